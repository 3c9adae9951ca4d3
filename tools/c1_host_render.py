import importlib, sys, time
sys.path.insert(0, '.')
import numpy as np
tor = importlib.import_module("trace-of-radiance_amd")
scene, cam = tor.random_scene(0xFACADE), tor.camera()
ref = None
for seeding, sn in ((tor.SEED_PIXEL, "pixel"), (tor.SEED_SAMPLE, "sample")):
    for accel, an in ((tor.ACCEL_NONE, "brute"), (tor.ACCEL_BLOCKS, "blocks")):
        cv = tor.new_canvas(216, 384, 100, 2.2)
        opt = tor.make_options(seeding=seeding, accel=accel)
        tor.render(cv, cam, scene.list(), 50, opt)
        ts = []
        for _ in range(3):
            t = time.perf_counter(); tor.render(cv, cam, scene.list(), 50, opt); ts.append(time.perf_counter() - t)
        print(f"C1 384x216x100 host canvas in/out, {sn} {an}: {min(ts)*1e3:.1f} ms  ({216*384*100/min(ts)/1e6:.0f} Msamples/s)")
