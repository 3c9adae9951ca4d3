"""The reference's default configuration (C1: 384x216, 100 spp, depth 50) through the host-canvas entry point
(tor_render_opt: scene upload + kernel + D2H), every seeding x accel mode."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tor = importlib.import_module("trace-of-radiance_amd")
scene, cam = tor.random_scene(0xFACADE), tor.camera()
for seeding, sn in ((tor.SEED_PIXEL, "pixel"), (tor.SEED_SAMPLE, "sample")):
    for accel, an in ((0, "float64 brute force"), (2, "f32"), (1, "blocks"), (3, "blocks+f32")):
        cv = tor.new_canvas(216, 384, 100, 2.2)
        opt = tor.make_options(seeding=seeding, accel=accel)
        tor.render(cv, cam, scene.list(), 50, opt)
        ts = []
        for _ in range(3):
            t = time.perf_counter()
            tor.render(cv, cam, scene.list(), 50, opt)
            ts.append(time.perf_counter() - t)
        print(f"C1 384x216x100 host canvas in/out, {sn:6s} {an:20s}: {min(ts) * 1e3:6.1f} ms  ({216 * 384 * 100 / min(ts) / 1e6:5.0f} Msamples/s)")
