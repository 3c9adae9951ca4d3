"""C1 (384x216x100, the reference's own main()) through tor_render(): ms and Msamples/s per pixel kernel / accel."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
tor = importlib.import_module("trace-of-radiance_amd")
scene, cam = tor.random_scene(0xFACADE), tor.camera()
sizes = [(216, 384, 100)] + [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for (h, w, spp) in sizes:
    ref = None
    for name, opt in (("lane brute", dict(pixel_kernel=1, accel=0)), ("lane accel3", dict(pixel_kernel=1, accel=3)),
                      ("wave", dict(pixel_kernel=2)), ("auto (tor_render)", None)):
        cv = tor.new_canvas(h, w, spp, 2.2)
        o = tor.make_options(**opt) if opt else None
        tor.render(cv, cam, scene.list(), 50, o)
        t = time.perf_counter()
        for _ in range(3):
            tor.render(cv, cam, scene.list(), 50, o)
        dt = (time.perf_counter() - t) / 3
        if ref is None:
            ref = cv.pixels.copy()
        print(f"{w}x{h}x{spp} {name:18s} {dt * 1e3:8.2f} ms  {h * w * spp / dt / 1e6:8.1f} Msamples/s  same={np.array_equal(ref, cv.pixels)}", flush=True)
