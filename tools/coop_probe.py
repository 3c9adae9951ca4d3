import importlib, os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
tor = importlib.import_module("trace-of-radiance_amd")
scene, cam = tor.random_scene(0xFACADE), tor.camera()
for (h, w, spp, depth) in ((4, 4, 1, 1), (4, 4, 1, 3), (8, 8, 2, 50), (36, 64, 16, 50)):
    for k in (1, 2):
        cv = tor.new_canvas(h, w, spp, 2.2)
        print("render", h, w, spp, depth, "kernel", k, flush=True)
        t = time.perf_counter()
        tor.render(cv, cam, scene.list(), depth, tor.make_options(pixel_kernel=k))
        print("   done %.1f ms  sum %.6f" % ((time.perf_counter() - t) * 1e3, cv.pixels.sum()), flush=True)
