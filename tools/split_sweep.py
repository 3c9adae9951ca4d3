"""TOR_SEED_PIXEL split mode: time per frame for several split fractions and frame sizes (one process per setting:
the knobs are read when the context is created)."""
import importlib, os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import time
    import numpy as np
    tor = importlib.import_module("trace-of-radiance_amd")
    scene, cam = tor.random_scene(0xFACADE), tor.camera()
    out = []
    for a in sys.argv[2:]:
        h, w, spp = (int(x) for x in a.split("x"))
        ref = tor.new_canvas(h, w, spp, 2.2)
        tor.render(ref, cam, scene.list(), 50, tor.make_options(pixel_kernel=1, accel=3))
        cv = tor.new_canvas(h, w, spp, 2.2)
        tor.render(cv, cam, scene.list(), 50)
        t = time.perf_counter()
        for _ in range(3):
            tor.render(cv, cam, scene.list(), 50)
        dt = (time.perf_counter() - t) / 3
        out.append(f"{w}x{h}x{spp}: {dt * 1e3:7.2f} ms {h * w * spp / dt / 1e6:7.1f} Ms/s same={np.array_equal(ref.pixels, cv.pixels)}")
    print(" | ".join(out), flush=True)
else:
    sizes = sys.argv[1:] or ["216x384x100", "288x512x100", "432x768x100", "720x1280x100"]
    for frac in (os.environ.get("FRACS", "0,-1,0.3,0.45").split(",")):
        env = dict(os.environ, TOR_SPLIT_FRAC=frac)
        r = subprocess.run([sys.executable, __file__, "child"] + sizes, capture_output=True, text=True, env=env, timeout=300)
        print(f"frac {frac}: " + (r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]), flush=True)
