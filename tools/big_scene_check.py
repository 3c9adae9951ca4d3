import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tor = importlib.import_module("trace-of-radiance_amd")
rng = np.random.default_rng(3)
for n, spread in ((20000, 60.0), (100000, 140.0)):
    recs = [[0, 0, -1000, 0, 0, -1000, 0, 0, 1, 1000, 0, .5, .5, .5, 0, 0]]
    xs = rng.uniform(-spread, spread, (n, 3)); xs[:, 1] = rng.uniform(0.2, 3.0, n)
    for i in range(n):
        x, y, z = xs[i]
        if i % 3:
            recs.append([0, x, y, z, x, y, z, 0, 1, 0.2, i % 3, .6, .5, .4, 0.1, 1.5])
        else:
            recs.append([1, x, y, z, x, y + 0.4, z, 0.0, 1.0, 0.2, 0, .3, .7, .4, 0.0, 1.5])
    recs = np.asarray(recs, dtype=np.float64)
    scene = tor.Scene.from_records(recs)
    cam = tor.camera(look_from=(spread * 1.2, spread * 0.3, spread * 0.5), look_at=(0, 1, 0), vertical_field_of_view=30.0,
                     aperture=0.05, focus_distance=spread)
    out = {}
    for accel in (0, 2, 1, 3):
        cv = tor.new_canvas(54, 96, 8, 2.2)
        t0 = time.perf_counter()
        tor.render(cv, cam, scene.list(), 20, tor.make_options(seeding=1, accel=accel))
        out[accel] = (cv.pixels.copy(), time.perf_counter() - t0)
    same = all(np.array_equal(out[0][0], out[a][0]) for a in (1, 2, 3))
    print(f"{n} objects: identical={same}  seconds: " + " ".join(f"a{a}:{out[a][1]:.2f}" for a in (0, 2, 1, 3)), flush=True)
