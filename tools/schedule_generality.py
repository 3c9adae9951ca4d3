#!/usr/bin/env python3
"""Is the SEED_PIXEL schedule (DESIGN 4.7 (HISTORY 4.9): chain-ordered tiles, two regions, arbiter priorities) and the chain hand-off
(DESIGN 4.7 (HISTORY 4.10)) tuned to ONE scene?  

Four workloads -- the bench scene and three others the tests already hold -- x {float64 brute force, tor_render()'s default
accelerations} x four settings of the scheduling machinery, every canvas compared with the first setting's:

    default        everything on
    sched off      TOR_HOT_FRAC=0 TOR_PRIO_SHIFT=0                             (no arbiter priorities)
    handoff off    TOR_MIGRATE=0                                               (round-2 behaviour: split mode / wave kernel)
    all off        both

    python tools/schedule_generality.py [--out gpurun_out/r3_schedule_generality.txt]
"""
import argparse
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

tor = importlib.import_module("trace-of-radiance_amd")
KNOBS = ("TOR_HOT_FRAC", "TOR_PRIO_SHIFT", "TOR_MIGRATE")
SETTINGS = [("default", {}), ("sched off", {"TOR_HOT_FRAC": "0", "TOR_PRIO_SHIFT": "0"}), ("handoff off", {"TOR_MIGRATE": "0"}),
            ("all off", {"TOR_HOT_FRAC": "0", "TOR_PRIO_SHIFT": "0", "TOR_MIGRATE": "0"})]


def multi_group_scene():
    """~300 objects, 3 time groups (one of general movers), hollow spheres, a big glass sphere (tests/test_gpu_parity.py)."""
    rng = np.random.default_rng(5)
    recs = [[0, 0, -1000, 0, 0, -1000, 0, 0, 1, 1000, 0, .5, .5, .5, 0, 0]]
    for i in range(300):
        x, z = rng.uniform(-9, 9, 2)
        kind, mat = i % 4, [0, 1, 2][i % 3]
        r = 0.25 if i % 17 else -0.25
        if kind == 0:
            recs.append([0, x, .25, z, x, .25, z, 0, 1, r, mat, .6, .5, .4, 0.2, 1.5])
        elif kind == 1:
            recs.append([1, x, .25, z, x, .25 + rng.uniform(0, .6), z, 0.0, 1.0, r, mat, .3, .7, .4, 0.1, 1.5])
        elif kind == 2:
            recs.append([1, x, .25, z, x + rng.uniform(-.5, .5), .4, z + rng.uniform(-.5, .5), 0.25, 0.75, r, mat, .3, .3, .8, 0.0, 1.4])
        else:
            recs.append([1, x, .25, z, x, .25, z + .3, -1.0, 2.0, r, mat, .8, .3, .3, 0.4, 1.3])
    recs.append([0, 0, 1, 0, 0, 1, 0, 0, 1, 1.0, 2, 0, 0, 0, 0, 1.5])
    return tor.Scene.from_records(np.asarray(recs, dtype=np.float64))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/r3_schedule_generality.txt")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--spp", type=int, default=100)
    args = ap.parse_args()
    H, W, spp = args.height, args.width, args.spp
    it = iter(tor.Animation(H, W, 0.005, 0.0, 7.2).scenes(6))
    for _ in range(38):
        acam, ascene, _t = next(it)
    workloads = [("random_scene (bench), camera outside", tor.random_scene(0xFACADE), tor.camera()),
                 ("animation frame 37, 1601 static spheres", ascene, acam),
                 ("300 objects, 3 time groups, hollow + big glass", multi_group_scene(), tor.camera(look_from=(10, 2.5, 4), aperture=0.05)),
                 ("random_scene, camera INSIDE the cloud", tor.random_scene(0xFACADE), tor.camera(look_from=(0.6, 0.7, 0.9), look_at=(4, 0.6, 0), vertical_field_of_view=60.0,
                                                                                                   aperture=0.02, focus_distance=3.0))]
    st = torch.cuda.current_stream().cuda_stream
    lines = [f"# tools/schedule_generality.py: TOR_SEED_PIXEL, {W}x{H}, {spp} spp, depth 50; ms per step (min of 3), Msamples/s; canvases vs `default`",
             "workload | accel | " + " | ".join(n for n, _ in SETTINGS) + " | worst regression of `default` vs the best other setting"]
    for wname, scene, cam in workloads:
        for accel in (0, 3):
            ref, cells, times = None, [], []
            for sname, env in SETTINGS:
                for k in KNOBS:
                    os.environ.pop(k, None)
                os.environ.update(env)
                ctx = tor.Context(0)
                for k in KNOBS:
                    os.environ.pop(k, None)
                ctx.upload(scene.list())
                buf = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
                o = tor.make_options(seeding=tor.SEED_PIXEL, accel=accel)
                ctx.render_device(cam, H, W, spp, 2.2, 50, o, buf.data_ptr(), st)
                torch.cuda.synchronize()
                ts = []
                for _ in range(3):
                    t = time.perf_counter()
                    ctx.render_device(cam, H, W, spp, 2.2, 50, o, buf.data_ptr(), st)
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t) * 1e3)
                if ref is None:
                    ref = buf.clone()
                same = bool(torch.equal(ref, buf))
                times.append(min(ts))
                cells.append(f"{min(ts):7.2f} ms {H * W * spp / min(ts) / 1e3:7.1f}{'' if same else ' CANVAS DIFFERS'}")
                ctx.close()
            loss = times[0] / min(times[1:]) - 1.0
            lines.append(f"{wname} | {accel} | " + " | ".join(cells) + f" | {loss * 100:+.1f} %")
            print(lines[-1], flush=True)
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out, "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
