#!/usr/bin/env python3
"""Is the float64 brute force (the metric's kernel, `integrate_kernel<1, 2, 3, 0, 0>`) fitted to random_scene's geometry?
(the plane screen ran only on common-height segments with y-only movers.)

Workloads -- the bench scene and scenes that share none of its coincidences -- x {default, TOR_PLANE=0 (second form / first form
for every object), TOR_SCREEN=0 (the reference's unfused loop)}, per-sample streams, accel 0.  Every canvas is compared with the
unscreened one and the resolve pass's candidate count with TOR_PLANE=0's (stage one must never change what stage two leaves).

    python tools/brute_generality.py [--out gpurun_out/r6_brute_generality.txt] [--spp 100]
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
from tools.schedule_generality import multi_group_scene  # noqa: E402

SETTINGS = [("default", {}), ("TOR_PLANE=0", {"TOR_PLANE": "0"}), ("TOR_SCREEN=0", {"TOR_SCREEN": "0"})]
KNOBS = ("TOR_PLANE", "TOR_SCREEN")


def cloud_scene(n=600, seed=11, movers=True):
    """A 3-D cloud: no common height, no common radius; a third static, a third moving along y, a third moving anywhere."""
    rng = np.random.default_rng(seed)
    recs = [[0, 0, -1000, 0, 0, -1000, 0, 0, 1, 1000, 0, .5, .5, .5, 0, 0]]
    for i in range(n):
        x, z = rng.uniform(-8, 8, 2)
        y = rng.uniform(0.2, 6.0)
        r = rng.uniform(0.12, 0.3)
        mat = i % 3
        k = i % 3 if movers else 0
        if k == 0:
            recs.append([0, x, y, z, x, y, z, 0, 1, r, mat, .6, .5, .4, 0.2, 1.5])
        elif k == 1:
            recs.append([1, x, y, z, x, y + rng.uniform(0, .5), z, 0.0, 1.0, r, mat, .3, .7, .4, 0.1, 1.5])
        else:
            recs.append([1, x, y, z, x + rng.uniform(-.4, .4), y + rng.uniform(-.3, .3), z + rng.uniform(-.4, .4), 0.0, 1.0, r, mat, .3, .3, .8, 0.0, 1.4])
    return tor.Scene.from_records(np.asarray(recs, dtype=np.float64))


def wall_scene(n=24):
    """A vertical wall of spheres in the plane z = 0 seen along x: every ground track crosses every column (the plane family's worst case)."""
    recs = [[0, 0, -1000, 0, 0, -1000, 0, 0, 1, 1000, 0, .5, .5, .5, 0, 0]]
    for i in range(n):
        for j in range(n):
            recs.append([0, 0.5 * i - 6, 0.25 + 0.5 * j, 0.0, 0, 0, 0, 0, 1, 0.22, (i + j) % 3, .6, .5, .4, 0.1, 1.5])
    return tor.Scene.from_records(np.asarray(recs, dtype=np.float64))


def workloads(H, W):
    it = iter(tor.Animation(H, W, 0.005, 0.0, 7.2).scenes(6))
    for _ in range(38):
        acam, ascene, _t = next(it)
    return [("random_scene (bench)", tor.random_scene(0xFACADE), tor.camera()),
            ("animation frame 37, 1601 static spheres", ascene, acam),
            ("300 objects, 3 time groups, hollow + big glass", multi_group_scene(), tor.camera(look_from=(10, 2.5, 4), aperture=0.05)),
            ("3-D cloud, 600 objects, static / y movers / general movers", cloud_scene(), tor.camera(look_from=(14, 5, 6), look_at=(0, 3, 0), aperture=0.05)),
            ("3-D cloud, 600 static", cloud_scene(movers=False), tor.camera(look_from=(14, 5, 6), look_at=(0, 3, 0), aperture=0.05)),
            ("wall of 576 spheres in z = 0 seen along x", wall_scene(), tor.camera(look_from=(16, 6, 0.3), look_at=(0, 6, 0), aperture=0.02)),
            ("random_scene, camera INSIDE the cloud", tor.random_scene(0xFACADE),
             tor.camera(look_from=(0.6, 0.7, 0.9), look_at=(4, 0.6, 0), vertical_field_of_view=60.0, aperture=0.02, focus_distance=3.0))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/r6_brute_generality.txt")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--spp", type=int, default=100)
    args = ap.parse_args()
    H, W, spp = args.height, args.width, args.spp
    st = torch.cuda.current_stream().cuda_stream
    lines = [f"# tools/brute_generality.py: TOR_SEED_SAMPLE, float64 brute force (accel 0), {W}x{H}, {spp} spp, depth 50; ms per step (min of 3), "
             "Msamples/s, candidates per closest-hit query; canvases vs TOR_SCREEN=0",
             "workload | objects | " + " | ".join(n for n, _ in SETTINGS) + " | default / TOR_PLANE=0 | default / TOR_SCREEN=0"]
    for wname, scene, cam in workloads(H, W):
        cells, times, canv, cands = [], [], [], []
        for sname, env in SETTINGS:
            for k in KNOBS:
                os.environ.pop(k, None)
            os.environ.update(env)
            ctx = tor.Context(0)
            for k in KNOBS:
                os.environ.pop(k, None)
            ctx.upload(scene.list())
            buf = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
            o = tor.make_options(seeding=tor.SEED_SAMPLE, accel=0)
            ctx.set_stats(True)
            ctx.render_device(cam, H, W, spp, 2.2, 50, o, buf.data_ptr(), st)
            torch.cuda.synchronize()
            s = ctx.last_stats()
            cands.append((int(s.candidates), int(s.hit_queries)))
            ctx.set_stats(False)
            ts = []
            for _ in range(3):
                t = time.perf_counter()
                ctx.render_device(cam, H, W, spp, 2.2, 50, o, buf.data_ptr(), st)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t) * 1e3)
            canv.append(buf.clone())
            times.append(min(ts))
            ctx.close()
        same = all(bool(torch.equal(canv[-1], c)) for c in canv)
        same_cand = cands[0] == cands[1]
        for (sname, _), t, (c, q) in zip(SETTINGS, times, cands):
            cells.append(f"{t:7.2f} ms {H * W * spp / t / 1e3:7.1f} {c / max(q, 1):5.2f} c/q")
        lines.append(f"{wname} | {len(scene)} | " + " | ".join(cells) + f" | x{times[1] / times[0]:.2f} | x{times[2] / times[0]:.2f}"
                     + ("" if same else " | CANVAS DIFFERS") + ("" if same_cand else f" | CANDIDATES DIFFER {cands[0]} vs {cands[1]}"))
        print(lines[-1], flush=True)
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out, "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
