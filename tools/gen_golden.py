"""Generates tests/golden/kat.json and tests/golden/small_canvases.npz from the CPU oracle.

Run only after tests/test_oracle_golden.py::test_oracle_reproduces_reference_image_bit_exact
passes: the oracle is then pinned to the reference's own image (media/book2_motion_blur.png),
and these files record its answers for the modes and intermediate values that image cannot
reach (per-sample seeding, portable math, fused arithmetic, RNG states, scene hash)."""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def main():
    O.build(force=True)
    L = O.lib()
    st = (C.c_uint64 * 4)()
    rng_cases = []
    for mode, args in [(1, [0xFACADE]), (1, [0]), (2, [0, 0]), (2, [1, 2]), (2, [215, 383]),
                       (2, [1079, 1919]), (3, [0, 0, 0]), (3, [1, 2, 3]), (3, [2159, 3839, 4095])]:
        if mode == 1:
            L.oracle_rng_seed1(args[0], st)
        elif mode == 2:
            L.oracle_rng_seed2(args[0], args[1], st)
        else:
            L.oracle_rng_seed3(*args, st)
        state = [hex(int(v)) for v in st]
        nxt = [hex(int(L.oracle_rng_next(st))) for _ in range(4)]
        rng_cases.append({"mode": mode, "args": args, "state": state, "next": nxt})
    objs, draws = O.random_scene(0xFACADE)
    cam = O.camera()
    r = O.render(216, 384, 100, cam, objs, collect_stats=True)
    s = r.stats
    kat = {
        "rng": rng_cases,
        "scene_sha256": hashlib.sha256(objs.tobytes()).hexdigest(),
        "scene_draws": draws,
        "camera": cam.tolist(),
        "c1_stats": {"hit_calls": int(s.hit_calls), "rng_draws": int(s.rng_draws),
                     "depth_exhausted": int(s.depth_exhausted), "absorbed": int(s.absorbed),
                     "scatter": [int(s.scatter_lambertian), int(s.scatter_metal), int(s.scatter_dielectric)],
                     "depth_hist": [int(v) for v in s.depth_hist]},
        "c1_canvas_sha256": hashlib.sha256(r.pixels.tobytes()).hexdigest(),
    }
    with open(os.path.join(G, "kat.json"), "w") as f:
        json.dump(kat, f, indent=1)
    canv = {}
    for seeding in (0, 1):
        for math in (0, 1):
            for arith in (0, 1):
                for accum in ((0,) if seeding == 0 else (0, 1)):
                    rr = O.render(36, 64, 16, cam, objs, seeding=seeding, math=math, arith=arith, accum=accum)
                    canv[f"c_{seeding}_{math}_{arith}_{accum}"] = rr.pixels
    np.savez_compressed(os.path.join(G, "small_canvases.npz"), **canv)
    print("wrote", G)


if __name__ == "__main__":
    main()
