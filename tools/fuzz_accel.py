"""Differential fuzz of the exact accelerations on the GPU: random scenes and cameras, every accel mode -- and the brute force
WITHOUT its conservative FMA screen (TOR_SCREEN=0, DESIGN 4.2: no plane screen, no second form, the reference's unfused
discriminant for every object) and the brute force with stage one of the screen FORCED onto every segment that carries a plane
table (TOR_PLANE=2: no gate) -- against the float64 brute-force canvas (default: the host's gate and the waves' votes decide per
segment), bit for bit.
Usage: python tools/fuzz_accel.py [seconds] [seed] [heights] [--oracle K] [--big K]
    heights: mixed (default: half of the scenes rest their spheres on 1-4 common heights; 15 % are dense, see below) | none (every sphere
             at its own height: no common-height segment anywhere -- the round-5 record) | dense (every scene: 1-3 clusters of nearly
             coincident spheres, the worst case of the pooled lists and of every candidate queue)
    --oracle K (round 6): every K-th scene is ALSO rendered by the CPU oracle (oracle/tor_oracle.c: the restatement of the reference,
             PORTABLE math, the accumulation of the stream mode) in both stream modes and the GPU brute force must equal it bit for bit --
             a GPU-vs-GPU diff cannot see a defect in what every loop family shares (layout, padding, segment order); the oracle
             walks the caller's list in the caller's order like hittables_lists.nim:48-55.  (The oracle is the checker here, as in tests/.)"""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tor = importlib.import_module("trace-of-radiance_amd")


HEIGHTS = "mixed"


def random_scene(rng):
    n = int(rng.choice([40, 90, 200, 485, 700, 1300]))
    spread = float(rng.choice([0.02, 1.0, 4.0, 12.0, 60.0, 3000.0]))
    shift = rng.uniform(-1, 1, 3) * float(rng.choice([0.0, 10.0, 1e3, 1e5]))
    rscale = spread / 12.0
    groups = [(0.0, 1.0), (-0.5, 0.5), (0.25, 2.0), (1.0, 0.0), (0.5, 0.5)][: int(rng.integers(1, 6))]   # (the last: time0 == time1, never a finite fraction)
    recs = []
    if rng.random() < 0.6:
        R = float(rng.choice([100.0, 1000.0])) * rscale
        recs.append([0, shift[0], shift[1] - R, shift[2], shift[0], shift[1] - R, shift[2], 0, 1, R, 0, .5, .5, .5, 0, 0])
    mover_frac = float(rng.choice([0.0, 0.5, 0.9]))
    general = rng.random() < 0.4
    per_object = rng.random() < 0.3     # movers along y and movers in general position in ONE scene (and in one time group)
    # spheres resting at a few common heights (bit-identical c0.y: the strict loop's common-height segments and the plane screen in
    # front of them) in half of the scenes; any height in the others
    levels = shift[1] + rng.uniform(0, 0.3 * spread, int(rng.integers(1, 5))) if (HEIGHTS == "mixed" and rng.random() < 0.5) else None
    # dense scenes (round 6, after the pooled survivor list of the cooperative resolve was found to overflow on them): 1-3 clusters of
    # nearly coincident spheres -- a ray through a cluster keeps EVERY object of every block it enters, and every larger sphere
    dense = HEIGHTS == "dense" or (HEIGHTS == "mixed" and rng.random() < 0.15)
    centres = shift + rng.uniform(-0.3, 0.3, (int(rng.integers(1, 4)), 3)) * spread * (1, 0.3, 1) + (0, 0.1 * spread, 0)
    tight = float(rng.choice([0.002, 0.02, 0.1])) * rscale
    while len(recs) < n:
        c = shift + np.array([rng.uniform(-spread, spread), rng.uniform(0, 0.3 * spread), rng.uniform(-spread, spread)])
        if dense and rng.random() < 0.9:
            c = centres[int(rng.integers(0, len(centres)))] + rng.uniform(-tight, tight, 3)
        if levels is not None and rng.random() < 0.9:
            c[1] = levels[int(rng.integers(0, len(levels)))]
        r = float(rng.choice([0.15, 0.2, 0.3, 0.45, 1.0])) * rscale * (1 if rng.random() > 0.03 else -1)
        mat = int(rng.integers(0, 3))
        alb = rng.uniform(0.1, 0.95, 3)
        fuzz, ri = rng.uniform(0, 0.6), rng.uniform(1.2, 1.8)
        if rng.random() >= mover_frac:
            recs.append([0, *c, *c, 0, 1, r, mat, *alb, fuzz, ri])
        else:
            t0, t1 = groups[int(rng.integers(0, len(groups)))]
            g = (rng.random() < 0.5) if per_object else general
            d = rng.uniform(-0.6, 0.6, 3) * rscale if g else np.array([0.0, rng.uniform(0, 0.6) * rscale, 0.0])
            recs.append([1, *c, *(c + d), t0, t1, r, mat, *alb, fuzz, ri])
    recs = np.asarray(recs, dtype=np.float64)
    inside = rng.random() < 0.3
    look_from = shift + (rng.uniform(-0.5, 0.5, 3) * spread + (0, 0.2 * spread, 0) if inside
                         else np.array([1.8, 0.7, 1.1]) * spread * rng.uniform(0.6, 2.0))
    shutter = [(0.0, 1.0), (-1.0, 3.0), (0.5, 0.5), (2.0, 1.0), (0.0, 0.0)][int(rng.integers(0, 5))]
    cam = tor.camera(look_from=tuple(look_from), look_at=tuple(shift + (0, 0.1 * spread, 0)),
                     vertical_field_of_view=float(rng.uniform(15, 80)), aperture=float(rng.uniform(0, 0.3)) * rscale,
                     focus_distance=float(rng.uniform(0.5, 2) * spread), shutter_open=shutter[0], shutter_close=shutter[1])
    return recs, cam


def main():
    argv = list(sys.argv[1:])
    big_every = 0
    if "--big" in argv:   # every K-th scene once more at 108 x 192 x 40 spp: hundreds of tiles, every wave of the machine, dedicated servers
        k = argv.index("--big")
        big_every = int(argv[k + 1])
        del argv[k:k + 2]
    n_big = 0
    oracle_every = 0
    if "--oracle" in argv:
        k = argv.index("--oracle")
        oracle_every = int(argv[k + 1])
        del argv[k:k + 2]
    budget = float(argv[0]) if len(argv) > 0 else 120.0
    rng = np.random.default_rng(int(argv[1]) if len(argv) > 1 else 12345)
    global HEIGHTS
    HEIGHTS = argv[2] if len(argv) > 2 else "mixed"
    O = None
    if oracle_every > 0:
        from oracle import oracle as O   # the checker (test infrastructure), never the product
        O.build()
    n_oracle = 0
    os.environ["TOR_SCREEN"] = "0"
    unscreened = tor.Context(0)      # (the knob is read when a context is made; tor.render() below uses the default context)
    os.environ.pop("TOR_SCREEN", None)
    os.environ["TOR_PLANE"] = "2"
    forced = tor.Context(0)
    os.environ.pop("TOR_PLANE", None)
    t0 = time.time()
    n_scenes = n_renders = bad = 0
    while time.time() - t0 < budget:
        recs, cam = random_scene(rng)
        scene = tor.Scene.from_records(recs)
        h, w = int(rng.choice([16, 24, 40])), int(rng.choice([26, 34, 64]))
        spp, depth = int(rng.choice([2, 6, 33, 40])), int(rng.choice([3, 12, 50]))
        for seeding in (0, 1):
            canv = []
            for accel in (0, 1, 2, 3):
                cv = tor.new_canvas(h, w, spp, 2.2)
                # (pixel_kernel=1: the lane kernel -- the wave-per-pixel kernel small frames would otherwise get ignores accel)
                tor.render(cv, cam, scene.list(), depth, tor.make_options(seeding=seeding, accel=accel, pixel_kernel=1))
                canv.append(cv.pixels.copy())
                n_renders += 1
            if O is not None and n_scenes % oracle_every == 0:
                # the arbiter: the reference's algorithm restated on the CPU, on the same list, camera and streams
                ocam = np.frombuffer(bytes(cam), dtype=np.float64).copy()   # TorCamera = 24 float64 in cameras.nim's field order
                want = O.render(h, w, spp, ocam, recs, max_depth=depth, seeding=seeding, math=O.MATH_PORTABLE, accum=seeding).pixels
                n_oracle += 1
                if not np.array_equal(canv[0], want, equal_nan=True):
                    bad += 1
                    print(f"MISMATCH scene {n_scenes} (n={len(recs)}) seeding {seeding} GPU brute force vs ORACLE: "
                          f"{int((canv[0] != want).sum())} values differ, max {np.nanmax(np.abs(canv[0] - want)):.3e}", flush=True)
            # the same brute force without the FMA screen: every object through the reference's unfused discriminant
            unscreened.upload(scene.list())
            buf = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda")
            unscreened.render_device(cam, h, w, spp, 2.2, depth, tor.make_options(seeding=seeding, accel=0, pixel_kernel=1), buf.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            plain = buf.cpu().numpy()
            n_renders += 1
            if not np.array_equal(canv[0], plain, equal_nan=True):
                bad += 1
                print(f"MISMATCH scene {n_scenes} (n={len(recs)}) seeding {seeding} TOR_SCREEN=0: {int((canv[0] != plain).sum())} values differ", flush=True)
            forced.upload(scene.list())
            forced.render_device(cam, h, w, spp, 2.2, depth, tor.make_options(seeding=seeding, accel=0, pixel_kernel=1), buf.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            n_renders += 1
            if not np.array_equal(canv[0], buf.cpu().numpy(), equal_nan=True):
                bad += 1
                print(f"MISMATCH scene {n_scenes} (n={len(recs)}) seeding {seeding} TOR_PLANE=2: {int((canv[0] != buf.cpu().numpy()).sum())} values differ", flush=True)
            if seeding == 0:  # the wave-per-pixel kernel (TorOptions.pixel_kernel = 2) against the lane kernel's brute force
                cv = tor.new_canvas(h, w, spp, 2.2)
                tor.render(cv, cam, scene.list(), depth, tor.make_options(seeding=0, accel=0, pixel_kernel=2))
                n_renders += 1
                if not np.array_equal(canv[0], cv.pixels, equal_nan=True):
                    bad += 1
                    print(f"MISMATCH scene {n_scenes} (n={len(recs)}) wave-per-pixel kernel: {int((canv[0] != cv.pixels).sum())} values differ", flush=True)
            if seeding == 0:
                # what tor_render() runs when nothing is set (round 6): the library picks the pixel kernel itself (TorOptions.pixel_kernel = 0)
                # -- the chain hand-off with its server waves (>= 32 spp, single-level culling layout), split mode, or one wave per pixel
                for accel in (3, 0):
                    cv = tor.new_canvas(h, w, spp, 2.2)
                    tor.render(cv, cam, scene.list(), depth, tor.make_options(seeding=0, accel=accel))
                    n_renders += 1
                    if not np.array_equal(canv[0], cv.pixels, equal_nan=True):
                        bad += 1
                        print(f"MISMATCH scene {n_scenes} (n={len(recs)}) library's own choice of pixel kernel, accel {accel}: "
                              f"{int((canv[0] != cv.pixels).sum())} values differ; note: {tor.last_note()!r}", flush=True)
            for accel in (1, 2, 3):
                if not np.array_equal(canv[0], canv[accel], equal_nan=True):
                    bad += 1
                    print(f"MISMATCH scene {n_scenes} (n={len(recs)}) seeding {seeding} accel {accel}: "
                          f"{int((canv[0] != canv[accel]).sum())} values differ", flush=True)
        if big_every > 0 and n_scenes % big_every == 0:
            # a frame large enough for the SEED_PIXEL schedule to matter: the cost probe, the tile order, dedicated server waves,
            # hand-offs from waves all over the machine -- the library's own choice of kernel against the lane kernel's brute force
            bh, bw, bspp = 108, 192, 40
            ref = tor.new_canvas(bh, bw, bspp, 2.2)
            tor.render(ref, cam, scene.list(), depth, tor.make_options(seeding=0, accel=0, pixel_kernel=1))
            n_renders += 1
            n_big += 1
            for accel, pk in ((3, 0), (0, 0), (3, 1)):
                cv = tor.new_canvas(bh, bw, bspp, 2.2)
                tor.render(cv, cam, scene.list(), depth, tor.make_options(seeding=0, accel=accel, pixel_kernel=pk))
                n_renders += 1
                if not np.array_equal(ref.pixels, cv.pixels, equal_nan=True):
                    bad += 1
                    print(f"MISMATCH scene {n_scenes} (n={len(recs)}) {bh}x{bw}x{bspp} accel {accel} pixel_kernel {pk}: "
                          f"{int((ref.pixels != cv.pixels).sum())} values differ; note: {tor.last_note()!r}", flush=True)
        n_scenes += 1
    if big_every > 0:
        print(f"  ({n_big} scenes also at 108x192x40 through the library's own choice of pixel kernel)", flush=True)
    print(f"fuzz (heights: {HEIGHTS}): {n_scenes} scenes, {n_renders} renders, {n_oracle} of them checked against the CPU oracle "
          f"(every {oracle_every or '-'}th scene, both stream modes), {bad} mismatches in {time.time() - t0:.0f} s", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
