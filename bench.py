#!/usr/bin/env python3
"""bench.py -- Msamples/s of the integrator hot path (render.nim:49-68) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one frame of the benchmark workload: BASELINE.json
configs[1], the reference's random_scene (seed 0xFACADE, 485 objects) at 1920x1080, 100 spp,
depth 50, rendered with the counter-based per-sample streams (TOR_SEED_SAMPLE) and the
reference's rounding (TOR_ARITH_STRICT).  Scene and camera are resident in HBM before the timed
region; the frame stays on the device.

N > 1 (one process per GPU, RCCL): image rows are dealt to the ranks round-robin (render.nim:55's
`parallelFor row` across GPUs), every rank renders its rows, then ONE all_gather of the
row shards (the framebuffer gather over xGMI) -- both inside the timed region.  Weak scaling:
samples per pixel grow with N (100*N), so every GPU traces the same number of samples as the
single-GPU run.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field meanings).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Algorithmic work per pixel-sample on the benchmark scene (BASELINE.md 4,
# re-measured by the oracle's counters: tests/golden/kat.json): 2.60 closest-hit queries x 485
# objects = 1262 ray/object tests x 32.8 float64 ops (reference formulation, nothing hoisted,
# nothing fused) + ~0.5 k for camera/scatter/sky.
FLOPS_PER_SAMPLE = 41.9e3
PEAK_FP64_VECTOR_TFLOPS = 78.6   # 256 CU x 4 SIMD x 16 lanes x 2 (FMA) x 2.4 GHz  (datasheet)
PEAK_FP64_NOFMA_TFLOPS = 39.3    # the same issue rate with add/mul only (the reference has no FMA)
PEAK_HBM_GBPS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=100, help="samples per pixel per GPU (x N ranks)")
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--seeding", choices=["sample", "pixel"], default="sample")
    ap.add_argument("--arith", choices=["strict", "fused"], default="strict")
    ap.add_argument("--accel", choices=["none", "blocks", "f32", "blocks+f32"], default="none",
                    help="none: the reference's brute-force closest hit (the metric's algorithm); blocks: exact block culling (SURVEY 8 f4)")
    ap.add_argument("--row-tile", type=int, default=1,
                    help="rows per shard tile; 1 = row-cyclic: every rank gets nrows/N rows (+-1) of statistically equal cost")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["c2", "c5"], default="c2",
                    help="c2: random_scene frame (default, the metric's config); c5: BASELINE configs[4], the animated "
                         "bouncing-spheres scene, 256 spp per frame, frames dealt round-robin to the GPUs (no collective)")
    ap.add_argument("--verify", action="store_true", help="N > 1: also render the whole frame on every rank and require the gathered frame to be identical")
    ap.add_argument("--no-accel-leg", action="store_true", help="skip the secondary TOR_ACCEL_BLOCKS measurement")
    ap.add_argument("--stats", action="store_true", help="also collect the kernel's workload counters (untimed extra step)")
    return ap.parse_args()


def measured_profile(W, H, spp, seeding, arith, accel="none"):
    """What the committed rocprofv3 PMC passes (profiles/traffic.json, rebuilt from the summaries by
    tools/update_traffic.py) measured for this configuration: HBM bytes per launch of the dominant kernel and
    its executed-instruction counters; {} otherwise."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        return t[f"{W}x{H}x{spp}:{seeding}:{arith}" + ("" if accel == "none" else ":" + accel)]
    except Exception:
        return {}


def measured_traffic(W, H, spp, seeding, arith, accel="none"):
    return measured_profile(W, H, spp, seeding, arith, accel).get("bytes")


def cpu_baseline(width, height, spp, depth, target_seconds):
    """The oracle in its reference-faithful mode (per-pixel streams, libm, no FMA), OpenMP over
    rows with schedule(dynamic,1) -- the analogue of Weave's parallelFor row (render.nim:55) --
    on all host cores, over every k-th row of the SAME frame so that it takes ~target_seconds."""
    from oracle import oracle as O
    objs, _ = O.random_scene(0xFACADE)
    cam = O.camera()
    cores = O.num_threads()
    # calibrate on ~one row per core, then size the strided sample for ~target_seconds
    step_cal = max(1, height // max(min(height, cores), 1))
    rows_cal = len(range(0, height, step_cal))
    t = time.perf_counter()
    O.render(height, width, spp, cam, objs, max_depth=depth, row_step=step_cal)
    rate = rows_cal * width * spp / max(time.perf_counter() - t, 1e-6)  # samples/s, rough
    want_rows = int(min(height, max(cores, rate * target_seconds / (width * spp))))
    step = max(1, height // max(want_rows, 1))
    rows = len(range(0, height, step))
    t = time.perf_counter()
    O.render(height, width, spp, cam, objs, max_depth=depth, row_step=step)
    dt = time.perf_counter() - t
    samples = rows * width * spp
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except Exception:
        pass
    return {
        "value": round(samples / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
        "sample": f"every {step}th row ({rows} of {height} rows) of the {width}x{height}x{spp}spp frame, "
                  f"depth {depth}: {samples / 1e6:.1f} Msamples in {dt:.1f} s; oracle/tor_oracle.c faithful mode "
                  f"(seed(row,col) streams, libm, -ffp-contract=off; bit-identical to the reference's golden PNG), "
                  f"OpenMP schedule(dynamic,1) over rows",
        "cpu_model": model,
    }


def bench_animation(args, tor, torch, dist, world, rank, local_rank):
    """BASELINE configs[4]: scenes_animated bouncing spheres, frame-parallel (frame f -> GPU f mod N,
    SURVEY 8e): a step = one frame per GPU: scene upload (1601 objects) + integrator.  No collective."""
    H, W = args.height, args.width
    spp = args.spp if args.spp != 100 else 256
    seeding = tor.SEED_SAMPLE if args.seeding == "sample" else tor.SEED_PIXEL
    arith = tor.ARITH_STRICT if args.arith == "strict" else tor.ARITH_FUSED
    n_steps = args.warmup + args.steps
    anim = tor.Animation(H, W, 0.005, 0.0, 7.2)   # 240 frames at skip 6
    frames = []
    for f, (cam, scene, t) in enumerate(anim.scenes(6)):
        if f >= n_steps * max(world, 1):
            break
        if f % max(world, 1) == (rank if world > 1 else 0):
            frames.append((cam, scene))
    ctx = tor.Context(local_rank % max(torch.cuda.device_count(), 1) if world > 1 else 0)
    opt = tor.make_options(seeding=seeding, arith=arith, accel=ACCEL_BITS[args.accel])
    buf = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def step(i):
        cam, scene = frames[i]
        ctx.upload(scene.list())
        ctx.render_device(cam, H, W, spp, 2.2, args.depth, opt, buf.data_ptr(), stream)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.warmup, n_steps):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    k_ms, k_n = ctx.kernel_ms_mean(args.steps)
    if rank == 0:
        total = H * W * spp * args.steps * max(world, 1)
        print(json.dumps({
            "metric": "Msamples/s (pixels x spp / s) on the animated bouncing-spheres scene", "value": round(total / elapsed / 1e6, 2),
            "unit": "Msamples/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64" if args.accel == "none" else "f64 (canvas bit-identical to the float64 path)", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[4]: scenes_animated (1601 static spheres per frame), {W}x{H}, {spp} spp, "
                                   f"depth {args.depth}, one frame per GPU per step", "seeding": args.seeding, "arith": args.arith,
                       "parallelism": f"frame f -> GPU f mod {max(world, 1)}, no collective"},
            "kernel_ms": round(k_ms, 3), "frames_per_s": round(args.steps * max(world, 1) / elapsed, 3)}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


ACCEL_BITS = {"none": 0, "blocks": 1, "f32": 2, "blocks+f32": 3}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = args.gpus
    # TOR_BENCH_BACKEND=gloo lets several ranks share ONE GPU (testing the N > 1 path on a 1-GPU box);
    # the real run is one rank per GPU over RCCL ("nccl")
    backend = os.environ.get("TOR_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=backend)
    else:
        torch.cuda.set_device(0)
        dev_index = 0
    tor = importlib.import_module("trace-of-radiance_amd")
    tor.ensure_built()   # a fresh checkout has no in-tree .so yet (file-locked: the ranks may race)

    H, W = args.height, args.width
    if args.workload == "c5":
        return bench_animation(args, tor, torch, dist, world, rank, local_rank)
    spp = args.spp * max(world, 1)  # weak scaling: per-GPU samples fixed
    seeding = tor.SEED_SAMPLE if args.seeding == "sample" else tor.SEED_PIXEL
    arith = tor.ARITH_STRICT if args.arith == "strict" else tor.ARITH_FUSED

    scene = tor.random_scene(0xFACADE)
    cam = tor.camera(aspect_ratio=W / H) if (W * 9 != H * 16) else tor.camera()
    ctx = tor.Context(dev_index)
    ctx.upload(scene.list())
    opt = tor.make_options(seeding=seeding, arith=arith, shard_index=rank if world > 1 else 0,
                           shard_count=max(world, 1), row_tile=args.row_tile,
                           accel=ACCEL_BITS[args.accel])
    tdist = importlib.import_module("trace-of-radiance_amd.distributed")
    plan = tdist.ShardPlan(H, args.row_tile, max(world, 1))
    frame = tdist.DistributedFrame(plan, W, rank if world > 1 else 0, torch.device("cuda"))
    my_rows = frame.my_rows
    assert list(my_rows) == list(tor.shard_rows(H, args.row_tile, rank if world > 1 else 0, max(world, 1)))
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        ctx.render_device(cam, H, W, spp, 2.2, args.depth, opt, frame.shard.data_ptr(), stream)
        if world > 1:
            frame.gather()   # RCCL all_gather of the row shards over xGMI + rows put in place

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if args.verify and world > 1:
        # the gathered frame must be the frame one process renders alone (any partition, any world size)
        full = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
        ctx.render_device(cam, H, W, spp, 2.2, args.depth,
                          tor.make_options(seeding=seeding, arith=arith, accel=opt.accel), full.data_ptr(), stream)
        torch.cuda.synchronize()
        same = bool(torch.equal(full, frame.frame))
        flag = torch.tensor([1 if same else 0], dtype=torch.int64, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        verified = bool(flag.item())
        if not verified:
            raise SystemExit(f"rank {rank}: gathered frame differs from the single-process frame")
    else:
        verified = None

    # dominant kernel: integrate_kernel, per-launch HIP events on the launch stream
    k_ms, k_n = ctx.kernel_ms_mean(args.steps)
    local_samples = len(my_rows) * W * spp
    total_samples = H * W * spp
    value = total_samples * args.steps / elapsed / 1e6

    result = None
    if rank == 0:
        k_rate = local_samples / (k_ms * 1e-3)          # samples/s inside the kernel, this rank
        tflops = k_rate * FLOPS_PER_SAMPLE / 1e12
        hbm_bytes = len(my_rows) * W * 24.0 * 2 + 64e3  # canvas write (+ atomics read-modify-write) + scene
        roof = {
            "bound": "valu_fp64", "kernel": "tor::integrate_kernel",
            "achieved": round(tflops, 3), "peak": PEAK_FP64_VECTOR_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tflops / PEAK_FP64_VECTOR_TFLOPS, 4),
            "frac_of_nofma_peak": round(tflops / PEAK_FP64_NOFMA_TFLOPS, 4),
            "nofma_peak": PEAK_FP64_NOFMA_TFLOPS,
            "flops_per_sample": FLOPS_PER_SAMPLE, "kernel_ms": round(k_ms, 3), "launches_averaged": k_n,
            "traffic": measured_traffic(W, H, spp, args.seeding, args.arith, args.accel) if world == 1 else None,
            "executed": measured_profile(W, H, spp, args.seeding, args.arith, args.accel).get("executed") if world == 1 else None,
            "hbm": {"bound": "hbm", "achieved": round(hbm_bytes / (k_ms * 1e-3) / 1e9, 4), "peak": PEAK_HBM_GBPS,
                    "unit": "GB/s", "frac": round(hbm_bytes / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 8),
                    "algorithmic_bytes_per_launch": hbm_bytes},
            "note": "the path is float64-VALU bound by design (scene in SGPRs via the scalar cache, 24 B of HBM "
                    "per pixel); the reference's arithmetic has no FMA, so 0.5 of the FMA peak is its ceiling",
        }
        if args.accel != "none":
            # with an exact acceleration the rate is still quoted against the reference's brute-force
            # float64 operation count, so frac can exceed 1; `executed` holds what the kernel really issued
            roof["note"] = ("--accel " + args.accel + ": achieved = samples/s x the reference's brute-force float64 "
                            "operation count (SURVEY 8d), not executed work -- the float32 pre-filter issues packed "
                            "float32 (peak 157.3 TFLOP/s) and the block culling skips tests; see `executed`")
            result_dtype = "f64 (canvas bit-identical to the float64 path)"
        else:
            result_dtype = "f64"
        result = {
            "metric": "Msamples/s (pixels\u00d7spp/s) on book-1 random_scene", "value": round(value, 2), "unit": "Msamples/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": result_dtype, "data": "synthetic",
            "config": {"workload": f"BASELINE configs[1]: random_scene seed 0xFACADE (485 objects), {W}x{H}, "
                                   f"{spp} spp ({args.spp} per GPU), depth {args.depth}",
                       "seeding": args.seeding, "arith": args.arith, "accel": args.accel,
                       "parallelism": f"row tiles of {args.row_tile} dealt to {max(world, 1)} rank(s)" +
                                      (" + RCCL all_gather of the framebuffer" if world > 1 else "")},
            "roofline": roof,
        }
        if verified is not None:
            result["gathered_frame_identical_to_single_process"] = verified
    if rank == 0 and world == 1 and args.accel == "none" and not args.no_accel_leg:
        # secondary legs (never the metric's value): the same frame with the exact accelerations -- bit-identical
        # canvas (tests/test_gpu_parity.py::test_block_culling_never_changes_a_pixel), checked here again
        ref_frame = frame.shard.clone()
        notes = {"f32": "every ray x every object, through the conservative packed-float32 pre-filter first "
                        "(tor_filter32.hpp); kept objects get the reference's float64 test",
                 "blocks": "SURVEY 8 f4: spatial blocks of 8 objects behind conservative boxes",
                 "blocks+f32": "both"}
        for name in ("f32", "blocks", "blocks+f32"):
            opt2 = tor.make_options(seeding=seeding, arith=arith, row_tile=args.row_tile, accel=ACCEL_BITS[name])
            ctx.render_device(cam, H, W, spp, 2.2, args.depth, opt2, frame.shard.data_ptr(), stream)
            torch.cuda.synchronize()
            same = bool(torch.equal(ref_frame, frame.shard))
            t1 = time.perf_counter()
            for _ in range(args.steps):
                ctx.render_device(cam, H, W, spp, 2.2, args.depth, opt2, frame.shard.data_ptr(), stream)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            prof2 = measured_profile(W, H, spp, args.seeding, args.arith, name)
            result["accel_" + name.replace("+", "_")] = {
                "value": round(total_samples * args.steps / dt2 / 1e6, 2), "unit": "Msamples/s",
                "traffic": prof2.get("bytes"),
                "valu_active_frac": (prof2.get("executed") or {}).get("valu_active_frac"),
                "ms_per_step": round(dt2 / args.steps * 1e3, 3), "canvas_identical_to_brute_force": same,
                "note": notes[name] + "; the metric's value above is the reference's float64 brute-force closest hit"}
    if args.stats and rank == 0:
        ctx.set_stats(True)
        step()
        torch.cuda.synchronize()
        st = ctx.last_stats()
        result["kernel_stats"] = {
            "hit_queries_per_sample": round(st.hit_queries / max(st.samples, 1), 4),
            "candidates_per_query": round(st.candidates / max(st.hit_queries, 1), 3),
            "lane_utilisation": round(st.hit_queries / max(st.lane_slots, 1), 4),
            "samples": int(st.samples),
        }
        ctx.set_stats(False)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(W, H, args.spp, args.depth, args.cpu_seconds)
    elif rank == 0:
        result["cpu_baseline"] = None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
