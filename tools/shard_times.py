#!/usr/bin/env python3
"""Predict multi-GPU STRONG scaling of the row-sharded render (render.nim:55 `parallelFor row` across GPUs) on ONE GPU.

For N in {1, 2, 4, 8}: shard k of N (row tiles dealt round-robin, tor_shard_rows) is rendered ALONE through
tor_render_device -- exactly what rank k of an N-GPU job runs before the framebuffer gather -- and timed.  An N-GPU
step cannot be faster than its slowest shard, so

    predicted speed-up(N) = T(1) / max_k T_k(N)          (gather excluded: 6.2 MB per rank at 1080p, 24.9 MB at 4K)

Modes: (a) SAMPLE float64 brute force (the metric's kernel), (b) SAMPLE with both exact accelerations,
(c) PIXEL with both exact accelerations = what tor_render() -- the drop-in a Nim caller gets -- runs by default.
Configs: BASELINE configs[2] (1920x1080, FIXED 1000 spp) and configs[3] (3840x2160x4096).

    python tools/shard_times.py [--configs c3,c4] [--modes a,b,c] [--shards 1,2,4,8] [--out profiles/r3_shard_times.json]

BASELINE configs[4] (the 240-frame animation, frame-parallel) needs no GPU here: its per-frame times are in the line of
`bench.py --workload c5 --frames 240` (`frames.table`), and a GPU's share of an N-GPU job is the sum of its frames:

    python tools/shard_times.py --c5-from profiles/r6_bench_c5_240.json [--out profiles/r6_shard_times.json]   (appends a "c5" block)

Dealing rules compared: `f mod N` (scenes_animated's natural order dealt round-robin: SURVEY 8e), contiguous blocks of 240 / N frames,
and LPT on the measured costs (the bound a cost-aware dealer could reach).
"""
import argparse
import importlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--c5-from" not in sys.argv:
    import torch  # noqa: E402
tor = importlib.import_module("trace-of-radiance_amd")

CONFIGS = {"c1k": (216, 384, 1000, "384x216 x 1000 spp (the reference's main() size at configs[2]'s spp)"),
           "c2": (1080, 1920, 100, "BASELINE configs[1] 1920x1080x100"),
           "c3": (1080, 1920, 1000, "BASELINE configs[2] 1920x1080x1000 (fixed spp: strong scaling)"),
           "c4": (2160, 3840, 4096, "BASELINE configs[3] 3840x2160x4096")}
MODES = {"a": ("SAMPLE float64 brute force", dict(seeding=tor.SEED_SAMPLE, accel=0)),
         "b": ("SAMPLE, TOR_ACCEL_BLOCKS|TOR_ACCEL_F32", dict(seeding=tor.SEED_SAMPLE, accel=3)),
         "c": ("PIXEL (render.nim:59-67 streams), TOR_ACCEL_BLOCKS|TOR_ACCEL_F32 = tor_render() default", dict(seeding=tor.SEED_PIXEL, accel=3))}


def predict_frames(table, shards=(1, 2, 4, 8)):
    """table: [[frame, upload_ms, kernel_ms], ...] of ONE GPU rendering every frame.  Returns {rule: {N: {...}}}."""
    cost = [(int(f), float(u) + float(k)) for f, u, k in table]
    t1 = sum(c for _, c in cost)
    out = {}
    for rule in ("f mod N", "contiguous blocks", "LPT on measured cost"):
        out[rule] = {}
        for N in shards:
            loads = [0.0] * N
            if rule == "f mod N":
                for i, (f, c) in enumerate(cost):
                    loads[i % N] += c
            elif rule == "contiguous blocks":
                per = (len(cost) + N - 1) // N
                for i, (f, c) in enumerate(cost):
                    loads[min(i // per, N - 1)] += c
            else:
                for f, c in sorted(cost, key=lambda x: -x[1]):
                    loads[loads.index(min(loads))] += c
            mx = max(loads)
            out[rule][str(N)] = {"max_ms": round(mx, 2), "imbalance_max_over_mean": round(mx / (t1 / N), 4),
                                 "predicted_speedup": round(t1 / mx, 3), "predicted_efficiency": round(t1 / mx / N, 4)}
    return t1, out


def c5_block(path):
    with open(path) as f:
        line = json.loads([l for l in f.read().splitlines() if l.strip().startswith("{")][-1])
    table = line["frames"]["table"]
    t1, pred = predict_frames(table)
    ks = [r[2] for r in table]
    return {"config": "c5", "config_desc": line["config"]["workload"], "mode_desc": f"{line['config']['seeding']} streams, accel {line['config']['accel']}",
            "source": os.path.relpath(path), "frames": len(table), "one_gpu_total_ms": round(t1, 1),
            "kernel_ms_per_frame": {"min": min(ks), "max": max(ks), "mean": round(sum(ks) / len(ks), 3),
                                    "first_20_mean": round(sum(ks[:20]) / 20, 3), "last_20_mean": round(sum(ks[-20:]) / 20, 3)},
            "what": "a GPU's share of the frame-parallel job = the sum of (upload + kernel) of its frames; predicted speed-up = T(1) / max share; no collective, "
                    "the per-GPU video stage and D2H ride along with the frames (bench line: with_video_stage)",
            "dealing": pred}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c5-from", default=None, help="bench line of `bench.py --workload c5 --frames 240`: predict frame-parallel scaling from its per-frame table (no GPU needed)")
    ap.add_argument("--configs", default="c3,c4")
    ap.add_argument("--modes", default="a,b,c")
    ap.add_argument("--shards", default="1,2,4,8")
    ap.add_argument("--row-tile", type=int, default=1)
    ap.add_argument("--repeat", type=int, default=1, help="timed launches per shard (the minimum is kept)")
    ap.add_argument("--out", default="gpurun_out/shard_times.json")
    args = ap.parse_args()
    if args.c5_from:
        block = c5_block(args.c5_from)
        result = {"runs": []}
        if os.path.exists(args.out):
            with open(args.out) as f:
                result = json.load(f)
        result["runs"] = [r for r in result.get("runs", []) if r.get("config") != "c5"] + [block]
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(result, f, indent=1)
        for rule, per in block["dealing"].items():
            print("c5", rule, {n: v["predicted_speedup"] for n, v in per.items()})
        print("wrote", args.out)
        return
    scene, cam = tor.random_scene(0xFACADE), tor.camera()
    ctx = tor.Context(0)
    ctx.upload(scene.list())
    st = torch.cuda.current_stream().cuda_stream
    result = {"what": __doc__.split("\n\n")[1].replace("\n", " "), "device": torch.cuda.get_device_name(0), "row_tile": args.row_tile, "runs": []}
    for cname in args.configs.split(","):
        H, W, spp, cdesc = CONFIGS[cname]
        for m in args.modes.split(","):
            mdesc, mopt = MODES[m]
            entry = {"config": cname, "config_desc": cdesc, "mode": m, "mode_desc": mdesc, "shards": {}}
            t1 = None
            for N in [int(x) for x in args.shards.split(",")]:
                times, kms = [], []
                for k in range(N):
                    o = tor.make_options(shard_index=k, shard_count=N, row_tile=args.row_tile, **mopt)
                    rows = len(tor.shard_rows(H, args.row_tile, k, N))
                    buf = torch.empty((rows, W, 3), dtype=torch.float64, device="cuda")
                    ctx.render_device(cam, H, W, 2, 2.2, 50, o, buf.data_ptr(), st)   # layouts, buffers
                    torch.cuda.synchronize()
                    best = None
                    for _ in range(args.repeat):
                        t = time.perf_counter()
                        ctx.render_device(cam, H, W, spp, 2.2, 50, o, buf.data_ptr(), st)
                        torch.cuda.synchronize()
                        dt = time.perf_counter() - t
                        best = dt if best is None or dt < best else best
                    times.append(best * 1e3)
                    kms.append(ctx.last_kernel_ms()[0])
                    del buf
                if N == 1:
                    t1 = times[0]
                mx, mn, mean = max(times), min(times), sum(times) / len(times)
                entry["shards"][str(N)] = {
                    "step_ms": [round(t, 2) for t in times], "integrator_kernel_ms": [round(t, 2) for t in kms],
                    "max_ms": round(mx, 2), "imbalance_max_over_mean": round(mx / mean, 4),
                    "predicted_speedup": round(t1 / mx, 3) if t1 else None,
                    "predicted_efficiency": round(t1 / mx / N, 4) if t1 else None,
                    "Msamples_per_s_aggregate": round(H * W * spp / (mx * 1e-3) / 1e6, 1)}
                print(cname, m, "N", N, json.dumps(entry["shards"][str(N)]), flush=True)
            result["runs"].append(entry)
            os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
            with open(args.out, "w") as f:
                json.dump(result, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
