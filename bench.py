#!/usr/bin/env python3
"""bench.py -- Msamples/s of the integrator hot path (render.nim:49-68) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (default): BASELINE.json configs[2] -- the configuration the target is quoted on -- the
reference's random_scene (seed 0xFACADE, 485 objects) at 1920x1080, 1000 spp, depth 50, rendered with the
counter-based per-sample streams (TOR_SEED_SAMPLE) and the reference's rounding and algorithm
(TOR_ARITH_STRICT, float64 brute-force closest hit).  A step = one pass of the hot path over one frame.

Two timed regions are reported on one GPU, each over the same K steps:
  value         -- the harness contract's region: scene and camera resident in HBM before the clock starts, the
                   frame stays on the device (tor_render_device).
  host_canvas   -- SURVEY 8(d)'s region, what a Nim caller of render() pays
                   (trace_of_radiance.nim:60-64): tor_render_opt on a HOST canvas = scene + camera H2D (the
                   scene is cached after the first call), kernels, D2H into canvas.pixels.
N > 1 (one process per GPU): image rows are dealt to the ranks round-robin (render.nim:55's `parallelFor row`
across GPUs), every rank renders its rows, then ONE gather of the row shards to rank 0 over xGMI --
tor_render_gather_device: RCCL inside the library (falls back to torch.distributed's all_gather, also RCCL,
if the library's communicator cannot be set up) -- both inside the timed region.  STRONG scaling by default: the
frame -- BASELINE configs[2], 1920x1080x1000 spp, the config the ">= 7x at 8 GPUs" target is quoted on -- is the same
whatever N (`--scaling weak` multiplies spp by N instead; `--config c4` selects BASELINE configs[3], 3840x2160x4096).
`python bench.py --gpus N` WITHOUT torchrun drives N devices from one process through the drop-in itself
(TorOptions.devices): the path a Nim host uses.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the field meanings).
"""
import argparse
import importlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Algorithmic work per pixel-sample on the benchmark scene: every closest-hit
# query tests all 485 objects; per test the reference's formulation costs 23 (static sphere, 90 objects) or
# 35 (moving sphere, 395 objects) float64 operations -- nothing hoisted, nothing fused -- + ~0.5 k per sample
# for camera/scatter/sky.  Queries per sample are MEASURED in the run (tor_last_stats), 2.60 at 1080p.
OPS_PER_TEST_STATIC, OPS_PER_TEST_MOVING, OPS_PER_SAMPLE_FIXED = 23.0, 35.0, 500.0
FLOPS_PER_SAMPLE_SURVEY = 41.9e3
PEAK_FP64_VECTOR_TFLOPS = 78.6   # 256 CU x 4 SIMD x 16 lanes x 2 (FMA) x 2.4 GHz  (datasheet)
PEAK_FP64_NOFMA_TFLOPS = 39.3    # the same issue rate with add/mul only (the reference has no FMA)
PEAK_HBM_GBPS = 8000.0           # MI355X_MICROARCH.md: 8 TB/s spec
ACCEL_BITS = {"none": 0, "blocks": 1, "f32": 2, "blocks+f32": 3}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1000, help="samples per pixel of the frame (--scaling weak: per GPU, x N ranks)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong (default) = the SAME frame on N GPUs -- BASELINE configs[2] stays 1920x1080x1000 spp, the config the "
                         ">= 7x at 8 GPUs target is quoted on; weak = spp x N, every GPU traces as many samples as the single-GPU run")
    ap.add_argument("--config", choices=["c1", "c2", "c3", "c4"], default=None,
                    help="preset sizes: c1 = BASELINE configs[0] geometry 384x216x100, c2 = configs[1] 1920x1080x100, c3 = configs[2] "
                         "1920x1080x1000 (default), c4 = configs[3] 3840x2160x4096 (the config BASELINE assigns to 8 GPUs)")
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--seeding", choices=["sample", "pixel"], default="sample")
    ap.add_argument("--arith", choices=["strict"], default="strict", help="(TOR_ARITH_FUSED was removed in round 5)")
    ap.add_argument("--accel", choices=list(ACCEL_BITS), default=None,
                    help="none (default for the frame workload): the reference's brute-force closest hit (the metric's algorithm); others: exact "
                         "accelerations (default for --workload c5: blocks+f32, what tor_render() runs by default and the sensible mode for 1601 spheres)")
    ap.add_argument("--row-tile", type=int, default=1,
                    help="rows per shard tile; 1 = row-cyclic: every rank gets nrows/N rows (+-1) of statistically equal cost")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", choices=["frame", "c5"], default="frame",
                    help="frame: random_scene frame (default, the metric's config); c5: BASELINE configs[4], the animated "
                         "bouncing-spheres scene, 256 spp per frame, frames dealt round-robin to the GPUs (no collective)")
    ap.add_argument("--frames", type=int, default=0,
                    help="--workload c5: render the first F frames of the animation IN ORDER (240 = all of BASELINE configs[4]); steps = the frames "
                         "this rank renders; per-frame table in the line; a second pass times the device video stage inside the region")
    ap.add_argument("--verify", action="store_true", help="N > 1: rank 0 also renders the whole frame alone and requires the gathered frame to be identical")
    ap.add_argument("--no-accel-leg", action="store_true", help="skip the secondary exact-acceleration measurements")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the host-canvas (SURVEY 8d) region")
    ap.add_argument("--no-stats", action="store_true", help="skip the untimed extra launch that collects the kernel's workload counters")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 --pmc passes (HBM traffic of one launch)")
    ap.add_argument("--aux-steps", type=int, default=3, help="steps of each secondary leg (host canvas, accelerations)")
    ap.add_argument("--gather", choices=["lib", "torch"], default="lib", help="N > 1: framebuffer gather inside the library (RCCL) or torch.distributed")
    ap.add_argument("--pmc-child", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.accel is None:
        args.accel = "blocks+f32" if args.workload == "c5" else "none"
    if args.config:
        args.width, args.height, args.spp = {"c1": (384, 216, 100), "c2": (1920, 1080, 100), "c3": (1920, 1080, 1000),
                                             "c4": (3840, 2160, 4096)}[args.config]
    return args


def frame_spp(args, world):
    """Samples per pixel of the rendered frame: fixed (strong scaling, default) or x N (weak)."""
    return args.spp * (max(world, 1) if args.scaling == "weak" else 1)


def config_name(W, H, spp, world, scaling="strong"):
    """Which BASELINE.json configs[] entry this run is (derived from the sizes of the RENDERED frame, never hard-coded)."""
    tail = "" if world == 1 else (f", the same frame on {world} GPUs (strong scaling)" if scaling == "strong" else
                                  f" weak-scaled to {world} GPUs ({spp} spp in all)")
    base_spp = spp if scaling == "strong" else spp // max(world, 1)
    if (W, H) == (1920, 1080) and base_spp == 100:
        return "BASELINE configs[1]" + tail
    if (W, H) == (1920, 1080) and base_spp == 1000:
        return "BASELINE configs[2]" + tail
    if (W, H) == (3840, 2160) and spp == 4096:
        return "BASELINE configs[3]" + ("" if world == 1 else f" on {world} GPUs")
    if (W, H) == (384, 216) and base_spp == 100:
        return "BASELINE configs[0] geometry on the GPU" + tail
    return "custom size (not a BASELINE config)" + tail


def git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "log", "-1", "--format=%h", "--", "profiles"], capture_output=True, text=True,
                              timeout=10).stdout.strip() or None
    except Exception:
        return None


def from_profile(W, H, spp, seeding, arith, accel="none"):
    """What the COMMITTED rocprofv3 PMC passes (profiles/traffic.json, rebuilt from the summaries by
    tools/update_traffic.py) measured for this configuration on an earlier run -- evidence, not a measurement of
    this run; {} when there is none."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        e = dict(t[f"{W}x{H}x{spp}:{seeding}:{arith}" + ("" if accel == "none" else ":" + accel)])
        e["_what"] = "replayed from committed rocprofv3 summaries (profiles/), NOT measured in this run"
        commit = git_head()
        if commit:
            e["_profiles_commit"] = commit
        return e
    except Exception:
        return {}


# ---------------------------------------------------------------------------------------------------------
# live HBM traffic: rocprofv3 --pmc around ONE launch of the same kernel/config, in a child process
# ---------------------------------------------------------------------------------------------------------
def pmc_child(spec):
    """Child of live_traffic(): one launch through the bare C ABI (ctypes only, no torch)."""
    vals = [int(x) for x in spec.split(",")]
    W, H, spp, depth, seeding, arith, accel = vals[:7]
    shard_index, shard_count, row_tile = (vals[7:10] if len(vals) >= 10 else (0, 1, 1))
    anim_frame = vals[10] if len(vals) >= 11 else -1     # >= 0: that frame of the bouncing-spheres animation (BASELINE configs[4])
    os.environ["TOR_NO_TORCH"] = "1"
    tor = importlib.import_module("trace-of-radiance_amd")
    if anim_frame >= 0:
        it = iter(tor.Animation(H, W, 0.005, 0.0, 7.2).scenes(6))
        for _ in range(anim_frame + 1):
            cam, scene, _t = next(it)
    else:
        scene, cam = tor.random_scene(0xFACADE), tor.camera(aspect_ratio=W / H) if (W * 9 != H * 16) else tor.camera()
    cv = tor.new_canvas(H, W, spp, 2.2)
    # two identical calls; live_traffic() reads the counters of the SECOND launch -- the steady state every timed step of the
    # host-canvas leg is in.  The first launch of a process (fresh allocations) fetches the canvas once from HBM and writes it
    # back as 64-byte lines on top: +65 MB read, +123 MB written at 1080p (profiles/r3_traffic_reconcile.txt, HISTORY 6)
    for _ in range(2):
        tor.render(cv, cam, scene.list(), depth, tor.make_options(seeding=seeding, arith=arith, accel=accel, shard_index=shard_index,
                                                                  shard_count=shard_count, row_tile=row_tile))
    print("pmc-child done", float(cv.pixels.mean()))


# rocprofv3 --pmc passes of the live measurement.  FETCH_SIZE and WRITE_SIZE cannot share a pass (MI355X_MICROARCH.md "rocprofv3
# PMC slots"); the SQ instruction counters fit one pass of eight (tools/profile_gpu.sh runs the same groups); GRBM_GUI_ACTIVE --
# the denominator of the issue utilisation -- gets its own, as in the committed summaries.
PMC_PASSES = (
    ("FETCH_SIZE",),
    ("WRITE_SIZE",),
    ("SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS"),
    ("GRBM_GUI_ACTIVE",),
)
PEAK_FP64_ISSUE_TLANEOPS = 39.3   # 256 CU x 4 SIMD x 16 lanes x 2.4 GHz: float64 lane-operations per second (an FMA is ONE), datasheet clock
SIMDS, XCDS = 1024, 8


def live_traffic(W, H, spp, depth, seeding, arith, accel, timeout_s=240, shard=(0, 1, 1), sq=True, anim_frame=-1):
    """Counters of ONE integrate_kernel launch of this configuration, from separate rocprofv3 --pmc passes around a child
    process (PMC_PASSES): HBM bytes -- FETCH_SIZE doubled (gfx950 tallies 128-byte requests at 64 B), WRITE_SIZE as is
    (uncalibrated), both in KiB, as the guide's HBM section says -- and, with sq=True, the executed-instruction view (VALU /
    float64 wave-instructions, VALU-active quad-cycles, GPU-active cycles).  None when rocprofv3 is missing or a traffic pass
    fails; a failing SQ pass only drops the `executed` block (its reason is kept)."""
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    import glob
    import sqlite3
    out = {}
    base = tempfile.mkdtemp(prefix="tor_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", TOR_NO_TORCH="1")
    spec = f"{W},{H},{spp},{depth},{seeding},{arith},{accel},{shard[0]},{shard[1]},{shard[2]},{anim_frame}"
    t_all = time.perf_counter()
    sq_note = None
    try:
        for n_pass, counters in enumerate(PMC_PASSES):
            traffic_pass = counters[0] in ("FETCH_SIZE", "WRITE_SIZE")
            if not traffic_pass and not sq:
                continue
            d = os.path.join(base, f"pass{n_pass}")
            try:
                r = subprocess.run([exe, "--pmc", *counters, "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
                                    "--pmc-child", spec], capture_output=True, text=True, timeout=timeout_s, cwd="/tmp", env=env)
                why = None if r.returncode == 0 else f"rocprofv3 --pmc {' '.join(counters)} failed: {r.stderr[-300:]}"
            except subprocess.TimeoutExpired:
                why = f"rocprofv3 --pmc {' '.join(counters)}: no result after {timeout_s} s"
            rows = []
            if why is None:
                dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
                if not dbs:
                    why = f"rocprofv3 --pmc {counters[0]}: no rocpd database"
                else:
                    con = sqlite3.connect(dbs[0])
                    rows = con.execute("select name, counter_name, sum(counter_value), max(duration), dispatch_id from pmc_events "
                                       "group by name, counter_name, dispatch_id order by dispatch_id").fetchall()
                    rows = [r_ for r_ in rows if "integrate_kernel" in r_[0]]
                    if not rows:
                        why = f"rocprofv3 --pmc {counters[0]}: no integrate_kernel dispatch in the database"
            if why is not None:
                if traffic_pass:
                    return None, why
                sq_note = why
                continue
            longest = max(r_[3] for r_ in rows)
            last = [r_ for r_ in rows if r_[3] >= 0.5 * longest][-1][4]   # the LAST frame launch (the 2-spp cost probes are tiny)
            for r_ in rows:
                if r_[4] == last:
                    out[r_[1]] = float(r_[2])
                    out[r_[1] + "_kernel_ms"] = float(r_[3]) / 1e6
                    out["kernel"] = r_[0]
    except Exception as e:  # noqa
        return None, f"live PMC pass failed: {e!r}"
    finally:
        shutil.rmtree(base, ignore_errors=True)
    out["bytes"] = (2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024.0
    out["pmc_seconds"] = round(time.perf_counter() - t_all, 1)
    out["what"] = ("second of two identical launches in a child process (steady state), one rocprofv3 --pmc pass per counter group; "
                   "FETCH_SIZE doubled (gfx950), WRITE_SIZE as is: for SEED_SAMPLE it is the float64 flush atomics -- each 8-byte atomic "
                   "is one 32-byte write request at the fabric -- plus the write-back of the canvas clear (profiles/r3_traffic_reconcile.txt)")
    if sq_note:
        out["executed_note"] = sq_note
    return out, None


def executed_from_counters(c, samples):
    """The executed-instruction view of one launch from live_traffic()'s SQ / GRBM counters (`samples` = the pixel-samples that
    launch traced).  Wave-instruction counts x 64 lanes = lane-operations; an FMA counts ONCE against the ISSUE peak (39.3 T/s)
    and twice in executed_tflops (78.6 peak).  SQ_ACTIVE_INST_VALU is in quad-cycles (x 4 = SIMD cycles with a VALU instruction
    in flight); GRBM_GUI_ACTIVE, summed over the 8 XCDs, / 8 = GPU-active cycles, x 1024 SIMDs = the SIMD cycles there were."""
    need = ("SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE")
    if not c or any(k not in c for k in need):
        return None
    k_s = c["SQ_INSTS_VALU_kernel_ms"] * 1e-3
    f64 = c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_FMA_F64"]
    lane_ops = f64 * 64.0 / k_s
    simd_cycles = c["GRBM_GUI_ACTIVE"] / XCDS * SIMDS
    clock_ghz = c["GRBM_GUI_ACTIVE"] / XCDS / (c["GRBM_GUI_ACTIVE_kernel_ms"] * 1e-3) / 1e9
    return {
        "source": "measured in this run (rocprofv3 --pmc around one launch of the same configuration, child process)",
        "kernel": c.get("kernel"), "kernel_ms": round(c["SQ_INSTS_VALU_kernel_ms"], 3),
        "valu_wave_instructions": c["SQ_INSTS_VALU"], "fp64_wave_instructions": f64,
        "fp64_fma_wave_instructions": c["SQ_INSTS_VALU_FMA_F64"],
        "salu_wave_instructions": c.get("SQ_INSTS_SALU"), "lds_wave_instructions": c.get("SQ_INSTS_LDS"),
        "valu_per_sample": round(c["SQ_INSTS_VALU"] * 64.0 / samples, 1), "fp64_per_sample": round(f64 * 64.0 / samples, 1),
        "fp64_lane_ops_per_s": lane_ops, "frac_fp64_issue": round(lane_ops / (PEAK_FP64_ISSUE_TLANEOPS * 1e12), 4),
        "executed_tflops": round((f64 + c["SQ_INSTS_VALU_FMA_F64"]) * 64.0 / k_s / 1e12, 3),
        "valu_issue_util": round(c["SQ_ACTIVE_INST_VALU"] * 4.0 / simd_cycles, 4),
        "gpu_clock_ghz": round(clock_ghz, 3),
        "formulae": "frac_fp64_issue = (ADD + MUL + FMA float64 wave-instructions) x 64 / kernel time / 39.3e12; valu_issue_util = "
                    "SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)",
    }


# ---------------------------------------------------------------------------------------------------------
# CPU baseline (oracle = checker; imported only here)
# ---------------------------------------------------------------------------------------------------------
def cpu_baseline(width, height, spp, depth, target_seconds, objs=None, cam=None, scene_name=None):
    """The oracle in its reference-faithful mode (per-pixel streams, libm, no FMA), OpenMP over rows with
    schedule(dynamic,1) -- the analogue of Weave's parallelFor row (render.nim:55) -- on all host cores.
    (1) BASELINE configs[0] in full (384x216x100, the reference's own main()): timed, and its PPM quantisation
        is compared with the reference's golden PNG IN THIS RUN;
    (2) every k-th row of the bench frame, sized for ~target_seconds: the `value`."""
    from oracle import oracle as O
    import numpy as np
    ref_objs, _ = O.random_scene(0xFACADE)
    ref_cam = O.camera()
    if objs is None:
        objs, cam = ref_objs, ref_cam
    cores = O.num_threads()
    t = time.perf_counter()
    c1 = O.render(216, 384, 100, ref_cam, ref_objs, max_depth=50)
    c1_dt = time.perf_counter() - t
    png_equal, png_note = None, None
    try:
        from PIL import Image
        g = np.array(Image.open(os.path.join(ROOT, "tests", "golden", "book2_motion_blur.png")).convert("RGB"))
        rgb = O.quantize_ppm(c1.pixels)
        png_equal = bool(np.array_equal(rgb, g))
        png_note = f"{int((rgb != g).sum())} of {g.size} 8-bit channels differ"
    except Exception as e:  # noqa
        png_note = f"not checked: {e!r}"
    rate = 216 * 384 * 100 / max(c1_dt, 1e-6)
    # a bounded sample of the bench frame: evenly spaced rows, parallelised over (row, 16-column) tiles so that a
    # few rows still load every core (Weave, too, splits rows and then columns)
    # ... in TWO halves (rows 0, 2k, 4k, ... and rows k, 3k, 5k, ...): the two rates bracket the run-to-run and row-to-row spread of a
    # sample this small (+-10 % between runs), the value is their pooled rate
    rate *= len(ref_objs) / max(len(objs), 1)      # (a closest-hit query walks the whole list: cost per sample ~ objects)
    want_rows = int(min(height, max(4, rate * target_seconds / (width * spp))))
    step = max(1, height // max(want_rows, 1))
    halves = []
    for begin in (0, step):
        n_rows = len(range(begin, height, 2 * step))
        if n_rows == 0:
            continue
        t = time.perf_counter()
        O.render(height, width, spp, cam, objs, max_depth=depth, rows=(begin, height), row_step=2 * step, col_block=16)
        halves.append((n_rows, time.perf_counter() - t))
    rows = sum(n for n, _ in halves)
    dt = sum(d for _, d in halves)
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
        "sample": (f"{scene_name}: " if scene_name else "") + f"every {step}th row ({rows} of {height} rows) of the {width}x{height}x{spp}spp frame, "
                  f"depth {depth}: {samples / 1e6:.1f} Msamples in {dt:.1f} s; oracle/tor_oracle.c faithful mode "
                  f"(seed(row,col) streams, libm, -ffp-contract=off), OpenMP schedule(dynamic,1) over (row, 16-column) tiles",
        "cpu_model": model,
        "halves": [round(n * width * spp / d / 1e6, 4) for n, d in halves],
        "spread": "the two interleaved halves of the row sample, timed separately (Msamples/s): their difference is the noise of `value`",
        "c1": {"workload": "BASELINE configs[0]: 384x216, 100 spp, depth 50 (trace_of_radiance.nim main())",
               "value": round(216 * 384 * 100 / c1_dt / 1e6, 4), "seconds": round(c1_dt, 3),
               "ppm_equals_reference_png": png_equal, "ppm_vs_png": png_note},
    }


def bench_animation(args, tor, torch, dist, world, rank, local_rank):
    """BASELINE configs[4]: scenes_animated bouncing spheres (scenes_animated.nim:176-225, the loop of
    trace_of_radiance_animation.nim:101-214), 1920x1080x256 spp, frame-parallel: frame f -> GPU f mod N, no collective.
    A step = one frame per GPU: scene upload (1601 objects, host layouts built per frame) + integrator; the frame stays on the device.
    --frames F: the first F frames of the animation in order (240 = the whole config), every frame's upload / kernel time in the line
    (`frames`: the table tools/shard_times.py --config c5 predicts frame-parallel scaling from), and a SECOND pass over the same
    frames with the device video stage inside the region (canvas -> RGB8 -> Y'CbCr 4:2:0 -> I_PCM slice bytes, D2H of the slice:
    io/color_conversions.nim:180-252, h264.nim)."""
    import numpy as np
    H, W = args.height, args.width
    spp = args.spp if args.spp != 1000 else 256
    seeding = tor.SEED_SAMPLE if args.seeding == "sample" else tor.SEED_PIXEL
    arith = tor.ARITH_STRICT
    accel = ACCEL_BITS[args.accel]
    N = max(world, 1)
    me = rank if world > 1 else 0
    if args.frames > 0:
        n_frames = args.frames                       # frames 0 .. F-1; warm-up = frame 0 rendered `warmup` times before the clock
        first_timed = 0
    else:
        n_frames = (args.warmup + args.steps) * N    # (the short form: the first `warmup` frames of every rank are untimed)
        first_timed = args.warmup * N
    anim = tor.Animation(H, W, 0.005, 0.0, 7.2)      # 240 frames at skip 6 (trace_of_radiance_animation.nim:116-119)
    frames = []                                      # (frame index, camera, scene) of this rank
    seen = 0
    for f, (cam, scene, t) in enumerate(anim.scenes(6)):
        if f >= n_frames:
            break
        seen = f + 1
        if f % N == me:
            frames.append((f, cam, scene))
    n_frames = seen                                  # (the animation has 240 frames)
    timed = [fr for fr in frames if fr[0] >= first_timed]
    dev = local_rank % max(torch.cuda.device_count(), 1) if world > 1 else 0
    ctx = tor.Context(dev)
    opt = tor.make_options(seeding=seeding, arith=arith, accel=accel)
    buf = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    slice_bytes = tor.h264_frame_bytes(W, H)
    d_slice = torch.empty(slice_bytes, dtype=torch.uint8, device="cuda")
    h_slice = torch.empty(slice_bytes, dtype=torch.uint8).pin_memory()

    def render(cam, scene, video):
        """One frame: returns (upload ms on the host, kernel ms by HIP events, encode ms by events | None)."""
        t = time.perf_counter()
        ctx.upload(scene.list())
        ctx.render_device(cam, 16, 16, 1, 2.2, 1, opt, buf.data_ptr(), stream)   # builds the layouts this mode uses (lazy) -- part of the upload
        torch.cuda.synchronize()
        up = (time.perf_counter() - t) * 1e3
        ctx.render_device(cam, H, W, spp, 2.2, args.depth, opt, buf.data_ptr(), stream)
        enc = None
        if video:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ctx.encode_frame_device(buf.data_ptr(), H, W, d_slice.data_ptr(), 0, 0, 0, stream)
            h_slice.copy_(d_slice, non_blocking=True)
            e1.record()
            torch.cuda.synchronize()
            enc = e0.elapsed_time(e1)
        else:
            torch.cuda.synchronize()                 # (the next upload overwrites the device scene: it waits for the frame anyway)
        return up, ctx.last_kernel_ms()[0], enc

    def timed_pass(video):
        for _ in range(args.warmup if args.frames > 0 else 0):
            render(frames[0][1], frames[0][2], video)
        for fr in frames:
            if fr[0] < first_timed:
                render(fr[1], fr[2], video)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        table = []
        for f, cam, scene in timed:
            up, k, enc = render(cam, scene, video)
            table.append((f, up, k, enc))
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, table

    # live counters of one frame's launch FIRST, while this process has not touched the GPU's memory much yet: read after the 480
    # scene uploads of a 240-frame run the L2's FETCH_SIZE / WRITE_SIZE came out 250 x too large (15 GB / 42 GB for a launch that
    # moves 24 MB / 165 MB: the same command gave the small numbers before the run and with --frames 3) -- the driver's clearing of the
    # freed scene buffers shares the L2 with the profiled kernel; the SQ counters were right either way
    mid = timed[len(timed) // 2]
    traffic, traffic_note = None, "skipped (--no-pmc)" if args.no_pmc else "N > 1: not read"
    if rank == 0 and not args.no_pmc and world == 1:
        traffic, traffic_note = live_traffic(W, H, spp, args.depth, seeding, arith, accel, anim_frame=mid[0])
    elapsed, table = timed_pass(False)
    steps = len(timed)                                # frames per GPU inside the region (every rank has the same number +- 1)
    total_frames = sum(1 for f in range(first_timed, n_frames))
    total = H * W * spp * total_frames
    k_ms = sum(r[2] for r in table) / max(len(table), 1)
    video = None
    if args.frames > 0:
        v_elapsed, v_table = timed_pass(True)
        video = {"value": round(total / v_elapsed / 1e6, 2), "unit": "Msamples/s", "frames_per_s": round(total_frames / v_elapsed, 3),
                 "ms_per_step": round(v_elapsed / max(steps, 1) * 1e3, 3),
                 "encode_ms_per_frame": round(sum(r[3] for r in v_table) / max(len(v_table), 1), 4),
                 "slice_bytes_per_frame": int(slice_bytes),
                 "region": "scene upload + integrator + encode_ipcm_kernel (canvas -> RGB8 -> Y'CbCr 4:2:0 -> I_PCM slice bytes) + D2H of the "
                           "slice into pinned host memory, per frame (what trace_of_radiance_animation.nim:181-196 pays per frame)",
                 "vs_value": round(elapsed / v_elapsed, 4)}
    result = None
    if rank == 0:
        # workload counters and the reference-formulation operation count on the middle frame of this rank (one extra, untimed launch)
        ctx.upload(mid[2].list())
        ctx.set_stats(True)
        ctx.render_device(mid[1], H, W, spp, 2.2, args.depth, opt, buf.data_ptr(), stream)
        torch.cuda.synchronize()
        st = ctx.last_stats()
        ctx.set_stats(False)
        n_obj = len(mid[2])
        n_moving = sum(1 for i in range(n_obj) if mid[2].objects[i].kind == tor.MOVING_SPHERE)
        q_per_sample = st.hit_queries / max(st.samples, 1)
        fps = q_per_sample * (n_moving * OPS_PER_TEST_MOVING + (n_obj - n_moving) * OPS_PER_TEST_STATIC) + OPS_PER_SAMPLE_FIXED
        k_rate = H * W * spp / (k_ms * 1e-3)
        tflops = k_rate * fps / 1e12
        hbm_bytes = H * W * 24.0 * (2 if seeding == tor.SEED_SAMPLE else 1) + n_obj * 128.0
        ex = executed_from_counters(traffic, H * W * spp)
        roof = {"bound": "valu_fp64", "kernel": (traffic or {}).get("kernel") or "tor::integrate_kernel", "achieved": round(tflops, 3),
                "peak": PEAK_FP64_VECTOR_TFLOPS, "unit": "TFLOP/s", "frac": round(tflops / PEAK_FP64_VECTOR_TFLOPS, 4),
                "flops_per_sample": round(fps, 1), "kernel_ms": round(k_ms, 3), "launches_averaged": len(table),
                "traffic": traffic["bytes"] if traffic else None, "traffic_detail": traffic if traffic else traffic_note,
                "traffic_scope": f"one launch = frame {mid[0]} (the middle frame of the timed range)",
                "workload_live": {"frame": mid[0], "objects": n_obj, "hit_queries_per_sample": round(q_per_sample, 4),
                                  "candidates_per_query": round(st.candidates / max(st.hit_queries, 1), 3),
                                  "blocks_entered_per_query": round(st.block_tests / 8.0 / max(st.hit_queries, 1), 3),
                                  "exact_tests_per_query": round(st.exact_tests / max(st.hit_queries, 1), 3),
                                  "lane_utilisation": round(st.hit_queries / max(st.lane_slots, 1), 4),
                                  "algorithmic_fp64_ops_per_sample": round(fps, 1),
                                  "formula": f"queries/sample x ({n_moving} moving x 35 + {n_obj - n_moving} static x 23 float64 ops, the reference's "
                                             "brute-force formulation, SURVEY 8d) + 500"},
                "executed_live": ex, "frac_executed": ex["frac_fp64_issue"] if ex else None,
                "hbm": {"bound": "hbm", "achieved": round(hbm_bytes / (k_ms * 1e-3) / 1e9, 4), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                        "frac": round(hbm_bytes / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 8), "algorithmic_bytes_per_launch": hbm_bytes},
                "note": "frac prices the throughput in the reference's units (every ray x all " + str(n_obj) + " objects in unfused float64, SURVEY 8d) and exceeds 1 with "
                        "the exact accelerations by design; executed_live (counters of one frame's launch, read in this run) holds the fractions of a roof: "
                        "frac_fp64_issue and valu_issue_util (the accelerated kernel issues packed float32 for its filters, so its float64 issue share is small)"}
        result = {
            "metric": "Msamples/s (pixels x spp / s) on the animated bouncing-spheres scene", "value": round(total / elapsed / 1e6, 2),
            "unit": "Msamples/s", "n_gpus": args.gpus, "steps": steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / max(steps, 1) * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64" if args.accel == "none" else "f64 (canvas bit-identical to the float64 path)", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[4]: scenes_animated (1601 static spheres per frame), {W}x{H}, {spp} spp, "
                                   f"depth {args.depth}, frames {first_timed}..{n_frames - 1} of 240, one frame per GPU per step",
                       "seeding": args.seeding, "arith": args.arith, "accel": args.accel,
                       "parallelism": f"frame f -> GPU f mod {N}, no collective",
                       "timed_region": "per frame: tor_scene_upload (host layouts + H2D) + integrator + finalize; the frame stays on the device"},
            "kernel_ms": round(k_ms, 3), "frames_per_s": round(total_frames / elapsed, 3),
            "scene_upload_ms_per_frame": round(sum(r[1] for r in table) / max(len(table), 1), 3),
            "roofline": roof}
        if video is not None:
            result["with_video_stage"] = video
        if args.frames > 0:
            result["frames"] = {"what": "per frame of this rank, in order: [frame, upload_ms (host: layouts + H2D), integrator kernel_ms (HIP events)]",
                                "table": [[r[0], round(r[1], 3), round(r[2], 3)] for r in table],
                                "kernel_ms_min_max": [round(min(r[2] for r in table), 3), round(max(r[2] for r in table), 3)]}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if args.no_cpu_baseline:
            result["cpu_baseline"] = {"value": None, "note": "skipped (--no-cpu-baseline)"}
        else:
            from oracle import oracle as O
            mid_f = timed[len(timed) // 2][0]
            ocam, oobjs = None, None
            for f, (c_, o_, _t) in enumerate(O.animation_scenes(H, W, 0.005, 0.0, 7.2, skip=6, max_frames=mid_f + 1)):
                ocam, oobjs = c_, o_
            result["cpu_baseline"] = cpu_baseline(W, H, spp, args.depth, args.cpu_seconds, objs=oobjs, cam=ocam,
                                                  scene_name=f"animation frame {mid_f} ({len(oobjs)} objects; the oracle's restatement of scenes_animated.nim, parity-unpinned)")
        emit_line(json.dumps(result))


def algorithmic_flops_per_sample(scene, tor, queries_per_sample=2.6022):
    """SURVEY 8(d)'s formulation: queries/sample x (moving x 35 + static x 23 float64 ops) + 500."""
    n_obj = len(scene)
    n_moving = sum(1 for i in range(n_obj) if scene.objects[i].kind == tor.MOVING_SPHERE)
    return queries_per_sample * (n_moving * OPS_PER_TEST_MOVING + (n_obj - n_moving) * OPS_PER_TEST_STATIC) + OPS_PER_SAMPLE_FIXED


def bench_single_process_multi_device(args, tor):
    """`python bench.py --gpus N` without torchrun: ONE process drives N devices through the drop-in itself
    (tor_render_opt with TorOptions.devices: a host thread + stream per device, row-cyclic shards, framebuffer
    gather) -- the path a Nim host takes.  Timed region = SURVEY 8(d): host canvas in, host canvas out.
    The line is self-contained: CPU baseline timed BEFORE the devices are busy, kernel-level roofline from
    the per-device HIP events of the launches, live PMC traffic of one shard's launch, the gather leg and the RCCL
    communicator's rank count, and the N = 1 value of the same invocation."""
    import torch
    H, W, N = args.height, args.width, args.gpus
    spp = frame_spp(args, N)
    n_dev = max(torch.cuda.device_count(), 1)
    devices = [k % n_dev for k in range(N)]
    n_distinct = len(set(devices))
    cpu = None
    if not args.no_cpu_baseline:
        cpu = cpu_baseline(W, H, args.spp, args.depth, args.cpu_seconds)   # the host cores are idle: no device work yet
    scene = tor.random_scene(0xFACADE)
    cam = tor.camera(aspect_ratio=W / H) if (W * 9 != H * 16) else tor.camera()
    seeding = tor.SEED_SAMPLE if args.seeding == "sample" else tor.SEED_PIXEL
    arith = tor.ARITH_STRICT
    accel = ACCEL_BITS[args.accel]
    opt = tor.make_options(seeding=seeding, arith=arith, accel=accel, row_tile=args.row_tile, devices=devices)
    cv = tor.new_canvas(H, W, spp, 2.2)
    for _ in range(args.warmup):
        tor.render(cv, cam, scene.list(), args.depth, opt)
    k_ms_steps = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tor.render(cv, cam, scene.list(), args.depth, opt)
        k_ms_steps.append(tor.last_device_kernel_ms())
    elapsed = time.perf_counter() - t0
    timing = tor.last_render_timing()
    note = tor.last_note()
    info = tor.last_gather_info()
    value = H * W * spp * args.steps / elapsed / 1e6
    # ---- the N = 1 value of this very invocation (same frame, device 0 alone, same host-canvas region) ----
    aux = max(1, min(args.aux_steps, args.steps))
    one = tor.new_canvas(H, W, spp, 2.2)
    opt1 = tor.make_options(seeding=seeding, arith=arith, accel=accel, device=devices[0])
    tor.render(one, cam, scene.list(), args.depth, opt1)
    t1 = time.perf_counter()
    for _ in range(aux):
        tor.render(one, cam, scene.list(), args.depth, opt1)
    dt1 = time.perf_counter() - t1
    n1_value = H * W * spp * aux / dt1 / 1e6
    import numpy as np
    verified = bool(np.array_equal(one.pixels, cv.pixels))
    if args.verify and not verified:
        raise SystemExit("multi-device canvas differs from the single-device canvas")
    # ---- kernel-level roofline: every device's integrate_kernel launch (HIP events on its own stream), averaged over the steps;
    #      the job's kernel phase ends with the slowest device ----
    per_dev = [sum(st[k] for st in k_ms_steps) / len(k_ms_steps) for k in range(N)] if k_ms_steps and all(len(st) == N for st in k_ms_steps) else []
    shard_rows = [len(tor.shard_rows(H, args.row_tile, k, N)) for k in range(N)]
    fps = algorithmic_flops_per_sample(scene, tor)
    roof = {"bound": "valu_fp64", "kernel": "tor::integrate_kernel (one launch per device and step)"}
    if per_dev and max(per_dev) > 0:
        k_max = max(per_dev)
        # one launch per device entry, side by side: the job's kernel phase is the slowest launch.  Entries that REPEAT an ordinal
        # (the single-GPU emulation of an N-GPU job) share that GPU: their launches interleave workgroup by workgroup, each event
        # pair covers only its own launch and none of them spans the phase -- the step time does (kernels are > 99.5 % of it)
        basis = "slowest device's integrate_kernel launch (HIP events)"
        if n_distinct < N:
            k_max = elapsed / args.steps * 1e3
            basis = "step time: the device list repeats an ordinal, the launches share one GPU and no single launch spans the kernel phase"
        roof["kernel_ms_basis"] = basis
        tflops = H * W * spp * fps / (k_max * 1e-3) / 1e12
        peak = PEAK_FP64_VECTOR_TFLOPS * n_distinct
        roof.update({"achieved": round(tflops, 3), "peak": peak, "unit": "TFLOP/s", "frac": round(tflops / peak, 4),
                     "frac_of_nofma_peak": round(tflops / (PEAK_FP64_NOFMA_TFLOPS * n_distinct), 4),
                     "kernel_ms_per_device": [round(m, 3) for m in per_dev], "kernel_ms": round(k_max, 3), "launches_averaged": len(k_ms_steps),
                     "flops_per_sample": round(fps, 1)})
    traffic, traffic_note = None, "skipped (--no-pmc)" if args.no_pmc else None
    if not args.no_pmc:
        traffic, traffic_note = live_traffic(W, H, spp, args.depth, seeding, arith, accel, shard=(0, N, args.row_tile))
    roof["traffic"] = traffic["bytes"] if traffic else None
    roof["traffic_detail"] = traffic if traffic else traffic_note
    roof["traffic_scope"] = f"one launch = shard 0 of {N} ({shard_rows[0]} rows)"
    roof["executed_live"] = executed_from_counters(traffic, shard_rows[0] * W * spp)
    roof["frac_executed"] = roof["executed_live"]["frac_fp64_issue"] if roof["executed_live"] else None
    roof["note"] = ("frac_executed / executed_live (counters of one shard's launch, read in this run) are fractions of a roof; frac is SURVEY 8(d)'s "
                    "reference-formulation operation count over the FMA peak and exceeds 1 by design")
    roof["hbm"] = {"bound": "hbm", "algorithmic_bytes_per_launch": shard_rows[0] * W * 24.0 * (2 if seeding == tor.SEED_SAMPLE else 1) + 64e3,
                   "peak": PEAK_HBM_GBPS, "unit": "GB/s"}
    if per_dev and per_dev[0] > 0:
        roof["hbm"]["achieved"] = round(roof["hbm"]["algorithmic_bytes_per_launch"] / (per_dev[0] * 1e-3) / 1e9, 4)
        roof["hbm"]["frac"] = round(roof["hbm"]["achieved"] / PEAK_HBM_GBPS, 8)
    emit_line(json.dumps({
        "metric": "Msamples/s (pixels×spp/s) on book-1 random_scene", "value": round(value, 2),
        "unit": "Msamples/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{config_name(W, H, spp, N, args.scaling)}: random_scene seed 0xFACADE (485 objects), {W}x{H}, {spp} spp, "
                               f"depth {args.depth}", "seeding": args.seeding, "arith": args.arith, "accel": args.accel,
                   "parallelism": f"ONE process, tor_render_opt with devices={devices}: row tiles of {args.row_tile} dealt to {N} "
                                  f"device contexts, framebuffer gather inside the library ({note})",
                   "timed_region": "SURVEY 8(d): host canvas in/out (scene cached after the first call)"},
        "gather": {"leg": info["leg"], "rccl_ranks": info["rccl_ranks"], "devices": info["devices"], "distinct_gpus": n_distinct, "note": note},
        "gather_leg": info["leg"], "rccl_ranks": info["rccl_ranks"],
        "single_gpu_same_run": {"value": round(n1_value, 2), "unit": "Msamples/s", "steps": aux, "ms_per_step": round(dt1 / aux * 1e3, 3),
                                "speedup": round(value / n1_value, 3), "region": "the same frame and region on devices[0] alone"},
        "last_call_ms": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in timing.items()},
        "canvas_identical_to_single_device": verified, "roofline": roof,
        "cpu_baseline": cpu if cpu is not None else {"value": None, "note": "skipped (--no-cpu-baseline)"}}))


_REAL_STDOUT = None


def claim_stdout():
    """stdout carries ONE line, the JSON line of rank 0: whatever else writes to file descriptor 1 from here on -- the banners RCCL
    and gloo print when a communicator comes up, children -- lands on stderr instead."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit_line(line):
    if _REAL_STDOUT is None:
        print(line, flush=True)
    else:
        sys.stdout.flush()
        os.write(_REAL_STDOUT, (line + "\n").encode())


def main():
    args = parse_args()
    if args.pmc_child:
        return pmc_child(args.pmc_child)
    claim_stdout()
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
    if world == 1 and n_gpus > 1:
        return bench_single_process_multi_device(args, tor)
    spp = frame_spp(args, world)  # strong scaling (default): the frame is the same whatever N
    seeding = tor.SEED_SAMPLE if args.seeding == "sample" else tor.SEED_PIXEL
    arith = tor.ARITH_STRICT
    accel = ACCEL_BITS[args.accel]

    scene = tor.random_scene(0xFACADE)
    cam = tor.camera(aspect_ratio=W / H) if (W * 9 != H * 16) else tor.camera()
    ctx = tor.Context(dev_index)
    ctx.upload(scene.list())
    shard = rank if world > 1 else 0
    opt = tor.make_options(seeding=seeding, arith=arith, shard_index=shard, shard_count=max(world, 1), row_tile=args.row_tile,
                           accel=accel)
    my_rows = tor.shard_rows(H, args.row_tile, shard, max(world, 1))
    stream = torch.cuda.current_stream().cuda_stream
    frame = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")   # world 1: the canvas; world > 1: rank 0's gathered frame

    # ---- N > 1: the framebuffer gather.  Preferred: RCCL inside the library (tor_render_gather_device). ----
    gather_kind = None
    tframe = None
    rccl_ranks = None
    if world > 1:
        gather_kind = "torch.distributed all_gather (RCCL)" if backend == "nccl" else f"torch.distributed all_gather ({backend})"
        lib_ok = False
        if args.gather == "lib" and backend == "nccl":
            try:
                uid = [tor.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                # watchdog: ncclCommInitRank that never returns (the usual failure on a fresh node) must not hang the run --
                # it runs in a helper thread (ctypes releases the GIL) and is abandoned at the deadline
                import threading
                box = {}

                def _init():
                    try:
                        ctx.comm_init_rank(uid[0], rank, world)
                        box["ok"] = True
                    except Exception as e:  # noqa
                        box["err"] = e
                th = threading.Thread(target=_init, daemon=True)
                th.start()
                th.join(float(os.environ.get("TOR_BENCH_RCCL_INIT_S", "120")))
                if th.is_alive():
                    raise RuntimeError("ncclCommInitRank did not return within the deadline")
                if "err" in box:
                    raise box["err"]
                lib_ok = True
            except Exception as e:  # noqa
                print(f"[bench rank {rank}] library RCCL communicator unavailable ({e}); using torch.distributed", file=sys.stderr, flush=True)
            flag = torch.tensor([1 if lib_ok else 0], dtype=torch.int64, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            lib_ok = bool(flag.item())
            if lib_ok:
                # prove the library's gather on a small frame before the clock depends on it: every rank renders its
                # shard of a 64-row frame, rank 0 gathers and compares with the frame it renders alone
                try:
                    th, tw = 64, 96
                    small = torch.zeros((th, tw, 3), dtype=torch.float64, device="cuda")
                    ctx.render_gather_device(cam, th, tw, 2, 2.2, 8, opt, 0, small.data_ptr(), stream)
                    # (a gather that does not complete is aborted at the deadline: poll, do not synchronise)
                    t_dead = time.perf_counter() + float(os.environ.get("TOR_BENCH_RCCL_CHECK_S", "60"))
                    while not torch.cuda.current_stream().query():
                        if time.perf_counter() > t_dead:
                            ctx.comm_abort()
                            raise RuntimeError("the library's RCCL gather did not complete within the deadline; communicator aborted")
                        time.sleep(0.002)
                    torch.cuda.synchronize()
                    ok = 1
                    if rank == 0:
                        alone = torch.zeros_like(small)
                        ctx.render_device(cam, th, tw, 2, 2.2, 8, tor.make_options(seeding=seeding, arith=arith, accel=accel),
                                          alone.data_ptr(), stream)
                        torch.cuda.synchronize()
                        ok = int(torch.equal(alone, small))
                except Exception as e:  # noqa
                    print(f"[bench rank {rank}] library gather failed its self-check ({e}); using torch.distributed", file=sys.stderr, flush=True)
                    ok = 0
                flag = torch.tensor([ok], dtype=torch.int64, device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                lib_ok = bool(flag.item())
                if not lib_ok:
                    ctx.comm_destroy()
        rccl_ranks = None
        if lib_ok:
            gather_kind = "tor_render_gather_device: RCCL send/recv gather to rank 0 inside libtor_mi355x + de-interleave kernel"
            rccl_ranks = ctx.comm_count()     # ncclCommCount of the communicator that carries the framebuffer
        else:
            rccl_ranks = dist.get_world_size() if backend == "nccl" else 0
            tdist = importlib.import_module("trace-of-radiance_amd.distributed")
            tframe = tdist.DistributedFrame(tdist.ShardPlan(H, args.row_tile, world), W, rank, torch.device("cuda"))
            assert list(tframe.my_rows) == list(my_rows)

    def step():
        if world == 1:
            ctx.render_device(cam, H, W, spp, 2.2, args.depth, opt, frame.data_ptr(), stream)
        elif tframe is None:
            ctx.render_gather_device(cam, H, W, spp, 2.2, args.depth, opt, 0, frame.data_ptr(), stream)
        else:
            ctx.render_device(cam, H, W, spp, 2.2, args.depth, opt, tframe.shard.data_ptr(), stream)
            tframe.gather()   # all_gather of the row shards over xGMI + rows put in place

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
    # dominant kernel: integrate_kernel, per-launch HIP events on the launch stream; N > 1: every rank's mean, the job's
    # kernel phase is the slowest rank's
    k_ms, k_n = ctx.kernel_ms_mean(args.steps)
    k_ms_ranks = [k_ms]
    if world > 1:
        allk = torch.zeros(world, dtype=torch.float64, device="cuda")
        allk[rank] = k_ms
        dist.all_reduce(allk, op=dist.ReduceOp.SUM)
        k_ms_ranks = [float(x) for x in allk.cpu()]
    # the N = 1 value of this very invocation: rank 0 renders the whole frame alone (the other ranks wait at the barrier below)
    n1 = None
    if world > 1:
        if rank == 0:
            aux1 = max(1, min(args.aux_steps, args.steps))
            full1 = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
            o1 = tor.make_options(seeding=seeding, arith=arith, accel=accel)
            ctx.render_device(cam, H, W, spp, 2.2, args.depth, o1, full1.data_ptr(), stream)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(aux1):
                ctx.render_device(cam, H, W, spp, 2.2, args.depth, o1, full1.data_ptr(), stream)
            torch.cuda.synchronize()
            dt1 = time.perf_counter() - t1
            n1 = {"value": round(H * W * spp * aux1 / dt1 / 1e6, 2), "unit": "Msamples/s", "steps": aux1, "ms_per_step": round(dt1 / aux1 * 1e3, 3),
                  "kernel_ms": round(ctx.kernel_ms_mean(aux1)[0], 3), "region": "the same frame, resident, on rank 0's GPU alone",
                  "frame_identical_to_gathered": bool(torch.equal(full1, frame if tframe is None else tframe.frame))}
            del full1
        dist.barrier()

    verified = None
    if args.verify and world > 1:
        # the gathered frame must be the frame one process renders alone (any partition, any world size)
        ok = 1
        if rank == 0:
            full = torch.empty((H, W, 3), dtype=torch.float64, device="cuda")
            ctx.render_device(cam, H, W, spp, 2.2, args.depth, tor.make_options(seeding=seeding, arith=arith, accel=accel),
                              full.data_ptr(), stream)
            torch.cuda.synchronize()
            ok = int(torch.equal(full, frame if tframe is None else tframe.frame))
        flag = torch.tensor([ok], dtype=torch.int64, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        verified = bool(flag.item())
        if not verified:
            raise SystemExit(f"rank {rank}: gathered frame differs from the single-process frame")

    local_samples = len(my_rows) * W * spp
    total_samples = H * W * spp
    value = total_samples * args.steps / elapsed / 1e6

    result = None
    if rank == 0:
        # ---- workload counters of this very configuration, measured now (one extra, untimed launch) ----
        live = None
        flops_per_sample = FLOPS_PER_SAMPLE_SURVEY
        if not args.no_stats:
            ctx.set_stats(True)
            one = torch.empty((len(my_rows), W, 3), dtype=torch.float64, device="cuda")
            ctx.render_device(cam, H, W, spp, 2.2, args.depth, opt, one.data_ptr(), stream)
            torch.cuda.synchronize()
            st = ctx.last_stats()
            ctx.set_stats(False)
            n_obj = len(scene)
            n_moving = sum(1 for i in range(n_obj) if scene.objects[i].kind == tor.MOVING_SPHERE)
            q_per_sample = st.hit_queries / max(st.samples, 1)
            ops_per_query = n_moving * OPS_PER_TEST_MOVING + (n_obj - n_moving) * OPS_PER_TEST_STATIC
            flops_per_sample = q_per_sample * ops_per_query + OPS_PER_SAMPLE_FIXED
            live = {
                "what": "tor_last_stats of one extra launch of this configuration (counters in the kernel, untimed)",
                "samples": int(st.samples), "hit_queries_per_sample": round(q_per_sample, 4),
                "object_tests_per_sample": round(q_per_sample * n_obj, 1),
                "candidates_per_query": round(st.candidates / max(st.hit_queries, 1), 3),
                "exact_tests_per_query": round(st.exact_tests / max(st.hit_queries, 1), 3),
                "lane_utilisation": round(st.hit_queries / max(st.lane_slots, 1), 4),
                "algorithmic_fp64_ops_per_sample": round(flops_per_sample, 1),
                "formula": f"queries/sample x ({n_moving} moving x 35 + {n_obj - n_moving} static x 23 float64 ops, the reference's "
                           f"un-hoisted formulation, SURVEY 8d) + 500",
            }
            del one
        k_rate = local_samples / (k_ms * 1e-3)          # samples/s inside the kernel, this rank
        if world > 1:                                   # N > 1: the whole frame over the slowest rank's kernel, against N GPUs
            k_rate = total_samples / (max(k_ms_ranks) * 1e-3) / world
        tflops = k_rate * flops_per_sample / 1e12
        hbm_bytes = len(my_rows) * W * 24.0 * (2 if seeding == tor.SEED_SAMPLE else 1) + 64e3  # canvas write (+ clear in SAMPLE mode) + scene
        traffic, traffic_note = None, "skipped (--no-pmc)" if args.no_pmc else None
        if world == 1 and not args.no_pmc:
            traffic, traffic_note = live_traffic(W, H, spp, args.depth, seeding, arith, accel)
        roof = {
            "bound": "valu_fp64", "kernel": "tor::integrate_kernel",
            "achieved": round(tflops, 3), "peak": PEAK_FP64_VECTOR_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tflops / PEAK_FP64_VECTOR_TFLOPS, 4),
            "frac_of_nofma_peak": round(tflops / PEAK_FP64_NOFMA_TFLOPS, 4),
            "nofma_peak": PEAK_FP64_NOFMA_TFLOPS,
            "flops_per_sample": round(flops_per_sample, 1), "kernel_ms": round(max(k_ms_ranks), 3), "launches_averaged": k_n,
            "kernel_ms_per_rank": [round(x, 3) for x in k_ms_ranks] if world > 1 else None,
            "scope": "per GPU (achieved and peak are one device's; N > 1: the frame's samples / N over the slowest rank's kernel time)",
            "traffic": traffic["bytes"] if traffic else None,
            "traffic_detail": traffic if traffic else traffic_note,
            "workload_live": live,
            # the EXECUTED-instruction view, measured in this run (the SQ / GRBM passes of live_traffic): the fraction of a roof that
            # `frac` -- the reference formulation's operation count over the kernel time, > 1 since the screens -- no longer is
            "executed_live": executed_from_counters(traffic, local_samples) if world == 1 else None,
            "from_profile": from_profile(W, H, spp, args.seeding, args.arith, args.accel) if world == 1 else None,
            "hbm": {"bound": "hbm", "achieved": round(hbm_bytes / (k_ms * 1e-3) / 1e9, 4), "peak": PEAK_HBM_GBPS,
                    "unit": "GB/s", "frac": round(hbm_bytes / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 8),
                    "algorithmic_bytes_per_launch": hbm_bytes},
            "note": "WHICH NUMBER IS A FRACTION OF A ROOF: `frac_executed` (= executed_live.frac_fp64_issue: float64 lane-operations the "
                    "kernel really issued per second over the 39.3 T/s float64 issue peak) and executed_live.valu_issue_util (share of the SIMD "
                    "cycles with a VALU instruction in flight), both from counters read in this run.  `frac` is SURVEY 8(d)'s figure -- samples/s x "
                    "the operations the REFERENCE's formulation would spend (41.9 k per sample, nothing hoisted or fused) over the FMA peak -- "
                    "which the survey itself says exceeds 1 once work is removed; it prices the throughput in the reference's units, it is "
                    "not a utilisation.  The path is float64-VALU bound by design (scene in SGPRs via the scalar cache, 24 B of HBM per pixel).  "
                    "The reference's arithmetic has no FMA "
                    "(0.5 of the FMA peak would be its ceiling); the object loop therefore SCREENS every ray x object pair with a "
                    "conservative float64 FMA form of the same geometry (csrc/tor_screen.hpp: 4 instructions per object for the distance to the "
                    "vertical plane through the ray, then the expanded quadratic -- 7 to 10 -- on the ~9 objects per query that keeps) and runs "
                    "the reference's unfused operations only on the candidates -- same canvas bit for bit (`unscreened` = without any of "
                    "it, `second_form_only` = without the plane stage)",
        }
        roof["frac_executed"] = roof["executed_live"]["frac_fp64_issue"] if roof.get("executed_live") else None
        if args.accel != "none":
            # with an exact acceleration the rate is still quoted against the reference's brute-force
            # float64 operation count, so frac can exceed 1
            roof["note"] = ("--accel " + args.accel + ": achieved = samples/s x the reference's brute-force float64 "
                            "operation count (SURVEY 8d), not executed work -- the float32 pre-filter issues packed "
                            "float32 (peak 157.3 TFLOP/s) and the block culling skips tests")
            result_dtype = "f64 (canvas bit-identical to the float64 path)"
        else:
            result_dtype = "f64"
        result = {
            "metric": "Msamples/s (pixels×spp/s) on book-1 random_scene", "value": round(value, 2), "unit": "Msamples/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": result_dtype, "data": "synthetic",
            "config": {"workload": f"{config_name(W, H, spp, max(world, 1), args.scaling)}: random_scene seed 0xFACADE (485 objects), {W}x{H}, "
                                   f"{spp} spp, depth {args.depth}",
                       "seeding": args.seeding, "arith": args.arith, "accel": args.accel,
                       "object_loop": ("every ray x every object in float64; conservative FMA screen in two stages (plane through the ray, then the "
                                       "expanded quadratic), candidates re-tested with the reference's unfused operations (TOR_SCREEN=0 turns "
                                       "the screen off, TOR_PLANE=0 its first stage)") if (args.accel == "none" and args.arith == "strict"
                                                                                                     and os.environ.get("TOR_SCREEN", "1") != "0") else "see accel / arith",
                       "timed_region": "`value`: scene + camera resident in HBM when the clock starts, the frame stays on the device -- the harness "
                                       "contract's region (inputs resident; a PCIe-inclusive rate is never `value`).  SURVEY 8(d)'s region -- what a Nim caller "
                                       "of render() pays, trace_of_radiance.nim:60-64: scene (cached) + camera H2D, kernels, D2H into canvas.pixels -- "
                                       "is `value_survey_8d` / `host_canvas`, measured in the same run over the same frame",
                       "parallelism": f"row tiles of {args.row_tile} dealt to {max(world, 1)} rank(s)" +
                                      (f" + {gather_kind}" if world > 1 else "")},
            "roofline": roof,
        }
        if verified is not None:
            result["gathered_frame_identical_to_single_process"] = verified
        if world > 1:
            result["rccl_ranks"] = rccl_ranks
            result["gather"] = {"kind": gather_kind, "rccl_ranks": rccl_ranks, "world": world}
            # at a glance: which leg moved the shards -- "rccl" = the library's own communicator (send/recv gather
            # inside libtor_mi355x), "torch" = the fall-back through torch.distributed's all_gather (RCCL as well under the nccl backend)
            result["gather_leg"] = "rccl" if (gather_kind or "").startswith("tor_render_gather_device") else "torch"
            if n1 is not None:
                n1["speedup"] = round(value / n1["value"], 3)
                result["single_gpu_same_run"] = n1
    aux = max(1, min(args.aux_steps, args.steps))
    if rank == 0 and world == 1 and not args.no_host_leg:
        # ---- SURVEY 8(d)'s region: what a Nim caller of render() pays (trace_of_radiance.nim:60-64) ----
        import numpy as np
        cv = tor.new_canvas(H, W, spp, 2.2)
        hopt = tor.make_options(seeding=seeding, arith=arith, accel=accel, device=dev_index)
        tor.render(cv, cam, scene.list(), args.depth, hopt)          # first call: context, scene upload, pinned staging
        first = tor.last_render_timing()
        t1 = time.perf_counter()
        for _ in range(aux):
            tor.render(cv, cam, scene.list(), args.depth, hopt)
        dth = time.perf_counter() - t1
        last = tor.last_render_timing()
        same = bool(np.array_equal(cv.pixels, frame.cpu().numpy()))
        hv = total_samples * aux / dth / 1e6
        result["value_survey_8d"] = round(hv, 2)
        result["host_canvas"] = {
            "value": round(hv, 2), "unit": "Msamples/s", "steps": aux, "ms_per_step": round(dth / aux * 1e3, 3),
            "region": "tor_render_opt on a host canvas: scene (cached) + camera H2D, kernels, D2H into canvas.pixels "
                      "(SURVEY 8d; trace_of_radiance.nim:60-64)",
            "vs_resident": round(hv / value, 4), "canvas_identical_to_resident_frame": same,
            "first_call_ms": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in first.items()},
            "steady_call_ms": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in last.items()},
        }
    if rank == 0 and world == 1 and args.accel == "none" and not args.no_accel_leg:
        # secondary legs (never the metric's value): the same frame with the exact accelerations -- bit-identical
        # canvas (tests/test_gpu_parity.py::test_block_culling_never_changes_a_pixel), checked here again
        ref_frame = frame.clone()
        if arith == tor.ARITH_STRICT:
            # the same brute force WITHOUT the conservative FMA screen (csrc/tor_screen.hpp): every object through the
            # reference's unfused discriminant (rounds 1-2) -- a second context, the knob is read when a context is made
            os.environ["TOR_SCREEN"] = "0"
            ctx0 = tor.Context(dev_index)
            os.environ.pop("TOR_SCREEN", None)
            ctx0.upload(scene.list())
            ctx0.render_device(cam, H, W, spp, 2.2, args.depth, opt, frame.data_ptr(), stream)
            torch.cuda.synchronize()
            same = bool(torch.equal(ref_frame, frame))
            t1 = time.perf_counter()
            for _ in range(aux):
                ctx0.render_device(cam, H, W, spp, 2.2, args.depth, opt, frame.data_ptr(), stream)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            result["unscreened"] = {
                "value": round(total_samples * aux / dt2 / 1e6, 2), "unit": "Msamples/s", "steps": aux, "ms_per_step": round(dt2 / aux * 1e3, 3),
                "canvas_identical_to_value_frame": same,
                "note": "TOR_SCREEN=0: 17 / 19 / 23 unfused float64 operations per ray x object in the wave-uniform loop (the reference's "
                        "discriminant as written) instead of the conservative FMA screen; candidates get the reference's exact test either way"}
            ctx0.close()
            # ... and with the screen's second form on EVERY object (round 4 before the plane stage)
            os.environ["TOR_PLANE"] = "0"
            ctx1 = tor.Context(dev_index)
            os.environ.pop("TOR_PLANE", None)
            ctx1.upload(scene.list())
            ctx1.render_device(cam, H, W, spp, 2.2, args.depth, opt, frame.data_ptr(), stream)
            torch.cuda.synchronize()
            same = bool(torch.equal(ref_frame, frame))
            t1 = time.perf_counter()
            for _ in range(aux):
                ctx1.render_device(cam, H, W, spp, 2.2, args.depth, opt, frame.data_ptr(), stream)
            torch.cuda.synchronize()
            dt3 = time.perf_counter() - t1
            result["second_form_only"] = {
                "value": round(total_samples * aux / dt3 / 1e6, 2), "unit": "Msamples/s", "steps": aux, "ms_per_step": round(dt3 / aux * 1e3, 3),
                "canvas_identical_to_value_frame": same,
                "note": "TOR_PLANE=0: the screen's second form (7 / 9 / 10 instructions per common-height static / static / mover along y) on "
                        "every ray x object, without the 4-instruction plane screen in front of it"}
            ctx1.close()
        notes = {"f32": "every ray x every object, through the conservative packed-float32 pre-filter first "
                        "(tor_filter32.hpp); kept objects get the reference's float64 test",
                 "blocks": "SURVEY 8 f4: spatial blocks of 8 objects behind conservative boxes",
                 "blocks+f32": "both"}
        for name in ("f32", "blocks", "blocks+f32"):
            opt2 = tor.make_options(seeding=seeding, arith=arith, row_tile=args.row_tile, accel=ACCEL_BITS[name])
            ctx.render_device(cam, H, W, spp, 2.2, args.depth, opt2, frame.data_ptr(), stream)
            torch.cuda.synchronize()
            same = bool(torch.equal(ref_frame, frame))
            t1 = time.perf_counter()
            for _ in range(aux):
                ctx.render_device(cam, H, W, spp, 2.2, args.depth, opt2, frame.data_ptr(), stream)
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            leg = {
                "value": round(total_samples * aux / dt2 / 1e6, 2), "unit": "Msamples/s", "steps": aux,
                "ms_per_step": round(dt2 / aux * 1e3, 3), "kernel_ms": round(ctx.kernel_ms_mean(aux)[0], 3), "canvas_identical_to_brute_force": same,
                "note": notes[name] + "; the metric's value above is the reference's float64 brute-force closest hit"}
            if name == "blocks+f32":
                # what this leg's kernel executes, read NOW (round 6; rounds 4-5 replayed a committed profile here): the same four
                # rocprofv3 --pmc passes as the metric's kernel, and the kernel's own workload counters (one extra untimed launch)
                if not args.no_pmc:
                    tr2, note2 = live_traffic(W, H, spp, args.depth, seeding, arith, ACCEL_BITS[name])
                    leg["executed_live"] = executed_from_counters(tr2, total_samples) if tr2 else None
                    leg["traffic"] = tr2["bytes"] if tr2 else None
                    if not tr2:
                        leg["executed_note"] = note2
                if not args.no_stats:
                    ctx.set_stats(True)
                    ctx.render_device(cam, H, W, spp, 2.2, args.depth, opt2, frame.data_ptr(), stream)
                    torch.cuda.synchronize()
                    st2 = ctx.last_stats()
                    ctx.set_stats(False)
                    q2 = max(st2.hit_queries, 1)
                    leg["workload_live"] = {"hit_queries_per_sample": round(st2.hit_queries / max(st2.samples, 1), 4),
                                            "blocks_entered_per_query": round(st2.block_tests / 8.0 / q2, 3),
                                            "float32_filter_tests_per_query": round(st2.block_tests / q2, 3),
                                            "exact_tests_per_query": round(st2.exact_tests / q2, 3),
                                            "lane_utilisation": round(st2.hit_queries / max(st2.lane_slots, 1), 4)}
            else:
                leg["from_profile"] = from_profile(W, H, spp, args.seeding, args.arith, name) or None
            result["accel_" + name.replace("+", "_")] = leg
        if args.seeding == "sample":
            # the reference's own stream layout (what tor_render() runs by default), brute force and with its default accelerations
            for name, bits in (("pixel_seeding", 0), ("pixel_seeding_default_accel", 3)):
                opt2 = tor.make_options(seeding=tor.SEED_PIXEL, arith=arith, row_tile=args.row_tile, accel=bits)
                ctx.render_device(cam, H, W, spp, 2.2, args.depth, opt2, frame.data_ptr(), stream)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(aux):
                    ctx.render_device(cam, H, W, spp, 2.2, args.depth, opt2, frame.data_ptr(), stream)
                torch.cuda.synchronize()
                dt2 = time.perf_counter() - t1
                result[name] = {"value": round(total_samples * aux / dt2 / 1e6, 2), "unit": "Msamples/s", "steps": aux,
                                "ms_per_step": round(dt2 / aux * 1e3, 3),
                                "note": "TOR_SEED_PIXEL (render.nim:59-67 streams), " +
                                        ("float64 brute force" if bits == 0 else "TOR_ACCEL_BLOCKS|TOR_ACCEL_F32 = tor_render()'s default")}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and world > 1 and not args.no_pmc:
        # HBM traffic of ONE launch of this run's kernel -- rank 0's shard -- now that the job is over and GPU 0 is free
        traffic, traffic_note = live_traffic(W, H, spp, args.depth, seeding, arith, accel, shard=(0, world, args.row_tile))
        result["roofline"]["traffic"] = traffic["bytes"] if traffic else None
        result["roofline"]["traffic_detail"] = traffic if traffic else traffic_note
        result["roofline"]["traffic_scope"] = f"one launch = rank 0's shard ({len(my_rows)} of {H} rows)"
        result["roofline"]["executed_live"] = executed_from_counters(traffic, local_samples)
        result["roofline"]["frac_executed"] = result["roofline"]["executed_live"]["frac_fp64_issue"] if result["roofline"]["executed_live"] else None
    if rank == 0:
        # (N > 1: after the process group is gone -- the other ranks have left, the host cores are free again)
        result["cpu_baseline"] = (cpu_baseline(W, H, args.spp, args.depth, args.cpu_seconds) if not args.no_cpu_baseline
                                  else {"value": None, "note": "skipped (--no-cpu-baseline)"})
        emit_line(json.dumps(result))


if __name__ == "__main__":
    main()
