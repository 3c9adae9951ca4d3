#!/usr/bin/env python3
"""Where do the HBM bytes of a bench step come from?  

Runs ONE configuration of the integrator under rocprofv3 -- one `--pmc` pass per counter group, each in its own child
process, kernel trace in a pass of its own -- and prints, per kernel of the step (canvas clear, cost probe, integrate_kernel,
finalize_kernel ...), the memory-side request counters of the L2:

    FETCH_SIZE, WRITE_SIZE                          (KiB; what bench.py's live `roofline.traffic` is made of)
    TCC_EA0_WRREQ_sum, TCC_EA0_WRREQ_64B_sum        (write requests to the fabric; 64-byte ones)
    TCC_EA0_ATOMIC_sum, TCC_ATOMIC_sum              (atomics forwarded to the memory side; atomics seen by the L2)
    TCC_EA0_RDREQ_sum, TCC_EA0_RDREQ_32B_sum
    TCC_HIT_sum, TCC_MISS_sum, TCC_WRITEBACK_sum

in three situations:
    cold      the first tor_render() of a process on a host canvas  (what bench.py's live_traffic() profiled in round 2)
    warm      the second identical tor_render() of the same process (what every timed step of the host-canvas leg is)
    resident  tor_render_device() into a device buffer, second call  (the bench's `value` region; tools/profile_gpu.sh)

    python tools/traffic_reconcile.py [--spp 1000] [--seeding sample] [--accel 0] [--out gpurun_out/r3_traffic_reconcile.txt]
"""
import argparse
import ctypes as C
import glob
import importlib
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = [["FETCH_SIZE"], ["WRITE_SIZE"], ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"], ["TCC_EA0_ATOMIC_sum", "TCC_ATOMIC_sum"],
          ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"], ["TCC_HIT_sum", "TCC_MISS_sum"], ["TCC_WRITEBACK_sum", "TCC_EA0_WRREQ_WRITE_DRAM_sum"]]


def child(spec):
    mode, W, H, spp, depth, seeding, accel = spec.split(",")
    W, H, spp, depth, seeding, accel = (int(x) for x in (W, H, spp, depth, seeding, accel))
    os.environ["TOR_NO_TORCH"] = "1"
    sys.path.insert(0, ROOT)
    tor = importlib.import_module("trace-of-radiance_amd")
    scene, cam = tor.random_scene(0xFACADE), tor.camera()
    opt = tor.make_options(seeding=seeding, accel=accel)
    if mode in ("cold", "warm"):
        cv = tor.new_canvas(H, W, spp, 2.2)
        for _ in range(1 if mode == "cold" else 2):
            tor.render(cv, cam, scene.list(), depth, opt)
        print("child done", float(cv.pixels.mean()))
    else:
        hip = C.CDLL("libamdhip64.so")
        ptr = C.c_void_p()
        assert hip.hipMalloc(C.byref(ptr), C.c_size_t(H * W * 24)) == 0
        ctx = tor.Context(0)
        ctx.upload(scene.list())
        for _ in range(2):
            ctx.render_device(cam, H, W, spp, 2.2, depth, opt, ptr.value, 0)
            assert hip.hipDeviceSynchronize() == 0
        print("child done (resident)")


def collect(exe, spec, counters, trace=False):
    base = tempfile.mkdtemp(prefix="tor_rec_", dir="/tmp")
    try:
        cmd = [exe] + (["--kernel-trace", "--stats"] if trace else ["--pmc"] + counters) + ["-d", base, "-o", "p", "--", sys.executable,
                                                                                             os.path.abspath(__file__), "--child", spec]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", TOR_NO_TORCH="1"))
        if r.returncode != 0:
            return None, r.stderr[-400:]
        dbs = glob.glob(os.path.join(base, "**", "*.db"), recursive=True)
        if not dbs:
            return None, "no database"
        con = sqlite3.connect(dbs[0])
        if trace:
            rows = con.execute("select name, start, duration from kernels order by start").fetchall()
            return [(n, d / 1e6) for n, _, d in rows], None
        rows = con.execute("select dispatch_id, name, counter_name, sum(counter_value) from pmc_events group by dispatch_id, name, counter_name "
                           "order by dispatch_id").fetchall()
        return rows, None
    finally:
        shutil.rmtree(base, ignore_errors=True)


def short(name):
    for key in ("integrate_kernel", "finalize_kernel", "tile_key_kernel", "tile_order_kernel", "coop_pixel_kernel", "gather_rows_kernel", "fillBuffer", "FillBuffer"):
        if key in name:
            if key == "integrate_kernel":
                return "integrate_kernel" + name[name.index("<"):name.index(">") + 1].replace(" ", "") if "<" in name else key
            return key
    return name[:40]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", default=None)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1000)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--seeding", choices=["sample", "pixel"], default="sample")
    ap.add_argument("--accel", type=int, default=0)
    ap.add_argument("--modes", default="cold,warm,resident")
    ap.add_argument("--out", default="gpurun_out/r3_traffic_reconcile.txt")
    args = ap.parse_args()
    if args.child:
        return child(args.child)
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    seeding = 1 if args.seeding == "sample" else 0
    lines = [f"# tools/traffic_reconcile.py: {args.width}x{args.height}x{args.spp} spp, depth {args.depth}, seeding {args.seeding}, accel {args.accel}",
             f"# algorithmic HBM bytes of integrate_kernel: {args.width * args.height * 24 / 1e6:.1f} MB (one 24-byte sum per pixel)"
             + (f" + {args.width * args.height * 24 / 1e6:.1f} MB canvas clear in front of it (SEED_SAMPLE)" if seeding else ""),
             "# one rocprofv3 pass (own process) per counter group; values summed over the L2 instances; per dispatch, in launch order"]
    for mode in args.modes.split(","):
        spec = f"{mode},{args.width},{args.height},{args.spp},{args.depth},{seeding},{args.accel}"
        lines.append("")
        lines.append(f"## {mode}")
        trace, err = collect(exe, spec, [], trace=True)
        if trace is None:
            lines.append(f"(kernel trace failed: {err})")
        else:
            lines.append("kernel trace: " + "; ".join(f"{short(n)} {ms:.3f} ms" for n, ms in trace))
        table = {}
        order = []
        for grp in GROUPS:
            rows, err = collect(exe, spec, grp)
            if rows is None:
                lines.append(f"({'+'.join(grp)} failed: {err})")
                continue
            # dispatch ids differ between passes only by a constant; index dispatches by their order inside the pass
            ids = []
            for did, name, cname, val in rows:
                if did not in ids:
                    ids.append(did)
                key = (ids.index(did), short(name))
                if key not in table:
                    table[key] = {}
                    order.append(key)
                table[key][cname] = val
        cols = [c for g in GROUPS for c in g]
        lines.append("dispatch | kernel | " + " | ".join(cols) + " | bytes = (2*FETCH + WRITE) KiB")
        for key in sorted(set(order)):
            t = table[key]
            b = (2 * t.get("FETCH_SIZE", 0) + t.get("WRITE_SIZE", 0)) * 1024
            lines.append(f"{key[0]} | {key[1]} | " + " | ".join(f"{t.get(c, float('nan')):.6g}" for c in cols) + f" | {b / 1e6:.1f} MB")
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
