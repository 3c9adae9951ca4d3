"""Rebuild profiles/traffic.json (what bench.py reports as roofline.traffic / roofline.executed) from the
committed rocprofv3 summaries, so the numbers cannot drift from the evidence.

    python tools/update_traffic.py
"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# key in traffic.json -> (summary file, the integrate_kernel variant <SEEDING, ARITH, WAVES_PER_SIMD, F32, BLOCKS> that mode launches:
# strict brute force runs behind the FMA screen = ARITH 2; the exact accelerations keep ARITH 0).  A summary whose dominant
# integrate_kernel is another variant is an ERROR (bench.py must not replay counters of a kernel it does not run).
CONFIGS = {
    "1920x1080x100:sample:strict": ("r6_c2_summary.txt", (1, 2, 3, 0, 0)),
    "1920x1080x100:pixel:strict": ("r6_c2_pixel_brute_summary.txt", (0, 2, 3, 0, 0)),
    "1920x1080x1000:sample:strict": ("r6_c3_summary.txt", (1, 2, 3, 0, 0)),
    "1920x1080x100:sample:strict:f32": ("r6_c2_f32_summary.txt", (1, 0, 3, 1, 0)),
    "1920x1080x100:sample:strict:blocks": ("r6_c2_blocks_summary.txt", (1, 0, 3, 0, 1)),
    "1920x1080x100:sample:strict:blocks+f32": ("r6_c2_accel3_summary.txt", (1, 0, 3, 1, 1)),
    "1920x1080x1000:sample:strict:blocks+f32": ("r6_c3_accel3_summary.txt", (1, 0, 3, 1, 1)),
    "1920x1080x100:pixel:strict:blocks+f32": ("r6_c2_pixel_default_summary.txt", (0, 0, 3, 1, 1)),
    "1920x1080x1000:pixel:strict:blocks+f32": ("r6_c3_pixel_default_summary.txt", (0, 0, 3, 1, 1)),
}


def counters(path, variant):
    """Counter totals of the launch of `variant` in a summary (the frame's launch: the longest dispatch of that variant)."""
    out = {}
    want = "integrate_kernel<%d, %d, %d, %d, %d>" % variant
    pat = re.compile(r"(integrate_kernel<[0-9, ]+>).*?\| ([A-Z][A-Z0-9_]+) \| \d+ \| ([0-9.e+]+) \| ([0-9.]+)")
    seen = {}
    for line in open(path):
        m = pat.search(line)
        if not m:
            continue
        name, counter, value, ms = m.group(1), m.group(2), float(m.group(3)), float(m.group(4))
        seen[name] = max(seen.get(name, 0.0), ms)
        if name == want and ms >= out.get("_ms_" + counter, 0.0):
            out[counter] = value
            out["_ms_" + counter] = ms
    if not seen:
        raise SystemExit(f"{path}: no integrate_kernel dispatch in the summary")
    dominant = max(seen, key=seen.get)
    if dominant != want:
        raise SystemExit(f"{path}: the dominant kernel is {dominant}, but this mode launches {want} -- re-profile (tools/profile_round6.sh)")
    out["_ms"] = seen[want]
    return {k: v for k, v in out.items() if not k.startswith("_ms_")}


def main():
    res = {"_note": "HBM bytes per integrate_kernel launch and executed-instruction counters from rocprofv3 PMC passes "
                    "(separate --pmc runs, one launch each; profiles/*_summary.txt; regenerate with tools/update_traffic.py). "
                    "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE reports half of a wide coalesced read; "
                    "WRITE_SIZE uncorrected). simd_cycles = GRBM_GUI_ACTIVE (summed over the 8 XCDs) * 128 SIMDs per XCD."}
    for key, (fname, variant) in CONFIGS.items():
        path = os.path.join(ROOT, "profiles", fname)
        if not os.path.exists(path):
            raise SystemExit(f"profiles/{fname} is missing (tools/profile_round6.sh makes it)")
        c = counters(path, variant)
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
            continue
        e = {"fetch_size_kb": c["FETCH_SIZE"], "write_size_kb": c["WRITE_SIZE"],
             "bytes": (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0, "kernel": "tor::integrate_kernel<%d, %d, %d, %d, %d>" % variant,
             "kernel_ms": c.get("_ms")}
        if "SQ_INSTS_VALU" in c and "GRBM_GUI_ACTIVE" in c:
            simd = c["GRBM_GUI_ACTIVE"] * 128.0
            ex = {"valu_wave_instructions": c["SQ_INSTS_VALU"],
                  "valu_active_quad_cycles": c.get("SQ_ACTIVE_INST_VALU"),
                  "simd_cycles": simd,
                  "valu_active_frac": round(4.0 * c["SQ_ACTIVE_INST_VALU"] / simd, 4) if "SQ_ACTIVE_INST_VALU" in c else None,
                  "fp64_add_mul_fma_wave_instructions": sum(c.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64")),
                  "f32_add_mul_fma_wave_instructions": sum(c.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32")),
                  "scalar_cache_hit_rate": round(c["SQC_DCACHE_HITS"] / (c["SQC_DCACHE_HITS"] + c["SQC_DCACHE_MISSES"]), 4) if "SQC_DCACHE_MISSES" in c else None,
                  "source": "profiles/" + fname}
            e["executed"] = ex
        res[key] = e
    with open(os.path.join(ROOT, "profiles", "traffic.json"), "w") as f:
        json.dump(res, f, indent=1)
    for k, v in res.items():
        if k != "_note":
            print(k, {a: b for a, b in v.items() if a != "executed"}, (v.get("executed") or {}).get("valu_active_frac"))


if __name__ == "__main__":
    main()
