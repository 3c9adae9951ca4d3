"""Summarise a rocprofv3 PC-sampling run (csv): samples of integrate_kernel by instruction class, by stall / issue state when the
columns exist, and by source line (the instruction comments of a -gline-tables-only build).
usage: python tools/pc_sample_summary.py <rocprofv3 output dir> <summary.txt>"""
import collections, csv, glob, os, re, sys
src, out = sys.argv[1], sys.argv[2]
files = [f for f in glob.glob(os.path.join(src, "**", "*.csv"), recursive=True) if "pc_sampling" in os.path.basename(f)]
lines = []
def P(*a):
    lines.append(" ".join(str(x) for x in a))
if not files:
    P("no pc_sampling csv under", src, ":", [os.path.basename(f) for f in glob.glob(os.path.join(src, "**", "*"), recursive=True)][:20])
    open(out, "w").write("\n".join(lines) + "\n"); print(lines[0]); sys.exit(0)
def klass(ins):
    op = ins.split()[0] if ins else "?"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "lane <-> scalar moves (SGPR spills, readfirstlane)"
    if op.startswith("v_") and "_f64" in op and op.startswith(("v_fma", "v_fmac", "v_mul", "v_add", "v_sub")): return "float64 add / mul / fma"
    if op.startswith("v_") and "_f64" in op: return "float64 other (div / sqrt / rcp steps, cmp, cvt, min / max, ldexp, frexp)"
    if op.startswith("v_pk_"): return "packed float32"
    if op.startswith("v_") and "_f32" in op: return "float32"
    if op.startswith(("v_mov", "v_accvgpr", "v_swap")): return "moves"
    if op.startswith("v_cndmask"): return "selects"
    if op.startswith(("v_alignbit", "v_bitop3", "v_bfe", "v_bfi", "v_lshl", "v_lshr", "v_ashr", "v_and", "v_or", "v_xor", "v_not", "v_perm")): return "bit / shift / logic"
    if op.startswith(("v_add", "v_sub", "v_mul", "v_mad", "v_addc", "v_subb", "v_mbcnt", "v_min", "v_max", "v_ffb", "v_bcnt")): return "integer arithmetic"
    if op.startswith("v_cmp"): return "integer compares"
    if op.startswith("v_"): return "other vector"
    if op.startswith("ds_"): return "LDS"
    if op.startswith(("global_", "flat_", "buffer_")): return "vector memory"
    if op.startswith("scratch_"): return "scratch (VGPR spills)"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith(("s_cbranch", "s_branch")): return "branches"
    if op.startswith("s_load"): return "scalar loads"
    if op.startswith("s_"): return "other scalar"
    return "other"
for f in files:
    rows = list(csv.DictReader(open(f)))
    if not rows:
        continue
    cols = list(rows[0].keys())
    P("file", os.path.basename(f), "rows", len(rows)); P("columns", cols)
    ins_col = next((c for c in cols if c.lower() == "instruction"), None)
    cm_col = next((c for c in cols if "comment" in c.lower()), None)
    if not ins_col:
        continue
    issued_col = next((c for c in cols if c.lower() in ("wave_issued", "wave_issued_instruction")), None)
    by_class, by_class_issued, by_line, by_op = collections.Counter(), collections.Counter(), collections.Counter(), collections.Counter()
    extra = {c: collections.Counter() for c in cols if c.lower() in ("instruction_type", "stall_reason", "wave_issued", "snapshot_dual_issue_valu", "arb_state_issue_valu", "arb_state_stall_valu")}
    n = 0
    for r in rows:
        ins = r[ins_col]
        n += 1
        k = klass(ins)
        by_class[k] += 1
        by_op[ins.split()[0] if ins else "?"] += 1
        if issued_col and r[issued_col] in ("1", "true", "True"):
            by_class_issued[k] += 1
        if cm_col and r[cm_col]:
            m = re.search(r"([\w./-]+\.(?:hip|inc|hpp|h)):(\d+)", r[cm_col])
            by_line[(os.path.basename(m.group(1)), int(m.group(2))) if m else ("?", 0)] += 1
        for c, ctr in extra.items():
            ctr[r[c]] += 1
    P(f"samples {n}")
    P("--- by instruction class (share of samples; issued = the sampled wave issued that instruction in the sampled cycle)")
    for k, v in by_class.most_common():
        P(f"  {k:75s} {v / n:7.4f}   issued {by_class_issued.get(k, 0) / max(n, 1):7.4f}")
    for c, ctr in extra.items():
        P(f"--- {c}:", {k: round(v / n, 4) for k, v in ctr.most_common(12)})
    P("--- top opcodes")
    for k, v in by_op.most_common(40):
        P(f"  {k:34s} {v / n:7.4f}")
    if by_line:
        byfile = collections.Counter()
        for (fn, ln), v in by_line.items():
            byfile[fn] += v
        P("--- by source file:", {k: round(v / n, 4) for k, v in byfile.most_common()})
        P("--- top source lines")
        for (fn, ln), v in by_line.most_common(70):
            P(f"  {fn:36s}:{ln:5d} {v / n:7.4f}")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:60]))
