"""Static instruction counts of one integrate_kernel variant PER LOOP of the compiler's own loop nest (the `Loop Header` / `in Loop`
annotations LLVM leaves in the ISA: make -C trace-of-radiance_amd/csrc asm), by instruction class -- and, with --weights FILE, a
dynamic estimate per bounce iteration: every loop's count times its measured trips (rocprofv3 PC sampling is not available on this
pool's boxes, so this is the only per-class view there is; the total is checked against SQ_INSTS_VALU of the same launch).
usage: python tools/isa_dynamic.py [name part] [--weights tools/isa_weights_<variant>.json] [--blocks] [--lines <.s built with -gline-tables-only>]
--lines: where the SPILL instructions sit -- v_readlane / v_writelane (scalar registers spilled to VGPR lanes) and scratch loads / stores
(VGPRs spilled to memory) by loop depth and source file, from the .loc tables of a line-table build of the same source
(hipcc <CXXFLAGS of csrc/Makefile> -gline-tables-only -S --cuda-device-only -o /tmp/tor_kernels_g.s tor_kernels.hip)."""
import collections, json, os, re, sys
path = os.environ.get("TOR_ASM") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "trace-of-radiance_amd", "lib", "asm", "tor_kernels.s")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
want = args[0] if args else "Li1ELi2ELi3ELi0ELi0E"
wfile = sys.argv[sys.argv.index("--weights") + 1] if "--weights" in sys.argv else None
text = open(path).read().split("\n")
i0 = next(i for i, l in enumerate(text) if re.match(r"^_ZN3tor16integrate_kernel\w*:", l) and want in l)
i1 = next(j for j in range(i0, len(text)) if "s_endpgm" in text[j])
body = text[i0:i1 + 1]

CLASSES = ["f64 add/mul/fma", "f64 other", "f32 / packed f32", "int arith", "bit/shift/logic", "compare", "select", "move", "lane<->scalar (SGPR spill, readfirstlane)",
           "other valu", "lds", "vmem", "scratch (VGPR spill)", "salu", "smem", "branch", "waitcnt"]
def klass(op):
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "lane<->scalar (SGPR spill, readfirstlane)"
    if op.startswith("v_") and "_f64" in op:
        return "f64 add/mul/fma" if op.startswith(("v_fma_f64", "v_fmac_f64", "v_mul_f64", "v_add_f64")) else ("compare" if op.startswith("v_cmp") else "f64 other")
    if op.startswith("v_pk_") or (op.startswith("v_") and ("_f32" in op or "_f16" in op)): return "compare" if op.startswith("v_cmp") else "f32 / packed f32"
    if op.startswith("v_cmp"): return "compare"
    if op.startswith("v_cndmask"): return "select"
    if op.startswith(("v_mov", "v_accvgpr", "v_swap")): return "move"
    if op.startswith(("v_alignbit", "v_bitop3", "v_bfe", "v_bfi", "v_lshl", "v_lshr", "v_ashr", "v_and", "v_or", "v_xor", "v_not", "v_perm", "v_bfm")): return "bit/shift/logic"
    if op.startswith(("v_add", "v_sub", "v_mul", "v_mad", "v_addc", "v_subb", "v_mbcnt", "v_min", "v_max", "v_ffb", "v_bcnt", "v_cvt")): return "int arith"
    if op.startswith("v_"): return "other valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "flat_", "buffer_")): return "vmem"
    if op.startswith("scratch_"): return "scratch (VGPR spill)"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_load"): return "smem"
    if op.startswith("s_"): return "salu"
    return None
VALU = set(CLASSES[:10])

# basic blocks and the loop each belongs to
blocks = []   # (label, first line, loop header or None, depth)
cur = {"label": "entry", "line": 0, "hdr": None, "depth": 0, "ins": collections.Counter(), "parents": []}
blocks.append(cur)
j = 0
while j < len(body):
    l = body[j]
    m = re.match(r"^(\.LBB\d+_\d+):(.*)$", l)
    if m:
        cur = {"label": m.group(1), "line": j, "hdr": None, "depth": 0, "ins": collections.Counter(), "parents": []}
        blocks.append(cur)
        k = j
        note = m.group(2)
        notes = [note]
        while k + 1 < len(body) and body[k + 1].strip().startswith(";"):
            k += 1
            notes.append(body[k])
        for nt in notes:
            mm = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", nt)
            if mm:
                cur["hdr"], cur["depth"] = "." + "L" + mm.group(1), int(mm.group(2))
            mm = re.search(r"Loop Header: Depth=(\d+)", nt)
            if mm:
                cur["hdr"], cur["depth"] = cur["label"], int(mm.group(1))
            mm = re.search(r"Parent Loop (BB\d+_\d+) Depth=(\d+)", nt)
            if mm:
                cur["parents"].append((".L" + mm.group(1), int(mm.group(2))))
    elif re.match(r"^; %bb\.\d+:", l):
        # a fall-through block: inherits nothing by itself; its loop is in the comment if any
        cur = {"label": l.split(":")[0].strip("; "), "line": j, "hdr": None, "depth": 0, "ins": collections.Counter(), "parents": []}
        blocks.append(cur)
        mm = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", l)
        if mm:
            cur["hdr"], cur["depth"] = ".L" + mm.group(1), int(mm.group(2))
    elif l.startswith("\t") and not l.strip().startswith((".", ";")):
        k_ = klass(l.split()[0])
        if k_:
            cur["ins"][k_] += 1
    j += 1
# loop tree: header -> parent header
parent = {}
for b in blocks:
    if b["hdr"] == b["label"] and b["parents"]:
        parent[b["label"]] = max(b["parents"], key=lambda x: x[1])[0]
    elif b["hdr"] == b["label"]:
        parent[b["label"]] = None
loops = collections.OrderedDict()
for b in blocks:
    h = b["hdr"]
    loops.setdefault(h, {"ins": collections.Counter(), "first": b["line"], "last": b["line"], "depth": b["depth"]})
    loops[h]["ins"].update(b["ins"])
    loops[h]["last"] = max(loops[h]["last"], b["line"])
    loops[h]["first"] = min(loops[h]["first"], b["line"])
weights = json.load(open(wfile)) if wfile else {}
def mult(h):
    """executions of loop h's body per bounce iteration = product of the trips along its chain of parents"""
    m_ = 1.0
    while h is not None:
        m_ *= float(weights.get("trips", {}).get(h, 1.0 if not wfile else 0.0 if h not in weights.get("trips", {}) else 1.0))
        h = parent.get(h)
    return m_
print(f"{text[i0][:64]}: {len(body)} lines, {len(loops)} loops (own instructions of each loop, children excluded)")
tot = collections.Counter()
for h, L in loops.items():
    own = L["ins"]
    v = sum(n for c, n in own.items() if c in VALU)
    w = mult(h) if h else (1.0 if not wfile else float(weights.get("outside", 0.0)))
    if "--blocks" in sys.argv or not wfile or w > 0:
        print(f"{str(h):12s} depth {L['depth']} lines {L['first']:5d}-{L['last']:5d} parent {str(parent.get(h)):12s} x{w:7.2f}  valu {v:4d} "
              f"f64 {own['f64 add/mul/fma']:4d} lane {own['lane<->scalar (SGPR spill, readfirstlane)']:3d} mov {own['move']:3d} sel {own['select']:3d} "
              f"cmp {own['compare']:3d} int {own['int arith']:3d} bit {own['bit/shift/logic']:3d} scr {own['scratch (VGPR spill)']:2d} lds {own['lds']:3d} salu {own['salu']:4d} br {own['branch']:3d}")
    for c, n in own.items():
        tot[c] += n * w
if wfile:
    v = sum(n for c, n in tot.items() if c in VALU)
    print(f"\nestimated per bounce iteration of a wave (weights: {wfile}; {weights.get('note', '')})")
    for c in CLASSES:
        print(f"  {c:45s} {tot[c]:8.1f}" + (f"   {tot[c] / v:6.3f} of the vector instructions" if c in VALU else ""))
    print(f"  vector instructions in all {v:8.1f}   (measured: {weights.get('measured_valu_per_iteration', '?')})")

if "--lines" in sys.argv:
    gpath = sys.argv[sys.argv.index("--lines") + 1]
    gtext = open(gpath).read().split("\n")
    files = {}
    for l in gtext:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', l)
        if m:
            files[int(m.group(1))] = m.group(3).split("/")[-1]
    g0 = next(i for i, l in enumerate(gtext) if re.match(r"^_ZN3tor16integrate_kernel\w*:", l) and want in l)
    g1 = next(j for j in range(g0, len(gtext)) if "s_endpgm" in gtext[j])
    cur, depth = ("?", 0), 0
    lane, scr = collections.Counter(), collections.Counter()
    for l in gtext[g0:g1]:
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        if l.startswith(".LBB") or l.startswith("; %bb"):
            mm = re.search(r"in Loop: Header=\S+ Depth=(\d+)", l)
            depth = int(mm.group(1)) if mm else (0 if "Depth" not in l else depth)
            mm = re.search(r"Loop Header: Depth=(\d+)", l)
            if mm:
                depth = int(mm.group(1))
            continue
        if "Loop Header: Depth=" in l:
            depth = int(re.search(r"Depth=(\d+)", l).group(1))
            continue
        t = l.strip().split()
        if not t:
            continue
        if t[0].startswith(("v_readlane", "v_writelane")):
            lane[(depth, cur[0], t[0][:10])] += 1
        if t[0].startswith("scratch_"):
            scr[(depth, cur[0], t[0])] += 1
    print("\nspill instructions by loop depth (0 = outside every loop, 1 = the bounce loop's own body, 2 = the pass / segment loop, 3+ = the inner loops) and source file")
    for k, v in sorted(lane.items()):
        print(f"  depth {k[0]}  {k[1]:34s} {k[2]:12s} {v:4d}")
    for k, v in sorted(scr.items()):
        print(f"  depth {k[0]}  {k[1]:34s} {k[2]:22s} {v:4d}")
    by_depth = collections.Counter()
    for k, v in lane.items():
        by_depth[k[0]] += v
    print("  lane reads / writes per depth:", dict(sorted(by_depth.items())))
