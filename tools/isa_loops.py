"""Static instruction mix of every loop of one integrate_kernel variant in the generated ISA (make -C trace-of-radiance_amd/csrc asm).
usage: python tools/isa_loops.py [substring of the mangled name, default the metric's kernel Li1ELi2ELi3ELi0ELi0E] [--dump A B]"""
import os, re, sys
path = os.environ.get("TOR_ASM") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "trace-of-radiance_amd", "lib", "asm", "tor_kernels.s")
want = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "Li1ELi2ELi3ELi0ELi0E"
text = open(path).read().split("\n")
i0 = next(i for i, l in enumerate(text) if re.match(r"^_ZN3tor16integrate_kernel\w*:", l) and want in l)
i1 = next(j for j in range(i0, len(text)) if "s_endpgm" in text[j])
body = text[i0:i1]
if "--dump" in sys.argv:
    k = sys.argv.index("--dump")
    print("\n".join(body[int(sys.argv[k + 1]):int(sys.argv[k + 2]) + 1]))
    sys.exit(0)
labels = {m.group(1): j for j, l in enumerate(body) if (m := re.match(r"^(\.LBB\w+):", l))}
loops = sorted({(labels[m.group(1)], j) for j, l in enumerate(body) if (m := re.search(r"s_c?branch\w*\s+(\.LBB\w+)", l)) and m.group(1) in labels and labels[m.group(1)] < j})
def stats(a, b):
    ins = [l.strip().split()[0] for l in body[a:b + 1] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    return (len(ins), sum(x.startswith("v_") for x in ins), sum(("_f64" in x) for x in ins), sum(x.startswith("s_") for x in ins),
            sum(x.startswith("ds_") for x in ins), sum(x.startswith(("s_cbranch", "s_branch")) for x in ins), sum(x.startswith("s_waitcnt") for x in ins))
print(f"{text[i0][:70]}: {len(body)} lines")
for a, b in loops:
    n, v, f, s, d, br, w = stats(a, b)
    print(f"lines {a:5d}-{b:5d}: {n:5d} instr, valu {v:5d} (f64 {f:4d}) salu {s:4d} (branches {br:3d}, waits {w:3d}) ds {d:3d}")
