"""Static instruction counts of one integrate_kernel variant by SOURCE LINE (compile with -gline-tables-only first:
    hipcc --offload-arch=gfx950 <CXXFLAGS of csrc/Makefile> -gline-tables-only -S --cuda-device-only -o /tmp/tor_kernels_g.s tor_kernels.hip).
usage: python tools/isa_lines.py /tmp/tor_kernels_g.s [name part, default the metric's kernel] [top N]
Per (file, line): vector / scalar instructions attributed by the .loc in force (inlined code counts at its own line)."""
import collections, re, sys
path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "Li1ELi2ELi3ELi0ELi0E"
top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
text = open(path).read().split("\n")
files = {}
for l in text:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', l)
    if m:
        files[int(m.group(1))] = m.group(3)
i0 = next(i for i, l in enumerate(text) if re.match(r"^_ZN3tor16integrate_kernel\w*:", l) and want in l)
i1 = next(j for j in range(i0, len(text)) if "s_endpgm" in text[j])
cur = None
v = collections.Counter(); s = collections.Counter()
for l in text[i0:i1]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    if l.startswith("\t") and not l.strip().startswith((".", ";")):
        op = l.split()[0]
        if op.startswith("v_") or op.startswith("ds_") or op.startswith("global_"):
            v[cur] += 1
        elif op.startswith("s_"):
            s[cur] += 1
tot_v, tot_s = sum(v.values()), sum(s.values())
print(f"{text[i0][:60]}: vector {tot_v}, scalar {tot_s}")
byfile = collections.Counter()
for k, n in v.items():
    byfile[k[0]] += n
print("vector by file:", dict(byfile.most_common()))
for (k, n) in v.most_common(top):
    print(f"{k[0]:34s}:{k[1]:5d}  vector {n:5d}  scalar {s.get(k, 0):5d}")
