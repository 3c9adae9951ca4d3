"""Scratch traffic of the integrate_kernel variants in the generated ISA (make -C trace-of-radiance_amd/csrc asm):
spill stores / loads in total and inside loops (a store inside the bounce loop is HBM write traffic on every iteration).
usage: python tools/isa_spills.py [substring of the mangled kernel name]"""
import os, re, sys
path = os.environ.get("TOR_ASM") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "trace-of-radiance_amd", "lib", "asm", "tor_kernels.s")
want = sys.argv[1] if len(sys.argv) > 1 else "integrate_kernel"
text = open(path).read().split("\n")
starts = [(i, m.group(1)) for i, l in enumerate(text) if (m := re.match(r"^(_ZN3tor\w+):", l))]
for k, (i0, name) in enumerate(starts):
    if want not in name:
        continue
    i1 = next((j for j in range(i0, len(text)) if "s_endpgm" in text[j]), len(text))
    body = text[i0:i1]
    labels = {m.group(1): j for j, l in enumerate(body) if (m := re.match(r"^(\.LBB\w+):", l))}
    loops = []
    for j, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\w+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < j:
            loops.append((labels[m.group(1)], j))
    def in_loop(j):
        return any(a <= j <= b for a, b in loops)
    st = [j for j, l in enumerate(body) if "scratch_store" in l]
    ld = [j for j, l in enumerate(body) if "scratch_load" in l]
    print(f"{name[24:64]:42s} {i1 - i0:6d} lines; scratch stores {len(st):3d} (in loops {sum(map(in_loop, st)):3d}); loads {len(ld):3d} (in loops {sum(map(in_loop, ld)):3d})")
