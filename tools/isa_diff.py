#!/usr/bin/env python3
"""Compare two `make -C trace-of-radiance_amd/csrc asm` outputs kernel by kernel (round 5: pruning the variant table must not
change a kept variant).  Labels carry the function's ordinal in the file (.LBB57_12), which shifts when functions are removed:
they are normalised; everything else must match byte for byte.

    python tools/isa_diff.py old.s new.s
"""
import re
import sys


def kernels(path):
    out, name, body = {}, None, []
    for ln in open(path):
        m = re.match(r"^(_Z\w+):\s", ln)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        if ln.startswith("\t.section") or ln.startswith(".Lfunc_end"):
            out[name] = body
            name = None
            continue
        t = re.sub(r"\.LBB\d+_", ".LBB_", ln)
        t = re.sub(r";.*$", "", t).rstrip()      # comments carry block ordinals too
        t = re.sub(r"\.Lfunc_end\d+", ".Lfunc_end", t)
        if t:
            body.append(t)
    return out


def main():
    a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
    gone = sorted(set(a) - set(b))
    new = sorted(set(b) - set(a))
    diff = [k for k in sorted(set(a) & set(b)) if a[k] != b[k]]
    print(f"{len(a)} kernels before, {len(b)} after; removed {len(gone)}, added {len(new)}, kept {len(set(a) & set(b))}, "
          f"kept-but-different {len(diff)}")
    for k in gone:
        print("  removed:", k)
    for k in new:
        print("  added:  ", k)
    for k in diff:
        n = sum(1 for x, y in zip(a[k], b[k]) if x != y) + abs(len(a[k]) - len(b[k]))
        print(f"  DIFFERS: {k}: {len(a[k])} -> {len(b[k])} lines, {n} differing")
    return 1 if diff else 0


if __name__ == "__main__":
    sys.exit(main())
