#!/bin/bash
# A/B of two builds of the library on the GPU box: section shares of the bounce iteration (tools/section_probe.py, float64 brute
# force rows only) for the in-tree library (default, and with the plane gate off: TOR_PLANE=2) and for TOR_AB_LIB (default:
# lib/libtor_mi355x_old.so, an older build of the kernels made from git (note: lib/*_r[0-9]*.so does not travel to the GPU box, .gpurunignore): `git archive <commit> trace-of-radiance_amd/csrc include | tar -x -C
# /tmp/x && make -C ...`).
# usage (inside gpurun): bash tools/ab_sections.sh [other.so]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OTHER=${1:-$R/trace-of-radiance_amd/lib/libtor_mi355x_old.so}
for rep in 1 2; do
for lib in "" "$OTHER"; do
  echo "== ${lib:-in-tree library}"
  TOR_AB_LIB=$lib SPP=${SPP:-50} python $R/tools/section_probe.py 2>&1 | grep "seeding 1 accel 0"
done
echo "== in-tree library, TOR_PLANE=2"
TOR_PLANE=2 SPP=${SPP:-50} python $R/tools/section_probe.py 2>&1 | grep "seeding 1 accel 0"
done
