#!/bin/bash
# A/B of two builds of the library in one gpurun call: tools/ab_rate.py for the in-tree library and for another build, alternating.
# usage (inside gpurun): bash tools/ab_rate.sh [other.so] [rounds]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OTHER=${1:-$R/trace-of-radiance_amd/lib/libtor_mi355x_old.so}
for rep in $(seq 1 ${2:-2}); do
  for lib in "" "$OTHER"; do
    TOR_AB_LIB=$lib python $R/tools/ab_rate.py 2>&1 | grep -v "amdgpu.ids\|TOR_AB_LIB set"
  done
done
