#!/bin/bash
# Re-profile only the accelerated modes (inside gpurun: bash tools/profile_accel_refresh.sh) -- same tags as tools/profile_round6.sh.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
COMMON="--no-accel-leg --no-pmc --no-host-leg --no-stats"
prof() { # tag, bench args
  local t=$1; shift
  bash tools/profile_gpu.sh $t "$@" $COMMON > $O/prof_$t.log 2>&1
  python tools/rocpd_summary.py $O/prof_$t $O/${t}_summary.txt > /dev/null
  rm -rf $O/prof_$t/*/
}
prof r6_c2_f32 --spp 100 --accel f32
prof r6_c2_blocks --spp 100 --accel blocks
prof r6_c2_accel3 --spp 100 --accel blocks+f32
prof r6_c3_accel3 --spp 1000 --accel blocks+f32
prof r6_c2_pixel_default --spp 100 --seeding pixel --accel blocks+f32
prof r6_c3_pixel_default --spp 1000 --seeding pixel --accel blocks+f32
