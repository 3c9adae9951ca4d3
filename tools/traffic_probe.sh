#!/bin/bash
# HBM traffic of one bench step (separate rocprofv3 --pmc passes for FETCH_SIZE and WRITE_SIZE).
# usage (inside gpurun): bash tools/traffic_probe.sh <tag> [bench.py args...]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/traffic_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --pmc $c -d $OUT/$c -o bench -- python $R/bench.py --no-cpu-baseline --no-accel-leg --steps 1 --warmup 0 "$@" > $OUT/$c.log 2>&1
done
python $R/tools/rocpd_summary.py $OUT $OUT/summary.txt > /dev/null
grep -E "integrate_kernel.*(FETCH_SIZE|WRITE_SIZE)" $OUT/summary.txt | awk -F'|' -v t=$TAG '{print t, $1, $2, $4}'
rm -rf $OUT/FETCH_SIZE $OUT/WRITE_SIZE
