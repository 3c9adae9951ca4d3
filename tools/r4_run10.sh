#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$(pwd)
COMMON="--no-accel-leg --no-pmc --no-host-leg --no-stats"
bash tools/profile_gpu.sh r4_c2 --spp 100 $COMMON > gpurun_out/prof_r4_c2.log 2>&1
python tools/rocpd_summary.py gpurun_out/prof_r4_c2 gpurun_out/r4_c2_summary.txt > /dev/null
rm -rf gpurun_out/prof_r4_c2/*/
cd $R
python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-host-leg --no-accel-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3', d['value'])"
grep -c . gpurun_out/r4_c2_summary.txt
