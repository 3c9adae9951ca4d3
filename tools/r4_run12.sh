#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_parity.py -x -q -m gpu -k "plane or c1_reference or edge_cases" 2>&1 | tail -3
SPP=20 timeout 300 python tools/section_probe.py 2>&1 | grep "accel 0"
for c in c2 c3; do
python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-host-leg --no-accel-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', d['value'], d.get('pixel_seeding',{}).get('value'))"
done
