#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -x -q -m gpu -k "c1_reference or edge_cases or sample_seeding_matches or pixel_seeding_matches or fma_screen" 2>&1 | tail -3
for v in "TOR_PLANE_LDS=1" "TOR_PLANE_LDS=0"; do
  echo "== $v"
  env $v SPP=20 timeout 300 python tools/section_probe.py 2>&1 | grep "seeding 1 accel 0"
  env $v timeout 600 python bench.py --config c2 --steps 5 --warmup 1 --no-cpu-baseline --no-pmc --no-host-leg --no-accel-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', d['value'])"
done
python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-host-leg --no-accel-leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3', d['value'])"
