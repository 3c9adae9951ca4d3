#!/bin/bash
# quick A/B of a kernel change: parity subset, section shares, the bench lines' rates
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_parity.py tests/test_gpu_round3.py -x -q -m gpu -k "plane or c1_reference or edge_cases or fma_screen or sample_seeding_matches or pixel_seeding_matches or block_culling_never" 2>&1 | tail -3
SPP=20 timeout 300 python tools/section_probe.py 2>&1 | grep "seeding 1 accel [03]\|seeding 0 accel 3"
for c in c2 c3; do
python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-host-leg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', d['value'], {k:d[k]['value'] for k in ('accel_f32','accel_blocks','accel_blocks_f32','pixel_seeding','pixel_seeding_default_accel') if k in d})"
done
