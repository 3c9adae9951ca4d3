#!/bin/bash
# round 4, plane screen: quick parity + A/B (TOR_PLANE=1 / 0) on configs[1] and configs[2]
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py tests/test_gpu_round4.py -x -q -m gpu > gpurun_out/r8_tests.log 2>&1
echo "tests rc=$?" > gpurun_out/r8_status.txt
tail -5 gpurun_out/r8_tests.log
for pl in 1 0; do
  TOR_PLANE=$pl timeout 600 python bench.py --config c2 --steps 5 --warmup 1 --no-cpu-baseline --no-pmc --no-host-leg > gpurun_out/r8_c2_plane$pl.json 2> gpurun_out/r8_c2_plane$pl.err
  TOR_PLANE=$pl timeout 600 python bench.py --config c3 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-host-leg --no-accel-leg > gpurun_out/r8_c3_plane$pl.json 2> gpurun_out/r8_c3_plane$pl.err
done
python - <<'PY'
import json
for f in ("r8_c2_plane1","r8_c2_plane0","r8_c3_plane1","r8_c3_plane0"):
    try:
        d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["value"], d.get("pixel_seeding",{}).get("value"), d.get("unscreened",{}).get("value"))
    except Exception as e:
        print(f, "ERR", e)
PY
