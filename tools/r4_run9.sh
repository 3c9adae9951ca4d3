#!/bin/bash
# round 4, plane screen: where the time goes (section probe) and LDS vs vector-cache records for stage two
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in "TOR_PLANE=1 TOR_PLANE_LDS=1" "TOR_PLANE=1 TOR_PLANE_LDS=0" "TOR_PLANE=0"; do
  echo "== $v" >> gpurun_out/r9_sections.txt
  env $v SPP=20 timeout 300 python tools/section_probe.py 2>&1 | grep "accel 0" >> gpurun_out/r9_sections.txt
  tag=$(echo "$v" | tr -d ' =' )
  env $v timeout 600 python bench.py --config c2 --steps 5 --warmup 1 --no-cpu-baseline --no-pmc --no-host-leg --no-accel-leg > gpurun_out/r9_c2_$tag.json 2> gpurun_out/r9_c2_$tag.err
  python -c "
import json,sys
d=json.loads(open('gpurun_out/r9_c2_$tag.json').read().strip().splitlines()[-1]); print('$tag', d['value'], d.get('pixel_seeding',{}).get('value'))" >> gpurun_out/r9_sections.txt
done
cat gpurun_out/r9_sections.txt
