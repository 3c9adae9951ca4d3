#!/bin/bash
# How much of the vector unit can 1, 2, 3 waves per SIMD of the metric's kernel keep busy?  (inside gpurun: bash tools/occupancy_probe.sh)
# TOR_BLOCKS_PER_CU caps the persistent grid's workgroups per CU (4 waves each, one per SIMD); bench.py reads the counters live.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r5_occupancy.txt
cd $R
echo "# tools/occupancy_probe.sh: configs[1] (1920x1080x100), float64 brute force, sample streams; live counters of bench.py (roofline.executed_live)" > $O
for b in 1 2 3; do
  TOR_BLOCKS_PER_CU=$b python bench.py --config c2 --no-cpu-baseline --no-accel-leg --no-host-leg > gpurun_out/occ_$b.json 2> gpurun_out/occ_$b.err
  python - $b >> $O <<'PY'
import json, sys
b = sys.argv[1]
d = [json.loads(l) for l in open(f"gpurun_out/occ_{b}.json") if l.startswith("{")][0]
e = d["roofline"].get("executed_live") or {}
print(f"workgroups per CU {b} (waves per SIMD {b}): {d['value']:8.1f} Msamples/s  kernel {e.get('kernel_ms')} ms  VALU issue utilisation {e.get('valu_issue_util')}  "
      f"VALU / sample {e.get('valu_per_sample')}  SALU wave-instructions {e.get('salu_wave_instructions'):.4g}")
PY
done
cat $O
