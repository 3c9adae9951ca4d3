#!/bin/bash
# round 4, final evidence: the whole GPU suite, then tools/profile_round4.sh (bench lines, rocprofv3 summaries, shard table, fuzz)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r4_gpu_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4_gpu_tests.log
tail -4 gpurun_out/r4_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/profile_round4.sh > gpurun_out/r4_profile.log 2>&1
tail -3 gpurun_out/r4_fuzz.txt
python - <<'PY'
import json
for f in ("r4_bench_c3","r4_bench_c2"):
    d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["value"], d["roofline"]["traffic"], {k:d[k]["value"] for k in ("unscreened","second_form_only","accel_f32","accel_blocks","accel_blocks_f32","pixel_seeding","pixel_seeding_default_accel") if k in d})
PY
