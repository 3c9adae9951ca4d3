# quick GPU check of a kernel change: fuzz, the bench's metric + secondary legs, the parity suites
set -x
mkdir -p gpurun_out/chk
( timeout 200 python tools/fuzz_accel.py ${FUZZ_S:-120} 31 2>&1 | tail -1 ) > gpurun_out/chk/fuzz.txt 2>&1; cat gpurun_out/chk/fuzz.txt
( timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-host-leg ) > gpurun_out/chk/bench_c3.json 2> gpurun_out/chk/bench_c3.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/chk/bench_c3.json').read().strip().splitlines()[-1])
print('VALUE', d['value'], d['ms_per_step'], d['roofline']['workload_live']['candidates_per_query'])
for k in ("unscreened","accel_f32","accel_blocks","accel_blocks_f32","pixel_seeding","pixel_seeding_default_accel"):
    if k in d: print(k, d[k]['value'], {kk:vv for kk,vv in d[k].items() if 'identical' in kk})
PY
( timeout 2400 python -m pytest tests -q -m gpu -x ${PYTEST_K:-} 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -6 ) > gpurun_out/chk/pytest.txt 2>&1; cat gpurun_out/chk/pytest.txt
