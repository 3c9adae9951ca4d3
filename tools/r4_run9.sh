set -x
mkdir -p gpurun_out/r4i
( timeout 260 python tools/fuzz_accel.py 120 21 2>&1 | tail -2 ) > gpurun_out/r4i/fuzz.txt 2>&1
cat gpurun_out/r4i/fuzz.txt
( timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-host-leg ) > gpurun_out/r4i/bench_c3.json 2> gpurun_out/r4i/bench_c3.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4i/bench_c3.json').read().strip().splitlines()[-1])
print('VALUE', d['value'], d['ms_per_step'])
for k in ("unscreened","accel_f32","accel_blocks","accel_blocks_f32","pixel_seeding","pixel_seeding_default_accel"):
    if k in d: print(k, d[k]['value'], {kk:vv for kk,vv in d[k].items() if 'identical' in kk})
PY
( timeout 600 python bench.py --workload c5 --accel blocks+f32 --steps 6 --warmup 2 ) > gpurun_out/r4i/bench_c5.json 2>/dev/null; cut -c1-200 gpurun_out/r4i/bench_c5.json
( timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_round4.py -q -m gpu -x -k "not configs3 and not configs2_sample" 2>&1 | tail -5 ) > gpurun_out/r4i/pytest.txt 2>&1
tail -3 gpurun_out/r4i/pytest.txt
