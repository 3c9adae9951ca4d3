set -x
mkdir -p gpurun_out/r4h
( timeout 2700 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 ) > gpurun_out/r4h/pytest_gpu.txt 2>&1
tail -4 gpurun_out/r4h/pytest_gpu.txt
( timeout 260 python tools/fuzz_accel.py 200 11 2>&1 | tail -3 ) > gpurun_out/r4h/fuzz.txt 2>&1
cat gpurun_out/r4h/fuzz.txt
( timeout 900 python bench.py --steps 5 --warmup 1 --cpu-seconds 6 ) > gpurun_out/r4h/bench_c3.json 2> gpurun_out/r4h/bench_c3.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4h/bench_c3.json').read().strip().splitlines()[-1])
print('VALUE', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])
for k in ("unscreened","accel_f32","accel_blocks","accel_blocks_f32","pixel_seeding","pixel_seeding_default_accel","host_canvas"):
    if k in d: print(k, d[k]['value'], {kk:vv for kk,vv in d[k].items() if 'identical' in kk})
PY
