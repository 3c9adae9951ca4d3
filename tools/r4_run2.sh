set -x
mkdir -p gpurun_out/r4b
( timeout 900 python bench.py --steps 5 --warmup 1 --cpu-seconds 5 --no-pmc ) > gpurun_out/r4b/bench_c3.json 2> gpurun_out/r4b/bench_c3.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4b/bench_c3.json').read().strip().splitlines()[-1])
print('VALUE', d['value'], d['ms_per_step'], d['roofline']['executed_live'])
for k in ("unscreened","accel_f32","accel_blocks","accel_blocks_f32","pixel_seeding","pixel_seeding_default_accel","host_canvas"):
    if k in d: print(k, d[k]['value'], {kk:vv for kk,vv in d[k].items() if 'identical' in kk})
PY
( timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -30 ) > gpurun_out/r4b/pytest_gpu.txt 2>&1
tail -5 gpurun_out/r4b/pytest_gpu.txt
