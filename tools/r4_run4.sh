set -x
mkdir -p gpurun_out/r4d
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -x -q -m gpu -k "not configs3 and not configs4 and not bench_multi" 2>&1 | tail -15 ) > gpurun_out/r4d/pytest_sub.txt 2>&1
tail -4 gpurun_out/r4d/pytest_sub.txt
( timeout 900 python bench.py --steps 5 --warmup 1 --cpu-seconds 5 --no-pmc ) > gpurun_out/r4d/bench_c3.json 2> gpurun_out/r4d/bench_c3.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4d/bench_c3.json').read().strip().splitlines()[-1])
print('VALUE', d['value'], d['ms_per_step'], d['roofline']['executed_live']['candidates_per_query'])
for k in ("unscreened","accel_f32","accel_blocks","accel_blocks_f32","pixel_seeding","pixel_seeding_default_accel","host_canvas"):
    if k in d: print(k, d[k]['value'], {kk:vv for kk,vv in d[k].items() if 'identical' in kk})
PY
( SPP=50 timeout 600 python tools/section_probe.py 2>&1 | grep seeding ) > gpurun_out/r4d/section_probe.txt 2>&1
cat gpurun_out/r4d/section_probe.txt
