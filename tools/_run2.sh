cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-accel-leg --no-host-leg > gpurun_out/res_c3.json 2> gpurun_out/res_c3.err; python - <<'PY'
import json
d=[json.loads(l) for l in open("gpurun_out/res_c3.json") if l.startswith("{")][0]
e=d["roofline"].get("executed_live") or {}
print(d["value"], {k:e.get(k) for k in ("valu_per_sample","fp64_per_sample","valu_issue_util","kernel_ms","salu_wave_instructions")})
PY
python bench.py --config c2 --no-cpu-baseline --no-accel-leg --no-host-leg --no-pmc > gpurun_out/res_c2.json 2> gpurun_out/res_c2.err; python -c "
import json
d=[json.loads(l) for l in open('gpurun_out/res_c2.json') if l.startswith('{')][0]; print('c2', d['value'])"
