cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-accel-leg --no-host-leg > gpurun_out/res_c3.json 2> gpurun_out/res_c3.err; python - <<'PY'
import json
d=[json.loads(l) for l in open("gpurun_out/res_c3.json") if l.startswith("{")][0]
e=d["roofline"].get("executed_live") or {}
print(d["value"], {k:e.get(k) for k in ("valu_per_sample","fp64_per_sample","valu_issue_util","kernel_ms")})
PY
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
