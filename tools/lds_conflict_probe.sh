#!/bin/bash
# Where do the LDS bank conflicts of the pooled resolve come from?  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of integrate_kernel<1,0,3,1,1>
# at configs[1] with the block records staged in LDS (default) and read from global memory (TOR_STAGE_LDS=0).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/lds_probe; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
for stage in default 0; do
  if [ $stage = 0 ]; then export TOR_STAGE_LDS=0; else unset TOR_STAGE_LDS; fi
  timeout 240 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d $O/stage_$stage -o p -- python $R/bench.py --config c2 --accel blocks+f32 --steps 1 --warmup 0 --no-cpu-baseline --no-accel-leg --no-pmc --no-host-leg --no-stats > $O/stage_$stage.log 2>&1
  python - <<PY
import glob, sqlite3
db = glob.glob("$O/stage_$stage/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
rows = con.execute("select name, counter_name, sum(counter_value), max(duration) from pmc_events where name like '%integrate_kernel<1, 0, 3, 1, 1>%' group by counter_name").fetchall()
print("TOR_STAGE_LDS=$stage", {r[1]: float(r[2]) for r in rows}, "kernel ms", max(r[3] for r in rows) / 1e6)
PY
done
