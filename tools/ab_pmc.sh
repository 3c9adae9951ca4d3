#!/bin/bash
# A/B of two builds on the GPU box with hardware counters: one rocprofv3 --pmc pass per counter group around bench.py's pmc child
# (one configs[1]-sized launch pair through the bare C ABI), for the in-tree library and for TOR_AB_LIB.
# usage (inside gpurun): bash tools/ab_pmc.sh [other.so]   -> gpurun_out/ab_pmc.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OTHER=${1:-$R/trace-of-radiance_amd/lib/libtor_mi355x_old.so}
OUT=$R/gpurun_out/ab_pmc.txt
mkdir -p $R/gpurun_out; : > $OUT
export TMPDIR=/tmp TOR_NO_TORCH=1
cd /tmp
SPEC=${SPEC:-1920,1080,100,50,1,0,0}
for lib in "" "$OTHER"; do
  echo "== ${lib:-in-tree library}" | tee -a $OUT
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_ACTIVE_INST_VALU" \
             "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES" \
             "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_BRANCH SQ_IFETCH SQ_INST_CYCLES_SALU" \
             "GRBM_GUI_ACTIVE"; do
    d=/tmp/abpmc_$$; rm -rf $d
    TOR_AB_LIB=$lib timeout 200 rocprofv3 --pmc $grp -d $d -o p -- python $R/bench.py --pmc-child $SPEC > /dev/null 2>&1
    python - "$d" <<'PY' | tee -a $OUT
import glob, sqlite3, sys
dbs = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)
con = sqlite3.connect(dbs[0])
rows = con.execute("select name, counter_name, sum(counter_value), max(duration), dispatch_id from pmc_events group by name, counter_name, dispatch_id order by dispatch_id").fetchall()
rows = [r for r in rows if "integrate_kernel" in r[0]]
last = max(r[4] for r in rows)
for r in rows:
    if r[4] == last:
        print(f"  {r[1]:28s} {r[2]:.6g}   kernel {r[3] / 1e6:.3f} ms")
PY
  done
done
