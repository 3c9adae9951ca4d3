#!/bin/bash
# Collects the rocprofv3 evidence for one bench configuration on the GPU box:
#   kernel trace + stats, then PMC counters in SEPARATE passes (never combined with tracing).
# usage (inside gpurun):  bash tools/profile_gpu.sh <tag> [bench.py args...]
# Output: gpurun_out/prof_<tag>/...; summarise with tools/rocpd_summary.py and commit under profiles/.
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $R/bench.py --no-cpu-baseline $*"
run() { # name, rocprof args...
  local name=$1; shift
  timeout 240 rocprofv3 "$@" -d $OUT/$name -o bench -- $BENCH --steps 1 --warmup 0 > $OUT/$name.log 2>&1
  grep -h '"metric"' $OUT/$name.log | cut -c1-300
}
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $BENCH --steps 3 --warmup 1 > $OUT/trace.log 2>&1
grep -h '"metric"' $OUT/trace.log | cut -c1-400
run pmc_sq_time  --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run pmc_sq_insts --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64
run pmc_sq_mix   --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM SQ_BUSY_CYCLES SQ_THREAD_CYCLES_VALU
run pmc_sq_f32   --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32
run pmc_sqc      --pmc SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_REQ
run pmc_fetch    --pmc FETCH_SIZE
run pmc_write    --pmc WRITE_SIZE
run pmc_grbm     --pmc GRBM_GUI_ACTIVE GRBM_COUNT
run pmc_lds      --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS_ATOMIC SQ_WAIT_INST_LDS SQ_IFETCH SQ_LEVEL_WAVES SQ_WAVES SQ_CYCLES
ls $OUT
