#!/bin/bash
# Round-5 evidence (inside gpurun: bash tools/profile_round5.sh [quick]): bench lines, rocprofv3 summaries of every mode bench.py
# replays from profiles/traffic.json (no entry may point at an older round's kernel: tools/update_traffic.py checks the variant), shard table, fuzz.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python bench.py > $O/r5_bench_c3.json 2> $O/r5_bench_c3.err
python bench.py --config c2 --cpu-seconds 6 > $O/r5_bench_c2.json 2> $O/r5_bench_c2.err
python bench.py --gpus 2 --steps 3 --warmup 1 --verify --cpu-seconds 6 > $O/r5_bench_2dev.json 2> $O/r5_bench_2dev.err
python bench.py --workload c5 --accel blocks+f32 --steps 6 --warmup 2 > $O/r5_bench_c5.json 2> $O/r5_bench_c5.err
COMMON="--no-accel-leg --no-pmc --no-host-leg --no-stats"
prof() { # tag, bench args
  local t=$1; shift
  bash tools/profile_gpu.sh $t "$@" $COMMON > $O/prof_$t.log 2>&1
  python tools/rocpd_summary.py $O/prof_$t $O/${t}_summary.txt > /dev/null
  rm -rf $O/prof_$t/*/
}
prof r5_c3 --spp 1000
prof r5_c2 --spp 100
prof r5_c2_pixel_brute --spp 100 --seeding pixel
prof r5_c2_f32 --spp 100 --accel f32
prof r5_c2_blocks --spp 100 --accel blocks
prof r5_c2_accel3 --spp 100 --accel blocks+f32
prof r5_c3_accel3 --spp 1000 --accel blocks+f32
prof r5_c2_pixel_default --spp 100 --seeding pixel --accel blocks+f32
prof r5_c3_pixel_default --spp 1000 --seeding pixel --accel blocks+f32
python tools/brute_generality.py --out $O/r5_brute_generality.txt > $O/r5_brute_generality.log 2>&1
SPP=50 python tools/section_probe.py > $O/r5_section_probe_c2x50.txt 2>&1
if [ "${1:-}" != "quick" ]; then
  python tools/shard_times.py --configs c3,c4 --modes a,b,c --repeat 2 --out $O/r5_shard_times.json > $O/r5_shard_times.log 2>&1
  timeout 400 python tools/fuzz_accel.py 300 4 > $O/r5_fuzz.txt 2>&1
fi
ls -la $O/*r5*
