#!/bin/bash
# Round-3 evidence (inside gpurun: bash tools/profile_round3.sh): bench lines, rocprofv3 summaries, shard table, fuzz.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd $R
python bench.py > $O/r3_bench_c3.json 2> $O/r3_bench_c3.err
python bench.py --config c2 --no-cpu-baseline > $O/r3_bench_c2.json 2> $O/r3_bench_c2.err
python bench.py --workload c5 --accel blocks+f32 --steps 6 --warmup 2 > $O/r3_bench_c5.json 2> $O/r3_bench_c5.err
python bench.py --gpus 2 --steps 2 --warmup 1 --verify --no-cpu-baseline > $O/r3_bench_2dev_strong.json 2> $O/r3_bench_2dev.err
COMMON="--no-accel-leg --no-pmc --no-host-leg --no-stats"
bash tools/profile_gpu.sh r3_c3 --spp 1000 $COMMON > $O/prof_r3_c3.log 2>&1
TOR_SCREEN=0 bash tools/profile_gpu.sh r3_c3_unscreened --spp 1000 $COMMON > $O/prof_r3_c3_unscreened.log 2>&1
bash tools/profile_gpu.sh r3_c2 --spp 100 $COMMON > $O/prof_r3_c2.log 2>&1
bash tools/profile_gpu.sh r3_c3_pixel_default --spp 1000 --seeding pixel --accel blocks+f32 $COMMON > $O/prof_r3_c3_pixel_default.log 2>&1
bash tools/profile_gpu.sh r3_c2_pixel_default --spp 100 --seeding pixel --accel blocks+f32 $COMMON > $O/prof_r3_c2_pixel_default.log 2>&1
for t in r3_c3 r3_c3_unscreened r3_c2 r3_c3_pixel_default r3_c2_pixel_default; do
  python tools/rocpd_summary.py $O/prof_$t $O/${t}_summary.txt > /dev/null
  rm -rf $O/prof_$t/*/
done
python tools/shard_times.py --configs c3,c4 --modes a,b,c --repeat 2 --out $O/r3_shard_times.json > $O/r3_shard_times.log 2>&1
python tools/mig_check.py 144x256x100 216x384x100 288x512x100 432x768x100 540x960x100 720x1280x100 900x1600x100 1080x1920x100 216x384x1000 540x960x1000 1080x1920x1000 > $O/r3_mig_check.log 2>&1
timeout 200 python tools/fuzz_accel.py 120 3 > $O/r3_fuzz.txt 2>&1
ls -la $O/*r3*
