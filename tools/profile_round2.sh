#!/bin/bash
# Round-2 evidence: rocprofv3 kernel trace + PMC passes (separate runs) for the bench's dominant kernel and the
# kernels VERDICT r1 named.  usage (inside gpurun): bash tools/profile_round2.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
COMMON="--no-accel-leg --no-pmc --no-host-leg --no-stats"
bash $R/tools/profile_gpu.sh r2_c3 --spp 1000 $COMMON > $R/gpurun_out/prof_r2_c3.log 2>&1
bash $R/tools/profile_gpu.sh r2_c2_accel3 --spp 100 --accel blocks+f32 $COMMON > $R/gpurun_out/prof_r2_c2_accel3.log 2>&1
bash $R/tools/profile_gpu.sh r2_c2_pixel_accel3 --spp 100 --seeding pixel --accel blocks+f32 $COMMON > $R/gpurun_out/prof_r2_c2_pixel_accel3.log 2>&1
bash $R/tools/profile_gpu.sh r2_c2_pixel_brute --spp 100 --seeding pixel $COMMON > $R/gpurun_out/prof_r2_c2_pixel_brute.log 2>&1
bash $R/tools/profile_gpu.sh r2_c1_wave_kernel --width 384 --height 216 --spp 100 --seeding pixel $COMMON > $R/gpurun_out/prof_r2_c1_wave.log 2>&1
for t in r2_c3 r2_c2_accel3 r2_c2_pixel_accel3 r2_c2_pixel_brute r2_c1_wave_kernel; do
  python $R/tools/rocpd_summary.py $R/gpurun_out/prof_$t $R/gpurun_out/${t}_summary.txt > /dev/null
  rm -rf $R/gpurun_out/prof_$t/*/  # the databases are large; the summaries are what gets committed
done
ls -la $R/gpurun_out/*_summary.txt
