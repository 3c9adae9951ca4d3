#!/bin/bash
# Round-6 evidence (inside gpurun: bash tools/profile_round6.sh [quick]): bench lines, rocprofv3 summaries of every mode bench.py
# replays from profiles/traffic.json (no entry may point at an older round's kernel: tools/update_traffic.py checks the variant), shard table, fuzz.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
python bench.py > $O/r6_bench_c3.json 2> $O/r6_bench_c3.err
python bench.py --config c2 --cpu-seconds 6 > $O/r6_bench_c2.json 2> $O/r6_bench_c2.err
python bench.py --gpus 2 --steps 3 --warmup 1 --verify --cpu-seconds 6 > $O/r6_bench_2dev.json 2> $O/r6_bench_2dev.err
# (configs[4] in full -- 240 frames, ~3 min -- is its own call: python bench.py --workload c5 --frames 240 > profiles/r6_bench_c5_240.json)
TOR_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --verify --cpu-seconds 6 > $O/r6_bench_torchrun2.json 2> $O/r6_bench_torchrun2.err
COMMON="--no-accel-leg --no-pmc --no-host-leg --no-stats"
prof() { # tag, bench args
  local t=$1; shift
  bash tools/profile_gpu.sh $t "$@" $COMMON > $O/prof_$t.log 2>&1
  python tools/rocpd_summary.py $O/prof_$t $O/${t}_summary.txt > /dev/null
  rm -rf $O/prof_$t/*/
}
prof r6_c3 --spp 1000
prof r6_c2 --spp 100
prof r6_c2_pixel_brute --spp 100 --seeding pixel
prof r6_c2_f32 --spp 100 --accel f32
prof r6_c2_blocks --spp 100 --accel blocks
prof r6_c2_accel3 --spp 100 --accel blocks+f32
prof r6_c3_accel3 --spp 1000 --accel blocks+f32
prof r6_c2_pixel_default --spp 100 --seeding pixel --accel blocks+f32
prof r6_c3_pixel_default --spp 1000 --seeding pixel --accel blocks+f32
prof r6_c5 --workload c5 --frames 3
python tools/brute_generality.py --out $O/r6_brute_generality.txt > $O/r6_brute_generality.log 2>&1
SPP=50 python tools/section_probe.py > $O/r6_section_probe_c2x50.txt 2>&1
if [ -f trace-of-radiance_amd/lib/libtor_mi355x_fine.so ]; then   # (make -C trace-of-radiance_amd/csrc fine)
  for sc in random anim; do TOR_AB_LIB=$R/trace-of-radiance_amd/lib/libtor_mi355x_fine.so SPP=50 python tools/fine_probe.py $sc; done > $O/r6_fine_probe_c2x50.txt 2>&1
fi
if [ "${1:-}" != "quick" ]; then
  python tools/shard_times.py --configs c3,c4 --modes a,b,c --repeat 2 --out $O/r6_shard_times.json > $O/r6_shard_times.log 2>&1
  python tools/shard_times.py --c5-from profiles/r6_bench_c5_240.json --out $O/r6_shard_times.json >> $O/r6_shard_times.log 2>&1
  timeout 400 python tools/fuzz_accel.py 300 4 mixed --oracle 2 > $O/r6_fuzz.txt 2>&1
fi
ls -la $O/*r6*
