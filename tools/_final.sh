cd $GRAFT_REPO_ROOT
bash tools/profile_round5.sh > gpurun_out/r5_profile_round5.log 2>&1
bash tools/occupancy_probe.sh > /dev/null 2>&1
TOR_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --verify --cpu-seconds 6 > gpurun_out/r5_bench_torchrun2.json 2> gpurun_out/r5_bench_torchrun2.err; echo "torchrun rc=$?"
cat gpurun_out/r5_occupancy.txt
