cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_round5.py -x -q -m gpu > gpurun_out/t_r5.log 2>&1; echo "round5 tests rc=$?"; tail -2 gpurun_out/t_r5.log
bash tools/profile_round5.sh quick > gpurun_out/r5_profile_round5.log 2>&1
bash tools/occupancy_probe.sh > /dev/null 2>&1
TOR_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --verify --cpu-seconds 6 > gpurun_out/r5_bench_torchrun2.json 2> gpurun_out/r5_bench_torchrun2.err; echo "torchrun rc=$?"
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/t_all.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" gpurun_out/t_all.log | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
