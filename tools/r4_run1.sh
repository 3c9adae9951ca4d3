set -x
mkdir -p gpurun_out/r4a
( timeout 1500 python -m pytest tests/test_gpu_round4.py -x -q -m gpu 2>&1 | tail -40 ) > gpurun_out/r4a/pytest_round4.txt 2>&1
( timeout 120 tools/ubench/mfma_f64_rate.bin ) > gpurun_out/r4a/mfma_f64_rate.txt 2>&1
( SPP=50 timeout 600 python tools/section_probe.py ) > gpurun_out/r4a/section_probe.txt 2>&1
( timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --cpu-seconds 6 ) > gpurun_out/r4a/bench_2dev.json 2> gpurun_out/r4a/bench_2dev.err
( timeout 900 python bench.py --steps 5 --warmup 1 --cpu-seconds 6 ) > gpurun_out/r4a/bench_c3.json 2> gpurun_out/r4a/bench_c3.err
tail -3 gpurun_out/r4a/pytest_round4.txt
cat gpurun_out/r4a/mfma_f64_rate.txt
