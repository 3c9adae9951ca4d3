set -x
mkdir -p gpurun_out/r4c
( timeout 2700 python -m pytest tests -q -m gpu 2>&1 | tail -30 ) > gpurun_out/r4c/pytest_gpu.txt 2>&1
tail -5 gpurun_out/r4c/pytest_gpu.txt
