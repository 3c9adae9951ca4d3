set -x
mkdir -p gpurun_out/r4f
( SPP=100 timeout 1200 python tools/fanout_sweep.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r4f/fanout_sweep.txt 2>&1
cat gpurun_out/r4f/fanout_sweep.txt
