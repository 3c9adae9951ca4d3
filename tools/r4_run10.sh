set -x
mkdir -p gpurun_out/r4j
( timeout 2700 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -12 ) > gpurun_out/r4j/pytest_gpu.txt 2>&1
cat gpurun_out/r4j/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
