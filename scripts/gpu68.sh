#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "=== graphed-step tests with the weight stream ==="
MN_WEIGHT_STREAM=1 timeout 600 python -m pytest tests/test_gpu_models.py -q -m gpu -k "graphed" -s 2>&1 | grep -E "eager|passed|failed|Error" | cut -c1-300 | head
for v in 0 1 0 1; do
  echo "=== MN_WEIGHT_STREAM=$v ==="
  MN_WEIGHT_STREAM=$v timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench68_$v.json 2>gpurun_out/bench68_$v.err; cut -c60-330 gpurun_out/bench68_$v.json; grep -i -E "error|Traceback" gpurun_out/bench68_$v.err | head -3
done
