#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/pytest69.log 2>&1; grep -E "^E  |FAILED|passed|failed" gpurun_out/pytest69.log | cut -c1-300 | head
for v in 0 1 0 1; do
  echo "=== MN_MULTI_WQ=$v ==="
  MN_MULTI_WQ=$v timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench69_$v.json 2>gpurun_out/bench69_$v.err; cut -c60-200 gpurun_out/bench69_$v.json; grep -i -E "Error|Traceback" gpurun_out/bench69_$v.err | head -3
done
echo "=== 2 ranks gloo ==="
MN_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | grep '"metric"' | cut -c1-200
