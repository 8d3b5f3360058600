#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "=== 2 ranks on one GPU (gloo), graphed ==="
MN_DIST_BACKEND=gloo MN_BENCH_SHARE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -2 | cut -c1-400
echo "=== 2 ranks, eager ==="
MN_DIST_BACKEND=gloo MN_BENCH_SHARE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-graph 2>&1 | tail -2 | cut -c1-400
