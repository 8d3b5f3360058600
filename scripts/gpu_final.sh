#!/bin/bash
# The round's evidence in one GPU-box call: GPU suite, batch-256 parity record, default bench line (PMC + CPU legs), rocprofv3 summaries of all seven workloads,
# and the two-rank line under gloo on the one GPU (functional check of the N > 1 path).   bash scripts/gpu_final.sh
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
MN_WORKLOADS="c2 c2b c1_w2a2 c1 c3 c4 c5" bash scripts/gpu_check.sh tests bench prof
( time python bench.py --steps 20 --warmup 5 --detail gpurun_out/bench_detail_steps20.json > gpurun_out/bench_driver_like.json 2> gpurun_out/bench_driver_like.err ) 2> gpurun_out/bench_driver_like.time
tail -3 gpurun_out/bench_driver_like.time
MN_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --detail gpurun_out/bench_detail_gloo2.json \
  > gpurun_out/bench_gloo2.json 2> gpurun_out/bench_gloo2.err
tail -c 600 gpurun_out/bench_gloo2.json
