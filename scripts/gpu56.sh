#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in 1024 512 2048; do
  echo "=== MN_PWD_CAP=$v ==="
  MN_PWD_CAP=$v timeout 200 python scripts/kbench.py --scheme sign8 --layers L2,L5,L8 --algos 3 --which dgrad --iters 30 2>&1 | grep dgrad
done
for v in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench56_$v.json 2>/dev/null; cut -c1-200 gpurun_out/bench56_$v.json
done
