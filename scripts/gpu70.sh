#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "sign8 or bnh or hot" 2>&1 | tail -2
for v in old new old new; do
  echo "=== $v ==="
  if [ $v = old ]; then export MN_LIB_PATH=$PWD/micronet_amd/lib/libmicronet_hip_nosplit.so; else unset MN_LIB_PATH; fi
  timeout 120 python scripts/kbench.py --scheme sign8 --layers L2,L5,L8 --algos 3 --which wgrad --iters 30 2>&1 | grep wgrad
done
for v in old new old new; do
  if [ $v = old ]; then export MN_LIB_PATH=$PWD/micronet_amd/lib/libmicronet_hip_nosplit.so; else unset MN_LIB_PATH; fi
  timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench70_$v.json 2>/dev/null; echo $v; cut -c60-200 gpurun_out/bench70_$v.json
done
