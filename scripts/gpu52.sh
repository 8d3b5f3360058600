#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "=== pytest gpu kernels ==="
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "3x3 or kxk or hot or sign8" > gpurun_out/pytest52.log 2>&1; grep -E "^E  |FAILED|passed|failed" gpurun_out/pytest52.log | cut -c1-300 | head -20
for v in old new old new; do
  echo "=== $v ==="
  if [ $v = old ]; then export MN_NO_K3D=1; else unset MN_NO_K3D; fi
  timeout 120 python scripts/kbench.py --scheme sign8 --layers L4,L7 --algos 3 --which dgrad --iters 30 2>&1 | grep dgrad
done
for v in old new; do
  if [ $v = old ]; then export MN_NO_K3D=1; else unset MN_NO_K3D; fi
  timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench52_$v.json 2>/dev/null; cut -c1-200 gpurun_out/bench52_$v.json
done
