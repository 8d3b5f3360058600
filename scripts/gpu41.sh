#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "=== pytest gpu kernels (sign8) ==="
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "sign8 or hot" 2>&1 | tail -4
for v in old new; do
  if [ $v = old ]; then export MN_NO_K3S=1; else unset MN_NO_K3S; fi
  echo "=== $v ==="
  timeout 120 python scripts/kbench.py --scheme sign8 --layers L4,L7 --algos 3 --which wgrad --iters 30 2>&1 | grep wgrad
done
for z in 16 8 4; do
  echo "=== new Z=$z ==="
  MN_K3S_Z=$z timeout 120 python scripts/kbench.py --scheme sign8 --layers L4,L7 --algos 3 --which wgrad --iters 30 2>&1 | grep wgrad
done
echo "=== bench ==="
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench41.json 2>gpurun_out/bench41.err; cut -c1-330 gpurun_out/bench41.json
MN_NO_K3S=1 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-330
