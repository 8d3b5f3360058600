#!/bin/bash
# LDS-staged wgrad on sign codes: parity + A/B against the direct-load kernel
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "=== pytest gpu kernels (sign / bnh / wgrad) ==="
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "sign or bnh or wgrad" 2>&1 | tail -4
for v in direct staged; do
  if [ $v = direct ]; then export MN_WG2_DIRECT=1; else unset MN_WG2_DIRECT; fi
  echo "=== $v ==="
  timeout 120 python scripts/kbench.py --scheme sign8 --layers L2,L5,L8 --algos 3 --which wgrad --iters 30 2>&1 | grep wgrad
done
for z in 128 64 32; do
  echo "=== staged Z=$z ==="
  MN_WG2_Z=$z timeout 120 python scripts/kbench.py --scheme sign8 --layers L2,L5,L8 --algos 3 --which wgrad --iters 30 2>&1 | grep wgrad
done
echo "=== bench staged ==="
timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
echo "=== bench direct ==="
MN_WG2_DIRECT=1 timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400
