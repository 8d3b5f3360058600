#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in 0 1 0 1; do
  echo "=== strided=$v ==="
  if [ $v = 1 ]; then export MN_WG2_STRIDED=1; else unset MN_WG2_STRIDED; fi
  timeout 120 python scripts/kbench.py --scheme sign8 --layers L2,L5,L8 --algos 3 --which wgrad --iters 30 2>&1 | grep wgrad
done
