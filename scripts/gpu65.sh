#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in 4 2; do
  for z in 0 128 64; do
    echo "=== MW=$v Z=$z ==="
    if [ $z = 0 ]; then unset MN_WG2_Z; else export MN_WG2_Z=$z; fi
    MN_WG2_MW=$v timeout 120 python scripts/kbench.py --scheme sign8 --layers L2,L5,L8 --algos 3 --which wgrad --iters 30 2>&1 | grep wgrad
  done
done
