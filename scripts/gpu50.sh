#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in 1024 512 256 1024 512; do
  echo "=== MN_C1_BLOCKS=$v ==="
  MN_C1_BLOCKS=$v timeout 100 python scripts/kbench.py --scheme real --layers L1 --algos 0 --iters 30 2>&1 | grep -E "fwd|wgrad"
done
