#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in 0 1 0 1; do
  echo "=== MN_BNH_FOLD=$v ==="
  MN_BNH_FOLD=$v timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench42_$v.json 2>/dev/null; cut -c1-200 gpurun_out/bench42_$v.json
done
