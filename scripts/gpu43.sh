#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "=== pytest gpu (all) ==="
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
for v in 1 0 1 0; do
  echo "=== MN_NO_KXK_STASH=$v ==="
  if [ $v = 1 ]; then export MN_NO_KXK_STASH=1; else unset MN_NO_KXK_STASH; fi
  timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench43_$v.json 2>gpurun_out/bench43_$v.err; cut -c1-200 gpurun_out/bench43_$v.json; tail -2 gpurun_out/bench43_$v.err
done
