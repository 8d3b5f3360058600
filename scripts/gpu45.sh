#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "=== pytest gpu (all) ==="
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/pytest45.log 2>&1; grep -E "^E  |FAILED|passed|failed" gpurun_out/pytest45.log | head -20
for v in 1 0; do
  echo "=== MN_NO_KXK_STASH=$v ==="
  if [ $v = 1 ]; then export MN_NO_KXK_STASH=1; else unset MN_NO_KXK_STASH; fi
  timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench45_$v.json 2>/dev/null; cut -c1-200 gpurun_out/bench45_$v.json
done
