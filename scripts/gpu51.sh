#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "=== pytest gpu (all) ==="
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/pytest51.log 2>&1; grep -E "^E  |FAILED|passed|failed" gpurun_out/pytest51.log | cut -c1-300 | head -20
for v in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench51_$v.json 2>/dev/null; cut -c1-200 gpurun_out/bench51_$v.json
done
