#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/pytest75.log 2>&1; grep -E "^E  |FAILED|passed|failed" gpurun_out/pytest75.log | cut -c1-300 | head -12
for w in c1 c2; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --steps 15 --warmup 3 2>/dev/null | tail -1 > gpurun_out/bench75_$w.json; echo $w; cut -c1-70,100-175 gpurun_out/bench75_$w.json
done
