#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "=== pytest gpu (first-layer, models) ==="
timeout 900 python -m pytest tests -q -m gpu -k "first or model or smoke or graph" > gpurun_out/pytest49.log 2>&1; grep -E "^E  |FAILED|passed|failed" gpurun_out/pytest49.log | cut -c1-300 | head -20
for v in old new old new; do
  echo "=== $v ==="
  if [ $v = old ]; then export MN_LIB_PATH=$PWD/micronet_amd/lib/libmicronet_hip_old.so; else unset MN_LIB_PATH; fi
  timeout 100 python scripts/kbench.py --scheme real --layers L1 --algos 0 --iters 30 2>&1 | grep -E "fwd|wgrad"
  timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench49_$v.json 2>/dev/null; cut -c1-200 gpurun_out/bench49_$v.json
done
