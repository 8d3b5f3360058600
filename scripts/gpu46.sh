#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "=== pytest gpu (all) ==="
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/pytest46.log 2>&1; grep -E "^E  |FAILED|passed|failed" gpurun_out/pytest46.log | cut -c1-300 | head -20
for v in old new old new; do
  echo "=== $v ==="
  if [ $v = old ]; then export MN_LIB_PATH=$PWD/micronet_amd/lib/libmicronet_hip_old.so; else unset MN_LIB_PATH; fi
  timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench46_$v.json 2>/dev/null; cut -c1-200 gpurun_out/bench46_$v.json
done
