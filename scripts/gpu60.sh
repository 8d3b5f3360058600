#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -q -m gpu -k "bnsign or bnh or pool or fused" > gpurun_out/pytest60.log 2>&1; grep -E "^E  |FAILED|passed|failed" gpurun_out/pytest60.log | cut -c1-300 | head
for v in old new old new; do
  if [ $v = old ]; then export MN_NO_BNH_POOLFAST=1; else unset MN_NO_BNH_POOLFAST; fi
  timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench60_$v.json 2>/dev/null; echo $v; cut -c60-200 gpurun_out/bench60_$v.json
done
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench60_new.json').read().strip().splitlines()[-1])
print({k: (x['launches_per_step'], x['avg_us']) for k, x in d['kernels'].items() if 'bnh' in k})
PY
