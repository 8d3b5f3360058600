#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu78.log 2>&1; grep -E "^E  |FAILED|passed|failed" gpurun_out/pytest_gpu78.log | cut -c1-300 | head -8
timeout 300 python bench.py --workload c1 --no-cpu-baseline --steps 15 --warmup 3 2>/dev/null | tail -1 > gpurun_out/bench78_c1.json; cut -c1-70,100-175 gpurun_out/bench78_c1.json
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench78_c1.json').read().strip().splitlines()[-1])
print({k: (x['launches_per_step'], x['avg_us'], x['GBps']) for k, x in d['kernels'].items() if 'pool' in k})
PY
