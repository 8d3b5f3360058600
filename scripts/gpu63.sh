#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/pytest63.log 2>&1; grep -E "^E  |FAILED|passed|failed" gpurun_out/pytest63.log | cut -c1-300 | head
for v in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench63_$v.json 2>/dev/null; cut -c60-200 gpurun_out/bench63_$v.json
done
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench63_2.json').read().strip().splitlines()[-1])
print({k: (x['launches_per_step'], x['avg_us']) for k, x in d['kernels'].items() if 'c1' in k or 'wgrad_s' in k})
PY
