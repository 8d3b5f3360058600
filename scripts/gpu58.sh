#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "3x3 or kxk or hot" > gpurun_out/pytest58.log 2>&1; grep -E "^E  |FAILED|passed|failed" gpurun_out/pytest58.log | cut -c1-300 | head
for v in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench58_$v.json 2>/dev/null; cut -c1-200 gpurun_out/bench58_$v.json
done
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench58_2.json').read().strip().splitlines()[-1])
print({k: (x['launches_per_step'], x['avg_us']) for k, x in d['kernels'].items() if k.startswith('k_k3s') or k.startswith('k_h_')})
PY
