#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -q -m gpu -k "classifier or SignIn or sign_in or packed" > gpurun_out/pytest53.log 2>&1; grep -E "^E  |FAILED|passed|failed" gpurun_out/pytest53.log | cut -c1-300 | head
for v in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench53_$v.json 2>/dev/null; cut -c1-200 gpurun_out/bench53_$v.json
done
python - <<'PY'
import json
d = json.loads(open('gpurun_out/bench53_2.json').read().strip().splitlines()[-1])
for k in ('k_sconv_fwd','k_sconv_dgrad','k_k3s_dgrad','k_pws<4, 4, 1>'):
    print(k, d['kernels'].get(k))
PY
