#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "=== pytest gpu (all) ==="
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/pytest57.log 2>&1; grep -E "^E  |FAILED|passed|failed" gpurun_out/pytest57.log | cut -c1-300 | head -20
for v in old new old new; do
  if [ $v = old ]; then export MN_NO_K3F=1; else unset MN_NO_K3F; fi
  timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench57_$v.json 2>/dev/null; echo $v; cut -c1-200 gpurun_out/bench57_$v.json
done
python - <<'PY'
import json
for v in ('old','new'):
    d = json.loads(open('gpurun_out/bench57_%s.json'%v).read().strip().splitlines()[-1])
    print(v, {k: (x['launches_per_step'], x['avg_us']) for k, x in d['kernels'].items() if k in ('k_k3s_fwd','k_kk<2, 3, 1>','k_h_stats','k_h_sign','k_bnh_partial<0>','k_bnh_apply<0>','k_bnh_apply<1>','k_bnh_partial<1>')})
PY
