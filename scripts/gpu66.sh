#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/pytest66.log 2>&1; grep -E "^E  |FAILED|passed|failed" gpurun_out/pytest66.log | cut -c1-300 | head
for v in old new old new; do
  if [ $v = old ]; then export MN_NO_STATS_H=1; else unset MN_NO_STATS_H; fi
  timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench66_$v.json 2>/dev/null; echo $v; cut -c60-200 gpurun_out/bench66_$v.json
done
python - <<'PY'
import json
for v in ('old','new'):
    d = json.loads(open('gpurun_out/bench66_%s.json'%v).read().strip().splitlines()[-1])
    print(v, {k: (x['launches_per_step'], x['avg_us']) for k, x in d['kernels'].items() if k.startswith('k_pws<') or k.startswith('k_h_')})
PY
