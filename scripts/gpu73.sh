#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in 1024 512 1024 512; do
  echo "=== MN_PW_CAP=$v ==="
  MN_PW_CAP=$v timeout 300 python bench.py --workload c1 --no-cpu-baseline --steps 15 --warmup 3 2>/dev/null | tail -1 > gpurun_out/bench73_$v.json; cut -c1-70,100-175 gpurun_out/bench73_$v.json
done
python - <<'PY'
import json
for v in (1024, 512):
    d = json.loads(open('gpurun_out/bench73_%d.json'%v).read().strip().splitlines()[-1])
    print(v, {k: (x['launches_per_step'], x['avg_us']) for k, x in d['kernels'].items() if k.startswith('k_pw<')})
PY
