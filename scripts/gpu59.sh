#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in 512 256 1024; do
  echo "=== blocks $v ==="
  MN_K3F_BLOCKS=$v MN_K3D_BLOCKS=$v timeout 300 python bench.py --no-cpu-baseline --steps 20 > gpurun_out/bench59_$v.json 2>/dev/null; cut -c1-200 gpurun_out/bench59_$v.json | cut -c60-200
  python - <<PY
import json
d = json.loads(open('gpurun_out/bench59_$v.json').read().strip().splitlines()[-1])
print({k: (x['launches_per_step'], x['avg_us']) for k, x in d['kernels'].items() if k.startswith('k_k3s')})
PY
done
