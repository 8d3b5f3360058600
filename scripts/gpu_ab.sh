#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=gpurun_out/ab; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "first" 2>&1 | tail -3
for v in new f32; do
  [ $v = f32 ] && export MN_C1_F32=1
  for w in c2 c3; do
    python bench.py --only $w --no-pmc --no-cpu-baseline --repeats 3 --detail $O/${w}_${v}.json > $O/${w}_${v}.out 2> $O/${w}_${v}.err
    python - $w $v <<'PY'
import json, os, sys
w, v = sys.argv[1], sys.argv[2]
d = json.load(open("gpurun_out/ab/%s_%s.json" % (w, v)))["sections"][w]
ks = {k: x for k, x in d["kernels"].items() if "c1" in k}
print(w, v, d["value"], d["ms_per_step"], {k: x["avg_us"] for k, x in ks.items()})
PY
  done
done
