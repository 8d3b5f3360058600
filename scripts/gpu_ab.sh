#!/bin/bash
# A/B of environment knobs on one workload each: bash scripts/gpu_ab.sh   (prints workload, knob, images/s, ms/step)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/ab; mkdir -p $O
run() { # workload, label, env...
  w=$1; l=$2; shift 2
  env "$@" timeout 200 python bench.py --only $w --no-pmc --no-cpu-baseline --no-kernel-timing --repeats 3 > $O/${w}_$l.json 2> $O/${w}_$l.err
  python - "$w" "$l" <<'PY'
import json, sys
w, l = sys.argv[1:3]
try:
    d = json.loads(open("gpurun_out/ab/%s_%s.json" % (w, l)).read().strip().splitlines()[-1])
    print(w, l, d["value"], d["ms_per_step"], d.get("window_ms"))
except Exception as e:
    print(w, l, "failed", e)
PY
}
# edit below: one line per measurement, e.g.
#   run c2 base MN_X=0
#   run c2 nopoolfold MN_BNH_POOL_FOLD=0
#   bash scripts/gpu_ab.sh c4:base c4:wgrad32:MN_QD_WGRAD32=1 c5:base c5:wgrad32:MN_QD_WGRAD32=1     (the unmeasured 32x32x16 backward-weight kernel)
#   bash scripts/gpu_ab.sh c2:base c2:hsfold:MN_HSIGN_FOLD=1 c1_w2a2:base c1_w2a2:hsfold:MN_HSIGN_FOLD=1                  (the unmeasured statistics-finals fold of the sign pass)
for spec in "$@"; do          # or pass "workload:label:ENV=VALUE" triples on the command line
  IFS=: read -r w l e <<< "$spec"
  run "$w" "$l" "${e:-MN_X=0}"
done
