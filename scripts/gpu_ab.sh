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
# pass "workload:label:ENV=VALUE" triples on the command line, e.g. (the knobs that remain are listed in micronet_amd/csrc/common.h):
#   bash scripts/gpu_ab.sh c2:base c2:two_kernel_backward:MN_PWB=0 c1_w2a2:base c1_w2a2:two_kernel_backward:MN_PWB=0
#   bash scripts/gpu_ab.sh c4:base c4:exact_terms:MN_GRAD_TERMS=3
for spec in "$@"; do
  IFS=: read -r w l e <<< "$spec"
  run "$w" "$l" "${e:-MN_X=0}"
done
