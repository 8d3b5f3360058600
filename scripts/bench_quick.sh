#!/bin/bash
# quick images/s of the given workloads (default: all seven), no PMC / CPU legs:  bash scripts/bench_quick.sh [c2 c1_w2a2 ...]   (MN_LIB_PATH selects a variant library)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
[ $# -eq 0 ] && set -- c2 c2b c1_w2a2 c1 c3 c4 c5
for w in "$@"; do
  python bench.py --only $w --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --no-dp-single --detail gpurun_out/quick_$w.json 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['value'], d['ms_per_step'])"
done
