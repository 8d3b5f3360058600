#!/bin/bash
# round 4, GPU call E: the whole GPU suite (no -x) + c3 / c5 bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4e; mkdir -p $O
python -m pytest tests -m gpu -q > $O/t_all.log 2>&1; echo "full gpu suite rc=$?"; tail -12 $O/t_all.log
python bench.py --only c3 --no-pmc --no-cpu-baseline --detail $O/c3_detail.json > $O/c3.json 2> $O/c3.err; echo "c3 rc=$?"; tail -c 260 $O/c3.json
python bench.py --only c5 --no-pmc --no-cpu-baseline --detail $O/c5_detail.json > $O/c5.json 2> $O/c5.err; echo "c5 rc=$?"; tail -c 260 $O/c5.json
python - <<'PY'
import json, os
for w in ("c3", "c5"):
    d = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r4e/%s_detail.json" % w)))
    s = d["sections"][w]
    print(w, s["value"], s["ms_per_step"], "fallbacks", s.get("stock_fallbacks"))
    for k, v in list(s["kernels"].items())[:14]:
        print("   %-34s %7.3f ms/step %5.1f x %7.1f us %7.1f GB/s" % (k[:34], v["ms_per_step"], v["launches_per_step"], v["avg_us"], v["GBps"]))
PY
