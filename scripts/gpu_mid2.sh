#!/bin/bash
# Mid-round validation of a batch of changes: the WHOLE GPU suite (no -x), smoke, then every workload's bench value without the PMC / CPU-baseline passes.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/mid2; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1
grep -E "^E  |FAILED|ERROR|passed|failed" $O/pytest_gpu.log | cut -c1-400 | head -40
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python bench.py --no-pmc --no-cpu-baseline --detail $O/bench_detail.json > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 900 $O/bench.json; echo; tail -3 $O/bench.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/mid2/bench_detail.json"))
    for w, sec in d["sections"].items():
        ks = sec.get("kernels", {})
        print(w, sec.get("value"), sec.get("ms_per_step"), "launches/step (timed kernels)", round(sum(v["launches_per_step"] for v in ks.values()), 1), sec.get("stock_fallbacks"))
        if w in ("c5", "c4"):
            for k, v in list(ks.items())[:14]:
                print("  %-36s %.4f ms x%.0f %.1f us" % (k, v["ms_per_step"], v["launches_per_step"], v["avg_us"]))
except Exception as e:
    print("no detail:", e)
PY
