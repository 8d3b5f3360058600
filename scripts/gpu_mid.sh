#!/bin/bash
# Mid-round check of a new path on a GPU box (short: GPU minutes are scarce): the named tests in ONE pytest process, then one workload's bench line with its per-kernel table.
#   bash scripts/gpu_mid.sh "<pytest -k expression>" <workload>
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/mid; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_parity_full.py -m gpu -q --tb=short -p no:cacheprovider -s -k "$1" > $O/pytest.log 2>&1
grep -E "^E  |FAILED|ERROR|passed|failed|worst rel" $O/pytest.log | cut -c1-700 | head -30
timeout 300 python bench.py --only ${2:-c1} --no-pmc --no-cpu-baseline --detail $O/bench_detail.json > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 1800 $O/bench.json; tail -5 $O/bench.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/mid/bench_detail.json"))
    for w, sec in d["sections"].items():
        print(w, sec.get("value"), sec.get("ms_per_step"), sec.get("stock_fallbacks"))
        for k, v in list(sec.get("kernels", {}).items())[:28]:
            print("  %-36s %.4f ms x%.0f %.1f us %s GB/s" % (k, v["ms_per_step"], v["launches_per_step"], v["avg_us"], v.get("GBps")))
except Exception as e:
    print("no detail:", e)
PY
