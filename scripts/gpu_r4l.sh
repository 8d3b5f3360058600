#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4l; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "grouped_3x3" 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_bnfuse_block.py -q 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_parity_full.py -q -k "c3" 2>&1 | tail -15
python bench.py --only c3 --no-pmc --no-cpu-baseline --repeats 3 --detail $O/c3_detail.json > $O/c3.json 2> $O/c3.err; echo "c3 rc=$?"; tail -3 $O/c3.err
python - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r4l/c3_detail.json")))["sections"]["c3"]
print("c3", d["value"], d["ms_per_step"], "fallbacks", d.get("stock_fallbacks"))
for k, v in list(d["kernels"].items())[:30]:
    print("   %-34s %7.3f ms/step %5.1f x %7.1f us" % (k[:34], v["ms_per_step"], v["launches_per_step"], v["avg_us"]))
PY
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -o c3 -- python $GRAFT_REPO_ROOT/bench.py --only c3 --steps 20 --warmup 5 --repeats 1 --no-pmc --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/$O/rocprof_c3.log 2>&1)
F=$(find /tmp/prof_c3 -name "*kernel_stats.csv" | head -1); cp "$F" $O/c3_kernel_stats.csv
python - <<'PY'
import csv, os
rows = list(csv.DictReader(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r4l/c3_kernel_stats.csv"))))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:45]:
    print("%-64s %6d %8.1f us %5.1f%%" % (r["Name"][:64], int(r["Calls"]), float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
