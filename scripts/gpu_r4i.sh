#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4i; mkdir -p $O
python scripts/dbg_c5_whole.py conv2_x.1 2>&1 | grep -v Warn | tail -20
echo "---- producer min/max hand-over off"
MN_NO_PRODUCER_MINMAX=1 python scripts/dbg_c5_whole.py conv2_x.1 2>&1 | grep -v Warn | tail -6
python bench.py --only c3 --no-pmc --no-cpu-baseline --repeats 3 --detail $O/c3_detail.json > $O/c3.json 2> $O/c3.err; echo "c3 rc=$?"
python - <<'PY'
import json, os
d = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r4i/c3_detail.json")))["sections"]["c3"]
print("c3", d["value"], d["ms_per_step"])
for k, v in list(d["kernels"].items())[:12]:
    print("   %-34s %7.3f ms/step %5.1f x %7.1f us" % (k[:34], v["ms_per_step"], v["launches_per_step"], v["avg_us"]))
PY
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -o c3 -- python $GRAFT_REPO_ROOT/bench.py --only c3 --steps 20 --warmup 5 --repeats 1 --no-pmc --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/$O/rocprof_c3.log 2>&1)
F=$(find /tmp/prof_c3 -name "*kernel_stats.csv" | head -1); cp "$F" $O/c3_kernel_stats.csv
python - <<'PY'
import csv, os
rows = list(csv.DictReader(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r4i/c3_kernel_stats.csv"))))
for r in rows:
    if any(t in r["Name"] for t in ("gram_reduce", "gram_stats", "k_bf_M", "prep", "gram<")):
        print("%-60s %6d %8.1f us" % (r["Name"][:60], int(r["Calls"]), float(r["AverageNs"]) / 1e3))
PY
