#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4o; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "thin_output or grouped_3x3 or first_layer_forward_with or qdense" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_parity_full.py -q -k "c3" > $O/parity_c3.log 2>&1; tail -3 $O/parity_c3.log
timeout 600 python -m pytest tests/test_gpu_bnfuse_block.py tests/test_gpu_models.py -q 2>&1 | tail -5
for w in c3 c5; do
python bench.py --only $w --no-pmc --no-cpu-baseline --repeats 3 --detail $O/${w}_detail.json > $O/$w.json 2> $O/$w.err; echo "$w rc=$?"
done
python - <<'PY'
import json, os
for w, f in (("c5", "c5_detail"), ("c3", "c3_detail")):
    d = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r4o/%s.json" % f)))["sections"][w]
    print(f, d["value"], d["ms_per_step"], "fallbacks", d.get("stock_fallbacks"))
    for k, v in list(d["kernels"].items())[:10]:
        print("   %-34s %7.3f ms/step %5.1f x %7.1f us" % (k[:34], v["ms_per_step"], v["launches_per_step"], v["avg_us"]))
PY
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -o c3 -- python $GRAFT_REPO_ROOT/bench.py --only c3 --steps 20 --warmup 5 --repeats 1 --no-pmc --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/$O/rocprof_c3.log 2>&1)
F=$(find /tmp/prof_c3 -name "*kernel_stats.csv" | head -1); cp "$F" $O/c3_kernel_stats.csv
python - <<'PY'
import csv, os
rows = list(csv.DictReader(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r4o/c3_kernel_stats.csv"))))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("launches/step", sum(int(r["Calls"]) for r in rows) / 28.0, "ms/step", tot / 28e6)
for r in rows[:40]:
    print("%-64s %6d %8.1f us %5.1f%%" % (r["Name"][:64], int(r["Calls"]), float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
