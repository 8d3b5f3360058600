#!/bin/bash
# round 4, GPU call F: first-layer patch Gram, scalar loads in the small kernels
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4g; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -k "iaobf or fq_maxpool or first_layer" -q > $O/t_kernels.log 2>&1; echo "kernels rc=$?"; tail -2 $O/t_kernels.log
python -m pytest tests/test_gpu_bnfuse_block.py tests/test_gpu_iao_ops.py -q > $O/t_block.log 2>&1; echo "block rc=$?"; tail -2 $O/t_block.log
python -m pytest tests/test_gpu_parity_full.py -k c3 -q > $O/t_parity_c3.log 2>&1; echo "parity c3 rc=$?"; tail -2 $O/t_parity_c3.log
python -m pytest tests/test_gpu_models.py -k "c3 or fall_through" -q > $O/t_models.log 2>&1; echo "models rc=$?"; tail -2 $O/t_models.log
python bench.py --only c3 --no-pmc --no-cpu-baseline --detail $O/c3_detail.json > $O/c3.json 2> $O/c3.err; echo "c3 rc=$?"; tail -c 200 $O/c3.json
mkdir -p /tmp/prof_c3
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -o c3 -- python $GRAFT_REPO_ROOT/bench.py --only c3 --steps 20 --warmup 5 --repeats 1 --no-pmc --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/$O/rocprof_c3.log 2>&1)
F=$(find /tmp/prof_c3 -name "*kernel_stats.csv" | head -1); cp "$F" $O/c3_kernel_stats.csv
python - <<'PY'
import csv, os
p = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r4g/c3_kernel_stats.csv")
rows = list(csv.DictReader(open(p)))
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print("total ms/step (28 steps): %.3f, launches/step %.0f" % (tot / 1e6 / 28, calls / 28))
for r in rows[:30]:
    print("%-80s %6d %9.3f ms %8.1f us %5.2f%%" % (r["Name"][:80], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
