#!/bin/bash
# round 4, GPU call H: LDS-staged small kernels; Gram block-count A/B; c3 kernel trace; c5 parity (whole-block stage)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4h; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -k "iaobf or first_layer" -q > $O/t_kernels.log 2>&1; echo "kernels rc=$?"; tail -2 $O/t_kernels.log
python -m pytest tests/test_gpu_bnfuse_block.py tests/test_gpu_models.py -q > $O/t_block.log 2>&1; echo "block+models rc=$?"; tail -3 $O/t_block.log
python -m pytest tests/test_gpu_parity_full.py -k c3 -q > $O/t_parity_c3.log 2>&1; echo "parity c3 rc=$?"; tail -2 $O/t_parity_c3.log
python -m pytest tests/test_gpu_parity_resnet.py -k "c5 or iao" -q > $O/t_parity_c5.log 2>&1; echo "parity c5 rc=$?"; tail -4 $O/t_parity_c5.log
for nb in 512 256 1024; do
  MN_GRAM_BLOCKS=$nb python bench.py --only c3 --no-pmc --no-cpu-baseline --repeats 3 --detail $O/c3_g$nb.json > $O/c3_g$nb.line 2> /dev/null
  python - "$nb" <<'PY'
import json, os, sys
nb = sys.argv[1]
d = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r4h/c3_g%s.json" % nb)))["sections"]["c3"]
k = d["kernels"]
pick = lambda n: next((v for kk, v in k.items() if kk.startswith(n)), {"avg_us": 0})
print("gram blocks", nb, "c3", d["value"], "img/s  gram", pick("k_bf_gram<4")["avg_us"], "us  dgrad", pick("k_bf_dgrad")["avg_us"])
PY
done
mkdir -p /tmp/prof_c3
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c3 -o c3 -- python $GRAFT_REPO_ROOT/bench.py --only c3 --steps 20 --warmup 5 --repeats 1 --no-pmc --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/$O/rocprof_c3.log 2>&1)
F=$(find /tmp/prof_c3 -name "*kernel_stats.csv" | head -1); cp "$F" $O/c3_kernel_stats.csv
python - <<'PY'
import csv, os
p = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r4h/c3_kernel_stats.csv")
rows = list(csv.DictReader(open(p)))
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print("total ms/step (28 steps): %.3f, launches/step %.0f" % (tot / 1e6 / 28, calls / 28))
for r in rows[:24]:
    print("%-80s %6d %9.3f ms %8.1f us %5.2f%%" % (r["Name"][:80], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
