#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4n; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "qdense or first_layer_forward_with" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_parity_resnet.py -q -k "c5" 2>&1 | tail -5
timeout 600 python -m pytest tests/test_gpu_bnfuse_block.py -q 2>&1 | tail -3
for w in c5 c3; do
python bench.py --only $w --no-pmc --no-cpu-baseline --repeats 3 --detail $O/${w}_detail.json > $O/$w.json 2> $O/$w.err; echo "$w rc=$?"
done
MN_QD_STE_SEPARATE=1 python bench.py --only c5 --no-pmc --no-cpu-baseline --repeats 3 --detail $O/c5_sep_detail.json > $O/c5_sep.json 2> $O/c5_sep.err
python - <<'PY'
import json, os
for w, f in (("c5", "c5_detail"), ("c5", "c5_sep_detail"), ("c3", "c3_detail")):
    d = json.load(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r4n/%s.json" % f)))["sections"][w]
    print(f, d["value"], d["ms_per_step"], "fallbacks", d.get("stock_fallbacks"))
    for k, v in list(d["kernels"].items())[:8]:
        print("   %-34s %7.3f ms/step %5.1f x %7.1f us" % (k[:34], v["ms_per_step"], v["launches_per_step"], v["avg_us"]))
PY
