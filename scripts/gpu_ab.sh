#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=gpurun_out/ab; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "iaobf" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_bnfuse_block.py -q 2>&1 | tail -2
python bench.py --only c3 --no-pmc --no-cpu-baseline --repeats 3 --detail $O/c3.json > $O/c3.out 2> $O/c3.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/ab/c3.json"))["sections"]["c3"]
print("c3", d["value"], d["ms_per_step"])
for k, v in list(d["kernels"].items())[:8]:
    print("   %-34s %7.3f ms/step %5.1f x %7.1f us" % (k[:34], v["ms_per_step"], v["launches_per_step"], v["avg_us"]))
PY
