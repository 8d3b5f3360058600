#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
R=$GRAFT_REPO_ROOT
cd $R
echo "=== pytest gpu (all) ==="; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu71.log 2>&1; echo "pytest rc=$?"; grep -v Warning gpurun_out/pytest_gpu71.log | tail -3
echo "=== smoke ==="; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "=== PMC traffic ==="; bash scripts/pmc_traffic.sh c2 > gpurun_out/pmc71.log 2>&1; tail -2 gpurun_out/pmc71.log
cp gpurun_out/traffic_c2.json profiles/traffic_c2.json
echo "=== bench ==="
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench71.json; cut -c1-900 gpurun_out/bench71.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof71 -o c2 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/bench71_prof.log 2>&1
cd $R
tail -1 gpurun_out/bench71_prof.log | cut -c1-250
find gpurun_out/prof71 -name '*kernel_trace.csv' -size +20M -delete
echo "=== other workloads ==="
for w in c1 c1_w2a2 c3 c4 c5; do
  timeout 400 python bench.py --workload $w --no-cpu-baseline --steps 15 --warmup 3 2>/dev/null | tail -1 > gpurun_out/bench71_$w.json; echo $w; cut -c1-260 gpurun_out/bench71_$w.json | cut -c1-70,100-260
done
