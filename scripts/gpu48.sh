mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
R=$GRAFT_REPO_ROOT
echo "=== pytest gpu (all) ==="; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu48.log 2>&1; echo "pytest rc=$?"; grep -v Warning gpurun_out/pytest_gpu48.log | tail -3
echo "=== smoke ==="; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "=== PMC traffic ==="; bash scripts/pmc_traffic.sh c2 > gpurun_out/pmc48.log 2>&1; tail -2 gpurun_out/pmc48.log
echo "=== bench ==="
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench48.json; cut -c1-700 gpurun_out/bench48.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof48 -o c2 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/bench48_prof.log 2>&1
cd $R
tail -1 gpurun_out/bench48_prof.log | cut -c1-250
find gpurun_out/prof48 -name '*kernel_trace.csv' -size +20M -delete
