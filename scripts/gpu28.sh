mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
R=$GRAFT_REPO_ROOT
echo "=== pytest gpu (all) ==="; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu28.log 2>&1; echo "pytest rc=$?"; grep -v Warning gpurun_out/pytest_gpu28.log | tail -3
echo "=== smoke ==="; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "=== PMC traffic ==="; bash scripts/pmc_traffic.sh c2 > gpurun_out/pmc28.log 2>&1; tail -3 gpurun_out/pmc28.log
echo "=== bench ==="
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench28.json; cut -c1-900 gpurun_out/bench28.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof28 -o c2 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/bench28_prof.log 2>&1
cd $R
tail -1 gpurun_out/bench28_prof.log | cut -c1-250
head -12 gpurun_out/prof28/c2_kernel_stats.csv | cut -c1-150
find gpurun_out/prof28 -name '*kernel_trace.csv' -size +20M -delete
