mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
R=$GRAFT_REPO_ROOT
echo "=== pytest gpu (all) ==="; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu25.log 2>&1; echo "pytest rc=$?"; grep -v Warning gpurun_out/pytest_gpu25.log | tail -4
echo "=== smoke ==="; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "=== PMC traffic ==="; bash scripts/pmc_traffic.sh c2 2>&1 | tail -30
echo "=== bench ==="
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/bench25.json; cut -c1-600 gpurun_out/bench25.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof25 -o c2 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph > $R/gpurun_out/bench25_prof.log 2>&1
cd $R
tail -1 gpurun_out/bench25_prof.log | cut -c1-250
python scripts/step_breakdown.py $(find gpurun_out/prof25 -name '*kernel_trace.csv' | head -1) 45
find gpurun_out/prof25 -name '*kernel_trace.csv' -size +20M -delete
rm -rf gpurun_out/pmc_FETCH_SIZE/*/*.db gpurun_out/pmc_WRITE_SIZE/*/*.db 2>/dev/null
