mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
R=$GRAFT_REPO_ROOT
echo "=== pytest gpu ==="; timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu14.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_gpu14.log
echo "=== smoke ==="; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
echo "=== bench ==="
timeout 600 python bench.py --steps 30 --warmup 5 2>&1 | tail -1 > gpurun_out/bench14.json; cut -c1-2500 gpurun_out/bench14.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof14 -o c2 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/bench14_prof.log 2>&1
cd $R
tail -1 gpurun_out/bench14_prof.log | cut -c1-200
find gpurun_out/prof14 -name '*kernel_stats.csv' | head
python scripts/step_breakdown.py $(find gpurun_out/prof14 -name '*kernel_trace.csv' | head -1) 30
# keep the merge-back under the size cap: drop the raw trace, keep the stats
find gpurun_out/prof14 -name '*kernel_trace.csv' -size +20M -delete
