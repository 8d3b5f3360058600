mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
echo "=== pytest gpu (models) ==="; timeout 1500 python -m pytest tests/test_gpu_models.py -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/pytest_gpu3.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/pytest_gpu3.log | tail -3
echo "=== rocprof bench ==="
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof3 -o c2 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench3_prof.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/bench3_prof.log | cut -c1-1200
find gpurun_out/prof3 -type f | head
f=$(find gpurun_out/prof3 -name "*kernel_stats.csv" | head -1); echo "stats file: $f"; head -45 "$f" | cut -c1-200
for t in 4 8 16; do timeout 300 python bench.py --cpu-only --cpu-threads $t --cpu-batch 64 --cpu-steps 2 2>/dev/null | tail -1 | cut -c1-100; done
