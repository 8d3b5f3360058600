mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
echo "=== pytest gpu ==="; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/pytest_gpu2.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/pytest_gpu2.log | tail -3
echo "=== smoke ==="; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke2.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke2.log
echo "=== rocprof bench ==="
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof2 -o c2 -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/bench2_prof.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
tail -1 gpurun_out/bench2_prof.log | cut -c1-1500
ls gpurun_out/prof2/* | head
f=$(ls gpurun_out/prof2/*/*kernel_stats.csv 2>/dev/null | head -1); echo "stats file: $f"; head -40 "$f"
echo "=== cpu thread sweep ==="
for t in 16 32 64 128; do timeout 300 python bench.py --cpu-only --cpu-threads $t --cpu-batch 64 --cpu-steps 2 2>/dev/null | tail -1 | cut -c1-200; done
