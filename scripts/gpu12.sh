mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
R=$GRAFT_REPO_ROOT
echo "=== pytest gpu ==="; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu13.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/pytest_gpu13.log
echo "=== bench ==="
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof13 -o c2 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/bench12_prof.log 2>&1
cd $R
python scripts/step_breakdown.py gpurun_out/prof13/c2_kernel_trace.csv 24
