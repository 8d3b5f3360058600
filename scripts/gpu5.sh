mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
echo "=== pytest gpu (all) ==="; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu5.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/pytest_gpu5.log
echo "=== bench ==="
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1500
