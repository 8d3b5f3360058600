mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
echo "=== pytest gpu ==="; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu7.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/pytest_gpu7.log
echo "=== kbench ==="
timeout 600 python scripts/kbench.py --algos 3 --layers L4,L7 2>&1 | tail -8
timeout 600 python scripts/kbench.py --algos 3 --scheme iao --layers L1 2>&1 | tail -3
echo "=== bench ==="
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-600
