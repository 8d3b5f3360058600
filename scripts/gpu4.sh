mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
echo "=== pytest gpu kernels ==="; timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu4.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu4.log
echo "=== kbench ==="
timeout 600 python scripts/kbench.py --algos 2,3 --json gpurun_out/kbench4.json 2>&1 | tail -40
