mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
echo "=== pytest gpu kernels ==="; timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu6.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu6.log
echo "=== kbench ==="
timeout 600 python scripts/kbench.py --algos 3 --layers L2,L4,L7,L9 --json gpurun_out/kbench6.json 2>&1 | tail -40
timeout 600 python scripts/kbench.py --algos 3 --scheme iao --layers L1,L2,L4 2>&1 | tail -40
timeout 300 python scripts/diag_wgrad.py 2>&1 | tail -4
