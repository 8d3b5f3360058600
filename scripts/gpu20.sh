mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 1200 python -m pytest tests/test_gpu_modules.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu20.log 2>&1; echo "pytest rc=$?"; grep -v Warning gpurun_out/pytest_gpu20.log | tail -8
