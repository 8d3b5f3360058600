mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu35.log 2>&1; echo "pytest rc=$?"; grep -v Warning gpurun_out/pytest_gpu35.log | tail -4
