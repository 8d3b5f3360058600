mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
echo "=== rocminfo ==="; rocminfo | grep -E "gfx|Compute Unit|Marketing" | head -8
echo "=== smoke ==="; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -5 gpurun_out/smoke.log
echo "=== pytest gpu ==="; timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -40 gpurun_out/pytest_gpu.log
echo "=== bench ==="; timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench1.log 2>&1; echo "bench rc=$?"; tail -5 gpurun_out/bench1.log
