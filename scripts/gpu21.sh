mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -m gpu -q -p no:cacheprovider -k "first" 2>&1 | tail -4
python scripts/kbench.py --layers L1 --algos 0 --scheme real 2>&1 | grep -v amdgpu.ids | tail -3
echo "=== bench ==="; timeout 900 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench21.json; cut -c1-2600 gpurun_out/bench21.json
