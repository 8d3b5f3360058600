mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
R=$GRAFT_REPO_ROOT
echo "=== pytest graphed ==="; timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q --tb=short -p no:cacheprovider -k "graphed" -s > gpurun_out/graphed.log 2>&1; grep -v Warning gpurun_out/graphed.log | grep -n "eager\|Error\|assert\|passed\|failed" | head -20
echo "=== bench ==="
timeout 900 python bench.py --steps 30 --warmup 5 --cpu-steps 4 2>&1 | tail -3 > gpurun_out/bench17.json; cut -c1-3000 gpurun_out/bench17.json


