mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
R=$GRAFT_REPO_ROOT
echo "=== pytest modules ==="; timeout 900 python -m pytest tests/test_gpu_modules.py -m gpu -q --tb=short -p no:cacheprovider -k "fused_conv_bn or graphed" -s 2>&1 | grep -v Warning | tail -5
for i in 1; do
echo "=== bench fused ==="; timeout 900 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | cut -c1-220
echo "=== bench unfused ==="; MN_BENCH_WBWTAB_KW="fuse_conv_bn=0" timeout 900 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | cut -c1-220
done
