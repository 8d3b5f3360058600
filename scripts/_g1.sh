cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out/fg
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -q --tb=short -p no:cacheprovider -k "first_conv or first_block" > gpurun_out/fg/k.log 2>&1; tail -5 gpurun_out/fg/k.log | cut -c1-300
MN_WORKLOADS='c2' bash scripts/gpu_check.sh prof > /dev/null 2>&1; grep -E 'k_c1' gpurun_out/prof_c2_summary.md | cut -c1-200; tail -1 gpurun_out/prof_c2_summary.md
