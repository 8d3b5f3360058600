export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "bnsign_fused or sign8" 2>&1 | tail -3
for nt in 8 4 2; do echo "== MN_PWS_NT=$nt"; MN_PWS_NT=$nt python scripts/kbench_fused.py 2>&1 | grep -v amdgpu.ids | grep "L2\|L5\|L8"; done
echo "== cap 2048 nt 8"; MN_PWS_CAP=2048 python scripts/kbench_fused.py 2>&1 | grep -v amdgpu.ids | grep "L2\|L8"
echo "== cap 512 nt 8"; MN_PWS_CAP=512 python scripts/kbench_fused.py 2>&1 | grep -v amdgpu.ids | grep "L2\|L8"
