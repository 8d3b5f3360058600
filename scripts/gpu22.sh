export PYTHONDONTWRITEBYTECODE=1
echo "== sign8 kernels"; python scripts/kbench.py --layers L2,L4,L5,L7,L8 --algos 3 --scheme sign8 2>&1 | grep -v amdgpu.ids
echo "== wgrad old"; MN_NO_WG2=1 python scripts/kbench.py --layers L2,L8 --algos 3 --scheme sign8 2>&1 | grep -v amdgpu.ids | grep wgrad
echo "=== bench new ==="; timeout 900 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | cut -c1-200
echo "=== bench old wgrad ==="; MN_NO_WG2=1 timeout 900 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | cut -c1-200
