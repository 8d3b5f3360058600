export PYTHONDONTWRITEBYTECODE=1
python scripts/kbench.py --layers L2,X2,X2b,L5,X5 --algos 3 --scheme sign8 2>&1 | grep -v amdgpu.ids
echo "== old wgrad"; MN_NO_WG2=1 python scripts/kbench.py --layers L2,X2,X2b --algos 3 --scheme sign8 2>&1 | grep -v amdgpu.ids | grep wgrad
