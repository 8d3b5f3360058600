export PYTHONDONTWRITEBYTECODE=1
python scripts/kbench.py --layers L1 --algos 0 --scheme real 2>&1 | grep -v amdgpu.ids | grep "fwd\|wgrad"
