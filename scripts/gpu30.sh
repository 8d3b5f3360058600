export PYTHONDONTWRITEBYTECODE=1
for mw in 4 2; do for z in 256 512; do echo "== MW=$mw Z=$z"; MN_WG2_MW=$mw MN_WG2_Z=$z python scripts/kbench.py --layers L2,L5,L8 --algos 3 --scheme sign8 2>&1 | grep -v amdgpu.ids | grep wgrad; done; done
