export PYTHONDONTWRITEBYTECODE=1
python scripts/kbench.py --layers L1 --algos 3,0 --scheme real 2>&1 | tail -7
