#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
echo "=== pytest gpu (all) ==="; timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu77.log 2>&1; echo "pytest rc=$?"; grep -v Warning gpurun_out/pytest_gpu77.log | tail -2
echo "=== smoke ==="; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "=== bench (default) ==="; timeout 900 python bench.py 2>/dev/null | tail -1 > gpurun_out/bench77.json; cut -c1-330 gpurun_out/bench77.json
echo "=== bench c1 ==="; timeout 300 python bench.py --workload c1 --no-cpu-baseline --steps 15 --warmup 3 2>/dev/null | tail -1 > gpurun_out/bench77_c1.json; cut -c1-70,100-175 gpurun_out/bench77_c1.json
