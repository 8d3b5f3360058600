#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu80.log 2>&1; grep -E "^E  |FAILED|passed|failed" gpurun_out/pytest_gpu80.log | cut -c1-300 | head -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c60-200
