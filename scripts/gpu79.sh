#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_modules.py -m gpu -q --tb=short -p no:cacheprovider -k "iao or c3 or c5 or graphed" > gpurun_out/pytest_gpu79.log 2>&1; grep -E "^E  |FAILED|passed|failed" gpurun_out/pytest_gpu79.log | cut -c1-300 | head -8
timeout 300 python bench.py --workload c3 --no-cpu-baseline --steps 12 --warmup 3 2>/dev/null | tail -1 > gpurun_out/bench79_c3.json; cut -c1-70,100-175 gpurun_out/bench79_c3.json
