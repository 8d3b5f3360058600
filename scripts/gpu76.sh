#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_modules.py tests/test_gpu_models.py -q -m gpu -k "dorefa or c1 or c4 or graphed" > gpurun_out/pytest76.log 2>&1; grep -E "^E  |FAILED|passed|failed" gpurun_out/pytest76.log | cut -c1-300 | head -12
