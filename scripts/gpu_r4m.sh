#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4m
timeout 900 python -m pytest tests/test_gpu_parity_full.py -q -k "c3" > gpurun_out/r4m/parity_c3.log 2>&1
grep -n "Error\|error\|assert\|FAILED\|passed\|failed" gpurun_out/r4m/parity_c3.log | cut -c1-1500 | tail -20
