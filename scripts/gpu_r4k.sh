#!/bin/bash
mkdir -p gpurun_out/r4k
timeout 800 python -m pytest tests/test_gpu_parity_resnet.py -q -k "c5" -x 2>&1 | tail -30
