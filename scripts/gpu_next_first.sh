#!/bin/bash
# First GPU call of the next round: hardware validation + A/B of the kernels that were written and emulator-checked without GPU time (DESIGN §7 "What comes next").
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash scripts/gpu_next_first.sh > gpurun_out/next_first.log 2>&1'
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
# 1. do they run correctly on the hardware?  (--runxfail: a failure here must be a failure)
timeout 600 python -m pytest tests/test_gpu_zz_unmeasured.py -m gpu -q --runxfail -x 2>&1 | tail -15
# 2. do they pay?  (same box, back to back; identical final_loss in the json = bit-identical step)
bash scripts/gpu_ab.sh c4:base c4:wgrad32:MN_QD_WGRAD32=1 c5:base c5:wgrad32:MN_QD_WGRAD32=1 c2:base c2:hsfold:MN_HSIGN_FOLD=1 c2:base2 c2:hsfold2:MN_HSIGN_FOLD=1
# 3. kernel level: the dense backward-weight on the four resnet18 layer shapes, default kernel vs MN_QD_WGRAD32=1
timeout 300 python scripts/bench_qd_wgrad.py 2>&1 | tail -8
