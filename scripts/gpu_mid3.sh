#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/mid3; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_parity_full.py tests/test_gpu_models.py tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_determinism.py -m gpu -q --tb=short -p no:cacheprovider -n 2 -k "pack_multi or (teacher_forced and (c2_nin or c1_nin_gc_dorefa_w2a2)) or fused_blocks_match_unfused or graphed or c2_nin_gc or c1_nin_gc or wbwtab_fused or lr_schedule" > $O/pytest.log 2>&1
grep -E "^E  |FAILED|ERROR|passed|failed" $O/pytest.log | cut -c1-500 | head -30
