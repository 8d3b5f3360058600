#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/mid3; mkdir -p $O
timeout 500 python -m pytest tests/test_gpu_parity_resnet.py tests/test_gpu_models.py tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_determinism.py -m gpu -q --tb=short -p no:cacheprovider -k "resnet_iao or c5 or iao_resnet or qdense_layer_iao or fused_blocks_match_unfused or (teacher_forced_resnet and c4) or c4_resnet or first_conv_qa or qconv_bnq_block or graphed" > $O/pytest.log 2>&1
grep -E "^E  |FAILED|ERROR|passed|failed" $O/pytest.log | cut -c1-500 | head -30
