#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/mid3; mkdir -p $O
timeout 500 python -m pytest tests/test_gpu_parity_full.py tests/test_gpu_models.py tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_determinism.py tests/test_gpu_inference.py -m gpu -q --tb=short -p no:cacheprovider -k "wbwtab or c2 or bnsign or bn_folded or maxpool_folded or byte_stash or sign" > $O/pytest.log 2>&1
grep -E "^E  |FAILED|ERROR|passed|failed" $O/pytest.log | cut -c1-500 | head -30
