#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "=== pytest gpu (modules, kernels kxk) ==="
timeout 900 python -m pytest tests/test_gpu_modules.py tests/test_gpu_kernels.py -x -q -m gpu -k "shuffle or kxk or fused or bnsign" 2>&1 | tail -4
for v in 1 0 1 0; do
  echo "=== MN_NO_KXK_STASH=$v ==="
  if [ $v = 1 ]; then export MN_NO_KXK_STASH=1; else unset MN_NO_KXK_STASH; fi
  timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench44_$v.json 2>/dev/null; cut -c1-200 gpurun_out/bench44_$v.json
done
