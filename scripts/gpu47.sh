#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "=== pytest gpu kernels (sign) ==="
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "sign or bnh" 2>&1 | tail -2
for v in old new; do
  echo "=== $v ==="
  if [ $v = old ]; then export MN_LIB_PATH=$PWD/micronet_amd/lib/libmicronet_hip_old.so; else unset MN_LIB_PATH; fi
  timeout 200 python scripts/kbench_fused.py 2>&1 | grep -v -i warn | head -30
done
for v in old new old new; do
  echo "=== bench $v ==="
  if [ $v = old ]; then export MN_LIB_PATH=$PWD/micronet_amd/lib/libmicronet_hip_old.so; else unset MN_LIB_PATH; fi
  timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench47_$v.json 2>/dev/null; cut -c1-200 gpurun_out/bench47_$v.json
done
