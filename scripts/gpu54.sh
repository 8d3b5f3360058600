#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "sign or bnh" 2>&1 | tail -2
for v in old new; do
  echo "=== $v ==="
  if [ $v = old ]; then export MN_LIB_PATH=$PWD/micronet_amd/lib/libmicronet_hip_old.so; else unset MN_LIB_PATH; fi
  timeout 200 python scripts/kbench_fused.py 2>&1 | grep -E "fused fwd|plain"
done
for v in old new old new; do
  if [ $v = old ]; then export MN_LIB_PATH=$PWD/micronet_amd/lib/libmicronet_hip_old.so; else unset MN_LIB_PATH; fi
  timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench54_$v.json 2>/dev/null; echo $v; cut -c1-200 gpurun_out/bench54_$v.json
done
