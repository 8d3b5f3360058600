#!/bin/bash
# ablation of k_pws_wgrad: where does the time go?
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for d in 0 1 2 3 4 5 7 0; do
  if [ $d = 0 ]; then unset MN_LIB_PATH; else export MN_LIB_PATH=$PWD/micronet_amd/lib/libmicronet_hip_dbg$d.so; fi
  echo "=== dbg $d ==="
  timeout 120 python scripts/kbench.py --scheme sign8 --layers L2,L5,L8 --algos 3 --which wgrad --iters 30 2>&1 | grep -v Warn
done
