#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in 1024 512 256 2048; do
  echo "=== MN_PWS_CAP=$v ==="
  MN_PWS_CAP=$v timeout 200 python scripts/kbench_fused.py 2>&1 | grep -E "fused fwd"
done
