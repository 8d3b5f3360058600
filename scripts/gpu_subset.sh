#!/bin/bash
# A SUBSET of the GPU suite in one pytest process (two xdist workers), for mid-round checks when GPU minutes are scarce:
#   bash scripts/gpu_subset.sh "<pytest -k expression>" [test files ...]
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/subset; mkdir -p $O
k=$1; shift
files=${@:-tests}
timeout 600 python -m pytest $files -m gpu -q --tb=short -p no:cacheprovider -n 2 -k "$k" > $O/pytest.log 2>&1
grep -E "^E  |FAILED|ERROR|passed|failed" $O/pytest.log | cut -c1-500 | head -30
