#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "=== long run ==="
timeout 300 python bench.py --no-cpu-baseline --steps 300 --warmup 20 2>/dev/null | tail -1 | cut -c60-230
echo "=== default run (as the driver calls it) ==="
timeout 600 python bench.py 2>/dev/null | tail -1 | cut -c1-330
echo "=== eager (no graph) ==="
timeout 300 python bench.py --no-cpu-baseline --no-graph --steps 20 2>/dev/null | tail -1 | cut -c60-230
