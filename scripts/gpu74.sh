#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof74 -o c1 -- python $R/bench.py --workload c1 --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/bench74_prof.log 2>&1
cd $R
find gpurun_out/prof74 -name '*kernel_trace.csv' -size +20M -delete
tail -1 gpurun_out/bench74_prof.log | cut -c1-160
