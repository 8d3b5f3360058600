mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof9 -o c2 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/bench9_prof.log 2>&1
cd $R
tail -1 gpurun_out/bench9_prof.log | cut -c1-400
python scripts/step_breakdown.py gpurun_out/prof9/c2_kernel_trace.csv 45
