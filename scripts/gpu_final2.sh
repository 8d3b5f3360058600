#!/bin/bash
# End-of-round measurement on a GPU box WITHOUT the test suite (run separately): smoke, the default bench line (PMC + CPU baseline), rocprofv3 summaries.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/final; mkdir -p $O
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --detail $O/bench_detail.json > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 2600 $O/bench.json; echo; wc -c $O/bench.json
for w in ${MN_WORKLOADS:-c2 c1 c5}; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o p -- python $OLDPWD/bench.py --only $w --steps 20 --warmup 5 --repeats 1 --no-pmc --no-cpu-baseline --no-kernel-timing > $OLDPWD/$O/prof_$w.log 2>&1)
  f=$(find /tmp/prof_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $O/${w}_kernel_stats.csv && python scripts/prof_summary.py $f $O/${w}_summary.md "rocprofv3 --kernel-trace --stats: bench.py --only $w --steps 20 --warmup 5 --repeats 1 --no-pmc --no-cpu-baseline --no-kernel-timing" && head -8 $O/${w}_summary.md
done
