#!/bin/bash
# The one parameterised GPU-box runner (run through scripts/grun.sh or gpurun directly, from the repo root):
#   bash scripts/gpu_check.sh [tests|parity|bench|prof]...      (default: tests bench)
#   tests  : pytest -m gpu (whole suite)            -> gpurun_out/pytest_gpu.log
#   parity : only tests/test_gpu_parity_full.py      -> gpurun_out/parity_r06/
#   bench  : smoke + bench.py (default workloads)    -> gpurun_out/bench.json
#   prof   : rocprofv3 --kernel-trace --stats of bench.py per workload -> gpurun_out/prof_<w>/
#   (PMC traffic / MFMA-busy: bench.py collects them itself -- roofline.traffic, gpurun_out/bench_detail.json)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
WL=${MN_WORKLOADS:-"c2 c1_w2a2"}
[ $# -eq 0 ] && set -- tests bench
for what in "$@"; do
  case $what in
    tests)
      timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1
      grep -E "^E  |FAILED|ERROR|passed|failed" gpurun_out/pytest_gpu.log | cut -c1-400 | head -12 ;;
    parity)
      timeout 900 python -m pytest tests/test_gpu_parity_full.py -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/pytest_parity.log 2>&1
      grep -E "^E  |FAILED|ERROR|passed|failed|worst" gpurun_out/pytest_parity.log | cut -c1-1500 | head -20 ;;
    bench)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
      timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -c 1500 gpurun_out/bench.json ;;
    prof)
      for w in $WL; do
        (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_$w -o p -- python $OLDPWD/bench.py --only $w --steps 20 --warmup 5 --repeats 1 --no-pmc --no-cpu-baseline --no-kernel-timing --detail /tmp/prof_detail_$w.json > $OLDPWD/gpurun_out/prof_$w.log 2>&1)
        f=$(find gpurun_out/prof_$w -name "*kernel_stats.csv" | head -1)
        [ -n "$f" ] && python scripts/prof_summary.py $f gpurun_out/prof_${w}_summary.md "rocprofv3 --kernel-trace --stats: bench.py --only $w --steps 20 --warmup 5 --repeats 1 --no-pmc --no-cpu-baseline --no-kernel-timing" && head -25 gpurun_out/prof_${w}_summary.md
      done ;;
  esac
done
