mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
R=$GRAFT_REPO_ROOT
echo "=== pytest gpu (new) ==="; timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -m gpu -q --tb=short -p no:cacheprovider -k "sign8 or packed or bnsign or fused_bn" -x 2>&1 | tail -15
echo "=== pytest gpu (all) ==="; timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu15.log 2>&1; echo "pytest rc=$?"; tail -6 gpurun_out/pytest_gpu15.log
echo "=== bench ==="
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench15.json; cut -c1-1800 gpurun_out/bench15.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof15 -o c2 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/bench15_prof.log 2>&1
cd $R
python scripts/step_breakdown.py $(find gpurun_out/prof15 -name '*kernel_trace.csv' | head -1) 32
find gpurun_out/prof15 -name '*kernel_trace.csv' -size +20M -delete
