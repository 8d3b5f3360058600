#!/bin/bash
# round 4, GPU call A: the new BN-fused IAO block (kernels, module chain, c3 parity) + c3 A/B + profile
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r4a; mkdir -p $O
python -m pytest tests/test_gpu_kernels.py -k iaobf -x -q > $O/t_kernels.log 2>&1; echo "kernels rc=$?"
python -m pytest tests/test_gpu_bnfuse_block.py -x -q > $O/t_block.log 2>&1; echo "block rc=$?"
python -m pytest tests/test_gpu_modules.py tests/test_gpu_iao_ops.py -x -q > $O/t_modules.log 2>&1; echo "modules rc=$?"
python -m pytest tests/test_gpu_parity_full.py -k c3 -x -q > $O/t_parity_c3.log 2>&1; echo "parity c3 rc=$?"
python -m pytest tests/test_gpu_models.py tests/test_gpu_determinism.py tests/test_gpu_inference.py -x -q > $O/t_models.log 2>&1; echo "models rc=$?"
python bench.py --only c3 --no-pmc --no-cpu-baseline --detail $O/c3_new_detail.json > $O/c3_new.json 2> $O/c3_new.err; echo "c3 new rc=$?"; tail -c 600 $O/c3_new.json
MN_NO_BNFUSE_BLOCK=1 python bench.py --only c3 --no-pmc --no-cpu-baseline --detail $O/c3_old_detail.json > $O/c3_old.json 2> $O/c3_old.err; echo "c3 old rc=$?"; tail -c 300 $O/c3_old.json
(cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_c3 -o c3 -- python $GRAFT_REPO_ROOT/bench.py --only c3 --steps 20 --warmup 5 --repeats 1 --no-pmc --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1)
cp $(find /tmp/prof_c3 -name "*kernel_stats.csv" | head -1) $O/c3_kernel_stats.csv 2>/dev/null
head -45 $O/c3_kernel_stats.csv | cut -c1-150
