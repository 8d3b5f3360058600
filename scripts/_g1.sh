cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out/fg
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -q --tb=short -p no:cacheprovider -k "first_conv or first_block" > gpurun_out/fg/k.log 2>&1; tail -25 gpurun_out/fg/k.log | cut -c1-300
for w in c2 c2b; do for v in 1 0; do
MN_FIRST_FUSED=$v timeout 300 python bench.py --only $w --steps 30 --warmup 5 --repeats 3 --no-pmc --no-cpu-baseline --no-kernel-timing --no-dp-single --detail /tmp/d.json 2>gpurun_out/fg/b_${w}.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$w fused=$v', d['value'], d['ms_per_step'], d['config'].get('final_loss'))"
done; done
MN_WORKLOADS='c2' bash scripts/gpu_check.sh prof > /dev/null 2>&1; grep -E 'k_c1' gpurun_out/prof_c2_summary.md | cut -c1-200
