export PYTHONDONTWRITEBYTECODE=1
MN_WG2_CW8=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -p no:cacheprovider -k "wgrad_direct or folded_in" 2>&1 | tail -2
echo "== 2x2 waves (64x64)"; python scripts/kbench.py --layers L2,L5,L8 --algos 3 --scheme sign8 2>&1 | grep -v amdgpu.ids | grep wgrad
echo "== 4x1 waves (32x128)"; MN_WG2_CW8=1 python scripts/kbench.py --layers L2,L5,L8 --algos 3 --scheme sign8 2>&1 | grep -v amdgpu.ids | grep wgrad
for v in 0 1; do echo "=== bench CW8=$v ==="; MN_WG2_CW8=$v timeout 900 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > /tmp/b.json; python -c "
import json; d=json.load(open('/tmp/b.json')); print(d['value'], d['ms_per_step'])
for k,v in d['kernels'].items():
    if 'wgrad<' in k and 'pws' in k: print('%-34s %8.1f us/step %5.1f x %7.1f us  %7.1f GB/s' % (k, v['ms_per_step']*1e3, v['launches_per_step'], v['avg_us'], v['GBps']))"; done
