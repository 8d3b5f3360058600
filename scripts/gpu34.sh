mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_modules.py -m gpu -q --tb=short -p no:cacheprovider -k "first" 2>&1 | grep -v Warning | tail -8
echo "=== bench lazy bn grad ==="; timeout 900 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench34.json; cut -c1-240 gpurun_out/bench34.json; python -c "
import json; d=json.load(open('gpurun_out/bench34.json'))
for k,v in d['kernels'].items(): print('%-34s %8.1f us/step %5.1f x %7.1f us  %7.1f GB/s' % (k, v['ms_per_step']*1e3, v['launches_per_step'], v['avg_us'], v['GBps']))" | grep "c1_\|bns"
