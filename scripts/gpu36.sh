mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
for v in fold nofold fold nofold; do
  if [ $v = nofold ]; then export MN_NO_BNH_FOLD=1; else unset MN_NO_BNH_FOLD; fi
  echo "=== bench $v ==="; timeout 900 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench36_$v.json; cut -c1-200 gpurun_out/bench36_$v.json | cut -c60-200; python -c "
import json; d=json.load(open('gpurun_out/bench36_$v.json'))
for k,v in d['kernels'].items(): print('%-34s %8.1f us/step %5.1f x %7.1f us  %7.1f GB/s' % (k, v['ms_per_step']*1e3, v['launches_per_step'], v['avg_us'], v['GBps']))" | grep "wgrad<4\|pwd\|bnh_apply"
done
