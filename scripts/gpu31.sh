mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
for w in c1 c1_w2a2 c3 c4 c5; do echo "=== $w ==="; timeout 600 python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step graph=%s' % d['config'].get('hip_graph'), d['config'].get('hip_graph_error', ''), 'loss', d['config']['final_loss'])
except Exception as e: print('ERR', e)
"; done
