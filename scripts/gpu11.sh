mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench11.json; cut -c1-2500 gpurun_out/bench11.json
bash scripts/pmc_traffic.sh c2 2>&1 | tail -15
