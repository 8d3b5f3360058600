export PYTHONDONTWRITEBYTECODE=1
echo "=== 2 ranks on one GPU, gloo, graphs ==="
MN_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 3 --batch 64 2>&1 | grep -v Warning | tail -1 | cut -c1-330
echo "=== 1 gpu eager vs graph ==="
timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-graph 2>&1 | tail -1 | cut -c1-200
timeout 600 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>&1 | tail -1 | cut -c1-200
