"""N>1 path on CPU: world_size-2 gloo. The DP layer is model-agnostic, so it is exercised with a plain fp32 net
(the quantised kernels need the GPU): 2 ranks x half batch must equal 1 process x full batch."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _net():
    torch.manual_seed(3)
    return nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(), nn.Conv2d(8, 8, 3, padding=1), nn.ReLU(),
                         nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(8, 10))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from micronet_amd import dp
    from micronet_amd.train import make_optimizer, synth_batch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    model = _net()
    if rank == 1:                       # deliberately different start: broadcast must fix it
        for p in model.parameters():
            p.data.add_(1.0)
    dp.broadcast_parameters(model)
    sync = dp.GradSync(model, bucket_bytes=1024)     # tiny buckets -> several collectives
    opt = make_optimizer(model, 0.01, 1e-5)
    x, y = synth_batch(8)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    for _ in range(3):
        dp.train_step_dp(model, opt, sync, xs, ys)
    if rank == 0:
        q.put([p.detach().numpy().copy() for p in model.parameters()])   # by value: the worker exits before the parent reads
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_process_full_batch():
    sys.path.insert(0, ROOT)
    from micronet_amd.train import make_optimizer, synth_batch, train_step
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    torch.set_num_threads(1)
    model = _net()
    opt = make_optimizer(model, 0.01, 1e-5)
    x, y = synth_batch(8)
    for _ in range(3):
        train_step(model, opt, x, y)
    for a, b in zip(got, model.parameters()):
        a = torch.from_numpy(a)
        assert torch.allclose(a, b.detach(), rtol=1e-4, atol=1e-6), (a - b).abs().max()


def test_single_process_is_a_noop():
    sys.path.insert(0, ROOT)
    from micronet_amd import dp
    m = _net()
    s = dp.GradSync(m)
    s.wait()
    assert s.world == 1 and len(s.buckets) == 1
