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


# ------------------------------------------------------------------------------------------------ observer ranges over the global batch
def _obs_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from micronet_amd import dp
    from oracle import torch_oracle as TO
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(5)
    xs = [torch.randn(8, 4, 6, 6) * (1 + s) + 0.3 * s for s in range(3)]          # 3 steps of a global batch of 8
    obs = TO.Observer("L", ema=True)
    for x in xs:
        shard = x[rank * 4:(rank + 1) * 4]
        lo, hi = shard.min().reshape(1), shard.max().reshape(1)
        dp.allreduce_minmax(lo, hi)                 # what the product's observer does between its two kernel launches
        obs(torch.cat([lo, hi]))                    # the ordinary (first-call / EMA) update on the two global extremes
    if rank == 0:
        q.put((obs.min_val.numpy().copy(), obs.max_val.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_observer_ranges_reduce_to_the_global_batch():
    """2 ranks x half batch: all-reduced (min, max) followed by the EMA update == the single-process observer on the full batch, bit for bit."""
    sys.path.insert(0, ROOT)
    from oracle import torch_oracle as TO
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_obs_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    lo, hi = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    torch.manual_seed(5)
    xs = [torch.randn(8, 4, 6, 6) * (1 + s) + 0.3 * s for s in range(3)]
    ref = TO.Observer("L", ema=True)
    for x in xs:
        ref(x)
    assert (lo == ref.min_val.numpy()).all() and (hi == ref.max_val.numpy()).all()


def test_gradient_bucket_layout_resnet18():
    """resnet18's 44.7 MB of fp32 gradients split into two buckets at the 32 MB cap, filled in reverse parameter order (the order backward
    produces them), every parameter exactly once; nin_gc (2.37 MB) is a single bucket."""
    sys.path.insert(0, ROOT)
    from micronet_amd import dp
    from micronet_amd.train import build_model
    m = build_model("resnet18")
    s = dp.GradSync(m)
    total = sum(p.numel() for p in m.parameters())
    assert total == 11173962 and len(s.buckets) == 2
    assert sum(b.flat.numel() for b in s.buckets) == total
    assert s.buckets[0].flat.numel() * 4 >= 32 << 20 and s.buckets[0].params[0] is list(m.parameters())[-1]
    seen = [id(p) for b in s.buckets for p in b.params]
    assert len(seen) == len(set(seen)) == len(list(m.parameters()))
    n = build_model("nin_gc")
    assert len(dp.GradSync(n).buckets) == 1 and sum(p.numel() for p in n.parameters()) == 591390


def test_sync_observers_marks_only_level_L_minmax_observers():
    sys.path.insert(0, ROOT)
    from micronet_amd import dp
    from micronet_amd.train import build_model
    from micronet.compression.quantization.wqaq.iao import quantize as Q
    from micronet.compression.quantization.wqaq.dorefa import quantize as D
    m = Q.prepare(build_model("resnet18"), inplace=True, a_bits=4, w_bits=4, q_type=0, q_level=0)
    n = dp.sync_observers(m)
    acts = sum(1 for mod in m.modules() if isinstance(mod, (Q.QuantConv2d, Q.QuantLinear)))
    adds = sum(1 for mod in m.modules() if isinstance(mod, Q.QuantAdd))
    assert adds == 8 and n >= acts + 3 * adds            # one per activation quantizer + (res, shortcut, union) per QuantAdd; weight observers are per-channel
    assert not any(getattr(o, "_mn_sync", False) for mod in m.modules() if isinstance(mod, (Q.QuantConv2d,)) for o in [mod.weight_quantizer.observer])
    assert dp.sync_observers(D.prepare(build_model("nin_gc"), inplace=True, a_bits=2, w_bits=2)) == 0


# ------------------------------------------------------------------------------------------------ graphed step vs cross-rank observer collectives
class _Synced(nn.Module):
    """stands for an IAO activation observer switched on by dp.sync_observers (a collective inside forward)"""
    _mn_sync = True

    def forward(self, x):
        return x


def _graph_fallback_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from micronet_amd import dp
    from micronet_amd.train import GraphedTrainStep, make_optimizer, synth_batch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    model = nn.Sequential(_net(), _Synced())
    opt = make_optimizer(model, 0.01, 1e-5)
    x, y = synth_batch(8)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    msgs = []
    try:                             # no device here: refused before anything is touched -> the caller is told to run the eager step
        GraphedTrainStep(model, opt, xs, ys)
        msgs.append("no error")
    except RuntimeError as e:
        msgs.append(str(e))
    sync = dp.GradSync(model)
    loss, _ = dp.train_step_dp(model, opt, sync, xs, ys)          # ... which works
    if rank == 0:
        q.put((msgs, float(loss)))
    dist.barrier()
    dist.destroy_process_group()


def test_graphed_step_refuses_host_tensors_on_every_rank_alike():
    """GraphedTrainStep without a device: it raises BEFORE touching the process group (all ranks alike, so nobody dead-locks) and names the eager step, which
    bench.py falls back to.  (With a device, models with cross-rank observer collectives inside forward are captured in segments: tests/test_gpu_dp.py.)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_graph_fallback_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    msgs, loss = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert len(msgs) == 1 and "eager DP step" in msgs[0] and "cpu" in msgs[0], msgs
    assert loss == loss


def _bn_net():
    torch.manual_seed(5)
    return nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU(), nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(8, 10))


def _replica_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from micronet_amd import dp
    from micronet_amd.train import make_optimizer, synth_batch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    model = _bn_net()
    dp.broadcast_parameters(model)
    rb = dp.replica_buffers(model)
    sync = dp.GradSync(model)
    opt = make_optimizer(model, 0.01, 0.0)
    x, y = synth_batch(8)
    xs, ys = x[rank * 4:(rank + 1) * 4], y[rank * 4:(rank + 1) * 4]
    # what rank 0's BatchNorm sees in step 1 (its own shard, the common start weights): the running mean every rank must hold after the step
    with torch.no_grad():
        ref = _bn_net()
        y0 = ref[0](x[0:4])
        want = 0.9 * torch.zeros(8) + 0.1 * y0.mean(dim=(0, 2, 3))
    dp.train_step_dp(model, opt, sync, xs, ys)
    got = model[1].running_mean.clone()
    for _ in range(2):
        dp.train_step_dp(model, opt, sync, xs, ys)
    state = [t.detach().clone() for t in list(model.parameters()) + list(model.buffers())]
    gathered = [None, None]
    dist.all_gather_object(gathered, state)
    if rank == 1:
        q.put(dict(step1_err=float((got - want).abs().max()), same=all(torch.equal(a.float(), b.float()) for a, b in zip(*gathered)), nbuf=len(rb.bufs)))
    dist.barrier()
    dist.destroy_process_group()


def test_replica_buffers_follow_rank_zero():
    """dp.ReplicaBuffers (the reference's nn.DataParallel semantics for buffers): after a step every rank holds rank 0's BatchNorm running statistics -- computed from
    rank 0's shard -- and, with the gradient all-reduce, the same parameters."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_replica_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got["same"] and got["nbuf"] == 2 and got["step1_err"] <= 1e-6, got
