"""The data-parallel path with the REAL quantised modules: 2 ranks (gloo; both on the one GPU of the test box -- RCCL needs one GPU per rank, the
collective layer is the same torch.distributed API) vs one process on the concatenated batch.
  * IAO net without BatchNorm: with dp.sync_observers every activation-quantizer range / scale equals the single-process full-batch one BIT FOR BIT
    (SURVEY 8e ii), and the all-reduced gradients equal the full-batch gradients to float round-off;
  * wbwtab nin_gc (packed sign activations, lazy gradients, fused blocks): two steps run, parameters stay bit-identical across the ranks."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _iao_net():
    torch.manual_seed(4)
    return nn.Sequential(nn.Conv2d(3, 16, 3, padding=1), nn.ReLU(), nn.Conv2d(16, 16, 3, padding=1, groups=2), nn.ReLU(), nn.MaxPool2d(2),
                         nn.Conv2d(16, 32, 1), nn.ReLU(), nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(32, 10))


def _worker(rank, world, port, q, what):
    sys.path.insert(0, ROOT)
    import torch.nn.functional as F
    from micronet_amd import dp
    from micronet_amd.train import build_model, make_optimizer, synth_batch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    x, y = synth_batch(16, device="cuda")
    xs, ys = x[rank * 8:(rank + 1) * 8], y[rank * 8:(rank + 1) * 8]
    if what == "iao":
        from micronet.compression.quantization.wqaq.iao import quantize as Q
        model = Q.prepare(_iao_net(), inplace=True, a_bits=8, w_bits=8, q_type=1, q_level=0).cuda().train()
        dp.broadcast_parameters(model)
        assert dp.sync_observers(model) > 0
        sync = dp.GradSync(model)
        for _ in range(2):                     # first call + one moving-average update
            loss = F.cross_entropy(model(xs), ys)
            model.zero_grad()
            loss.backward()
            sync.wait()
        if rank == 0:
            q.put(({k: v.cpu().numpy().copy() for k, v in model.state_dict().items() if "activation_quantizer" in k},
                   [p.grad.cpu().numpy().copy() for p in model.parameters()]))
    else:
        from micronet.compression.quantization.wbwtab import quantize as Q
        model = Q.prepare(build_model("nin_gc"), inplace=True, A=2, W=3).cuda().train()
        dp.broadcast_parameters(model)
        sync = dp.GradSync(model)
        opt = make_optimizer(model, 0.01, 0.0)
        for _ in range(2):
            dp.train_step_dp(model, opt, sync, xs, ys)
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        other = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(other, flat)
        if rank == 0:
            q.put(bool(torch.equal(other[0], other[1])) and bool(torch.isfinite(flat).all()))
    dist.barrier()
    dist.destroy_process_group()


def _run(what):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, what)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    return got


def test_iao_observer_ranges_and_gradients_match_the_global_batch():
    import torch.nn.functional as F
    from micronet_amd.train import synth_batch
    from micronet.compression.quantization.wqaq.iao import quantize as Q
    bufs, grads = _run("iao")
    x, y = synth_batch(16, device="cuda")
    model = Q.prepare(_iao_net(), inplace=True, a_bits=8, w_bits=8, q_type=1, q_level=0).cuda().train()
    for _ in range(2):
        loss = F.cross_entropy(model(x), y)
        model.zero_grad()
        loss.backward()
    sd = model.state_dict()
    assert len(bufs) >= 8
    for k, v in bufs.items():
        assert (sd[k].cpu().numpy() == v).all(), k                      # ranges, scales, zero points: bit-identical to the full batch
    for p, g in zip(model.parameters(), grads):
        ref = p.grad.cpu()
        assert float((torch.from_numpy(g) - ref).abs().max()) <= 1e-5 * float(ref.abs().max().clamp_min(1e-12)) + 1e-9


def test_wbwtab_fused_net_stays_in_sync_across_ranks():
    assert _run("wbwtab") is True
