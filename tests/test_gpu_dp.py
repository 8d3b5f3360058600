"""The data-parallel path with the REAL quantised modules: 2 ranks (gloo; both on the one GPU of the test box -- RCCL needs one GPU per rank, the
collective layer is the same torch.distributed API) vs one process on the concatenated batch.
  * IAO net without BatchNorm: with dp.sync_observers every activation-quantizer range / scale equals the single-process full-batch one BIT FOR BIT
    (SURVEY 8e ii), and the all-reduced gradients equal the full-batch gradients to float round-off;
  * wbwtab nin_gc (packed sign activations, lazy gradients, fused blocks): two steps run, parameters stay bit-identical across the ranks;
  * the graph-replayed IAO step (forward captured in segments between the range collectives) vs the eager data-parallel step."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import numpy as np
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
    elif what == "replica":
        # the reference's DataParallel buffer semantics (dp.ReplicaBuffers): no range collective in forward -- the graph-replayed step is ONE graph A, the ranks use
        # their own shard's ranges within a step, rank 0's observer / BatchNorm state is what every rank holds after it
        from micronet_amd.train import GraphedTrainStep
        from micronet.compression.quantization.wqaq.iao import quantize as Q
        torch.manual_seed(7)
        m = Q.prepare(build_model("resnet18"), inplace=True, a_bits=4, w_bits=4, q_type=0, q_level=0).cuda().train()
        dp.broadcast_parameters(m)
        rb = dp.replica_buffers(m)
        o = make_optimizer(m, 0.01, 1e-5)
        g = GraphedTrainStep(m, o, xs, ys, warmup=2)
        for _ in range(3):
            loss = g.step()[0]
        torch.cuda.synchronize()
        g.finish()
        sd = {k: v.detach().float().cpu() for k, v in m.state_dict().items()}
        gathered = [None, None]
        dist.all_gather_object(gathered, sd)
        if rank == 0:
            bad = [k for k in gathered[0] if not torch.equal(gathered[0][k], gathered[1][k]) and not (torch.isnan(gathered[0][k]).all() and torch.isnan(gathered[1][k]).all())]
            q.put(dict(segments=len(g.segments), synced=sum(int(getattr(mm, "_mn_sync", False)) for mm in m.modules()), bad=bad, loss=float(loss),
                       nbuf=len(rb.bufs), flat=int(rb.flat.numel())))
    elif what.startswith("segmented"):
        # the graph-replayed data-parallel step of an IAO model (range collectives inside forward -> captured in segments) next to the eager DP step: learning rate 0
        # (parameters frozen, observers and BN statistics still move), so the two runs stay comparable step by step
        from micronet_amd.train import GraphedTrainStep
        from micronet.compression.quantization.wqaq.iao import quantize as Q

        def fresh():
            if what == "segmented_resnet18":
                torch.manual_seed(7)
                net, kw = build_model("resnet18"), dict(a_bits=4, w_bits=4, q_type=0, q_level=0)
            else:
                net, kw = _iao_net(), dict(a_bits=8, w_bits=8, q_type=1, q_level=0)
            m = Q.prepare(net, inplace=True, **kw).cuda().train()
            dp.broadcast_parameters(m)
            assert dp.sync_observers(m) > 0
            return m, make_optimizer(m, 0.0, 0.0)
        m1, o1 = fresh()
        sync = dp.GradSync(m1)
        for _ in range(5):
            dp.train_step_dp(m1, o1, sync, xs, ys)
        m2, o2 = fresh()
        g = GraphedTrainStep(m2, o2, xs, ys, warmup=3)           # 3 eager steps (collectives included), then replays
        for _ in range(2):
            loss, _ = g.step()
        g.finish()
        torch.cuda.synchronize()
        sd1, sd2 = m1.state_dict(), m2.state_dict()
        ranges = [k for k in sd1 if k.endswith(("min_val", "max_val", "scale", "zero_point"))]
        bad = [k for k in ranges if not torch.equal(sd1[k], sd2[k])]
        gerr = max(float((p1.grad - p2.grad).abs().max()) / max(float(p1.grad.abs().max()), 1e-12) for p1, p2 in zip(m1.parameters(), m2.parameters()))
        for grp in o2.param_groups:                               # and two real steps: the ranks stay bit-identical
            grp["lr"] = 0.01
        for _ in range(2):
            loss, _ = g.step()
        flat = torch.cat([p.detach().reshape(-1) for p in m2.parameters()])
        other = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(other, flat)
        if rank == 0:
            q.put(dict(segments=len(g.segments), ranges=len(ranges), bad=bad, gerr=gerr, loss=float(loss), in_sync=bool(torch.equal(other[0], other[1])),
                       moved=float((flat - torch.cat([p.detach().reshape(-1) for p in m1.parameters()])).abs().max())))
    elif what == "buckets":
        # the graphed data-parallel step of IAO resnet18 (44.7 MB of gradients) with ONE gradient bucket and with TWO (graph A1 | late bucket's all-reduce overlapping
        # graph A2 | early bucket | graph B): same kernels on the same data, the same element-wise sums over the ranks -> the same parameters
        from micronet_amd.train import GraphedTrainStep
        from micronet.compression.quantization.wqaq.iao import quantize as Q

        def run(nb):
            os.environ["MN_DP_BUCKETS"] = nb
            torch.manual_seed(7)
            m = Q.prepare(build_model("resnet18"), inplace=True, a_bits=4, w_bits=4, q_type=0, q_level=0).cuda().train()
            dp.broadcast_parameters(m)
            assert dp.sync_observers(m) > 0
            o = make_optimizer(m, 0.01, 0.0)
            g = GraphedTrainStep(m, o, xs, ys, warmup=2)
            for _ in range(3):
                loss, _ = g.step()
            g.finish()
            torch.cuda.synchronize()
            return m, g, float(loss)
        m1, g1, l1 = run("1")
        m2, g2, l2 = run("2")
        os.environ.pop("MN_DP_BUCKETS", None)
        flat = torch.cat([p.detach().reshape(-1) for p in m2.parameters()])
        other = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(other, flat)
        f1 = torch.cat([p.detach().reshape(-1) for p in m1.parameters()])
        if rank == 0:
            q.put(dict(two=(g1.graph_a2 is not None, g2.graph_a2 is not None), reason=g2.one_bucket_reason, buckets=(g2.flat.numel(), g2.flat2.numel() if g2.flat2 is not None else 0),
                       segments=(len(g1.segments), len(g2.segments)), losses=(l1, l2), same=bool(torch.equal(f1, flat)), perr=float((f1 - flat).abs().max()),
                       in_sync=bool(torch.equal(other[0], other[1]))))
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


@pytest.mark.parametrize("what", ["segmented", "segmented_resnet18"])
def test_segmented_graph_step_equals_the_eager_dp_step(what):
    """IAO data parallel, graph-replayed: forward is captured in segments cut at the observers' range collectives (train.GraphedTrainStep).  With the learning rate
    at 0 every observer range / scale / zero point after 3 eager + 2 replayed steps is BIT-identical to five eager DP steps, the reduced gradients agree to float
    round-off (atomic accumulation order), and with a real learning rate the two ranks' parameters stay bit-identical."""
    r = _run(what)
    print(what, r)
    assert r["segments"] >= 4 and r["ranges"] >= 8
    assert r["bad"] == [], r["bad"][:5]
    assert r["gerr"] <= 2e-5, r["gerr"]
    assert r["in_sync"] and r["loss"] == r["loss"] and 0 < r["moved"] <= 0.021


def test_two_gradient_buckets_equal_one():
    """resnet18 (IAO): graph A1 -> all-reduce of the late bucket (conv4_x, conv5_x, fc: 40 MB) overlapping graph A2 -> all-reduce of the early bucket (2.6 MB) -> graph B
    leaves the parameters of a three-step run where the one-bucket step leaves them (gloo: element-wise sums in rank order), on both ranks."""
    r = _run("buckets")
    print(r)
    assert r["two"] == (False, True), r
    assert r["buckets"][0] > 8 * r["buckets"][1] > 0, r["buckets"]
    assert r["segments"][0] == r["segments"][1] >= 4
    assert r["in_sync"] and r["losses"][0] == r["losses"][0]
    assert r["same"] or r["perr"] <= 1e-6, r


def _single_rank_worker(port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MN_DP_SINGLE="1")
    from micronet_amd import dp
    from micronet_amd.train import GraphedTrainStep, make_optimizer, synth_batch
    from micronet.compression.quantization.wqaq.iao import quantize as Q
    torch.cuda.set_device(0)
    x, y = synth_batch(16, device="cuda")

    def fresh():
        m = Q.prepare(_iao_net(), inplace=True, a_bits=8, w_bits=8, q_type=1, q_level=0).cuda().train()
        return m, make_optimizer(m, 0.01, 0.0)
    m1, o1 = fresh()
    g1 = GraphedTrainStep(m1, o1, x, y, warmup=2)            # no process group yet: the ordinary single-GPU step, one graph
    l1 = [float(g1.step()[0]) for _ in range(3)]
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    assert dp.active()
    m2, o2 = fresh()
    assert dp.sync_observers(m2) > 0
    g2 = GraphedTrainStep(m2, o2, x, y, warmup=2)            # the data-parallel step on one rank: segments + range collectives + gradient all-reduce + graph B
    l2 = [float(g2.step()[0]) for _ in range(3)]
    torch.cuda.synchronize()
    sd1, sd2 = m1.state_dict(), m2.state_dict()
    keys = [k for k in sd1 if k.endswith(("min_val", "max_val", "scale", "zero_point"))]
    q.put(dict(segments=(len(g1.segments), len(g2.segments)), dp=(g1.dp, g2.dp), losses=(l1, l2), bad=[k for k in keys if not torch.equal(sd1[k], sd2[k])],
               perr=max(float((a - b).abs().max()) for a, b in zip(m1.parameters(), m2.parameters()))))
    dist.destroy_process_group()


def test_single_rank_dp_step_equals_the_plain_step():
    """MN_DP_SINGLE=1 (bench.py's `dp_single_rank` leg): the whole data-parallel step on an RCCL group of ONE rank.  A global range over one rank is the local range
    and a mean over one rank the gradient itself, so it must reproduce the ordinary single-GPU step: ranges bit for bit, parameters to Adam-amplified round-off."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_single_rank_worker, args=(35500 + os.getpid() % 2000, q))
    p.start()
    r = q.get(timeout=300)
    p.join(timeout=300)
    assert p.exitcode == 0
    print(r)
    assert r["segments"][0] == 0 and r["segments"][1] >= 4 and r["dp"] == (False, True)
    assert r["bad"] == [], r["bad"][:5]
    assert r["perr"] <= 0.05 and all(abs(a - b) <= 2e-2 * max(1.0, abs(a)) for a, b in zip(*r["losses"])), r


def test_replica_buffer_semantics_need_no_collective_in_forward():
    """dp.replica_buffers (what nn.DataParallel does in wqaq/iao/main.py:496-500: every replica updates its buffers from its own shard, device 0's survive): the
    graph-replayed resnet18 IAO step is one graph A (no segments, no synced observer), and after every step both ranks hold the same parameters AND the same buffers --
    rank 0's observer ranges / scales / zero points and BatchNorm running statistics, handed over by one flat broadcast next to the gradient all-reduce."""
    r = _run("replica")
    assert r["segments"] == 0 and r["synced"] == 0, r
    assert r["bad"] == [], r["bad"][:8]
    assert r["nbuf"] > 50 and np.isfinite(r["loss"])
