"""Whole-net QAT on the MI355X.

(1) ``test_layerwise_teacher_forced``: the CPU oracle runs the whole net (forward + backward) and every quantised
    layer's input / output / incoming gradient is recorded; each product layer is then fed the SAME input and gradient on
    the GPU and must reproduce output, input gradient and parameter gradients to the float-accumulate tolerance.  This is
    the parity statement for real activations, real layer shapes and real weights.
(2) ``test_training_trajectory_smoke_vs_reference``: SMOKE, free-running 3 Adam steps vs the reference's CPU trajectory
    (tests/golden/models.npz).  Quantised nets are chaotic -- with binary activations one sign flip (a BN output within
    1e-7 of zero) cascades through every following layer -- so this comparison is statistical: losses close, gradient
    norms close; logits are compared tightly only for the schemes that are not binary."""
import importlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = {
    "c1_nin_gc_dorefa_w8a8": ("nin_gc", "wqaq.dorefa", dict(a_bits=8, w_bits=8), 8, 1e-5),
    "c2_nin_gc_wbwtab_w3a2": ("nin_gc", "wbwtab", dict(A=2, W=3), 8, 0.0),
    "c2b_nin_gc_wbwtab_w2a2": ("nin_gc", "wbwtab", dict(A=2, W=2), 8, 0.0),
    "c3_nin_gc_iao_w8a8_bnfuse": ("nin_gc", "wqaq.iao", dict(a_bits=8, w_bits=8, q_type=0, q_level=0, weight_observer=0, bn_fuse=True), 8, 1e-5),
    "c4_resnet18_dorefa_w2a2": ("resnet18", "wqaq.dorefa", dict(a_bits=2, w_bits=2), 4, 1e-5),
    "c5_resnet18_iao_w4a4": ("resnet18", "wqaq.iao", dict(a_bits=4, w_bits=4, q_type=0, q_level=0), 4, 1e-5),
    "nin_dorefa_w4a4": ("nin", "wqaq.dorefa", dict(a_bits=4, w_bits=4), 4, 1e-5),
}


@pytest.mark.parametrize("key", list(CFG))
def test_training_trajectory_smoke_vs_reference(golden, key):
    """SMOKE test (free-running, batch 4-8): three optimizer steps stay near the reference's trajectory.  A free-running low-bit net is chaotic -- one activation or
    weight code that lands on the other side of a rounding boundary (legitimately: a different summation order of a float accumulate) moves the logits by ~1e-2 --
    so the bounds here are loose; parity proper is the teacher-forced stage-wise suite (test_layerwise_teacher_forced, test_gpu_parity_full, test_gpu_parity_resnet)."""
    from micronet_amd.train import build_model, make_optimizer, synth_batch, train_step
    arch, scheme, kw, B, wd = CFG[key]
    quantize = importlib.import_module("micronet.compression.quantization.%s.quantize" % scheme)
    model = quantize.prepare(build_model(arch), inplace=True, **kw).cuda()
    opt = make_optimizer(model, 0.01, wd)
    x, y = synth_batch(B, device="cuda")
    model.train()
    ref_losses = golden.meta["surface"][key]["losses"]
    losses = []
    for step in range(3):
        out = model(x)
        loss = torch.nn.functional.cross_entropy(out, y)
        opt.zero_grad()
        loss.backward()
        if step == 0:
            ref = golden.mo[f"{key}_logits0"]
            err = np.max(np.abs(out.detach().cpu().numpy() - ref)) / max(np.max(np.abs(ref)), 1e-6)
            chaotic = "wbwtab" in key or key.startswith(("c4", "c5"))      # 2-bit / binary nets: one code flip (a rounding-level
                                                                            # change of a summation order) moves whole gradients
            print(key, "logits0 rel err", err)
            # 8-bit nets: measured 1.5e-2 (generic kernels) / 2.4e-2 (fused BN-fuse blocks) on the same box for c3 -- both are code flips of a batch-8 net, not
            # arithmetic error (the teacher-forced stages of the same configuration agree to <= 1e-5)
            assert err <= (1.0 if chaotic else 5e-2), ("logits0", err)
            gn_ref = golden.meta["surface"][key]["gradnorm0"]
            gmax = max(gn_ref.values())
            bad = []
            for n_, p in model.named_parameters():
                r = gn_ref[n_]
                if r < 1e-4 * gmax:
                    continue            # conv biases in front of a BatchNorm: the true gradient is 0, both sides hold round-off
                gnorm = float(p.grad.double().norm())
                if abs(gnorm - r) > (0.5 if chaotic else 0.1) * r:
                    bad.append((n_, gnorm, r))
            assert len(bad) <= max(1, len(gn_ref) // 10), bad[:5]
        opt.step()
        losses.append(float(loss))
    print(key, "gpu", losses, "ref", ref_losses)
    # (measured on the ORACLE itself, nin_gc IAO W8A8 at batch 8: forcing ONE weight code of one layer to the neighbouring level moves loss0 by up to 1.2e-3 and
    # the logits by up to 1.8e-2 of their maximum; scaling one block's input by 1 + 2e-6 moves loss0 by 8e-4 -- a free-running comparison cannot be tighter
    # than a couple of such flips)
    assert abs(losses[0] - ref_losses[0]) <= (6e-2 if chaotic else 5e-3)
    # low-bit resnets at lr 0.01 are unstable (c5 collapses to loss 0.13 in 2 steps; c4's third loss swings between 1.7 and
    # 3.3 with the summation order of a single kernel): the free-running comparison covers the first two steps there
    nsteps = 2 if key.startswith(("c4", "c5")) else 3
    assert all(abs(a - b) <= 0.2 * max(1.0, abs(b)) for a, b in zip(losses[:nsteps], ref_losses[:nsteps])), (losses, ref_losses)
    model.eval()
    out = model(x)
    assert torch.isfinite(out).all()


def test_state_dict_round_trip_with_reference_layout(golden):
    """state_dict produced on the GPU loads back and keeps the reference key layout (checkpoint interchange)."""
    from micronet_amd.train import build_model
    from micronet.compression.quantization.wqaq.iao import quantize
    model = quantize.prepare(build_model("nin_gc"), inplace=True, a_bits=8, w_bits=8, bn_fuse=True).cuda().train()
    x = torch.randn(4, 3, 32, 32, device="cuda")
    model(x)
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    assert [[k, list(v.shape)] for k, v in sd.items()] == golden.meta["surface"]["c3_nin_gc_iao_w8a8_bnfuse"]["state"]
    model2 = quantize.prepare(build_model("nin_gc", seed=5), inplace=True, a_bits=8, w_bits=8, bn_fuse=True).cuda()
    model2.load_state_dict(sd)
    model.eval(), model2.eval()
    assert torch.equal(model(x), model2(x))


# ------------------------------------------------------------------------------------------------ teacher forcing
def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("key", list(CFG))
def test_layerwise_teacher_forced(key):
    from oracle import torch_oracle as TO
    from micronet_amd.train import build_model, synth_batch
    arch, scheme, kw, B, wd = CFG[key]
    quantize = importlib.import_module("micronet.compression.quantization.%s.quantize" % scheme)
    prod = quantize.prepare(build_model(arch), inplace=True, **kw).cuda().train()
    orc = TO.prepare(build_model(arch), scheme.split(".")[-1], inplace=True, **kw).train()
    unit_types = (TO.OConv2d, TO.OLinear, TO.OBinAct, TO.OQuantWrap, TO.OQuantAdd)
    rec = {}

    def hook(name):
        def fn(mod, inputs, output):
            r = rec.setdefault(name, {})
            r["in"] = [i.detach().clone() for i in inputs]
            r["out"] = output.detach().clone()
            r["gin"] = [None] * len(inputs)
            for k_, i in enumerate(inputs):
                if i.requires_grad:
                    i.register_hook(lambda g, k_=k_, r=r: r["gin"].__setitem__(k_, g.detach().clone()))
            output.register_hook(lambda g, r=r: r.__setitem__("gout", g.detach().clone()))
        return fn

    units = [(n, m) for n, m in orc.named_modules() if isinstance(m, unit_types)]
    for n, m in units:
        # give every unit a private copy of its inputs so the recorded input gradient is THIS unit's contribution only
        # (a residual block's input also feeds the shortcut)
        m.register_forward_pre_hook(lambda mod, inputs: tuple(i.clone() if i.requires_grad else i for i in inputs))
        m.register_forward_hook(hook(n))
    torch.set_num_threads(min(32, __import__("os").cpu_count()))
    x, y = synth_batch(B)
    loss = torch.nn.functional.cross_entropy(orc(x), y)
    loss.backward()
    pmods = dict(prod.named_modules())
    oparams = {n: dict(m.named_parameters(recurse=False)) for n, m in units}
    report = []
    for n, om in units:
        pm, r = pmods[n], rec[n]
        if getattr(pm, "in_shuffle_groups", 0):
            pm.in_shuffle_groups = 0      # the recorded input is already shuffled (the oracle's block shuffles in front of its conv);
                                          # the folded permutation has its own tests (test_gpu_kernels / test_gpu_modules)
        ins = [i.cuda().requires_grad_(True) for i in r["in"]]
        for p in pm.parameters():
            p.grad = None
        out = pm(*ins)
        bnfuse = type(pm).__name__ == "QuantBNFuseConv2d"
        tol = 1e-5
        e_out = _rel(out, r["out"])
        out.backward(r["gout"].cuda())
        errs = {"y": e_out}
        for k_, g in enumerate(r["gin"]):
            if g is not None:
                errs["dx%d" % k_] = _rel(ins[k_].grad, g)
        pn = dict(pm.named_parameters(recurse=False))
        for pname, op in oparams[n].items():
            if op.grad is None or pname not in pn:
                continue
            if pname == "bias" and bnfuse:
                continue                      # mathematically zero (cancels through the batch mean)
            if pname == "bias" and r["gout"].dim() == 4:
                # d bias = sum of gout; in front of a BatchNorm the true value is 0, so compare on the scale of sum|gout|
                scale = r["gout"].abs().sum(dim=(0, 2, 3)).max()
                errs["dbias"] = float((pn[pname].grad.cpu().double() - op.grad.double()).abs().max() / scale)
                continue
            errs["d" + pname] = _rel(pn[pname].grad, op.grad)
        # wbwtab convs sit behind a BatchNorm: the incoming gradient sums to ~0 per channel, so d weight is a heavily
        # cancelling sum and the REFERENCE's own fp32 summation order shows up at the 1e-5 level.  For these units both
        # sides are measured against an fp64 evaluation of the same oracle module (the quantizer decisions -- ternary
        # threshold, signs -- do not depend on the precision for generic weights): ours must be within the tolerance of
        # the exact value, or at least as close to it as the reference's fp32 CPU result is.
        ref_slack = {}
        if scheme == "wbwtab" and isinstance(om, TO.OConv2d):
            import copy
            om64 = copy.deepcopy(om).double()
            for p_ in om64.parameters():
                p_.grad = None
            ins64 = [i.double().requires_grad_(True) for i in r["in"]]
            om64(*ins64).backward(r["gout"].double())
            p64 = dict(om64.named_parameters(recurse=False))
            if "weight" in pn and p64["weight"].grad is not None:
                g64 = p64["weight"].grad
                sc64 = g64.abs().max().clamp_min(1e-300)
                e_ours = float((pn["weight"].grad.double().cpu() - g64).abs().max() / sc64)
                e_ref = float((oparams[n]["weight"].grad.double() - g64).abs().max() / sc64)
                errs["dweight"] = e_ours
                ref_slack["dweight"] = 2.0 * e_ref
        report.append((n, type(pm).__name__, {k_: float("%.1e" % v) for k_, v in errs.items()}))
        for k_, v in errs.items():
            # DoReFa d weight: the element holding max|tanh w| receives -sum(du*t/2)/M^2, a cancelling fp32 sum whose
            # rounding in the REFERENCE depends on ATen's summation order (ours is accumulated in fp64)
            lim = 1e-4 if (k_ == "dweight" and "dorefa" in scheme) else tol
            lim = max(lim, ref_slack.get(k_, 0.0))
            assert v <= lim, (key, n, type(pm).__name__, k_, v, ref_slack, report[-3:])
    print(key, "worst per-layer rel err:", max(max(e.values()) for _, _, e in report))


@pytest.mark.parametrize("key", ["c1_nin_gc_dorefa_w8a8", "c2_nin_gc_wbwtab_w3a2"])
def test_graphed_train_step_matches_eager(key):
    """GraphedTrainStep (whole step captured in a HIP graph, Adam step count in device memory) trains like the eager loop."""
    from micronet_amd.train import GraphedTrainStep, build_model, make_optimizer, synth_batch, train_step
    arch, scheme, kw, B, wd = CFG[key]
    quantize = importlib.import_module("micronet.compression.quantization.%s.quantize" % scheme)
    x, y = synth_batch(16, device="cuda")

    def fresh():
        m = quantize.prepare(build_model(arch), inplace=True, **kw).cuda().train()
        return m, make_optimizer(m, 0.01, wd)
    m1, o1 = fresh()
    eager = [float(train_step(m1, o1, x, y)[0].detach()) for _ in range(6)]
    m2, o2 = fresh()
    g = GraphedTrainStep(m2, o2, x, y, warmup=2)          # 2 eager steps, then replays
    graphed = [float(g.step()[0]) for _ in range(4)]
    g.finish()
    print(key, "eager", eager, "graphed", graphed)
    assert all(int(st["step"]) == 6 for st in o2.state.values())
    chaotic = "wbwtab" in key
    for a, b in zip(eager[2:], graphed):
        assert abs(a - b) <= (0.25 if chaotic else 2e-2) * max(1.0, abs(a)), (eager, graphed)   # eager vs captured allocation changes nothing numerically, but low-bit nets amplify any last-bit difference of the warm-up steps
    if not chaotic:
        for (n_, p1), (_, p2) in zip(m1.named_parameters(), m2.named_parameters()):
            assert float((p1 - p2).abs().max()) <= 0.3 * max(1.0, float(p1.abs().max())), n_     # Adam at lr 0.01 amplifies the round-off of the atomic wgrad
    # new data through the static input tensors
    x2, y2 = synth_batch(16, seed=99, device="cuda")
    g.data.copy_(x2); g.target.copy_(y2)
    assert torch.isfinite(g.step()[0])


def test_graphed_step_follows_lr_schedule():
    """the reference edits param_group['lr'] every epoch (wbwtab/main.py:62-66 adjust_learning_rate).  The captured Adam launch
    reads lr / weight_decay from device memory, refreshed before each replay: a replayed step after an lr edit must equal the eager
    step with the same lr (lr = 0 is the sharpest check: parameters must not move at all)."""
    from micronet_amd.train import GraphedTrainStep, build_model, make_optimizer, synth_batch
    from micronet.compression.quantization.wqaq.dorefa import quantize
    x, y = synth_batch(16, device="cuda")
    m = quantize.prepare(build_model("nin_gc"), inplace=True, a_bits=8, w_bits=8).cuda().train()
    o = make_optimizer(m, 0.01, 1e-5)
    g = GraphedTrainStep(m, o, x, y, warmup=2)
    g.step()
    before = [p.detach().clone() for p in m.parameters()]
    for grp in o.param_groups:
        grp["lr"] = 0.0
    g.step()
    torch.cuda.synchronize()
    assert all(torch.equal(a, p.detach()) for a, p in zip(before, m.parameters())), "lr = 0 after capture must freeze the parameters"
    for grp in o.param_groups:
        grp["lr"] = 0.001
    g.step()
    torch.cuda.synchronize()
    moved = max(float((a - p.detach()).abs().max()) for a, p in zip(before, m.parameters()))
    assert 0 < moved <= 0.0011, moved            # Adam moves every element by at most ~lr per step
    sd = o.state_dict()                          # state_dict() syncs the replayed step count
    assert all(int(st["step"]) == 5 for st in sd["state"].values())
    for grp in o.param_groups:
        grp["betas"] = (0.5, 0.999)
    with pytest.raises(Exception):
        g.step()


def test_wbwtab_fused_pipeline_usage_scenarios(golden):
    """The packed / fused wbwtab pipeline under the ways a training script uses a model: odd batch sizes (kernels fall back or
    clamp), eval + no_grad inference, deepcopy, state_dict interchange with the un-fused module graph and with the reference's
    key layout, batch 1."""
    import copy
    from micronet_amd.train import build_model
    from micronet.compression.quantization.wbwtab import quantize
    fused = quantize.prepare(build_model("nin_gc"), inplace=True, A=2, W=3).cuda().train()
    plain = quantize.prepare(build_model("nin_gc"), inplace=True, A=2, W=3, fuse_bn_act=False, fold_shuffle=False,
                             packed_activations=False, fuse_conv_bn=False).cuda().train()
    assert list(fused.state_dict().keys()) == list(plain.state_dict().keys())
    assert [[k, list(v.shape)] for k, v in fused.state_dict().items()] == golden.meta["surface"]["c2_nin_gc_wbwtab_w3a2"]["state"]
    for B in (1, 3, 5, 8):
        x = torch.randn(B, 3, 32, 32, device="cuda")
        out = fused(x)
        assert out.shape == (B, 10) and torch.isfinite(out).all()
        out.square().mean().backward()
        assert all(torch.isfinite(p.grad).all() for p in fused.parameters())
        fused.zero_grad()
    # same weights -> same function (up to BatchNorm-output ties) in eval mode, where the running statistics normalise
    plain.load_state_dict(fused.state_dict())
    fused.eval(), plain.eval()
    x = torch.randn(16, 3, 32, 32, device="cuda")
    with torch.no_grad():
        of, op = fused(x), plain(x)
    assert of.shape == op.shape and float((of.argmax(1) == op.argmax(1)).float().mean()) >= 0.8
    clone = copy.deepcopy(fused)
    with torch.no_grad():
        assert torch.equal(clone(x), of)
    # non-square images, H*W not a multiple of 64
    fused.train()
    y = fused.model[0:3](torch.randn(2, 3, 24, 40, device="cuda"))
    assert y.shape[2:] == (24, 40)
    y.float().sum().backward() if y.requires_grad else None


@pytest.mark.parametrize("key", ["c1_nin_gc_dorefa_w8a8", "c2_nin_gc_wbwtab_w3a2", "c3_nin_gc_iao_w8a8_bnfuse", "c4_resnet18_dorefa_w2a2", "c5_resnet18_iao_w4a4"])
def test_bench_workloads_never_fall_through_to_stock_operators(key):
    """a module of this package hands a geometry its kernels do not cover to the stock torch operator and counts it (ops.note_fallback); for every
    benched net one training step must leave that counter EMPTY -- no MIOpen / ATen convolution, BatchNorm or pooling on the hot path."""
    from micronet_amd import ops
    from micronet_amd.train import build_model, make_optimizer, synth_batch, train_step
    arch, scheme, kw, B, wd = CFG[key]
    quantize = importlib.import_module("micronet.compression.quantization.%s.quantize" % scheme)
    model = quantize.prepare(build_model(arch), inplace=True, **kw).cuda().train()
    opt = make_optimizer(model, 0.01, wd)
    x, y = synth_batch(8, device="cuda")
    ops.fallback_counts(reset=True)
    train_step(model, opt, x, y)
    assert ops.fallback_counts() == {}, ops.fallback_counts()


def test_iao_resnet_quantadd_observers_read_producer_partials():
    """every QuantAdd of the IAO ResNet takes both input ranges from the (min, max) partials its producers left (BatchNorm apply pass, the previous block's
    QuantAdd + ReLU, the stem's BatchNorm + ReLU): no pass over the two tensors (mn_iao_qadd_observe) in a training step"""
    from micronet_amd import ops
    from micronet_amd.train import build_model, make_optimizer, synth_batch, train_step
    arch, scheme, kw, B, wd = CFG["c5_resnet18_iao_w4a4"]
    quantize = importlib.import_module("micronet.compression.quantization.%s.quantize" % scheme)
    model = quantize.prepare(build_model(arch), inplace=True, **kw).cuda().train()
    opt = make_optimizer(model, 0.01, wd)
    x, y = synth_batch(8, device="cuda")
    calls = {"full": 0, "partials": 0}
    real_full, real_part = ops.iao_qadd_observe, ops.iao_qadd_observe_partials
    ops.iao_qadd_observe = lambda *a, **k: (calls.__setitem__("full", calls["full"] + 1), real_full(*a, **k))[1]
    ops.iao_qadd_observe_partials = lambda *a, **k: (calls.__setitem__("partials", calls["partials"] + 1), real_part(*a, **k))[1]
    try:
        train_step(model, opt, x, y)
    finally:
        ops.iao_qadd_observe, ops.iao_qadd_observe_partials = real_full, real_part
    assert calls == {"full": 0, "partials": 8}, calls


def _first_block_run(model, x, monkeypatch, **knobs):
    """One forward + backward of the net's FIRST block (conv + BatchNorm + activation of models/nin_gc.py:53-59) -> (output as float, parameter gradients)."""
    import copy
    from micronet_amd import ops
    for k, v in knobs.items():
        monkeypatch.setattr(ops, k, v)
    blk = copy.deepcopy(model.model[0]).train()
    out = blk(x)
    of = out.to_float() if hasattr(out, "to_float") else (out.materialize() if hasattr(out, "materialize") else out)          # SignTensor / QActTensor / plain
    gen = torch.Generator(device="cuda").manual_seed(5)
    gout = torch.randn(of.shape, device="cuda", generator=gen)
    out.backward(gout)
    return of.detach().clone(), {n: p.grad.detach().clone() for n, p in blk.named_parameters()}, blk


def test_first_block_fused_vs_unfused(monkeypatch):
    """The fused first block (ops.FirstConvLazy -> FirstConvBNSign: conv + Gram-data statistics + BatchNorm + sign in one kernel, backward on pass bits) against
    the unfused kernels and the two-pass backward: same codes, gradients bit-identical to the one-pass unfused path and within float rounding of the two-pass one;
    a foreign consumer of the un-computed conv output sees the tensor the reference produces."""
    from micronet_amd import ops
    from micronet_amd.sign_tensor import LazyConvOut
    from micronet_amd.train import build_model
    from micronet.compression.quantization.wbwtab import quantize
    torch.manual_seed(3)
    model = quantize.prepare(build_model("nin_gc"), inplace=True, A=2, W=3).cuda().train()
    assert model.model[0].conv.lazy_for_bn is True
    x = torch.randn(16, 3, 32, 32, device="cuda")
    o_f, g_f, blk = _first_block_run(model, x, monkeypatch, FIRST_FUSED=True)
    o_u, g_u, _ = _first_block_run(model, x, monkeypatch, FIRST_FUSED=False)
    o_2, g_2, _ = _first_block_run(model, x, monkeypatch, FIRST_FUSED=False, FIRST_GRAM=False)
    assert torch.equal(o_f, o_u)
    for n in g_f:
        assert torch.equal(g_f[n], g_u[n]), n
    flips = float((o_u != o_2).float().mean())          # statistics from the Gram data vs from y: signs differ only at BatchNorm ties
    assert flips <= 1e-5
    for n in g_u:
        if n.endswith("conv.bias"):          # sum of dy: zero but for rounding, in every path
            assert float(g_u[n].abs().max()) <= 1e-4 * float(g_u["bn.bias"].abs().max()) and float(g_2[n].abs().max()) <= 1e-4 * float(g_2["bn.bias"].abs().max())
            continue
        scale = float(g_2[n].abs().max())
        assert float((g_u[n] - g_2[n]).abs().max()) <= (2e-3 if flips else 2e-5) * scale, n
    monkeypatch.setattr(ops, "FIRST_FUSED", True)
    monkeypatch.setattr(ops, "FIRST_GRAM", True)
    # a second backward through the retained graph gives the same gradients (nothing the backward needs is consumed by the first one)
    import copy
    blk2 = copy.deepcopy(model.model[0]).train()
    out2 = blk2(x)
    gout = torch.randn(out2.shape, device="cuda")
    out2.backward(gout, retain_graph=True)
    g_a = {n: p.grad.detach().clone() for n, p in blk2.named_parameters()}
    blk2.zero_grad()
    out2.backward(gout)
    for n, p in blk2.named_parameters():
        assert torch.equal(p.grad, g_a[n]), n
    y = blk.conv(x)
    assert isinstance(y, LazyConvOut) and y.recipe["kind"] == "first"
    ref = torch.nn.functional.conv2d(x, blk.conv.weight, blk.conv.bias, padding=blk.conv.padding)
    assert float(((y + 0.0) - ref).abs().max()) <= 2e-5 * float(ref.abs().max())          # any torch operator materialises it
    (y * 1.0).square().mean().backward()          # ... and the conv's ordinary backward runs behind it
    assert torch.isfinite(blk.conv.weight.grad).all()


def test_first_block_dorefa_mask_backward(monkeypatch):
    """The DoReFa first block: default (conv, BatchNorm + ReLU + quantizer pass that leaves the pass nibbles, one-pass backward on them) against the fused forward
    (knob MN_FIRST_FUSED_QA) and against the two-pass backward."""
    from micronet_amd import ops
    from micronet_amd.train import build_model
    from micronet.compression.quantization.wqaq.dorefa import quantize
    torch.manual_seed(4)
    model = quantize.prepare(build_model("nin_gc"), inplace=True, a_bits=4, w_bits=4).cuda().train()
    assert model.model[0].conv.lazy_for_bn == "qa"
    x = torch.randn(16, 3, 32, 32, device="cuda")
    o_d, g_d, _ = _first_block_run(model, x, monkeypatch, FIRST_FUSED_QA=False)
    o_f, g_f, _ = _first_block_run(model, x, monkeypatch, FIRST_FUSED_QA=True)
    o_2, g_2, _ = _first_block_run(model, x, monkeypatch, FIRST_FUSED_QA=False, FIRST_GRAM=False)
    assert torch.equal(o_d, o_f)
    for n in g_d:
        assert torch.equal(g_d[n], g_f[n]), n
    assert float((o_d - o_2).abs().max()) <= 1e-4 * float(o_2.abs().max())          # statistics from the Gram data vs from y
    for n in g_d:
        if n.endswith("conv.bias"):
            continue
        assert float((g_d[n] - g_2[n]).abs().max()) <= 2e-3 * float(g_2[n].abs().max()), n


def test_iao_resnet_shortcut_gradient_folded_into_backward_data(monkeypatch):
    """ops.RES_ADD_FOLD: the identity shortcut's gradient of an IAO residual block is added inside the first conv's backward-data store (mn_actq.dx_add) instead of by
    autograd's accumulate kernel.  a + b == b + a: every gradient of a whole-net step is bit-identical with the fold on and off; the five identity blocks of
    resnet18 fold, the three downsampling ones do not."""
    import copy
    from micronet_amd import ops
    from micronet_amd.train import build_model, synth_batch
    arch, scheme, kw, B, wd = CFG["c5_resnet18_iao_w4a4"]
    quantize = importlib.import_module("micronet.compression.quantization.%s.quantize" % scheme)
    torch.manual_seed(9)
    base = quantize.prepare(build_model(arch), inplace=True, **kw).cuda().train()
    x, y = synth_batch(8, device="cuda")
    grads, folds = {}, {}
    real, real_bn = ops.IaoQuantAdd.apply, ops.IaoQuantAddBN.apply          # (round 6: the block's QuantAdd is IaoQuantAddBN when the BatchNorm in front stays un-computed)
    for fold in (True, False):
        monkeypatch.setattr(ops, "RES_ADD_FOLD", fold)
        model = copy.deepcopy(base)
        n = {"tok": 0}

        def counting(*a, _n=n, _real=real):
            _n["tok"] += int(len(a) > 7 and a[7] is not None)
            return _real(*a)
        monkeypatch.setattr(ops.IaoQuantAdd, "apply", staticmethod(counting))
        monkeypatch.setattr(ops.IaoQuantAddBN, "apply", staticmethod(lambda *a, _n=n: (_n.__setitem__("tok", _n["tok"] + int(len(a) > 7 and a[7] is not None)), real_bn(*a))[1]))
        try:
            torch.nn.functional.cross_entropy(model(x), y).backward()
        finally:
            monkeypatch.setattr(ops.IaoQuantAdd, "apply", real)
            monkeypatch.setattr(ops.IaoQuantAddBN, "apply", real_bn)
        grads[fold] = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        folds[fold] = n["tok"]
    assert folds == {True: 5, False: 0}, folds
    for k in grads[True]:
        assert torch.equal(grads[True][k], grads[False][k]), k


def _iao_resnet_two_steps(monkeypatch, iaoq_knobs, batch=8):
    """two training steps of the IAO resnet18 (observer first call, then the EMA update) -> (losses, parameters, every buffer), + the prepared model of step 2"""
    from micronet_amd.quantization.wqaq.iao import quantize as iaoq
    from micronet_amd.train import build_model, make_optimizer, synth_batch, train_step
    arch, scheme, kw, B, wd = CFG["c5_resnet18_iao_w4a4"]
    for k, v in iaoq_knobs.items():
        monkeypatch.setattr(iaoq, k, v)
    model = iaoq.prepare(build_model(arch), inplace=True, **kw).cuda().train()
    opt = make_optimizer(model, 0.01, wd)
    losses = []
    for step in range(2):
        x, y = synth_batch(batch, seed=77 + step, device="cuda")
        losses.append(float(train_step(model, opt, x, y)[0]))
    return losses, {k: v.detach().clone() for k, v in model.state_dict().items()}, model


@pytest.mark.parametrize("key", ["c2_nin_gc_wbwtab_w3a2", "c1_nin_gc_dorefa_w8a8"])
def test_tail_fused_vs_stock(monkeypatch, key):
    """Round 6: the net's tail -- BatchNorm2d -> ReLU -> AvgPool2d over the map (models/nin_gc.py:136-147) and nn.CrossEntropyLoss() (wqaq/dorefa/main.py:87-92) -- as
    three launches (mn_bnrelu_gap_fwd / _bwd, mn_cross_entropy_fwd + mn_scale_by) against MIOpen's BatchNorm / ATen's ReLU, pooling, softmax and nll_loss on the same
    prepared net: same loss and the same gradients of EVERY parameter to fp32 round-off, no fallback, and the stock modules when the knob is off."""
    import copy
    import torch.nn.functional as Fn
    from micronet_amd import nn as mnn, ops
    from micronet_amd.train import build_model, synth_batch
    arch, scheme, kw, B, wd = CFG[key]
    quantize = importlib.import_module("micronet.compression.quantization.%s.quantize" % scheme)
    base = quantize.prepare(build_model(arch), inplace=True, **kw).cuda().train()
    assert isinstance(base.model[-2].bn, mnn.TailBNMixin)
    x, y = synth_batch(32, device="cuda")
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(mnn, "TAIL_FUSED", fused)
        m = copy.deepcopy(base)
        ops.fallback_counts(reset=True)
        loss = ops.cross_entropy(m(x), y) if fused else Fn.cross_entropy(m(x), y)
        loss.backward()
        assert ops.fallback_counts() == {}, ops.fallback_counts()
        res[fused] = (float(loss), {k: p.grad.detach().clone() for k, p in m.named_parameters()}, {k: v.detach().clone() for k, v in m.state_dict().items()})
    assert abs(res[True][0] - res[False][0]) <= 2e-6 * abs(res[False][0]), (res[True][0], res[False][0])
    gscale = max(float(g.abs().max()) for g in res[False][1].values())
    for k, g0 in res[False][1].items():
        g1 = res[True][1][k]
        # (a conv bias in front of a BatchNorm has an exactly zero gradient: what both runs hold there is round-off of the net's gradient scale)
        assert float((g1 - g0).abs().max()) <= 2e-5 * float(g0.abs().max()) + 1e-6 * gscale, (k, float((g1 - g0).abs().max()), float(g0.abs().max()), gscale)
    for k, v0 in res[False][2].items():          # running statistics, counters
        v1 = res[True][2][k]
        assert torch.allclose(v1.float(), v0.float(), rtol=2e-5, atol=1e-7, equal_nan=True), k


@pytest.mark.parametrize("scheme,kw,up_call,final_call", [
    ("wbwtab", dict(A=2, W=3), ("mn_conv2d_bwd_bnh_up",), "mn_bnh_bwd_sums_final"),
    ("wqaq.dorefa", dict(a_bits=2, w_bits=2), ("mn_conv2d_bwd_qa_up", "mn_conv2d_bwd_codes_up"), "mn_qa_bwd_sums_final"),
])
def test_upstream_sums_ride_on_the_next_blocks_backward(monkeypatch, scheme, kw, up_call, final_call):
    """Round 6, ops.UpSums: the one-launch backward of a pointwise block (k_pwb<.., UP>) also forms the BatchNorm-backward sums of the block in front -- its dx IS that
    block's incoming gradient -- so that block only finishes partials instead of streaming (gradient, stash) again (wbwtab: k_bnh_partial, DoReFa: k_qa_partial).
    Whole nin_gc step with the hand-over on and off: the hand-over happens for the un-pooled pointwise pairs and every gradient agrees to the summation order."""
    import copy
    from micronet_amd import ops
    from micronet_amd.train import build_model, synth_batch
    quantize = importlib.import_module("micronet.compression.quantization.%s.quantize" % scheme)
    torch.manual_seed(11)
    base = quantize.prepare(build_model("nin_gc"), inplace=True, **kw).cuda().train()
    x, y = synth_batch(32, device="cuda")
    real = ops._call
    res, calls = {}, {}
    monkeypatch.setattr(ops, "UP_SUMS_PLAIN", True)          # (k-bit blocks: also behind a plain gradient, which the default leaves alone -- slower there)
    for on in (True, False):
        monkeypatch.setattr(ops, "UP_SUMS_FOLD", on)
        n = {}
        monkeypatch.setattr(ops, "_call", lambda name, *a, _n=n: (_n.__setitem__(name, _n.get(name, 0) + 1), real(name, *a))[1])
        m = copy.deepcopy(base)
        ops.fallback_counts(reset=True)
        ops.cross_entropy(m(x), y).backward()
        assert ops.fallback_counts() == {}, ops.fallback_counts()
        res[on] = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
        calls[on] = n
    monkeypatch.setattr(ops, "_call", real)
    ups = {on: sum(calls[on].get(c, 0) for c in up_call) for on in (True, False)}
    # nin_gc (models/nin_gc.py:18-59): the consumer is a pointwise block k_pwb covers, the block in front left a byte / 16-bit stash and is not pooled
    if scheme == "wbwtab":          # + the two pointwise blocks behind a 3x3 block (layers 4 | 5, 7 | 8: mn_conv2d_bwd_bnh_up9)
        assert calls[True].get("mn_conv2d_bwd_bnh_up9", 0) == 2 and calls[False].get("mn_conv2d_bwd_bnh_up9", 0) == 0, calls[True]
        ups[True] += 2
    assert ups[True] >= 2 and calls[True].get(final_call, 0) == ups[True], calls[True]
    assert ups[False] == 0 and calls[False].get(final_call, 0) == 0, calls[False]
    gscale = max(float(g.abs().max()) for g in res[False].values())
    for k, g0 in res[False].items():
        g1 = res[True][k]
        assert float((g1 - g0).abs().max()) <= 2e-5 * float(g0.abs().max()) + 1e-6 * gscale, (k, float((g1 - g0).abs().max()), float(g0.abs().max()))


@pytest.mark.parametrize("codes,add", [(True, True), (True, False), (False, True)])
def test_iao_resnet_block_fused_passes(monkeypatch, codes, add):
    """Round 6, the IAO BasicBlock (models/resnet.py:17-29, 60-65 under wqaq/iao/quantize.py:492-507, 1484-1498) in fewer passes:
    ``codes``: the activation between the two convs stays un-computed (LazyBNAct) -- its range comes from the first conv's accumulator extrema, the second conv pulls
    its codes from the first conv's output in one pass;  ``add``: the BatchNorm(s) in front of the block's QuantAdd stay un-computed too -- ONE pass normalises,
    quantises, adds and rectifies, and each BatchNorm's backward reads the block-output gradient with the clip-STE / ReLU decisions as bits.
    Same codes, same bits, same observer state, same order of summation: two training steps are bit-identical to the unfused modules -- losses, every parameter, every
    buffer -- and all eight blocks of resnet18 take the paths (nothing materialised)."""
    from micronet_amd import ops
    n = {"pull": 0, "mat": 0, "addbn": 0, "add": 0}
    real_pull, real_mat, real_addbn, real_add = ops.iao_bn_apply_codes, ops.LazyBNActToFloat.apply, ops.IaoQuantAddBN.apply, ops.IaoQuantAdd.apply
    monkeypatch.setattr(ops, "iao_bn_apply_codes", lambda *a, **k: (n.__setitem__("pull", n["pull"] + 1), real_pull(*a, **k))[1])
    monkeypatch.setattr(ops.LazyBNActToFloat, "apply", staticmethod(lambda *a: (n.__setitem__("mat", n["mat"] + 1), real_mat(*a))[1]))
    monkeypatch.setattr(ops.IaoQuantAddBN, "apply", staticmethod(lambda *a: (n.__setitem__("addbn", n["addbn"] + 1), real_addbn(*a))[1]))
    monkeypatch.setattr(ops.IaoQuantAdd, "apply", staticmethod(lambda *a: (n.__setitem__("add", n["add"] + 1), real_add(*a))[1]))
    l1, s1, m1 = _iao_resnet_two_steps(monkeypatch, dict(_FUSE_BN_CODES=codes, _FUSE_BN_ADD=add))
    assert n == {"pull": 16 if codes else 0, "mat": 0, "addbn": 16 if add else 0, "add": 0 if add else 16}, n
    l0, s0, _ = _iao_resnet_two_steps(monkeypatch, dict(_FUSE_BN_CODES=False, _FUSE_BN_ADD=False))
    assert l1 == l0, (l1, l0)
    for k in s0:
        assert torch.equal(s1[k], s0[k]) or (torch.isnan(s1[k]).all() and torch.isnan(s0[k]).all()), k
    if codes:
        # a foreign consumer of the un-computed activation sees the float32 tensor the unfused BatchNorm + ReLU writes
        from micronet_amd.sign_tensor import LazyBNAct
        blk = m1.conv2_x[0].residual_function
        x = torch.randn(4, 64, 32, 32, device="cuda")
        lazy = blk[2](blk[1](blk[0](x)))
        assert isinstance(lazy, LazyBNAct)
        a = lazy + 0.0
        assert type(a) is torch.Tensor and a.shape == lazy.shape and float(a.min()) == 0.0


@pytest.mark.parametrize("key", ["c2_nin_gc_wbwtab_w3a2", "c1_nin_gc_dorefa_w8a8", "c5_resnet18_iao_w4a4"])
def test_forward_without_backward_does_not_leak(key):
    """A grad-enabled training-mode forward that is never backpropagated (a skipped step, an exception) must not keep its activations alive: the hand-over objects
    the fused paths hang on tensors (ops.FirstConvRecord, ops.ResidualToken, LazyConvOut recipes) hold autograd nodes weakly or not at all -- no reference cycle through
    C++ saved tensors."""
    import gc
    from micronet_amd.train import build_model, synth_batch
    arch, scheme, kw, B, wd = CFG[key]
    quantize = importlib.import_module("micronet.compression.quantization.%s.quantize" % scheme)
    model = quantize.prepare(build_model(arch), inplace=True, **kw).cuda().train()
    x, _ = synth_batch(8, device="cuda")
    used = []
    for _ in range(5):
        out = model(x)
        del out
        gc.collect()
        torch.cuda.synchronize()
        used.append(torch.cuda.memory_allocated())
    assert used[4] == used[2] == used[3], used


def test_pooled_codes_come_from_the_sign_pass(monkeypatch):
    """Round 6: a BatchNorm+sign block in front of nn.MaxPool2d(2, 2) (models/nin_gc.py:88,119; prepare() marks it ``pool_next``) writes the pooled sign codes in its own
    sign pass (mn_qconv_bnsign_fwd_stash_pool); the pool module hands them on instead of launching mn_maxpool2x2_sign8_fwd.  Same logits and the same gradients to the
    bit with the hand-over on and off; both pools of nin_gc take it."""
    import copy
    from micronet_amd import ops
    from micronet_amd.train import build_model, synth_batch
    from micronet.compression.quantization.wbwtab import quantize
    torch.manual_seed(13)
    base = quantize.prepare(build_model("nin_gc"), inplace=True, A=2, W=3).cuda().train()
    assert sum(int(getattr(m, "pool_next", False)) for m in base.modules()) == 2
    x, y = synth_batch(32, device="cuda")
    real = ops._call
    res = {}
    for on in (True, False):
        monkeypatch.setattr(ops, "POOL_IN_SIGN_PASS", on)
        n = {}
        monkeypatch.setattr(ops, "_call", lambda name, *a, _n=n: (_n.__setitem__(name, _n.get(name, 0) + 1), real(name, *a))[1])
        m = copy.deepcopy(base)
        out = m(x)
        ops.cross_entropy(out, y).backward()
        res[on] = (out.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters()}, n)
    monkeypatch.setattr(ops, "_call", real)
    assert res[True][2].get("mn_qconv_bnsign_fwd_stash_pool", 0) == 2 and res[True][2].get("mn_maxpool2x2_sign8_fwd", 0) == 0, res[True][2]
    assert res[False][2].get("mn_qconv_bnsign_fwd_stash_pool", 0) == 0 and res[False][2].get("mn_maxpool2x2_sign8_fwd", 0) == 2, res[False][2]
    assert torch.equal(res[True][0], res[False][0])
    for k, g0 in res[False][1].items():
        assert torch.equal(res[True][1][k], g0), k
