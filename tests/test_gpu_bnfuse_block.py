"""The BN-fused IAO blocks behind the module surface (prepare(bn_fuse=True) on ConvBNReLU chains): the fused pipeline with its LIVE hand-overs -- (min, max)
partials from a block's epilogue to the next block's observer, the ReLU mask applied by the consumer's backward-data kernel -- against the torch-CPU oracle of the
reference (wqaq/iao/quantize.py:837-994 + models/nin_gc.py:53-59), and against the product's own generic path (raw conv + statistics passes)."""
import copy
import importlib

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from oracle import torch_oracle as TO  # noqa: E402
from micronet_amd.models.nin_gc import ConvBNReLU  # noqa: E402

KW = dict(a_bits=8, w_bits=8, q_type=0, q_level=0, weight_observer=0, bn_fuse=True)


def _q():
    return importlib.import_module("micronet.compression.quantization.wqaq.iao.quantize")


def _chain(seed=0):
    torch.manual_seed(seed)
    net = nn.Sequential(ConvBNReLU(32, 64, kernel_size=1, groups=2),
                        ConvBNReLU(64, 64, kernel_size=1, groups=2, channel_shuffle=1, shuffle_groups=2),
                        ConvBNReLU(64, 32, kernel_size=1, groups=1, channel_shuffle=1, shuffle_groups=2))
    for m in net.modules():
        if isinstance(m, nn.BatchNorm2d):
            nn.init.uniform_(m.weight, 0.4, 1.3)
            nn.init.normal_(m.bias, 0, 0.2)
    return net


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _run(model, x, g, steps):
    recs = []
    for _ in range(steps):
        for p in model.parameters():
            p.grad = None
        xx = x.clone().requires_grad_(True)
        out = model(xx)
        out.backward(g)
        recs.append(dict(out=out.detach().clone(), dx=xx.grad.clone(), **{"d_" + n: p.grad.clone() for n, p in model.named_parameters()}))
    return recs


@pytest.mark.parametrize("steps", [2])
def test_chain_vs_oracle_with_live_handover(steps):
    Q = _q()
    base = _chain()
    x = torch.randn(8, 32, 8, 8) * 1.3 + 0.2
    g = torch.randn(8, 32, 8, 8)
    orc = TO.prepare(copy.deepcopy(base), "iao", inplace=True, **KW).train()
    o64 = TO.prepare(copy.deepcopy(base), "iao", inplace=True, **KW).double().train()
    ours = Q.prepare(copy.deepcopy(base), inplace=True, **KW).cuda().train()
    assert all(b.conv.relu_fused for b in ours) and ours[1].conv.in_shuffle_groups == 2 and ours[1].channel_shuffle_flag == 0
    from micronet_amd import ops
    masks = []
    real_mask = ops.relu_mask
    ops.relu_mask = lambda g_, a_: (masks.append(1), real_mask(g_, a_))[1]
    try:
        r_ours = _run(ours, x.cuda(), g.cuda(), steps)
    finally:
        ops.relu_mask = real_mask
    # only the LAST block masks its own gradient (nobody behind it): the two hand-overs in front of it were pre-masked by their consumers
    assert len(masks) == steps, masks
    r_ref, r_64 = _run(orc, x, g, steps), _run(o64, x.double(), g.double(), steps)
    for s in range(steps):
        for k in r_ref[s]:
            ko = k
            e32, e64 = _rel(r_ours[s][ko], r_ref[s][k]), _rel(r_ours[s][ko], r_64[s][k])
            own = _rel(r_ref[s][k], r_64[s][k])
            if k.endswith(".conv.bias"):       # zero in exact arithmetic (a bias in front of batch statistics): rounding noise on both sides
                assert float(r_ours[s][ko].abs().max()) <= 1e-5 * float(r_ref[s]["d_0.conv.beta"].abs().max())
                continue
            assert e32 <= 1e-5 or e64 <= max(1e-5, 2 * own), (s, k, e32, e64, own)
    ob = dict(orc.named_buffers())
    for n, b in ours.named_buffers():          # running statistics of every block after `steps` steps (the oracle names them the same)
        if n.endswith("running_mean") or n.endswith("running_var"):
            assert _rel(b, ob[n]) <= 1e-5, n


def test_fused_equals_generic_path_and_hooks_see_true_gradients():
    """the fused blocks against the product's generic path (raw statistics conv, separate ReLU) on the same weights; and with a tensor hook on an intermediate
    activation the consumer must NOT pre-mask (the hook would see a masked gradient): the hooked gradient equals the generic path's"""
    Q = _q()
    base = _chain(1)
    x = (torch.randn(4, 32, 8, 8) * 1.3 + 0.2).cuda()
    g = torch.randn(4, 32, 8, 8).cuda()
    fused = Q.prepare(copy.deepcopy(base), inplace=True, **KW).cuda().train()
    plain = Q.prepare(copy.deepcopy(base), inplace=True, fuse_blocks=False, **KW).cuda().train()
    assert not any(getattr(b.conv, "relu_fused", False) for b in plain)
    rf, rp = _run(fused, x, g, 2), _run(plain, x, g, 2)
    for s in range(2):
        for k in rf[s]:
            if k.endswith(".conv.bias"):
                continue
            assert _rel(rf[s][k], rp[s][k]) <= 2e-5, (s, k, _rel(rf[s][k], rp[s][k]))
    # hook on the activation between block 0 and block 1
    seen = {}

    def run_hooked(model):
        for p in model.parameters():
            p.grad = None
        xx = x.clone().requires_grad_(True)
        a0 = model[0](xx)
        a0.register_hook(lambda gr: seen.__setitem__(id(model), gr.detach().clone()))
        model[2](model[1](a0)).backward(g)
        return xx.grad.clone()
    dxf, dxp = run_hooked(fused), run_hooked(plain)
    assert _rel(seen[id(fused)], seen[id(plain)]) <= 2e-5
    assert _rel(dxf, dxp) <= 2e-5


def _mixed_chain(seed=2):
    torch.manual_seed(seed)
    net = nn.Sequential(ConvBNReLU(3, 32, kernel_size=5, padding=2),                                  # the first layer: image input, first-layer kernels
                        ConvBNReLU(32, 32, kernel_size=1, groups=2),                                    # pointwise fused
                        nn.MaxPool2d(2, 2),                                                             # -> QuantMaxPool2d (fused quantizer + pool)
                        ConvBNReLU(32, 64, kernel_size=3, padding=1, groups=2, channel_shuffle=1, shuffle_groups=2),      # generic fused (k x k)
                        ConvBNReLU(64, 16, kernel_size=1, groups=1, channel_shuffle=1, shuffle_groups=2))
    for m in net.modules():
        if isinstance(m, nn.BatchNorm2d):
            nn.init.uniform_(m.weight, 0.4, 1.3)
            nn.init.normal_(m.bias, 0, 0.2)
    return net


def test_mixed_chain_vs_oracle():
    """first layer (5 x 5 on the image) -> pointwise -> QuantMaxPool2d -> grouped 3 x 3 with a folded shuffle -> pointwise: every hand-over of the nin_gc pipeline
    (ReLU masks pre-applied by the pool / the next block, (min, max) partials from epilogues and from the pool) against the oracle, two steps"""
    Q = _q()
    base = _mixed_chain()
    x = torch.randn(4, 3, 16, 16)
    g = torch.randn(4, 16, 8, 8)
    orc = TO.prepare(copy.deepcopy(base), "iao", inplace=True, **KW).train()
    o64 = TO.prepare(copy.deepcopy(base), "iao", inplace=True, **KW).double().train()
    ours = Q.prepare(copy.deepcopy(base), inplace=True, **KW).cuda().train()
    steps = 2

    def run(model, xx_, g_):
        recs = []
        for _ in range(steps):
            for p in model.parameters():
                p.grad = None
            out = model(xx_)
            out.backward(g_)
            recs.append(dict(out=out.detach().clone(), **{"d_" + n: p.grad.clone() for n, p in model.named_parameters()}))
        return recs
    r_ours, r_ref, r_64 = run(ours, x.cuda(), g.cuda()), run(orc, x, g), run(o64, x.double(), g.double())
    # which kernel family ran each block: first-layer Gram path (generic node), pointwise, the grouped 3 x 3 image-resident kernels behind the pool, pointwise
    assert [b.conv.__dict__.get("_mn_path") for b in ours if hasattr(b, "conv")] == ["generic", "pw", "g3", "pw"]
    for s in range(steps):
        for k in r_ref[s]:
            if k.endswith(".conv.bias"):
                continue
            e32, e64, own = _rel(r_ours[s][k], r_ref[s][k]), _rel(r_ours[s][k], r_64[s][k]), _rel(r_ref[s][k], r_64[s][k])
            assert e32 <= 1e-5 or e64 <= max(1e-5, 2 * own), (s, k, e32, e64, own)


@pytest.mark.parametrize("hw", [16, 32])
def test_grouped_3x3_family_equals_generic_path(hw):
    """the grouped 3 x 3 block behind the pool on csrc/iao_g3.hip (statistics from the accumulators, d y_raw recomputed, shuffle in the addressing) against the
    product's generic path (raw conv written + statistics passes + materialised shuffle) on the same weights: 8 x 8 and 16 x 16 maps, two steps"""
    Q = _q()
    base = _mixed_chain(4)
    x = torch.randn(4, 3, hw, hw).cuda()
    g = torch.randn(4, 16, hw // 2, hw // 2).cuda()

    def run(model):
        recs = []
        for _ in range(2):
            for p in model.parameters():
                p.grad = None
            out = model(x)
            out.backward(g)
            recs.append(dict(out=out.detach().clone(), **{"d_" + n: p.grad.clone() for n, p in model.named_parameters()}))
        return recs
    a = Q.prepare(copy.deepcopy(base), inplace=True, **KW).cuda().train()
    ra = run(a)
    assert a[3].conv.__dict__.get("_mn_path") == "g3"
    import micronet_amd.quantization.wqaq.iao.quantize as QI          # (the module that owns the knob; `micronet.compression...` may be an alias package)
    keep = QI._FUSE_G3
    QI._FUSE_G3 = False
    try:
        b = Q.prepare(copy.deepcopy(base), inplace=True, **KW).cuda().train()
        rb = run(b)
    finally:
        QI._FUSE_G3 = keep
    assert b[3].conv.__dict__.get("_mn_path") == "generic"
    for s in range(2):
        for k in ra[s]:
            if k.endswith(".conv.bias"):
                continue
            assert _rel(ra[s][k], rb[s][k]) <= 2e-5, (s, k, _rel(ra[s][k], rb[s][k]))


def test_conv_module_called_directly_keeps_the_reference_contract():
    """the conv of a fused block called on its own returns the UN-rectified output (what the reference's QuantBNFuseConv2d.forward returns) and accepts a gradient
    with respect to it -- the wrapper recomputes it for a foreign consumer"""
    Q = _q()
    base = _chain(3)
    x = (torch.randn(4, 32, 8, 8) * 1.3 + 0.2)
    g = torch.randn(4, 64, 8, 8)
    ours = Q.prepare(copy.deepcopy(base), inplace=True, **KW).cuda().train()
    orc = TO.prepare(copy.deepcopy(base), "iao", inplace=True, **KW).train()
    xo = x.clone().requires_grad_(True)
    yo = orc[0].conv(xo)
    yo.backward(g)
    xp = x.cuda().requires_grad_(True)
    yp = ours[0].conv(xp)
    assert type(yp).__name__ == "LazyReluConvOut"
    assert _rel(yp, yo) <= 1e-5 and float(yp.detach().min().cpu()) < 0          # not rectified
    torch.autograd.backward([yp], [g.cuda()])
    assert _rel(xp.grad, xo.grad) <= 1e-5
    assert _rel(ours[0].conv.weight.grad, orc[0].conv.weight.grad) <= 1e-5


def test_forward_without_backward_does_not_leak():
    """a grad-enabled training-mode forward that is never backpropagated (a skipped step, an exception) must free x, a and qw of every fused block: the ReLU
    node keeps its output through save_for_backward, not on ctx (that was a reference cycle only backward broke)."""
    import gc
    Q = _q()
    net = Q.prepare(_chain(5), inplace=True, a_bits=8, w_bits=8, q_type=0, q_level=0, weight_observer=0, bn_fuse=True).cuda().train()
    x = torch.randn(16, 32, 16, 16, device="cuda")
    for _ in range(3):
        net(x)
    gc.collect()
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    for _ in range(10):
        net(x)
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() <= base + (1 << 20), (torch.cuda.memory_allocated(), base)          # no gc.collect(): nothing may depend on the cycle collector
