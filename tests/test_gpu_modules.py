"""The torch.nn.Module surface on the MI355X vs (a) the golden vectors produced by the real reference and (b) the
torch-CPU oracle on fresh seeded inputs.  Tolerances: quantised codes / observer statistics bit-exact, fp32 channel
sums <= 5e-7 rel, float conv accumulate <= 1e-5 * max|ref| (BN-fuse included: its batch statistics are accumulated in fp64)."""
import importlib

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from oracle import torch_oracle as TO  # noqa: E402


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30)


def _q(scheme):
    return importlib.import_module("micronet.compression.quantization.%s.quantize" % scheme)


def _build(c, variant, m):
    base = c["base"]
    kw = dict(stride=c["stride"], padding=c["padding"], dilation=c["dilation"], groups=c["groups"], bias=c["bias"])
    cin, cout, k = c["cin"], c["cout"], c["k"]
    if variant.startswith("dorefa"):
        bits = int(variant[-1])
        mod = _q("wqaq.dorefa").QuantConv2d(cin, cout, k, a_bits=bits, w_bits=bits, **kw)
    elif variant.startswith("wbwtab"):
        mod = _q("wbwtab").QuantConv2d(cin, cout, k, W=int(variant[-1]), **kw)
    else:
        iao = _q("wqaq.iao")
        cfg = {"iao_w8a8_sym_c": dict(a_bits=8, w_bits=8, q_type=0, q_level=0),
               "iao_w4a4_sym_c": dict(a_bits=4, w_bits=4, q_type=0, q_level=0),
               "iao_w8a8_asym_l": dict(a_bits=8, w_bits=8, q_type=1, q_level=1),
               "iao_bnfuse_w8a8": dict(a_bits=8, w_bits=8, q_type=0, q_level=0)}[variant]
        mod = (iao.QuantBNFuseConv2d if "bnfuse" in variant else iao.QuantConv2d)(cin, cout, k, **kw, **cfg)
        if "bnfuse" in variant:
            mod.gamma.data = torch.from_numpy(m[f"{base}_gamma"].copy())
            mod.beta.data = torch.from_numpy(m[f"{base}_beta"].copy())
    mod.weight.data = torch.from_numpy(m[f"{base}_w"].copy())
    if c["bias"]:
        mod.bias.data = torch.from_numpy(m[f"{base}_b"].copy())
    return mod.cuda().train()


def test_conv_modules_vs_reference_golden(golden):
    m = golden.m
    worst = {}
    for c in golden.meta["modules"]:
        base, v = c["base"], c["variant"]
        mod = _build(c, v, m)
        x = m[f"{base}_xbin"] if v.startswith("wbwtab") else m[f"{base}_xreal"]
        tol = 1e-5
        for s in range(c["steps"]):
            for p in mod.parameters():
                p.grad = None
            xt = torch.from_numpy(x.copy()).cuda().requires_grad_(True)
            y = mod(xt)
            y.backward(torch.from_numpy(m[f"{base}_g"].copy()).cuda())
            pre = f"{base}_{v}_s{s}"
            errs = dict(y=rel_err(y.detach().cpu(), m[f"{pre}_y"]), dx=rel_err(xt.grad.cpu(), m[f"{pre}_dx"]),
                        dw=rel_err(mod.weight.grad.cpu(), m[f"{pre}_d_weight"]))
            if c["bias"] and "bnfuse" not in v:   # with BN folded in, d bias is mathematically 0 (round-off on both sides)
                errs["db"] = rel_err(mod.bias.grad.cpu(), m[f"{pre}_d_bias"])
            if "bnfuse" in v:
                errs["dgamma"] = rel_err(mod.gamma.grad.cpu(), m[f"{pre}_d_gamma"])
                errs["dbeta"] = rel_err(mod.beta.grad.cpu(), m[f"{pre}_d_beta"])
            for k_, e in errs.items():
                worst[(v, k_)] = max(worst.get((v, k_), 0.0), e)
                assert e <= tol, (pre, k_, e)
        if v.startswith("iao"):      # observer / qparam buffers: exact (bn-fuse weights: via batch statistics -> 1e-5)
            for name, buf in mod.named_buffers():
                ref = m[f"{base}_{v}_buf_{name}"]
                got = buf.detach().cpu().numpy()
                if "bnfuse" in v:
                    assert rel_err(got, ref) <= 2e-5, (base, v, name)
                else:
                    assert np.array_equal(got.reshape(-1), ref.reshape(-1)), (base, v, name)
        if v == "wbwtab_w2":          # the in-place mutated weight
            assert np.max(np.abs(mod.weight.detach().cpu().numpy() - m[f"{base}_{v}_par_weight"])) <= 2e-7
        if "bnfuse" in v:
            mod.eval()
            assert rel_err(mod(torch.from_numpy(x.copy()).cuda()).detach().cpu(), m[f"{base}_{v}_eval_y"]) <= tol
    print("worst rel errors:", {k: float("%.2e" % e) for k, e in sorted(worst.items())})


def test_linear_and_add_vs_reference_golden(golden):
    m = golden.m
    x, g = m["lin_x"], m["lin_g"]
    for vname, ctor, steps in (
        ("dorefa_w4a4", lambda: _q("wqaq.dorefa").QuantLinear(32, 10, a_bits=4, w_bits=4), 1),
        ("iao_w8a8_sym_fc", lambda: _q("wqaq.iao").QuantLinear(32, 10, a_bits=8, w_bits=8, q_type=0, q_level=0), 2),
    ):
        mod = ctor()
        mod.weight.data = torch.from_numpy(m["lin_w"].copy())
        mod.bias.data = torch.from_numpy(m["lin_b"].copy())
        mod = mod.cuda().train()
        for s in range(steps):
            for p in mod.parameters():
                p.grad = None
            xt = torch.from_numpy(x.copy()).cuda().requires_grad_(True)
            y = mod(xt)
            y.backward(torch.from_numpy(g.copy()).cuda())
            pre = f"lin_{vname}_s{s}"
            assert rel_err(y.detach().cpu(), m[f"{pre}_y"]) <= 1e-5
            assert rel_err(xt.grad.cpu(), m[f"{pre}_dx"]) <= 1e-5
            assert rel_err(mod.weight.grad.cpu(), m[f"{pre}_d_weight"]) <= 1e-5
            assert rel_err(mod.bias.grad.cpu(), m[f"{pre}_d_bias"]) <= 1e-5
    iao = _q("wqaq.iao")
    for q_type, bits in ((0, 4), (1, 8)):
        qa = iao.QuantAdd(a_bits=bits, q_type=q_type).cuda().train()
        key = f"qadd_t{q_type}b{bits}"
        for s in range(2):
            a = torch.from_numpy(m[f"{key}_s{s}_a"].copy()).cuda().requires_grad_(True)
            c = torch.from_numpy(m[f"{key}_s{s}_c"].copy()).cuda().requires_grad_(True)
            y = qa(a, c)
            y.backward(torch.from_numpy(m[f"{key}_s{s}_g"].copy()).cuda())
            # one fp32 add of two exactly-quantised values: bit-exact
            assert np.array_equal(y.detach().cpu().numpy(), m[f"{key}_s{s}_y"])
            assert np.array_equal(a.grad.cpu().numpy(), m[f"{key}_s{s}_da"])
            assert np.array_equal(c.grad.cpu().numpy(), m[f"{key}_s{s}_dc"])
            assert np.array_equal(qa.activation_quantizer.scale.cpu().numpy(), m[f"{key}_s{s}_scale"])
            assert np.array_equal(qa.activation_quantizer.zero_point.cpu().numpy(), m[f"{key}_s{s}_zp"])


def test_standalone_quantizers_keep_the_reference_call_pattern(golden):
    """scripts call ``m.weight_quantizer(m.weight)`` and read ``activation_quantizer.scale``."""
    q = golden.q
    d = _q("wqaq.dorefa")
    x = torch.from_numpy(q["dorefa_act4_x"].copy()).cuda().requires_grad_(True)
    y = d.ActivationQuantizer(a_bits=4)(x)
    y.backward(torch.from_numpy(q["dorefa_act4_g"].copy()).cuda())
    assert np.array_equal(y.detach().cpu().numpy(), q["dorefa_act4_y"]) and np.array_equal(x.grad.cpu().numpy(), q["dorefa_act4_dx"])
    assert np.array_equal(d.Round.apply(x.detach()).cpu().numpy(), np.sign(q["dorefa_act4_x"]) * np.floor(np.abs(q["dorefa_act4_x"]) + np.float32(0.5)))
    w = _q("wbwtab")
    xb = torch.from_numpy(q["binact_x"].copy()).cuda().requires_grad_(True)
    yb = w.ActivationQuantizer(A=2)(xb)
    yb.backward(torch.from_numpy(q["binact_g"].copy()).cuda())
    assert np.array_equal(yb.detach().cpu().numpy(), q["binact_y"]) and np.array_equal(xb.grad.cpu().numpy(), q["binact_dx"])
    t, thr = w.Ternary.apply(torch.from_numpy(q["ternary_w"].copy()).cuda())
    ok = ~np.isnan(q["ternary_y"])
    assert np.array_equal(t.cpu().numpy()[ok], np.sign(q["ternary_y"])[ok])
    assert d.ActivationQuantizer(a_bits=32)(x) is x


def _convt_golden():
    import json
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    return np.load(os.path.join(here, "convt.npz")), json.load(open(os.path.join(here, "convt_meta.json")))


@pytest.mark.parametrize("idx", range(5))
def test_conv_transpose_modules_vs_reference_golden(idx):
    """QuantConvTranspose2d of ALL THREE schemes (dorefa 126-174, wbwtab 198-244, iao 510-636) on the GPU against vectors produced by the reference's own classes
    (tests/golden/make_golden.py --convt-only): outputs / gradients <= 1e-5 of max|ref|, IAO observer ranges and scales bit-exact, the W = 2 in-place weight update
    to the last place of the Cin mean."""
    m, meta = _convt_golden()
    c = meta["cases"][idx]
    cin, cout, k, st, pd, op, H, W, Nb = c["shape"]
    base = "convt_" + c["name"]
    q = _q({"dorefa": "wqaq.dorefa", "wbwtab": "wbwtab", "iao": "wqaq.iao"}[c["scheme"]])
    mod = q.QuantConvTranspose2d(cin, cout, k, stride=st, padding=pd, output_padding=op, **c["kw"])
    mod.weight.data = torch.from_numpy(m[base + "_w"].copy())
    mod.bias.data = torch.from_numpy(m[base + "_b"].copy())
    mod = mod.cuda().train()
    for s in range(c["steps"]):
        for p_ in mod.parameters():
            p_.grad = None
        xt = torch.from_numpy(m[base + "_x"].copy()).cuda().requires_grad_(True)
        y = mod(xt)
        y.backward(torch.from_numpy(m[base + "_g"].copy()).cuda())
        pre = f"{base}_s{s}"
        assert rel_err(y.detach().cpu().numpy(), m[pre + "_y"]) <= 1e-5, pre
        assert rel_err(xt.grad.cpu().numpy(), m[pre + "_dx"]) <= 1e-5, pre
        assert rel_err(mod.weight.grad.cpu().numpy(), m[pre + "_d_weight"]) <= 1e-5, pre
        assert rel_err(mod.bias.grad.cpu().numpy(), m[pre + "_d_bias"]) <= 1e-5, pre
    if c["scheme"] == "iao":
        sd = {k_: v.detach().cpu().numpy() for k_, v in mod.state_dict().items()}
        for name in ("activation_quantizer.scale", "activation_quantizer.observer.min_val", "activation_quantizer.observer.max_val", "activation_quantizer.zero_point",
                     "weight_quantizer.scale", "weight_quantizer.observer.min_val", "weight_quantizer.observer.max_val"):
            assert sd[name].shape == m[f"{base}_buf_{name}"].shape, name
            assert np.array_equal(sd[name], m[f"{base}_buf_{name}"]), name
    if c["scheme"] == "wbwtab" and c["kw"]["W"] == 2:
        # the in-place mean-centre / clamp of the Parameter (wbwtab/quantize.py:98-102): the kernel sums the Cin axis in fp64 and rounds once, ATen adds fp32 partials --
        # the mean, and with it every stored weight, may differ in the last place
        got, ref = mod.weight.detach().cpu().numpy(), m[base + "_par_weight"]
        assert np.max(np.abs(got - ref)) <= 2.0 ** -22 * max(1.0, float(np.max(np.abs(ref))))


@pytest.mark.parametrize("scheme,cfg,okw", [
    ("wqaq.dorefa", dict(a_bits=2, w_bits=2), ("dorefa", dict(a_bits=2, w_bits=2))),
    ("wbwtab", dict(A=2, W=3), ("wbwtab", dict(A=2, W=3))),
    ("wqaq.iao", dict(a_bits=4, w_bits=4, q_type=1, q_level=0, weight_observer=1), ("iao", dict(a_bits=4, w_bits=4, q_type=1, q_level=0, weight_observer=1))),
    ("wqaq.iao", dict(a_bits=8, w_bits=8, bn_fuse=True, bn_fuse_calib=True), ("iao", dict(a_bits=8, w_bits=8, bn_fuse=True, bn_fuse_calib=True))),
])
def test_small_net_vs_torch_oracle(scheme, cfg, okw):
    """A 4-conv net with pools: product on the GPU vs the oracle on the CPU from the same init, one training step."""
    def net():
        torch.manual_seed(11)
        return nn.Sequential(nn.Conv2d(3, 16, 3, padding=1), nn.BatchNorm2d(16), nn.ReLU(inplace=True),
                             nn.Conv2d(16, 32, 3, padding=1, groups=2), nn.BatchNorm2d(32), nn.ReLU(inplace=True),
                             nn.MaxPool2d(2), nn.Conv2d(32, 32, 1, groups=4), nn.BatchNorm2d(32), nn.ReLU(inplace=True),
                             nn.Conv2d(32, 10, 1), nn.BatchNorm2d(10), nn.ReLU(inplace=True), nn.AvgPool2d(8), nn.Flatten())
    from micronet_amd.train import make_optimizer, synth_batch, train_step
    x, y = synth_batch(16)
    x = x[:, :, :16, :16].contiguous()
    prod = _q(scheme).prepare(net(), inplace=True, **cfg).cuda().train()
    orc = TO.prepare(net(), okw[0], inplace=True, **okw[1]).train()
    out_p = prod(x.cuda())
    out_o = orc(x)
    # free-running low-bit nets are chaotic: a BatchNorm output within 1e-7 of a rounding boundary flips an activation code
    # when the (mathematically equivalent) summation order of a conv changes, and the flip propagates.  The tight, per-layer
    # statement is test_gpu_models.py::test_layerwise_teacher_forced; here the nets must agree statistically.
    chaotic = scheme in ("wbwtab", "wqaq.dorefa")
    e_out = rel_err(out_p.detach().cpu(), out_o.detach())
    assert e_out <= (0.15 if chaotic else 2e-3), e_out
    lp = torch.nn.functional.cross_entropy(out_p, y.cuda())
    lo = torch.nn.functional.cross_entropy(out_o, y)
    lp.backward()
    lo.backward()
    assert abs(float(lp) - float(lo)) <= (2e-2 if chaotic else 1e-3)
    gp = dict(prod.named_parameters())
    gmax = max(float(p.grad.norm()) for p in orc.parameters() if p.grad is not None)
    for n_, p in orc.named_parameters():
        if p.grad is None or float(p.grad.norm()) < 1e-4 * gmax:
            continue                    # conv bias in front of BatchNorm: true gradient 0
        e = rel_err(gp[n_].grad.cpu(), p.grad)
        assert e <= (0.6 if chaotic else 0.25), (n_, e)       # free-running: a handful of activation-code flips perturb deep-layer gradients;
                                        # the tight per-layer statement is test_gpu_models.py::test_layerwise_teacher_forced


def test_wbwtab_fused_bn_binact_matches_unfused():
    """prepare(fuse_bn_act=True) (BatchNorm2dBinAct: one fused kernel) vs the plain BatchNorm2d + ActivationQuantizer modules:
    same binary activations except at BatchNorm outputs within rounding of zero, same gradients, same running statistics and
    state_dict keys."""
    w = _q("wbwtab")

    def net():
        torch.manual_seed(3)
        return nn.Sequential(nn.Conv2d(3, 32, 3, padding=1), nn.BatchNorm2d(32), nn.ReLU(inplace=True),
                             nn.Conv2d(32, 64, 3, padding=1, groups=2), nn.BatchNorm2d(64), nn.ReLU(inplace=True),
                             nn.Conv2d(64, 10, 1)).cuda().train()
    fused = w.prepare(net(), inplace=True, A=2, W=3)
    plain = w.prepare(net(), inplace=True, A=2, W=3, fuse_bn_act=False)
    assert type(fused[1]).__name__ == "BatchNorm2dBinAct" and type(plain[1]) is nn.BatchNorm2d
    assert list(fused.state_dict().keys()) == list(plain.state_dict().keys())
    x = torch.randn(8, 3, 16, 16, device="cuda")
    acts = {}
    fused[2].register_forward_hook(lambda m, i, o: acts.__setitem__("f", o.detach().clone()))
    plain[2].register_forward_hook(lambda m, i, o: acts.__setitem__("p", o.detach().clone()))
    yf, yp = fused(x), plain(x)
    flips = (acts["f"] != acts["p"]).float().mean().item()
    assert flips <= 1e-4, flips
    assert torch.all(acts["f"].abs() == 1)
    (yf.square().mean()).backward()
    (yp.square().mean()).backward()
    assert rel_err(fused[0].weight.grad.cpu(), plain[0].weight.grad.cpu()) <= (2e-2 if flips else 2e-5)
    assert rel_err(fused[1].weight.grad.cpu(), plain[1].weight.grad.cpu()) <= (2e-2 if flips else 2e-5)
    assert rel_err(fused[1].running_var.cpu(), plain[1].running_var.cpu()) <= 1e-5
    assert int(fused[1].num_batches_tracked) == 1
    fused.eval(), plain.eval()
    ef, ep = fused(x), plain(x)
    assert rel_err(ef.detach().cpu(), ep.detach().cpu()) <= 0.2


def test_wbwtab_folded_channel_shuffle_is_bit_identical():
    """prepare(fold_shuffle=True) moves ConvBNReLU's channel shuffle into the conv kernels' addressing: outputs and every
    gradient must be bit-identical to the materialised shuffle (same products, same summation order)."""
    from micronet_amd.models.nin_gc import ConvBNReLU
    w = _q("wbwtab")

    def net():
        torch.manual_seed(5)
        return nn.Sequential(ConvBNReLU(3, 16, 3, padding=1), ConvBNReLU(16, 32, 1, groups=2, channel_shuffle=1, shuffle_groups=2),
                             ConvBNReLU(32, 32, 3, padding=1, groups=2, channel_shuffle=1, shuffle_groups=4),
                             ConvBNReLU(32, 10, 1)).cuda().train()
    a = w.prepare(net(), inplace=True, A=2, W=3, fold_shuffle=True, fuse_conv_bn=False)
    b = w.prepare(net(), inplace=True, A=2, W=3, fold_shuffle=False, fuse_conv_bn=False)
    assert a[1].conv.in_shuffle_groups == 2 and a[1].channel_shuffle_flag == 0 and b[1].channel_shuffle_flag == 1
    x = torch.randn(8, 3, 16, 16, device="cuda")
    ya, yb = a(x), b(x)
    assert torch.equal(ya, yb)
    ya.square().mean().backward()
    yb.square().mean().backward()
    for (n_, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if n_.startswith("2.conv"):
            # the folded net hands the 3x3 conv packed codes (k_k3s_wgrad), the other one the float tensor torch's shuffle made
            # (k_kk_wgrad): same exact products, different summation order
            if n_.endswith("bias"):            # d bias in front of a BatchNorm is a sum that cancels to ~0: absolute tolerance
                assert (pa.grad - pb.grad).abs().max().item() <= 1e-4 * a[2].conv.weight.grad.abs().max().item(), n_
            else:
                assert rel_err(pa.grad.cpu(), pb.grad.cpu()) <= 1e-5, n_
        elif n_.startswith(("1.", "2.")):      # our kernels: deterministic, bit-identical
            assert torch.equal(pa.grad, pb.grad), n_
        else:                                  # plain nn.Conv2d layers run MIOpen's (atomic) backward-weight: equal to rounding
            assert rel_err(pa.grad.cpu(), pb.grad.cpu()) <= 1e-5, n_


def test_wbwtab_packed_activations_are_bit_identical():
    """prepare(packed_activations=True): the fused BN+sign hands its +-1 output to the next conv / max-pool as int8
    (SignTensor).  Same codes, same products, same summation order: logits, running statistics and every gradient must be
    bit-identical to the float32 hand-off; a foreign consumer (forward hook, the plain last conv) sees the float32 values."""
    from micronet_amd.models.nin_gc import ConvBNReLU
    from micronet_amd.sign_tensor import SignTensor
    w = _q("wbwtab")

    def net():
        torch.manual_seed(7)
        return nn.Sequential(ConvBNReLU(3, 32, 5, padding=2), ConvBNReLU(32, 32, 1, groups=2),
                             ConvBNReLU(32, 64, 1, groups=2, channel_shuffle=1, shuffle_groups=2), nn.MaxPool2d(2, 2),
                             ConvBNReLU(64, 64, 3, padding=1, groups=4, channel_shuffle=1, shuffle_groups=2),
                             ConvBNReLU(64, 64, 1, groups=2, channel_shuffle=1, shuffle_groups=4), nn.MaxPool2d(2, 2),
                             ConvBNReLU(64, 10, 1), nn.AvgPool2d(4)).cuda().train()
    a = w.prepare(net(), inplace=True, A=2, W=3, packed_activations=True, fuse_conv_bn=False)
    b = w.prepare(net(), inplace=True, A=2, W=3, packed_activations=False)
    assert type(a[3]).__name__ == "MaxPool2dSign" and type(b[3]) is nn.MaxPool2d
    # the classifier conv on sign codes (Conv2dSignIn) sums in another order than MIOpen: keep the stock layer on both sides so that
    # this test isolates the hand-off (Conv2dSignIn has its own test)
    assert type(a[7].conv).__name__ == "Conv2dSignIn"
    a[7].conv.__class__ = nn.Conv2d
    seen = {}
    a[1].register_forward_hook(lambda m, i, o: seen.__setitem__("a", (type(i[0]), type(o), o.detach().float().clone())))
    b[1].register_forward_hook(lambda m, i, o: seen.__setitem__("b", o.detach().clone()))
    x = torch.randn(8, 3, 16, 16, device="cuda")
    ya, yb = a(x), b(x)
    assert seen["a"][0] is SignTensor and seen["a"][1] is SignTensor           # the packed hand-off really happened
    assert torch.equal(seen["a"][2], seen["b"]) and torch.all(seen["b"].abs() == 1)
    assert type(ya) is torch.Tensor and torch.equal(ya, yb)
    ya.square().mean().backward()
    yb.square().mean().backward()
    for (n_, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if n_.startswith(("0.conv", "7.conv")):    # plain nn.Conv2d layers run MIOpen's (atomic) backward-weight: equal to rounding
            assert rel_err(pa.grad.cpu(), pb.grad.cpu()) <= 1e-5, n_
        elif n_ == "4.conv.weight":                # 3x3 on packed codes: k_k3s_wgrad; on the float hand-off: k_kk_wgrad -- same products, other order
            assert rel_err(pa.grad.cpu(), pb.grad.cpu()) <= 1e-5, n_
        elif n_ == "4.conv.bias":                  # a sum that cancels to ~0 in front of a BatchNorm: absolute tolerance
            assert (pa.grad - pb.grad).abs().max().item() <= 1e-4 * a[4].conv.weight.grad.abs().max().item(), n_
        else:
            assert torch.equal(pa.grad, pb.grad), n_
    for (n_, ba), (_, bb) in zip(a.named_buffers(), b.named_buffers()):
        assert torch.equal(ba, bb), n_
    a.eval(), b.eval()
    assert torch.equal(a(x), b(x))


def test_wbwtab_fused_conv_bn_matches_unfused():
    """prepare(fuse_conv_bn=True): a quantised pointwise conv between a SignTensor and a BatchNorm2dBinAct returns a LazyConvOut
    and the fused kernels (mn_qconv_bnsign_*) do conv + statistics + normalisation + sign without ever storing the conv
    output.  Same activations except at BatchNorm outputs within rounding of zero, same gradients / running statistics to
    rounding; a foreign consumer of the lazy output (a forward hook on the conv) sees the real convolution result."""
    from micronet_amd.models.nin_gc import ConvBNReLU
    from micronet_amd.sign_tensor import LazyConvOut, SignTensor
    w = _q("wbwtab")

    def net():
        torch.manual_seed(11)
        return nn.Sequential(ConvBNReLU(3, 64, 5, padding=2), ConvBNReLU(64, 64, 1, groups=2),
                             ConvBNReLU(64, 128, 1, groups=2, channel_shuffle=1, shuffle_groups=2), nn.MaxPool2d(2, 2),
                             ConvBNReLU(128, 128, 3, padding=1, groups=8, channel_shuffle=1, shuffle_groups=2),
                             ConvBNReLU(128, 128, 1, groups=4, channel_shuffle=1, shuffle_groups=8),
                             ConvBNReLU(128, 10, 1), nn.AvgPool2d(8)).cuda().train()
    a = w.prepare(net(), inplace=True, A=2, W=3)
    b = w.prepare(net(), inplace=True, A=2, W=3, fuse_conv_bn=False)
    assert a[1].conv.lazy_for_bn and a[2].conv.lazy_for_bn and a[4].conv.lazy_for_bn and a[0].conv.lazy_for_bn          # (block 0: the un-quantised first conv, ops.FirstConvLazy)
    assert not b[1].conv.lazy_for_bn and not b[0].conv.lazy_for_bn
    seen = {}

    def conv_hook(m, i, o):     # a foreign consumer: .clone() materialises the lazy output; reference = stock conv on the same input
        with torch.no_grad():
            xin = ops.channel_shuffle(i[0].to_float(), m.in_shuffle_groups)
            ref = torch.nn.functional.conv2d(xin, m.weight_quantizer(m.weight), m.bias, groups=m.groups)
        seen["lazy"] = (type(o), o.detach().clone(), ref)
    from micronet_amd import ops
    a[2].conv.register_forward_hook(conv_hook)
    # the first fused block sees bit-identical inputs on both sides (block 0: fused and unfused first block give identical codes, test_first_block_fused_vs_unfused)
    a[1].register_forward_hook(lambda m, i, o: seen.__setitem__("sa", o.detach().float().clone()))
    b[1].register_forward_hook(lambda m, i, o: seen.__setitem__("sb", o.detach().float().clone()))
    b[1].conv.register_forward_hook(lambda m, i, o: seen.__setitem__("yb", o.detach().double().clone()))
    x = torch.randn(16, 3, 16, 16, device="cuda")
    ya, yb = a(x), b(x)
    assert seen["lazy"][0] is LazyConvOut
    assert rel_err(seen["lazy"][1].cpu(), seen["lazy"][2].cpu()) <= 1e-6
    # a conv output takes only ~2K+1 distinct values per channel, so a BatchNorm output within rounding of zero is shared by
    # many elements: compare signs away from those ties (z from an fp64 evaluation of the unfused conv output)
    yc = seen["yb"]
    z = (yc - yc.mean(dim=(0, 2, 3), keepdim=True)) / torch.sqrt(yc.var(dim=(0, 2, 3), unbiased=False, keepdim=True) + 1e-5)
    away = z.abs() > 1e-6
    assert torch.equal(seen["sa"][away], seen["sb"][away])
    flips = (seen["sa"] != seen["sb"]).float().mean().item()
    print("sign flips at ties after the first fused block:", flips)
    ya.square().mean().backward()
    yb.square().mean().backward()
    assert all(torch.isfinite(p_.grad).all() for p_ in a.parameters() if p_.grad is not None)
    for (n_, ba), (_, bb) in zip(a.named_buffers(), b.named_buffers()):
        if n_.startswith(("0.", "1.")):          # blocks with bit-identical inputs on both sides
            assert rel_err(ba.float().cpu(), bb.float().cpu()) <= 1e-5, n_
    # gradients, teacher-forced on one block (a tie flips a whole level of a channel and then cascades through the following
    # binary blocks, but the backward of a block depends only on its input codes and the incoming gradient)
    for blk in (1, 2, 4):
        ba_, bb_ = a[blk], b[blk]
        cin = ba_.conv.in_channels
        codes = (torch.randint(0, 2, (8, cin, 8, 8), device="cuda", dtype=torch.int8) * 2 - 1)
        xa, xb = SignTensor(codes.clone()).requires_grad_(True), SignTensor(codes.clone()).requires_grad_(True)
        for m in (ba_, bb_):
            for p_ in m.parameters():
                p_.grad = None
        oa, ob = ba_(xa), bb_(xb)
        gout = torch.randn(oa.shape, device="cuda")
        oa.backward(gout)
        ob.backward(gout)
        assert rel_err(xa.grad.cpu(), xb.grad.cpu()) <= 2e-5, blk
        for (n_, pa), (_, pb) in zip(ba_.named_parameters(), bb_.named_parameters()):
            if n_ == "conv.bias":
                continue            # bias in front of a BatchNorm: the true gradient is zero, both sides hold round-off
            assert rel_err(pa.grad.cpu(), pb.grad.cpu()) <= 2e-5, (blk, n_, rel_err(pa.grad.cpu(), pb.grad.cpu()))
    a.eval(), b.eval()
    ea, eb = a(x), b(x)
    assert rel_err(ea.detach().cpu(), eb.detach().cpu()) <= 0.2


def test_first_conv_runs_on_our_kernels_and_matches_torch():
    """prepare() switches the un-quantised first conv to Conv2dFirst (conv_first.hip); output and gradients match nn.Conv2d."""
    from micronet_amd.nn import Conv2dFirst
    from micronet_amd import ops
    for mod in ("wbwtab", "wqaq.dorefa"):
        q = _q(mod)
        net = nn.Sequential(nn.Conv2d(3, 96, 5, padding=2), nn.BatchNorm2d(96), nn.ReLU(), nn.Conv2d(96, 32, 1), nn.BatchNorm2d(32), nn.ReLU(),
                            nn.Conv2d(32, 10, 1)).cuda()
        p = q.prepare(net, inplace=False)
        assert type(p[0]) is Conv2dFirst and isinstance(p[0], nn.Conv2d) and list(p.state_dict()) == list(net.state_dict())
    torch.manual_seed(0)
    ref = nn.Conv2d(3, 96, 5, padding=2).cuda()
    ours = Conv2dFirst(3, 96, 5, padding=2).cuda()
    ours.load_state_dict(ref.state_dict())
    x = torch.randn(16, 3, 32, 32, device="cuda")
    assert ops.first_conv_supported(x.shape, ours.weight.shape, ours.stride, ours.padding, ours.dilation, ours.groups)
    yr, yo = ref(x), ours(x)
    assert rel_err(yo.detach().cpu(), yr.detach().cpu()) <= 2e-6
    g = torch.randn_like(yr)
    yr.backward(g), yo.backward(g)
    assert rel_err(ours.weight.grad.cpu(), ref.weight.grad.cpu()) <= 1e-5
    assert rel_err(ours.bias.grad.cpu(), ref.bias.grad.cpu()) <= 1e-5
    # a geometry the kernels do not cover (stride 2) silently takes the stock path
    s2 = Conv2dFirst(3, 8, 3, stride=2, padding=1).cuda()
    assert s2(x).shape == (16, 8, 16, 16)


def test_wbwtab_pool_gradient_stays_pooled_and_matches():
    """The sign max-pool hands its input gradient on in pooled form (LazyPoolGrad); the fused block in front expands it inside
    its backward kernels.  Teacher-forced on block + pool: identical input codes and output gradient with the fusion on / off."""
    from micronet_amd import ops
    from micronet_amd.models.nin_gc import ConvBNReLU
    from micronet_amd.sign_tensor import SignTensor
    w = _q("wbwtab")
    torch.manual_seed(13)
    net = nn.Sequential(ConvBNReLU(3, 64, 3, padding=1), ConvBNReLU(64, 128, 1, groups=2, channel_shuffle=1, shuffle_groups=2), nn.MaxPool2d(2, 2),
                        ConvBNReLU(128, 10, 1), nn.AvgPool2d(8)).cuda().train()
    q = w.prepare(net, inplace=True, A=2, W=3)
    blk, pool = q[1], q[2]
    codes = (torch.randint(0, 2, (8, 64, 16, 16), device="cuda", dtype=torch.int8) * 2 - 1)
    gout = torch.randn(8, 128, 8, 8, device="cuda")
    res = {}
    for lazy in (True, False):
        ops.LAZY_POOL_GRAD = lazy
        try:
            for p_ in blk.parameters():
                p_.grad = None
            blk.bn.running_mean.zero_(); blk.bn.running_var.fill_(1.0)
            x = SignTensor(codes.clone()).requires_grad_(True)
            out = pool(blk(x))
            assert isinstance(out, SignTensor)
            out.backward(gout)
            res[lazy] = (x.grad.clone(), blk.conv.weight.grad.clone(), blk.bn.weight.grad.clone(), blk.bn.bias.grad.clone(), out.to_float())
        finally:
            ops.LAZY_POOL_GRAD = True
    assert torch.equal(res[True][4], res[False][4])
    for a_, b_, name in zip(res[True][:4], res[False][:4], ("dx", "dweight", "dgamma", "dbeta")):
        assert rel_err(a_.cpu(), b_.cpu()) <= 2e-6, (name, rel_err(a_.cpu(), b_.cpu()))
    # a foreign consumer of the lazy gradient (a tensor hook) sees the expanded gradient
    x = SignTensor(codes.clone()).requires_grad_(True)
    mid = blk(x)
    seen = {}
    mid.register_hook(lambda g_: seen.__setitem__("g", (type(g_).__name__, (g_ * 1.0).clone())))
    pool(mid).backward(gout)
    ref = torch.zeros(8, 128, 16, 16, device="cuda").requires_grad_(True)
    assert seen["g"][1].shape == (8, 128, 16, 16) and float(seen["g"][1].abs().sum()) > 0
    assert torch.allclose(seen["g"][1].sum(), gout.sum(), rtol=1e-4)
    # and the whole net still trains
    y = q(torch.randn(8, 3, 16, 16, device="cuda"))
    y.square().mean().backward()
    assert all(torch.isfinite(p_.grad).all() for p_ in q.parameters() if p_.grad is not None)


def test_wbwtab_last_conv_reads_sign_codes():
    """prepare() turns the un-quantised LAST conv into Conv2dSignIn: with a packed input it runs on the sign-code classifier kernels;
    output and all gradients match the stock nn.Conv2d applied to the unpacked float32 tensor."""
    from micronet_amd.nn import Conv2dSignIn
    from micronet_amd.sign_tensor import SignTensor
    w = _q("wbwtab")
    q = w.prepare(nn.Sequential(nn.Conv2d(3, 32, 3, padding=1), nn.BatchNorm2d(32), nn.ReLU(), nn.Conv2d(32, 64, 1), nn.BatchNorm2d(64), nn.ReLU(),
                                nn.Conv2d(64, 10, 1)).cuda(), inplace=True)
    assert type(q[6]) is Conv2dSignIn
    torch.manual_seed(3)
    ours, ref = Conv2dSignIn(256, 10, 1).cuda(), nn.Conv2d(256, 10, 1).cuda()
    ref.load_state_dict(ours.state_dict())
    codes = (torch.randint(0, 2, (8, 256, 8, 8), device="cuda", dtype=torch.int8) * 2 - 1)
    xs = SignTensor(codes.clone()).requires_grad_(True)
    xf = codes.float().requires_grad_(True)
    yo, yr = ours(xs), ref(xf)
    assert type(yo) is torch.Tensor and rel_err(yo.detach().cpu(), yr.detach().cpu()) <= 2e-6
    g = torch.randn_like(yr)
    yo.backward(g), yr.backward(g)
    assert rel_err(xs.grad.cpu(), xf.grad.cpu()) <= 2e-6
    assert rel_err(ours.weight.grad.cpu(), ref.weight.grad.cpu()) <= 1e-5 and rel_err(ours.bias.grad.cpu(), ref.bias.grad.cpu()) <= 1e-5
    assert rel_err(ours(xf.detach()).detach().cpu(), yr.detach().cpu()) <= 1e-5        # a float32 input takes the stock path


def test_first_block_bn_gradient_stays_lazy_and_matches():
    """Behind the first conv (no backward-data) the fused BatchNorm+sign backward only computes its sums and hands d loss / d y on as a
    LazyBNGrad; the first-layer backward-weight forms dy in registers.  Same expressions: gradients are bit-identical to the two-step path."""
    from micronet_amd import ops
    from micronet_amd.models.nin_gc import ConvBNReLU
    w = _q("wbwtab")
    torch.manual_seed(17)
    net = nn.Sequential(ConvBNReLU(3, 64, 5, padding=2), ConvBNReLU(64, 64, 1, groups=2), ConvBNReLU(64, 10, 1), nn.AvgPool2d(16)).cuda().train()
    q = w.prepare(net, inplace=True, A=2, W=3)
    x = torch.randn(8, 3, 16, 16, device="cuda")
    res = {}
    ops.FIRST_GRAM = False          # (the two-pass form of the lazy hand-over; the one-pass form on the image's Gram data: test_gpu_models.test_first_block_fused_vs_unfused)
    try:
        for lazy in (True, False):
            ops.LAZY_BN_GRAD = lazy
            try:
                q.zero_grad()
                torch.manual_seed(1)
                q(x).square().mean().backward()
                res[lazy] = [p_.grad.clone() for p_ in q.parameters()]
            finally:
                ops.LAZY_BN_GRAD = True
        for (n_, _), a_, b_ in zip(q.named_parameters(), res[True], res[False]):
            assert torch.equal(a_, b_), n_
        # a hook on the conv output's gradient is a foreign consumer: it sees the expanded dy
        seen = {}
        h = q[0].conv.register_full_backward_hook(lambda m, gi, go: seen.__setitem__("go", (go[0] * 1.0).abs().sum().item()))
        q.zero_grad()
        q(x).square().mean().backward()
        h.remove()
        assert seen["go"] > 0
        for a_, p_ in zip(res[True], q.parameters()):
            assert torch.allclose(a_, p_.grad, rtol=1e-4, atol=1e-7)
    finally:
        ops.FIRST_GRAM = True
    # the default path (fused first block, one-pass backward): the same foreign consumer gets dy from a recomputed conv output
    seen = {}
    h = q[0].conv.register_full_backward_hook(lambda m, gi, go: seen.__setitem__("go", (go[0] * 1.0).abs().sum().item()))
    q.zero_grad()
    q(x).square().mean().backward()
    h.remove()
    assert seen["go"] > 0 and all(torch.isfinite(p_.grad).all() for p_ in q.parameters())


def test_bn_backward_folded_into_conv_backward_matches():
    """ops.FOLD_BN_INTO_CONV_BWD: the fused block's BatchNorm+sign backward is not applied by a streaming pass but inside the conv's
    backward-data / backward-weight (dy formed from (da, h) in registers).  Teacher-forced on one block: same gradients to rounding."""
    from micronet_amd import ops
    from micronet_amd.models.nin_gc import ConvBNReLU
    from micronet_amd.sign_tensor import SignTensor
    w = _q("wbwtab")
    torch.manual_seed(19)
    net = nn.Sequential(ConvBNReLU(3, 128, 3, padding=1), ConvBNReLU(128, 256, 1, groups=2, channel_shuffle=1, shuffle_groups=2),
                        ConvBNReLU(256, 10, 1), nn.AvgPool2d(8)).cuda().train()
    q = w.prepare(net, inplace=True, A=2, W=3)
    blk = q[1]
    codes = (torch.randint(0, 2, (8, 128, 8, 8), device="cuda", dtype=torch.int8) * 2 - 1)
    gout = torch.randn(8, 256, 8, 8, device="cuda")
    res = {}
    old = ops.FOLD_BN_INTO_CONV_BWD
    for fold in (True, False):
        ops.FOLD_BN_INTO_CONV_BWD = fold
        try:
            for p_ in blk.parameters():
                p_.grad = None
            x = SignTensor(codes.clone()).requires_grad_(True)
            blk(x).backward(gout)
            res[fold] = [x.grad.clone()] + [p_.grad.clone() for n_, p_ in blk.named_parameters() if n_ != "conv.bias"]
        finally:
            ops.FOLD_BN_INTO_CONV_BWD = old
    for a_, b_ in zip(res[True], res[False]):
        assert rel_err(a_.cpu(), b_.cpu()) <= 1e-5


def test_dorefa_fused_bn_relu_matches_unfused():
    """prepare(fuse_bn_act=True) of the DoReFa scheme: BatchNorm2d + ReLU of a ConvBNReLU block as one fused op (BatchNorm2dReLU, the ReLU a
    no-op subclass) against the reference's module graph (MIOpen BatchNorm + ATen ReLU): logits, running statistics and every gradient."""
    from micronet_amd.models.nin_gc import ConvBNReLU
    w = _q("wqaq.dorefa")

    def net():
        torch.manual_seed(11)
        return nn.Sequential(ConvBNReLU(3, 32, 5, padding=2), ConvBNReLU(32, 32, 1, groups=2), nn.MaxPool2d(2, 2),
                             ConvBNReLU(32, 64, 3, padding=1, groups=2), ConvBNReLU(64, 10, 1), nn.AvgPool2d(8)).cuda().train()
    a = w.prepare(net(), inplace=True, a_bits=8, w_bits=8)
    b = w.prepare(net(), inplace=True, a_bits=8, w_bits=8, fuse_bn_act=False)
    assert type(a[0].bn).__name__ == "BatchNorm2dReLU" and isinstance(a[0].bn, nn.BatchNorm2d) and isinstance(a[0].relu, nn.ReLU)
    assert type(a[2]).__name__ == "MaxPool2dF32" and isinstance(a[2], nn.MaxPool2d) and type(b[2]) is nn.MaxPool2d
    assert type(b[0].bn) is nn.BatchNorm2d and type(b[0].relu) is nn.ReLU
    assert list(a.state_dict().keys()) == list(b.state_dict().keys())
    x = torch.randn(8, 3, 16, 16, device="cuda")
    ya, yb = a(x), b(x)
    assert rel_err(ya.detach().cpu(), yb.detach().cpu()) <= 2e-5
    ya.square().mean().backward()
    yb.square().mean().backward()
    scale = max(p_.grad.abs().max().item() for n_, p_ in b.named_parameters() if n_.endswith("conv.weight"))
    for (n_, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if n_.endswith("conv.bias"):           # a sum that cancels to ~0 in front of a BatchNorm: absolute tolerance
            assert (pa.grad - pb.grad).abs().max().item() <= 1e-4 * scale, n_
        else:
            assert rel_err(pa.grad.cpu(), pb.grad.cpu()) <= 2e-4, n_      # 8-bit activation codes may flip at a rounding boundary
    for (n_, ba), (_, bb) in zip(a.named_buffers(), b.named_buffers()):
        assert rel_err(ba.float().cpu(), bb.float().cpu()) <= 1e-5, n_
    a.eval(), b.eval()
    with torch.no_grad():
        assert rel_err(a(x).cpu(), b(x).cpu()) <= 2e-5


@pytest.mark.parametrize("bits,arch", [(2, "nin_gc"), (4, "nin_gc"), (3, "nin_gc"), (2, "nin"), (8, "nin_gc"), (6, "nin_gc")])
def test_dorefa_fused_blocks_match_unfused(bits, arch):
    """prepare(fuse_blocks=True) (activation codes in one byte, 16-bit integer conv stash, BatchNorm + ReLU + max-pool + next-layer quantizer in streaming
    kernels, shuffle folded) is numerically the SAME function as the unfused module graph: identical logits, gradients to float round-off, identical
    BatchNorm buffers -- in training and in eval mode, and under torch.no_grad()."""
    from micronet_amd.sign_tensor import QActTensor
    from micronet_amd.train import build_model, synth_batch
    Q = _q("wqaq.dorefa")
    x, y = synth_batch(16, device="cuda")
    # More than 4 bits (the wide kernels: 32-bit stash).  The batch statistics of the two paths differ in the last bit (block sums of acc^2 in fp32 there, exact
    # integers at <= 4 bits, fp32 sums of y on the unfused side), and at these widths that is enough to move a few activation codes across a rounding boundary
    # -- which the reference's OWN function amplifies to the per-cent level at initialisation: the CPU oracle of nin_gc at batch 16 answers a 1e-7 relative
    # per-channel perturbation of its BatchNorm outputs with a 5 % (6 bit) / 1.7 % (8 bit) change of the logits, and with none at 4 bits (measured, round 4).
    # So in TRAINING mode the two graphs are compared as a smoke test at those widths (same function up to such flips); the tight checks of the wide kernels are
    # the EVAL-mode comparison below (identical running statistics on both sides: identical codes), the kernel tests on the layer shapes
    # (test_qconv_bnq_hot_shapes_w8a8) and the batch-256 teacher-forced parity against the oracle (test_gpu_parity_full.py, c1 W8A8: every stage <= 1e-5).
    wide = bits > 4
    tl, tg, tk, tb = (0.3, 1.0, 1.0, 2e-2) if wide else (1e-6, 2e-5, 2e-4, 2e-6)
    a = Q.prepare(build_model(arch), inplace=True, a_bits=bits, w_bits=bits).cuda().train()
    b = Q.prepare(build_model(arch), inplace=True, a_bits=bits, w_bits=bits, fuse_blocks=False).cuda().train()
    assert list(a.state_dict().keys()) == list(b.state_dict().keys())
    kinds = []
    hooks = [m.register_forward_hook(lambda mod, i, o: kinds.append(type(o).__name__)) for m in a.model]
    oa, ob = a(x), b(x)
    for h in hooks:
        h.remove()
    if arch == "nin_gc":
        assert kinds.count("QActTensor") >= 8, kinds          # the blocks really run fused
    assert float((oa - ob).abs().max()) <= tl * float(ob.abs().max()), "logits"
    torch.nn.functional.cross_entropy(oa, y).backward()
    torch.nn.functional.cross_entropy(ob, y).backward()
    gmax = max(float(p.grad.abs().max()) for p in b.parameters())
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if n.endswith("conv.bias") and float(pb.grad.abs().max()) < 1e-4 * gmax:
            continue                                          # a conv bias in front of a BatchNorm: the true gradient is 0, both sides hold round-off
        d = (pa.grad - pb.grad).abs()
        scale = pb.grad.abs().max().clamp_min(1e-30)
        if n.endswith("conv.weight") and pa.dim() == 4:
            # the DoReFa weight quantizer routes the gradient of its global max |tanh w| to ONE element: a sum of ~all other terms of both signs
            # (ill-conditioned; the full-batch parity test judges it against fp64).  Everything else to 2e-5, that element to 2e-4.
            k = int(pb.detach().abs().argmax())
            assert float(d.flatten()[k] / scale) <= tk, (n, "arg-max element", float(d.flatten()[k] / scale))
            d = d.flatten().clone()
            d[k] = 0
        e = float(d.max() / scale)
        assert e <= tg, (n, e)
    bufs_b = dict(b.named_buffers())
    for (n, ba), (_, bb) in zip(a.named_buffers(), b.named_buffers()):
        if ba.dtype.is_floating_point:
            scale = float(bb.abs().max().clamp_min(1e-6))
            if n.endswith("running_mean"):          # a mean is accurate relative to the spread of the data, not to its own (possibly tiny) magnitude: the first block's
                scale = max(scale, float(bufs_b[n[:-len("running_mean")] + "running_var"].max().sqrt()))          # comes from the image's Gram data on the fused side
            assert float((ba - bb).abs().max()) <= tb * scale, n
        else:
            assert torch.equal(ba, bb), n
    b.load_state_dict(a.state_dict())
    a.eval(), b.eval()
    with torch.no_grad():
        ea, eb = a(x), b(x)
    assert float((ea - eb).abs().max()) <= (5e-6 if wide else 1e-6) * float(eb.abs().max())          # (same running statistics on both sides: no flips at any width)
    # a foreign consumer of a block's output sees the fp32 activation of the reference (hooks, feature taps)
    a.train()
    feats = []
    h = a.model[1].register_forward_hook(lambda mod, i, o: feats.append(o.float().mean().item() if isinstance(o, QActTensor) else None))
    a(x)
    h.remove()
    assert feats and feats[0] is not None and feats[0] >= 0
