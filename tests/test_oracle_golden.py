"""Pin oracle/ against the golden vectors produced by the real reference (tests/golden/make_golden.py).

Tolerances (stated once, used below):
  * every elementwise quantizer value, code, observer statistic and STE gradient: BIT-EXACT;
  * DoReFa weight path: bit-exact when fed torch-CPU's tanh (MKL VML in the reference environment) -- numpy's libm tanh differs in the last ulp,
    so with np.tanh at most a handful of codes may move at rounding boundaries;
  * float conv accumulate (numpy fp64 einsum vs the reference's MKLDNN fp32): |diff| <= 1e-5 * max|ref|.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import np_oracle as O
from oracle import torch_oracle as TO


def eq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("bits", [2, 3, 4, 8])
def test_dorefa_act(golden, bits):
    q = golden.q
    x, g = q[f"dorefa_act{bits}_x"], q[f"dorefa_act{bits}_g"]
    y, codes = O.dorefa_act_fwd(x, bits)
    assert eq(y, q[f"dorefa_act{bits}_y"])
    assert codes.min() >= 0 and codes.max() <= 2 ** bits - 1 and eq(codes, np.round(codes))
    assert eq(O.dorefa_act_bwd(g, x, bits), q[f"dorefa_act{bits}_dx"])


@pytest.mark.parametrize("bits", [2, 4, 8])
def test_dorefa_weight(golden, bits):
    q = golden.q
    w, g, th = q[f"dorefa_w{bits}_w"], q[f"dorefa_w{bits}_g"], q[f"dorefa_w{bits}_tanh"]
    y, k, t, M = O.dorefa_w_fwd(w, bits, tanh_w=th)
    assert eq(y, q[f"dorefa_w{bits}_y"])
    dw = O.dorefa_w_bwd(g, w, bits, tanh_w=th)
    ref = q[f"dorefa_w{bits}_dw"]
    # two autograd branches are summed in autograd's order: <= 1e-6 rel
    assert np.max(np.abs(dw - ref)) <= 1e-6 * np.max(np.abs(ref))
    # libm tanh instead of the reference host's: codes may differ only at rounding boundaries
    y2, k2, _, _ = O.dorefa_w_fwd(w, bits)
    assert (k2 != k).sum() <= 2


def test_wbwtab(golden):
    q = golden.q
    assert eq(O.binact_fwd(q["binact_x"]), q["binact_y"])
    assert eq(O.binact_bwd(q["binact_g"], q["binact_x"]), q["binact_dx"])
    out, t, alpha, thr, cnt = O.ternary_w_fwd(q["ternary_w"])
    ref = q["ternary_y"]
    assert np.isnan(out[4]).all() and np.isnan(ref[4]).all()          # all-zero channel -> 0/0
    ok = ~np.isnan(ref)
    assert eq(np.sign(out[ok]), np.sign(ref[ok]))                      # ternary CODES: bit-exact
    assert eq(t[ok], np.sign(ref[ok]))
    # alpha is an fp32 SUM over the channel: summation order (numpy pairwise vs ATen vectorised) moves the last ulp
    assert np.max(np.abs(out[ok] - ref[ok])) <= 5e-7 * np.max(np.abs(ref[ok]))
    dw, ref = O.ternary_w_bwd(q["ternary_g"], q["ternary_w"]), q["ternary_dw"]
    ok = ~np.isnan(ref)
    assert eq(np.isnan(dw), np.isnan(ref))
    assert np.max(np.abs(dw[ok] - ref[ok])) <= 1e-6 * np.max(np.abs(ref[ok]))
    out, w_new, b, alpha = O.binary_w_fwd(q["binary_w"])
    assert np.max(np.abs(w_new - q["binary_w_after"])) <= 2e-7      # mean over Cin is an fp32 sum (order-dependent ulp)
    assert eq(np.sign(out), np.sign(q["binary_y"]))                  # binary CODES: bit-exact
    assert np.max(np.abs(out - q["binary_y"])) <= 5e-7 * np.max(np.abs(out))
    dw, ref = O.binary_w_bwd(q["binary_g"], w_new), q["binary_dw"]
    assert np.max(np.abs(dw - ref)) <= 1e-6 * np.max(np.abs(ref))


def test_iao_quantizers(golden):
    q = golden.q
    for c in golden.meta["iao"]:
        key, bits, q_type, is_act = c["key"], c["bits"], c["q_type"], c["kind"] == "act"
        mn = mx = None
        for s in range(3):
            x, g = q[f"{key}_s{s}_x"], q[f"{key}_s{s}_g"]
            cmin, cmax = O.observe(x, c["level"])
            mn, mx = O.observer_update(c["obs"], s == 0, mn, mx, cmin, cmax)
            assert eq(mn, q[f"{key}_s{s}_min"]) and eq(mx, q[f"{key}_s{s}_max"]), key
            scale, zp = O.iao_qparams(mn, mx, bits, q_type, is_act)
            assert eq(scale, q[f"{key}_s{s}_scale"]) and eq(zp, q[f"{key}_s{s}_zp"]), key
            y, codes = O.iao_fq_fwd(x, scale, zp, bits, q_type, is_act)
            assert eq(y, q[f"{key}_s{s}_y"]), key
            dx = O.iao_fq_bwd(g, x, scale, zp, mn, mx, bits, q_type, is_act)
            assert eq(dx, q[f"{key}_s{s}_dx"]), key
        y, _ = O.iao_fq_fwd(q[f"{key}_eval_x"], scale, zp, bits, q_type, is_act)
        assert eq(y, q[f"{key}_eval_y"]), key


def test_conv_accumulate(golden):
    """numpy fp64 conv vs the reference's fp32 conv on the same quantised operands."""
    m = golden.m
    for c in golden.meta["modules"]:
        if c["variant"] != "wbwtab_w3":
            continue
        base = c["base"]
        x, w, g = m[f"{base}_xbin"], m[f"{base}_w"], m[f"{base}_g"]
        b = m[f"{base}_b"] if c["bias"] else None
        wq = O.ternary_w_fwd(w)[0]
        kw = dict(stride=c["stride"], padding=c["padding"], dilation=c["dilation"], groups=c["groups"])
        y = O.conv2d_fwd(x, wq, b, **kw)
        ref = m[f"{base}_wbwtab_w3_s0_y"]
        assert np.max(np.abs(y - ref)) <= 1e-5 * np.max(np.abs(ref)), base
        dx, dwq, db = O.conv2d_bwd(g, x, wq, **kw)
        ref = m[f"{base}_wbwtab_w3_s0_dx"]
        assert np.max(np.abs(dx - ref)) <= 1e-5 * np.max(np.abs(ref)), base
        dw = O.ternary_w_bwd(dwq.astype(np.float32), w)
        ref = m[f"{base}_wbwtab_w3_s0_d_weight"]
        assert np.max(np.abs(dw - ref)) <= 1e-5 * np.max(np.abs(ref)), base


# ------------------------------------------------------------------ torch oracle, module level (bit-exact on CPU)
def _mk(c, variant, m):
    import torch.nn as nn
    base = c["base"]
    src = nn.Conv2d(c["cin"], c["cout"], c["k"], c["stride"], c["padding"], c["dilation"], c["groups"], c["bias"])
    src.weight.data = torch.from_numpy(m[f"{base}_w"].copy())
    if c["bias"]:
        src.bias.data = torch.from_numpy(m[f"{base}_b"].copy())
    if variant.startswith("dorefa"):
        bits = int(variant[-1])
        return TO.OConv2d(src, "dorefa", a_bits=bits, w_bits=bits)
    if variant.startswith("wbwtab"):
        return TO.OConv2d(src, "wbwtab", W=int(variant[-1]))
    cfg = {"iao_w8a8_sym_c": dict(a_bits=8, w_bits=8, q_type=0, q_level=0),
           "iao_w4a4_sym_c": dict(a_bits=4, w_bits=4, q_type=0, q_level=0),
           "iao_w8a8_asym_l": dict(a_bits=8, w_bits=8, q_type=1, q_level=1),
           "iao_bnfuse_w8a8": dict(a_bits=8, w_bits=8, q_type=0, q_level=0)}[variant]
    if "bnfuse" in variant:
        bn = nn.BatchNorm2d(c["cout"])
        bn.weight.data = torch.from_numpy(m[f"{base}_gamma"].copy())
        bn.bias.data = torch.from_numpy(m[f"{base}_beta"].copy())
        return TO.OBNFuseConv2d(src, bn, **cfg)
    return TO.OConv2d(src, "iao", **cfg)


def test_torch_oracle_modules(golden):
    m = golden.m
    torch.set_num_threads(8)
    for c in golden.meta["modules"]:
        base, v = c["base"], c["variant"]
        mod = _mk(c, v, m).train()
        x = m[f"{base}_xbin"] if v.startswith("wbwtab") else m[f"{base}_xreal"]
        for s in range(c["steps"]):
            for p in mod.parameters():
                p.grad = None
            xt = torch.from_numpy(x.copy()).requires_grad_(True)
            y = mod(xt)
            y.backward(torch.from_numpy(m[f"{base}_g"].copy()))
            pre = f"{base}_{v}_s{s}"
            assert eq(y.detach().numpy(), m[f"{pre}_y"]), pre
            assert eq(xt.grad.numpy(), m[f"{pre}_dx"]), pre
            assert eq(mod.weight.grad.numpy(), m[f"{pre}_d_weight"]), pre
            if "bnfuse" in v:
                assert eq(mod.gamma.grad.numpy(), m[f"{pre}_d_gamma"]) and eq(mod.beta.grad.numpy(), m[f"{pre}_d_beta"])
        if "bnfuse" in v:
            mod.eval()
            assert eq(mod(torch.from_numpy(x.copy())).detach().numpy(), m[f"{base}_{v}_eval_y"])
            assert eq(mod.running_var.numpy(), m[f"{base}_{v}_buf_running_var"])


MODEL_CFG = {
    "c1_nin_gc_dorefa_w8a8": ("nin_gc", "dorefa", dict(a_bits=8, w_bits=8), 8, 1e-5),
    "c2_nin_gc_wbwtab_w3a2": ("nin_gc", "wbwtab", dict(A=2, W=3), 8, 0.0),
    "c2b_nin_gc_wbwtab_w2a2": ("nin_gc", "wbwtab", dict(A=2, W=2), 8, 0.0),
    "c3_nin_gc_iao_w8a8_bnfuse": ("nin_gc", "iao", dict(a_bits=8, w_bits=8, q_type=0, q_level=0, weight_observer=0, bn_fuse=True), 8, 1e-5),
    "c4_resnet18_dorefa_w2a2": ("resnet18", "dorefa", dict(a_bits=2, w_bits=2), 4, 1e-5),
    "c5_resnet18_iao_w4a4": ("resnet18", "iao", dict(a_bits=4, w_bits=4, q_type=0, q_level=0), 4, 1e-5),
    "nin_dorefa_w4a4": ("nin", "dorefa", dict(a_bits=4, w_bits=4), 4, 1e-5),
}


@pytest.mark.parametrize("key", list(MODEL_CFG))
def test_torch_oracle_models(golden, key):
    """3 Adam steps of the whole net: same losses / logits as the reference, bit for bit, on CPU."""
    from micronet_amd.train import build_model, synth_batch
    torch.set_num_threads(8)
    arch, scheme, cfg, B, wd = MODEL_CFG[key]
    model = build_model(arch)
    TO.prepare(model, scheme, inplace=True, **cfg)
    opt = TO.make_optimizer(model, 0.01, wd)
    x, y = synth_batch(B)
    model.train()
    losses = []
    for step in range(3):
        loss, out = TO.train_step(model, opt, x, y)
        if step == 0:
            assert eq(out.detach().numpy(), golden.mo[f"{key}_logits0"])
        losses.append(float(loss.detach()))
    assert losses == golden.meta["surface"][key]["losses"]
    model.eval()
    assert eq(model(x).detach().numpy(), golden.mo[f"{key}_eval_logits"])


# ------------------------------------------------------------------------------------------------ IAO: the rest of the surface (f2)
def _oracle_op(op, kw):
    import torch.nn as nn
    ctor = {"relu": lambda: nn.ReLU(), "leakyrelu": lambda: nn.LeakyReLU(0.1), "sigmoid": lambda: nn.Sigmoid(),
            "maxpool": lambda: nn.MaxPool2d(2, 2, 0), "maxpool3": lambda: nn.MaxPool2d(3, 2, 1), "avgpool": lambda: nn.AvgPool2d(2, 2, 0),
            "avgpool4": lambda: nn.AvgPool2d(4, 4, 0), "adaptiveavgpool": lambda: nn.AdaptiveAvgPool2d((1, 1))}[op]
    return TO.OQuantWrap(ctor(), kw["a_bits"], kw.get("q_type", 0), kw.get("qaft", False), kw.get("ptq", False), kw.get("percentile", 0.9999))


def test_iao_ops_oracle_vs_reference_golden():
    """Quant{ReLU,LeakyReLU,Sigmoid,MaxPool2d,AvgPool2d,AdaptiveAvgPool2d} in QAT / PTQ / QAFT mode: the oracle is bit-exact."""
    import iao_ops_cases as IC
    g, meta = IC.load()
    n = 0
    for c in meta["ops"]:
        if c["op"] in ("bnfuse", "hist"):
            continue
        IC.run_op(_oracle_op(c["op"], c["kw"]), g, c["op"], c["mode"], "cpu", exact=True)
        n += 1
    assert n == 40


def test_histogram_observer_oracle_vs_reference_golden():
    import iao_ops_cases as IC
    g, meta = IC.load()
    hs = [c for c in meta["ops"] if c["op"] == "hist"]
    for i, c in enumerate(hs):
        ho = TO.HistObserver(c["percentile"])
        for s_ in range(2):
            ho(torch.from_numpy(g[f"hist_{i}_s{s_}_x"].copy()))
            assert eq(ho.max_val.numpy(), g[f"hist_{i}_s{s_}_max"]), (i, s_)


@pytest.mark.parametrize("variant", ["calib", "qaft", "pretrained", "calib_pretrained", "ptq"])
def test_bnfuse_variants_oracle_vs_reference_golden(variant):
    """QuantBNFuseConv2d with bn_fuse_calib / qaft / pretrained_model / ptq (ref 837-994): same ATen ops in the same order -> bit-exact
    buffers, outputs and gradients to fp32 round-off of autograd's accumulation order."""
    import torch.nn as nn
    import iao_ops_cases as IC
    g, meta = IC.load()
    kw = [c for c in meta["ops"] if c["op"] == "bnfuse" and c["mode"] == variant][0]["kw"]
    conv = nn.Conv2d(8, 12, 3, padding=1, groups=2, bias=False)
    bn = nn.BatchNorm2d(12)
    conv.weight.data = torch.from_numpy(g["bnf_w"].copy())
    bn.weight.data, bn.bias.data = torch.from_numpy(g["bnf_gamma"].copy()), torch.from_numpy(g["bnf_beta"].copy())
    bn.running_mean.copy_(torch.from_numpy(g["bnf_rm"])); bn.running_var.copy_(torch.from_numpy(g["bnf_rv"]))
    m = TO.OBNFuseConv2d(conv, bn, a_bits=8, w_bits=8, q_type=0, q_level=0, **kw).train()
    for s_ in range(2):
        for p in m.parameters():
            p.grad = None
        x = torch.from_numpy(g[f"ops_x{s_}"].copy()).requires_grad_(True)
        y = m(x)
        y.backward(torch.from_numpy(g[f"bnf_g{s_}"].copy()))
        key = f"bnf_{variant}_s{s_}"
        assert eq(y.detach().numpy(), g[f"{key}_y"]), (variant, s_)
        assert eq(m.running_mean.numpy(), g[f"{key}_buf_running_mean"]) and eq(m.running_var.numpy(), g[f"{key}_buf_running_var"])
        for got, ref in ((x.grad, g[f"{key}_dx"]), (m.weight.grad, g[f"{key}_d_weight"]), (m.gamma.grad, g[f"{key}_d_gamma"]), (m.beta.grad, g[f"{key}_d_beta"])):
            assert np.max(np.abs(got.numpy() - ref)) <= 2e-6 * max(np.max(np.abs(ref)), 1e-30), (variant, s_)
    m.eval()
    assert eq(m(torch.from_numpy(g["ops_x0"].copy())).detach().numpy(), g[f"bnf_{variant}_eval_y"])


# ------------------------------------------------------------------------------------------------ inference graphs (f3): the oracle's bn_fuse restatements
INF_CASES = {"inf_wbwtab_w3": ("wbwtab", dict(A=2, W=3), 0.0), "inf_wbwtab_w2": ("wbwtab", dict(A=2, W=2), 0.0),
             "inf_iao_w8a8": ("iao", dict(a_bits=8, w_bits=8, q_type=0, q_level=0, bn_fuse=True), 1e-5)}


def _load_inference_golden():
    import json
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    return np.load(os.path.join(here, "inference.npz")), json.load(open(os.path.join(here, "inference_meta.json")))


@pytest.mark.parametrize("key", list(INF_CASES))
def test_bn_fused_inference_graphs_oracle_vs_reference_golden(key):
    """oracle/torch_oracle.py:bn_fuse_wbwtab / bn_fuse_iao against the graphs the REFERENCE's own bn_fuse functions produce (wbwtab/bn_fuse/bn_fuse.py:20-107,
    wqaq/iao/bn_fuse/bn_fuse.py:20-80; tests/golden/make_golden.py:gen_inference): train the oracle net three steps (bit-identical to the reference's training, so
    the trained state must equal the golden one), fold, and compare every folded weight / bias, two intermediate stage outputs and the logits bit for bit."""
    from micronet_amd.models import nin_gc
    from micronet_amd.train import init_like_main, synth_batch
    g, meta = _load_inference_golden()
    scheme, kw, wd = INF_CASES[key]
    torch.set_num_threads(8)
    torch.manual_seed(1)
    model = init_like_main(nin_gc.Net(cfg=meta["cfg"]))
    TO.prepare(model, scheme, inplace=True, **kw)
    opt = TO.make_optimizer(model, 0.01, wd)
    x, y = synth_batch(4)
    model.train()
    for _ in range(3):
        TO.train_step(model, opt, x, y)
    for n_, p in model.named_parameters():
        assert eq(p.detach().numpy(), g[f"{key}_trained_{n_}"]), n_
    fused = (TO.bn_fuse_wbwtab(model, kw["W"]) if scheme == "wbwtab" else TO.bn_fuse_iao(model)).eval()
    convs = [(n_, m) for n_, m in fused.named_modules() if isinstance(m, torch.nn.Conv2d)]
    assert [n_ for n_, _ in convs] == [n_ for n_, _ in meta["cases"][key]["convs"]]
    kinds = ["Conv2d" if type(m) is torch.nn.Conv2d else "QuantConv2d" for _, m in convs]
    assert kinds == [t for _, t in meta["cases"][key]["convs"]]
    for n_, m in convs:
        assert eq(m.weight.detach().numpy(), g[f"{key}_fused_{n_}_weight"]) and eq(m.bias.detach().numpy(), g[f"{key}_fused_{n_}_bias"]), n_
    with torch.no_grad():
        model.eval()
        assert eq(model(x).numpy(), g[f"{key}_train_eval_logits"])
        t, outs = x, []
        for st in fused.model:
            t = st(t)
            outs.append(t)
        assert eq(outs[1].numpy(), g[f"{key}_fused_stage1"]) and eq(outs[8].numpy(), g[f"{key}_fused_stage8"])
        assert eq(fused(x).numpy(), g[f"{key}_fused_logits"])


def _convt_golden():
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    return np.load(os.path.join(here, "convt.npz")), json.load(open(os.path.join(here, "convt_meta.json")))


@pytest.mark.parametrize("idx", range(5))
def test_conv_transpose_oracle_vs_reference_golden(idx):
    """QuantConvTranspose2d of the three schemes (dorefa 126-174, wbwtab 198-244, iao 510-636): TO.OConvTranspose2d reproduces the reference's outputs, gradients and
    observer buffers on the reference's own inputs (tests/golden/make_golden.py --convt-only)."""
    m, meta = _convt_golden()
    c = meta["cases"][idx]
    cin, cout, k, st, pd, op, H, W, Nb = c["shape"]
    base = "convt_" + c["name"]
    torch.set_num_threads(8)
    src = torch.nn.ConvTranspose2d(cin, cout, k, st, pd, op)
    src.weight.data = torch.from_numpy(m[base + "_w"].copy())
    src.bias.data = torch.from_numpy(m[base + "_b"].copy())
    mod = TO.OConvTranspose2d(src, c["scheme"], **c["kw"]).train()
    for s in range(c["steps"]):
        for p_ in mod.parameters():
            p_.grad = None
        xt = torch.from_numpy(m[base + "_x"].copy()).requires_grad_(True)
        y = mod(xt)
        y.backward(torch.from_numpy(m[base + "_g"].copy()))
        pre = f"{base}_s{s}"
        assert eq(y.detach().numpy(), m[pre + "_y"]), pre
        assert eq(xt.grad.numpy(), m[pre + "_dx"]), pre
        assert eq(mod.weight.grad.numpy(), m[pre + "_d_weight"]), pre
        assert eq(mod.bias.grad.numpy(), m[pre + "_d_bias"]), pre
    if c["scheme"] == "iao":
        for q_, o_ in (("activation_quantizer", mod.aq), ("weight_quantizer", mod.wq)):
            assert eq(o_.scale.numpy().reshape(-1), m[f"{base}_buf_{q_}.scale"].reshape(-1)), q_
            assert eq(o_.observer.min_val.numpy().reshape(-1), m[f"{base}_buf_{q_}.observer.min_val"].reshape(-1)), q_
    if c["scheme"] == "wbwtab" and c["kw"]["W"] == 2:
        assert eq(mod.weight.detach().numpy(), m[base + "_par_weight"])          # mean-centred and clamped in place (wbwtab/quantize.py:98-102)
