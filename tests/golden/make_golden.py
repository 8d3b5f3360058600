"""Generate golden vectors from the REAL reference (666DZY666/micronet) on CPU.

Run in the build container only (the reference is mounted read-only at
/root/reference and does not exist on the GPU box):

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports the reference's three ``quantize.py`` modules and model files
unmodified, drives them on seeded + adversarial inputs, and stores inputs and
outputs as small ``.npz`` fixtures next to this file.  The fixtures pin
``oracle/`` (tests/test_oracle_golden.py) and, on the GPU box, the HIP path
(tests/test_gpu_kernels.py, tests/test_gpu_modules.py, tests/test_gpu_models.py).  Nothing here is copied from the reference: it is
only *executed*.
"""
import json
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
REF = os.environ.get("MICRONET_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import numpy as np
import torch
import torch.nn as nn

import micronet.compression.quantization.wqaq.dorefa.quantize as ref_dorefa
import micronet.compression.quantization.wbwtab.quantize as ref_wbwtab
import micronet.compression.quantization.wqaq.iao.quantize as ref_iao
from micronet.models import nin_gc as ref_nin_gc, nin as ref_nin, resnet as ref_resnet

HERE = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)


def T(a):
    return torch.from_numpy(np.array(a, dtype=np.float32, copy=True))  # always copy: the reference mutates W in place


def N(t):
    return t.detach().cpu().numpy().copy()


def rng(seed):
    return np.random.default_rng(seed)


# ---------------------------------------------------------------- quantizers
def adversarial_act(bits):
    """values that land on rounding ties / clamp edges of the DoReFa act chain."""
    n = 2 ** bits - 1
    s = np.float32(1.0 / n)
    vals = [0.0, -0.0, 10.0, 10.000001, 9.999999, -1e-30, 1e-30, 1e-40, -5.0, 25.0, np.float32(0.49999997) * s * 10]
    for k in range(n + 1):
        for d in (-0.5, -0.25, 0.0, 0.25, 0.5):
            v = np.float32((k + d)) * s * np.float32(10.0)
            vals += [v, np.nextafter(np.float32(v), np.float32(100)), np.nextafter(np.float32(v), np.float32(-100))]
    return np.array(vals, dtype=np.float32)


def gen_dorefa(out):
    for bits in (2, 3, 4, 8):
        r = rng(100 + bits)
        x = np.concatenate([(r.standard_normal(600) * 6).astype(np.float32), adversarial_act(bits)])
        g = r.standard_normal(x.shape).astype(np.float32)
        xt = T(x).requires_grad_(True)
        q = ref_dorefa.ActivationQuantizer(a_bits=bits)
        y = q(xt)
        y.backward(T(g))
        out[f"dorefa_act{bits}_x"] = x
        out[f"dorefa_act{bits}_g"] = g
        out[f"dorefa_act{bits}_y"] = N(y)
        out[f"dorefa_act{bits}_dx"] = N(xt.grad)
    for bits in (2, 4, 8):
        r = rng(200 + bits)
        w = (r.standard_normal((8, 4, 3, 3)) * 0.4).astype(np.float32)
        if bits == 4:  # two-way tie for the global max |tanh|
            w[0, 0, 0, 0] = 1.7
            w[5, 2, 1, 1] = -1.7
        g = r.standard_normal(w.shape).astype(np.float32)
        wt = T(w).requires_grad_(True)
        q = ref_dorefa.WeightQuantizer(w_bits=bits)
        y = q(wt)
        y.backward(T(g))
        out[f"dorefa_w{bits}_w"] = w
        out[f"dorefa_w{bits}_g"] = g
        out[f"dorefa_w{bits}_tanh"] = N(torch.tanh(T(w)))
        out[f"dorefa_w{bits}_y"] = N(y)
        out[f"dorefa_w{bits}_dw"] = N(wt.grad)


def gen_wbwtab(out):
    r = rng(300)
    x = np.concatenate([(r.standard_normal(500) * 1.2).astype(np.float32),
                        np.array([0.0, -0.0, 1.0, -1.0, np.nextafter(np.float32(1), np.float32(0)),
                                  np.nextafter(np.float32(1), np.float32(2)), np.nextafter(np.float32(-1), np.float32(0)),
                                  np.nextafter(np.float32(-1), np.float32(-2)), 1e-38, -1e-38, 1e-45, 3.0, -3.0],
                                 dtype=np.float32)])
    g = r.standard_normal(x.shape).astype(np.float32)
    xt = T(x).requires_grad_(True)
    y = ref_wbwtab.ActivationQuantizer(A=2)(xt)
    y.backward(T(g))
    out["binact_x"], out["binact_g"], out["binact_y"], out["binact_dx"] = x, g, N(y), N(xt.grad)

    # ternary: random + a constant-magnitude channel + an all-zero channel (NaN alpha)
    w = (r.standard_normal((6, 4, 3, 3)) * 0.1).astype(np.float32)
    w[2] = np.where(r.standard_normal((4, 3, 3)) > 0, 0.25, -0.25).astype(np.float32)
    w[4] = 0.0
    g = r.standard_normal(w.shape).astype(np.float32)
    wt = T(w).requires_grad_(True)
    y = ref_wbwtab.WeightQuantizer(W=3)(wt)
    y.backward(T(g))
    out["ternary_w"], out["ternary_g"], out["ternary_y"], out["ternary_dw"] = w, g, N(y), N(wt.grad)

    w = (r.standard_normal((6, 4, 3, 3)) * 0.8).astype(np.float32)
    g = r.standard_normal(w.shape).astype(np.float32)
    p = nn.Parameter(T(w))
    y = ref_wbwtab.WeightQuantizer(W=2)(p)
    y.backward(T(g))
    out["binary_w"], out["binary_g"], out["binary_y"] = w, g, N(y)
    out["binary_w_after"], out["binary_dw"] = N(p.data), N(p.grad)


def gen_iao(out):
    """Three successive training-mode calls of each quantizer flavour."""
    cases = []
    for q_type in (0, 1):
        for bits in (4, 8):
            cases.append(("act", "L", "ema", q_type, bits, (3, 5, 4, 4)))
            cases.append(("w", "C", "minmax", q_type, bits, (6, 4, 3, 3)))
            cases.append(("w", "C", "ema", q_type, bits, (6, 4, 3, 3)))
            cases.append(("w", "L", "minmax", q_type, bits, (6, 4, 3, 3)))
            cases.append(("w", "FC", "minmax", q_type, bits, (7, 20)))
    meta = []
    for ci, (kind, level, obs, q_type, bits, shape) in enumerate(cases):
        r = rng(400 + ci)
        oc = shape[0] if level in ("C", "FC") else None
        observer = (ref_iao.MinMaxObserver if obs == "minmax" else ref_iao.MovingAverageMinMaxObserver)(
            q_level=level, out_channels=oc)
        cls = ref_iao.SymmetricQuantizer if q_type == 0 else ref_iao.AsymmetricQuantizer
        q = cls(bits=bits, observer=observer, activation_weight_flag=1 if kind == "act" else 0)
        q.train()
        key = f"iao{ci}"
        meta.append(dict(key=key, kind=kind, level=level, obs=obs, q_type=q_type, bits=bits, shape=list(shape)))
        for step in range(3):
            scale_in = [1.0, 2.5, 0.3][step]
            x = (r.standard_normal(shape) * scale_in + (0.4 if q_type == 1 else 0.0)).astype(np.float32)
            if step == 2 and kind == "w" and level == "C":
                x[1] = 0.0  # zero channel -> eps scale
            g = r.standard_normal(shape).astype(np.float32)
            xt = T(x).requires_grad_(True)
            y = q(xt)
            y.backward(T(g))
            out[f"{key}_s{step}_x"] = x
            out[f"{key}_s{step}_g"] = g
            out[f"{key}_s{step}_y"] = N(y)
            out[f"{key}_s{step}_dx"] = N(xt.grad)
            out[f"{key}_s{step}_min"] = N(q.observer.min_val)
            out[f"{key}_s{step}_max"] = N(q.observer.max_val)
            out[f"{key}_s{step}_scale"] = N(q.scale)
            out[f"{key}_s{step}_zp"] = N(q.zero_point)
        # eval-mode call (no observer update)
        q.eval()
        x = (r.standard_normal(shape) * 1.5).astype(np.float32)
        out[f"{key}_eval_x"] = x
        out[f"{key}_eval_y"] = N(q(T(x)))
    return meta


# ------------------------------------------------------------------- modules
CONV_CASES = [
    # name, cin, cout, k, stride, pad, dil, groups, bias, H, W, N
    ("g2_3x3", 8, 12, 3, 1, 1, 1, 2, True, 6, 6, 2),
    ("g4_1x1", 16, 16, 1, 1, 0, 1, 4, False, 4, 4, 3),
    ("s2_3x3", 6, 10, 3, 2, 1, 1, 1, False, 8, 8, 2),
    ("s2_1x1", 8, 16, 1, 2, 0, 1, 1, False, 8, 8, 2),
    ("d2_3x3", 4, 6, 3, 1, 2, 2, 1, True, 7, 7, 2),
    ("k5_first", 3, 16, 5, 1, 2, 1, 1, True, 8, 8, 2),
]


def run_module(mod, x, g, steps=1):
    res = {}
    for s in range(steps):
        for p in mod.parameters():
            p.grad = None
        xt = T(x).requires_grad_(True)
        y = mod(xt)
        y.backward(T(g))
        res[f"s{s}_y"] = N(y)
        res[f"s{s}_dx"] = N(xt.grad)
        for n_, p in mod.named_parameters():
            res[f"s{s}_d_{n_}"] = N(p.grad)
    for n_, b in mod.named_buffers():
        res[f"buf_{n_}"] = N(b)
    for n_, p in mod.named_parameters():
        res[f"par_{n_}"] = N(p.data)
    return res


def gen_modules(out):
    meta = []
    for ci, (name, cin, cout, k, st, pd, dl, gr, bias, H, W, Nb) in enumerate(CONV_CASES):
        r = rng(500 + ci)
        w = (r.standard_normal((cout, cin // gr, k, k)) * 0.3).astype(np.float32)
        b = (r.standard_normal(cout) * 0.1).astype(np.float32) if bias else None
        Ho = (H + 2 * pd - dl * (k - 1) - 1) // st + 1
        Wo = (W + 2 * pd - dl * (k - 1) - 1) // st + 1
        x_real = (r.standard_normal((Nb, cin, H, W)) * 4).astype(np.float32)
        x_bin = np.where(r.standard_normal((Nb, cin, H, W)) > 0, 1.0, -1.0).astype(np.float32)
        g = r.standard_normal((Nb, cout, Ho, Wo)).astype(np.float32)
        gamma = (r.random(cout) + 0.5).astype(np.float32)
        beta = (r.standard_normal(cout) * 0.1).astype(np.float32)
        base = f"conv_{name}"
        out[f"{base}_w"], out[f"{base}_xreal"], out[f"{base}_xbin"], out[f"{base}_g"] = w, x_real, x_bin, g
        out[f"{base}_gamma"], out[f"{base}_beta"] = gamma, beta
        if b is not None:
            out[f"{base}_b"] = b
        kw = dict(stride=st, padding=pd, dilation=dl, groups=gr, bias=bias)
        variants = {
            "dorefa_w2a2": (lambda: ref_dorefa.QuantConv2d(cin, cout, k, a_bits=2, w_bits=2, **kw), x_real, 1),
            "dorefa_w8a8": (lambda: ref_dorefa.QuantConv2d(cin, cout, k, a_bits=8, w_bits=8, **kw), x_real, 1),
            "wbwtab_w3": (lambda: ref_wbwtab.QuantConv2d(cin, cout, k, W=3, **kw), x_bin, 1),
            "wbwtab_w2": (lambda: ref_wbwtab.QuantConv2d(cin, cout, k, W=2, **kw), x_bin, 1),
            "iao_w8a8_sym_c": (lambda: ref_iao.QuantConv2d(cin, cout, k, a_bits=8, w_bits=8, q_type=0, q_level=0, **kw), x_real, 2),
            "iao_w4a4_sym_c": (lambda: ref_iao.QuantConv2d(cin, cout, k, a_bits=4, w_bits=4, q_type=0, q_level=0, **kw), x_real, 2),
            "iao_w8a8_asym_l": (lambda: ref_iao.QuantConv2d(cin, cout, k, a_bits=8, w_bits=8, q_type=1, q_level=1, **kw), x_real, 2),
            "iao_bnfuse_w8a8": (lambda: ref_iao.QuantBNFuseConv2d(cin, cout, k, a_bits=8, w_bits=8, q_type=0, q_level=0, **kw), x_real, 2),
        }
        for vname, (ctor, x, steps) in variants.items():
            torch.manual_seed(0)
            m = ctor()
            m.weight.data = T(w)
            if bias:
                m.bias.data = T(b)
            if "bnfuse" in vname:
                m.gamma.data = T(gamma)
                m.beta.data = T(beta)
            m.train()
            res = run_module(m, x, g, steps)
            if "bnfuse" in vname:  # eval forward with the running stats
                m.eval()
                res["eval_y"] = N(m(T(x)))
            for kk, vv in res.items():
                out[f"{base}_{vname}_{kk}"] = vv
            meta.append(dict(base=base, variant=vname, steps=steps, cin=cin, cout=cout, k=k, stride=st, padding=pd,
                             dilation=dl, groups=gr, bias=bias))
    # linear
    r = rng(600)
    w = (r.standard_normal((10, 32)) * 0.2).astype(np.float32)
    b = (r.standard_normal(10) * 0.1).astype(np.float32)
    x = (r.standard_normal((5, 32)) * 3).astype(np.float32)
    g = r.standard_normal((5, 10)).astype(np.float32)
    out["lin_w"], out["lin_b"], out["lin_x"], out["lin_g"] = w, b, x, g
    for vname, ctor, steps in (
        ("dorefa_w4a4", lambda: ref_dorefa.QuantLinear(32, 10, a_bits=4, w_bits=4), 1),
        ("iao_w8a8_sym_fc", lambda: ref_iao.QuantLinear(32, 10, a_bits=8, w_bits=8, q_type=0, q_level=0), 2),
    ):
        m = ctor()
        m.weight.data = T(w)
        m.bias.data = T(b)
        m.train()
        for kk, vv in run_module(m, x, g, steps).items():
            out[f"lin_{vname}_{kk}"] = vv
    # QuantAdd: two training calls, then eval
    for q_type, bits in ((0, 4), (1, 8)):
        r = rng(700 + q_type)
        qa = ref_iao.QuantAdd(a_bits=bits, q_type=q_type)
        qa.train()
        key = f"qadd_t{q_type}b{bits}"
        for s in range(2):
            a = (r.standard_normal((2, 4, 5, 5)) * (1 + s)).astype(np.float32)
            c = (r.standard_normal((2, 4, 5, 5)) * 0.5 + 0.3).astype(np.float32)
            g = r.standard_normal((2, 4, 5, 5)).astype(np.float32)
            at, ct = T(a).requires_grad_(True), T(c).requires_grad_(True)
            y = qa(at, ct)
            y.backward(T(g))
            out[f"{key}_s{s}_a"], out[f"{key}_s{s}_c"], out[f"{key}_s{s}_g"] = a, c, g
            out[f"{key}_s{s}_y"], out[f"{key}_s{s}_da"], out[f"{key}_s{s}_dc"] = N(y), N(at.grad), N(ct.grad)
            out[f"{key}_s{s}_scale"] = N(qa.activation_quantizer.scale)
            out[f"{key}_s{s}_zp"] = N(qa.activation_quantizer.zero_point)
    return meta


# --------------------------------------------------------------------- models
def init_like_main(model):
    """mirror of dorefa/main.py:289-297."""
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.Linear):
            nn.init.normal_(m.weight, 0, 0.01)
            if m.bias is not None:
                nn.init.zeros_(m.bias)


MODEL_CASES = {
    # key: (model ctor, prepare fn, batch, weight_decay)
    "c1_nin_gc_dorefa_w8a8": (lambda: ref_nin_gc.Net(), lambda m: ref_dorefa.prepare(m, inplace=True, a_bits=8, w_bits=8), 8, 1e-5),
    "c2_nin_gc_wbwtab_w3a2": (lambda: ref_nin_gc.Net(), lambda m: ref_wbwtab.prepare(m, inplace=True, A=2, W=3), 8, 0.0),
    "c2b_nin_gc_wbwtab_w2a2": (lambda: ref_nin_gc.Net(), lambda m: ref_wbwtab.prepare(m, inplace=True, A=2, W=2), 8, 0.0),
    "c3_nin_gc_iao_w8a8_bnfuse": (lambda: ref_nin_gc.Net(), lambda m: ref_iao.prepare(
        m, inplace=True, a_bits=8, w_bits=8, q_type=0, q_level=0, weight_observer=0, bn_fuse=True), 8, 1e-5),
    "c4_resnet18_dorefa_w2a2": (lambda: ref_resnet.resnet18(), lambda m: ref_dorefa.prepare(m, inplace=True, a_bits=2, w_bits=2), 4, 1e-5),
    "c5_resnet18_iao_w4a4": (lambda: ref_resnet.resnet18(), lambda m: ref_iao.prepare(
        m, inplace=True, a_bits=4, w_bits=4, q_type=0, q_level=0), 4, 1e-5),
    "nin_dorefa_w4a4": (lambda: ref_nin.Net(), lambda m: ref_dorefa.prepare(m, inplace=True, a_bits=4, w_bits=4), 4, 1e-5),
}


def synth_batch(B):
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, 3, 32, 32, generator=g)
    y = torch.randint(0, 10, (B,), generator=g)
    return x, y


def gen_models(out):
    surface = {}
    for key, (ctor, prep, B, wd) in MODEL_CASES.items():
        torch.manual_seed(1)
        model = ctor()
        init_like_main(model)
        prep(model)
        surface[key] = dict(
            modules=[(n_, type(m).__name__) for n_, m in model.named_modules()],
            state=[(k, list(v.shape)) for k, v in model.state_dict().items()],
        )
        params = [{"params": [v], "lr": 0.01, "weight_decay": wd} for _, v in model.named_parameters()]
        opt = torch.optim.Adam(params, lr=0.01, weight_decay=wd)
        crit = nn.CrossEntropyLoss()
        x, y = synth_batch(B)
        model.train()
        losses = []
        for step in range(3):
            o = model(x)
            loss = crit(o, y)
            opt.zero_grad()
            loss.backward()
            if step == 0:
                out[f"{key}_logits0"] = N(o)
                gn = {}
                for n_, p in model.named_parameters():
                    gn[n_] = float(p.grad.double().norm())
                surface[key]["gradnorm0"] = gn
                # one small gradient slice for a quantised mid layer
                names = [n_ for n_, p in model.named_parameters() if p.dim() == 4]
                mid = names[len(names) // 2]
                out[f"{key}_grad0_mid"] = N(dict(model.named_parameters())[mid].grad)[:8]
                surface[key]["mid"] = mid
            opt.step()
            losses.append(float(loss))
        model.eval()
        out[f"{key}_eval_logits"] = N(model(x))
        surface[key]["losses"] = losses
        print(key, losses)
    return surface


# ---------------------------------------------------------------- IAO: the rest of the module surface
def gen_iao_ops(out):
    """QuantReLU / LeakyReLU / Sigmoid / MaxPool2d / AvgPool2d / AdaptiveAvgPool2d (ref 1160-1438) in QAT (sym / asym), PTQ (HistogramObserver,
    ref 116-139) and QAFT mode; QuantBNFuseConv2d with bn_fuse_calib (ref 957-972), in QAFT mode and with pretrained_model; QuantConv2d in
    PTQ mode.  Two training steps (observer first-call + EMA), then an eval forward."""
    meta = []
    r = rng(900)
    x = (r.standard_normal((3, 8, 12, 12)) * 2.5).astype(np.float32)
    x2 = (r.standard_normal((3, 8, 12, 12)) * 4.0 + 0.5).astype(np.float32)
    out["ops_x0"], out["ops_x1"] = x, x2
    ops = {
        "relu": (lambda **kw: ref_iao.QuantReLU(inplace=False, **kw), (3, 8, 12, 12)),
        "leakyrelu": (lambda **kw: ref_iao.QuantLeakyReLU(negative_slope=0.1, inplace=False, **kw), (3, 8, 12, 12)),
        "sigmoid": (lambda **kw: ref_iao.QuantSigmoid(**kw), (3, 8, 12, 12)),
        "maxpool": (lambda **kw: ref_iao.QuantMaxPool2d(kernel_size=2, stride=2, padding=0, **kw), (3, 8, 6, 6)),
        "maxpool3": (lambda **kw: ref_iao.QuantMaxPool2d(kernel_size=3, stride=2, padding=1, **kw), (3, 8, 6, 6)),
        "avgpool": (lambda **kw: ref_iao.QuantAvgPool2d(kernel_size=2, stride=2, padding=0, **kw), (3, 8, 6, 6)),
        "avgpool4": (lambda **kw: ref_iao.QuantAvgPool2d(kernel_size=4, stride=4, padding=0, **kw), (3, 8, 3, 3)),
        "adaptiveavgpool": (lambda **kw: ref_iao.QuantAdaptiveAvgPool2d(output_size=(1, 1), **kw), (3, 8, 1, 1)),
    }
    modes = {"sym8": dict(a_bits=8, q_type=0), "asym8": dict(a_bits=8, q_type=1), "sym4": dict(a_bits=4, q_type=0),
             "ptq8": dict(a_bits=8, q_type=0, ptq=True, percentile=0.999), "qaft8": dict(a_bits=8, q_type=0, qaft=True)}
    for oname, (ctor, oshape) in ops.items():
        g = [r.standard_normal(oshape).astype(np.float32) for _ in range(2)]
        out[f"ops_{oname}_g0"], out[f"ops_{oname}_g1"] = g
        for mname, kw in modes.items():
            m = ctor(**kw)
            m.train()
            key = f"ops_{oname}_{mname}"
            for s_, (xi, gi) in enumerate(((x, g[0]), (x2, g[1]))):
                xt = T(xi).requires_grad_(True)
                y = m(xt)
                y.backward(T(gi))
                out[f"{key}_s{s_}_y"], out[f"{key}_s{s_}_dx"] = N(y), N(xt.grad)
                for n_, b in m.named_buffers():
                    out[f"{key}_s{s_}_buf_{n_}"] = N(b)
            m.eval()
            out[f"{key}_eval_y"] = N(m(T(x)))
            meta.append(dict(op=oname, mode=mname, kw=kw))
    # ---- BN-fuse conv variants + PTQ conv
    cin, cout, k = 8, 12, 3
    w = (r.standard_normal((cout, cin // 2, k, k)) * 0.3).astype(np.float32)
    gamma = (r.random(cout) + 0.5).astype(np.float32)
    beta = (r.standard_normal(cout) * 0.1).astype(np.float32)
    rm = (r.standard_normal(cout) * 0.2).astype(np.float32)
    rv = (r.random(cout) + 0.5).astype(np.float32)
    gy = [r.standard_normal((3, cout, 12, 12)).astype(np.float32) for _ in range(2)]
    out["bnf_w"], out["bnf_gamma"], out["bnf_beta"], out["bnf_rm"], out["bnf_rv"], out["bnf_g0"], out["bnf_g1"] = w, gamma, beta, rm, rv, gy[0], gy[1]
    variants = {
        "calib": dict(bn_fuse_calib=True),
        "qaft": dict(qaft=True),
        "pretrained": dict(pretrained_model=True),
        "calib_pretrained": dict(bn_fuse_calib=True, pretrained_model=True),
        "ptq": dict(ptq=True, percentile=0.999),
    }
    for vname, kw in variants.items():
        torch.manual_seed(0)
        m = ref_iao.QuantBNFuseConv2d(cin, cout, k, padding=1, groups=2, bias=False, a_bits=8, w_bits=8, q_type=0, q_level=0, **kw)
        m.weight.data, m.gamma.data, m.beta.data = T(w), T(gamma), T(beta)
        m.running_mean.copy_(T(rm)); m.running_var.copy_(T(rv))
        m.train()
        key = f"bnf_{vname}"
        for s_, (xi, gi) in enumerate(((x, gy[0]), (x2, gy[1]))):
            for p in m.parameters():
                p.grad = None
            xt = T(xi).requires_grad_(True)
            y = m(xt)
            y.backward(T(gi))
            out[f"{key}_s{s_}_y"], out[f"{key}_s{s_}_dx"] = N(y), N(xt.grad)
            for n_, p in m.named_parameters():
                if p.grad is not None:
                    out[f"{key}_s{s_}_d_{n_}"] = N(p.grad)
            for n_, b in m.named_buffers():
                out[f"{key}_s{s_}_buf_{n_}"] = N(b)
        m.eval()
        out[f"{key}_eval_y"] = N(m(T(x)))
        meta.append(dict(op="bnfuse", mode=vname, kw=kw))
    torch.manual_seed(0)
    m = ref_iao.QuantConv2d(cin, cout, k, padding=1, groups=2, bias=True, a_bits=8, w_bits=8, q_type=0, q_level=0, ptq=True, percentile=0.999)
    m.weight.data = T(w)
    b = (r.standard_normal(cout) * 0.1).astype(np.float32)
    out["ptqconv_b"] = b
    m.bias.data = T(b)
    m.train()
    for s_, (xi, gi) in enumerate(((x, gy[0]), (x2, gy[1]))):
        for p in m.parameters():
            p.grad = None
        xt = T(xi).requires_grad_(True)
        y = m(xt)
        y.backward(T(gi))
        out[f"ptqconv_s{s_}_y"], out[f"ptqconv_s{s_}_dx"], out[f"ptqconv_s{s_}_d_weight"] = N(y), N(xt.grad), N(m.weight.grad)
        for n_, bb in m.named_buffers():
            out[f"ptqconv_s{s_}_buf_{n_}"] = N(bb)
    # ---- HistogramObserver on its own: sizes around the k-th value index arithmetic, ties, first call + EMA
    for i, (n, pct) in enumerate(((1000, 0.9999), (4096, 0.999), (100000, 0.9999), (37, 0.9), (65536, 0.99999))):
        ho = ref_iao.HistogramObserver(q_level="L", percentile=pct)
        for s_ in range(2):
            v = (r.standard_normal(n) * (1 + s_)).astype(np.float32)
            if i == 1:
                v = np.round(v * 4) / 4          # many exact ties
            out[f"hist_{i}_s{s_}_x"] = v
            ho(T(v))
            out[f"hist_{i}_s{s_}_max"] = N(ho.max_val)
        meta.append(dict(op="hist", n=n, percentile=pct))
    return meta


# ---------------------------------------------------------------- inference graphs: the reference's own bn_fuse functions
SMALL_CFG = [32, 32, 32, 64, 64, 64, 128, 128]          # nin_gc at an eighth of the width (every group count of the net still divides): small fixtures


def _load_ref_script(rel_dir, name):
    """Load <REF>/micronet/compression/quantization/<rel_dir>/bn_fuse/bn_fuse.py the way it runs as a script: its `import quantize` / `from models import ...`
    resolve through the sys.path entries it appends (its parent directory, the package root)."""
    import importlib.util
    qdir = os.path.join(REF, "micronet", "compression", "quantization", *rel_dir.split("/"))
    saved_path, saved_mods = list(sys.path), {k: sys.modules.get(k) for k in ("quantize", "models")}
    sys.path[:0] = [qdir, os.path.join(REF, "micronet")]
    for k in ("quantize", "models"):
        sys.modules.pop(k, None)
    try:
        spec = importlib.util.spec_from_file_location("ref_%s_%s" % (rel_dir.replace("/", "_"), name), os.path.join(qdir, "bn_fuse", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.path[:] = saved_path
        for k, v in saved_mods.items():
            sys.modules.pop(k, None)
            if v is not None:
                sys.modules[k] = v
    return mod


def _train3(model, x, y, wd):
    params = [{"params": [v], "lr": 0.01, "weight_decay": wd} for _, v in model.named_parameters()]
    opt = torch.optim.Adam(params, lr=0.01, weight_decay=wd)
    crit = nn.CrossEntropyLoss()
    model.train()
    for _ in range(3):
        loss = crit(model(x), y)
        opt.zero_grad()
        loss.backward()
        opt.step()


def _stage_outputs(model, x):
    outs, t = [], x
    for st in model.model:
        t = st(t)
        outs.append(t)
    return outs


def gen_inference(out):
    """Folded inference graphs produced by the REFERENCE's functions -- wbwtab/bn_fuse/bn_fuse.py:20-107 (bn_fuse / bn_fuse_module / model_bn_fuse, with the two
    globals its __main__ sets: bin_bn_fuse_num = number of ActivationQuantizer modules, args.W) and wqaq/iao/bn_fuse/bn_fuse.py:20-80 -- from models trained
    for three steps with the reference.  Stored: the initial state_dict to start from, the trained state_dict, every folded conv's weight / bias, the eval-mode
    logits of the training graph and of the folded graph, and two intermediate stage outputs of the folded graph.
    The IAO script passes a `device=` keyword that this version of the reference's QuantConv2d does not accept (the script is stale); the generator strips that one
    keyword and runs the function otherwise unmodified."""
    import argparse
    import copy
    meta = {}
    x, y = synth_batch(4)
    for W in (3, 2):
        bnf = _load_ref_script("wbwtab", "bn_fuse")
        Q = bnf.quantize
        key = "inf_wbwtab_w%d" % W
        torch.manual_seed(1)
        base = ref_nin_gc.Net(cfg=SMALL_CFG)
        init_like_main(base)
        train = copy.deepcopy(base)
        Q.prepare(train, inplace=True, A=2, W=W)
        for k_, v in train.state_dict().items():
            out[f"{key}_init_{k_}"] = N(v)
        _train3(train, x, y, 0.0)
        inf = copy.deepcopy(base)
        Q.prepare(inf, inplace=True, A=2, W=W, quant_inference=True)
        inf.load_state_dict(train.state_dict())
        for k_, v in train.state_dict().items():
            out[f"{key}_trained_{k_}"] = N(v)
        bnf.args = argparse.Namespace(W=W, A=2)
        bnf.bn_counter = 0
        bnf.bin_bn_fuse_num = sum(isinstance(m, Q.ActivationQuantizer) for m in inf.modules())
        fused = bnf.model_bn_fuse(inf, inplace=False)
        train.eval(), fused.eval()
        with torch.no_grad():
            out[f"{key}_train_eval_logits"] = N(train(x))
            outs = _stage_outputs(fused, x)
            out[f"{key}_fused_logits"] = N(fused(x))
            out[f"{key}_fused_stage1"] = N(outs[1])
            out[f"{key}_fused_stage8"] = N(outs[8])
        convs = [(n_, m) for n_, m in fused.named_modules() if isinstance(m, nn.Conv2d)]
        for n_, m in convs:
            out[f"{key}_fused_{n_}_weight"], out[f"{key}_fused_{n_}_bias"] = N(m.weight), N(m.bias)
        meta[key] = dict(W=W, bin_bn_fuse_num=int(bnf.bin_bn_fuse_num), convs=[(n_, type(m).__name__) for n_, m in convs],
                         modules=[(n_, type(m).__name__) for n_, m in fused.named_modules()])
        print(key, "bin_bn_fuse_num", bnf.bin_bn_fuse_num, [t for _, t in meta[key]["convs"]])
    # IAO
    bnf = _load_ref_script("wqaq/iao", "bn_fuse")
    Q = bnf.quantize
    key = "inf_iao_w8a8"
    kw = dict(a_bits=8, w_bits=8, q_type=0, q_level=0)
    torch.manual_seed(1)
    base = ref_nin_gc.Net(cfg=SMALL_CFG)
    init_like_main(base)
    train = copy.deepcopy(base)
    Q.prepare(train, inplace=True, bn_fuse=1, **kw)
    for k_, v in train.state_dict().items():
        out[f"{key}_init_{k_}"] = N(v)
    _train3(train, x, y, 1e-5)
    for k_, v in train.state_dict().items():
        out[f"{key}_trained_{k_}"] = N(v)
    bnf.args = argparse.Namespace(**kw)
    bnf.device = "cpu"
    import types

    def ctor_without_device(*a, **k):
        k.pop("device", None)
        return Q.QuantConv2d(*a, **k)
    bnf.quantize = types.SimpleNamespace(QuantConv2d=ctor_without_device, QuantBNFuseConv2d=Q.QuantBNFuseConv2d)      # what bn_fuse.py reads from its `quantize`
    inf = copy.deepcopy(train)          # (the script loads the same state into a graph prepared with quant_inference=True; QuantBNFuseConv2d ignores that flag)
    fused = bnf.model_bn_fuse(inf, inplace=False)
    train.eval(), fused.eval()
    with torch.no_grad():
        out[f"{key}_train_eval_logits"] = N(train(x))
        outs = _stage_outputs(fused, x)
        out[f"{key}_fused_logits"] = N(fused(x))
        out[f"{key}_fused_stage1"] = N(outs[1])
        out[f"{key}_fused_stage8"] = N(outs[8])
    convs = [(n_, m) for n_, m in fused.named_modules() if isinstance(m, nn.Conv2d)]
    for n_, m in convs:
        out[f"{key}_fused_{n_}_weight"], out[f"{key}_fused_{n_}_bias"] = N(m.weight), N(m.bias)
        out[f"{key}_fused_{n_}_ascale"], out[f"{key}_fused_{n_}_wscale"] = N(m.activation_quantizer.scale), N(m.weight_quantizer.scale)
    meta[key] = dict(convs=[(n_, type(m).__name__) for n_, m in convs], modules=[(n_, type(m).__name__) for n_, m in fused.named_modules()], **kw)
    print(key, [t for _, t in meta[key]["convs"]])
    return meta


# ---------------------------------------------------------------- QuantConvTranspose2d of the three schemes (dorefa 126-174, wbwtab 198-244, iao 510-636)
CONVT_CASES = [
    # name, scheme, ctor kwargs (beyond the shape), steps
    ("dorefa_w4a4", "dorefa", dict(a_bits=4, w_bits=4), 1),
    ("wbwtab_w2", "wbwtab", dict(W=2), 1),
    ("wbwtab_w3", "wbwtab", dict(W=3), 1),
    ("iao_sym_w8a8", "iao", dict(a_bits=8, w_bits=8, q_type=0), 2),
    ("iao_asym_w4a4_ma", "iao", dict(a_bits=4, w_bits=4, q_type=1, weight_observer=1), 2),
]


def gen_convt(out):
    """groups = 1, dilation = 1, bias = True only: the reference's dorefa / wbwtab classes hand (dilation, groups, bias) to nn.ConvTranspose2d in the positions of
    (groups, bias, dilation) (dorefa/quantize.py:142-153, wbwtab/quantize.py:214-225), so every other combination builds a different layer than the one asked for."""
    meta = []
    for ci, (name, scheme, kw, steps) in enumerate(CONVT_CASES):
        r = rng(900 + ci)
        cin, cout, k, st, pd, op, H, W, Nb = 8, 6, 3, 2, 1, 1, 8, 8, 2
        w = (r.standard_normal((cin, cout, k, k)) * 0.3).astype(np.float32)
        b = (r.standard_normal(cout) * 0.1).astype(np.float32)
        x = (r.standard_normal((Nb, cin, H, W)) * 4).astype(np.float32)
        if scheme == "wbwtab":
            x = np.where(x > 0, 1.0, -1.0).astype(np.float32)            # the layer's input is a BinaryActivation output
        Ho, Wo = (H - 1) * st - 2 * pd + (k - 1) + op + 1, (W - 1) * st - 2 * pd + (k - 1) + op + 1
        g = r.standard_normal((Nb, cout, Ho, Wo)).astype(np.float32)
        ref = {"dorefa": ref_dorefa, "wbwtab": ref_wbwtab, "iao": ref_iao}[scheme]
        # dorefa / wbwtab: `bias=1` -- the swapped hand-over makes nn's dilation = the `bias` argument, and torch >= 2 rejects dilation (True, True)
        # ("must be tuple of ints, but found element of type bool"): with the default bias=True these two reference classes cannot run a forward here at all;
        # the integer 1 is the one value that is both a valid dilation and a true bias flag.  The reference itself is executed unmodified.
        extra = {} if scheme == "iao" else dict(bias=1)
        mod = ref.QuantConvTranspose2d(cin, cout, k, stride=st, padding=pd, output_padding=op, **extra, **kw)
        assert tuple(mod.weight.shape) == w.shape and mod.groups == 1 and tuple(mod.dilation) == (1, 1) and mod.bias is not None
        mod.weight.data = T(w)
        mod.bias.data = T(b)
        mod.train()
        base = f"convt_{name}"
        out[base + "_x"], out[base + "_w"], out[base + "_b"], out[base + "_g"] = x, w, b, g
        for key, val in run_module(mod, x, g, steps=steps).items():
            out[f"{base}_{key}"] = val
        meta.append(dict(name=name, scheme=scheme, kw=kw, steps=steps, shape=[cin, cout, k, st, pd, op, H, W, Nb]))
    return meta


def main():
    if "--inference-only" in sys.argv:      # regenerate only inference.npz (the older fixture files stay byte-identical)
        o = {}
        inf_meta = gen_inference(o)
        np.savez_compressed(os.path.join(HERE, "inference.npz"), **o)
        with open(os.path.join(HERE, "inference_meta.json"), "w") as f:
            json.dump(dict(cases=inf_meta, cfg=SMALL_CFG, torch=torch.__version__, reference_version="1.12.0"), f, indent=1)
        print("inference.npz", os.path.getsize(os.path.join(HERE, "inference.npz")) // 1024, "KiB")
        return
    if "--convt-only" in sys.argv:      # regenerate only convt.npz (round 5; the older fixture files stay byte-identical)
        o = {}
        meta = gen_convt(o)
        np.savez_compressed(os.path.join(HERE, "convt.npz"), **o)
        with open(os.path.join(HERE, "convt_meta.json"), "w") as f:
            json.dump(dict(cases=meta, torch=torch.__version__, reference_version="1.12.0"), f, indent=1)
        print("convt.npz", os.path.getsize(os.path.join(HERE, "convt.npz")) // 1024, "KiB")
        return
    if "--ops-only" in sys.argv:      # regenerate only iao_ops.npz (the three older fixture files stay byte-identical)
        o = {}
        ops_meta = gen_iao_ops(o)
        np.savez_compressed(os.path.join(HERE, "iao_ops.npz"), **o)
        with open(os.path.join(HERE, "iao_ops_meta.json"), "w") as f:
            json.dump(dict(ops=ops_meta, torch=torch.__version__, reference_version="1.12.0"), f, indent=1)
        print("iao_ops.npz", os.path.getsize(os.path.join(HERE, "iao_ops.npz")) // 1024, "KiB")
        return
    q = {}
    gen_dorefa(q)
    gen_wbwtab(q)
    iao_meta = gen_iao(q)
    np.savez_compressed(os.path.join(HERE, "quantizers.npz"), **q)
    m = {}
    mod_meta = gen_modules(m)
    np.savez_compressed(os.path.join(HERE, "modules.npz"), **m)
    mo = {}
    surface = gen_models(mo)
    np.savez_compressed(os.path.join(HERE, "models.npz"), **mo)
    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump(dict(iao=iao_meta, modules=mod_meta, surface=surface,
                       torch=torch.__version__, reference_version="1.12.0"), f, indent=1)
    for fn in ("quantizers.npz", "modules.npz", "models.npz", "meta.json"):
        print(fn, os.path.getsize(os.path.join(HERE, fn)) // 1024, "KiB")
    o = {}
    ops_meta = gen_iao_ops(o)
    np.savez_compressed(os.path.join(HERE, "iao_ops.npz"), **o)
    with open(os.path.join(HERE, "iao_ops_meta.json"), "w") as f:
        json.dump(dict(ops=ops_meta, torch=torch.__version__, reference_version="1.12.0"), f, indent=1)
    print("iao_ops.npz", os.path.getsize(os.path.join(HERE, "iao_ops.npz")) // 1024, "KiB")
    o = {}
    inf_meta = gen_inference(o)
    np.savez_compressed(os.path.join(HERE, "inference.npz"), **o)
    with open(os.path.join(HERE, "inference_meta.json"), "w") as f:
        json.dump(dict(cases=inf_meta, cfg=SMALL_CFG, torch=torch.__version__, reference_version="1.12.0"), f, indent=1)
    print("inference.npz", os.path.getsize(os.path.join(HERE, "inference.npz")) // 1024, "KiB")
    o = {}
    meta = gen_convt(o)
    np.savez_compressed(os.path.join(HERE, "convt.npz"), **o)
    with open(os.path.join(HERE, "convt_meta.json"), "w") as f:
        json.dump(dict(cases=meta, torch=torch.__version__, reference_version="1.12.0"), f, indent=1)
    print("convt.npz", os.path.getsize(os.path.join(HERE, "convt.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
