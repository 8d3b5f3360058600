"""The rest of the IAO module surface on the MI355X against vectors generated from the reference itself
(tests/golden/iao_ops.npz, make_golden.py: gen_iao_ops): Quant{ReLU, LeakyReLU, Sigmoid, MaxPool2d, AvgPool2d, AdaptiveAvgPool2d} in QAT
(sym / asym), PTQ (HistogramObserver) and QAFT mode, QuantBNFuseConv2d with bn_fuse_calib / qaft / pretrained_model / ptq, QuantConv2d in PTQ
mode, and the HistogramObserver's k-th value selection.

Tolerances: quantizer buffers (observer range, scale, zero point) and everything that is a pure function of quantised values through exact
ops (ReLU, LeakyReLU, max-pool) BIT-EXACT; sigmoid (device expf vs Sleef) and average pooling (summation order) <= 1e-6 rel; convolutions
<= 1e-5 rel (north_star)."""
import numpy as np
import pytest
import torch

import iao_ops_cases as IC

pytestmark = pytest.mark.gpu

EXACT_OPS = ("relu", "leakyrelu", "maxpool", "maxpool3")


def _product_op(op, kw):
    from micronet.compression.quantization.wqaq.iao import quantize as Q
    ctor = {"relu": lambda: Q.QuantReLU(inplace=False, **kw), "leakyrelu": lambda: Q.QuantLeakyReLU(negative_slope=0.1, inplace=False, **kw),
            "sigmoid": lambda: Q.QuantSigmoid(**kw), "maxpool": lambda: Q.QuantMaxPool2d(kernel_size=2, stride=2, padding=0, **kw),
            "maxpool3": lambda: Q.QuantMaxPool2d(kernel_size=3, stride=2, padding=1, **kw),
            "avgpool": lambda: Q.QuantAvgPool2d(kernel_size=2, stride=2, padding=0, **kw),
            "avgpool4": lambda: Q.QuantAvgPool2d(kernel_size=4, stride=4, padding=0, **kw),
            "adaptiveavgpool": lambda: Q.QuantAdaptiveAvgPool2d(output_size=(1, 1), **kw)}[op]
    return ctor().cuda()


def test_iao_ops_vs_reference_golden():
    g, meta = IC.load()
    worst = {}
    for c in meta["ops"]:
        if c["op"] in ("bnfuse", "hist"):
            continue
        m = _product_op(c["op"], c["kw"])
        errs = IC.run_op(m, g, c["op"], c["mode"], "cuda", exact=c["op"] in EXACT_OPS)
        for k_, e in errs.items():
            worst[c["op"]] = max(worst.get(c["op"], 0.0), e)
            assert e <= 1e-6, (c, k_, e)
        # buffers after the second training step: bit-exact
        key = f"ops_{c['op']}_{c['mode']}_s1_buf_"
        for n_, b in m.named_buffers():
            ref = g[key + n_]
            assert np.array_equal(b.detach().cpu().numpy().reshape(-1), ref.reshape(-1)), (c, n_, b, ref)
    print("worst rel err per op:", {k: float("%.2e" % v) for k, v in worst.items()})


def test_histogram_observer_kth_value_exact():
    """mn_hist_observe (3-pass radix select + EMA on the device) == torch.kthvalue of the reference, bit for bit; plus large random sizes vs
    torch.kthvalue on the same device."""
    from micronet.compression.quantization.wqaq.iao import quantize as Q
    g, meta = IC.load()
    hs = [c for c in meta["ops"] if c["op"] == "hist"]
    for i, c in enumerate(hs):
        ho = Q.HistogramObserver(q_level="L", percentile=c["percentile"]).cuda()
        for s_ in range(2):
            ho(torch.from_numpy(g[f"hist_{i}_s{s_}_x"].copy()).cuda())
            assert np.array_equal(ho.max_val.cpu().numpy(), g[f"hist_{i}_s{s_}_max"]), (i, s_, ho.max_val, g[f"hist_{i}_s{s_}_max"])
    gen = torch.Generator(device="cuda").manual_seed(3)
    for n, pct in ((1 << 24, 0.9999), (12345677, 0.5), (1 << 20, 1.0), (999, 0.001 + 1e-9)):
        x = torch.randn(n, device="cuda", generator=gen) * 3
        x[:7] = torch.tensor([0.0, -0.0, float("inf"), -1e-40, 1e-40, 65504.0, -3.0], device="cuda")
        k = max(1, int(pct * n))
        ho = Q.HistogramObserver(q_level="L", percentile=k / n + 1e-12 if int((k / n) * n) != k else k / n).cuda()
        if int(ho.percentile * n) != k:
            continue
        ho(x)
        ref = torch.kthvalue(x.abs().view(-1), k)[0]
        assert torch.equal(ho.max_val.view(()), ref), (n, pct, ho.max_val, ref)


@pytest.mark.parametrize("variant", ["calib", "qaft", "pretrained", "calib_pretrained", "ptq"])
def test_bnfuse_variants_vs_reference_golden(variant):
    from micronet.compression.quantization.wqaq.iao import quantize as Q
    g, meta = IC.load()
    kw = [c for c in meta["ops"] if c["op"] == "bnfuse" and c["mode"] == variant][0]["kw"]
    m = Q.QuantBNFuseConv2d(8, 12, 3, padding=1, groups=2, bias=False, a_bits=8, w_bits=8, q_type=0, q_level=0, **kw)
    m.weight.data, m.gamma.data, m.beta.data = (torch.from_numpy(g[k].copy()) for k in ("bnf_w", "bnf_gamma", "bnf_beta"))
    m.running_mean.copy_(torch.from_numpy(g["bnf_rm"])); m.running_var.copy_(torch.from_numpy(g["bnf_rv"]))
    m = m.cuda().train()
    rel = lambda a, b: float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))
    for s_ in range(2):
        for p in m.parameters():
            p.grad = None
        x = torch.from_numpy(g[f"ops_x{s_}"].copy()).cuda().requires_grad_(True)
        y = m(x)
        y.backward(torch.from_numpy(g[f"bnf_g{s_}"].copy()).cuda())
        key = f"bnf_{variant}_s{s_}"
        errs = dict(y=rel(y.detach().cpu().numpy(), g[f"{key}_y"]), dx=rel(x.grad.cpu().numpy(), g[f"{key}_dx"]),
                    dw=rel(m.weight.grad.cpu().numpy(), g[f"{key}_d_weight"]), dgamma=rel(m.gamma.grad.cpu().numpy(), g[f"{key}_d_gamma"]),
                    dbeta=rel(m.beta.grad.cpu().numpy(), g[f"{key}_d_beta"]),
                    rm=rel(m.running_mean.cpu().numpy(), g[f"{key}_buf_running_mean"]), rv=rel(m.running_var.cpu().numpy(), g[f"{key}_buf_running_var"]))
        print(variant, s_, {k: float("%.2e" % v) for k, v in errs.items()})
        for k_, e in errs.items():
            assert e <= 1e-5, (variant, s_, k_, e)
        for n_ in ("activation_quantizer.scale", "activation_quantizer.observer.min_val", "activation_quantizer.observer.max_val"):
            got = dict(m.named_buffers())[n_].detach().cpu().numpy()
            assert np.array_equal(got.reshape(-1), g[f"{key}_buf_{n_}"].reshape(-1)), (variant, s_, n_)
    m.eval()
    assert rel(m(torch.from_numpy(g["ops_x0"].copy()).cuda()).detach().cpu().numpy(), g[f"bnf_{variant}_eval_y"]) <= 1e-5


def test_ptq_conv_vs_reference_golden():
    from micronet.compression.quantization.wqaq.iao import quantize as Q
    g, _ = IC.load()
    m = Q.QuantConv2d(8, 12, 3, padding=1, groups=2, bias=True, a_bits=8, w_bits=8, q_type=0, q_level=0, ptq=True, percentile=0.999)
    m.weight.data, m.bias.data = torch.from_numpy(g["bnf_w"].copy()), torch.from_numpy(g["ptqconv_b"].copy())
    m = m.cuda().train()
    rel = lambda a, b: float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-30))
    for s_ in range(2):
        m.weight.grad = None
        x = torch.from_numpy(g[f"ops_x{s_}"].copy()).cuda().requires_grad_(True)
        y = m(x)
        y.backward(torch.from_numpy(g[f"bnf_g{s_}"].copy()).cuda())
        assert rel(y.detach().cpu().numpy(), g[f"ptqconv_s{s_}_y"]) <= 1e-5
        assert rel(x.grad.cpu().numpy(), g[f"ptqconv_s{s_}_dx"]) <= 1e-5
        assert rel(m.weight.grad.cpu().numpy(), g[f"ptqconv_s{s_}_d_weight"]) <= 1e-5
        for n_, b in m.named_buffers():
            assert np.array_equal(b.detach().cpu().numpy().reshape(-1), g[f"ptqconv_s{s_}_buf_{n_}"].reshape(-1)), (s_, n_)


def test_observer_from_producer_partials_equals_observer_on_tensor():
    """The fused BN+ReLU forward of an IAO ResNet leaves per-block (min, max) of its output (``_mn_minmax``); the observer + update_qparams of the conv that reads
    it then run from those partials (one launch, no pass over the activation).  Same ranges, scales and qparams as the ordinary observer, bit for bit; an in-place
    write into the tensor invalidates the hand-over (the observer reads the tensor again)."""
    from micronet.compression.quantization.wqaq.iao import quantize as Q
    from micronet_amd.quantization.wqaq.dorefa.quantize import BatchNorm2dReLU
    torch.manual_seed(3)
    bn = BatchNorm2dReLU(32).cuda().train()
    bn.emit_minmax = True
    y = torch.randn(9, 32, 8, 8, device="cuda") * 2
    qa = Q.SymmetricQuantizer(bits=4, observer=Q.MovingAverageMinMaxObserver(q_level="L", out_channels=None), activation_weight_flag=1).cuda().train()
    qb = Q.SymmetricQuantizer(bits=4, observer=Q.MovingAverageMinMaxObserver(q_level="L", out_channels=None), activation_weight_flag=1).cuda().train()
    for step in range(3):          # first call, then two moving-average updates
        a = bn(y + step)
        assert getattr(a, "_mn_minmax", None) is not None
        qp1 = qa.qparams(a)                                   # from the producer's partials
        qp2 = qb.qparams(a.detach().clone())                  # the ordinary observer on the tensor
        for u, v in ((qp1, qp2), (qa.observer.min_val, qb.observer.min_val), (qa.observer.max_val, qb.observer.max_val), (qa.scale, qb.scale)):
            assert torch.equal(u, v), step
    a = bn(y)
    a.add_(1.0)                                               # written in place: the partials are stale and must not be used
    qp1, qp2 = qa.qparams(a), qb.qparams(a.detach().clone())
    assert torch.equal(qp1, qp2) and torch.equal(qa.observer.max_val, qb.observer.max_val)
