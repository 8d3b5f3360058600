"""Inference-graph tooling (micronet_amd/inference.py): the deployed graph computes the same function as the trained QAT graph in eval
mode -- the check the reference's ``quant_model_test.py`` / ``bn_fused_model_test.py`` scripts make by comparing test accuracy, made here on the
tensors themselves."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu


def _trained(scheme, arch, kw, steps=2, wd=1e-5):
    from micronet_amd.train import build_model, make_optimizer, synth_batch, train_step
    Q = importlib.import_module("micronet.compression.quantization.%s.quantize" % scheme)
    m = Q.prepare(build_model(arch), inplace=True, **kw).cuda().train()
    opt = make_optimizer(m, 0.01, wd)
    x, y = synth_batch(32, device="cuda")
    for _ in range(steps):
        train_step(m, opt, x, y)
    return Q, m, x


@pytest.mark.parametrize("bits", [2, 8])
def test_dorefa_prequantized_inference_graph(bits):
    """quant_model_test.py:189-191: weights stored fake-quantised, quant_inference=True skips the weight quantizer: same eval output."""
    from micronet_amd import inference
    from micronet_amd.train import build_model
    Q, T, x = _trained("wqaq.dorefa", "nin_gc", dict(a_bits=bits, w_bits=bits))
    I = Q.prepare(build_model("nin_gc"), inplace=True, a_bits=bits, w_bits=bits, quant_inference=True).cuda()
    I.load_state_dict(T.state_dict())
    assert inference.prequantize_weights(I) == 8                       # every QuantConv2d (the first conv stays fp32, ref 206)
    T.eval(), I.eval()
    with torch.no_grad():
        a, b = T(x), I(x)
        # stage by stage on the SAME input: the inference graph contracts the stored fp32 weights (fp32-exact kernels), the training graph the integer
        # codes: equal to float round-off
        ta = x
        for mt, mi in (zip(T.model, I.model) if bits == 8 else ()):          # (the 2-bit training graph pools inside its fused blocks: stages do not align)
            ya, yb = mt(ta), mi(ta.float() if hasattr(ta, "float") else ta)
            assert float((ya.float() - yb.float()).abs().max()) <= 2e-6 * float(ya.float().abs().max().clamp_min(1e-30))
            ta = ya
    # free-running: at 8 bit a 5e-7 difference flips activation codes at rounding boundaries in the following layers (a code step is 1/255)
    assert float((a - b).abs().max()) <= (1e-6 if bits == 2 else 2e-2) * float(a.abs().max()), float((a - b).abs().max())
    n = 2 ** bits - 1
    for m in I.modules():
        if isinstance(m, Q.QuantConv2d):                               # stored weights are on the quantizer's grid (2k - n) / n
            k = (m.weight * n + n) / 2
            assert float((k - k.round()).abs().max()) <= 1e-4


@pytest.mark.parametrize("W", [3, 2])
def test_wbwtab_bn_fused_inference_graph(W):
    """wbwtab/bn_fuse/bn_fuse.py:20-107: BatchNorm folded into the conv; in front of a binary activation the fold keeps the weights ternary / binary.
    Binary nets amplify rounding at sign ties (a BatchNorm output within 1e-7 of zero), so the two graphs are compared on the logits with a small
    tolerance and on the predicted classes."""
    from micronet_amd import inference
    from micronet_amd.train import build_model
    Q, T, x = _trained("wbwtab", "nin_gc", dict(A=2, W=W), wd=0.0)
    I = Q.prepare(build_model("nin_gc"), inplace=True, A=2, W=W, quant_inference=True).cuda()
    I.load_state_dict(T.state_dict())
    inference.prequantize_weights(I)
    F = inference.wbwtab_model_bn_fuse(I, W=W)
    assert not any(isinstance(m, torch.nn.BatchNorm2d) for m in F.modules())
    # the folded convs in front of binary activations still hold ternary / binary codes x alpha: at most 3 distinct |values| per output channel
    qc = [m for m in F.modules() if isinstance(m, Q.QuantConv2d)]
    assert len(qc) == 7
    for m in qc:
        w = m.weight.detach().flatten(1)
        assert all(len(torch.unique(w[o].abs())) <= 2 for o in range(0, w.shape[0], 37))
    T.eval(), I.eval(), F.eval()
    with torch.no_grad():
        t, i, f = T(x), I(x), F(x)
    assert float((t - i).abs().max()) <= 1e-6 * float(t.abs().max())                                      # pre-quantisation alone is exact
    # ---- block by block on the SAME (teacher-forced) input: every stage of the folded graph gets the training graph's input of that stage; a binary block's
    # output must carry the same signs except where the folded pre-activation is a tie (|c| within 1e-5 of the channel's scale: there the fold's
    # y - mean + beta std / gamma and the BatchNorm's gamma (y - mean) / std + beta legitimately round to different sides); the last (real-valued) block
    # to 1e-5.  A sign error on a gamma < 0 channel or a lost channel shuffle of a replaced conv moves half of a block's outputs.
    from micronet_amd.sign_tensor import SignTensor
    tof = lambda v: v.to_float() if isinstance(v, SignTensor) else v.float()
    rec_in, rec_out, pre = {}, {}, {}
    hooks = []
    def recorder(k):
        def fn(mod, inp, out):          # (returns None: a hook's return value would replace the output)
            rec_in[k], rec_out[k] = inp[0], out
        return fn
    for k, m in enumerate(T.model):
        hooks.append(m.register_forward_hook(recorder(k)))
    with torch.no_grad():
        T(x)
    for h in hooks:
        h.remove()
    n_bin = 0
    for k, (mt, mf) in enumerate(zip(T.model, F.model)):
        act = getattr(mf, "relu", None)
        def pre_rec(mod, inp, k=k):
            pre[k] = tof(inp[0])
        hk = act.register_forward_pre_hook(pre_rec) if isinstance(act, torch.nn.Module) else None
        with torch.no_grad():
            of = tof(mf(rec_in[k]))
        if hk is not None:
            hk.remove()
        ot = tof(rec_out[k])
        binary = bool(((ot == 1) | (ot == -1)).all()) and k in pre
        if binary:
            n_bin += 1
            c = pre[k]
            tie = c.abs() <= 1e-5 * c.abs().amax(dim=(0, 2, 3), keepdim=True).clamp_min(1e-30)
            if of.shape != c.shape:          # (a pooled block: compare through the pool on the tie mask too)
                tie = torch.nn.functional.max_pool2d(tie.float(), 2, 2) > 0 if c.shape[2] == 2 * of.shape[2] else tie
            bad = (of != ot)
            assert float(bad.float().mean()) <= 1e-3 and bool(tie[bad].all()) if tie.shape == bad.shape else float(bad.float().mean()) <= 1e-4, \
                ("stage %d: folded block disagrees with BatchNorm + sign away from ties" % k, int(bad.sum()), float(bad.float().mean()))
        else:
            assert float((of - ot).abs().max()) <= 1e-5 * float(ot.abs().max().clamp_min(1e-30)), ("stage %d" % k, float((of - ot).abs().max()))
    assert n_bin >= 7
    agree = float((t.argmax(1) == f.argmax(1)).float().mean())
    rel = float((t - f).abs().max() / t.abs().max())
    print("W", W, "bn-fused vs train graph: max rel logit diff", rel, "class agreement", agree)
    assert agree >= 0.9 and rel <= 0.2


def test_iao_bn_fused_inference_graph():
    """wqaq/iao/bn_fuse/bn_fuse.py:20-80 + pre-quantisation: QuantBNFuseConv2d -> QuantConv2d(quant_inference=True) with folded, quantised weights."""
    from micronet_amd import inference
    Q, T, x = _trained("wqaq.iao", "nin_gc", dict(a_bits=8, w_bits=8, q_type=0, q_level=0, weight_observer=0, bn_fuse=True))
    T.eval()
    with torch.no_grad():
        t = T(x)
    I = inference.iao_model_bn_fuse(T)
    assert not any(isinstance(m, Q.QuantBNFuseConv2d) for m in I.modules())
    assert inference.prequantize_weights(I) == 9
    I.eval()
    with torch.no_grad():
        i = I(x)
        ta = x
        for mt, mi in zip(T.model, I.model):          # stage by stage on the same input: float round-off only
            ya, yb = mt(ta), mi(ta)
            assert float((ya - yb).abs().max()) <= 1e-5 * float(ya.abs().max().clamp_min(1e-30))
            ta = ya
    # free-running: 8-bit activation codes flip at rounding boundaries under a 5e-7 perturbation (a code step is 1/255 of the range)
    assert float((t - i).abs().max()) <= 2e-2 * float(t.abs().max()), float((t - i).abs().max() / t.abs().max())
    sd_keys = [k for k in I.state_dict() if "running" in k or "gamma" in k]
    assert not sd_keys


@pytest.mark.parametrize("bits", [2, 4])
def test_dorefa_resnet_inference_graph_on_int8_matrix_cores(bits):
    """SURVEY 8 (f3), second half: the deployed DoReFa ResNet (weights stored fake-quantised, quant_model_test.py:189-191) runs its convolutions on
    v_mfma_i32_16x16x64_i8 -- activation codes and weight codes as bytes, exact i32 accumulators (k_qd_fwd8) -- and computes the same function as the
    training graph in eval mode (bit for bit: same codes, same integer sums) and as the un-fused inference graph on the fp32-weight kernels."""
    import ctypes as C
    from micronet_amd import _lib, inference
    from micronet_amd.train import build_model
    Q, T, x = _trained("wqaq.dorefa", "resnet18", dict(a_bits=bits, w_bits=bits))
    I = Q.prepare(build_model("resnet18"), inplace=True, a_bits=bits, w_bits=bits, quant_inference=True).cuda()
    U = Q.prepare(build_model("resnet18"), inplace=True, a_bits=bits, w_bits=bits, quant_inference=True, fuse_blocks=False).cuda()
    I.load_state_dict(T.state_dict()), U.load_state_dict(T.state_dict())
    assert inference.prequantize_weights(I) == 20 and inference.prequantize_weights(U) == 20          # 16 block convs + 3 shortcuts + the classifier
    T.eval(), I.eval(), U.eval()
    lib = _lib.get_lib()
    with torch.no_grad():
        a = T(x)
        lib.mn_profile_enable(1)
        b = I(x)
        torch.cuda.synchronize()
        buf = (_lib.ProfEntry * 192)()
        n = lib.mn_profile_collect(buf, 192)
        lib.mn_profile_enable(0)
        u = U(x)
    names = [buf[i].name.decode() for i in range(n)]
    convs = [k for k in names if k.startswith("k_qd_") or k.startswith("k_kk") or k.startswith("k_conv")]
    assert convs and all(k.startswith("k_qd_fwd8<") for k in convs), names            # every quantised conv of the deployed graph: the int8 kernel
    assert torch.equal(a, b), float((a - b).abs().max())
    assert float((u - b).abs().max()) <= 1e-5 * float(u.abs().max()), float((u - b).abs().max())
    # weights NOT on the quantizer's grid (quant_inference without pre-quantisation): the reference convolves them as they are -- so do we (no code kernels)
    R = Q.prepare(build_model("resnet18"), inplace=True, a_bits=bits, w_bits=bits, quant_inference=True).cuda().eval()
    R.load_state_dict(T.state_dict())
    R2 = Q.prepare(build_model("resnet18"), inplace=True, a_bits=bits, w_bits=bits, quant_inference=True, fuse_blocks=False).cuda().eval()
    R2.load_state_dict(T.state_dict())
    with torch.no_grad():
        r, r2 = R(x), R2(x)
    assert float((r - r2).abs().max()) <= 1e-4 * float(r2.abs().max()), float((r - r2).abs().max())


# ---- the folded graphs against the REFERENCE's own (tests/golden/inference.npz: make_golden.py:gen_inference runs wbwtab/bn_fuse/bn_fuse.py:20-107 and
# wqaq/iao/bn_fuse/bn_fuse.py:20-80 on models the reference trained; tests/test_oracle_golden.py pins oracle/torch_oracle.py:bn_fuse_* to the same vectors)
def _inference_golden():
    import json
    import os
    import numpy as np
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    return np.load(os.path.join(here, "inference.npz")), json.load(open(os.path.join(here, "inference_meta.json")))


def _small_net(meta):
    from micronet_amd.models import nin_gc
    from micronet_amd.train import init_like_main
    torch.manual_seed(1)
    return init_like_main(nin_gc.Net(cfg=meta["cfg"]))


def _rel(a, ref):
    import numpy as np
    a = a.detach().double().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    return float(np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-30))


@pytest.mark.parametrize("W", [3, 2])
def test_wbwtab_bn_fused_graph_vs_reference_golden(W):
    """micronet_amd.inference.wbwtab_model_bn_fuse on the state the reference trained: the same module kinds per conv, every folded weight / bias equal to the
    reference's fold (<= 1e-6: the same few fp32 ops), and the folded graph run on the gfx950 kernels stage by stage ON THE REFERENCE'S OWN stage inputs --
    +-1 outputs equal except at sign ties of the reference's pre-activation, logits within float round-off of the last (fp32) conv."""
    import numpy as np
    from micronet_amd import inference
    from oracle import torch_oracle as TO
    from micronet_amd.train import synth_batch
    Q = importlib.import_module("micronet.compression.quantization.wbwtab.quantize")
    g, meta = _inference_golden()
    key = "inf_wbwtab_w%d" % W
    I = Q.prepare(_small_net(meta), inplace=True, A=2, W=W, quant_inference=True)
    I.load_state_dict({k[len(key) + 9:]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith(key + "_trained_")})
    F = inference.wbwtab_model_bn_fuse(I.cuda(), W=W).eval()
    convs = [(n_, m) for n_, m in F.named_modules() if isinstance(m, torch.nn.Conv2d)]
    assert [n_ for n_, _ in convs] == [n_ for n_, _ in meta["cases"][key]["convs"]]
    assert ["QuantConv2d" if isinstance(m, Q.QuantConv2d) else "Conv2d" for _, m in convs] == [t for _, t in meta["cases"][key]["convs"]]
    for n_, m in convs:
        assert _rel(m.weight, g[f"{key}_fused_{n_}_weight"]) <= 1e-6 and _rel(m.bias, g[f"{key}_fused_{n_}_bias"]) <= 1e-6, n_
    # the oracle's folded graph (pinned bit for bit to the reference's in tests/test_oracle_golden.py) supplies every stage's input / output on the CPU
    orc = TO.prepare(_small_net(meta), "wbwtab", inplace=True, A=2, W=W)
    orc.load_state_dict({k[len(key) + 9:]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith(key + "_trained_")})
    OF = TO.bn_fuse_wbwtab(orc, W).eval()
    x, _ = synth_batch(4)
    with torch.no_grad():
        t = x
        for i, (so, sp) in enumerate(zip(OF.model, F.model)):
            ref = so(t)
            got = sp(t.cuda())
            got = got.to_float() if hasattr(got, "to_float") else got
            if bool(((ref == 1) | (ref == -1)).all()):
                bad = (got.cpu() != ref)
                assert float(bad.float().mean()) <= 2e-4, (i, float(bad.float().mean()))         # sign ties of a pre-activation within round-off of 0
            else:
                assert _rel(got, ref.double().numpy()) <= 1e-5, (i, _rel(got, ref.double().numpy()))
            t = ref
        assert _rel(OF(x), g[f"{key}_fused_logits"].astype(np.float64)) <= 1e-6          # (bit-equal on the host that generated the goldens: tests/test_oracle_golden.py)
        lg = F(x.cuda())
        assert bool((lg.argmax(1).cpu() == torch.from_numpy(g[f"{key}_fused_logits"]).argmax(1)).float().mean() >= 0.75)
    # ---- the reference's DEPLOYMENT flow: weights stored pre-quantised (wbwtab/quant_model_test/quant_model_para.py: ``m.weight.data = weight_quantizer(m.weight)``),
    # THEN the fold.  The folded convs in front of a sign then hold codes x alpha (ref 36-55 only flips signs there), and the product's folded graph runs LOW-BIT:
    # +-1 activations as one byte end to end, integer weight codes on the matrix cores, conv + bias + sign in the fused kernels (no fp32 activation between two
    # quantised layers).  Checked against the oracle's folded graph stage by stage on the oracle's own stage inputs, handed over PACKED where they are +-1.
    from micronet_amd.sign_tensor import SignTensor, LazyConvOut
    orc2 = TO.prepare(_small_net(meta), "wbwtab", inplace=True, A=2, W=W)
    orc2.load_state_dict({k[len(key) + 9:]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith(key + "_trained_")})
    with torch.no_grad():
        for m in orc2.modules():
            if isinstance(m, TO.OConv2d) and m.scheme == "wbwtab":
                m.weight.data = TO.wbwtab_weight(m.weight, W).detach().clone()
    OF2 = TO.bn_fuse_wbwtab(orc2, W).eval()
    I2 = Q.prepare(_small_net(meta), inplace=True, A=2, W=W, quant_inference=True)
    I2.load_state_dict(orc2.state_dict())
    F2 = inference.wbwtab_model_bn_fuse(I2.cuda(), W=W).eval()
    qc2 = [m for m in F2.modules() if isinstance(m, Q.QuantConv2d)]
    assert len(qc2) == 7 and all(m.stored_codes for m in qc2)                       # every folded quantised conv kept codes x alpha
    assert sum(bool(m.lazy_for_bn) for m in qc2) >= 4                               # ... and most hand their un-computed result to the fused sign
    seen = []
    with torch.no_grad():
        t = x
        for i, (so, sp) in enumerate(zip(OF2.model, F2.model)):
            ref = so(t)
            pm1 = i > 0 and bool(((t == 1) | (t == -1)).all())
            inp = SignTensor(t.to(torch.int8).cuda().contiguous()) if pm1 else t.cuda()
            hooks = [m.register_forward_hook(lambda mod, i_, o_: seen.append(type(o_).__name__)) for m in sp.modules() if isinstance(m, Q.QuantConv2d)]
            got = sp(inp)
            for h_ in hooks:
                h_.remove()
            if bool(((ref == 1) | (ref == -1)).all()):
                assert isinstance(got, SignTensor), (i, type(got))                  # the deployed stage emits packed signs
                bad = (got.to_float().cpu() != ref)
                assert float(bad.float().mean()) <= 2e-4, (i, float(bad.float().mean()))
            else:
                got = got.to_float() if hasattr(got, "to_float") else got
                assert _rel(got, ref.double().numpy()) <= 1e-5, (i, _rel(got, ref.double().numpy()))
            t = ref
        assert seen.count("LazyConvOut") >= 4, seen                                 # conv + bias + sign really ran as ONE fused op on the codes
        lg2 = F2(x.cuda())
        assert bool((lg2.argmax(1).cpu() == OF2(x).argmax(1)).float().mean() >= 0.75)


def test_iao_bn_fused_graph_vs_reference_golden():
    """micronet_amd.inference.iao_model_bn_fuse on the state the reference trained: folded weights / biases and the copied quantizer scales equal the reference's,
    every stage of the folded graph on the reference's own stage input within 1e-5 (8-bit activation codes flip only at rounding ties: <= 1e-4 of the elements may
    differ by one step)."""
    import numpy as np
    from micronet_amd import inference
    from oracle import torch_oracle as TO
    from micronet_amd.train import synth_batch
    Q = importlib.import_module("micronet.compression.quantization.wqaq.iao.quantize")
    g, meta = _inference_golden()
    key = "inf_iao_w8a8"
    kw = dict(a_bits=8, w_bits=8, q_type=0, q_level=0)
    T = Q.prepare(_small_net(meta), inplace=True, bn_fuse=True, **kw)
    T.load_state_dict({k[len(key) + 9:]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith(key + "_trained_")})
    F = inference.iao_model_bn_fuse(T.cuda()).eval()
    convs = [(n_, m) for n_, m in F.named_modules() if isinstance(m, torch.nn.Conv2d)]
    assert [n_ for n_, _ in convs] == [n_ for n_, _ in meta["cases"][key]["convs"]] and all(type(m) is Q.QuantConv2d and m.quant_inference for _, m in convs)
    for n_, m in convs:
        assert _rel(m.weight, g[f"{key}_fused_{n_}_weight"]) <= 1e-6 and _rel(m.bias, g[f"{key}_fused_{n_}_bias"]) <= 1e-6, n_
        assert np.array_equal(m.activation_quantizer.scale.cpu().numpy().reshape(-1), g[f"{key}_fused_{n_}_ascale"].reshape(-1))
        assert np.array_equal(m.weight_quantizer.scale.cpu().numpy().reshape(-1), g[f"{key}_fused_{n_}_wscale"].reshape(-1))
    orc = TO.prepare(_small_net(meta), "iao", inplace=True, bn_fuse=True, **kw)
    sd = {k[len(key) + 9:]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith(key + "_trained_")}
    ren = lambda k: k.replace("activation_quantizer.", "aq.").replace("weight_quantizer.", "wq.").replace("quant_min_val", "qmin").replace("quant_max_val", "qmax")
    missing = orc.load_state_dict({ren(k): v for k, v in sd.items()}, strict=False)
    assert not [k for k in missing.missing_keys if not k.endswith(("qmin", "qmax"))], missing
    OF = TO.bn_fuse_iao(orc).eval()
    x, _ = synth_batch(4)
    with torch.no_grad():
        assert _rel(OF(x), g[f"{key}_fused_logits"].astype(np.float64)) <= 1e-6          # (bit-equal on the host that generated the goldens: tests/test_oracle_golden.py)
        t = x
        for i, (so, sp) in enumerate(zip(OF.model, F.model)):
            ref = so(t)
            got = sp(t.cuda()).cpu()
            d = (got - ref).abs() / ref.abs().max().clamp_min(1e-30)
            assert float((d > 1e-5).float().mean()) <= 1e-4 and float(d.max()) <= 2e-2, (i, float(d.max()), float((d > 1e-5).float().mean()))
            t = ref


def test_stored_codes_verdict_follows_the_weight_tensor():
    """`stored_codes` (the deployed wbwtab layer contracts integer codes) is a verdict about ONE state of the weight tensor.  Module moves keep it; new
    weights (`weight.data = ...`, load_state_dict) drop it, and the layer then convolves what is stored -- like the reference's quant_inference branch
    (wbwtab/quantize.py:181-185)."""
    from micronet_amd import inference
    Q = importlib.import_module("micronet.compression.quantization.wbwtab.quantize")
    torch.manual_seed(3)
    m = Q.QuantConv2d(16, 32, 1, W=3, quant_inference=True).cuda()
    inference.prequantize_weights(torch.nn.Sequential(m))
    assert m._codes_valid()
    m = m.cpu()
    assert m._codes_valid()                                     # Module._apply carries the verdict to the new tensor (same values)
    m = m.cuda()
    assert m._codes_valid()
    x = torch.sign(torch.randn(2, 16, 8, 8, device="cuda"))
    x[x == 0] = 1
    y0 = m(x)
    ref0 = torch.nn.functional.conv2d(x.cpu(), m.weight.detach().cpu(), m.bias.detach().cpu())
    assert float((y0.cpu() - ref0).abs().max()) <= 1e-5 * float(ref0.abs().max())
    w_new = torch.randn_like(m.weight)                          # NOT codes x alpha
    m.weight.data = w_new
    assert not m._codes_valid()
    y1 = m(x)
    ref1 = torch.nn.functional.conv2d(x.cpu(), w_new.cpu(), m.bias.detach().cpu())
    assert float((y1.cpu() - ref1).abs().max()) <= 1e-5 * float(ref1.abs().max())
    inference.mark_stored_codes(m)                              # (a wrong verdict would now decode sign x max|w| ...)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.load_state_dict(sd)                                       # ... and load_state_dict drops it again
    assert not m.stored_codes
