"""Checks of the BN-fused IAO block kernels (micronet_amd/csrc/iao_bnfuse.hip) through the C ABI, written once for the CPU emulation build and the gfx950
build: the pipeline gram -> prep_fwd -> conv(+ReLU, min/max) -> bwd_weight -> prep_bwd -> bwd_data against the torch-CPU oracle of the reference's
QuantBNFuseConv2d (oracle/torch_oracle.py:OBNFuseConv2d = wqaq/iao/quantize.py:837-994) followed by the block's ReLU, evaluated in fp32 (the reference) and
in fp64 (the conditioning reference for sums that cancel)."""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from micronet_amd import _lib
from oracle import torch_oracle as TO

CASES = [
    # N, C, O, H, W, groups, shuffle, bias
    dict(N=3, C=32, O=32, H=4, W=8, groups=2, shuffle=0, bias=True),
    dict(N=2, C=64, O=32, H=8, W=8, groups=2, shuffle=2, bias=True),
    dict(N=2, C=256, O=256, H=4, W=4, groups=2, shuffle=2, bias=False),      # Cg = Mg = 128: the nin_gc tile
    dict(N=5, C=48, O=96, H=2, W=6, groups=1, shuffle=0, bias=True),         # Cg = 48 (padded to 64), Mg = 96 (three K-steps)
    dict(N=4, C=32, O=32, H=16, W=24, groups=2, shuffle=2, bias=True),       # 24 slabs: three partial Gram tiles per group (Z = 3), several chunks per wave
]


def _shuffle(x, g):
    n, c, h, w = x.shape
    return x.view(n, g, c // g, h, w).transpose(1, 2).contiguous().view(n, c, h, w)


def _oracle(case, seed, dtype, steps):
    """steps: list of (x_phys, g); returns per step dict of outputs / gradients, and the module after the last step."""
    torch.manual_seed(seed)
    conv = nn.Conv2d(case["C"], case["O"], 1, groups=case["groups"], bias=case["bias"])
    bn = nn.BatchNorm2d(case["O"])
    with torch.no_grad():
        bn.weight.uniform_(0.3, 1.2)
        bn.bias.normal_(0, 0.2)
    m = TO.OBNFuseConv2d(conv, bn, a_bits=8, w_bits=8, q_type=0, q_level=0, weight_observer=0).train()
    if dtype == torch.float64:
        m = m.double()
    recs = []
    for x_phys, g in steps:
        xp = x_phys.to(dtype).clone().requires_grad_(True)
        xin = _shuffle(xp, case["shuffle"]) if case["shuffle"] > 1 else xp
        for p in m.parameters():
            p.grad = None
        out = torch.relu(m(xin))
        out.backward(g.to(dtype))
        recs.append(dict(out=out.detach(), dx=xp.grad.detach(), dw=m.weight.grad.detach().clone(), db=None if m.bias is None else m.bias.grad.detach().clone(),
                         dgamma=m.gamma.grad.detach().clone(), dbeta=m.beta.grad.detach().clone(), rm=m.running_mean.clone(), rv=m.running_var.clone(),
                         wmin=m.wq.observer.min_val.clone(), wmax=m.wq.observer.max_val.clone(), wscale=m.wq.scale.clone(),
                         amin=m.aq.observer.min_val.clone(), amax=m.aq.observer.max_val.clone()))
    return recs, m


def _close(name, got, ref32, ref64, tol=1e-5):
    got = np.asarray(got, dtype=np.float64).reshape(-1)
    r32 = ref32.detach().double().numpy().reshape(-1)
    r64 = ref64.detach().double().numpy().reshape(-1)
    scale = max(np.abs(r64).max(), 1e-30)
    e_ref = np.abs(got - r32).max() / scale
    e64 = np.abs(got - r64).max() / scale
    own = np.abs(r32 - r64).max() / scale                    # the reference's own fp32 error
    assert e_ref <= tol or e64 <= max(tol, 2.0 * own), "%s: vs fp32 reference %.3g, vs fp64 %.3g (reference's own error %.3g)" % (name, e_ref, e64, own)
    return e_ref, e64


def check_iaobf_pointwise(be, case, seed=0, nsteps=2, relu_mask=True):
    N, Cc, O, H, W, G, sg = case["N"], case["C"], case["O"], case["H"], case["W"], case["groups"], case["shuffle"]
    rng = torch.Generator().manual_seed(1000 + seed)
    steps = []
    for _ in range(nsteps):
        x = torch.relu(torch.randn(N, Cc, H, W, generator=rng) * 1.5 + 0.4)          # what a ReLU block hands over
        g = torch.randn(N, O, H, W, generator=rng)
        steps.append((x, g))
    r32, m32 = _oracle(case, seed, torch.float32, steps)
    r64, _ = _oracle(case, seed, torch.float64, steps)
    torch.manual_seed(seed)
    conv = nn.Conv2d(Cc, O, 1, groups=G, bias=case["bias"])        # the same initial parameters as the oracle's
    bn = nn.BatchNorm2d(O)
    with torch.no_grad():
        bn.weight.uniform_(0.3, 1.2)
        bn.bias.normal_(0, 0.2)
    lib = be.lib
    geom = _lib.ConvGeom(N, Cc, H, W, O, 1, 1, 1, 1, 0, 0, 1, 1, G, sg)
    Cg = Cc // G
    n = float(N * H * W)
    w = be.to_dev(conv.weight.detach().numpy().reshape(O, Cg))
    bias = be.to_dev(conv.bias.detach().numpy()) if case["bias"] else None
    gamma, beta = be.to_dev(bn.weight.detach().numpy()), be.to_dev(bn.bias.detach().numpy())
    rm, rv = be.to_dev(np.zeros(O)), be.to_dev(np.ones(O))
    wmin, wmax, wscale, wzp = be.to_dev(np.zeros(O)), be.to_dev(np.zeros(O)), be.to_dev(np.ones(O)), be.to_dev(np.zeros(O))
    amin, amax, ascale, azp = be.to_dev(np.zeros(1)), be.to_dev(np.zeros(1)), be.to_dev(np.ones(1)), be.to_dev(np.zeros(1))
    assert lib.mn_iaobf_gram_supported(C.byref(geom)) == 1 and lib.mn_iaobf_bwd_data_supported(C.byref(geom)) == 1
    worst = {}
    for it, (x_t, g_t) in enumerate(steps):
        first = int(it == 0)
        x = be.to_dev(x_t.numpy())
        # activation observer + qparams of the input (per tensor, moving average): the existing entry points
        ows = be.empty(int(lib.mn_iao_observe_ws_floats(1, x_t.numel())) + 4)
        be.call("mn_iao_observe", be.ptr(x), 1, x_t.numel(), 1, first, 0.1, be.ptr(amin), be.ptr(amax), be.ptr(ows), be.stream)
        aqp = be.empty(4)
        be.call("mn_iao_qparams", be.ptr(amin), be.ptr(amax), 1, 8, 0, 1, 1, be.ptr(ascale), be.ptr(azp), be.ptr(aqp), be.stream)
        # Gram data
        nb = int(lib.mn_iaobf_gram_ws_bytes(C.byref(geom)))
        ws = be.empty(nb // 4 + 4)
        gram = be.empty(2 * G * Cg * Cg)          # doubles
        sx = be.empty(2 * Cc)
        be.call("mn_iaobf_gram", C.byref(geom), be.ptr(x), be.ptr(gram), be.ptr(sx), be.ptr(ws), nb, be.stream)
        xl = (_shuffle(x_t, sg) if sg > 1 else x_t).double()
        xg = xl.permute(1, 0, 2, 3).reshape(G, Cg, -1)
        gram_ref = torch.einsum("gcp,gdp->gcd", xg, xg).numpy()
        gram_got = np.frombuffer(be.to_host(gram).tobytes(), dtype=np.float64).reshape(G, Cg, Cg)
        sx_got = np.frombuffer(be.to_host(sx).tobytes(), dtype=np.float64)
        # a partial tile accumulates <= 2048 pixels x 6 term products in ONE fp32 accumulator per entry (the emulator adds them strictly in order: the worst case);
        # the Z partials are then summed in fp64.  What the statistics need is checked right below (running mean / var, weight scale within 1e-5 of the reference).
        assert np.abs(gram_got - gram_ref).max() <= 6e-6 * np.abs(gram_ref).max(), ("gram", np.abs(gram_got - gram_ref).max() / np.abs(gram_ref).max())
        assert np.abs(sx_got - xg.sum(-1).reshape(-1).numpy()).max() <= 2e-6 * np.abs(xg.sum(-1)).max().item()
        # forward preparation
        stats, kfold, bias_f, qw, wqp = be.empty(2 * O), be.empty(O), be.empty(O), be.empty((O, Cg)), be.empty((O, 4))
        stats_raw, vc = be.empty(2 * O), be.empty((O, Cg))
        be.call("mn_iaobf_gram_stats", be.ptr(w), be.ptr(bias), be.ptr(gram), be.ptr(sx), O, Cg, G, n, be.ptr(stats_raw), be.ptr(vc), be.stream)
        be.call("mn_iaobf_prep_fwd", be.ptr(w), be.ptr(bias), be.ptr(gamma), be.ptr(beta), O, Cg, be.ptr(stats_raw), 1e-5, 0.1, first,
                be.ptr(rm), be.ptr(rv), 8, 0, 0, first, 0.1, be.ptr(wmin), be.ptr(wmax), be.ptr(wscale), be.ptr(wzp), be.ptr(stats), be.ptr(kfold), be.ptr(bias_f),
                be.ptr(qw), be.ptr(wqp), be.stream)
        worst["rm%d" % it] = _close("running_mean", be.to_host(rm), r32[it]["rm"], r64[it]["rm"])
        worst["rv%d" % it] = _close("running_var", be.to_host(rv), r32[it]["rv"], r64[it]["rv"])
        worst["wmin%d" % it] = _close("weight observer min", be.to_host(wmin), r32[it]["wmin"], r64[it]["wmin"])
        worst["wmax%d" % it] = _close("weight observer max", be.to_host(wmax), r32[it]["wmax"], r64[it]["wmax"])
        worst["wscale%d" % it] = _close("weight scale", be.to_host(wscale), r32[it]["wscale"], r64[it]["wscale"])
        assert np.array_equal(be.to_host(amin), r32[it]["amin"].numpy()) and np.array_equal(be.to_host(amax), r32[it]["amax"].numpy())
        # quantised conv + ReLU + (min, max) partials
        aq = be.actq(2, 8, 0, aqp)
        wq = be.wq(3, 8, 0, 4, wqp)
        cnt = int(lib.mn_conv2d_fwd_act_mm_count(C.byref(geom), C.byref(aq), C.byref(wq)))
        assert cnt > 0
        mm = be.empty(2 * cnt)
        a = be.empty((N, O, H, W))
        nbf = int(lib.mn_conv2d_ws_bytes(C.byref(geom), 0, 0))
        wsf = be.empty(nbf // 4 + 4)
        be.call("mn_conv2d_fwd_act", C.byref(geom), C.byref(aq), C.byref(wq), be.ptr(x), be.ptr(qw), be.ptr(bias_f), be.ptr(a), 1, be.ptr(mm), be.ptr(wsf), nbf, be.stream)
        a_h = be.to_host(a)
        worst["out%d" % it] = _close("relu(out)", a_h, r32[it]["out"], r64[it]["out"])
        mm_h = be.to_host(mm)
        assert mm_h[:cnt].min() == a_h.min() and mm_h[cnt:].max() == a_h.max()
        # backward: the block's own ReLU mask, quantised backward-weight, preparation, fused backward-data
        gy_h = (g_t.numpy() * (a_h > 0)).astype(np.float32)
        gy = be.to_dev(gy_h)
        dwq, dbf = be.empty((O, Cg)), be.empty(O)
        nbw = int(lib.mn_conv2d_ws_bytes(C.byref(geom), 2, 0))
        wsw = be.empty(nbw // 4 + 4)
        be.call("mn_conv2d_bwd_weight", C.byref(geom), C.byref(aq), be.ptr(gy), be.ptr(x), be.ptr(dwq), be.ptr(dbf), be.ptr(wsw), nbw, 0, be.stream)
        dw, dbias, dgamma, dbeta, coef = be.empty((O, Cg)), (be.empty(O) if case["bias"] else None), be.empty(O), be.empty(O), be.empty(4 * O)
        be.call("mn_iaobf_prep_bwd", be.ptr(dwq), be.ptr(dbf), be.ptr(w), be.ptr(bias), be.ptr(gamma), be.ptr(stats), be.ptr(wqp), O, Cg, G, be.ptr(vc), be.ptr(sx), n,
                1e-5, 8, 0, be.ptr(dw), be.ptr(dbias), be.ptr(dgamma), be.ptr(dbeta), be.ptr(coef), be.stream)
        worst["dw%d" % it] = _close("dw", be.to_host(dw), r32[it]["dw"], r64[it]["dw"])
        worst["dgamma%d" % it] = _close("dgamma", be.to_host(dgamma), r32[it]["dgamma"], r64[it]["dgamma"])
        worst["dbeta%d" % it] = _close("dbeta", be.to_host(dbeta), r32[it]["dbeta"], r64[it]["dbeta"])
        if case["bias"]:
            # the gradient of a conv bias in front of a BatchNorm is zero in exact arithmetic; the reference's value is rounding noise relative to dbeta
            assert np.abs(be.to_host(dbias)).max() <= 1e-5 * max(np.abs(r64[it]["dbeta"].numpy()).max(), 1e-30)
        nbd = int(lib.mn_iaobf_bwd_data_ws_bytes(C.byref(geom)))
        wsd = be.empty(nbd // 4 + 4)
        dx = be.empty((N, Cc, H, W))
        be.call("mn_iaobf_bwd_data", C.byref(geom), C.byref(aq), be.ptr(gy), be.ptr(x), be.ptr(w), be.ptr(qw), be.ptr(wqp), be.ptr(coef), be.ptr(sx), int(relu_mask),
                be.ptr(dx), be.ptr(wsd), nbd, be.stream)
        mask = (x_t > 0).to(torch.float32) if relu_mask else torch.ones_like(x_t)
        worst["dx%d" % it] = _close("dx", be.to_host(dx), r32[it]["dx"] * mask, r64[it]["dx"] * mask.double())
        # next step: updated parameters would come from the optimizer; the test keeps them (the oracle does too)
    return worst


def check_fq_maxpool(be, shape=(2, 6, 8, 16), bits=8, q_type=0, seed=0, relu_mask=False):
    """mn_iao_fq_maxpool2x2_fwd / _bwd against the oracle's QuantMaxPool2d (OQuantWrap(nn.MaxPool2d(2, 2)) = wqaq/iao/quantize.py:1347-1359), first training step
    (the observer takes its range from this tensor): values, gradient routing (ATen's first maximum) and the clip-STE bit for bit; ties are frequent (a coarse grid)."""
    lib = be.lib
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(*shape, generator=gen) * 2.0
    if relu_mask:
        x = torch.relu(x)
    x[0, 0, 0, :4] = torch.tensor([0.5, 0.5, 0.5, 0.5])            # an exact tie inside a window
    N, Cc, H, W = shape
    g = torch.randn(N, Cc, H // 2, W // 2, generator=gen)
    m = TO.OQuantWrap(nn.MaxPool2d(2, 2, 0), bits, q_type).train()
    xr = x.clone().requires_grad_(True)
    y_ref = m(xr)
    y_ref.backward(g)
    dx_ref = xr.grad * ((x > 0).float() if relu_mask else 1.0)
    xd = be.to_dev(x.numpy())
    amin, amax, ascale, azp = be.to_dev(np.zeros(1)), be.to_dev(np.zeros(1)), be.to_dev(np.ones(1)), be.to_dev(np.zeros(1))
    ows = be.empty(int(lib.mn_iao_observe_ws_floats(1, x.numel())) + 4)
    be.call("mn_iao_observe", be.ptr(xd), 1, x.numel(), 1, 1, 0.1, be.ptr(amin), be.ptr(amax), be.ptr(ows), be.stream)
    qp = be.empty(4)
    be.call("mn_iao_qparams", be.ptr(amin), be.ptr(amax), 1, bits, q_type, 1, 1, be.ptr(ascale), be.ptr(azp), be.ptr(qp), be.stream)
    assert lib.mn_iao_fq_maxpool2x2_supported(H, W) == 1
    cnt = int(lib.mn_iao_fq_maxpool2x2_mm_count(N * Cc, H, W))
    y, idx, mm = be.empty((N, Cc, H // 2, W // 2)), be.to_dev_u8(np.zeros((N, Cc, H // 2, W // 2))), be.empty(2 * cnt)
    be.call("mn_iao_fq_maxpool2x2_fwd", be.ptr(xd), N * Cc, H, W, be.ptr(qp), bits, q_type, be.ptr(y), be.ptr(idx), be.ptr(mm), be.stream)
    y_h = be.to_host(y)
    assert np.array_equal(y_h, y_ref.detach().numpy())
    mm_h = be.to_host(mm)
    assert mm_h[:cnt].min() == y_h.min() and mm_h[cnt:].max() == y_h.max()
    dx = be.empty(shape)
    gd = be.to_dev(g.numpy())
    be.call("mn_iao_fq_maxpool2x2_bwd", be.ptr(gd), be.ptr(idx), be.ptr(xd), N * Cc, H, W, be.ptr(qp), bits, q_type, int(relu_mask), be.ptr(dx), be.stream)
    assert np.array_equal(be.to_host(dx), dx_ref.numpy())


def check_stream_helpers(be, n=4 * 1000 + 8, seed=0):
    """mn_add_relu_mask / mn_relu_mm against numpy, bit for bit."""
    rng = np.random.RandomState(seed)
    a, b, x = (rng.randn(n).astype(np.float32) for _ in range(3))
    x[::7] = 0.0
    ad, bd, xd = be.to_dev(a), be.to_dev(b), be.to_dev(x)
    for bb, xx in ((bd, xd), (None, xd), (bd, None)):
        out = be.empty(n)
        be.call("mn_add_relu_mask", be.ptr(ad), be.ptr(bb), be.ptr(xx), be.ptr(out), n, be.stream)
        ref = a + (b if bb is not None else 0)
        ref = np.where(x > 0, ref, 0).astype(np.float32) if xx is not None else ref.astype(np.float32)
        assert np.array_equal(be.to_host(out), ref)
    cnt = int(be.lib.mn_relu_mm_count(n))
    y, mm = be.empty(n), be.empty(2 * cnt)
    be.call("mn_relu_mm", be.ptr(ad), be.ptr(y), n, be.ptr(mm), be.stream)
    ref = np.maximum(a, 0)
    assert np.array_equal(be.to_host(y), ref)
    mmh = be.to_host(mm)
    assert mmh[:cnt].min() == ref.min() and mmh[cnt:].max() == ref.max()


def check_iao_codes_at_boundaries(be, seed=0):
    """The division-free activation-code path of the pointwise code-domain kernels (iao_code_fast, qgemm_dev.h) on inputs that sit ON and within a few ulp of every
    rounding boundary (k + 0.5) * scale: forward and backward-weight must use exactly the codes of the reference expression clamp(rha(x / s)) -- one wrong code moves
    an output by a whole weight step (>= 1e-3 relative), the test allows 1e-6."""
    lib = be.lib
    rng = np.random.RandomState(seed)
    N, Cc, H, W, O = 2, 32, 8, 8, 32
    amax = np.float32(100.0)
    sc = np.float32(np.float32(amax) / np.float32(127.5))          # update_qparams (293-305): float_range / ((qmax - qmin) / 2), activations: qmin = -128
    ks = rng.randint(-127, 127, size=(N, Cc, H, W)).astype(np.float32)
    x = ((ks + np.float32(0.5)) * sc).astype(np.float32)
    ulps = rng.randint(-6, 7, size=x.shape)
    for u in range(1, 7):
        x = np.where(ulps >= u, np.nextafter(x, np.float32(np.inf)), x)
        x = np.where(ulps <= -u, np.nextafter(x, np.float32(-np.inf)), x)
    x = x.astype(np.float32)
    x.reshape(-1)[0], x.reshape(-1)[1] = amax, -amax
    x = np.clip(x, -amax, amax)
    wcode = rng.randint(-127, 128, size=(O, Cc)).astype(np.float32)
    wsc = (rng.rand(O).astype(np.float32) * 0.01 + 0.001).astype(np.float32)
    qw = (wcode * wsc[:, None]).astype(np.float32)
    code_ref = np.clip(np.sign(x / sc) * np.floor(np.abs(x / sc) + np.float32(0.5)), -128, 127).astype(np.float64)
    y_ref = np.einsum("oc,nchw->nohw", wcode.astype(np.float64), code_ref) * (wsc.astype(np.float64)[None, :, None, None] * np.float64(sc))
    xd = be.to_dev(x)
    amin_d, amax_d, asc_d, azp_d = be.to_dev(np.zeros(1)), be.to_dev(np.zeros(1)), be.to_dev(np.ones(1)), be.to_dev(np.zeros(1))
    ows = be.empty(int(lib.mn_iao_observe_ws_floats(1, x.size)) + 4)
    be.call("mn_iao_observe", be.ptr(xd), 1, x.size, 1, 1, 0.1, be.ptr(amin_d), be.ptr(amax_d), be.ptr(ows), be.stream)
    qp = be.empty(4)
    be.call("mn_iao_qparams", be.ptr(amin_d), be.ptr(amax_d), 1, 8, 0, 1, 1, be.ptr(asc_d), be.ptr(azp_d), be.ptr(qp), be.stream)
    assert be.to_host(asc_d)[0] == sc
    geom = _lib.ConvGeom(N, Cc, H, W, O, 1, 1, 1, 1, 0, 0, 1, 1, 1, 0)
    aq = be.actq(2, 8, 0, qp)
    wsd = be.to_dev(wsc)
    wq = be.wq(3, 8, 0, 1, wsd)
    y = be.conv_fwd(geom, aq, xd, be.to_dev(qw), None, 3, wq=wq)
    err = np.abs(be.to_host(y).astype(np.float64) - y_ref).max() / np.abs(y_ref).max()
    assert err <= 1e-6, ("forward codes", err)
    g = rng.randn(N, O, H, W).astype(np.float32)
    dw, _ = be.conv_bwd_weight(geom, aq, be.to_dev(g), xd, 3, bias=False)
    dw_ref = np.einsum("nohw,nchw->oc", g.astype(np.float64), code_ref) * np.float64(sc)
    errw = np.abs(be.to_host(dw).reshape(O, Cc).astype(np.float64) - dw_ref).max() / np.abs(dw_ref).max()
    assert errw <= 2e-6, ("backward-weight codes", errw)


def check_gram_patch(be, N=3, Cin=3, H=8, W=12, k=5, seed=0):
    """mn_iaobf_gram in patch mode (the first layer): Gram matrix and column sums of the im2col matrix of the image, against torch's unfold in fp64."""
    lib = be.lib
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Cin, H, W, generator=gen) * 1.7 + 0.3
    geom = _lib.ConvGeom(N, Cin, H, W, 16, k, k, 1, 1, k // 2, k // 2, 1, 1, 1, 0)
    assert lib.mn_iaobf_gram_supported(C.byref(geom)) == 1 and lib.mn_iaobf_bwd_data_supported(C.byref(geom)) == 0
    K = Cin * k * k
    nb = int(lib.mn_iaobf_gram_ws_bytes(C.byref(geom)))
    ws, gram, sx = be.empty(nb // 4 + 4), be.empty(2 * K * K), be.empty(2 * K)
    be.call("mn_iaobf_gram", C.byref(geom), be.ptr(be.to_dev(x.numpy())), be.ptr(gram), be.ptr(sx), be.ptr(ws), nb, be.stream)
    cols = torch.nn.functional.unfold(x.double(), k, padding=k // 2)            # [N, K, H * W], rows ordered (c, dy, dx) like weight.reshape(O, K)
    cols = cols.permute(1, 0, 2).reshape(K, -1)
    ref = (cols @ cols.t()).numpy()
    got = np.frombuffer(be.to_host(gram).tobytes(), dtype=np.float64).reshape(K, K)
    sxg = np.frombuffer(be.to_host(sx).tobytes(), dtype=np.float64)
    assert np.abs(got - ref).max() <= 6e-6 * np.abs(ref).max(), np.abs(got - ref).max() / np.abs(ref).max()
    assert np.abs(sxg - cols.sum(1).numpy()).max() <= 2e-6 * np.abs(cols.sum(1).numpy()).max()


# ------------------------------------------------------------------------------------------------ grouped 3 x 3 family (csrc/iao_g3.hip)
G3_CASES = [
    # N, groups, HW (8 -> 8 x 8 maps, four images per tile; 16 -> 16 x 16, one image per tile), channel shuffle, bias, blocks (MN_G3_BLOCKS: tiles per block)
    dict(N=8, G=2, HW=8, shuffle=0, bias=True, blocks=2),          # two tiles per block: the double-buffered pipeline and the running statistics merge
    dict(N=3, G=1, HW=16, shuffle=0, bias=False, blocks=0),        # one tile per block, three blocks per group
    dict(N=4, G=2, HW=16, shuffle=2, bias=True, blocks=4),         # channel shuffle in the addressing, two tiles per block
]


def check_g3(be, case, seed=0):
    """Every entry point of the grouped 3 x 3 BN-fused family against an fp64 evaluation of the same expression (torch conv2d + autograd on the CPU)."""
    import os
    import torch.nn.functional as F
    N, G, HW, sg = case["N"], case["G"], case["HW"], case["shuffle"]
    Cc, O = 16 * G, 32 * G
    rng = np.random.RandomState(100 + seed)
    if case["blocks"]:
        os.environ["MN_G3_BLOCKS"] = str(case["blocks"])
    else:
        os.environ.pop("MN_G3_BLOCKS", None)
    try:
        lib = be.lib
        geom = _lib.ConvGeom(N, Cc, HW, HW, O, 3, 3, 1, 1, 1, 1, 1, 1, G, sg)
        assert lib.mn_iaobf_g3_supported(C.byref(geom)) == 1
        s_g = np.float32(0.05)
        codes = rng.randint(-128, 128, size=(N, Cc, HW, HW)).astype(np.float32)
        codes[rng.rand(*codes.shape) < 0.3] = 0.0          # (a pooled ReLU output has many zeros)
        x_h = (codes * s_g).astype(np.float32)
        w_h = (rng.randn(O, 16, 3, 3) * 0.2).astype(np.float32)
        b_h = (rng.randn(O) * 0.3).astype(np.float32) if case["bias"] else None
        xgrid = be.to_dev(np.array([s_g, 0, -128, 127], dtype=np.float32))
        x, w, bias = be.to_dev(x_h), be.to_dev(w_h), (be.to_dev(b_h) if b_h is not None else None)
        nb = int(lib.mn_iaobf_g3_ws_bytes(C.byref(geom)))
        ws = be.empty(nb // 4 + 4)
        shuf = (lambda t: _shuffle(t, sg)) if sg > 1 else (lambda t: t)
        xp64 = torch.from_numpy(x_h).double().requires_grad_(True)
        xl64 = shuf(xp64)
        w64 = torch.from_numpy(w_h).double()
        b64 = torch.from_numpy(b_h).double() if b_h is not None else None
        rel = lambda got, ref: float(np.abs(np.asarray(got, dtype=np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30))
        worst = {}
        # ---- batch statistics of the raw convolution
        stats = be.empty(2 * O)
        be.call("mn_iaobf_g3_stats", C.byref(geom), be.ptr(x), be.ptr(xgrid), 8, be.ptr(w), be.ptr(bias), be.ptr(stats), be.ptr(ws), nb, be.stream)
        y64 = F.conv2d(xl64, w64, b64, padding=1, groups=G).detach()
        mean64, var64 = y64.mean(dim=(0, 2, 3)).numpy(), y64.var(dim=(0, 2, 3), unbiased=True).numpy()
        st = be.to_host(stats)
        worst["mean"], worst["var"] = rel(st[:O], mean64), rel(st[O:], var64)
        assert worst["mean"] <= 2e-6 and worst["var"] <= 2e-6, worst
        # ---- quantised convolution + ReLU + (min, max)
        s_a = np.float32(0.04)          # x / s_a reaches +-160: the clamp of the 8-bit activation quantizer is exercised
        amax = np.float32(np.abs(x_h).max()) / s_a
        aqp_h = np.array([s_a, 0, -amax, amax], dtype=np.float32)
        aqp = be.to_dev(aqp_h)
        s_w = (0.002 + 0.004 * rng.rand(O)).astype(np.float32)
        wcodes = rng.randint(-127, 128, size=(O, 16, 3, 3)).astype(np.float32)
        qw_h = (wcodes * s_w[:, None, None, None]).astype(np.float32)
        wqp_h = np.stack([s_w, np.zeros(O, np.float32), -np.full(O, 127, np.float32), np.full(O, 127, np.float32)], axis=1).astype(np.float32)
        bf_h = (rng.randn(O) * 0.5).astype(np.float32)
        qw, wqp, bias_f = be.to_dev(qw_h), be.to_dev(wqp_h), be.to_dev(bf_h)
        cnt = int(lib.mn_iaobf_g3_mm_count(C.byref(geom)))
        assert cnt > 0
        out, mm = be.empty((N, O, HW, HW)), be.empty(2 * cnt)
        be.call("mn_iaobf_g3_fwd", C.byref(geom), be.ptr(x), be.ptr(aqp), 8, be.ptr(qw), be.ptr(wqp), be.ptr(bias_f), 1, be.ptr(out), be.ptr(mm), be.stream)
        v32 = (x_h / s_a).astype(np.float32)
        r32 = np.sign(v32) * np.floor(np.abs(v32) + np.float32(0.5))
        xq_h = (np.clip(r32, -128, 127) * s_a).astype(np.float32)
        xq64 = shuf(torch.from_numpy(xq_h).double())
        qw64, bf64 = torch.from_numpy(qw_h).double(), torch.from_numpy(bf_h).double()
        out64 = torch.relu(F.conv2d(xq64, qw64, bf64, padding=1, groups=G)).numpy()
        out_h = be.to_host(out)
        worst["out"] = rel(out_h, out64)
        assert worst["out"] <= 2e-6, worst
        assert ((out_h > 0) == (out64 > 1e-4 * np.abs(out64).max())).mean() > 0.999
        mm_h = be.to_host(mm)
        assert mm_h[:cnt].min() == out_h.min() and mm_h[cnt:].max() == out_h.max()
        out_nr = be.empty((N, O, HW, HW))
        be.call("mn_iaobf_g3_fwd", C.byref(geom), be.ptr(x), be.ptr(aqp), 8, be.ptr(qw), be.ptr(wqp), be.ptr(bias_f), 0, be.ptr(out_nr), None, be.stream)
        assert np.array_equal(np.maximum(be.to_host(out_nr), 0), out_h)
        # ---- d y_raw
        coef_h = np.concatenate([rng.randn(O) * 1e-3, rng.randn(O) * 1e-2, np.zeros(2 * O)]).astype(np.float32)
        coef = be.to_dev(coef_h)
        dy = be.empty((N, O, HW, HW))
        be.call("mn_iaobf_g3_dyraw", C.byref(geom), be.ptr(x), be.ptr(xgrid), 8, be.ptr(w), be.ptr(bias), be.ptr(stats), be.ptr(coef), be.ptr(dy), be.stream)
        dy64 = (coef_h[:O].astype(np.float64)[None, :, None, None] + coef_h[O:2 * O].astype(np.float64)[None, :, None, None] *
                (y64.numpy() - st[:O].astype(np.float64)[None, :, None, None]))
        dy_h = be.to_host(dy)
        worst["dy"] = rel(dy_h, dy64)
        assert worst["dy"] <= 2e-6, worst
        # ---- backward-weight: quantised path (masked d out x activation codes, + d bias), then the statistics path accumulated on top (d y_raw x grid codes)
        g_h = rng.randn(N, O, HW, HW).astype(np.float32)
        gy = be.to_dev(g_h)
        gm64 = torch.from_numpy(g_h * (out_h > 0)).double()
        dw, dbf = be.empty((O, 16, 3, 3)), be.empty(O)
        be.call("mn_iaobf_g3_bwd_weight", C.byref(geom), be.ptr(gy), be.ptr(out), be.ptr(x), be.ptr(aqp), 8, 0, be.ptr(dw), be.ptr(dbf), be.ptr(ws), nb, be.stream)
        wv = qw64.clone().requires_grad_(True)
        (F.conv2d(xq64, wv, None, padding=1, groups=G) * gm64).sum().backward()
        dw_h = be.to_host(dw)
        worst["dwq"] = rel(dw_h, wv.grad.numpy())
        worst["dbf"] = rel(be.to_host(dbf), gm64.sum(dim=(0, 2, 3)).numpy())
        assert worst["dwq"] <= 2e-6 and worst["dbf"] <= 2e-6, worst
        # (pre-masked gradient, mask = NULL: the same numbers)
        gmask = be.to_dev((g_h * (out_h > 0)).astype(np.float32))
        dw2 = be.empty((O, 16, 3, 3))
        be.call("mn_iaobf_g3_bwd_weight", C.byref(geom), be.ptr(gmask), None, be.ptr(x), be.ptr(aqp), 8, 0, be.ptr(dw2), None, be.ptr(ws), nb, be.stream)
        assert np.array_equal(be.to_host(dw2), dw_h)
        be.call("mn_iaobf_g3_bwd_weight", C.byref(geom), be.ptr(dy), None, be.ptr(x), be.ptr(xgrid), 8, 1, be.ptr(dw), None, be.ptr(ws), nb, be.stream)
        wr = w64.clone().requires_grad_(True)
        (F.conv2d(xl64.detach(), wr, None, padding=1, groups=G) * torch.from_numpy(dy_h).double()).sum().backward()
        worst["dw_sum"] = rel(be.to_host(dw), wv.grad.numpy() + wr.grad.numpy())
        assert worst["dw_sum"] <= 2e-6, worst
        # ---- backward-data: clip-STE(W_q^T d out) + W^T d y_raw, with and without the ReLU mask of the block in front
        for relu_in in (0, 1):
            dx = be.empty((N, Cc, HW, HW))
            be.call("mn_iaobf_g3_bwd_data", C.byref(geom), be.ptr(gy), be.ptr(out), be.ptr(dy), be.ptr(x), be.ptr(aqp), 8, be.ptr(qw), be.ptr(wqp), be.ptr(w), relu_in,
                    be.ptr(dx), be.stream)
            xa = torch.from_numpy(xq_h).double().requires_grad_(True)
            (F.conv2d(shuf(xa), qw64, None, padding=1, groups=G) * gm64).sum().backward()
            passes = (r32 >= -128) & (r32 <= 127) & (v32 <= aqp_h[3]) & (v32 >= aqp_h[2])
            xb = torch.from_numpy(x_h).double().requires_grad_(True)
            (F.conv2d(shuf(xb), w64, None, padding=1, groups=G) * torch.from_numpy(dy_h).double()).sum().backward()
            ref = xa.grad.numpy() * passes + xb.grad.numpy()
            if relu_in:
                ref = ref * (x_h > 0)
            worst["dx%d" % relu_in] = rel(be.to_host(dx), ref)
            assert worst["dx%d" % relu_in] <= 5e-6, worst          # (288 x 9 term products in one fp32 accumulator, in order on the emulator)
        return worst
    finally:
        os.environ.pop("MN_G3_BLOCKS", None)


def check_first_layer_act(be, N=3, Cin=3, H=8, W=12, O=40, k=5, seed=0):
    """mn_conv2d_fwd_act on the first-layer kernels (real operands): relu(conv) equals the plain forward's values rectified bit for bit, the (min, max) partials
    reduce to the extremes of what was stored; relu = 0 returns the plain forward."""
    rng = np.random.RandomState(seed)
    x = be.to_dev(rng.randn(N, Cin, H, W).astype(np.float32))
    w = be.to_dev((rng.randn(O, Cin, k, k) * 0.2).astype(np.float32))
    b = be.to_dev(rng.randn(O).astype(np.float32))
    geom = _lib.ConvGeom(N, Cin, H, W, O, k, k, 1, 1, k // 2, k // 2, 1, 1, 1, 0)
    lib = be.lib
    assert lib.mn_conv2d_first_supported(C.byref(geom), 0) == 1
    none = be.actq(0)
    cnt = int(lib.mn_conv2d_fwd_act_mm_count(C.byref(geom), C.byref(none), None))
    assert cnt > 0
    nb = int(lib.mn_conv2d_ws_bytes(C.byref(geom), 0, 0))
    ws = be.empty(nb // 4 + 4)
    y0, y1, y2, mm = be.empty((N, O, H, W)), be.empty((N, O, H, W)), be.empty((N, O, H, W)), be.empty(2 * cnt)
    be.call("mn_conv2d_fwd", C.byref(geom), C.byref(none), None, be.ptr(x), be.ptr(w), be.ptr(b), be.ptr(y0), be.ptr(ws), nb, 0, be.stream)
    be.call("mn_conv2d_fwd_act", C.byref(geom), C.byref(none), None, be.ptr(x), be.ptr(w), be.ptr(b), be.ptr(y1), 1, be.ptr(mm), be.ptr(ws), nb, be.stream)
    be.call("mn_conv2d_fwd_act", C.byref(geom), C.byref(none), None, be.ptr(x), be.ptr(w), be.ptr(b), be.ptr(y2), 0, None, be.ptr(ws), nb, be.stream)
    y0h, y1h, mmh = be.to_host(y0), be.to_host(y1), be.to_host(mm)
    assert np.array_equal(y1h, np.maximum(y0h, 0)) and np.array_equal(be.to_host(y2), y0h)
    assert mmh[:cnt].min() == y1h.min() and mmh[cnt:].max() == y1h.max()


# ------------------------------------------------------------------------------------------------ thin-output pointwise family (csrc/iao_thin.hip)
def check_thin(be, N=3, Cc=128, O=10, HW=(4, 8), shuffle=0, bias=True, seed=0):
    """Every entry point of the thin-output family against an fp64 evaluation of the same expression."""
    import torch.nn.functional as F
    H, W = HW
    rng = np.random.RandomState(200 + seed)
    lib = be.lib
    geom = _lib.ConvGeom(N, Cc, H, W, O, 1, 1, 1, 1, 0, 0, 1, 1, 1, shuffle)
    assert lib.mn_iaobf_thin_supported(C.byref(geom)) == 1
    x_h = np.maximum(rng.randn(N, Cc, H, W) * 1.2 + 0.3, 0).astype(np.float32)
    w_h = (rng.randn(O, Cc) * 0.1).astype(np.float32)
    b_h = (rng.randn(O) * 0.3).astype(np.float32) if bias else None
    x, w, b = be.to_dev(x_h), be.to_dev(w_h), (be.to_dev(b_h) if bias else None)
    shuf = (lambda t: _shuffle(t, shuffle)) if shuffle > 1 else (lambda t: t)
    rel = lambda got, ref: float(np.abs(np.asarray(got, dtype=np.float64) - ref).max() / max(np.abs(ref).max(), 1e-30))
    worst = {}
    wt = be.empty((Cc, 16))
    be.call("mn_iaobf_thin_pack", be.ptr(w), O, Cc, be.ptr(wt), be.stream)
    wt_h = be.to_host(wt)
    assert np.array_equal(wt_h[:, :O], w_h.T) and not wt_h[:, O:].any()
    # raw convolution
    y = be.empty((N, O, H, W))
    be.call("mn_iaobf_thin_fwd", C.byref(geom), be.ptr(x), None, 8, be.ptr(wt), be.ptr(b), 0, be.ptr(y), None, be.stream)
    xl64 = shuf(torch.from_numpy(x_h).double())
    w64 = torch.from_numpy(w_h).double().reshape(O, Cc, 1, 1)
    y64 = F.conv2d(xl64, w64, torch.from_numpy(b_h).double() if bias else None).numpy()
    worst["y_raw"] = rel(be.to_host(y), y64)
    assert worst["y_raw"] <= 2e-6, worst
    # quantised convolution + ReLU + (min, max)
    s_a = np.float32(np.abs(x_h).max() / 100.0)          # x / s_a reaches 100... some values clamp at 127 only if above: make a few clamp
    s_a = np.float32(s_a * 0.7)
    amax = np.float32(np.abs(x_h).max()) / s_a
    aqp_h = np.array([s_a, 0, -amax, amax], dtype=np.float32)
    aqp = be.to_dev(aqp_h)
    s_w = (0.001 + 0.002 * rng.rand(O)).astype(np.float32)
    qw_h = (rng.randint(-127, 128, size=(O, Cc)).astype(np.float32) * s_w[:, None]).astype(np.float32)
    bf_h = (rng.randn(O) * 0.5).astype(np.float32)
    qw, bias_f = be.to_dev(qw_h), be.to_dev(bf_h)
    qwt = be.empty((Cc, 16))
    be.call("mn_iaobf_thin_pack", be.ptr(qw), O, Cc, be.ptr(qwt), be.stream)
    cnt = int(lib.mn_iaobf_thin_mm_count(C.byref(geom)))
    out, mm = be.empty((N, O, H, W)), be.empty(2 * cnt)
    be.call("mn_iaobf_thin_fwd", C.byref(geom), be.ptr(x), be.ptr(aqp), 8, be.ptr(qwt), be.ptr(bias_f), 1, be.ptr(out), be.ptr(mm), be.stream)
    v32 = (x_h / s_a).astype(np.float32)
    r32 = np.sign(v32) * np.floor(np.abs(v32) + np.float32(0.5))
    xq_h = (np.clip(r32, -128, 127) * s_a).astype(np.float32)
    assert (r32 > 127).any()
    xq64 = shuf(torch.from_numpy(xq_h).double())
    qw64 = torch.from_numpy(qw_h).double().reshape(O, Cc, 1, 1)
    out64 = torch.relu(F.conv2d(xq64, qw64, torch.from_numpy(bf_h).double())).numpy()
    out_h = be.to_host(out)
    worst["out"] = rel(out_h, out64)
    assert worst["out"] <= 2e-6, worst
    mm_h = be.to_host(mm)
    assert mm_h[:cnt].min() == out_h.min() and mm_h[cnt:].max() == out_h.max()
    # backward-weight: quantised path (+ d bias), then the raw path accumulated on top
    g_h = (rng.randn(N, O, H, W) * (out_h > 0)).astype(np.float32)
    dy_h = (rng.randn(N, O, H, W) * 0.1).astype(np.float32)
    gy, dy = be.to_dev(g_h), be.to_dev(dy_h)
    dw, dbf = be.empty((O, Cc)), be.empty(O)
    be.call("mn_iaobf_thin_bwd_weight", C.byref(geom), be.ptr(gy), be.ptr(x), be.ptr(aqp), 8, 0, be.ptr(dw), be.ptr(dbf), be.stream)
    g64, d64 = torch.from_numpy(g_h).double(), torch.from_numpy(dy_h).double()
    dwq64 = torch.einsum("nohw,nchw->oc", g64, xq64).numpy()
    worst["dwq"] = rel(be.to_host(dw), dwq64)
    worst["dbf"] = rel(be.to_host(dbf), g64.sum(dim=(0, 2, 3)).numpy())
    assert worst["dwq"] <= 2e-6 and worst["dbf"] <= 2e-6, worst
    be.call("mn_iaobf_thin_bwd_weight", C.byref(geom), be.ptr(dy), be.ptr(x), None, 8, 1, be.ptr(dw), None, be.stream)
    dwr64 = torch.einsum("nohw,nchw->oc", d64, xl64).numpy()
    worst["dw_sum"] = rel(be.to_host(dw), dwq64 + dwr64)
    assert worst["dw_sum"] <= 2e-6, worst
    # backward-data
    for relu_in in (0, 1):
        dx = be.empty((N, Cc, H, W))
        be.call("mn_iaobf_thin_bwd_data", C.byref(geom), be.ptr(gy), be.ptr(dy), be.ptr(x), be.ptr(aqp), 8, be.ptr(qwt), be.ptr(wt), relu_in, be.ptr(dx), be.stream)
        xa = torch.from_numpy(xq_h).double().requires_grad_(True)
        (F.conv2d(shuf(xa), qw64) * g64).sum().backward()
        xb = torch.from_numpy(x_h).double().requires_grad_(True)
        (F.conv2d(shuf(xb), w64) * d64).sum().backward()
        passes = (r32 >= -128) & (r32 <= 127) & (v32 <= aqp_h[3]) & (v32 >= aqp_h[2])
        ref = xa.grad.numpy() * passes + xb.grad.numpy()
        if relu_in:
            ref = ref * (x_h > 0)
        worst["dx%d" % relu_in] = rel(be.to_host(dx), ref)
        assert worst["dx%d" % relu_in] <= 2e-6, worst
    return worst


def check_gram_stats_variance_clamp(be, O=8, Cg=16, n=4096.0):
    """the batch variance from Gram data, w^T (G - n xbar xbar^T) w / (n - 1), can cancel to a slightly NEGATIVE number for a near-constant channel with a
    large mean; sqrt(var + eps) downstream would poison the running statistics and the folded weights with NaN for good.  Gram data of a constant input, rounded DOWN
    in its last bits: the variance written must be exactly 0, never negative."""
    c = 37.25
    gram = np.full((1, Cg, Cg), n * c * c * (1.0 - 3e-16), dtype=np.float64)
    sx = np.full(Cg, n * c, dtype=np.float64)
    w = np.random.default_rng(5).standard_normal((O, Cg)).astype(np.float32)
    d_gram, d_sx = be.to_dev(gram.view(np.float32).reshape(-1)), be.to_dev(sx.view(np.float32).reshape(-1))
    stats, vc = be.empty(2 * O), be.empty((O, Cg))
    be.call("mn_iaobf_gram_stats", be.ptr(be.to_dev(w)), None, be.ptr(d_gram), be.ptr(d_sx), O, Cg, 1, float(n), be.ptr(stats), be.ptr(vc), be.stream)
    got = be.to_host(stats)
    q_ref = np.einsum("oi,ij,oj->o", w.astype(np.float64), gram[0] - n * np.outer(sx / n, sx / n), w.astype(np.float64))
    assert (q_ref < 0).any(), "the case no longer provokes a negative sum"
    assert np.all(got[O:] >= 0.0) and np.all(np.isfinite(got)), got[O:]
