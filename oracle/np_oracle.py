"""CPU oracle (numpy) for micronet's fake-quantized conv hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped product path
(``micronet_amd/``) may import this file; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, and
only as the checker.

This is a restatement -- in plain numpy fp32 arithmetic -- of the algorithm in
the reference repo ``666DZY666/micronet`` (paths below are relative to the
reference root, ``micronet/compression/quantization/``):

  * ``wqaq/dorefa/quantize.py``  Round 11-21, ActivationQuantizer 36-46,
    WeightQuantizer 61-73
  * ``wbwtab/quantize.py``       BinaryActivation 11-36, BinaryWeight 40-51,
    Ternary 55-75, meancenter_clamp_convparams 98-102, WeightQuantizer 116-149
  * ``wqaq/iao/quantize.py``     observers 15-113, Round 144-168,
    Quantizer.forward 214-240, Signed/Unsigned ranges 243-288,
    Symmetric/Asymmetric update_qparams 293-321, QuantBNFuseConv2d.forward
    837-994, QuantAdd.forward 1484-1498

Parity pinning: the reference holds NO golden vectors or known-answer tests
for this path.  The oracle is therefore pinned
against outputs of the reference itself, imported and executed on CPU in the
build container: ``tests/golden/make_golden.py`` generates the fixtures in
``tests/golden/*.npz`` from the real reference modules, and
``tests/test_oracle_golden.py`` checks every function below against them
(bit-exact for everything except ``tanh``-dependent values and float conv
accumulation, whose tolerances are stated in the test).

Every forward returns fp32 arrays computed with fp32 IEEE operations in the
same order as the reference's ATen op chain; every backward is the explicit
formula autograd produces for that chain.
"""
import numpy as np

F32 = np.float32


def _f(x):
    return np.asarray(x, dtype=F32)


def rha(v):
    """round-half-away in fp32: sign(v) * floor(|v| + 0.5).

    dorefa/quantize.py:13-16, iao/quantize.py:158-160.  Note the ``+ 0.5`` is an
    fp32 add, so rha(0.49999997) == 1.
    """
    v = _f(v)
    return (np.sign(v) * np.floor(np.abs(v) + F32(0.5))).astype(F32)


# --------------------------------------------------------------------------
# DoReFa  (wqaq/dorefa/quantize.py)
# --------------------------------------------------------------------------
def dorefa_scale(bits):
    """python float 1/(2^bits-1) cast to fp32 when it meets an fp32 tensor (43, 70)."""
    return F32(1.0 / float(2 ** bits - 1))


def dorefa_act_fwd(x, a_bits):
    """ActivationQuantizer.forward (36-46). Returns (y, codes)."""
    x = _f(x)
    if a_bits == 32:
        return x.copy(), None
    assert a_bits != 1
    s = dorefa_scale(a_bits)
    t = x * F32(0.1)
    c = np.minimum(np.maximum(t, F32(0)), F32(1))
    # torch.clamp propagates NaN; np.minimum/maximum do as well.
    j = rha(c / s)
    return (j * s).astype(F32), j


def dorefa_act_bwd(g, x, a_bits):
    """autograd of the chain: mul(s) -> Round STE -> div(s) -> clamp mask -> mul(0.1)."""
    g = _f(g)
    x = _f(x)
    if a_bits == 32:
        return g.copy()
    s = dorefa_scale(a_bits)
    t = x * F32(0.1)
    d = (g * s) / s
    mask = (t >= F32(0)) & (t <= F32(1))
    d = np.where(mask, d, F32(0)).astype(F32)
    return (d * F32(0.1)).astype(F32)


def dorefa_w_fwd(w, w_bits, tanh_w=None):
    """WeightQuantizer.forward (61-73). Returns (out, codes, t, M).

    ``tanh_w`` lets the caller inject tanh(w) computed by torch-CPU (MKL VML vsTanh in the reference environment), which
    is not bit-identical to numpy's libm tanh.
    """
    w = _f(w)
    if w_bits == 32:
        return w.copy(), None, None, None
    assert w_bits != 1
    s = dorefa_scale(w_bits)
    t = _f(np.tanh(w)) if tanh_w is None else _f(tanh_w)
    M = np.max(np.abs(t)).astype(F32)
    u = (t / F32(2)) / M + F32(0.5)
    k = rha(u / s)
    q = k * s
    out = F32(2) * q - F32(1)
    return out.astype(F32), k, t, M


def dorefa_w_bwd(g, w, w_bits, tanh_w=None):
    """Backward of WeightQuantizer (Appendix A2), incl. the path through max|t|."""
    g = _f(g)
    w = _f(w)
    if w_bits == 32:
        return g.copy()
    s = dorefa_scale(w_bits)
    t = _f(np.tanh(w)) if tanh_w is None else _f(tanh_w)
    a = np.abs(t)
    M = np.max(a).astype(F32)
    dq = g * F32(2)
    dk = dq * s
    du = dk / s                      # STE through round, then d(u/s)/du
    v = t / F32(2)
    dv = du / M                      # d(v/M)/dv
    dt1 = dv / F32(2)
    # d(v/M)/dM = -v/M^2 ; autograd: grad * (-v / (M*M))
    dM = np.sum((-du * v / (M * M)).astype(F32), dtype=F32)
    amax_mask = (a == M)
    cnt = F32(amax_mask.sum())
    dt2 = np.where(amax_mask, (dM / cnt) * np.sign(t), F32(0)).astype(F32)
    dt = dt1 + dt2
    return (dt * (F32(1) - t * t)).astype(F32)


# --------------------------------------------------------------------------
# WbWtAb  (wbwtab/quantize.py)
# --------------------------------------------------------------------------
def binact_fwd(x):
    """BinaryActivation.forward (13-19): sign(x), with 0 (and -0) -> +1."""
    x = _f(x)
    y = np.sign(x)
    y[y == 0] = 1
    return y.astype(F32)


def binact_bwd(g, x):
    """BinaryActivation.backward (22-36): saturating STE, zero where |x| >= 1."""
    g = _f(g).copy()
    x = _f(x)
    g[x >= F32(1.0)] = 0
    g[x <= F32(-1.0)] = 0
    return g


def _chan_mean_abs(w):
    """torch.mean(|w|, (3,2,1), keepdim=True): fp32 sum then divide."""
    a = np.abs(w).reshape(w.shape[0], -1)
    n = a.shape[1]
    return (a.sum(axis=1, dtype=F32) / F32(n)).reshape(-1, 1, 1, 1).astype(F32)


def ternary_w_fwd(w):
    """WeightQuantizer W==3 branch (132-146) + Ternary.forward (57-68).

    Returns (out, t, alpha, thr, cnt).  An all-zero channel gives 0/0 = NaN.
    """
    w = _f(w)
    E = _chan_mean_abs(w)
    thr = E * F32(0.7)
    t = np.sign(np.sign(w + thr) + np.sign(w - thr)).astype(F32)
    a = np.abs(w)
    gt = a > thr
    a_th = np.where(gt, a, F32(0)).astype(F32)
    ssum = a_th.reshape(w.shape[0], -1).sum(axis=1, dtype=F32).reshape(-1, 1, 1, 1)
    cnt = gt.reshape(w.shape[0], -1).sum(axis=1).astype(F32).reshape(-1, 1, 1, 1)
    with np.errstate(invalid="ignore", divide="ignore"):
        alpha = (ssum / cnt).astype(F32)
        out = (t * alpha).astype(F32)
    return out, t, alpha, thr, cnt


def ternary_w_bwd(g, w):
    """Appendix A4: dW = g*alpha + sign(w)*[|w|>thr]/cnt * sum_o(g*t)."""
    g = _f(g)
    w = _f(w)
    out, t, alpha, thr, cnt = ternary_w_fwd(w)
    gt = np.abs(w) > thr
    gsum = (g * t).reshape(w.shape[0], -1).sum(axis=1, dtype=F32).reshape(-1, 1, 1, 1)
    with np.errstate(invalid="ignore", divide="ignore"):
        d_alpha_path = np.where(gt, np.sign(w) * (gsum / cnt), F32(0)).astype(F32)
        return (g * alpha + d_alpha_path).astype(F32)


def binary_w_center_clamp(w):
    """meancenter_clamp_convparams (98-102): returns the NEW weight.data."""
    w = _f(w)
    mean = (w.sum(axis=1, keepdims=True, dtype=F32) / F32(w.shape[1])).astype(F32)
    w2 = w - mean
    return np.minimum(np.maximum(w2, F32(-1.0)), F32(1.0)).astype(F32)


def binary_w_fwd(w):
    """WeightQuantizer W==2 branch (121-130). Returns (out, w_new, b, alpha)."""
    w_new = binary_w_center_clamp(w)
    alpha = _chan_mean_abs(w_new)
    b = np.sign(w_new)
    b[b == 0] = 1
    b = b.astype(F32)
    return (b * alpha).astype(F32), w_new, b, alpha


def binary_w_bwd(g, w_new):
    """Appendix A5 (w_new = the centred/clamped weight the forward saw)."""
    g = _f(g)
    w_new = _f(w_new)
    alpha = _chan_mean_abs(w_new)
    b = np.sign(w_new)
    b[b == 0] = 1
    n = F32(w_new[0].size)
    gsum = (g * b).reshape(g.shape[0], -1).sum(axis=1, dtype=F32).reshape(-1, 1, 1, 1)
    return (g * alpha + np.sign(w_new) * (gsum / n)).astype(F32)


# --------------------------------------------------------------------------
# IAO  (wqaq/iao/quantize.py)
# --------------------------------------------------------------------------
def observe(x, q_level):
    """ObserverBase.forward (23-36): (min, max) at level 'L' / 'C' / 'FC'."""
    x = _f(x)
    if q_level == "L":
        return x.min().reshape(1).astype(F32), x.max().reshape(1).astype(F32)
    if q_level == "C":
        f = x.reshape(x.shape[0], -1)
        shp = (x.shape[0],) + (1,) * (x.ndim - 1)
        return f.min(axis=1).reshape(shp).astype(F32), f.max(axis=1).reshape(shp).astype(F32)
    if q_level == "FC":
        return x.min(axis=1, keepdims=True).astype(F32), x.max(axis=1, keepdims=True).astype(F32)
    raise ValueError(q_level)


def observer_update(kind, first, old_min, old_max, cur_min, cur_max, momentum=0.1):
    """MinMaxObserver.update_range 62-74 / MovingAverageMinMaxObserver 101-113.

    kind: 'minmax' | 'ema'.  ``first`` is ``num_flag == 0``.
    ``(1 - momentum)`` and ``momentum`` are python doubles that become fp32
    scalars when they multiply an fp32 tensor.
    """
    if first:
        return _f(cur_min).copy(), _f(cur_max).copy()
    if kind == "minmax":
        return np.minimum(cur_min, old_min).astype(F32), np.maximum(cur_max, old_max).astype(F32)
    a = F32(1 - momentum)
    b = F32(momentum)
    return (a * _f(old_min) + b * _f(cur_min)).astype(F32), (a * _f(old_max) + b * _f(cur_max)).astype(F32)


def iao_qrange(bits, q_type, is_activation):
    """Signed/UnsignedQuantizer ranges (243-288)."""
    if q_type == 0:
        if is_activation:
            return F32(-(1 << (bits - 1))), F32((1 << (bits - 1)) - 1)
        return F32(-((1 << (bits - 1)) - 1)), F32((1 << (bits - 1)) - 1)
    if is_activation:
        return F32(0), F32((1 << bits) - 1)
    return F32(0), F32((1 << bits) - 2)


EPS32 = F32(np.finfo(np.float32).eps)


def iao_qparams(min_val, max_val, bits, q_type, is_activation):
    """update_qparams: symmetric 293-305, asymmetric 310-321. -> (scale, zp)."""
    qmin, qmax = iao_qrange(bits, q_type, is_activation)
    min_val = _f(min_val)
    max_val = _f(max_val)
    if q_type == 0:
        quant_range = F32(float(qmax - qmin) / 2)
        float_range = np.maximum(np.abs(min_val), np.abs(max_val))
        scale = np.maximum(float_range / quant_range, EPS32).astype(F32)
        return scale, np.zeros_like(scale)
    quant_range = F32(float(qmax - qmin))
    float_range = max_val - min_val
    scale = np.maximum(float_range / quant_range, EPS32).astype(F32)
    zp = (np.sign(min_val) * np.floor(np.abs(min_val / scale) + F32(0.5))).astype(F32)
    return scale, zp


def iao_fq_fwd(x, scale, zp, bits, q_type, is_activation):
    """Quantizer.forward 227-239. Returns (y, codes = clamp(r)+zp)."""
    x = _f(x)
    qmin, qmax = iao_qrange(bits, q_type, is_activation)
    v = x / scale - zp
    r = rha(v)
    c = np.minimum(np.maximum(r, qmin), qmax)
    codes = (c + zp).astype(F32)
    return (codes * scale).astype(F32), codes


def iao_fq_bwd(g, x, scale, zp, min_val, max_val, bits, q_type, is_activation):
    """Backward of Quantizer.forward (Appendix A8)."""
    g = _f(g)
    x = _f(x)
    qmin, qmax = iao_qrange(bits, q_type, is_activation)
    v = x / scale - zp
    r = rha(v)
    lo = _f(min_val) / scale - zp
    hi = _f(max_val) / scale - zp
    if q_type == 0:
        hi = np.maximum(np.abs(lo), np.abs(hi))
        lo = -hi
    d = (g * scale)                      # d/d(codes) of codes*scale
    d = np.where((r >= qmin) & (r <= qmax), d, F32(0)).astype(F32)   # clamp backward
    d = np.where((v > hi) | (v < lo), F32(0), d).astype(F32)         # Round clip-STE 166-167
    return (d / scale).astype(F32)


# --------------------------------------------------------------------------
# Convolution (the ATen call the reference makes: F.conv2d / F.linear)
# --------------------------------------------------------------------------
def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def conv2d_fwd(x, w, b=None, stride=1, padding=0, dilation=1, groups=1, acc=np.float64):
    """F.conv2d semantics (zero padding), accumulated in ``acc`` precision."""
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    dh, dw = _pair(dilation)
    x = np.asarray(x)
    w = np.asarray(w)
    N, C, H, W = x.shape
    O, Cg, KH, KW = w.shape
    assert C == Cg * groups and O % groups == 0
    Ho = (H + 2 * ph - dh * (KH - 1) - 1) // sh + 1
    Wo = (W + 2 * pw - dw * (KW - 1) - 1) // sw + 1
    xp = np.zeros((N, C, H + 2 * ph, W + 2 * pw), dtype=acc)
    xp[:, :, ph:ph + H, pw:pw + W] = x
    y = np.zeros((N, O, Ho, Wo), dtype=acc)
    Og = O // groups
    wa = w.astype(acc)
    for g in range(groups):
        xg = xp[:, g * Cg:(g + 1) * Cg]
        wg = wa[g * Og:(g + 1) * Og]
        for r in range(KH):
            for s in range(KW):
                patch = xg[:, :, r * dh: r * dh + (Ho - 1) * sh + 1: sh, s * dw: s * dw + (Wo - 1) * sw + 1: sw]
                y[:, g * Og:(g + 1) * Og] += np.einsum("nchw,oc->nohw", patch, wg[:, :, r, s], optimize=True)
    if b is not None:
        y += np.asarray(b, dtype=acc).reshape(1, -1, 1, 1)
    return y


def conv2d_bwd(gy, x, w, stride=1, padding=0, dilation=1, groups=1, acc=np.float64):
    """Returns (dx, dw, db) of conv2d_fwd."""
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    dh, dw_ = _pair(dilation)
    x = np.asarray(x)
    w = np.asarray(w)
    gy = np.asarray(gy).astype(acc)
    N, C, H, W = x.shape
    O, Cg, KH, KW = w.shape
    Ho, Wo = gy.shape[2], gy.shape[3]
    Og = O // groups
    xp = np.zeros((N, C, H + 2 * ph, W + 2 * pw), dtype=acc)
    xp[:, :, ph:ph + H, pw:pw + W] = x
    dxp = np.zeros_like(xp)
    dwt = np.zeros(w.shape, dtype=acc)
    wa = w.astype(acc)
    for g in range(groups):
        gyg = gy[:, g * Og:(g + 1) * Og]
        for r in range(KH):
            for s in range(KW):
                sl = (slice(None), slice(g * Cg, (g + 1) * Cg),
                      slice(r * dh, r * dh + (Ho - 1) * sh + 1, sh),
                      slice(s * dw_, s * dw_ + (Wo - 1) * sw + 1, sw))
                dwt[g * Og:(g + 1) * Og, :, r, s] = np.einsum("nohw,nchw->oc", gyg, xp[sl], optimize=True)
                dxp[sl] += np.einsum("nohw,oc->nchw", gyg, wa[g * Og:(g + 1) * Og, :, r, s], optimize=True)
    dx = dxp[:, :, ph:ph + H, pw:pw + W]
    db = gy.sum(axis=(0, 2, 3))
    return dx, dwt, db


def linear_fwd(x, w, b=None, acc=np.float64):
    y = np.asarray(x, dtype=acc) @ np.asarray(w, dtype=acc).T
    if b is not None:
        y = y + np.asarray(b, dtype=acc)
    return y


def bn_batch_stats(o):
    """QuantBNFuseConv2d 853-855: mean and UNBIASED var over (N,H,W) per channel."""
    o64 = np.asarray(o, dtype=np.float64)
    n = o64.shape[0] * o64.shape[2] * o64.shape[3]
    mean = o64.mean(axis=(0, 2, 3))
    var = o64.var(axis=(0, 2, 3), ddof=1) if n > 1 else np.full_like(mean, np.nan)
    return mean, var


# ---------------------------------------------------------------------------- input pipeline
def cifar_augment(images_u8, index, ox, oy, flip, pad=4, mean=(0.4914, 0.4822, 0.4465), std=(0.2023, 0.1994, 0.2010)):
    """wqaq/dorefa/main.py:203-210 restated for given random draws: RandomCrop(H, padding=pad) = zero-pad the uint8 HWC image by `pad`, take the H x W
    window at (oy, ox); RandomHorizontalFlip = reverse the columns when flip; ToTensor = CHW, float32(pixel).div(255); Normalize = sub(mean).div(std)
    (float32 tensors).  torchvision is not installed in the build container, so this follows its documented semantics (functional.pad fill=0,
    functional.crop, functional.hflip, to_tensor, normalize)."""
    images_u8 = np.asarray(images_u8, dtype=np.uint8)
    n, H, W, C_ = images_u8.shape
    out = np.empty((len(index), C_, H, W), dtype=F32)
    m, s = np.asarray(mean, dtype=F32).reshape(-1, 1, 1), np.asarray(std, dtype=F32).reshape(-1, 1, 1)
    for b, i in enumerate(index):
        padded = np.zeros((H + 2 * pad, W + 2 * pad, C_), dtype=np.uint8)
        padded[pad:pad + H, pad:pad + W] = images_u8[i]
        crop = padded[oy[b]:oy[b] + H, ox[b]:ox[b] + W]
        if flip[b]:
            crop = crop[:, ::-1]
        t = (crop.transpose(2, 0, 1).astype(F32) / F32(255)).astype(F32)
        out[b] = ((t - m) / s).astype(F32)
    return out
