"""Kernel-level parity cases shared by the CPU-emulation run (not gpu) and the MI355X run (-m gpu).

Each check drives the C ABI through ``abi_driver.Backend`` and compares with ``oracle/np_oracle.py`` and the golden
fixtures.  Tolerances: integer/quantize steps bit-exact; fp32 sums over a channel (alpha, mean) <= 5e-7 rel;
float conv accumulate <= 1e-5 * max|ref| (north_star), checked against an fp64 evaluation of the same products.
"""
import ctypes as C

import numpy as np

from oracle import np_oracle as O

F = np.float32


def eq(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


def close(a, b, rel):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = max(np.max(np.abs(b[np.isfinite(b)])) if np.isfinite(b).any() else 0.0, 1e-30)
    ok = np.isfinite(b)
    return a.shape == b.shape and eq(np.isnan(a), np.isnan(b)) and (np.max(np.abs(a[ok] - b[ok])) if ok.any() else 0.0) <= rel * scale


# ----------------------------------------------------------------------------- quantizers vs golden
def check_dorefa_act(be, q, bits):
    x, g = q[f"dorefa_act{bits}_x"], q[f"dorefa_act{bits}_g"]
    n = x.size
    dx_, dg = be.to_dev(x), be.to_dev(g)
    y, dxo = be.empty(n), be.empty(n)
    be.call("mn_dorefa_act_fwd", be.ptr(dx_), be.ptr(y), n, bits, be.stream)
    be.call("mn_dorefa_act_bwd", be.ptr(dg), be.ptr(dx_), be.ptr(dxo), n, bits, be.stream)
    assert eq(be.to_host(y), q[f"dorefa_act{bits}_y"])
    assert eq(be.to_host(dxo), q[f"dorefa_act{bits}_dx"])
    # unaligned / odd length path
    y2 = be.empty(n)
    xs = be.to_dev(np.concatenate([[0], x]))
    if be.kind == "emu":
        be.call("mn_dorefa_act_fwd", C.c_void_p(xs.ctypes.data + 4), be.ptr(y2), n, bits, be.stream)
    else:
        be.call("mn_dorefa_act_fwd", C.c_void_p(xs.data_ptr() + 4), be.ptr(y2), n, bits, be.stream)
    assert eq(be.to_host(y2), q[f"dorefa_act{bits}_y"])
    r = be.empty(n)
    be.call("mn_round_half_away", be.ptr(dx_), be.ptr(r), n, be.stream)
    assert eq(be.to_host(r), O.rha(x))


def check_dorefa_w(be, q, bits):
    w, g = q[f"dorefa_w{bits}_w"], q[f"dorefa_w{bits}_g"]
    n = w.size
    dw_, dg = be.to_dev(w), be.to_dev(g)
    ws = be.empty(int(be.lib.mn_dorefa_w_ws_floats(n)))
    qw, dwo = be.empty(n), be.empty(n)
    be.call("mn_dorefa_w_fwd", be.ptr(dw_), be.ptr(qw), n, bits, be.ptr(ws), be.stream)
    be.call("mn_dorefa_w_bwd", be.ptr(dg), be.ptr(dw_), be.ptr(dwo), n, bits, be.ptr(ws), be.stream)
    ref = q[f"dorefa_w{bits}_y"].reshape(-1)
    got = be.to_host(qw).reshape(-1)
    s = 1.0 / (2 ** bits - 1)
    # codes: identical except where the kernels' (correctly rounded) tanh and torch-CPU (MKL VML) tanh differ in the last ulp at a rounding boundary
    codes_got, codes_ref = np.round((got + 1) / 2 / s), np.round((ref + 1) / 2 / s)
    assert (codes_got != codes_ref).sum() <= 2, (codes_got != codes_ref).sum()
    same = codes_got == codes_ref
    assert eq(got[same], ref[same])
    dref = q[f"dorefa_w{bits}_dw"].reshape(-1)
    assert np.max(np.abs(be.to_host(dwo).reshape(-1) - dref)) <= 2e-6 * np.max(np.abs(dref))


def check_wbwtab(be, q):
    x, g = q["binact_x"], q["binact_g"]
    n = x.size
    dx_, dg = be.to_dev(x), be.to_dev(g)
    y, dxo = be.empty(n), be.empty(n)
    be.call("mn_binact_fwd", be.ptr(dx_), be.ptr(y), n, be.stream)
    be.call("mn_binact_bwd", be.ptr(dg), be.ptr(dx_), be.ptr(dxo), n, be.stream)
    assert eq(be.to_host(y), q["binact_y"]) and eq(be.to_host(dxo), q["binact_dx"])
    # ternary
    w, g = q["ternary_w"], q["ternary_g"]
    Oc, K = w.shape[0], w[0].size
    dw_, dg = be.to_dev(w), be.to_dev(g)
    qw, st, dwo = be.empty(w.shape), be.empty((Oc, 4)), be.empty(w.shape)
    be.call("mn_ternary_w_fwd", be.ptr(dw_), be.ptr(qw), be.ptr(st), Oc, K, be.stream)
    be.call("mn_ternary_w_bwd", be.ptr(dg), be.ptr(dw_), be.ptr(st), be.ptr(dwo), Oc, K, be.stream)
    got, ref = be.to_host(qw), q["ternary_y"]
    ok = ~np.isnan(ref)
    assert eq(np.isnan(got), np.isnan(ref))
    assert eq(np.sign(got[ok]), np.sign(ref[ok]))                       # ternary codes bit-exact
    assert np.max(np.abs(got[ok] - ref[ok])) <= 5e-7 * np.max(np.abs(ref[ok]))
    got, ref = be.to_host(dwo), q["ternary_dw"]
    assert eq(np.isnan(got), np.isnan(ref))
    assert np.max(np.abs(got[ok] - ref[ok])) <= 2e-6 * np.max(np.abs(ref[ok]))
    # binary (in-place centre + clamp)
    w, g = q["binary_w"], q["binary_g"]
    Oc, Cc, R = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
    dw_, dg = be.to_dev(w), be.to_dev(g)
    qw, al, dwo = be.empty(w.shape), be.empty(Oc), be.empty(w.shape)
    be.call("mn_binary_w_fwd", be.ptr(dw_), be.ptr(qw), be.ptr(al), Oc, Cc, R, be.stream)
    be.call("mn_binary_w_bwd", be.ptr(dg), be.ptr(dw_), be.ptr(al), be.ptr(dwo), Oc, Cc * R, be.stream)
    assert np.max(np.abs(be.to_host(dw_) - q["binary_w_after"])) <= 2e-7
    got, ref = be.to_host(qw), q["binary_y"]
    assert eq(np.sign(got), np.sign(ref)) and np.max(np.abs(got - ref)) <= 5e-7 * np.max(np.abs(ref))
    got, ref = be.to_host(dwo), q["binary_dw"]
    assert np.max(np.abs(got - ref)) <= 2e-6 * np.max(np.abs(ref))


def check_iao(be, q, meta):
    for c in meta:
        key, bits, q_type, is_act = c["key"], c["bits"], c["q_type"], int(c["kind"] == "act")
        shape = c["shape"]
        rows = 1 if c["level"] == "L" else shape[0]
        cols = int(np.prod(shape)) // rows
        mn, mx, sc, zp, qp = (be.to_dev(np.zeros(rows)), be.to_dev(np.zeros(rows)), be.to_dev(np.ones(rows)),
                              be.to_dev(np.zeros(rows)), be.empty((rows, 4)))
        ws = be.empty(max(4, int(be.lib.mn_iao_observe_ws_floats(rows, cols))))
        obs_kind = 0 if c["obs"] == "minmax" else 1
        for s in range(3):
            x, g = q[f"{key}_s{s}_x"], q[f"{key}_s{s}_g"]
            dx_, dg = be.to_dev(x), be.to_dev(g)
            be.call("mn_iao_observe", be.ptr(dx_), rows, cols, obs_kind, int(s == 0), 0.1, be.ptr(mn), be.ptr(mx), be.ptr(ws), be.stream)
            be.call("mn_iao_qparams", be.ptr(mn), be.ptr(mx), rows, bits, q_type, is_act, 1, be.ptr(sc), be.ptr(zp), be.ptr(qp), be.stream)
            assert eq(be.to_host(mn).reshape(-1), q[f"{key}_s{s}_min"].reshape(-1)), key
            assert eq(be.to_host(mx).reshape(-1), q[f"{key}_s{s}_max"].reshape(-1)), key
            assert eq(be.to_host(sc).reshape(-1), q[f"{key}_s{s}_scale"].reshape(-1)), key
            assert eq(be.to_host(zp).reshape(-1), q[f"{key}_s{s}_zp"].reshape(-1)), key
            y, dxo = be.empty(x.shape), be.empty(x.shape)
            be.call("mn_iao_fq_fwd", be.ptr(dx_), be.ptr(y), rows, cols, be.ptr(qp), bits, q_type, is_act, be.stream)
            be.call("mn_iao_fq_bwd", be.ptr(dg), be.ptr(dx_), be.ptr(dxo), rows, cols, be.ptr(qp), bits, q_type, is_act, be.stream)
            assert eq(be.to_host(y), q[f"{key}_s{s}_y"]), key
            assert eq(be.to_host(dxo), q[f"{key}_s{s}_dx"]), key
        # eval: snapshot only (update = 0)
        x = q[f"{key}_eval_x"]
        dx_, y = be.to_dev(x), be.empty(x.shape)
        be.call("mn_iao_qparams", be.ptr(mn), be.ptr(mx), rows, bits, q_type, is_act, 0, be.ptr(sc), be.ptr(zp), be.ptr(qp), be.stream)
        be.call("mn_iao_fq_fwd", be.ptr(dx_), be.ptr(y), rows, cols, be.ptr(qp), bits, q_type, is_act, be.stream)
        assert eq(be.to_host(y), q[f"{key}_eval_y"]), key


def check_bn_stats(be, shape=(5, 6, 4, 8), seed=0):
    r = np.random.default_rng(seed)
    o = (r.standard_normal(shape) * 2 + 3).astype(F)
    N, Cc, HW = shape[0], shape[1], shape[2] * shape[3]
    do_, st = be.to_dev(o), be.empty((2, Cc))
    ws = be.empty(int(be.lib.mn_bn_stats_ws_floats(N, Cc, HW)) + 2)
    be.call("mn_bn_stats_fwd", be.ptr(do_), N, Cc, HW, be.ptr(st), be.ptr(ws), be.stream)
    mean, var = O.bn_batch_stats(o)
    got = be.to_host(st)
    assert np.max(np.abs(got[0] - mean)) <= 1e-6 * np.max(np.abs(mean)) and np.max(np.abs(got[1] - var)) <= 2e-6 * np.max(np.abs(var))
    dm, dv = r.standard_normal(Cc).astype(F), r.standard_normal(Cc).astype(F)
    d_o = be.empty(shape)
    ddm, ddv = be.to_dev(dm), be.to_dev(dv)      # keep the device buffers alive across the call
    be.call("mn_bn_stats_bwd", be.ptr(do_), be.ptr(st), be.ptr(ddm), be.ptr(ddv), be.ptr(d_o), N, Cc, HW, be.stream)
    n = N * HW
    ref = dm.reshape(1, -1, 1, 1) / n + dv.reshape(1, -1, 1, 1) * 2 * (o.astype(np.float64) - mean.reshape(1, -1, 1, 1)) / (n - 1)
    assert close(be.to_host(d_o), ref, 2e-6)


# ----------------------------------------------------------------------------- convolution vs numpy fp64
def _quant_x(x, mode, bits, qp, q_type):
    if mode == 1:
        return O.dorefa_act_fwd(x, bits)[0]
    if mode == 2:
        return O.iao_fq_fwd(x, F(qp[0]), F(qp[1]), bits, q_type, True)[0]
    return x


def _ste(gx, x, mode, bits, qp, q_type):
    gx = gx.astype(F)
    if mode == 1:
        return O.dorefa_act_bwd(gx, x, bits)
    if mode == 2:
        qmin, qmax = O.iao_qrange(bits, q_type, True)
        v = x / F(qp[0]) - F(qp[1])
        r = O.rha(v)
        d = gx * F(qp[0])
        d = np.where((r >= qmin) & (r <= qmax), d, F(0))
        d = np.where((v > F(qp[3])) | (v < F(qp[2])), F(0), d)
        return (d / F(qp[0])).astype(F)
    return gx


def make_coded_weights(r, w_shape, wmode, wbits=8):
    """Fake-quantised fp32 weights of the given scheme (what the weight quantizer kernels emit) + the wq descriptor fields."""
    O = w_shape[0]
    if wmode == 1:      # ternary / binary: t * alpha[o]
        t = r.integers(-1, 2, size=w_shape).astype(F)
        alpha = (np.abs(r.standard_normal(O)) * 0.2 + 0.05).astype(F).reshape(-1, 1, 1, 1)
        t.reshape(O, -1)[:, 0] = 1          # every channel has a non-zero code, so max|w| recovers alpha
        return (t * alpha).astype(F), dict(mode=1), None
    if wmode == 2:      # dorefa: 2 * (k * s) - 1
        n = 2 ** wbits - 1
        s = F(1.0 / n)
        k = r.integers(0, n + 1, size=w_shape).astype(F)
        q = (k * s).astype(F)
        return (F(2) * q - F(1)).astype(F), dict(mode=2, bits=wbits), None
    if wmode == 3:      # iao symmetric per-channel: code * scale[o]
        qmax = 2 ** (wbits - 1) - 1
        code = r.integers(-qmax, qmax + 1, size=w_shape).astype(F)
        scale = (np.abs(r.standard_normal(O)) * 0.01 + 0.002).astype(F)
        return (code * scale.reshape(-1, 1, 1, 1)).astype(F), dict(mode=3, bits=wbits, q_type=0, per_channel=1), scale
    raise ValueError(wmode)


def check_conv(be, x_shape, w_shape, stride=1, padding=0, dilation=1, groups=1, bias=True, mode=0, bits=8, q_type=0,
               algos=(1, 2), seed=0, binary_x=False, expect_mfma=None, rel=1e-5, wmode=0, wbits=8, expect_qgemm=None, in_shuffle=0,
               sign8=False, want_dbias=True, expect_kernels=None):
    """fwd / bwd_data / bwd_weight of one geometry on every requested algo vs numpy fp64 on the same fp32 operands.
    wmode != 0: the weights are fake-quantised (ternary / dorefa / iao) and algo 3 (code-domain bf16 MFMA) is exercised.
    sign8: the +-1 activations are handed over as int8 codes (MN_ACTQ_SIGN8, the packed output of mn_bnsign_fwd_i8)."""
    r = np.random.default_rng(seed)
    if sign8:
        binary_x, mode = True, 0
    x = (np.where(r.standard_normal(x_shape) > 0, 1.0, -1.0) if binary_x else r.standard_normal(x_shape) * 4).astype(F)
    wq = None
    if wmode:
        w, wkw, wscale = make_coded_weights(r, w_shape, wmode, wbits)
        wq = be.wq(scale=be.to_dev(wscale) if wscale is not None else None, **wkw)
    else:
        w = (r.standard_normal(w_shape) * 0.3).astype(F)
    b = (r.standard_normal(w_shape[0]) * 0.2).astype(F) if bias else None
    g = be.geom(x_shape, w_shape, stride, padding, dilation, groups)
    x_phys = x
    if in_shuffle > 1:          # the kernels see the physical tensor; the reference convolves channel_shuffle(x_phys)
        g.in_shuffle = in_shuffle
        n_, c_, h_, w_ = x.shape
        x = np.ascontiguousarray(x_phys.reshape(n_, in_shuffle, c_ // in_shuffle, h_, w_).transpose(0, 2, 1, 3, 4).reshape(x.shape))
    qp = None
    if mode == 2:
        mn, mx = F(x.min()), F(x.max())
        sc, zp = O.iao_qparams(mn.reshape(1), mx.reshape(1), bits, q_type, True)
        lo, hi = mn / sc[0] - zp[0], mx / sc[0] - zp[0]
        if q_type == 0:
            hi = max(abs(lo), abs(hi)); lo = -hi
        qp = np.array([sc[0], zp[0], lo, hi], dtype=F)
    qx = _quant_x(x, mode, bits, qp, q_type)
    kw = dict(stride=stride, padding=padding, dilation=dilation, groups=groups)
    y_ref = O.conv2d_fwd(qx, w, b, **kw)
    gy = r.standard_normal(y_ref.shape).astype(F)
    dqx_ref, dw_ref, db_ref = O.conv2d_bwd(gy, qx, w, **kw)
    dx_ref = _ste(dqx_ref, x, mode, bits, qp, q_type)
    if in_shuffle > 1:          # gradient w.r.t. the physical tensor = inverse shuffle of the logical gradient
        n_, c_, h_, w_ = x.shape
        dx_ref = np.ascontiguousarray(dx_ref.reshape(n_, c_ // in_shuffle, in_shuffle, h_, w_).transpose(0, 2, 1, 3, 4).reshape(x.shape))
    dX, dW, dB, dG = (be.to_dev_i8(x_phys) if sign8 else be.to_dev(x_phys)), be.to_dev(w), (be.to_dev(b) if bias else None), be.to_dev(gy)
    dqp = be.to_dev(qp) if qp is not None else None
    aq = be.actq(3 if sign8 else mode, bits, q_type, dqp, flags=1 if (binary_x and mode == 0) else 0)
    sup = [bool(be.lib.mn_conv2d_mfma_supported(C.byref(g), k)) for k in range(3)]
    supq = [bool(be.lib.mn_conv2d_qgemm_supported(C.byref(g), C.byref(aq), C.byref(wq) if wq is not None else None, k)) for k in range(3)]
    if expect_mfma is not None:
        assert sup == [expect_mfma] * 3 if isinstance(expect_mfma, bool) else sup == list(expect_mfma), (sup, expect_mfma)
    if expect_qgemm is not None:
        assert supq == ([expect_qgemm] * 3 if isinstance(expect_qgemm, bool) else list(expect_qgemm)), (supq, expect_qgemm)
    out = {}
    for algo in algos:
        ok = lambda k: (algo == 1 or algo == 0) or (algo == 2 and sup[k]) or (algo == 3 and supq[k])
        lk = lambda: be.lib.mn_last_kernel().decode()
        if ok(0):
            y = be.to_host(be.conv_fwd(g, aq, dX, dW, dB, algo, wq=wq))
            assert close(y, y_ref, rel), ("fwd", algo, np.max(np.abs(y - y_ref)), np.max(np.abs(y_ref)))
            assert expect_kernels is None or lk().startswith(expect_kernels[0]), lk()
        if ok(1):
            dx = be.to_host(be.conv_bwd_data(g, aq, dG, dW, dX, algo, wq=wq))
            # the STE mask multiplies a float; compare where the mask passes with the float tolerance
            assert close(dx, dx_ref, rel), ("bwd_data", algo, np.max(np.abs(dx - dx_ref)), np.max(np.abs(dx_ref)))
            assert expect_kernels is None or lk().startswith(expect_kernels[1]), lk()
        if ok(2):
            dw, db = be.conv_bwd_weight(g, aq, dG, dX, algo, bias=want_dbias)
            dw = be.to_host(dw)
            assert close(dw, dw_ref, rel), ("bwd_weight", algo, np.max(np.abs(dw - dw_ref)), np.max(np.abs(dw_ref)))
            assert expect_kernels is None or lk().startswith(expect_kernels[2]), lk()
            if want_dbias:
                assert close(be.to_host(db), db_ref, rel), ("dbias", algo)
        out[algo] = True
    return sup, supq


# geometry list: (x_shape, w_shape, kwargs) -- covers every tiler branch with small tensors
SMALL_CONV_CASES = [
    # 1x1 grouped, image smaller than a tile (NI > 1), N not a multiple of NI
    dict(x_shape=(3, 16, 4, 4), w_shape=(16, 4, 1, 1), groups=4, bias=False, expect_mfma=True),
    # 3x3 pad 1 grouped (nin_gc L7-like: 8x8, Cg=16, Mg=32)
    dict(x_shape=(3, 32, 8, 8), w_shape=(64, 16, 3, 3), padding=1, groups=2, expect_mfma=True),
    # partial-image tiles (16x16 -> 2 tiles per image), dense 3x3, Mg=40 (padded m-tile), Cg=6 (padded channels)
    dict(x_shape=(2, 6, 16, 16), w_shape=(40, 6, 3, 3), padding=1, expect_mfma=True),
    # stride 2 3x3 (resnet downsample): fwd strided, bwd-data by zero insertion
    dict(x_shape=(2, 8, 16, 16), w_shape=(24, 8, 3, 3), stride=2, padding=1, bias=False, expect_mfma=True),
    # stride 2 1x1 shortcut
    dict(x_shape=(2, 8, 16, 16), w_shape=(16, 8, 1, 1), stride=2, bias=False, expect_mfma=True),
    # 5x5 pad 2, Cin=3 (first layer), several m-blocks
    dict(x_shape=(2, 3, 8, 8), w_shape=(160, 3, 5, 5), padding=2, expect_mfma=True),
    # dilation 2
    dict(x_shape=(2, 4, 8, 8), w_shape=(8, 4, 3, 3), padding=2, dilation=2, expect_mfma=True),
    # 32-wide rows (4 rows per tile), Cg=128 -> several channel chunks
    dict(x_shape=(1, 128, 8, 32), w_shape=(16, 128, 1, 1), expect_mfma=True),
    # shapes the tiler rejects -> direct kernels only
    dict(x_shape=(2, 8, 6, 6), w_shape=(12, 4, 3, 3), padding=1, groups=2, expect_mfma=False),
    dict(x_shape=(2, 4, 7, 7), w_shape=(6, 4, 3, 3), padding=2, dilation=2, expect_mfma=False),
    # linear layer as 1x1 conv on 1x1 images
    dict(x_shape=(5, 32, 1, 1), w_shape=(10, 32, 1, 1), expect_mfma=False),
]


# pointwise (1x1 stride 1) geometries for the code-domain kernels: (kwargs, what they cover)
QGEMM_PW_CASES = [
    # tiny: one chunk, partly masked pixels (3*16 = 48 < 64), Cg=4 padded to 32, Mg=4
    dict(x_shape=(3, 16, 4, 4), w_shape=(16, 4, 1, 1), groups=4, bias=False),
    # two K-steps (Cg=40 -> Kp=64), Mg=24 (NT=2, padded rows), several chunks, images smaller than a chunk
    dict(x_shape=(5, 80, 4, 8), w_shape=(48, 40, 1, 1), groups=2),
    # Mg=72 -> two m-blocks of 64, Cg=32, 8x8 images (1 chunk per image)
    dict(x_shape=(3, 32, 8, 8), w_shape=(72, 32, 1, 1)),
    # Mg=10 (single 16-row tile), Cg=96 (3 K-steps): the nin_gc L9 pattern
    dict(x_shape=(2, 96, 8, 8), w_shape=(10, 96, 1, 1)),
]

# k x k geometries for the code-domain kernels (tile = 256 output pixels = whole rows)
QGEMM_KXK_CASES = [
    # 3x3 pad 1 grouped, 8x8 images (4 images per tile, N=3 -> masked slot), Cg=16, Mg=32: the nin_gc L7 pattern
    dict(x_shape=(3, 32, 8, 8), w_shape=(64, 16, 3, 3), padding=1, groups=2),
    # 16x16 image = one tile, Cg=6 (padded to 16), Mg=40 (NT=4 with masked rows)
    dict(x_shape=(2, 6, 16, 16), w_shape=(40, 6, 3, 3), padding=1),
    # 5x5 pad 2, Cin=3: the first-layer pattern (iao)
    dict(x_shape=(2, 3, 8, 8), w_shape=(24, 3, 5, 5), padding=2),
    # dilation 2
    dict(x_shape=(2, 4, 8, 8), w_shape=(8, 4, 3, 3), padding=2, dilation=2),
    # two channel chunks (Cg=40), 32-wide rows (8 rows per tile, 2 tiles per image)
    dict(x_shape=(1, 40, 16, 32), w_shape=(16, 40, 3, 3), padding=1, bias=False),
    # stride 2 (forward only in the code domain)
    dict(x_shape=(2, 8, 16, 16), w_shape=(24, 8, 3, 3), stride=2, padding=1, bias=False),
]


def check_adam(be, sizes=(5, 4099, 2048, 1), steps=3, lr=0.01, wd=1e-5, seed=0):
    """mn_adam_step vs torch.optim.Adam (CPU, fp32) on the same parameters / gradients for a few steps."""
    import torch
    from micronet_amd import _lib
    r = np.random.default_rng(seed)
    ps = [r.standard_normal(n).astype(F) for n in sizes]
    tp = [torch.nn.Parameter(torch.from_numpy(p.copy())) for p in ps]
    opt = torch.optim.Adam([{"params": [t], "lr": lr * (1 + i), "weight_decay": wd * i} for i, t in enumerate(tp)], lr=lr)
    dp = [be.to_dev(p) for p in ps]
    dm = [be.to_dev(np.zeros_like(p)) for p in ps]
    dv = [be.to_dev(np.zeros_like(p)) for p in ps]
    for step in range(1, steps + 1):
        gs = [(r.standard_normal(n) * 0.1).astype(F) for n in sizes]
        for t, g in zip(tp, gs):
            t.grad = torch.from_numpy(g.copy())
        opt.step()
        dg = [be.to_dev(g) for g in gs]
        arr = (_lib.AdamTensor * len(sizes))()
        for i, n in enumerate(sizes):
            arr[i] = _lib.AdamTensor(be.ptr(dp[i]).value, be.ptr(dg[i]).value, be.ptr(dm[i]).value, be.ptr(dv[i]).value, n,
                                     lr * (1 + i), wd * i)
        be.call("mn_adam_step", arr, len(sizes), step, 0.9, 0.999, 1e-8, be.stream)
        for i, t in enumerate(tp):
            got, ref = be.to_host(dp[i]), t.detach().numpy()
            assert np.max(np.abs(got - ref)) <= 2e-6 * max(1.0, np.max(np.abs(ref))), (step, i, np.max(np.abs(got - ref)))


def check_bnsign(be, shape=(6, 5, 4, 8), seed=0, training=True):
    """mn_bnsign_fwd/bwd vs an fp64 numpy evaluation of BatchNorm2d (batch or running statistics) + BinaryActivation."""
    r = np.random.default_rng(seed)
    N, Cc, H, W = shape
    HW = H * W
    y = (r.standard_normal(shape) * 1.5 + r.standard_normal((1, Cc, 1, 1))).astype(F)
    gamma, beta = (r.standard_normal(Cc) * 0.5 + 1).astype(F), (r.standard_normal(Cc) * 0.3).astype(F)
    rm, rv = (r.standard_normal(Cc) * 0.1).astype(F), (np.abs(r.standard_normal(Cc)) + 0.5).astype(F)
    da = r.standard_normal(shape).astype(F)
    eps, mom = 1e-5, 0.1
    y64 = y.astype(np.float64)
    n = N * HW
    if training:
        mean = y64.mean(axis=(0, 2, 3)); var_b = y64.var(axis=(0, 2, 3)); var_u = var_b * n / (n - 1)
    else:
        mean, var_b = rm.astype(np.float64), rv.astype(np.float64)
    invstd = 1.0 / np.sqrt(var_b + eps)
    zh = (y64 - mean.reshape(1, -1, 1, 1)) * invstd.reshape(1, -1, 1, 1)
    z = zh * gamma.reshape(1, -1, 1, 1) + beta.reshape(1, -1, 1, 1)
    a_ref = np.where(z < 0, -1.0, 1.0)
    dz = np.where((z > -1) & (z < 1), da.astype(np.float64), 0.0)
    dbeta_ref, dgamma_ref = dz.sum(axis=(0, 2, 3)), (dz * zh).sum(axis=(0, 2, 3))
    gi = (gamma * invstd).reshape(1, -1, 1, 1)
    if training:
        dy_ref = gi * (dz - dbeta_ref.reshape(1, -1, 1, 1) / n - zh * dgamma_ref.reshape(1, -1, 1, 1) / n)
    else:
        dy_ref = gi * dz
    dY, dG, dB, dRM, dRV, dDA = be.to_dev(y), be.to_dev(gamma), be.to_dev(beta), be.to_dev(rm), be.to_dev(rv), be.to_dev(da)
    save, a, dy, dgam, dbet = be.empty((2, Cc)), be.empty(shape), be.empty(shape), be.empty(Cc), be.empty(Cc)
    ws = be.empty(int(be.lib.mn_bnsign_ws_floats(Cc)) + 2)
    be.call("mn_bnsign_fwd", be.ptr(dY), N, Cc, HW, be.ptr(dG), be.ptr(dB), eps, mom, int(training), be.ptr(dRM), be.ptr(dRV),
            be.ptr(save), be.ptr(a), be.ptr(ws), be.stream)
    be.call("mn_bnsign_bwd", be.ptr(dDA), be.ptr(dY), be.ptr(save), be.ptr(dG), be.ptr(dB), N, Cc, HW, int(training), be.ptr(dy),
            be.ptr(dgam), be.ptr(dbet), be.ptr(ws), be.stream)
    a8 = be.empty_i8(shape)                      # the packed (int8) output must hold the same signs
    be.call("mn_bnsign_fwd_i8", be.ptr(dY), N, Cc, HW, be.ptr(dG), be.ptr(dB), eps, 0.0, int(training), be.ptr(dRM), be.ptr(dRV),
            be.ptr(be.empty((2, Cc))), be.ptr(a8), be.ptr(ws), be.stream)
    a_got = be.to_host(a)
    assert np.array_equal(be.to_host(a8).astype(F), a_got)
    safe = np.abs(z) > 1e-5                      # away from the sign tie
    assert np.array_equal(a_got[safe], a_ref[safe]) and np.all(np.abs(a_got) == 1.0)
    sv = be.to_host(save)
    assert np.max(np.abs(sv[0] - mean)) <= 1e-6 * max(1.0, np.max(np.abs(mean))) and np.max(np.abs(sv[1] - invstd) / invstd) <= 2e-6
    if training:
        assert np.max(np.abs(be.to_host(dRM) - ((1 - mom) * rm + mom * mean))) <= 1e-6
        assert np.max(np.abs(be.to_host(dRV) - ((1 - mom) * rv + mom * var_u))) <= 2e-6 * np.max(var_u)
    else:
        assert eq(be.to_host(dRM), rm) and eq(be.to_host(dRV), rv)
    edge = (np.abs(np.abs(z) - 1) < 1e-5).any()
    tol = 1e-3 if edge else 1e-5
    assert close(be.to_host(dbet), dbeta_ref, tol) and close(be.to_host(dgam), dgamma_ref, tol)
    assert close(be.to_host(dy), dy_ref, tol)


def check_bnrelu(be, shape=(6, 5, 4, 8), seed=0, training=True, plain=False):
    """mn_bnrelu_fwd/bwd vs an fp64 numpy evaluation of relu(BatchNorm2d(y)) (batch or running statistics) and its backward; plain: mn_bn2d_fwd/bwd, the
    same passes without the activation (nn.BatchNorm2d)."""
    r = np.random.default_rng(seed)
    N, Cc, H, W = shape
    HW = H * W
    y = (r.standard_normal(shape) * 1.5 + r.standard_normal((1, Cc, 1, 1))).astype(F)
    gamma, beta = (r.standard_normal(Cc) * 0.5 + 1).astype(F), (r.standard_normal(Cc) * 0.3).astype(F)
    rm, rv = (r.standard_normal(Cc) * 0.1).astype(F), (np.abs(r.standard_normal(Cc)) + 0.5).astype(F)
    da = r.standard_normal(shape).astype(F)
    eps, mom = 1e-5, 0.1
    y64 = y.astype(np.float64)
    n = N * HW
    if training:
        mean = y64.mean(axis=(0, 2, 3)); var_b = y64.var(axis=(0, 2, 3)); var_u = var_b * n / (n - 1)
    else:
        mean, var_b = rm.astype(np.float64), rv.astype(np.float64)
    invstd = 1.0 / np.sqrt(var_b + eps)
    zh = (y64 - mean.reshape(1, -1, 1, 1)) * invstd.reshape(1, -1, 1, 1)
    z = zh * gamma.reshape(1, -1, 1, 1) + beta.reshape(1, -1, 1, 1)
    a_ref = z if plain else np.maximum(z, 0.0)
    dz = da.astype(np.float64) if plain else np.where(z > 0, da.astype(np.float64), 0.0)
    dbeta_ref, dgamma_ref = dz.sum(axis=(0, 2, 3)), (dz * zh).sum(axis=(0, 2, 3))
    gi = (gamma * invstd).reshape(1, -1, 1, 1)
    dy_ref = gi * (dz - dbeta_ref.reshape(1, -1, 1, 1) / n - zh * dgamma_ref.reshape(1, -1, 1, 1) / n) if training else gi * dz
    dY, dG, dB, dRM, dRV, dDA = be.to_dev(y), be.to_dev(gamma), be.to_dev(beta), be.to_dev(rm), be.to_dev(rv), be.to_dev(da)
    save, a, dy, dgam, dbet = be.empty((2, Cc)), be.empty(shape), be.empty(shape), be.empty(Cc), be.empty(Cc)
    ws = be.empty(int(be.lib.mn_bnsign_ws_floats(Cc)) + 2)
    fn = "mn_bn2d" if plain else "mn_bnrelu"
    be.call(fn + "_fwd", be.ptr(dY), N, Cc, HW, be.ptr(dG), be.ptr(dB), eps, mom, int(training), be.ptr(dRM), be.ptr(dRV),
            be.ptr(save), be.ptr(a), be.ptr(ws), be.stream)
    be.call(fn + "_bwd", be.ptr(dDA), be.ptr(dY), be.ptr(save), be.ptr(dG), be.ptr(dB), N, Cc, HW, int(training), be.ptr(dy),
            be.ptr(dgam), be.ptr(dbet), be.ptr(ws), be.stream)
    assert close(be.to_host(a), a_ref, 2e-6) and (plain or np.all(be.to_host(a) >= 0))
    if not plain:       # the variant that leaves per-block (min, max) of its output for the next layer's IAO observer: same output, and the observer update from
        # the partials equals mn_iao_observe on the tensor itself, bit for bit
        cnt = int(be.lib.mn_bnrelu_mm_count(N, Cc, HW))
        a2, mm = be.empty(shape), be.empty(2 * cnt)
        dRM2, dRV2, save2 = be.to_dev(rm), be.to_dev(rv), be.empty((2, Cc))
        be.call("mn_bnrelu_fwd_mm", be.ptr(dY), N, Cc, HW, be.ptr(dG), be.ptr(dB), eps, mom, int(training), be.ptr(dRM2), be.ptr(dRV2), be.ptr(save2), be.ptr(a2), be.ptr(ws),
                be.ptr(mm), be.stream)
        assert np.array_equal(be.to_host(a2), be.to_host(a))
        for kind, first in ((1, 1), (1, 0), (0, 0)):
            m1, M1, m2, M2 = be.to_dev(np.array([-0.25], dtype=F)), be.to_dev(np.array([0.75], dtype=F)), be.to_dev(np.array([-0.25], dtype=F)), be.to_dev(np.array([0.75], dtype=F))
            wso = be.empty(int(be.lib.mn_iao_observe_ws_floats(1, N * Cc * HW)))
            be.call("mn_iao_observe", be.ptr(a2), 1, N * Cc * HW, kind, first, 0.1, be.ptr(m1), be.ptr(M1), be.ptr(wso), be.stream)
            be.call("mn_iao_observe_partials", be.ptr(mm), cnt, kind, first, 0.1, be.ptr(m2), be.ptr(M2), be.stream)
            assert np.array_equal(be.to_host(m1), be.to_host(m2)) and np.array_equal(be.to_host(M1), be.to_host(M2)), (kind, first)
            # + the quantizer's update_qparams in the same launch == mn_iao_observe_partials then mn_iao_qparams
            m3, M3 = be.to_dev(np.array([-0.25], dtype=F)), be.to_dev(np.array([0.75], dtype=F))
            sc1, zp1, qp1, sc3, zp3, qp3 = be.empty(1), be.empty(1), be.empty((1, 4)), be.empty(1), be.empty(1), be.empty((1, 4))
            be.call("mn_iao_qparams", be.ptr(m2), be.ptr(M2), 1, 4, 0, 1, 1, be.ptr(sc1), be.ptr(zp1), be.ptr(qp1), be.stream)
            be.call("mn_iao_observe_partials_qparams", be.ptr(mm), cnt, kind, first, 0.1, be.ptr(m3), be.ptr(M3), 4, 0, 1, be.ptr(sc3), be.ptr(zp3), be.ptr(qp3), be.stream)
            for u, v in ((m2, m3), (M2, M3), (sc1, sc3), (zp1, zp3), (qp1, qp3)):
                assert np.array_equal(be.to_host(u), be.to_host(v))
    sv = be.to_host(save)
    assert np.max(np.abs(sv[0] - mean)) <= 1e-6 * max(1.0, np.max(np.abs(mean))) and np.max(np.abs(sv[1] - invstd) / invstd) <= 2e-6
    if training:
        assert np.max(np.abs(be.to_host(dRM) - ((1 - mom) * rm + mom * mean))) <= 1e-6
        assert np.max(np.abs(be.to_host(dRV) - ((1 - mom) * rv + mom * var_u))) <= 2e-6 * np.max(var_u)
    edge = (not plain) and (np.abs(z) < 1e-5).any()              # an element within rounding of the ReLU kink
    tol = 1e-3 if edge else 1e-5
    assert close(be.to_host(dbet), dbeta_ref, tol) and close(be.to_host(dgam), dgamma_ref, tol)
    assert close(be.to_host(dy), dy_ref, tol)


def check_pool_sign8(be, shape=(3, 5, 8, 16), seed=0):
    """mn_maxpool2x2_sign8_fwd/bwd vs torch CPU max_pool2d (forward values and the gradient routing to the first maximum)."""
    import torch
    r = np.random.default_rng(seed)
    a = np.where(r.standard_normal(shape) > 0.3, 1, -1).astype(np.int8)       # mostly -1: windows with 0, 1, several +1
    N, Cc, H, W = shape
    t = torch.from_numpy(a.astype(F)).requires_grad_(True)
    out = torch.nn.functional.max_pool2d(t, 2, 2)
    g = r.standard_normal(tuple(out.shape)).astype(F)
    out.backward(torch.from_numpy(g))
    dA, dG = be.to_dev_i8(a), be.to_dev(g)
    o8, din = be.empty_i8((N, Cc, H // 2, W // 2)), be.empty(shape)
    be.call("mn_maxpool2x2_sign8_fwd", be.ptr(dA), N * Cc, H, W, be.ptr(o8), be.stream)
    be.call("mn_maxpool2x2_sign8_bwd", be.ptr(dG), be.ptr(dA), N * Cc, H, W, be.ptr(din), be.stream)
    assert np.array_equal(be.to_host(o8).astype(F), out.detach().numpy())
    assert np.array_equal(be.to_host(din), t.grad.numpy())


def _check_pwb(be, g, wq, d_da, h8, a8, chan, sums, training, dW, dA, dx_ref, dw_ref, db_ref, db_tol, x_shape, w_shape, Oc):
    """mn_conv2d_bwd_bnh (k_pwb: backward-data and backward-weight of the block in one launch, (da, h) read once) against the two-step path's results."""
    if not be.lib.mn_conv2d_bwd_bnh_supported(C.byref(g), C.byref(wq), 1 if a8 is not None else 0):
        return
    nb = int(be.lib.mn_conv2d_bwd_bnh_ws_bytes(C.byref(g)))
    assert nb > 0
    for with_bias in (True, False):
        ws, dx3, dw3, db3 = be.empty(nb // 4 + 4), be.empty(x_shape), be.empty(w_shape), be.empty(Oc)
        be.call("mn_conv2d_bwd_bnh", C.byref(g), C.byref(wq), be.ptr(d_da), be.ptr(h8), be.ptr(a8) if a8 is not None else None, be.ptr(chan), be.ptr(sums),
                int(training), be.ptr(dW), be.ptr(dA), be.ptr(dx3), be.ptr(dw3), be.ptr(db3) if with_bias else None, be.ptr(ws), nb, be.stream)
        assert be.lib.mn_last_kernel().decode() == ("k_pwb<2, 0, 0>" if a8 is not None else "k_pwb<1, 0, 0>")
        assert close(be.to_host(dx3), dx_ref, 5e-6), ("k_pwb dx", np.max(np.abs(be.to_host(dx3) - dx_ref)) / np.max(np.abs(dx_ref)))
        assert close(be.to_host(dw3), dw_ref, 5e-6), ("k_pwb dw", np.max(np.abs(be.to_host(dw3) - dw_ref)) / np.max(np.abs(dw_ref)))
        if with_bias:
            assert np.max(np.abs(be.to_host(db3) - db_ref)) <= db_tol, "k_pwb dbias"
    # ... with the BatchNorm-backward sums of an upstream BatchNorm+sign block as a by-product (mn_conv2d_bwd_bnh_up): dx / dw bit-identical to the plain launch, the
    # finished sums equal to mn_bnh_bwd_sums on (dx, upstream stash) up to the summation order
    N_, Cin, H_, W_ = x_shape
    K_up = 128
    splits = int(be.lib.mn_conv2d_bwd_bnh_up_splits(C.byref(g), C.byref(wq), 1 if a8 is not None else 0, K_up))
    assert splits > 0 and int(be.lib.mn_conv2d_bwd_bnh_up_splits(C.byref(g), C.byref(wq), 1 if a8 is not None else 0, 255)) == 0
    r = np.random.default_rng(1234 + Cin + N_)
    nnz = r.integers(K_up // 2, K_up + 1, Cin).astype(F)
    up_h = np.floor(r.random(x_shape) * (nnz.reshape(1, -1, 1, 1) + 1)).astype(np.uint8)
    flip = np.where(r.random(Cin) < 0.5, -1.0, 1.0).astype(F)
    Lc, Uc = -np.floor(r.random(Cin) * 40).astype(F), np.floor(r.random(Cin) * 40).astype(F)
    Lc[1], Uc[1] = 5.0, -5.0                      # an empty interval
    Lc[2], Uc[2] = -1e9, 1e9                      # everything passes
    Lc[3] = np.nan                                # a poisoned channel: nothing passes
    Lc[4], Uc[4] = 300.0, 400.0                   # beyond every admissible value
    up_chan = np.stack([np.zeros(Cin, F), flip, Lc, Uc, (r.standard_normal(Cin) * 0.05).astype(F), (r.standard_normal(Cin) * 0.3).astype(F), np.ones(Cin, F), nnz]).astype(F)
    d_uh, d_uc = be.to_dev_u8(up_h), be.to_dev(up_chan)
    ws, dx4, dw4, db4 = be.empty(nb // 4 + 4), be.empty(x_shape), be.empty(w_shape), be.empty(Oc)
    part = be.empty(Cin * splits * 4 + 2)          # [C][splits][2] doubles
    be.call("mn_conv2d_bwd_bnh_up", C.byref(g), C.byref(wq), be.ptr(d_da), be.ptr(h8), be.ptr(a8) if a8 is not None else None, be.ptr(chan), be.ptr(sums),
            int(training), be.ptr(dW), be.ptr(dA), be.ptr(dx4), be.ptr(dw4), be.ptr(db4), be.ptr(ws), nb, be.ptr(d_uh), be.ptr(d_uc), be.ptr(part), be.stream)
    assert be.lib.mn_last_kernel().decode() == ("k_pwb<2, 0, 0, 1>" if a8 is not None else "k_pwb<1, 0, 0, 1>")
    assert np.array_equal(be.to_host(dx4), be.to_host(dx3)) and np.array_equal(be.to_host(dw4), be.to_host(dw3))
    s_up, dg_up, db_up = be.empty((2, Cin)), be.empty(Cin), be.empty(Cin)
    be.call("mn_bnh_bwd_sums_final", be.ptr(part), splits, N_, Cin, H_, W_, be.ptr(dg_up), be.ptr(db_up), be.ptr(s_up), be.stream)
    s_ref, dg_ref, db_ref2 = be.empty((2, Cin)), be.empty(Cin), be.empty(Cin)
    ws2 = be.empty(int(be.lib.mn_bnsign_ws_floats(Cin)) + 2)
    be.call("mn_bnh_bwd_sums", be.ptr(dx4), be.ptr(d_uh), None, be.ptr(d_uc), N_, Cin, H_, W_, be.ptr(dg_ref), be.ptr(db_ref2), be.ptr(s_ref), be.ptr(ws2), be.stream)
    got, ref = be.to_host(s_up).astype(np.float64), be.to_host(s_ref).astype(np.float64)
    # the sums cancel: judge against the sum of magnitudes (an fp64 evaluation from the same dx)
    dxh = be.to_host(dx4).astype(np.float64)
    acc = 2.0 * up_h.astype(np.float64) - nnz.reshape(1, -1, 1, 1)
    u = acc * flip.reshape(1, -1, 1, 1)
    with np.errstate(invalid="ignore"):
        mask = (u >= Lc.reshape(1, -1, 1, 1)) & (u <= Uc.reshape(1, -1, 1, 1))
    dz = np.where(mask, dxh, 0.0)
    zh = acc * up_chan[4].astype(np.float64).reshape(1, -1, 1, 1) + up_chan[5].astype(np.float64).reshape(1, -1, 1, 1)
    e1, e2 = dz.sum(axis=(0, 2, 3)), (dz * zh).sum(axis=(0, 2, 3))
    m1, m2 = np.abs(dz).sum(axis=(0, 2, 3)) + 1e-30, np.abs(dz * zh).sum(axis=(0, 2, 3)) + 1e-30
    assert np.max(np.abs(got[0] - e1) / m1) <= 2e-6 and np.max(np.abs(got[1] - e2) / m2) <= 2e-6, ("k_pwb upstream sums", np.max(np.abs(got[0] - e1) / m1), np.max(np.abs(got[1] - e2) / m2))
    assert np.max(np.abs(ref[0] - e1) / m1) <= 2e-6 and np.max(np.abs(ref[1] - e2) / m2) <= 2e-6
    assert np.all(got[:, 1] == 0) and np.all(got[:, 3] == 0) and np.all(got[:, 4] == 0)
    assert np.array_equal(be.to_host(dg_up), be.to_host(s_up)[1]) and np.array_equal(be.to_host(db_up), be.to_host(s_up)[0])
    if a8 is None and W_ >= 8 and (W_ & (W_ - 1)) == 0:
        _check_pwb_up9(be, g, wq, d_da, h8, chan, sums, training, dW, dA, dx3, dw3, nb, x_shape, w_shape, Oc)
    _check_pwb.count = getattr(_check_pwb, "count", 0) + 1


def _check_pwb_up9(be, g, wq, d_da, h8, chan, sums, training, dW, dA, dx3, dw3, nb, x_shape, w_shape, Oc):
    """mn_conv2d_bwd_bnh_up9 (k_pwb<1, 0, 0, 3>): the block in front is a 3x3 / padding-1 BatchNorm+sign block -- [17][C] constants, the stash offset nnz per border
    class of the pixel.  dx / dw bit-identical to the plain launch; the finished sums against mn_bnh_bwd_sums on (dx, upstream stash) and an fp64 evaluation."""
    N_, Cin, H_, W_ = x_shape
    K_up = 144
    splits = int(be.lib.mn_conv2d_bwd_bnh_up9_splits(C.byref(g), C.byref(wq), K_up))
    assert splits > 0 and int(be.lib.mn_conv2d_bwd_bnh_up9_splits(C.byref(g), C.byref(wq), 255)) == 0
    r = np.random.default_rng(4321 + Cin + N_ + W_)
    # nnz of the nine classes: the full count in the middle, fewer taps at the borders (what k_row_nnz9 leaves: any non-negative integers serve the arithmetic)
    full = r.integers(K_up // 2, K_up + 1, Cin)
    nnz9 = np.stack([np.floor(full * f).astype(np.int64) for f in (4 / 9, 6 / 9, 4 / 9, 6 / 9, 1.0, 6 / 9, 4 / 9, 6 / 9, 4 / 9)]).astype(F)       # [9][C]
    rc = np.where(np.arange(H_) == 0, 0, np.where(np.arange(H_) == H_ - 1, 2, 1))
    cc = np.where(np.arange(W_) == 0, 0, np.where(np.arange(W_) == W_ - 1, 2, 1))
    cls = 3 * rc[:, None] + cc[None, :]                                       # [H][W]
    nnz_px = nnz9[cls.reshape(-1)].reshape(H_, W_, Cin).transpose(2, 0, 1)[None]      # [1][C][H][W]
    up_h = np.floor(r.random(x_shape) * (nnz_px + 1)).astype(np.uint8)
    flip = np.where(r.random(Cin) < 0.5, -1.0, 1.0).astype(F)
    Lc, Uc = -np.floor(r.random(Cin) * 40).astype(F), np.floor(r.random(Cin) * 40).astype(F)
    Lc[1], Uc[1] = 5.0, -5.0                      # an empty interval
    Lc[2], Uc[2] = -1e9, 1e9                      # everything passes
    Lc[3] = np.nan                                # a poisoned channel: nothing passes
    up_chan = np.zeros((17, Cin), F)
    up_chan[1], up_chan[2], up_chan[3] = flip, Lc, Uc
    up_chan[4], up_chan[5], up_chan[6], up_chan[7] = (r.standard_normal(Cin) * 0.05).astype(F), (r.standard_normal(Cin) * 0.3).astype(F), 1.0, -1.0
    up_chan[8:17] = nnz9
    d_uh, d_uc = be.to_dev_u8(up_h), be.to_dev(up_chan)
    ws, dx4, dw4, db4 = be.empty(nb // 4 + 4), be.empty(x_shape), be.empty(w_shape), be.empty(Oc)
    part = be.empty(Cin * splits * 4 + 2)
    be.call("mn_conv2d_bwd_bnh_up9", C.byref(g), C.byref(wq), be.ptr(d_da), be.ptr(h8), be.ptr(chan), be.ptr(sums), int(training), be.ptr(dW), be.ptr(dA), be.ptr(dx4),
            be.ptr(dw4), be.ptr(db4), be.ptr(ws), nb, be.ptr(d_uh), be.ptr(d_uc), be.ptr(part), be.stream)
    assert be.lib.mn_last_kernel().decode() == "k_pwb<1, 0, 0, 3>"
    assert np.array_equal(be.to_host(dx4), be.to_host(dx3)) and np.array_equal(be.to_host(dw4), be.to_host(dw3))
    s_up, dg_up, db_up = be.empty((2, Cin)), be.empty(Cin), be.empty(Cin)
    be.call("mn_bnh_bwd_sums_final", be.ptr(part), splits, N_, Cin, H_, W_, be.ptr(dg_up), be.ptr(db_up), be.ptr(s_up), be.stream)
    s_ref, dg_ref, db_ref2 = be.empty((2, Cin)), be.empty(Cin), be.empty(Cin)
    ws2 = be.empty(int(be.lib.mn_bnsign_ws_floats(Cin)) + 2)
    be.call("mn_bnh_bwd_sums", be.ptr(dx4), be.ptr(d_uh), None, be.ptr(d_uc), N_, Cin, H_, W_, be.ptr(dg_ref), be.ptr(db_ref2), be.ptr(s_ref), be.ptr(ws2), be.stream)
    got, ref = be.to_host(s_up).astype(np.float64), be.to_host(s_ref).astype(np.float64)
    dxh = be.to_host(dx4).astype(np.float64)
    acc = 2.0 * up_h.astype(np.float64) - nnz_px
    u = acc * flip.reshape(1, -1, 1, 1)
    with np.errstate(invalid="ignore"):
        mask = (u >= Lc.reshape(1, -1, 1, 1)) & (u <= Uc.reshape(1, -1, 1, 1))
    dz = np.where(mask, dxh, 0.0)
    zh = acc * up_chan[4].astype(np.float64).reshape(1, -1, 1, 1) + up_chan[5].astype(np.float64).reshape(1, -1, 1, 1)
    e1, e2 = dz.sum(axis=(0, 2, 3)), (dz * zh).sum(axis=(0, 2, 3))
    m1, m2 = np.abs(dz).sum(axis=(0, 2, 3)) + 1e-30, np.abs(dz * zh).sum(axis=(0, 2, 3)) + 1e-30
    e_got = (np.max(np.abs(got[0] - e1) / m1), np.max(np.abs(got[1] - e2) / m2))
    assert e_got[0] <= 2e-6 and e_got[1] <= 2e-6, ("k_pwb<UP 3> upstream sums", e_got)
    assert np.max(np.abs(ref[0] - e1) / m1) <= 2e-6 and np.max(np.abs(ref[1] - e2) / m2) <= 2e-6
    assert np.all(got[:, 1] == 0) and np.all(got[:, 3] == 0)
    assert np.array_equal(be.to_host(dg_up), be.to_host(s_up)[1]) and np.array_equal(be.to_host(db_up), be.to_host(s_up)[0])
    _check_pwb_up9.count = getattr(_check_pwb_up9, "count", 0) + 1


def check_qconv_bnsign(be, x_shape, w_shape, groups=1, bias=True, in_shuffle=0, training=True, seed=0, pooled=False, stash=False, padding=0, **_):
    """mn_qconv_bnsign_fwd/bwd (conv + BatchNorm + sign on packed codes; y never stored) vs an fp64 numpy evaluation of the
    same block on the same +-1 input and ternary-coded weights."""
    r = np.random.default_rng(seed)
    N, Cin, H, W = x_shape
    Oc = w_shape[0]
    HW = H * W
    a_in = np.where(r.standard_normal(x_shape) > 0, 1, -1).astype(np.int8)
    w, wkw, _ = make_coded_weights(r, w_shape, 1)
    b = (r.standard_normal(Oc) * 0.2).astype(F) if bias else None
    gamma, beta = (r.standard_normal(Oc) * 0.5 + 1).astype(F), (r.standard_normal(Oc) * 0.3).astype(F)
    rm, rv = (r.standard_normal(Oc) * 0.1).astype(F), (np.abs(r.standard_normal(Oc)) * 20 + 5).astype(F)
    da = r.standard_normal((N, Oc, H, W)).astype(F)
    eps, mom = 1e-5, 0.1
    x_log = a_in.astype(F)
    if in_shuffle > 1:
        x_log = np.ascontiguousarray(x_log.reshape(N, in_shuffle, Cin // in_shuffle, H, W).transpose(0, 2, 1, 3, 4).reshape(x_shape))
    y64 = O.conv2d_fwd(x_log, w, b, padding=padding, groups=groups).astype(np.float64)      # exact: integer sums times alpha (+ bias), rounded once to fp32
    n = N * HW
    if training:
        mean = y64.mean(axis=(0, 2, 3)); var_b = y64.var(axis=(0, 2, 3)); var_u = var_b * n / (n - 1)
    else:
        mean, var_b = rm.astype(np.float64), rv.astype(np.float64)
    invstd = 1.0 / np.sqrt(var_b + eps)
    zh = (y64 - mean.reshape(1, -1, 1, 1)) * invstd.reshape(1, -1, 1, 1)
    z = zh * gamma.reshape(1, -1, 1, 1) + beta.reshape(1, -1, 1, 1)
    a_ref = np.where(z < 0, -1, 1)
    dz = np.where((z > -1) & (z < 1), da.astype(np.float64), 0.0)
    dbeta_ref, dgamma_ref = dz.sum(axis=(0, 2, 3)), (dz * zh).sum(axis=(0, 2, 3))
    gi = (gamma * invstd).reshape(1, -1, 1, 1)
    dy_ref = gi * (dz - dbeta_ref.reshape(1, -1, 1, 1) / n - zh * dgamma_ref.reshape(1, -1, 1, 1) / n) if training else gi * dz

    g = be.geom(x_shape, w_shape, padding=padding, groups=groups)
    g.in_shuffle = in_shuffle
    wq = be.wq(**wkw)
    if stash:       # the stash forward also covers k x k convolutions (same-size output: stride 1, padding = (k - 1) / 2)
        assert be.lib.mn_qconv_bnsign_stash_supported(C.byref(g), C.byref(wq)) == 1
        nb = max(int(be.lib.mn_qconv_bnsign_stash_ws_bytes(C.byref(g))), 4 * int(be.lib.mn_bnsign_ws_floats(Oc)))
    else:
        assert be.lib.mn_qconv_bnsign_supported(C.byref(g), C.byref(wq)) == 1
        nb = int(be.lib.mn_qconv_bnsign_ws_bytes(C.byref(g)))
    ws = be.empty(nb // 4 + 8)
    dA, dW, dB = be.to_dev_i8(a_in), be.to_dev(w), (be.to_dev(b) if bias else None)
    dG, dBe, dRM, dRV, dDA = be.to_dev(gamma), be.to_dev(beta), be.to_dev(rm), be.to_dev(rv), be.to_dev(da)
    save, a8 = be.empty((2, Oc)), be.empty_i8((N, Oc, H, W))
    if stash:      # forward that also stashes the integer conv result in one byte per element + the per-channel constants
        h8, chan = be.empty_i8((N, Oc, H, W)), be.empty((int(be.lib.mn_qconv_bnsign_stash_chan_rows(C.byref(g))), Oc))
        nbt = be.to_dev_i64([41])
        be.call("mn_qconv_bnsign_fwd_stash", C.byref(g), C.byref(wq), be.ptr(dA), be.ptr(dW), be.ptr(dB), be.ptr(dG), be.ptr(dBe), eps, mom, int(training),
                be.ptr(dRM), be.ptr(dRV), be.ptr(nbt), be.ptr(save), be.ptr(a8), be.ptr(h8), be.ptr(chan), be.ptr(ws), nb, be.stream)
        check_qconv_bnsign.last_fwd_kernel = be.lib.mn_last_kernel().decode()
        assert int(be.to_host(nbt)[0]) == (42 if training else 41)         # BatchNorm's forward counter: incremented by the statistics launch
        if training and w_shape[2] == 1 and H % 2 == 0 and W % 16 == 0 and be.lib.mn_qconv_bnsign_fwd_stash_pool_supported(C.byref(g), C.byref(wq)):
            # the same forward with the POOLED sign codes written by the sign pass (a 2x2 / stride-2 max-pool behind the block): every output bit for bit, and
            # a_pool == mn_maxpool2x2_sign8_fwd(a)
            dRM2, dRV2, nbt2 = be.to_dev(rm), be.to_dev(rv), be.to_dev_i64([41])
            save2, a82, h82, chan2 = be.empty((2, Oc)), be.empty_i8((N, Oc, H, W)), be.empty_i8((N, Oc, H, W)), be.empty((8, Oc))
            ap = be.empty_i8((N, Oc, H // 2, W // 2))
            ws_p = be.empty(nb // 4 + 8)
            be.call("mn_qconv_bnsign_fwd_stash_pool", C.byref(g), C.byref(wq), be.ptr(dA), be.ptr(dW), be.ptr(dB), be.ptr(dG), be.ptr(dBe), eps, mom, 1,
                    be.ptr(dRM2), be.ptr(dRV2), be.ptr(nbt2), be.ptr(save2), be.ptr(a82), be.ptr(ap), be.ptr(h82), be.ptr(chan2), be.ptr(ws_p), nb, be.stream)
            assert np.array_equal(be.to_host(a82), be.to_host(a8)) and np.array_equal(be.to_host(h82), be.to_host(h8)) and np.array_equal(be.to_host(chan2), be.to_host(chan))
            assert np.array_equal(be.to_host(save2), be.to_host(save)) and np.array_equal(be.to_host(dRM2), be.to_host(dRM)) and int(be.to_host(nbt2)[0]) == 42
            o8 = be.empty_i8((N, Oc, H // 2, W // 2))
            be.call("mn_maxpool2x2_sign8_fwd", be.ptr(a8), N * Oc, H, W, be.ptr(o8), be.stream)
            assert np.array_equal(be.to_host(ap), be.to_host(o8)), "pooled codes from the sign pass"
            check_qconv_bnsign.pool_sign_checked = getattr(check_qconv_bnsign, "pool_sign_checked", 0) + 1
        acc_ref = O.conv2d_fwd(x_log, np.sign(w).astype(F), None, padding=padding, groups=groups)
        # nnz per pixel: the non-zero weights that meet a non-zero input (= all of them, except at the zero-padded border of a 3x3 block)
        nnz = O.conv2d_fwd(np.ones_like(x_log), (w != 0).astype(F), None, padding=padding, groups=groups)
        assert np.array_equal((acc_ref + nnz) % 2, np.zeros_like(acc_ref))
        assert np.array_equal(be.to_host(h8).view(np.uint8).astype(np.int64), ((acc_ref + nnz) / 2).astype(np.int64))
    else:
        be.call("mn_qconv_bnsign_fwd", C.byref(g), C.byref(wq), be.ptr(dA), be.ptr(dW), be.ptr(dB), be.ptr(dG), be.ptr(dBe), eps, mom, int(training),
                be.ptr(dRM), be.ptr(dRV), be.ptr(save), be.ptr(a8), be.ptr(ws), nb, be.stream)
    dy, dgam, dbet = be.empty((N, Oc, H, W)), be.empty(Oc), be.empty(Oc)
    if pooled:
        # a 2x2 max-pool behind the block: the kernel gets the POOLED gradient + the block's own output codes; the reference routes
        # it through torch's max_pool2d backward on those same codes and then takes the ordinary path
        import torch
        gp = r.standard_normal((N, Oc, H // 2, W // 2)).astype(F)
        t_ = torch.from_numpy(be.to_host(a8).astype(F)).requires_grad_(True)
        torch.nn.functional.max_pool2d(t_, 2, 2).backward(torch.from_numpy(gp))
        da = t_.grad.numpy().astype(F)
        dz = np.where((z > -1) & (z < 1), da.astype(np.float64), 0.0)
        dbeta_ref, dgamma_ref = dz.sum(axis=(0, 2, 3)), (dz * zh).sum(axis=(0, 2, 3))
        dy_ref = gi * (dz - dbeta_ref.reshape(1, -1, 1, 1) / n - zh * dgamma_ref.reshape(1, -1, 1, 1) / n) if training else gi * dz
        dGP = be.to_dev(gp)
        if stash:
            sums = be.empty((2, Oc))
            be.call("mn_bnh_bwd_sums", be.ptr(dGP), be.ptr(h8), be.ptr(a8), be.ptr(chan), N, Oc, H, W, be.ptr(dgam), be.ptr(dbet), be.ptr(sums), be.ptr(ws), be.stream)
            be.call("mn_bnh_bwd_apply", be.ptr(dGP), be.ptr(h8), be.ptr(a8), be.ptr(chan), be.ptr(sums), N, Oc, H, W, int(training), be.ptr(dy), be.stream)
            if be.lib.mn_conv2d_bnh_pool_supported(C.byref(g), C.byref(wq)):
                # the consumers of dy forming it themselves from (pooled gradient, own codes, h): against the two-step path on the dy written above
                aq8 = be.actq(3)
                dx_ref2 = be.to_host(be.conv_bwd_data(g, aq8, dy, dW, None, 3, wq=wq))
                dw_ref2, db_ref2 = be.conv_bwd_weight(g, aq8, dy, dA, 3, bias=True)
                nb1 = be.lib.mn_conv2d_ws_bytes(C.byref(g), 1, 0)
                ws1, dx2 = be.empty(max(4, nb1 // 4 + 4)), be.empty((N, x_shape[1], H, W))
                be.call("mn_conv2d_bwd_data_bnh_pool", C.byref(g), C.byref(wq), be.ptr(dGP), be.ptr(h8), be.ptr(a8), be.ptr(chan), be.ptr(sums), int(training), be.ptr(dW),
                        be.ptr(dx2), be.ptr(ws1), nb1, be.stream)
                nb2 = be.lib.mn_conv2d_ws_bytes(C.byref(g), 2, 0)
                ws2, dw2, db2 = be.empty(max(4, nb2 // 4 + 4)), be.empty(w.shape), be.empty(Oc)
                be.call("mn_conv2d_bwd_weight_bnh_pool", C.byref(g), be.ptr(dGP), be.ptr(h8), be.ptr(a8), be.ptr(chan), be.ptr(sums), int(training), be.ptr(dA), be.ptr(dw2),
                        be.ptr(db2), be.ptr(ws2), nb2, be.stream)
                assert close(be.to_host(dx2), dx_ref2, 5e-6), ("pooled fold dx", np.max(np.abs(be.to_host(dx2) - dx_ref2)) / np.max(np.abs(dx_ref2)))
                assert close(be.to_host(dw2), be.to_host(dw_ref2), 5e-6), "pooled fold dw"
                sc_db = max(np.max(np.abs(be.to_host(dy))) * 1e-4, 1e-30)
                assert np.max(np.abs(be.to_host(db2) - be.to_host(db_ref2))) <= sc_db * N * H * W
                check_qconv_bnsign.pool_fold_checked = getattr(check_qconv_bnsign, "pool_fold_checked", 0) + 1
                _check_pwb(be, g, wq, dGP, h8, a8, chan, sums, training, dW, dA, dx_ref2, be.to_host(dw_ref2), be.to_host(db_ref2), sc_db * N * H * W, x_shape, w.shape, Oc)
        else:
          be.call("mn_qconv_bnsign_bwd_pooled", C.byref(g), C.byref(wq), be.ptr(dA), be.ptr(dW), be.ptr(dB), be.ptr(dG), be.ptr(dBe), be.ptr(save),
                be.ptr(dGP), be.ptr(a8), int(training), be.ptr(dy), be.ptr(dgam), be.ptr(dbet), be.ptr(ws), nb, be.stream)
    elif stash:
        sums = be.empty((2, Oc))
        be.call("mn_bnh_bwd_sums", be.ptr(dDA), be.ptr(h8), None, be.ptr(chan), N, Oc, H, W, be.ptr(dgam), be.ptr(dbet), be.ptr(sums), be.ptr(ws), be.stream)
        be.call("mn_bnh_bwd_apply", be.ptr(dDA), be.ptr(h8), None, be.ptr(chan), be.ptr(sums), N, Oc, H, W, int(training), be.ptr(dy), be.stream)
        if be.lib.mn_conv2d_bnh_supported(C.byref(g), C.byref(wq)):
            # the consumers of dy forming it themselves from (da, h): against the two-step path on the dy computed above
            aq8 = be.actq(3)
            dx_ref2 = be.to_host(be.conv_bwd_data(g, aq8, dy, dW, None, 3, wq=wq))
            dw_ref2, db_ref2 = be.conv_bwd_weight(g, aq8, dy, dA, 3, bias=True)
            nb1 = be.lib.mn_conv2d_ws_bytes(C.byref(g), 1, 0)
            ws1, dx2 = be.empty(max(4, nb1 // 4 + 4)), be.empty((N, x_shape[1], H, W))
            be.call("mn_conv2d_bwd_data_bnh", C.byref(g), C.byref(wq), be.ptr(dDA), be.ptr(h8), be.ptr(chan), be.ptr(sums), int(training), be.ptr(dW),
                    be.ptr(dx2), be.ptr(ws1), nb1, be.stream)
            nb2 = be.lib.mn_conv2d_ws_bytes(C.byref(g), 2, 0)
            ws2, dw2, db2 = be.empty(max(4, nb2 // 4 + 4)), be.empty(w.shape), be.empty(Oc)
            be.call("mn_conv2d_bwd_weight_bnh", C.byref(g), be.ptr(dDA), be.ptr(h8), be.ptr(chan), be.ptr(sums), int(training), be.ptr(dA), be.ptr(dw2),
                    be.ptr(db2), be.ptr(ws2), nb2, be.stream)
            assert close(be.to_host(dx2), dx_ref2, 5e-6), np.max(np.abs(be.to_host(dx2) - dx_ref2)) / np.max(np.abs(dx_ref2))
            assert close(be.to_host(dw2), be.to_host(dw_ref2), 5e-6)
            sc_db = max(np.max(np.abs(be.to_host(dy))) * 1e-4, 1e-30)      # d bias in front of a BatchNorm is a sum that cancels to ~0
            assert np.max(np.abs(be.to_host(db2) - be.to_host(db_ref2))) <= sc_db * N * H * W
            _check_pwb(be, g, wq, dDA, h8, None, chan, sums, training, dW, dA, dx_ref2, be.to_host(dw_ref2), be.to_host(db_ref2), sc_db * N * H * W, x_shape, w.shape, Oc)
    else:
        be.call("mn_qconv_bnsign_bwd", C.byref(g), C.byref(wq), be.ptr(dA), be.ptr(dW), be.ptr(dB), be.ptr(dG), be.ptr(dBe), be.ptr(save), be.ptr(dDA),
                int(training), be.ptr(dy), be.ptr(dgam), be.ptr(dbet), be.ptr(ws), nb, be.stream)
    sv = be.to_host(save)
    assert np.max(np.abs(sv[0] - mean)) <= 2e-6 * max(1.0, np.max(np.abs(mean))) and np.max(np.abs(sv[1] - invstd) / invstd) <= 4e-6
    a_got = be.to_host(a8).astype(np.int64)
    safe = np.abs(z) > 2e-5
    assert np.all(np.abs(a_got) == 1) and np.array_equal(a_got[safe], a_ref[safe])
    if training:
        assert np.max(np.abs(be.to_host(dRM) - ((1 - mom) * rm + mom * mean))) <= 2e-6 * max(1.0, np.max(np.abs(mean)))
        assert np.max(np.abs(be.to_host(dRV) - ((1 - mom) * rv + mom * var_u))) <= 4e-6 * np.max(np.maximum(var_u, rv))
    else:
        assert eq(be.to_host(dRM), rm) and eq(be.to_host(dRV), rv)
    edge = (np.abs(np.abs(z) - 1) < 2e-5).any()
    tol = 1e-3 if edge else 2e-5
    assert close(be.to_host(dbet), dbeta_ref, tol) and close(be.to_host(dgam), dgamma_ref, tol)
    assert close(be.to_host(dy), dy_ref, tol)
    # the plain forward on sign codes (PWS_Y through mn_conv2d_fwd) against the same y
    aq = be.actq(3)
    y = be.to_host(be.conv_fwd(g, aq, dA, dW, dB, 3, wq=wq))
    assert close(y, y64, 1e-6)


def check_deployed_sign_block(be, x_shape, w_shape, groups=1, in_shuffle=0, padding=0, seed=0, must_support=True):
    """The DEPLOYED binary block (wbwtab/bn_fuse/bn_fuse.py:36-55: the BatchNorm folded into the conv's bias, the weights still codes x alpha): sign(conv(a) + b)
    on packed +-1 codes through mn_qconv_bnsign_fwd_stash in eval mode with IDENTITY statistics (gamma 1, beta 0, mean 0, var 1, eps 0) -- the fused kernels then
    evaluate ((y - 0) * 1) * 1 + 0 = y in fp32, so the result must equal sign(fl(fl(acc * alpha) + b)) (0 -> +1) BIT FOR BIT: acc is an exact integer."""
    r = np.random.default_rng(seed)
    N, Cin, H, W = x_shape
    Oc = w_shape[0]
    a_in = np.where(r.standard_normal(x_shape) > 0, 1, -1).astype(np.int8)
    w, wkw, _ = make_coded_weights(r, w_shape, 1)
    b = (r.standard_normal(Oc) * 3.0).astype(F)
    b[::5] = 0                                                # channels with a zero bias: exact zeros of acc * alpha + b occur (sign(0) = +1)
    x_log = a_in.astype(F)
    if in_shuffle > 1:
        x_log = np.ascontiguousarray(x_log.reshape(N, in_shuffle, Cin // in_shuffle, H, W).transpose(0, 2, 1, 3, 4).reshape(x_shape))
    g = be.geom(x_shape, w_shape, padding=padding, groups=groups)
    g.in_shuffle = in_shuffle
    wq = be.wq(**wkw)
    sup = int(be.lib.mn_qconv_bnsign_stash_supported(C.byref(g), C.byref(wq)))
    if not sup:
        assert not must_support, "deployed block: geometry not covered by the fused sign kernels"
        return False
    acc = O.conv2d_fwd(x_log, np.sign(w).astype(F), None, padding=padding, groups=groups)                    # exact integers
    alpha = np.abs(w).reshape(Oc, -1).max(axis=1).astype(F)
    y = (acc.astype(F) * alpha.reshape(1, -1, 1, 1)).astype(F) + b.reshape(1, -1, 1, 1)                       # the kernels' fp32 chain
    a_ref = np.where(y.astype(F) < 0, -1, 1).astype(np.int8)
    nb = max(int(be.lib.mn_qconv_bnsign_stash_ws_bytes(C.byref(g))), 4 * int(be.lib.mn_bnsign_ws_floats(Oc)))
    ws = be.empty(nb // 4 + 8)
    one, zero = np.ones(Oc, dtype=F), np.zeros(Oc, dtype=F)
    dA, dW, dB = be.to_dev_i8(a_in), be.to_dev(w), be.to_dev(b)
    dG, dBe, dRM, dRV = be.to_dev(one), be.to_dev(zero), be.to_dev(zero), be.to_dev(one)
    save, a8 = be.empty((2, Oc)), be.empty_i8((N, Oc, H, W))
    h8, chan = be.empty_i8((N, Oc, H, W)), be.empty((int(be.lib.mn_qconv_bnsign_stash_chan_rows(C.byref(g))), Oc))
    be.call("mn_qconv_bnsign_fwd_stash", C.byref(g), C.byref(wq), be.ptr(dA), be.ptr(dW), be.ptr(dB), be.ptr(dG), be.ptr(dBe), 0.0, 0.0, 0,
            be.ptr(dRM), be.ptr(dRV), None, be.ptr(save), be.ptr(a8), be.ptr(h8), be.ptr(chan), be.ptr(ws), nb, be.stream)
    sv = be.to_host(save)
    assert np.array_equal(sv[0], zero) and np.array_equal(sv[1], one), "identity statistics: mean 0, invstd exactly 1"
    got = be.to_host(a8).view(np.int8)
    assert np.array_equal(got, a_ref), ("deployed sign block", int((got != a_ref).sum()), got.size)
    # the same rule for a block fed by the fp32 first conv: sign(y) as bytes from mn_bnsign_fwd_i8 with the identity statistics
    yf = (r.standard_normal((N, Oc, H, W)) * 2).astype(F)
    yf[:, :, 0, :2] = 0
    yf[:, :, 1, :2] = -0.0
    dY = be.to_dev(yf)
    a8b = be.empty_i8((N, Oc, H, W))
    ws2 = be.empty(int(be.lib.mn_bnsign_ws_floats(Oc)) + 8)
    be.call("mn_bnsign_fwd_i8", be.ptr(dY), N, Oc, H * W, be.ptr(dG), be.ptr(dBe), 0.0, 0.0, 0, be.ptr(dRM), be.ptr(dRV), be.ptr(save), be.ptr(a8b), be.ptr(ws2), be.stream)
    assert np.array_equal(be.to_host(a8b).view(np.int8), np.where(yf < 0, -1, 1).astype(np.int8)), "sign(y) bytes (0 and -0 -> +1)"
    return True


DEPLOYED_CASES = [
    dict(x_shape=(2, 96, 8, 8), w_shape=(96, 48, 1, 1), groups=2),
    dict(x_shape=(3, 80, 8, 16), w_shape=(96, 40, 1, 1), groups=2, in_shuffle=2),
    dict(x_shape=(2, 32, 8, 8), w_shape=(64, 16, 3, 3), groups=2, padding=1),
    dict(x_shape=(2, 32, 16, 16), w_shape=(48, 8, 3, 3), groups=4, padding=1, in_shuffle=2),
    # the layers of the small golden net (tests/golden/inference_meta.json cfg 32-32-32-64-64-64-128-128) the deployment-flow test runs packed
    dict(x_shape=(4, 32, 32, 32), w_shape=(32, 16, 1, 1), groups=2),
    dict(x_shape=(4, 64, 16, 16), w_shape=(64, 16, 1, 1), groups=4, in_shuffle=4),
    dict(x_shape=(4, 128, 8, 8), w_shape=(128, 16, 1, 1), groups=8, in_shuffle=32),
]


def check_sign_classifier(be, N=3, Cc=96, H=4, W=8, Oc=10, bias=True, seed=0):
    """mn_signconv1x1_small_fwd / mn_conv1x1_small_bwd_data (+ backward-weight through mn_conv2d_bwd_weight on the codes) vs fp64."""
    r = np.random.default_rng(seed)
    a = np.where(r.standard_normal((N, Cc, H, W)) > 0, 1, -1).astype(np.int8)
    w = (r.standard_normal((Oc, Cc, 1, 1)) * 0.1).astype(F)
    b = (r.standard_normal(Oc) * 0.2).astype(F) if bias else None
    assert be.lib.mn_signconv1x1_small_supported(Cc, H * W, Oc) == 1
    y_ref = O.conv2d_fwd(a.astype(F), w, b)
    gy = r.standard_normal(y_ref.shape).astype(F)
    dx_ref, dw_ref, db_ref = O.conv2d_bwd(gy, a.astype(F), w)
    dA, dW, dB, dG = be.to_dev_i8(a), be.to_dev(w), (be.to_dev(b) if bias else None), be.to_dev(gy)
    y, dx = be.empty((N, Oc, H, W)), be.empty((N, Cc, H, W))
    be.call("mn_signconv1x1_small_fwd", be.ptr(dA), be.ptr(dW), be.ptr(dB), be.ptr(y), N, Cc, H * W, Oc, be.stream)
    be.call("mn_conv1x1_small_bwd_data", be.ptr(dG), be.ptr(dW), be.ptr(dx), N, Cc, H * W, Oc, be.stream)
    assert close(be.to_host(y), y_ref, 2e-6) and close(be.to_host(dx), dx_ref, 2e-6)
    g = be.geom((N, Cc, H, W), (Oc, Cc, 1, 1))
    dw, db = be.conv_bwd_weight(g, be.actq(3), dG, dA, 0, bias=True)
    assert close(be.to_host(dw), dw_ref, 1e-5) and close(be.to_host(db), db_ref, 1e-5)


def check_code_classifier(be, N=3, Cc=96, H=4, W=8, Oc=10, bits=2, bias=True, seed=0):
    """mn_codeconv1x1_small_fwd (k-bit activation codes) + mn_conv2d_bwd_weight with MN_ACTQ_CODE8 on a small tile (the generic pointwise kernel reading
    bytes) vs the float conv of the de-quantised activation."""
    r = np.random.default_rng(seed)
    nlev = (1 << bits) - 1
    j = r.integers(0, nlev + 1, size=(N, Cc, H, W)).astype(np.uint8)
    q = (j.astype(np.float64) / nlev).astype(F)
    w = (r.standard_normal((Oc, Cc, 1, 1)) * 0.1).astype(F)
    b = (r.standard_normal(Oc) * 0.2).astype(F) if bias else None
    y_ref = O.conv2d_fwd(q, w, b)
    gy = r.standard_normal(y_ref.shape).astype(F)
    _, dw_ref, db_ref = O.conv2d_bwd(gy, q, w)
    dJ, dW, dB, dG = be.to_dev_u8(j), be.to_dev(w), (be.to_dev(b) if bias else None), be.to_dev(gy)
    y = be.empty((N, Oc, H, W))
    be.call("mn_codeconv1x1_small_fwd", be.ptr(dJ), bits, be.ptr(dW), be.ptr(dB), be.ptr(y), N, Cc, H * W, Oc, be.stream)
    assert close(be.to_host(y), y_ref, 2e-6)
    g = be.geom((N, Cc, H, W), (Oc, Cc, 1, 1))
    dw, db = be.conv_bwd_weight(g, be.actq(4, bits), dG, dJ, 0, bias=True)
    assert close(be.to_host(dw), dw_ref, 1e-5) and close(be.to_host(db), db_ref, 1e-5)


def check_first_conv_bn_wgrad(be, x_shape=(3, 3, 8, 8), Oc=24, k=5, training=True, seed=0):
    """mn_bnsign_bwd_sums + mn_conv2d_bwd_weight_first_bn (dy formed inside the first-layer backward-weight) vs the two-step path
    mn_bnsign_bwd -> mn_conv2d_bwd_weight on the same tensors: same expressions, so the results agree to the last bit."""
    r = np.random.default_rng(seed)
    N, Cin, H, W = x_shape
    x = r.standard_normal(x_shape).astype(F)
    yb = (r.standard_normal((N, Oc, H, W)) * 1.5).astype(F)
    da = r.standard_normal((N, Oc, H, W)).astype(F)
    gamma, beta = (r.standard_normal(Oc) * 0.5 + 1).astype(F), (r.standard_normal(Oc) * 0.3).astype(F)
    mean, var = yb.mean(axis=(0, 2, 3)), yb.var(axis=(0, 2, 3))
    save = np.stack([mean, 1.0 / np.sqrt(var + 1e-5)]).astype(F)
    g = be.geom(x_shape, (Oc, Cin, k, k), padding=k // 2)
    assert be.lib.mn_conv2d_first_supported(C.byref(g), 2) == 1
    dX, dY, dDA, dS, dG, dB = be.to_dev(x), be.to_dev(yb), be.to_dev(da), be.to_dev(save), be.to_dev(gamma), be.to_dev(beta)
    HW = H * W
    ws = be.empty(int(be.lib.mn_bnsign_ws_floats(Oc)) + 2)
    dy, dgam, dbet = be.empty((N, Oc, H, W)), be.empty(Oc), be.empty(Oc)
    be.call("mn_bnsign_bwd", be.ptr(dDA), be.ptr(dY), be.ptr(dS), be.ptr(dG), be.ptr(dB), N, Oc, HW, int(training), be.ptr(dy), be.ptr(dgam), be.ptr(dbet),
            be.ptr(ws), be.stream)
    dw_ref, db_ref = be.conv_bwd_weight(g, be.actq(0), dy, dX, 0, bias=True)
    sums, dgam2, dbet2 = be.empty((2, Oc)), be.empty(Oc), be.empty(Oc)
    be.call("mn_bnsign_bwd_sums", be.ptr(dDA), be.ptr(dY), be.ptr(dS), be.ptr(dG), be.ptr(dB), N, Oc, HW, be.ptr(dgam2), be.ptr(dbet2), be.ptr(sums),
            be.ptr(ws), be.stream)
    assert eq(be.to_host(dgam2), be.to_host(dgam)) and eq(be.to_host(dbet2), be.to_host(dbet))
    nb = be.lib.mn_conv2d_ws_bytes(C.byref(g), 2, 0)
    ws2 = be.empty(max(4, nb // 4 + 4))
    dw, db = be.empty((Oc, Cin, k, k)), be.empty(Oc)
    be.call("mn_conv2d_bwd_weight_first_bn", C.byref(g), be.ptr(dDA), be.ptr(dY), be.ptr(dS), be.ptr(dG), be.ptr(dB), be.ptr(sums), int(training),
            be.ptr(dX), be.ptr(dw), be.ptr(db), be.ptr(ws2), nb, be.stream)
    assert eq(be.to_host(dw), be.to_host(dw_ref)) and eq(be.to_host(db), be.to_host(db_ref))


def check_first_conv_qa_wgrad(be, x_shape=(3, 3, 8, 8), Oc=24, k=5, training=True, quant=1, bits=2, seed=0):
    """mn_qa_bwd_sums + mn_conv2d_bwd_weight_first_qa (dy of the DoReFa block BatchNorm + ReLU + next-layer quantizer formed inside the first-layer
    backward-weight) vs the two-step path mn_qa_bwd_apply -> mn_conv2d_bwd_weight: same expressions, so the results agree to the last bit."""
    r = np.random.default_rng(seed)
    N, Cin, H, W = x_shape
    x = r.standard_normal(x_shape).astype(F)
    yb = (r.standard_normal((N, Oc, H, W)) * 1.5).astype(F)
    dq = r.standard_normal((N, Oc, H, W)).astype(F)
    gamma, beta = (r.standard_normal(Oc) * 2.0 + 3.0).astype(F), (r.standard_normal(Oc) * 2.0 + 3.0).astype(F)      # activations on both sides of the clamp's upper edge (a = 10)
    mean, var = yb.mean(axis=(0, 2, 3)), yb.var(axis=(0, 2, 3))
    save = np.stack([mean, 1.0 / np.sqrt(var + 1e-5)]).astype(F)
    g = be.geom(x_shape, (Oc, Cin, k, k), padding=k // 2)
    assert be.lib.mn_conv2d_first_supported(C.byref(g), 2) == 1
    dX, dY, dDQ, dS, dG, dB = be.to_dev(x), be.to_dev(yb), be.to_dev(dq), be.to_dev(save), be.to_dev(gamma), be.to_dev(beta)
    chan = be.empty((9, Oc))
    be.call("mn_qa_chan_from_save", be.ptr(dS), be.ptr(dG), be.ptr(dB), Oc, be.ptr(chan), be.stream)
    ws = be.empty(int(be.lib.mn_qa_ws_floats(Oc)) + 2)
    sums, dgam, dbet, dy = be.empty((2, Oc)), be.empty(Oc), be.empty(Oc), be.empty((N, Oc, H, W))
    be.call("mn_qa_bwd_sums", 1, be.ptr(dY), be.ptr(chan), be.ptr(dDQ), N, Oc, H, W, bits, 0, quant, be.ptr(dgam), be.ptr(dbet), be.ptr(sums), be.ptr(ws), be.stream)
    be.call("mn_qa_bwd_apply", 1, be.ptr(dY), be.ptr(chan), be.ptr(sums), be.ptr(dDQ), N, Oc, H, W, bits, 0, quant, int(training), be.ptr(dy), be.stream)
    dw_ref, db_ref = be.conv_bwd_weight(g, be.actq(0), dy, dX, 0, bias=True)
    nb = be.lib.mn_conv2d_ws_bytes(C.byref(g), 2, 0)
    ws2 = be.empty(max(4, nb // 4 + 4))
    dw, db = be.empty((Oc, Cin, k, k)), be.empty(Oc)
    be.call("mn_conv2d_bwd_weight_first_qa", C.byref(g), be.ptr(dDQ), be.ptr(dY), be.ptr(chan), be.ptr(sums), bits, quant, int(training),
            be.ptr(dX), be.ptr(dw), be.ptr(db), be.ptr(ws2), nb, be.stream)
    assert np.abs(be.to_host(dw_ref)).max() > 0
    assert eq(be.to_host(dw), be.to_host(dw_ref)) and eq(be.to_host(db), be.to_host(db_ref))


def _im2col_gram(x, k):
    """fp64 Gram data of the im2col rows of x ("same" padding): G [K+1][K+1] with the feature K = 1."""
    N, Cin, H, W = x.shape
    pad = k // 2
    xp = np.zeros((N, Cin, H + 2 * pad, W + 2 * pad), dtype=np.float64)
    xp[:, :, pad:pad + H, pad:pad + W] = x
    cols = [xp[:, c, r:r + H, s:s + W].reshape(-1) for c in range(Cin) for r in range(k) for s in range(k)]
    f = np.stack(cols + [np.ones(N * H * W)], axis=0)
    return f @ f.T


def check_first_conv_xgram(be, x_shape, k, seed=0):
    """mn_conv2d_first_xgram alone against the fp64 Gram data of the im2col rows (for geometries whose other kernels are covered elsewhere: here the tile loop)."""
    r = np.random.default_rng(seed)
    N, Cin, H, W = x_shape
    K_ = Cin * k * k
    x = r.standard_normal(x_shape).astype(F)
    g = be.geom(x_shape, (8, Cin, k, k), padding=k // 2)
    assert be.lib.mn_conv2d_first_supported(C.byref(g), 2) == 1
    dX = be.to_dev(x)
    nbg = int(be.lib.mn_conv2d_first_xgram_ws_bytes(C.byref(g)))
    wsg, gram = be.empty(nbg // 4 + 4), be.empty(2 * 80 * 80)
    be.call("mn_conv2d_first_xgram", C.byref(g), be.ptr(dX), be.ptr(gram), be.ptr(wsg), nbg, be.stream)
    G = be.to_host(gram).view(np.float64).reshape(80, 80)
    assert close(G[:K_ + 1, :K_ + 1], _im2col_gram(x, k), 2e-6) and G[K_, K_] == N * H * W


def check_first_conv_gram_bwd(be, x_shape=(3, 3, 8, 8), Oc=24, k=5, kind="bn", quant=1, bits=2, bias=True, seed=0, tol=2e-5):
    """The one-pass backward of the first block (mn_conv2d_first_xgram + mn_conv2d_bwd_first_bn_gram / _qa_gram: the BatchNorm backward folded into per-channel
    algebra on Gram data of x) against the two-pass path (sums, then the fold in the backward-weight's operand load) on the SAME tensors -- here y really is
    conv(x, w) + b, which the identity needs.  Different summation order: agreement to fp32 rounding, not bit for bit."""
    r = np.random.default_rng(seed)
    N, Cin, H, W = x_shape
    K = Cin * k * k
    x = r.standard_normal(x_shape).astype(F)
    w = (r.standard_normal((Oc, Cin, k, k)) * 0.2).astype(F)
    b = (r.standard_normal(Oc) * 0.3).astype(F) if bias else None
    da = r.standard_normal((N, Oc, H, W)).astype(F)
    if kind == "bn":
        gamma, beta = (r.standard_normal(Oc) * 0.5 + 1).astype(F), (r.standard_normal(Oc) * 0.3).astype(F)
    else:
        gamma, beta = (r.standard_normal(Oc) * 2.0 + 3.0).astype(F), (r.standard_normal(Oc) * 2.0 + 3.0).astype(F)
    g = be.geom(x_shape, (Oc, Cin, k, k), padding=k // 2)
    assert be.lib.mn_conv2d_first_supported(C.byref(g), 2) == 1
    dX, dW, dBi, dDA, dG, dB = be.to_dev(x), be.to_dev(w), (be.to_dev(b) if bias else None), be.to_dev(da), be.to_dev(gamma), be.to_dev(beta)
    dY = be.conv_fwd(g, be.actq(0), dX, dW, dBi, 0)
    yh = be.to_host(dY).astype(np.float64)
    mean, var = yh.mean(axis=(0, 2, 3)), yh.var(axis=(0, 2, 3))
    dS = be.to_dev(np.stack([mean, 1.0 / np.sqrt(var + 1e-5)]).astype(F))
    HW = H * W
    # Gram data
    nbg = int(be.lib.mn_conv2d_first_xgram_ws_bytes(C.byref(g)))
    assert nbg > 0
    wsg, gram = be.empty(nbg // 4 + 4), be.empty(2 * 80 * 80)
    be.call("mn_conv2d_first_xgram", C.byref(g), be.ptr(dX), be.ptr(gram), be.ptr(wsg), nbg, be.stream)
    G = be.to_host(gram).view(np.float64).reshape(80, 80)
    G_ref = _im2col_gram(x, k)
    assert close(G[:K + 1, :K + 1], G_ref, 2e-6) and G[K, K] == N * HW
    # forward statistics from the Gram data vs the statistics of y itself; the apply-only pass vs the full BatchNorm + sign forward on the same statistics
    rm, rv = be.to_dev(np.full(Oc, 0.25, dtype=F)), be.to_dev(np.full(Oc, 2.0, dtype=F))
    sv = be.empty((2, Oc))
    be.call("mn_conv2d_first_gram_bnstats", C.byref(g), be.ptr(dW), be.ptr(dBi), be.ptr(gram), 1e-5, 0.1, be.ptr(rm), be.ptr(rv), be.ptr(sv), be.stream)
    svh = be.to_host(sv)
    assert np.abs(svh[0] - mean).max() <= 2e-6 * max(np.abs(mean).max(), np.sqrt(var).max()) and np.abs(svh[1] * np.sqrt(var + 1e-5) - 1).max() <= 1e-5
    n_el = N * HW
    assert np.abs(be.to_host(rm) - (0.9 * 0.25 + 0.1 * mean)).max() <= 1e-6 and np.abs(be.to_host(rv) / (0.9 * 2.0 + 0.1 * var * n_el / (n_el - 1)) - 1).max() <= 1e-5
    if kind == "bn" and HW % 4 == 0:
        wsb = be.empty(int(be.lib.mn_bnsign_ws_floats(Oc)) + 2)
        sv2, a_full, a_app = be.empty((2, Oc)), be.empty((N, Oc, H, W)), be.empty((N, Oc, H, W))
        be.call("mn_bnsign_fwd", be.ptr(dY), N, Oc, HW, be.ptr(dG), be.ptr(dB), 1e-5, 0.1, 1, None, None, be.ptr(sv2), be.ptr(a_full), be.ptr(wsb), be.stream)
        be.call("mn_bnsign_apply", be.ptr(dY), N, Oc, HW, be.ptr(dG), be.ptr(dB), be.ptr(sv2), be.ptr(a_app), 0, be.stream)
        assert eq(be.to_host(a_app), be.to_host(a_full))
        a8 = be.empty_i8((N, Oc, H, W))
        be.call("mn_bnsign_apply", be.ptr(dY), N, Oc, HW, be.ptr(dG), be.ptr(dB), be.ptr(sv2), be.ptr(a8), 1, be.stream)
        assert eq(be.to_host(a8).astype(F), be.to_host(a_full))
    # two-pass reference
    if kind == "bn":
        ws = be.empty(int(be.lib.mn_bnsign_ws_floats(Oc)) + 2)
        dy, dgam, dbet = be.empty((N, Oc, H, W)), be.empty(Oc), be.empty(Oc)
        be.call("mn_bnsign_bwd", be.ptr(dDA), be.ptr(dY), be.ptr(dS), be.ptr(dG), be.ptr(dB), N, Oc, HW, 1, be.ptr(dy), be.ptr(dgam), be.ptr(dbet), be.ptr(ws), be.stream)
    else:
        chan = be.empty((9, Oc))
        be.call("mn_qa_chan_from_save", be.ptr(dS), be.ptr(dG), be.ptr(dB), Oc, be.ptr(chan), be.stream)
        ws = be.empty(int(be.lib.mn_qa_ws_floats(Oc)) + 2)
        sums, dgam, dbet, dy = be.empty((2, Oc)), be.empty(Oc), be.empty(Oc), be.empty((N, Oc, H, W))
        be.call("mn_qa_bwd_sums", 1, be.ptr(dY), be.ptr(chan), be.ptr(dDA), N, Oc, H, W, bits, 0, quant, be.ptr(dgam), be.ptr(dbet), be.ptr(sums), be.ptr(ws), be.stream)
        be.call("mn_qa_bwd_apply", 1, be.ptr(dY), be.ptr(chan), be.ptr(sums), be.ptr(dDA), N, Oc, H, W, bits, 0, quant, 1, be.ptr(dy), be.stream)
    dw_ref, _ = be.conv_bwd_weight(g, be.actq(0), dy, dX, 0, bias=True)
    # one pass
    nb = be.lib.mn_conv2d_ws_bytes(C.byref(g), 2, 0)
    ws2 = be.empty(max(4, nb // 4 + 4))
    dw, db, dgam2, dbet2 = be.empty((Oc, Cin, k, k)), be.empty(Oc), be.empty(Oc), be.empty(Oc)
    if kind == "bn":
        be.call("mn_conv2d_bwd_first_bn_gram", C.byref(g), be.ptr(dDA), be.ptr(dY), be.ptr(dS), be.ptr(dG), be.ptr(dB), be.ptr(dW), be.ptr(dBi), be.ptr(gram),
                be.ptr(dX), be.ptr(dw), be.ptr(db), be.ptr(dgam2), be.ptr(dbet2), be.ptr(ws2), nb, be.stream)
    else:
        be.call("mn_conv2d_bwd_first_qa_gram", C.byref(g), be.ptr(dDA), be.ptr(dY), be.ptr(chan), bits, quant, be.ptr(dW), be.ptr(dBi), be.ptr(gram),
                be.ptr(dX), be.ptr(dw), be.ptr(db), be.ptr(dgam2), be.ptr(dbet2), be.ptr(ws2), nb, be.stream)
    dw_ref_h = be.to_host(dw_ref)
    assert np.abs(dw_ref_h).max() > 0
    assert close(be.to_host(dw), dw_ref_h, tol), np.abs(be.to_host(dw) - dw_ref_h).max() / np.abs(dw_ref_h).max()
    assert close(be.to_host(dgam2), be.to_host(dgam), tol) and close(be.to_host(dbet2), be.to_host(dbet), tol)
    assert np.abs(be.to_host(db)).max() <= 1e-4 * max(np.abs(be.to_host(dbet)).max(), 1e-30)


def check_first_conv_gram_conditioning(be, x_shape=(16, 3, 16, 16), Oc=16, k=3, seed=0):
    """The Gram-data statistics of the first block on what conditions them worst: un-normalised 0..255 images (mean^2 >> variance) that are smooth (neighbouring taps
    almost equal) under zero-sum difference filters (var_y << |w|^2 lambda_max(G)).  The kernel accumulates the Gram data of mean-shifted features and rebuilds the
    raw data in fp64 (k_c1_gram_unshift), so mean / invstd still agree with the statistics of y = conv(x, w) itself."""
    r = np.random.default_rng(seed)
    N, Cin, H, W = x_shape
    base = r.uniform(60, 200, size=(N, Cin, 1, 1))
    ramp = np.linspace(0, 1, W)[None, None, None, :] * r.uniform(-20, 20, size=(N, Cin, 1, 1)) + np.linspace(0, 1, H)[None, None, :, None] * r.uniform(-20, 20, size=(N, Cin, 1, 1))
    x = np.clip(base + ramp + r.standard_normal(x_shape) * 2.0, 0, 255).astype(F)
    w = (r.standard_normal((Oc, Cin, k, k)) * 0.2)
    w[: Oc // 2] -= w[: Oc // 2].mean(axis=(1, 2, 3), keepdims=True)          # half of the filters zero-sum over all taps (edge / difference filters)
    w = w.astype(F)
    b = (r.standard_normal(Oc) * 0.3).astype(F)
    g = be.geom(x_shape, (Oc, Cin, k, k), padding=k // 2)
    dX, dW, dBi = be.to_dev(x), be.to_dev(w), be.to_dev(b)
    import torch
    yh = torch.nn.functional.conv2d(torch.from_numpy(x.astype(np.float64)), torch.from_numpy(w.astype(np.float64)), torch.from_numpy(b.astype(np.float64)), 1, k // 2).numpy()
    mean, var = yh.mean(axis=(0, 2, 3)), yh.var(axis=(0, 2, 3))
    nbg = int(be.lib.mn_conv2d_first_xgram_ws_bytes(C.byref(g)))
    wsg, gram = be.empty(nbg // 4 + 4), be.empty(2 * 80 * 80)
    be.call("mn_conv2d_first_xgram", C.byref(g), be.ptr(dX), be.ptr(gram), be.ptr(wsg), nbg, be.stream)
    sv = be.empty((2, Oc))
    be.call("mn_conv2d_first_gram_bnstats", C.byref(g), be.ptr(dW), be.ptr(dBi), be.ptr(gram), 1e-5, 0.1, None, None, be.ptr(sv), be.stream)
    svh = be.to_host(sv)
    K = Cin * k * k
    G = be.to_host(gram).view(np.float64).reshape(80, 80)
    assert close(G[:K + 1, :K + 1], _im2col_gram(x, k), 2e-6)
    assert np.abs(svh[0] - mean).max() <= 2e-6 * max(np.abs(mean).max(), np.sqrt(var).max())
    err = np.abs(svh[1] * np.sqrt(var + 1e-5) - 1).max()
    assert err <= 2e-5, err          # (the un-shifted fp32 accumulation: 1e-3 ... 1e-1 on these inputs)


def check_first_conv_fused(be, x_shape=(3, 3, 8, 8), Oc=24, k=5, act=1, bits=2, bias=True, seed=0):
    """The fused first block (mn_conv2d_first_bnact_fwd: conv + BatchNorm + sign / ReLU + quantizer in one kernel, codes + pass bits instead of y; backward on
    (da, mask4)) against the unfused kernels on the same statistics: identical codes, identical masks, bit-identical gradients."""
    r = np.random.default_rng(seed)
    N, Cin, H, W = x_shape
    HW = H * W
    x = r.standard_normal(x_shape).astype(F)
    w = (r.standard_normal((Oc, Cin, k, k)) * 0.2).astype(F)
    b = (r.standard_normal(Oc) * 0.3).astype(F) if bias else None
    da = r.standard_normal((N, Oc, H, W)).astype(F)
    if act == 1:
        gamma, beta = (r.standard_normal(Oc) * 0.5 + 1).astype(F), (r.standard_normal(Oc) * 0.3).astype(F)
    else:
        gamma, beta = (r.standard_normal(Oc) * 2.0 + 3.0).astype(F), (r.standard_normal(Oc) * 2.0 + 3.0).astype(F)
    g = be.geom(x_shape, (Oc, Cin, k, k), padding=k // 2)
    dX, dW, dBi, dDA, dG, dB = be.to_dev(x), be.to_dev(w), (be.to_dev(b) if bias else None), be.to_dev(da), be.to_dev(gamma), be.to_dev(beta)
    dY = be.conv_fwd(g, be.actq(0), dX, dW, dBi, 0)
    nbg = int(be.lib.mn_conv2d_first_xgram_ws_bytes(C.byref(g)))
    wsg, gram, sv = be.empty(nbg // 4 + 4), be.empty(2 * 80 * 80), be.empty((2, Oc))
    be.call("mn_conv2d_first_xgram", C.byref(g), be.ptr(dX), be.ptr(gram), be.ptr(wsg), nbg, be.stream)
    be.call("mn_conv2d_first_gram_bnstats", C.byref(g), be.ptr(dW), be.ptr(dBi), be.ptr(gram), 1e-5, 0.1, None, None, be.ptr(sv), be.stream)
    # host: z exactly as the kernels evaluate it (fp32, no contraction)
    yh, svh = be.to_host(dY), be.to_host(sv)
    c4 = lambda v: v.astype(F)[None, :, None, None]
    z = ((yh - c4(svh[0])) * c4(svh[1])) * c4(gamma) + c4(beta)
    if act == 1:
        codes_ref = np.where(z < 0, -1, 1).astype(np.int8)
        lo_bits, hi_bits = (z > -1) & (z < 1), np.zeros_like(z, dtype=bool)
    else:
        chan = be.empty((9, Oc))
        be.call("mn_qa_chan_from_save", be.ptr(sv), be.ptr(dG), be.ptr(dB), Oc, be.ptr(chan), be.stream)
        cref = be.to_dev_u8(np.zeros((N, Oc, H, W), dtype=np.uint8))
        be.call("mn_qa_fwd", 1, be.ptr(dY), be.ptr(chan), N, Oc, H, W, bits, 0, be.ptr(cref), None, be.stream)
        codes_ref = be.to_host(cref)
        a = np.where(z > 0, z, 0).astype(F)
        t = a * F(0.1)
        lo_bits, hi_bits = z > 0, (z > 0) & (t >= 0) & (t <= 1)
    pack = lambda bts: (bts.reshape(N, Oc, HW // 4, 4) * np.array([1, 2, 4, 8])).sum(-1).astype(np.uint8)
    mask_ref = pack(lo_bits) | (pack(hi_bits) << 4)
    codes = be.to_dev_i8(np.zeros((N, Oc, H, W), dtype=np.int8)) if act == 1 else be.to_dev_u8(np.zeros((N, Oc, H, W), dtype=np.uint8))
    mask4 = be.to_dev_u8(np.zeros((N, Oc, HW // 4), dtype=np.uint8))
    be.call("mn_conv2d_first_bnact_fwd", C.byref(g), be.ptr(dX), be.ptr(dW), be.ptr(dBi), be.ptr(sv), be.ptr(dG), be.ptr(dB), act, bits, be.ptr(codes), be.ptr(mask4), be.stream)
    assert eq(be.to_host(codes), codes_ref)
    assert eq(be.to_host(mask4), mask_ref)
    if act == 2:          # the unfused forward pass that leaves the same nibbles
        c2_, m2_ = be.to_dev_u8(np.zeros((N, Oc, H, W), dtype=np.uint8)), be.to_dev_u8(np.zeros((N, Oc, HW // 4), dtype=np.uint8))
        be.call("mn_qa_fwd_f32_mask", be.ptr(dY), be.ptr(chan), N, Oc, H, W, bits, be.ptr(c2_), be.ptr(m2_), be.stream)
        assert eq(be.to_host(c2_), codes_ref) and eq(be.to_host(m2_), mask_ref)
    # backward: (da, mask4) vs (da, y)
    nb = be.lib.mn_conv2d_ws_bytes(C.byref(g), 2, 0)
    ws2 = be.empty(max(4, nb // 4 + 4))
    for quant in ((0,) if act == 1 else (1, 0)):
        ref = [be.empty((Oc, Cin, k, k)), be.empty(Oc), be.empty(Oc), be.empty(Oc)]
        out = [be.empty((Oc, Cin, k, k)), be.empty(Oc), be.empty(Oc), be.empty(Oc)]
        if act == 1:
            be.call("mn_conv2d_bwd_first_bn_gram", C.byref(g), be.ptr(dDA), be.ptr(dY), be.ptr(sv), be.ptr(dG), be.ptr(dB), be.ptr(dW), be.ptr(dBi), be.ptr(gram),
                    be.ptr(dX), be.ptr(ref[0]), be.ptr(ref[1]), be.ptr(ref[2]), be.ptr(ref[3]), be.ptr(ws2), nb, be.stream)
        else:
            be.call("mn_conv2d_bwd_first_qa_gram", C.byref(g), be.ptr(dDA), be.ptr(dY), be.ptr(chan), bits, quant, be.ptr(dW), be.ptr(dBi), be.ptr(gram),
                    be.ptr(dX), be.ptr(ref[0]), be.ptr(ref[1]), be.ptr(ref[2]), be.ptr(ref[3]), be.ptr(ws2), nb, be.stream)
        be.call("mn_conv2d_bwd_first_mask_gram", C.byref(g), be.ptr(dDA), be.ptr(mask4), quant, be.ptr(sv), be.ptr(dG), be.ptr(dW), be.ptr(dBi), be.ptr(gram),
                be.ptr(dX), be.ptr(out[0]), be.ptr(out[1]), be.ptr(out[2]), be.ptr(out[3]), be.ptr(ws2), nb, be.stream)
        assert np.abs(be.to_host(ref[0])).max() > 0
        for a_, b_ in zip(out, ref):
            assert eq(be.to_host(a_), be.to_host(b_))


def check_ternary_multi(be, seed=0):
    """mn_ternary_w_fwd_multi / mn_ternary_w_bwd_multi (one launch over several weight tensors) bit-identical to the per-tensor entry points."""
    r = np.random.default_rng(seed)
    shapes = [(24, 3 * 25), (40, 16), (7, 144), (64, 9)]
    ws = [(r.standard_normal(sh) * 0.3).astype(F) for sh in shapes]
    gs = [r.standard_normal(sh).astype(F) for sh in shapes]
    n = len(ws)
    dW, dG = [be.to_dev(w) for w in ws], [be.to_dev(g) for g in gs]
    q1, s1, d1 = [be.empty(sh) for sh in shapes], [be.empty((sh[0], 4)) for sh in shapes], [be.empty(sh) for sh in shapes]
    for i, sh in enumerate(shapes):
        be.call("mn_ternary_w_fwd", be.ptr(dW[i]), be.ptr(q1[i]), be.ptr(s1[i]), sh[0], sh[1], be.stream)
        be.call("mn_ternary_w_bwd", be.ptr(dG[i]), be.ptr(dW[i]), be.ptr(s1[i]), be.ptr(d1[i]), sh[0], sh[1], be.stream)
    q2, s2, d2 = [be.empty(sh) for sh in shapes], [be.empty((sh[0], 4)) for sh in shapes], [be.empty(sh) for sh in shapes]
    PA, LA = C.c_void_p * n, C.c_int64 * n
    arr = lambda ts: PA(*[be.ptr(t).value for t in ts])
    Os, Ks = LA(*[sh[0] for sh in shapes]), LA(*[sh[1] for sh in shapes])
    be.call("mn_ternary_w_fwd_multi", arr(dW), arr(q2), arr(s2), Os, Ks, n, be.stream)
    be.call("mn_ternary_w_bwd_multi", arr(dG), arr(dW), arr(s2), arr(d2), Os, Ks, n, be.stream)
    for i in range(n):
        assert eq(be.to_host(q1[i]), be.to_host(q2[i])) and eq(be.to_host(s1[i]), be.to_host(s2[i])) and eq(be.to_host(d1[i]), be.to_host(d2[i])), i


def check_binary_multi(be, seed=0):
    """mn_binary_w_fwd_multi / mn_binary_w_bwd_multi (one launch over several conv weights, each mean-centred and clamped IN PLACE) bit-identical to the per-tensor
    entry points: the mutated weights, the quantised weights, alpha and the gradients."""
    r = np.random.default_rng(seed)
    shapes = [(24, 16, 1), (40, 3, 25), (7, 128, 1), (64, 16, 9)]          # [O][C][R]
    ws = [(r.standard_normal(sh) * 0.8).astype(F) for sh in shapes]
    gs = [r.standard_normal(sh).astype(F) for sh in shapes]
    n = len(ws)
    w1, w2, dG = [be.to_dev(w) for w in ws], [be.to_dev(w) for w in ws], [be.to_dev(g) for g in gs]
    q1, a1, d1 = [be.empty(sh) for sh in shapes], [be.empty(sh[0]) for sh in shapes], [be.empty(sh) for sh in shapes]
    for i, sh in enumerate(shapes):
        be.call("mn_binary_w_fwd", be.ptr(w1[i]), be.ptr(q1[i]), be.ptr(a1[i]), sh[0], sh[1], sh[2], be.stream)
        be.call("mn_binary_w_bwd", be.ptr(dG[i]), be.ptr(w1[i]), be.ptr(a1[i]), be.ptr(d1[i]), sh[0], sh[1] * sh[2], be.stream)
    q2, a2, d2 = [be.empty(sh) for sh in shapes], [be.empty(sh[0]) for sh in shapes], [be.empty(sh) for sh in shapes]
    PA, LA = C.c_void_p * n, C.c_int64 * n
    arr = lambda ts: PA(*[be.ptr(t).value for t in ts])
    Os, Cs, Rs = LA(*[sh[0] for sh in shapes]), LA(*[sh[1] for sh in shapes]), LA(*[sh[2] for sh in shapes])
    be.call("mn_binary_w_fwd_multi", arr(w2), arr(q2), arr(a2), Os, Cs, Rs, n, be.stream)
    be.call("mn_binary_w_bwd_multi", arr(dG), arr(w2), arr(a2), arr(d2), Os, Cs, Rs, n, be.stream)
    for i in range(n):
        assert not eq(be.to_host(w2[i]), ws[i])          # mutated in place
        assert eq(be.to_host(w1[i]), be.to_host(w2[i])) and eq(be.to_host(q1[i]), be.to_host(q2[i])) and eq(be.to_host(a1[i]), be.to_host(a2[i])) and \
            eq(be.to_host(d1[i]), be.to_host(d2[i])), i


def check_pool_f32(be, shape=(3, 5, 8, 16), seed=0):
    """mn_maxpool2x2_f32_fwd/bwd vs torch CPU max_pool2d: values, the gradient routing (ties -> first maximum, NaN wins) bit-exact."""
    import torch
    r = np.random.default_rng(seed)
    x = np.round(r.standard_normal(shape) * 2).astype(F) * 0.5          # coarse grid: many ties inside windows
    x[0, 0, 0, 1] = np.nan; x[0, 0, 1, 0] = np.nan                      # two NaN in one window: the later one is the argmax
    if shape[0] > 1 and shape[2] > 2:
        x[1, 1, 2, 4] = np.nan
    N, Cc, H, W = shape
    t = torch.from_numpy(x.copy()).requires_grad_(True)
    y_ref = torch.nn.functional.max_pool2d(t, 2, 2)
    gy = r.standard_normal(tuple(y_ref.shape)).astype(F)
    y_ref.backward(torch.from_numpy(gy))
    dX, dG = be.to_dev(x), be.to_dev(gy)
    y, dx = be.empty(tuple(y_ref.shape)), be.empty(shape)
    idx = be.empty_i8(tuple(y_ref.shape))
    assert be.lib.mn_maxpool2x2_f32_supported(H, W) == 1
    be.call("mn_maxpool2x2_f32_fwd", be.ptr(dX), N * Cc, H, W, be.ptr(y), be.ptr(idx), be.stream)
    be.call("mn_maxpool2x2_f32_bwd", be.ptr(dG), be.ptr(idx), N * Cc, H, W, be.ptr(dx), be.stream)
    assert np.array_equal(be.to_host(y), y_ref.detach().numpy(), equal_nan=True)
    assert np.array_equal(be.to_host(dx), t.grad.numpy())


def check_qconv_bnq(be, x_shape, w_shape, groups=1, bias=True, in_shuffle=0, training=True, seed=0, pooled=False, padding=0, a_bits=2, w_bits=2,
                    out_bits=None, quant=1, **_):
    """The k-bit (DoReFa) fused block -- mn_qconv_bnq_fwd_stash + mn_qa_fwd + mn_qa_bwd_sums/_apply + mn_conv2d_bwd_weight/_bwd_data on activation codes --
    vs a numpy evaluation of the unfused chain (wqaq/dorefa/quantize.py:36-46 act quantizer, 107-122 conv; BatchNorm2d; ReLU; 2x2 max-pool):
      stash == the exact integer conv result; batch statistics / activation to fp32 round-off; codes equal away from rounding ties;
      masks and pool routing recomputed from the same values; all gradients <= 1e-5 rel of an fp64 evaluation."""
    import torch
    r = np.random.default_rng(seed)
    out_bits = out_bits or a_bits
    N, Cin, H, W = x_shape
    Oc = w_shape[0]
    HW = H * W
    na, nw = 2 ** a_bits - 1, 2 ** w_bits - 1
    s_a, s_o = F(1.0 / na), F(1.0 / (2 ** out_bits - 1))
    codes_in = r.integers(0, na + 1, size=x_shape).astype(np.uint8)
    kw_ = r.integers(0, nw + 1, size=w_shape)
    wcode = (2 * kw_ - nw).astype(np.float64)
    w = (F(2.0) * (kw_.astype(F) * F(1.0 / nw)) - F(1.0)).astype(F)          # the fake-quantised fp32 weights the reference holds (2 k s - 1)
    b = (r.standard_normal(Oc) * 0.2).astype(F) if bias else None
    gamma, beta = (r.standard_normal(Oc) * 0.5 + 1).astype(F), (r.standard_normal(Oc) * 0.3).astype(F)
    rm, rv = (r.standard_normal(Oc) * 0.1).astype(F), (np.abs(r.standard_normal(Oc)) * 2 + 0.5).astype(F)
    eps, mom = 1e-5, 0.1
    x_log = codes_in.astype(np.float64)
    if in_shuffle > 1:
        x_log = np.ascontiguousarray(x_log.reshape(N, in_shuffle, Cin // in_shuffle, H, W).transpose(0, 2, 1, 3, 4).reshape(x_shape))
    tconv = lambda xx, ww: torch.nn.functional.conv2d(torch.from_numpy(xx), torch.from_numpy(ww), None, 1, padding, 1, groups).numpy()
    acc = tconv(x_log, wcode)                                                # exact integers (float64)
    alpha = F(F(1.0 / nw) * s_a)
    y = (acc.astype(F) * alpha + (b.reshape(1, -1, 1, 1) if bias else F(0))).astype(F)      # the fp32 chain of the kernels' epilogue
    y64 = y.astype(np.float64)
    n = N * HW
    if training:
        # exact statistics of y = alpha * acc + bias (what the integer sums give), not of its fp32 rounding
        ye = acc * float(alpha) + (b.astype(np.float64).reshape(1, -1, 1, 1) if bias else 0.0)
        mean = ye.mean(axis=(0, 2, 3)); var_b = ye.var(axis=(0, 2, 3)); var_u = var_b * n / (n - 1)
    else:
        mean, var_b = rm.astype(np.float64), rv.astype(np.float64)
    mean32 = mean.astype(F)
    inv32 = (F(1.0) / np.sqrt(var_b.astype(F) + F(eps))).astype(F)
    zh = ((y - mean32.reshape(1, -1, 1, 1)) * inv32.reshape(1, -1, 1, 1)).astype(F)
    z = (zh * gamma.reshape(1, -1, 1, 1) + beta.reshape(1, -1, 1, 1)).astype(F)
    a = np.maximum(z, F(0)).astype(F)

    g = be.geom(x_shape, w_shape, padding=padding, groups=groups)
    g.in_shuffle = in_shuffle
    wq = be.wq(mode=2, bits=w_bits)
    assert be.lib.mn_qconv_bnq_supported(C.byref(g), C.byref(wq), a_bits) == 1, "bnq not supported for this case"
    sbits = int(be.lib.mn_qconv_bnq_stash_bits(C.byref(g), C.byref(wq), a_bits))
    Kc = (Cin // groups) * w_shape[2] * w_shape[3]
    assert sbits == (32 if (a_bits == 8 or Kc * na * nw > 32767) else 16), sbits      # wide stash: 8-bit codes, or an accumulator beyond int16
    assert np.abs(acc).max() <= (32767 if sbits == 16 else 2 ** 24)
    kind = 2 if sbits == 32 else 0                                           # mn_qa_*'s in_kind
    nb = max(int(be.lib.mn_qconv_bnq_ws_bytes(C.byref(g))), 4 * int(be.lib.mn_qa_ws_floats(Oc)))
    ws = be.empty(nb // 4 + 8)
    dX = be.to_dev_i8(codes_in.view(np.int8))
    dW, dB = be.to_dev(w), (be.to_dev(b) if bias else None)
    dG, dBe, dRM, dRV = be.to_dev(gamma), be.to_dev(beta), be.to_dev(rm), be.to_dev(rv)
    save, chan = be.empty((2, Oc)), be.empty((9, Oc))
    stash = be.empty_i8((N, Oc, H, (sbits // 8) * W))                        # int16 / int32 [N][Oc][H][W]
    nbt = be.to_dev_i64([7])
    be.call("mn_qconv_bnq_fwd_stash", C.byref(g), C.byref(wq), be.ptr(dX), a_bits, be.ptr(dW), be.ptr(dB), be.ptr(dG), be.ptr(dBe), eps, mom, int(training),
            be.ptr(dRM), be.ptr(dRV), be.ptr(nbt), be.ptr(save), be.ptr(stash), be.ptr(chan), be.ptr(ws), nb, be.stream)
    assert int(be.to_host(nbt)[0]) == (8 if training else 7)
    st = be.to_host(stash).view(np.int16 if sbits == 16 else np.int32).reshape(N, Oc, H, W)
    assert np.array_equal(st.astype(np.float64), acc), ("stash != exact integer conv result", float(np.abs(st - acc).max()))
    sv = be.to_host(save)
    assert np.max(np.abs(sv[0] - mean32)) <= 2e-6 * max(np.max(np.abs(mean32)), 1e-3) + 1e-7, "mean"
    assert np.max(np.abs(sv[1] - inv32) / inv32) <= 5e-6, "invstd"
    if training:
        assert close(be.to_host(dRM), ((1 - mom) * rm + mom * mean).astype(F), 1e-6) and close(be.to_host(dRV), ((1 - mom) * rv + mom * var_u).astype(F), 1e-5)
    # ---- streaming forward: the fp32 activation and the codes of the NEXT quantizer (through the 2x2 max-pool if pooled)
    Ho, Wo = (H // 2, W // 2) if pooled else (H, W)
    codes_out = be.empty_i8((N, Oc, Ho, Wo))
    act = be.empty((N, Oc, Ho, Wo))
    be.call("mn_qa_fwd", kind, be.ptr(stash), be.ptr(chan), N, Oc, H, W, out_bits, int(pooled), be.ptr(codes_out), be.ptr(act), be.stream)
    # the kernels use THEIR (mean, invstd): recompute the reference chain with them so that every downstream comparison is exact up to ties
    m_k, i_k = sv[0].astype(F), sv[1].astype(F)
    zh = ((y - m_k.reshape(1, -1, 1, 1)) * i_k.reshape(1, -1, 1, 1)).astype(F)
    z = (zh * gamma.reshape(1, -1, 1, 1) + beta.reshape(1, -1, 1, 1)).astype(F)
    a = np.maximum(z, F(0)).astype(F)
    if pooled:
        ta = torch.from_numpy(a.copy()).requires_grad_(True)
        ap_t = torch.nn.functional.max_pool2d(ta, 2, 2)
        a_out = ap_t.detach().numpy()
    else:
        a_out = a
    got_act = be.to_host(act)
    assert np.array_equal(got_act, a_out), ("activation", float(np.abs(got_act - a_out).max()))
    c_ref = np.sign(a_out * F(0.1)) * np.floor(np.abs(np.clip(a_out * F(0.1), 0, 1).astype(F) / s_o) + F(0.5))
    assert np.array_equal(be.to_host(codes_out).view(np.uint8).astype(np.float64), c_ref.astype(np.float64)), "codes of the next quantizer"
    # ---- backward of BatchNorm + ReLU (+ pool) (+ the quantizer's clip-STE when quant)
    dq = r.standard_normal((N, Oc, Ho, Wo)).astype(F)
    if quant:
        t_ = a_out * F(0.1)
        d_ = ((dq * s_o) / s_o).astype(F)
        d_ = np.where((t_ >= 0) & (t_ <= 1), d_, F(0)) * F(0.1)
    else:
        d_ = dq
    if pooled:
        ap_t.backward(torch.from_numpy(d_.astype(F)))
        da = ta.grad.numpy()
    else:
        da = d_
    dz = np.where(z > 0, da, F(0)).astype(np.float64)
    dbeta_ref, dgamma_ref = dz.sum(axis=(0, 2, 3)), (dz * zh.astype(np.float64)).sum(axis=(0, 2, 3))
    gi = (gamma.astype(np.float64) * i_k.astype(np.float64)).reshape(1, -1, 1, 1)
    dy_ref = gi * (dz - dbeta_ref.reshape(1, -1, 1, 1) / n - zh.astype(np.float64) * dgamma_ref.reshape(1, -1, 1, 1) / n) if training else gi * dz
    dDQ = be.to_dev(dq)
    dy, dgam, dbet, sums = be.empty((N, Oc, H, W)), be.empty(Oc), be.empty(Oc), be.empty((2, Oc))
    be.call("mn_qa_bwd_sums", kind, be.ptr(stash), be.ptr(chan), be.ptr(dDQ), N, Oc, H, W, out_bits, int(pooled), int(quant), be.ptr(dgam), be.ptr(dbet), be.ptr(sums),
            be.ptr(ws), be.stream)
    be.call("mn_qa_bwd_apply", kind, be.ptr(stash), be.ptr(chan), be.ptr(sums), be.ptr(dDQ), N, Oc, H, W, out_bits, int(pooled), int(quant), int(training), be.ptr(dy),
            be.stream)
    sc = max(np.max(np.abs(dz)) * np.sqrt(n), 1e-30)
    assert np.max(np.abs(be.to_host(dbet) - dbeta_ref)) <= 2e-6 * sc and np.max(np.abs(be.to_host(dgam) - dgamma_ref)) <= 2e-6 * sc * max(1.0, np.abs(zh).max()), "dgamma / dbeta"
    assert close(be.to_host(dy), dy_ref, 1e-5), ("dy", float(np.max(np.abs(be.to_host(dy) - dy_ref)) / np.max(np.abs(dy_ref))))
    # the two calls as two launches (mn_qa_bwd: the apply pass finishes the sums itself): every output to the bit
    dy2, dgam2, dbet2, sums2 = be.empty((N, Oc, H, W)), be.empty(Oc), be.empty(Oc), be.empty((2, Oc))
    be.call("mn_qa_bwd", kind, be.ptr(stash), be.ptr(chan), be.ptr(dDQ), N, Oc, H, W, out_bits, int(pooled), int(quant), int(training), be.ptr(dgam2), be.ptr(dbet2),
            be.ptr(sums2), be.ptr(dy2), be.ptr(ws), be.stream)
    assert eq(be.to_host(dy2), be.to_host(dy)) and eq(be.to_host(dgam2), be.to_host(dgam)) and eq(be.to_host(dbet2), be.to_host(dbet)) and eq(be.to_host(sums2), be.to_host(sums))
    # ---- backward of the conv on codes: dx (no STE here) and dw = s_a * sum dy * j
    gy = dy_ref.astype(F)
    dGY = be.to_dev(gy)
    aq = be.actq(4, a_bits)
    tx = torch.from_numpy(x_log * float(s_a)).requires_grad_(True)
    tw = torch.from_numpy(w.astype(np.float64)).requires_grad_(True)
    torch.nn.functional.conv2d(tx, tw, None, 1, padding, 1, groups).backward(torch.from_numpy(gy.astype(np.float64)))
    dx_ref, dw_ref = tx.grad.numpy(), tw.grad.numpy()
    if in_shuffle > 1:      # dx is delivered in the PHYSICAL channel order
        dx_ref = np.ascontiguousarray(dx_ref.reshape(N, Cin // in_shuffle, in_shuffle, H, W).transpose(0, 2, 1, 3, 4).reshape(x_shape))
    dx = be.empty(x_shape)
    nb1 = int(be.lib.mn_conv2d_ws_bytes(C.byref(g), 1, 0))
    ws1 = be.empty(nb1 // 4 + 8)
    be.call("mn_conv2d_bwd_data", C.byref(g), C.byref(aq), C.byref(wq), be.ptr(dGY), be.ptr(dW), None, be.ptr(dx), be.ptr(ws1), nb1, 0, be.stream)
    assert close(be.to_host(dx), dx_ref, 1e-5), ("dx", float(np.max(np.abs(be.to_host(dx) - dx_ref)) / np.max(np.abs(dx_ref))))
    dw, db = be.empty(w_shape), (be.empty(Oc) if bias else None)
    nb2 = int(be.lib.mn_conv2d_ws_bytes(C.byref(g), 2, 0))
    ws2 = be.empty(nb2 // 4 + 8)
    be.call("mn_conv2d_bwd_weight", C.byref(g), C.byref(aq), be.ptr(dGY), be.ptr(dX), be.ptr(dw), be.ptr(db), be.ptr(ws2), nb2, 0, be.stream)
    assert close(be.to_host(dw), dw_ref, 1e-5), ("dw", float(np.max(np.abs(be.to_host(dw) - dw_ref)) / np.max(np.abs(dw_ref))))
    if bias:
        db_ref = gy.astype(np.float64).sum(axis=(0, 2, 3))
        assert np.max(np.abs(be.to_host(db) - db_ref)) <= 1e-5 * max(np.abs(gy).sum(axis=(0, 2, 3)).max(), 1e-30), "dbias"
    # ---- both gradients in one launch (k_pwb): from the plain gradient, and -- un-pooled blocks -- with the block's backward formed from (dq, stash) inside
    if be.lib.mn_conv2d_bwd_bnh_supported(C.byref(g), C.byref(wq), 0):
        nb3 = int(be.lib.mn_conv2d_bwd_bnh_ws_bytes(C.byref(g)))
        ws3, dx3, dw3, db3 = be.empty(nb3 // 4 + 8), be.empty(x_shape), be.empty(w_shape), (be.empty(Oc) if bias else None)
        be.call("mn_conv2d_bwd_codes", C.byref(g), C.byref(wq), be.ptr(dGY), be.ptr(dW), be.ptr(dX), a_bits, be.ptr(dx3), be.ptr(dw3), be.ptr(db3), be.ptr(ws3), nb3, be.stream)
        assert be.lib.mn_last_kernel().decode() == "k_pwb<0, 1, 0>"
        assert close(be.to_host(dx3), dx_ref, 1e-5), ("k_pwb dx", float(np.max(np.abs(be.to_host(dx3) - dx_ref)) / np.max(np.abs(dx_ref))))
        assert close(be.to_host(dw3), dw_ref, 1e-5), ("k_pwb dw", float(np.max(np.abs(be.to_host(dw3) - dw_ref)) / np.max(np.abs(dw_ref))))
        if bias:
            assert np.max(np.abs(be.to_host(db3) - db_ref)) <= 1e-5 * max(np.abs(gy).sum(axis=(0, 2, 3)).max(), 1e-30), "k_pwb dbias"
        if not pooled:
            # reference: the two-kernel path on the dy the apply pass wrote above
            dx_t, dw_t, db_t = be.empty(x_shape), be.empty(w_shape), (be.empty(Oc) if bias else None)
            be.call("mn_conv2d_bwd_data", C.byref(g), C.byref(aq), C.byref(wq), be.ptr(dy), be.ptr(dW), None, be.ptr(dx_t), be.ptr(ws1), nb1, 0, be.stream)
            be.call("mn_conv2d_bwd_weight", C.byref(g), C.byref(aq), be.ptr(dy), be.ptr(dX), be.ptr(dw_t), be.ptr(db_t), be.ptr(ws2), nb2, 0, be.stream)
            dx4, dw4, db4 = be.empty(x_shape), be.empty(w_shape), (be.empty(Oc) if bias else None)
            be.call("mn_conv2d_bwd_qa", C.byref(g), C.byref(wq), be.ptr(dDQ), be.ptr(stash), sbits, be.ptr(chan), be.ptr(sums), out_bits, int(quant), int(training),
                    be.ptr(dW), be.ptr(dX), a_bits, be.ptr(dx4), be.ptr(dw4), be.ptr(db4), be.ptr(ws3), nb3, be.stream)
            assert be.lib.mn_last_kernel().decode() == "k_pwb<3, 1, %d>" % (1 if sbits == 32 else 0)
            assert close(be.to_host(dx4), be.to_host(dx_t), 5e-6), ("k_pwb<3> dx", float(np.max(np.abs(be.to_host(dx4) - be.to_host(dx_t))) / np.max(np.abs(be.to_host(dx_t)))))
            assert close(be.to_host(dw4), be.to_host(dw_t), 5e-6), ("k_pwb<3> dw", float(np.max(np.abs(be.to_host(dw4) - be.to_host(dw_t))) / np.max(np.abs(be.to_host(dw_t)))))
            if bias:
                sc_db = max(np.max(np.abs(be.to_host(dy))) * 1e-4, 1e-30)      # d bias in front of a BatchNorm is a sum that cancels to ~0
                assert np.max(np.abs(be.to_host(db4) - be.to_host(db_t))) <= sc_db * N * H * W, "k_pwb<3> dbias"
        if not pooled and sbits == 16:
            _check_pwb_up2(be, g, wq, r, dGY, dDQ, stash, chan, sums, out_bits, quant, training, dW, dX, a_bits, dx3, dw3, dx4, dw4, nb3, x_shape, w_shape, seed)
        check_qconv_bnq.pwb_checked = getattr(check_qconv_bnq, "pwb_checked", 0) + 1


def _check_pwb_up2(be, g, wq, r, dGY, dDQ, stash, chan, sums, out_bits, quant, training, dW, dX, a_bits, dx3, dw3, dx4, dw4, nb3, x_shape, w_shape, seed):
    """mn_conv2d_bwd_codes_up / mn_conv2d_bwd_qa_up (k_pwb<.., UP 2>): dx / dw bit-identical to the plain launches; the by-product -- sum dz, sum dz zhat of a k-bit
    block in front, finished by mn_qa_bwd_sums_final -- against mn_qa_bwd_sums on (that dx, the upstream 16-bit stash) and an fp64 evaluation of the same masks."""
    N, Cin, H, W = x_shape
    splits = int(be.lib.mn_conv2d_bwd_bnh_up_splits(C.byref(g), C.byref(wq), 0, 1))
    assert splits > 0
    # an upstream block: integers in the 16-bit stash, constants that put the ReLU kink and the clamp edge inside the range for most channels
    up_st = r.integers(-300, 301, size=x_shape).astype(np.int16)
    alpha_u = np.full(Cin, 0.01, F)
    bias_u = (r.standard_normal(Cin) * 0.1).astype(F)
    mean_u = (r.standard_normal(Cin) * 0.2).astype(F)
    inv_u = (np.abs(r.standard_normal(Cin)) * 0.5 + 0.5).astype(F)
    ga_u = (r.standard_normal(Cin) * 2.0 + 4.0).astype(F)
    be_u = (r.standard_normal(Cin) * 2.0 + 3.0).astype(F)
    ga_u[1] = -ga_u[1] - 0.5            # a decreasing channel
    ga_u[2] = 0.0                       # gamma == 0: element-wise masks
    be_u[3] = np.nan                    # a poisoned channel: element-wise masks, NaN comparisons
    ga_u[4], be_u[4] = 1e-3, -5.0       # nothing passes the ReLU
    up_chan = np.zeros((9, Cin), F)
    up_chan[0], up_chan[1], up_chan[2], up_chan[3], up_chan[4], up_chan[5] = alpha_u, bias_u, mean_u, inv_u, ga_u, be_u
    up_chan[8] = ga_u * inv_u
    d_us, d_uc = be.to_dev_i8(up_st.view(np.int8).reshape(N, Cin, H, 2 * W)), be.to_dev(up_chan)
    for up_quant in (1, 0):
        for form in ("codes", "qa"):
            ws, dx5, dw5 = be.empty(nb3 // 4 + 8), be.empty(x_shape), be.empty(w_shape)
            part = be.empty(Cin * splits * 4 + 2)
            if form == "codes":
                be.call("mn_conv2d_bwd_codes_up", C.byref(g), C.byref(wq), be.ptr(dGY), be.ptr(dW), be.ptr(dX), a_bits, be.ptr(dx5), be.ptr(dw5), None, be.ptr(ws), nb3,
                        be.ptr(d_us), be.ptr(d_uc), up_quant, be.ptr(part), be.stream)
                assert be.lib.mn_last_kernel().decode() == "k_pwb<0, 1, 0, 2>"
                assert np.array_equal(be.to_host(dx5), be.to_host(dx3)) and np.array_equal(be.to_host(dw5), be.to_host(dw3))
            else:
                be.call("mn_conv2d_bwd_qa_up", C.byref(g), C.byref(wq), be.ptr(dDQ), be.ptr(stash), be.ptr(chan), be.ptr(sums), out_bits, int(quant), int(training),
                        be.ptr(dW), be.ptr(dX), a_bits, be.ptr(dx5), be.ptr(dw5), None, be.ptr(ws), nb3, be.ptr(d_us), be.ptr(d_uc), up_quant, be.ptr(part), be.stream)
                assert be.lib.mn_last_kernel().decode() == "k_pwb<3, 1, 0, 2>"
                assert np.array_equal(be.to_host(dx5), be.to_host(dx4)) and np.array_equal(be.to_host(dw5), be.to_host(dw4))
            s_up, dg_up, db_up = be.empty((2, Cin)), be.empty(Cin), be.empty(Cin)
            be.call("mn_qa_bwd_sums_final", be.ptr(part), splits, Cin, be.ptr(dg_up), be.ptr(db_up), be.ptr(s_up), be.stream)
            s_ref, dg_ref, db_ref = be.empty((2, Cin)), be.empty(Cin), be.empty(Cin)
            ws2 = be.empty(int(be.lib.mn_qa_ws_floats(Cin)) + 2)
            be.call("mn_qa_bwd_sums", 0, be.ptr(d_us), be.ptr(d_uc), be.ptr(dx5), N, Cin, H, W, a_bits, 0, up_quant, be.ptr(dg_ref), be.ptr(db_ref), be.ptr(s_ref), be.ptr(ws2),
                    be.stream)
            got, ref = be.to_host(s_up).astype(np.float64), be.to_host(s_ref).astype(np.float64)
            # fp64 evaluation from the same dx with the kernels' fp32 chain for the masks
            dxh = be.to_host(dx5)
            sa = F(1.0 / (2 ** a_bits - 1))
            with np.errstate(invalid="ignore"):
                yv = (up_st.astype(F) * alpha_u.reshape(1, -1, 1, 1) + bias_u.reshape(1, -1, 1, 1)).astype(F)
                zh = ((yv - mean_u.reshape(1, -1, 1, 1)) * inv_u.reshape(1, -1, 1, 1)).astype(F)
                z = (zh * ga_u.reshape(1, -1, 1, 1) + be_u.reshape(1, -1, 1, 1)).astype(F)
                a_ = np.where(z > 0, z, F(0)).astype(F)          # (NaN > 0 is false)
                if up_quant:
                    t_ = (a_ * F(0.1)).astype(F)
                    d_ = ((dxh * sa) / sa).astype(F)
                    d_ = (np.where((t_ >= 0) & (t_ <= 1), d_, F(0)) * F(0.1)).astype(F)
                else:
                    d_ = dxh
                dz = np.where(z > 0, d_, F(0)).astype(np.float64)
            zh64 = np.nan_to_num(zh.astype(np.float64))
            e1, e2 = dz.sum(axis=(0, 2, 3)), (dz * zh64).sum(axis=(0, 2, 3))
            m1, m2 = np.abs(dz).sum(axis=(0, 2, 3)) + 1e-30, np.abs(dz * zh64).sum(axis=(0, 2, 3)) + 1e-30
            e_got = (np.max(np.abs(got[0] - e1) / m1), np.max(np.abs(got[1] - e2) / m2))
            assert e_got[0] <= 2e-6 and e_got[1] <= 2e-6, ("k_pwb<UP 2> upstream sums", form, up_quant, e_got)
            assert np.max(np.abs(ref[0] - e1) / m1) <= 2e-6 and np.max(np.abs(ref[1] - e2) / m2) <= 2e-6
            assert got[0, 4] == 0 and got[1, 4] == 0 and got[0, 3] == 0
            assert np.array_equal(be.to_host(dg_up), be.to_host(s_up)[1]) and np.array_equal(be.to_host(db_up), be.to_host(s_up)[0])
    _check_pwb_up2.count = getattr(_check_pwb_up2, "count", 0) + 1


# k-bit blocks on the geometries k_pwb covers (groups of 128 -> 128 channels): 16-bit and 32-bit stash, pooled (plain-gradient form only), eval mode, no quantizer behind
PWB_BNQ_CASES = [
    dict(x_shape=(3, 256, 4, 8), w_shape=(256, 128, 1, 1), groups=2, in_shuffle=2),
    dict(x_shape=(2, 128, 8, 8), w_shape=(128, 128, 1, 1), bias=False, training=False),
    dict(x_shape=(2, 256, 8, 8), w_shape=(256, 128, 1, 1), groups=2, pooled=True),
    dict(x_shape=(3, 256, 4, 8), w_shape=(256, 128, 1, 1), groups=2, in_shuffle=2, a_bits=8, w_bits=8),
    dict(x_shape=(5, 128, 4, 8), w_shape=(128, 128, 1, 1), quant=0, a_bits=4, w_bits=4, out_bits=3),
]


def check_pwb_bnq(be):
    before, before_up = getattr(check_qconv_bnq, "pwb_checked", 0), getattr(_check_pwb_up2, "count", 0)
    for i, case in enumerate(PWB_BNQ_CASES):
        check_qconv_bnq(be, seed=440 + i, **case)
    assert getattr(check_qconv_bnq, "pwb_checked", 0) - before == len(PWB_BNQ_CASES), "k_pwb did not take these geometries"
    assert getattr(_check_pwb_up2, "count", 0) - before_up == 3, "the upstream-sums variants (k_pwb<.., UP 2>) ran on the three un-pooled 16-bit cases"


BNQ_CASES = [
    dict(x_shape=(2, 96, 8, 8), w_shape=(96, 48, 1, 1), groups=2),
    dict(x_shape=(3, 80, 8, 16), w_shape=(96, 40, 1, 1), groups=2, in_shuffle=2, pooled=True),
    dict(x_shape=(2, 128, 4, 8), w_shape=(48, 128, 1, 1), bias=False, training=False),
    dict(x_shape=(2, 32, 8, 8), w_shape=(64, 16, 3, 3), groups=2, padding=1),
    dict(x_shape=(2, 32, 16, 16), w_shape=(48, 8, 3, 3), groups=4, padding=1, in_shuffle=2, pooled=True, a_bits=3, w_bits=3, out_bits=2),
    dict(x_shape=(4, 48, 8, 8), w_shape=(48, 48, 1, 1), quant=0, a_bits=4, w_bits=4),
    # the WIDE variants (8-bit codes / accumulators beyond int16: 32-bit stash, XENC 2 of the grouped kernels) -- W8A8 is the reference's CPU configuration
    dict(x_shape=(2, 96, 8, 8), w_shape=(96, 48, 1, 1), groups=2, a_bits=8, w_bits=8),
    dict(x_shape=(3, 80, 8, 16), w_shape=(96, 40, 1, 1), groups=2, in_shuffle=2, pooled=True, a_bits=8, w_bits=8),
    dict(x_shape=(2, 128, 4, 8), w_shape=(48, 128, 1, 1), bias=False, training=False, a_bits=8, w_bits=8),
    dict(x_shape=(2, 32, 8, 8), w_shape=(64, 16, 3, 3), groups=2, padding=1, a_bits=8, w_bits=8),
    dict(x_shape=(2, 32, 16, 16), w_shape=(48, 8, 3, 3), groups=4, padding=1, in_shuffle=2, pooled=True, a_bits=8, w_bits=8, out_bits=8),
    dict(x_shape=(4, 48, 8, 8), w_shape=(48, 48, 1, 1), quant=0, a_bits=6, w_bits=5),
    dict(x_shape=(2, 32, 8, 8), w_shape=(64, 16, 3, 3), groups=2, padding=1, a_bits=5, w_bits=6, out_bits=5),
]


# pointwise backward-weight, wave-specialised kernel (tiles of 128: Mg or Cg above 64) on tiles that are NOT full and on short / odd step ranges
WGRAD_SPEC_CASES = [
    dict(x_shape=(4, 160, 8, 8), w_shape=(192, 80, 1, 1), groups=2),                  # Mg = 96, Cg = 80: clamped rows; 8 steps over 4 blocks per tile
    dict(x_shape=(3, 96, 4, 8), w_shape=(80, 96, 1, 1)),                              # 3 steps in ONE block: fewer steps than the prefetch depth, odd count
    dict(x_shape=(5, 144, 4, 8), w_shape=(144, 72, 1, 1), groups=2, in_shuffle=2),    # 5 steps, shuffled input channels
]


# geometries k_pwb covers (groups of 128 -> 128 channels, H*W a multiple of 32): fewer steps than the prefetch depth, an odd count, more blocks than steps,
# shuffled channels, no bias, four groups
PWB_CASES = [
    dict(x_shape=(3, 256, 4, 8), w_shape=(256, 128, 1, 1), groups=2, in_shuffle=2),          # 3 steps per group
    dict(x_shape=(1, 128, 8, 8), w_shape=(128, 128, 1, 1), bias=False),                      # one group, 2 steps
    dict(x_shape=(5, 512, 4, 8), w_shape=(512, 128, 1, 1), groups=4, in_shuffle=4),          # 5 steps, four groups
]
# planes 16 and 32 pixels wide (un-pooled only): the border classes of mn_conv2d_bwd_bnh_up9 inside a lane's run of 16 pixels
PWB_CASES_WIDE = [
    dict(x_shape=(1, 128, 4, 16), w_shape=(128, 128, 1, 1)),
    dict(x_shape=(1, 128, 2, 32), w_shape=(128, 128, 1, 1), bias=False),
]


def check_pwb(be, pooled_too=True, light=False):
    """light (the CPU emulator, where a case takes ~20 s): every geometry once and every mode once -- case 0 training, case 1 eval, case 2 pooled -- instead of the full
    cross product the GPU runs."""
    before, before9 = getattr(_check_pwb, "count", 0), getattr(_check_pwb_up9, "count", 0)
    runs = 0
    for i, case in enumerate(PWB_CASES):
        modes = [("train", i == 0), ("eval", i == 1), ("pooled", i == 2)] if light else [("train", True), ("eval", True), ("pooled", pooled_too)]
        for mode, on in modes:
            if not on:
                continue
            runs += 1
            if mode == "train":
                check_qconv_bnsign(be, seed=400 + i, stash=True, **case)
            elif mode == "eval":
                check_qconv_bnsign(be, seed=410 + i, stash=True, training=False, **case)
            else:
                check_qconv_bnsign(be, seed=420 + i, stash=True, pooled=True, **case)
    before_ps = getattr(check_qconv_bnsign, "pool_sign_checked", 0)
    for i, case in enumerate(PWB_CASES_WIDE):
        check_qconv_bnsign(be, seed=430 + i, stash=True, **case)
    assert getattr(check_qconv_bnsign, "pool_sign_checked", 0) - before_ps == len(PWB_CASES_WIDE), "mn_qconv_bnsign_fwd_stash_pool did not run on the wide planes"
    assert getattr(_check_pwb, "count", 0) - before == runs + len(PWB_CASES_WIDE), "k_pwb did not take these geometries"
    assert getattr(_check_pwb_up9, "count", 0) - before9 >= len(PWB_CASES_WIDE) + (2 if light else 2 * len(PWB_CASES)), "mn_conv2d_bwd_bnh_up9 did not run on the un-pooled cases"


def check_wgrad_spec(be):
    for i, case in enumerate(WGRAD_SPEC_CASES):
        check_qconv_bnsign(be, seed=300 + i, stash=True, **case)
        check_qconv_bnq(be, seed=310 + i, **case)


def check_iao_bnfold(be, O_=24, K_=37, bias=True, shared_var=True, seed=0):
    """mn_iao_bnfold_fwd / _bwd vs an fp64 evaluation of wqaq/iao/quantize.py:900-956's fold expressions and their analytic gradients."""
    r = np.random.default_rng(seed)
    w = (r.standard_normal((O_, K_)) * 0.2).astype(F)
    b = (r.standard_normal(O_) * 0.1).astype(F) if bias else None
    gamma, beta = (r.random(O_) + 0.5).astype(F), (r.standard_normal(O_) * 0.2).astype(F)
    mean, vb = (r.standard_normal(O_) * 0.3).astype(F), (r.random(O_) * 2 + 0.1).astype(F)
    vw = vb if shared_var else (r.random(O_) * 2 + 0.1).astype(F)
    eps = 1e-5
    dwf, dbf = r.standard_normal((O_, K_)).astype(F), r.standard_normal(O_).astype(F)
    D = np.float64
    kw, kb = gamma.astype(D) / np.sqrt(vw.astype(D) + eps), gamma.astype(D) / np.sqrt(vb.astype(D) + eps)
    wf_ref = w.astype(D) * kw[:, None]
    bm = (b.astype(D) if bias else 0.0) - mean.astype(D)
    bf_ref = beta.astype(D) + bm * kb
    S = (dwf.astype(D) * w.astype(D)).sum(axis=1)
    Dd = dbf.astype(D) * bm
    ref = dict(dw=dwf.astype(D) * kw[:, None], dbias=dbf.astype(D) * kb, dgamma=S / np.sqrt(vw.astype(D) + eps) + Dd / np.sqrt(vb.astype(D) + eps), dbeta=dbf.astype(D),
               dmean=-dbf.astype(D) * kb, dvb=Dd * gamma.astype(D) * -0.5 * (vb.astype(D) + eps) ** -1.5, dvw=S * gamma.astype(D) * -0.5 * (vw.astype(D) + eps) ** -1.5)
    dW, dB = be.to_dev(w), (be.to_dev(b) if bias else None)
    dG, dBe, dM, dVb, dVw = be.to_dev(gamma), be.to_dev(beta), be.to_dev(mean), be.to_dev(vb), be.to_dev(vw)
    wf, bf = be.empty((O_, K_)), be.empty(O_)
    be.call("mn_iao_bnfold_fwd", be.ptr(dW), be.ptr(dB), be.ptr(dG), be.ptr(dBe), be.ptr(dM), be.ptr(dVb), be.ptr(dVw), eps, O_, K_, be.ptr(wf), be.ptr(bf), be.stream)
    assert close(be.to_host(wf), wf_ref, 2e-7) and close(be.to_host(bf), bf_ref, 5e-7)
    out = {k: be.empty((O_, K_) if k == "dw" else O_) for k in ("dw", "dbias", "dgamma", "dbeta", "dmean", "dvb", "dvw")}
    dDwf, dDbf = be.to_dev(dwf), be.to_dev(dbf)
    be.call("mn_iao_bnfold_bwd", be.ptr(dDwf), be.ptr(dDbf), be.ptr(dW), be.ptr(dB), be.ptr(dG), be.ptr(dM), be.ptr(dVb), be.ptr(dVw), eps, O_, K_,
            be.ptr(out["dw"]), be.ptr(out["dbias"]) if bias else None, be.ptr(out["dgamma"]), be.ptr(out["dbeta"]), be.ptr(out["dmean"]), be.ptr(out["dvb"]),
            be.ptr(out["dvw"]), be.stream)
    for k, v in ref.items():
        if k == "dbias" and not bias:
            continue
        assert close(be.to_host(out[k]), v, 1e-6), k


def check_qa_code_exact(be, bits=2, seed=0):
    """The k-bit activation code of mn_qa_fwd, j = rha(clamp(0.1 a, 0, 1) / s) (wqaq/dorefa/quantize.py:43-45), bit-exact on inputs AT the rounding
    boundaries (k + 0.5) s / 0.1 and a few ulps around them, at the clamp edges, and on random values: the kernel takes a multiply-based shortcut and
    evaluates the division only next to a boundary."""
    r = np.random.default_rng(seed)
    n = (1 << bits) - 1
    s32 = F(1.0) / F(n)
    vals = [F(0), F(-1), F(10), F(10.000001), F(9.999999), F(1e-8), F(1e9)]
    for k in range(n):
        a0 = F((k + 0.5) * float(s32) / 0.1)
        u = a0.view(np.int32) if hasattr(a0, "view") else np.float32(a0).view(np.int32)
        for d in range(-6, 7):
            vals.append(np.int32(int(u) + d).view(np.float32))
    vals = np.array(vals, dtype=F)
    rnd = (r.random(8192 - vals.size % 8192) * 11 - 0.5).astype(F)
    a = np.concatenate([vals, rnd]).astype(F)
    a = a[: (a.size // 64) * 64]
    N, Cc, H, W = 1, 1, a.size // 8, 8
    y = a.reshape(N, Cc, H, W)
    save = np.array([[0.0], [1.0]], dtype=F)
    chan = be.empty((9, Cc))
    be.call("mn_qa_chan_from_save", be.ptr(be_keep(be, save)), be.ptr(be_keep(be, np.ones(Cc, dtype=F))), be.ptr(be_keep(be, np.zeros(Cc, dtype=F))), Cc, be.ptr(chan), be.stream)
    codes = be.empty_u8((N, Cc, H, W)) if hasattr(be, "empty_u8") else None
    dY = be.to_dev(y)
    if codes is None:
        codes = be.to_dev_u8(np.zeros((N, Cc, H, W), dtype=np.uint8))
    be.call("mn_qa_fwd", 1, be.ptr(dY), be.ptr(chan), N, Cc, H, W, bits, 0, be.ptr(codes), None, be.stream)
    got = be.to_host(codes).reshape(-1).astype(np.int64)
    z = ((y.reshape(-1) - F(0)) * F(1)) * F(1) + F(0)
    act = np.where(z > 0, z, F(0)).astype(F)
    c = np.minimum(np.maximum((act * F(0.1)).astype(F), F(0)), F(1)).astype(F)
    ref = np.floor(((c / s32).astype(F) + F(0.5)).astype(F)).astype(np.int64)
    assert np.array_equal(got, ref), (bits, np.nonzero(got != ref)[0][:8])


_KEEP = []


def be_keep(be, arr):
    t = be.to_dev(arr)
    _KEEP.append(t)
    return t


def check_qa_thresholds(be, bits=2, pool=False, seed=0):
    """mn_qa_fwd on the int16 stash (integer-threshold path for <= 3-bit codes) vs the element-wise fp32 chain in numpy: random stash values over the whole
    int16 range and around every code boundary, channels with positive, negative, tiny and huge slopes, saturated and all-zero channels."""
    r = np.random.default_rng(seed)
    n = (1 << bits) - 1
    s32 = F(1.0) / F(n)
    Cc, N, H, W = 12, 2, 8, 16
    alpha = (r.random(Cc) * 0.02 + 0.001).astype(F)
    bias = (r.standard_normal(Cc) * 0.1).astype(F)
    mean = (r.standard_normal(Cc) * 0.5).astype(F)
    invstd = (r.random(Cc) * 3 + 0.2).astype(F)
    ga = (r.standard_normal(Cc) * 1.5).astype(F)              # both signs: decreasing chains too
    beb = (r.standard_normal(Cc) * 2 + 3).astype(F)
    ga[0], beb[0] = F(0), F(5)                                 # constant channel (code fixed)
    ga[1], beb[1] = F(1e-6), F(-1)                             # never positive
    alpha[2], invstd[2], ga[2] = F(1.0), F(10.0), F(-3.0)      # steep, decreasing
    chan = np.stack([alpha, bias, mean, invstd, ga, beb, alpha * invstd, (bias - mean) * invstd, ga * invstd]).astype(F)
    st = r.integers(-32768, 32768, size=(N, Cc, H, W)).astype(np.int16)
    st[:, :, :, : W // 2] = r.integers(-600, 600, size=(N, Cc, H, W // 2)).astype(np.int16)      # where the boundaries of typical channels are
    st[0, :, 0, 0], st[0, :, 0, 1] = -32768, 32767
    v = st.astype(F)
    sh = (1, Cc, 1, 1)
    y = (v * alpha.reshape(sh)).astype(F) + bias.reshape(sh)
    zh = ((y - mean.reshape(sh)).astype(F) * invstd.reshape(sh)).astype(F)
    z = ((zh * ga.reshape(sh)).astype(F) + beb.reshape(sh)).astype(F)
    a = np.where(z > 0, z, F(0)).astype(F)
    if pool:
        a = a.reshape(N, Cc, H // 2, 2, W // 2, 2).max(axis=(3, 5))
    c = np.minimum(np.maximum((a * F(0.1)).astype(F), F(0)), F(1)).astype(F)
    ref = np.floor(((c / s32).astype(F) + F(0.5)).astype(F)).astype(np.int64)
    dS = be.to_dev_i16(st) if hasattr(be, "to_dev_i16") else None
    dC = be.to_dev(chan)
    out = be.to_dev_u8(np.zeros(ref.shape, dtype=np.uint8))
    be.call("mn_qa_fwd", 0, be.ptr(dS), be.ptr(dC), N, Cc, H, W, bits, int(pool), be.ptr(out), None, be.stream)
    got = be.to_host(out).astype(np.int64)
    assert np.array_equal(got, ref), (bits, pool, np.argwhere(got != ref)[:5])


def check_qa_interval_masks(be, in_kind=0, quant=1, bits=2, seed=0):
    """The backward masks of the k-bit block as ONE interval of the streamed value per channel (qa_mask_interval: mn_qa_bwd_sums / mn_qa_bwd_apply, non-pooled) against
    the element-wise fp32 chain in numpy, BIT FOR BIT: channels with positive, negative, tiny and huge slopes, a constant channel (gamma = 0: element-wise fall-back),
    channels that never / always pass, values on both sides of every boundary.  in_kind 0: int16 stash, 2: int32 stash, 1: fp32 y."""
    r = np.random.default_rng(seed)
    n = (1 << bits) - 1
    s32 = F(1.0) / F(n)
    Cc, N, H, W = 12, 2, 8, 16
    alpha = (r.random(Cc) * 0.02 + 0.001).astype(F)
    bias = (r.standard_normal(Cc) * 0.1).astype(F)
    mean = (r.standard_normal(Cc) * 0.5).astype(F)
    invstd = (r.random(Cc) * 3 + 0.2).astype(F)
    ga = (r.standard_normal(Cc) * 1.5).astype(F)
    beb = (r.standard_normal(Cc) * 2 + 1).astype(F)
    ga[0], beb[0] = F(0), F(5)                                 # constant channel: element-wise fall-back (always passes the ReLU, clamp decides)
    ga[1], beb[1] = F(1e-6), F(-1)                             # never positive
    alpha[2], invstd[2], ga[2] = F(1.0), F(10.0), F(-3.0)      # steep, decreasing
    ga[3], beb[3] = F(1e-4), F(3.0)                            # always inside (0, 10]
    ga[4], beb[4] = F(2.0), F(50.0)                            # mostly beyond the clamp
    if in_kind == 1:
        alpha[:] = 1; bias[:] = 0
    chan = np.stack([alpha, bias, mean, invstd, ga, beb, alpha * invstd, (bias - mean) * invstd, ga * invstd]).astype(F)
    sh = (1, Cc, 1, 1)
    if in_kind == 1:
        v = (r.standard_normal((N, Cc, H, W)) * 4).astype(F)
        # values right at the channel's boundaries: solve z = 0 and 0.1 z = 1 for y in fp64, then step a few ulps around the fp32 neighbours
        for c in range(Cc):
            if ga[c] == 0:
                continue
            for zt in (0.0, 10.0):
                y0 = F((zt - float(beb[c])) / float(ga[c]) / float(invstd[c]) + float(mean[c]))
                u = np.float32(y0).view(np.int32)
                for d in range(-8, 9):
                    v[d % N, c, (d + 8) % H, (3 * d + (5 if zt else 0)) % W] = np.int32(int(u) + d).view(np.float32)
        dev_in = be.to_dev(v)
    else:
        lim = 32768 if in_kind == 0 else (1 << 24)
        st = r.integers(-lim, lim, size=(N, Cc, H, W)).astype(np.int64)
        st[:, :, :, : W // 2] = r.integers(-600, 600, size=(N, Cc, H, W // 2))          # where the boundaries of typical channels are
        for c in range(Cc):                                                            # ... and the exact neighbourhood of each channel's boundaries
            if ga[c] == 0 or alpha[c] == 0:
                continue
            for zt in (0.0, 10.0):
                v0 = int(round((((zt - float(beb[c])) / float(ga[c])) / float(invstd[c]) + float(mean[c]) - float(bias[c])) / float(alpha[c])))
                for d in range(-6, 7):
                    st[d % N, c, (d + 6) % H, (2 * d + (7 if zt else 0)) % W] = min(max(v0 + d, -lim), lim - 1)
        st[0, :, 0, 0], st[0, :, 0, 1] = -lim, lim - 1
        v = st.astype(F)
        dev_in = be.to_dev_i16(st.astype(np.int16)) if in_kind == 0 else be.to_dev(st.astype(np.int32).view(F))
    y = v if in_kind == 1 else ((v * alpha.reshape(sh)).astype(F) + bias.reshape(sh)).astype(F)
    zh = ((y - mean.reshape(sh)).astype(F) * invstd.reshape(sh)).astype(F)
    z = ((zh * ga.reshape(sh)).astype(F) + beb.reshape(sh)).astype(F)
    a = np.where(z > 0, z, F(0)).astype(F)
    dq = r.standard_normal((N, Cc, H, W)).astype(F)
    if quant:
        t = (a * F(0.1)).astype(F)
        d = ((dq * s32).astype(F) / s32).astype(F)
        d = (np.where((t >= 0) & (t <= 1), d, F(0)) * F(0.1)).astype(F)
    else:
        d = dq
    dz = np.where(z > 0, d, F(0)).astype(F)
    assert 0 < (dz != 0).mean() < 1
    sums = np.stack([r.standard_normal(Cc), r.standard_normal(Cc)]).astype(F) * 10
    nf = F(N) * F(H * W)
    k1, k2 = (sums[0] / nf).astype(F), (sums[1] / nf).astype(F)
    gi = chan[8]
    dy_ref = (gi.reshape(sh) * ((dz - k1.reshape(sh)).astype(F) - (zh * k2.reshape(sh)).astype(F)).astype(F)).astype(F)
    dC, dQ, dS = be.to_dev(chan), be.to_dev(dq), be.to_dev(sums)
    dy = be.empty((N, Cc, H, W))
    be.call("mn_qa_bwd_apply", in_kind, be.ptr(dev_in), be.ptr(dC), be.ptr(dS), be.ptr(dQ), N, Cc, H, W, bits, 0, int(quant), 1, be.ptr(dy), be.stream)
    got = be.to_host(dy)
    assert np.array_equal(got, dy_ref), (in_kind, quant, np.argwhere(got != dy_ref)[:5], float(np.abs(got - dy_ref).max()))
    # the sums pass sees the same dz: sum dz and sum dz * zhat per channel (fp64 of fp32 terms)
    ws = be.empty(int(be.lib.mn_qa_ws_floats(Cc)) + 8)
    dgam, dbet, sm = be.empty(Cc), be.empty(Cc), be.empty((2, Cc))
    be.call("mn_qa_bwd_sums", in_kind, be.ptr(dev_in), be.ptr(dC), be.ptr(dQ), N, Cc, H, W, bits, 0, int(quant), be.ptr(dgam), be.ptr(dbet), be.ptr(sm), be.ptr(ws), be.stream)
    s1 = dz.astype(np.float64).sum(axis=(0, 2, 3))
    sc1 = np.abs(dz).astype(np.float64).sum(axis=(0, 2, 3)).max()
    assert np.max(np.abs(be.to_host(dbet) - s1)) <= 1e-6 * max(sc1, 1e-30), "sum dz"


# ----------------------------------------------------------------------------- dense layers on activation codes (qgemm_dense.hip): the ResNet family
def check_qdense(be, x_shape, Oc, k=3, stride=1, a_bits=2, w_bits=2, seed=0, prepack=False):
    """mn_qconv_bnq_fwd_stash / mn_conv2d_bwd_data / mn_conv2d_bwd_weight on activation codes for a DENSE layer (groups = 1, C and O multiples of 64; 3x3 stride 1 / 2,
    1x1 stride 2: models/resnet.py:7-65 under wqaq/dorefa/quantize.py:107-122): the stash (16 or 32 bits by mn_qconv_bnq_stash_bits) equals the exact integer conv,
    the batch statistics follow, and both gradients agree with an fp64 evaluation of torch's conv backward to the float-accumulate tolerance."""
    import torch
    r = np.random.default_rng(seed)
    N, Cin, H, W = x_shape
    pad = 1 if k == 3 else 0
    w_shape = (Oc, Cin, k, k)
    na, nw = 2 ** a_bits - 1, 2 ** w_bits - 1
    codes = r.integers(0, na + 1, size=x_shape).astype(np.uint8)
    kw_ = r.integers(0, nw + 1, size=w_shape)
    wcode = (2 * kw_ - nw).astype(np.float64)
    w = (F(2.0) * (kw_.astype(F) * F(1.0 / nw)) - F(1.0)).astype(F)
    g = be.geom(x_shape, w_shape, stride=stride, padding=pad)
    wq = be.wq(mode=2, bits=w_bits)
    assert be.lib.mn_qconv_bnq_supported(C.byref(g), C.byref(wq), a_bits) == 1, "dense layer not supported"
    sb = int(be.lib.mn_qconv_bnq_stash_bits(C.byref(g), C.byref(wq), a_bits))
    assert sb == (32 if Cin * k * k * na * nw > 32767 else 16)
    acc = torch.nn.functional.conv2d(torch.from_numpy(codes.astype(np.float64)), torch.from_numpy(wcode), None, stride, pad).numpy()
    Ho, Wo = acc.shape[2:]
    nb = int(be.lib.mn_qconv_bnq_ws_bytes(C.byref(g)))
    ws = be.empty(nb // 4 + 8)
    dX, dW = be.to_dev_u8(codes), be.to_dev(w)
    if prepack:         # the weight codes written once by mn_qd_pack_multi (both fragment orders) instead of by every call
        pb = int(be.lib.mn_qd_packed_bytes(C.byref(g)))
        assert pb == (Oc * Cin * k * k * 2 + 255) // 256 * 256
        pk_f, pk_b = be.empty_i8((pb,)), be.empty_i8((pb,))
        PA, LA = C.c_void_p * 1, C.c_int64 * 1
        be.call("mn_qd_pack_multi", PA(be.ptr(dW).value), PA(be.ptr(pk_f).value), PA(be.ptr(pk_b).value), LA(Oc), LA(Cin), LA(k * k), None, None, 1, w_bits, be.stream)
        wq.packed_fwd, wq.packed_bwd = be.ptr(pk_f).value, be.ptr(pk_b).value
        dW_call = be.to_dev(np.full(w_shape, np.nan, dtype=F))       # the fp32 weights must not be read again
    else:
        dW_call = dW
    gamma, beta = (r.standard_normal(Oc) * 0.5 + 1).astype(F), (r.standard_normal(Oc) * 0.3).astype(F)
    rm, rv = be.to_dev(np.zeros(Oc)), be.to_dev(np.ones(Oc))
    save, chan = be.empty((2, Oc)), be.empty((9, Oc))
    stash = be.empty_i8((N, Oc, Ho, Wo * (sb // 8)))
    nbt = be.to_dev_i64([0])
    dGa, dBe = be.to_dev(gamma), be.to_dev(beta)
    be.call("mn_qconv_bnq_fwd_stash", C.byref(g), C.byref(wq), be.ptr(dX), a_bits, be.ptr(dW_call), None, be.ptr(dGa), be.ptr(dBe), 1e-5, 0.1, 1,
            be.ptr(rm), be.ptr(rv), be.ptr(nbt), be.ptr(save), be.ptr(stash), be.ptr(chan), be.ptr(ws), nb, be.stream)
    # weight codes of <= 7 bits fit signed bytes: the forward runs on the int8 matrix cores (k_qd_fwd8), else on bf16 (k_qd_fwd)
    assert be.lib.mn_last_kernel().decode().startswith("k_qd_fwd8<" if w_bits <= 7 else "k_qd_fwd<"), be.lib.mn_last_kernel()
    st = be.to_host(stash).view(np.int16 if sb == 16 else np.int32).reshape(N, Oc, Ho, Wo)
    assert np.array_equal(st.astype(np.float64), acc), "stash != exact integer conv result"
    alpha = float(F(F(1.0 / nw) * F(1.0 / na)))
    ye = acc * alpha
    sv = be.to_host(save)
    assert np.max(np.abs(sv[0] - ye.mean(axis=(0, 2, 3)))) <= 2e-6 * max(np.abs(ye.mean(axis=(0, 2, 3))).max(), 1e-3) + 1e-7, "mean"
    assert np.max(np.abs(sv[1] * np.sqrt(ye.var(axis=(0, 2, 3)) + 1e-5) - 1)) <= 5e-6, "invstd"
    assert int(be.to_host(nbt)[0]) == 1
    gy = r.standard_normal((N, Oc, Ho, Wo)).astype(F)
    dGY = be.to_dev(gy)
    aq = be.actq(4, a_bits)
    tx = torch.from_numpy(codes.astype(np.float64) / na).requires_grad_(True)
    tw = torch.from_numpy(w.astype(np.float64)).requires_grad_(True)
    torch.nn.functional.conv2d(tx, tw, None, stride, pad).backward(torch.from_numpy(gy.astype(np.float64)))
    dx = be.empty(x_shape)
    nb1 = int(be.lib.mn_conv2d_ws_bytes(C.byref(g), 1, 0))
    ws1 = be.empty(nb1 // 4 + 8)
    be.call("mn_conv2d_bwd_data", C.byref(g), C.byref(aq), C.byref(wq), be.ptr(dGY), be.ptr(dW_call), None, be.ptr(dx), be.ptr(ws1), nb1, 0, be.stream)
    assert be.lib.mn_last_kernel().decode().startswith("k_qd_dgrad"), be.lib.mn_last_kernel()
    assert close(be.to_host(dx), tx.grad.numpy(), 1e-5), "dx"
    dw = be.empty(w_shape)
    nb2 = int(be.lib.mn_conv2d_ws_bytes(C.byref(g), 2, 0))
    ws2 = be.empty(nb2 // 4 + 8)
    be.call("mn_conv2d_bwd_weight", C.byref(g), C.byref(aq), be.ptr(dGY), be.ptr(dX), be.ptr(dw), None, be.ptr(ws2), nb2, 0, be.stream)
    # stride 1 / 3 x 3 with power-of-two images: the straight-line second organisation (round 6); the strided layers keep k_qd_wgrad
    assert be.lib.mn_last_kernel().decode().startswith("k_qd_wgrad2<" if (k == 3 and stride == 1) else "k_qd_wgrad<"), be.lib.mn_last_kernel()
    assert close(be.to_host(dw), tw.grad.numpy(), 1e-5), "dw"


# (x shape, O, k, stride): small enough for the CPU emulator; every tile shape of the planners (W = 4 .. 32, images per tile > 1 with N not a multiple, several chunks,
# several output / input channel tiles, stride 2, the 1x1 shortcut, the 32-bit stash)
QDENSE_CASES = [
    ((2, 64, 8, 8), 64, 3, 1), ((3, 128, 4, 4), 64, 3, 1), ((1, 64, 16, 16), 128, 3, 1), ((1, 64, 8, 32), 64, 3, 1), ((5, 128, 8, 8), 128, 3, 2),
    ((1, 64, 32, 32), 64, 3, 2), ((2, 64, 16, 16), 128, 1, 2), ((2, 128, 8, 8), 64, 1, 2), ((1, 512, 4, 4), 64, 3, 1),
    ((2, 64, 32, 32), 64, 3, 1),          # a band of rows per tile (halo rows hold data): k_qd_wgrad2's 20-unit patch
]


def check_hsign_fold(be, kxk_cases, full=False):
    """The default since round 5 (MN_HSIGN_FOLD=0 turns it off): k_pws_stats_prep's work -- batch statistics from the partial rows, running statistics, the integer thresholds,
    nnz, the counter -- evaluated inside the streaming sign pass (k_h_sign_prep: every block for itself, block 0 of a channel writes): the stashed pointwise and
    3 x 3 blocks against the same references as the two-launch path (sign codes and stash bit for bit, statistics, counter, and the backward that reads `chan`)."""
    for case in ((0, 1, 2, 3) if full else (1,)):
        check_qconv_bnsign(be, seed=220 + case, stash=True, **QGEMM_PW_CASES[case])
        assert check_qconv_bnsign.last_fwd_kernel == "k_h_sign_prep", check_qconv_bnsign.last_fwd_kernel
        check_qconv_bnsign(be, seed=225 + case, stash=True, training=False, **QGEMM_PW_CASES[case])        # eval: the sign comes from the MFMA pass, nothing to fold
        if case in (1, 2):
            check_qconv_bnsign(be, seed=230 + case, stash=True, pooled=True, **QGEMM_PW_CASES[case])
    check_qconv_bnsign(be, seed=181, stash=True, x_shape=(2, 256, 4, 8), w_shape=(256, 128, 1, 1), groups=2, in_shuffle=2)          # KS = 4, two m-blocks: the nin_gc pattern
    seen = []
    for i, kw in enumerate(kxk_cases if full else kxk_cases[:2]):
        for training in (True, False):
            check_qconv_bnsign(be, seed=260 + i, stash=True, training=training, **kw)
            seen.append(check_qconv_bnsign.last_fwd_kernel)
    assert seen.count("k_h_sign_prep") >= 4, seen          # the staged-image 3 x 3 path (nin_gc's pattern) in both modes; the generic k x k path keeps two launches


CHILD = r"""
import sys
sys.path.insert(0, sys.argv[1])
import abi_driver
import kernel_cases as K
be = abi_driver.Backend(sys.argv[2])
%s
print("child ok")
"""


def run_child(body, backend, env, timeout):
    """Run `body` (python statements using `be` and `K`) in a child process with extra environment: the library reads its MN_* knobs once per process."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-c", CHILD % body, here, backend], env=dict(os.environ, **env), capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0 and "child ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def check_qg_pack_multi(be, seed=0):
    """mn_qg_pack_multi: the forward and backward-data weight-code images of several pointwise layers in ONE launch; the entry points handed those images
    (mn_wq.packed_fwd / packed_bwd) must return BIT-IDENTICAL results to the calls that pack for themselves -- sign-code blocks (ternary weights) and k-bit blocks
    (DoReFa weights), the backward-data with the BatchNorm fold included.  The poisoned `w` of the pre-packed calls proves the images are what is read."""
    r = np.random.default_rng(seed)
    layers = [  # (x_shape, w_shape, groups, in_shuffle, scheme)    scheme 1: sign codes x ternary weights, 2: 2-bit codes x DoReFa W2
        ((3, 80, 8, 16), (96, 40, 1, 1), 2, 2, 1), ((2, 96, 8, 8), (96, 48, 1, 1), 2, 0, 1), ((4, 160, 8, 8), (192, 80, 1, 1), 2, 0, 1),
        ((2, 96, 8, 8), (96, 48, 1, 1), 2, 0, 2), ((3, 80, 8, 16), (96, 40, 1, 1), 2, 2, 2),
    ]
    G, WQ_, W_, WH, OUT, keep, per = [], [], [], [], [], [], []
    for (xs, wsh, groups, shuf, scheme) in layers:
        g = be.geom(xs, wsh, groups=groups)
        g.in_shuffle = shuf
        if scheme == 1:
            w, wkw, _ = make_coded_weights(r, wsh, 1)
        else:
            w, wkw, _ = make_coded_weights(r, wsh, 2, 2)
        dW = be.to_dev(w)
        imgs = []
        for which in (0, 1):
            nb = int(be.lib.mn_qg_packed_bytes(C.byref(g), which))
            assert nb > 0, (xs, wsh, which)
            buf = be.empty(nb // 4 + 4)
            wq = be.wq(**wkw)
            G.append(g); WQ_.append(wq); W_.append(dW); WH.append(which); OUT.append(buf); imgs.append(buf)
        keep.append(dW)
        per.append((xs, wsh, groups, shuf, scheme, g, wkw, dW, w, imgs))
    n = len(G)
    GP, WP, PA, IA = C.POINTER(type(G[0])) * n, C.POINTER(type(WQ_[0])) * n, C.c_void_p * n, C.c_int32 * n
    be.call("mn_qg_pack_multi", n, GP(*[C.pointer(g) for g in G]), WP(*[C.pointer(q) for q in WQ_]), PA(*[be.ptr(w) for w in W_]), IA(*WH), PA(*[be.ptr(o) for o in OUT]),
            be.stream)
    for (xs, wsh, groups, shuf, scheme, g, wkw, dW, w, imgs) in per:
        N, Cin, H, W = xs
        Oc = wsh[0]
        dWn = be.to_dev(np.full(wsh, np.nan, dtype=F))
        gy = be.to_dev(r.standard_normal((N, Oc, H, W)).astype(F))
        nb1 = int(be.lib.mn_conv2d_ws_bytes(C.byref(g), 1, 0))
        res = []
        for packed in (False, True):
            wq = be.wq(**wkw)
            if packed:
                wq.packed_fwd, wq.packed_bwd = be.ptr(imgs[0]).value, be.ptr(imgs[1]).value
            wsrc = dWn if packed else dW
            out = {}
            if scheme == 1:
                a_in = np.where(np.random.default_rng(7).standard_normal(xs) > 0, 1, -1).astype(np.int8)
                dA = be.to_dev_i8(a_in)
                nb = max(int(be.lib.mn_qconv_bnsign_stash_ws_bytes(C.byref(g))), 4 * int(be.lib.mn_bnsign_ws_floats(Oc)))
                ws = be.empty(nb // 4 + 8)
                gam, bet = be.to_dev(np.linspace(0.5, 1.5, Oc).astype(F)), be.to_dev(np.linspace(-0.3, 0.3, Oc).astype(F))
                rm, rv = be.to_dev(np.zeros(Oc, dtype=F)), be.to_dev(np.ones(Oc, dtype=F))
                save, a8, h8 = be.empty((2, Oc)), be.empty_i8((N, Oc, H, W)), be.empty_i8((N, Oc, H, W))
                chan = be.empty((int(be.lib.mn_qconv_bnsign_stash_chan_rows(C.byref(g))), Oc))
                be.call("mn_qconv_bnsign_fwd_stash", C.byref(g), C.byref(wq), be.ptr(dA), be.ptr(wsrc), None, be.ptr(gam), be.ptr(bet), 1e-5, 0.1, 1, be.ptr(rm), be.ptr(rv),
                        None, be.ptr(save), be.ptr(a8), be.ptr(h8), be.ptr(chan), be.ptr(ws), nb, be.stream)
                out["a"], out["h"] = be.to_host(a8).copy(), be.to_host(h8).copy()
                sums, dgam, dbet = be.empty((2, Oc)), be.empty(Oc), be.empty(Oc)
                ws3 = be.empty(int(be.lib.mn_bnsign_ws_floats(Oc)) + 8)
                be.call("mn_bnh_bwd_sums", be.ptr(gy), be.ptr(h8), None, be.ptr(chan), N, Oc, H, W, be.ptr(dgam), be.ptr(dbet), be.ptr(sums), be.ptr(ws3), be.stream)
                ws1, dx = be.empty(max(4, nb1 // 4 + 4)), be.empty(xs)
                be.call("mn_conv2d_bwd_data_bnh", C.byref(g), C.byref(wq), be.ptr(gy), be.ptr(h8), be.ptr(chan), be.ptr(sums), 1, be.ptr(wsrc), be.ptr(dx), be.ptr(ws1), nb1,
                        be.stream)
                out["dx_bnh"] = be.to_host(dx).copy()
                aq = be.actq(3)
            else:
                codes = np.random.default_rng(8).integers(0, 4, size=xs).astype(np.uint8)
                dX = be.to_dev_i8(codes.view(np.int8))
                nb = max(int(be.lib.mn_qconv_bnq_ws_bytes(C.byref(g))), 4 * int(be.lib.mn_qa_ws_floats(Oc)))
                ws = be.empty(nb // 4 + 8)
                gam, bet = be.to_dev(np.linspace(0.5, 1.5, Oc).astype(F)), be.to_dev(np.linspace(-0.3, 0.3, Oc).astype(F))
                rm, rv = be.to_dev(np.zeros(Oc, dtype=F)), be.to_dev(np.ones(Oc, dtype=F))
                save, chan, stash = be.empty((2, Oc)), be.empty((9, Oc)), be.empty_i8((N, Oc, H, 2 * W))
                be.call("mn_qconv_bnq_fwd_stash", C.byref(g), C.byref(wq), be.ptr(dX), 2, be.ptr(wsrc), None, be.ptr(gam), be.ptr(bet), 1e-5, 0.1, 1, be.ptr(rm), be.ptr(rv),
                        None, be.ptr(save), be.ptr(stash), be.ptr(chan), be.ptr(ws), nb, be.stream)
                out["stash"], out["chan"] = be.to_host(stash).copy(), be.to_host(chan).copy()
                aq = be.actq(4, 2)
            ws1, dx = be.empty(max(4, nb1 // 4 + 4)), be.empty(xs)
            be.call("mn_conv2d_bwd_data", C.byref(g), C.byref(aq), C.byref(wq), be.ptr(gy), be.ptr(wsrc), None, be.ptr(dx), be.ptr(ws1), nb1, 0, be.stream)
            out["dx"] = be.to_host(dx).copy()
            res.append(out)
        for k in res[0]:
            assert np.array_equal(res[0][k], res[1][k], equal_nan=True), ("pre-packed != self-packed", k, xs, wsh, scheme)
            assert np.isfinite(res[1][k].astype(np.float64)).all(), ("NaN from the poisoned weights: the image was not used", k)


def _lib_actq_iao(be, bits, qp):
    return be.actq(2, bits, 0, qp)


def check_qr(be, shape=(3, 6, 4, 8), in_kind=0, res_kind=1, bits=2, training=True, with_dq2=False, with_gf=True, seed=0):
    """mn_qr_fwd / mn_qr_bwd_sums / mn_qr_bwd_apply -- the end of a residual block: u = bn(y) + res, a = relu(u) -> codes + fp32; backward through the ReLU, the
    clip-STE of the code readers and both BatchNorms -- vs a numpy evaluation of the same fp32 chain (models/resnet.py:60-65, wqaq/dorefa/quantize.py:36-46)."""
    r = np.random.default_rng(seed)
    N, Cc, H, W = shape
    n = N * H * W
    s_o = F(1.0 / (2 ** bits - 1))

    def make_branch(kind):          # kind 0 / 2: integer stash (int16 / int32) with chan constants; 1: fp32
        if kind == 1:
            y = (r.standard_normal(shape) * 2).astype(F)
            alpha, bias = np.ones(Cc, F), np.zeros(Cc, F)
            raw = y
        else:
            raw = r.integers(-300, 300, size=shape).astype(np.int32)
            alpha, bias = (np.abs(r.standard_normal(Cc)) * 0.01 + 0.005).astype(F), (r.standard_normal(Cc) * 0.1).astype(F)
            y = (raw.astype(F) * alpha.reshape(1, -1, 1, 1) + bias.reshape(1, -1, 1, 1)).astype(F)
        mean, inv = y.mean(axis=(0, 2, 3)).astype(F), (F(1.0) / np.sqrt(y.var(axis=(0, 2, 3)).astype(F) + F(1e-5))).astype(F)
        ga, be_ = (r.standard_normal(Cc) * 0.5 + 1).astype(F), (r.standard_normal(Cc) * 0.3).astype(F)
        chan = np.stack([alpha, bias, mean, inv, ga, be_, (alpha * inv).astype(F), ((bias - mean) * inv).astype(F), (ga * inv).astype(F)]).astype(F)
        zh = ((y - mean.reshape(1, -1, 1, 1)) * inv.reshape(1, -1, 1, 1)).astype(F)
        z = (zh * ga.reshape(1, -1, 1, 1) + be_.reshape(1, -1, 1, 1)).astype(F)
        dev = be.to_dev(raw) if kind == 1 else (be.to_dev_i16(raw.astype(np.int16)) if kind == 0 else _to_dev_i32(be, raw))
        return dev, be.to_dev(chan), zh, z, (ga * inv).astype(np.float64)

    src, chan, zh, z, gi = make_branch(in_kind)
    res_dev = res_chan = None
    zhs = gis = None
    if res_kind == 1:
        res = np.maximum(r.standard_normal(shape), 0).astype(F)
        res_dev = be.to_dev(res)
        u = (z + res).astype(F)
    elif res_kind >= 2:
        res_dev, res_chan, zhs, zs, gis = make_branch(0 if res_kind == 2 else 2)
        u = (z + zs).astype(F)
    else:
        u = z
    a = np.maximum(u, F(0)).astype(F)
    codes, act = be.empty_i8(shape), be.empty(shape)
    be.call("mn_qr_fwd", in_kind, be.ptr(src), be.ptr(chan), res_kind, be.ptr(res_dev), be.ptr(res_chan), N, Cc, H, W, bits, be.ptr(codes), be.ptr(act), be.stream)
    assert np.array_equal(be.to_host(act), a), "activation"
    c_ref = np.floor(np.abs(np.clip(a * F(0.1), 0, 1).astype(F) / s_o) + F(0.5))
    assert np.array_equal(be.to_host(codes).view(np.uint8).astype(np.float64), c_ref.astype(np.float64)), "codes"
    # backward
    dq = r.standard_normal(shape).astype(F)
    dq2 = r.standard_normal(shape).astype(F) if with_dq2 else None
    gf = r.standard_normal(shape).astype(F) if with_gf else None

    def ste(gq):
        t_ = a * F(0.1)
        d_ = ((gq * s_o) / s_o).astype(F)
        return (np.where((t_ >= 0) & (t_ <= 1), d_, F(0)) * F(0.1)).astype(F)
    da = ste(dq)
    if dq2 is not None:
        da = (da + ste(dq2)).astype(F)
    if gf is not None:
        da = (da + gf).astype(F)
    du_ref = np.where(u > 0, da, F(0)).astype(F)
    d64 = du_ref.astype(np.float64)
    s1, s2 = d64.sum(axis=(0, 2, 3)), (d64 * zh.astype(np.float64)).sum(axis=(0, 2, 3))
    du, dy, dys = be.empty(shape), be.empty(shape), (be.empty(shape) if res_kind >= 2 else None)
    dgam, dbet, sums = be.empty(Cc), be.empty(Cc), be.empty((2, Cc))
    dgam_s, dbet_s, sums_s = (be.empty(Cc), be.empty(Cc), be.empty((2, Cc))) if res_kind >= 2 else (None, None, None)
    ws = be.empty(int(be.lib.mn_qr_ws_floats(Cc)))
    dDQ, dDQ2, dGF = be.to_dev(dq), (be.to_dev(dq2) if dq2 is not None else None), (be.to_dev(gf) if gf is not None else None)      # (kept alive across the call)
    be.call("mn_qr_bwd_sums", in_kind, be.ptr(src), be.ptr(chan), res_kind, be.ptr(res_dev), be.ptr(res_chan), be.ptr(dDQ), be.ptr(dDQ2), be.ptr(dGF), N, Cc, H, W, bits,
            be.ptr(du), be.ptr(dgam), be.ptr(dbet), be.ptr(sums), be.ptr(dgam_s), be.ptr(dbet_s), be.ptr(sums_s), be.ptr(ws), be.stream)
    assert np.array_equal(be.to_host(du), du_ref), "du"
    sc = max(np.max(np.abs(d64)) * np.sqrt(n), 1e-30)
    assert np.max(np.abs(be.to_host(dbet) - s1)) <= 2e-6 * sc and np.max(np.abs(be.to_host(dgam) - s2)) <= 2e-6 * sc * max(1.0, np.abs(zh).max()), "dgamma / dbeta"
    be.call("mn_qr_bwd_apply", in_kind, be.ptr(src), be.ptr(chan), be.ptr(sums), res_kind, be.ptr(res_dev), be.ptr(res_chan), be.ptr(sums_s), be.ptr(du), N, Cc, H, W,
            int(training), be.ptr(dy), be.ptr(dys), be.stream)
    k1, k2 = (s1 / n, s2 / n) if training else (np.zeros(Cc), np.zeros(Cc))
    dy_ref = gi.reshape(1, -1, 1, 1) * (d64 - k1.reshape(1, -1, 1, 1) - zh.astype(np.float64) * k2.reshape(1, -1, 1, 1))
    assert close(be.to_host(dy), dy_ref, 1e-5), "dy"
    if res_kind >= 2:
        s3 = (d64 * zhs.astype(np.float64)).sum(axis=(0, 2, 3))
        assert np.max(np.abs(be.to_host(dgam_s) - s3)) <= 2e-6 * sc * max(1.0, np.abs(zhs).max()) and np.max(np.abs(be.to_host(dbet_s) - s1)) <= 2e-6 * sc, "shortcut dgamma / dbeta"
        k2s = s3 / n if training else np.zeros(Cc)
        dys_ref = gis.reshape(1, -1, 1, 1) * (d64 - k1.reshape(1, -1, 1, 1) - zhs.astype(np.float64) * k2s.reshape(1, -1, 1, 1))
        assert close(be.to_host(dys), dys_ref, 1e-5), "dy of the shortcut conv"
    # the two calls as two launches (mn_qr_bwd): every output to the bit
    du2, dy2, dys2 = be.empty(shape), be.empty(shape), (be.empty(shape) if res_kind >= 2 else None)
    dgam2, dbet2, sums2 = be.empty(Cc), be.empty(Cc), be.empty((2, Cc))
    dgam_s2, dbet_s2, sums_s2 = (be.empty(Cc), be.empty(Cc), be.empty((2, Cc))) if res_kind >= 2 else (None, None, None)
    be.call("mn_qr_bwd", in_kind, be.ptr(src), be.ptr(chan), res_kind, be.ptr(res_dev), be.ptr(res_chan), be.ptr(dDQ), be.ptr(dDQ2), be.ptr(dGF), N, Cc, H, W, bits,
            int(training), be.ptr(du2), be.ptr(dgam2), be.ptr(dbet2), be.ptr(sums2), be.ptr(dgam_s2), be.ptr(dbet_s2), be.ptr(sums_s2), be.ptr(dy2), be.ptr(dys2), be.ptr(ws),
            be.stream)
    for a_, b_ in ((du2, du), (dy2, dy), (dgam2, dgam), (dbet2, dbet), (sums2, sums)) + (((dys2, dys), (dgam_s2, dgam_s), (dbet_s2, dbet_s), (sums_s2, sums_s)) if res_kind >= 2 else ()):
        assert eq(be.to_host(a_), be.to_host(b_))


def _to_dev_i32(be, a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    if be.kind == "emu":
        return a.copy()
    return be.torch.from_numpy(a.copy()).cuda()


def check_qlinear(be, N=5, Cc=70, Oc=10, mode=1, bits=2, bias=True, seed=0):
    """mn_qlinear_fwd / _bwd_data / _bwd_weight (the ResNet classifier, wqaq/dorefa/quantize.py:192-199 / wqaq/iao/quantize.py:1150-1157) vs numpy; mode 0 none, 1 DoReFa, 2 IAO."""
    r = np.random.default_rng(seed)
    x = (r.standard_normal((N, Cc)) * 4).astype(F)
    w = (r.standard_normal((Oc, Cc)) * 0.3).astype(F)
    b = (r.standard_normal(Oc) * 0.2).astype(F) if bias else None
    gy = r.standard_normal((N, Oc)).astype(F)
    qp = None
    if mode == 2:
        qp_np = np.array([0.11, 0.0, -8.0, 7.0], dtype=F)
        qp = be.to_dev(qp_np)
    aq = be.actq(mode, bits, 0, qp)
    xq = _quant_x(x, mode, bits, qp_np if mode == 2 else None, 0)
    y = be.empty((N, Oc))
    dX, dW, dG = be.to_dev(x), be.to_dev(w), be.to_dev(gy)
    dB = be.to_dev(b) if bias else None
    be.call("mn_qlinear_fwd", C.byref(aq), be.ptr(dX), be.ptr(dW), be.ptr(dB), be.ptr(y), N, Cc, Oc, be.stream)
    y_ref = xq.astype(np.float64) @ w.astype(np.float64).T + (b.astype(np.float64) if bias else 0.0)
    assert close(be.to_host(y), y_ref, 1e-5), "y"
    dx = be.empty((N, Cc))
    be.call("mn_qlinear_bwd_data", C.byref(aq), be.ptr(dG), be.ptr(dW), be.ptr(dX), be.ptr(dx), N, Cc, Oc, be.stream)
    dx_ref = _ste((gy.astype(np.float64) @ w.astype(np.float64)).astype(F), x, mode, bits, qp_np if mode == 2 else None, 0)
    assert close(be.to_host(dx), dx_ref, 1e-5), "dx"
    dw, db = be.empty((Oc, Cc)), (be.empty(Oc) if bias else None)
    be.call("mn_qlinear_bwd_weight", C.byref(aq), be.ptr(dG), be.ptr(dX), be.ptr(dw), be.ptr(db), N, Cc, Oc, be.stream)
    assert close(be.to_host(dw), gy.astype(np.float64).T @ xq.astype(np.float64), 1e-5), "dw"
    if bias:
        assert close(be.to_host(db), gy.astype(np.float64).sum(axis=0), 1e-5), "db"


def check_qdense_iao(be, x_shape, Oc, k=3, stride=1, a_bits=4, w_bits=4, bias=False, seed=0):
    """The IAO QuantConv2d of the ResNets (wqaq/iao/quantize.py:492-507; symmetric activation quantizer in the conv, symmetric per-channel weights) on the dense
    kernels: mn_conv2d_fwd / _bwd_data / _bwd_weight with MN_ACTQ_IAO + MN_WQ_IAO route to k_qd_* and agree with the fp64 evaluation of the fake-quantised conv."""
    pad = 1 if k == 3 else 0
    check_conv(be, x_shape, (Oc, x_shape[1], k, k), stride=stride, padding=pad, bias=bias, mode=2, bits=a_bits, q_type=0, wmode=3, wbits=w_bits, algos=(3,), seed=seed,
               expect_qgemm=True, want_dbias=bias, expect_kernels=("k_qd_fwd", "k_qd_dgrad", "k_qd_wgrad"))
    # the activation codes of the forward kept in a caller-owned buffer (mn_actq.codes) and reused by backward-weight: same dw without reading x again
    r = np.random.default_rng(seed + 1000)
    w_shape = (Oc, x_shape[1], k, k)
    x = (r.standard_normal(x_shape) * 4).astype(F)
    w, wkw, wscale = make_coded_weights(r, w_shape, 3, w_bits)
    dWs = be.to_dev(wscale)
    wq = be.wq(scale=dWs, **wkw)
    g = be.geom(x_shape, w_shape, stride, pad)
    mn, mx = F(x.min()), F(x.max())
    sc, zp = O.iao_qparams(mn.reshape(1), mx.reshape(1), a_bits, 0, True)
    hi = max(abs(mn / sc[0]), abs(mx / sc[0]))
    dqp = be.to_dev(np.array([sc[0], zp[0], -hi, hi], dtype=F))
    aq = be.actq(2, a_bits, 0, dqp)
    nc = int(be.lib.mn_conv2d_iao_codes_bytes(C.byref(g), C.byref(aq), C.byref(wq)))
    assert nc == (int(np.prod(x_shape)) + 255) // 256 * 256
    codes = be.empty_i8((nc,))
    dX, dW = be.to_dev(x), be.to_dev(w)
    y1 = be.to_host(be.conv_fwd(g, aq, dX, dW, None, 3, wq=wq))
    aq.codes = be.ptr(codes).value
    y2 = be.to_host(be.conv_fwd(g, aq, dX, dW, None, 3, wq=wq))
    assert np.array_equal(y1, y2)
    gy = r.standard_normal(y1.shape).astype(F)
    dG = be.to_dev(gy)
    aq0 = be.actq(2, a_bits, 0, dqp)
    dw_a, _ = be.conv_bwd_weight(g, aq0, dG, dX, 3, bias=False)
    dw_b, _ = be.conv_bwd_weight(g, aq, dG, be.to_dev(np.full(x_shape, np.nan, dtype=F)), 3, bias=False)
    assert np.array_equal(be.to_host(dw_a), be.to_host(dw_b))
    # the IAO weight codes written once by mn_qd_pack_multi (codes = rint(w / scale[o])): same forward and backward-data without reading w again
    dx1 = be.to_host(be.conv_bwd_data(g, aq0, dG, dW, dX, 3, wq=wq))
    # the clip-STE decisions kept as bits by the forward (mn_actq.ste_mask) and applied by backward-data without reading x: the same dx, to the bit -- with the
    # range above (nothing clipped) and with half of it (a good part of the elements clipped: both conditions of iao_fq_grad decide)
    for shrink in (1.0, 0.5):
        sc2, zp2 = O.iao_qparams((mn * F(shrink)).reshape(1), (mx * F(shrink)).reshape(1), a_bits, 0, True)
        hi2 = max(abs(mn * F(shrink) / sc2[0]), abs(mx * F(shrink) / sc2[0]))
        dqp2 = be.to_dev(np.array([sc2[0], zp2[0], -hi2, hi2], dtype=F))
        aqx = be.actq(2, a_bits, 0, dqp2)
        dxx = be.to_host(be.conv_bwd_data(g, aqx, dG, dW, dX, 3, wq=wq))
        smask = be.empty_i8((nc // 8,))
        aqm = be.actq(2, a_bits, 0, dqp2)
        aqm.codes, aqm.ste_mask = be.ptr(codes).value, be.ptr(smask).value
        be.conv_fwd(g, aqm, dX, dW, None, 3, wq=wq)
        dxm = be.to_host(be.conv_bwd_data(g, aqm, dG, dW, be.to_dev(np.full(x_shape, np.nan, dtype=F)), 3, wq=wq))
        assert np.array_equal(dxm, dxx), ("ste_mask", shrink)
        if shrink < 1.0:
            assert 0.002 * dxx.size < np.count_nonzero(dxx == 0) < 0.9 * dxx.size          # (the clip is exercised)
    aq.codes = be.ptr(codes).value
    be.conv_fwd(g, aq, dX, dW, None, 3, wq=wq)          # (the codes of the original range again, for what follows)
    pb = int(be.lib.mn_qd_packed_bytes(C.byref(g)))
    pk_f, pk_b = be.empty_i8((pb,)), be.empty_i8((pb,))
    PA, LA, IA = C.c_void_p * 1, C.c_int64 * 1, C.c_int32 * 1
    be.call("mn_qd_pack_multi", PA(be.ptr(dW).value), PA(be.ptr(pk_f).value), PA(be.ptr(pk_b).value), LA(Oc), LA(x_shape[1]), LA(k * k), PA(be.ptr(dWs).value), IA(1), 1,
            w_bits, be.stream)
    wq.packed_fwd, wq.packed_bwd = be.ptr(pk_f).value, be.ptr(pk_b).value
    dWn = be.to_dev(np.full(w_shape, np.nan, dtype=F))
    y3 = be.to_host(be.conv_fwd(g, aq0, dX, dWn, None, 3, wq=wq))
    dx3 = be.to_host(be.conv_bwd_data(g, aq0, dG, dWn, dX, 3, wq=wq))
    assert np.array_equal(y1, y3) and np.array_equal(dx1, dx3)
    # a tensor added to dx in the store, after the clip-STE (mn_actq.dx_add: the identity shortcut's gradient of a residual block): == the separate add, to the bit
    assert be.lib.mn_conv2d_bwd_data_add_supported(C.byref(g), C.byref(aq0), C.byref(wq)) == 1
    addend = r.standard_normal(x_shape).astype(F)
    dA = be.to_dev(addend)
    aqa = be.actq(2, a_bits, 0, dqp)
    aqa.dx_add = be.ptr(dA).value
    dx4 = be.to_host(be.conv_bwd_data(g, aqa, dG, dWn, dX, 3, wq=wq))
    assert np.array_equal(dx4, dx1 + addend)
    # ---- the exact sums of the integer accumulator from the forward's epilogue (mn_actq.stats) and the BatchNorm behind the conv from them (mn_bn_fwd_acc: ONE pass)
    # == the ordinary statistics pass + apply pass over y (mn_bnrelu_fwd / mn_bn2d_fwd), to the round-off of fp32 statistics
    aqs = be.actq(2, a_bits, 0, dqp)
    rows = int(be.lib.mn_conv2d_iao_stats_rows(C.byref(g), C.byref(aqs), C.byref(wq)))
    assert rows > 0, "no epilogue statistics for this dense IAO layer"
    stats = be.to_dev(np.full((rows, Oc, 4), np.nan, dtype=F))          # rows * Oc * 2 doubles
    aqs.stats = be.ptr(stats).value
    cb = (r.standard_normal(Oc) * 0.3).astype(F) if bias else None
    dCB = be.to_dev(cb) if bias else None
    dY = be.conv_fwd(g, aqs, dX, dWn, dCB, 3, wq=wq)
    yv = be.to_host(dY)
    N_, _, Ho, Wo = yv.shape
    HW = Ho * Wo
    st = be.to_host(stats).view(np.float64).reshape(rows, Oc, 2).sum(axis=0)
    al = (F(sc[0]) * wscale.astype(F)).astype(np.float64)
    acc = (yv.astype(np.float64) - (cb.astype(np.float64).reshape(1, -1, 1, 1) if bias else 0.0)) / al.reshape(1, -1, 1, 1)
    acc = np.rint(acc)
    assert np.array_equal(st[:, 0], acc.sum(axis=(0, 2, 3))) and np.array_equal(st[:, 1], (acc * acc).sum(axis=(0, 2, 3))), "epilogue sums of acc / acc^2 are not exact"
    if HW % 4 == 0:
        gamma, beta = (r.standard_normal(Oc) * 0.5 + 1).astype(F), (r.standard_normal(Oc) * 0.3).astype(F)
        rm, rv = (r.standard_normal(Oc) * 0.1).astype(F), (np.abs(r.standard_normal(Oc)) + 0.5).astype(F)
        dGa, dBe = be.to_dev(gamma), be.to_dev(beta)
        ws = be.empty(int(be.lib.mn_bnsign_ws_floats(Oc)) + 8)
        for act, fn in ((1, "mn_bnrelu_fwd_mm"), (2, "mn_bn2d_fwd_mm")):
            cnt = int(be.lib.mn_bnrelu_mm_count(N_, Oc, HW))
            rm1, rv1, rm2, rv2 = be.to_dev(rm), be.to_dev(rv), be.to_dev(rm), be.to_dev(rv)
            s1, s2, a1, a2, mm1, mm2 = be.empty((2, Oc)), be.empty((2, Oc)), be.empty(yv.shape), be.empty(yv.shape), be.empty(2 * cnt), be.empty(2 * cnt)
            be.call(fn, be.ptr(dY), N_, Oc, HW, be.ptr(dGa), be.ptr(dBe), 1e-5, 0.1, 1, be.ptr(rm1), be.ptr(rv1), be.ptr(s1), be.ptr(a1), be.ptr(ws), be.ptr(mm1), be.stream)
            be.call("mn_bn_fwd_acc", be.ptr(dY), N_, Oc, HW, be.ptr(dGa), be.ptr(dBe), 1e-5, 0.1, be.ptr(rm2), be.ptr(rv2), be.ptr(s2), be.ptr(a2), be.ptr(mm2), act,
                    be.ptr(stats), rows, be.ptr(dqp), be.ptr(dWs), 1, be.ptr(dCB) if bias else None, be.stream)
            sv1, sv2 = be.to_host(s1), be.to_host(s2)
            assert np.max(np.abs(sv1[0] - sv2[0])) <= 2e-6 * max(np.max(np.abs(sv1[0])), 1e-3) + 1e-7, ("mean", act)
            assert np.max(np.abs(sv1[1] - sv2[1]) / sv1[1]) <= 5e-6, ("invstd", act)
            assert close(be.to_host(rm2), be.to_host(rm1), 2e-6) and close(be.to_host(rv2), be.to_host(rv1), 1e-5), ("running statistics", act)
            o1, o2 = be.to_host(a1), be.to_host(a2)
            assert np.max(np.abs(o1 - o2)) <= 1e-5 * max(np.max(np.abs(o1)), 1e-6), ("output", act, float(np.max(np.abs(o1 - o2))))
            # its (min, max) partials describe ITS output exactly
            m2 = be.to_host(mm2)
            assert m2[:cnt].min() == o2.min() and m2[cnt:].max() == o2.max(), ("min / max partials", act)
            check_bn_lazy_pull(be, r, g, aqs, wq, dX, dWn, dCB, dqp, dWs, bias, yv, dGa, dBe, rm, rv, s2, rm2, rv2, o2, act, a_bits, w_bits)


def check_bn_lazy_pull(be, r, g, aqs, wq, dX, dWn, dCB, dqp, dWs, bias, yv, dGa, dBe, rm, rv, s2, rm2, rv2, o2, act, a_bits, w_bits):
    """The BatchNorm [+ ReLU] behind a dense IAO conv as an un-computed activation whose consumer pulls (round 6; models/resnet.py:17-29 under
    wqaq/iao/quantize.py:492-507): the conv's epilogue leaves the extrema of its accumulator (mn_actq.acc_mm); mn_bn_acc_prep == mn_bn_fwd_acc's statistics bit for
    bit AND the per-channel (min, max) of the activation exactly, without a pass; mn_bn_apply_codes == the next dense conv's own code pass on the fp32 activation
    (codes and clip-STE bits, bit for bit); that conv under MN_ACTQ_CODES_GIVEN == the conv on the fp32 activation, never reading x."""
    N_, Oc, Ho, Wo = yv.shape
    HW = Ho * Wo
    rows = int(be.lib.mn_conv2d_iao_stats_rows(C.byref(g), C.byref(aqs), C.byref(wq)))
    stats = be.to_dev(np.full((rows, Oc, 4), np.nan, dtype=F))
    accmm = be.to_dev(np.full((rows, Oc, 2), np.nan, dtype=F))          # rows * Oc * 2 int32
    aqm = be.actq(2, aqs.bits, 0, dqp)
    aqm.stats, aqm.acc_mm = be.ptr(stats).value, be.ptr(accmm).value
    dY = be.conv_fwd(g, aqm, dX, dWn, dCB, 3, wq=wq)
    assert np.array_equal(be.to_host(dY), yv)
    am = be.to_host(accmm).view(np.int32).reshape(rows, Oc, 2)
    sc0 = float(be.to_host(dqp)[0])
    al = (F(sc0) * be.to_host(dWs).astype(F)).astype(np.float64)
    cbv = be.to_host(dCB).astype(np.float64).reshape(1, -1, 1, 1) if bias else 0.0
    acc = np.rint((yv.astype(np.float64) - cbv) / al.reshape(1, -1, 1, 1))
    assert np.array_equal(am[:, :, 0].min(axis=0), acc.min(axis=(0, 2, 3))) and np.array_equal(am[:, :, 1].max(axis=0), acc.max(axis=(0, 2, 3))), "extrema of acc"
    rm3, rv3, s3, mm3 = be.to_dev(rm), be.to_dev(rv), be.empty((2, Oc)), be.empty(2 * Oc)
    be.call("mn_bn_acc_prep", N_, Oc, HW, be.ptr(dGa), be.ptr(dBe), 1e-5, 0.1, be.ptr(rm3), be.ptr(rv3), be.ptr(s3), act, be.ptr(stats), be.ptr(accmm), rows,
            be.ptr(dqp), be.ptr(dWs), 1, be.ptr(dCB) if bias else None, be.ptr(mm3), be.stream)
    assert np.array_equal(be.to_host(s3), be.to_host(s2)) and np.array_equal(be.to_host(rm3), be.to_host(rm2)) and np.array_equal(be.to_host(rv3), be.to_host(rv2)), \
        ("prep statistics", act)
    m3 = be.to_host(mm3)
    assert np.array_equal(m3[:Oc], o2.min(axis=(0, 2, 3))) and np.array_equal(m3[Oc:], o2.max(axis=(0, 2, 3))), ("per-channel extrema of the activation", act)
    a3 = be.empty(yv.shape)
    be.call("mn_bn_apply", be.ptr(dY), N_, Oc, HW, be.ptr(dGa), be.ptr(dBe), be.ptr(s3), act, be.ptr(a3), be.stream)
    assert np.array_equal(be.to_host(a3), o2), ("mn_bn_apply", act)
    if HW % 8 or Oc % 64 or Wo % 4:
        return
    # the consumer: a dense 3 x 3 IAO conv on that activation, its quantizer's range = the activation's own (and half of it: the clip conditions decide)
    w2_shape = (64, Oc, 3, 3)
    w2, wkw2, wscale2 = make_coded_weights(r, w2_shape, 3, w_bits)
    dW2, dWs2 = be.to_dev(w2), be.to_dev(wscale2)
    wq2 = be.wq(scale=dWs2, **wkw2)
    g2 = be.geom(yv.shape, w2_shape, 1, 1)
    for shrink in (1.0, 0.5):
        mn, mx = F(o2.min() * shrink), F(o2.max() * shrink)
        sc, zp = O.iao_qparams(mn.reshape(1), mx.reshape(1), a_bits, 0, True)
        hi = max(abs(mn / sc[0]), abs(mx / sc[0]))
        dqp2 = be.to_dev(np.array([sc[0], zp[0], -hi, hi], dtype=F))
        aq2 = be.actq(2, a_bits, 0, dqp2)
        nc = int(be.lib.mn_conv2d_iao_codes_bytes(C.byref(g2), C.byref(aq2), C.byref(wq2)))
        if nc <= 0:
            return
        c_ref, m_ref = be.empty_i8((nc,)), be.empty_i8((nc // 8,))
        aq2.codes, aq2.ste_mask = be.ptr(c_ref).value, be.ptr(m_ref).value
        y_ref = be.to_host(be.conv_fwd(g2, aq2, be.to_dev(o2), dW2, None, 3, wq=wq2))
        c_new, m_new = be.empty_i8((nc,)), be.empty_i8((nc // 8,))
        be.call("mn_bn_apply_codes", be.ptr(dY), N_, Oc, HW, be.ptr(dGa), be.ptr(dBe), be.ptr(s3), act, be.ptr(dqp2), a_bits, be.ptr(c_new), be.ptr(m_new), be.stream)
        n = o2.size
        assert np.array_equal(be.to_host(c_new)[:n], be.to_host(c_ref)[:n]), ("codes", act, shrink)
        assert np.array_equal(be.to_host(m_new)[:n // 8], be.to_host(m_ref)[:n // 8]), ("clip-STE bits", act, shrink)
        aq3 = be.actq(2, a_bits, 0, dqp2, flags=2)          # MN_ACTQ_CODES_GIVEN
        aq3.codes, aq3.ste_mask = be.ptr(c_new).value, be.ptr(m_new).value
        y_new = be.to_host(be.conv_fwd(g2, aq3, be.to_dev(np.full(o2.shape, np.nan, dtype=F)), dW2, None, 3, wq=wq2))
        assert np.array_equal(y_new, y_ref), ("conv on handed-over codes", act, shrink)
        if shrink < 1.0 and act == 2:
            bits = np.unpackbits(be.to_host(m_new)[:n // 8].view(np.uint8))
            assert 0.002 * n < np.count_nonzero(bits == 0) < 0.9 * n          # (the clip is exercised)


def check_tail(be, shape=(37, 10, 8, 8), seed=0):
    """The tail of the reference's nets (models/nin_gc.py:136-147; wqaq/dorefa/main.py:87-92): mn_bnrelu_gap_fwd / _bwd == BatchNorm2d (training) -> ReLU ->
    AvgPool2d over the map, and mn_cross_entropy_fwd / mn_scale_by == nn.CrossEntropyLoss() with its backward, against fp64 evaluations."""
    r = np.random.default_rng(seed)
    N_, Cc, H, W = shape
    HW = H * W
    y = (r.standard_normal(shape) * 1.7 + 0.4).astype(F)
    gamma, beta = (r.standard_normal(Cc) * 0.5 + 1).astype(F), (r.standard_normal(Cc) * 0.3).astype(F)
    rm, rv = (r.standard_normal(Cc) * 0.1).astype(F), (np.abs(r.standard_normal(Cc)) + 0.5).astype(F)
    dY, dGa, dBe, dRm, dRv = be.to_dev(y), be.to_dev(gamma), be.to_dev(beta), be.to_dev(rm), be.to_dev(rv)
    save, pooled = be.empty((2, Cc)), be.empty((N_, Cc))
    be.call("mn_bnrelu_gap_fwd", be.ptr(dY), N_, Cc, HW, be.ptr(dGa), be.ptr(dBe), 1e-5, 0.1, be.ptr(dRm), be.ptr(dRv), be.ptr(save), be.ptr(pooled), be.stream)
    y64 = y.astype(np.float64)
    mean, var = y64.mean(axis=(0, 2, 3)), y64.var(axis=(0, 2, 3))
    invstd = 1.0 / np.sqrt(var + 1e-5)
    zh = (y64 - mean.reshape(1, -1, 1, 1)) * invstd.reshape(1, -1, 1, 1)
    z = zh * gamma.astype(np.float64).reshape(1, -1, 1, 1) + beta.astype(np.float64).reshape(1, -1, 1, 1)
    a = np.maximum(z, 0.0)
    assert close(be.to_host(pooled), a.mean(axis=(2, 3)), 2e-6), "pooled"
    sv = be.to_host(save)
    assert close(sv[0], mean, 2e-6) and close(sv[1], invstd, 5e-6), "save"
    n = N_ * HW
    assert close(be.to_host(dRm), 0.9 * rm + 0.1 * mean, 2e-6) and close(be.to_host(dRv), 0.9 * rv + 0.1 * var * n / (n - 1), 5e-6), "running statistics"
    gp = r.standard_normal((N_, Cc)).astype(F)
    dy, dga, dbe = be.empty(shape), be.empty(Cc), be.empty(Cc)
    be.call("mn_bnrelu_gap_bwd", be.ptr(be.to_dev(gp)), be.ptr(dY), be.ptr(save), be.ptr(dGa), be.ptr(dBe), N_, Cc, HW, be.ptr(dy), be.ptr(dga), be.ptr(dbe), be.stream)
    dz = np.where(z > 0, gp.astype(np.float64).reshape(N_, Cc, 1, 1) / HW, 0.0)
    s1, s2 = dz.sum(axis=(0, 2, 3)), (dz * zh).sum(axis=(0, 2, 3))
    dy_ref = (gamma.astype(np.float64) * invstd).reshape(1, -1, 1, 1) * (dz - s1.reshape(1, -1, 1, 1) / n - zh * s2.reshape(1, -1, 1, 1) / n)
    assert close(be.to_host(dy), dy_ref, 1e-5) and close(be.to_host(dga), s2, 1e-5) and close(be.to_host(dbe), s1, 1e-5), "backward"
    # ---- the loss
    K_ = Cc
    x = (r.standard_normal((N_, K_)) * 3).astype(F)
    t = r.integers(0, K_, N_).astype(np.int64)
    t[3] = -100          # ignore_index
    loss, dl = be.empty(1), be.empty((N_, K_))
    dT = be.to_dev(t.view(F).reshape(-1)) if be.kind == "emu" else be.torch.from_numpy(t).cuda()
    be.call("mn_cross_entropy_fwd", be.ptr(be.to_dev(x)), be.ptr(dT), N_, K_, -100, be.ptr(loss), be.ptr(dl), be.stream)
    x64 = x.astype(np.float64)
    lse = np.log(np.exp(x64 - x64.max(axis=1, keepdims=True)).sum(axis=1)) + x64.max(axis=1)
    valid = t >= 0
    li = np.where(valid, lse - x64[np.arange(N_), np.where(valid, t, 0)], 0.0)
    assert abs(float(be.to_host(loss)[0]) - li.sum() / valid.sum()) <= 2e-6 * abs(li.sum() / valid.sum()), "loss"
    sm = np.exp(x64 - lse.reshape(-1, 1))
    oh = np.zeros_like(sm); oh[np.arange(N_)[valid], t[valid]] = 1.0
    dref = np.where(valid.reshape(-1, 1), (sm - oh) / valid.sum(), 0.0)
    assert close(be.to_host(dl), dref, 2e-6), "d logits"
    out = be.empty((N_, K_))
    be.call("mn_scale_by", be.ptr(dl), be.ptr(be.to_dev(np.array([0.5], dtype=F))), be.ptr(out), N_ * K_, be.stream)
    assert np.array_equal(be.to_host(out), be.to_host(dl) * F(0.5))


def check_iao_qadd_bn(be, shape=(3, 6, 4, 8), bits=4, q_type=0, relu=True, scbn=False, shrink=0.6, seed=0):
    """The END of an IAO residual block in one pass (round 6): mn_iao_qadd_bn_fwd == mn_bn_apply (per side) -> mn_iao_qadd_fwd_mm, and mn_iao_qadd_bn_bwd ==
    mn_iao_qadd_bwd -> mn_bn2d_bwd (per side), bit for bit -- output, (min, max) of the output, dy / dgamma / dbeta of both BatchNorms, the identity shortcut's
    gradient -- with a quantizer range that clips a good part of both inputs (both clip-STE conditions and the ReLU mask decide)."""
    r = np.random.default_rng(seed)
    N_, Cc, H, W = shape
    HW, n = H * W, int(np.prod(shape))
    y_r, x_s = (r.standard_normal(shape) * 2 + 0.3).astype(F), (r.standard_normal(shape) * 1.5 - 0.2).astype(F)
    g = r.standard_normal(shape).astype(F)
    mk = lambda: (np.stack([r.standard_normal(Cc) * 0.3, 1.0 / np.sqrt(np.abs(r.standard_normal(Cc)) + 0.5)]).astype(F), (r.standard_normal(Cc) * 0.5 + 1).astype(F),
                  (r.standard_normal(Cc) * 0.3).astype(F))
    (sv_r, ga_r, be_r), (sv_s, ga_s, be_s) = mk(), mk()
    dYr, dXs, dG = be.to_dev(y_r), be.to_dev(x_s), be.to_dev(g)
    dSr, dGr, dBr, dSs, dGs, dBs = (be.to_dev(v) for v in (sv_r, ga_r, be_r, sv_s, ga_s, be_s))
    # ---- unfused: the BatchNorm outputs exist
    a_r = be.empty(shape)
    be.call("mn_bn_apply", be.ptr(dYr), N_, Cc, HW, be.ptr(dGr), be.ptr(dBr), be.ptr(dSr), 2, be.ptr(a_r), be.stream)
    if scbn:
        a_s = be.empty(shape)
        be.call("mn_bn_apply", be.ptr(dXs), N_, Cc, HW, be.ptr(dGs), be.ptr(dBs), be.ptr(dSs), 2, be.ptr(a_s), be.stream)
    else:
        a_s = dXs
    av, bv = be.to_host(a_r), be.to_host(a_s)
    mn, mx = F(min(av.min(), bv.min()) * shrink), F(max(av.max(), bv.max()) * shrink)
    sc, zp = O.iao_qparams(mn.reshape(1), mx.reshape(1), bits, q_type, True)
    lo, hi = mn / sc[0] - zp[0], mx / sc[0] - zp[0]
    if q_type == 0:
        hi = max(abs(lo), abs(hi)); lo = -hi
    dqp = be.to_dev(np.array([sc[0], zp[0], lo, hi], dtype=F))
    cnt1 = int(be.lib.mn_iao_qadd_mm_count(n))
    o1, mm1 = be.empty(shape), be.empty(2 * cnt1)
    be.call("mn_iao_qadd_fwd_mm", be.ptr(a_r), be.ptr(a_s), be.ptr(o1), n, be.ptr(dqp), bits, q_type, int(relu), be.ptr(mm1), be.stream)
    da1, db1 = be.empty(shape), be.empty(shape)
    be.call("mn_iao_qadd_bwd", be.ptr(dG), be.ptr(a_r), be.ptr(a_s), be.ptr(da1), be.ptr(db1), n, be.ptr(dqp), bits, q_type, int(relu), be.stream)
    ws = be.empty(int(be.lib.mn_bnsign_ws_floats(Cc)) + 8)
    dy1, dga1, dbe1 = be.empty(shape), be.empty(Cc), be.empty(Cc)
    be.call("mn_bn2d_bwd", be.ptr(da1), be.ptr(dYr), be.ptr(dSr), be.ptr(dGr), be.ptr(dBr), N_, Cc, HW, 1, be.ptr(dy1), be.ptr(dga1), be.ptr(dbe1), be.ptr(ws), be.stream)
    if scbn:
        dys1, dgas1, dbes1 = be.empty(shape), be.empty(Cc), be.empty(Cc)
        be.call("mn_bn2d_bwd", be.ptr(db1), be.ptr(dXs), be.ptr(dSs), be.ptr(dGs), be.ptr(dBs), N_, Cc, HW, 1, be.ptr(dys1), be.ptr(dgas1), be.ptr(dbes1), be.ptr(ws), be.stream)
    # ---- fused
    cnt2 = int(be.lib.mn_bnrelu_mm_count(N_, Cc, HW))
    o2, mm2 = be.empty(shape), be.empty(2 * cnt2)
    bits_r, bits_s = be.empty_i8((n // 8,)), be.empty_i8((n // 8,))
    be.call("mn_iao_qadd_bn_fwd", be.ptr(dYr), be.ptr(dSr), be.ptr(dGr), be.ptr(dBr), be.ptr(dXs), be.ptr(dSs) if scbn else None, be.ptr(dGs) if scbn else None,
            be.ptr(dBs) if scbn else None, N_, Cc, HW, be.ptr(dqp), bits, q_type, int(relu), be.ptr(o2), be.ptr(mm2), be.ptr(bits_r), be.ptr(bits_s), be.stream)
    ov1, ov2 = be.to_host(o1), be.to_host(o2)
    assert np.array_equal(ov1, ov2), "output"
    m1, m2 = be.to_host(mm1), be.to_host(mm2)
    assert m1[:cnt1].min() == m2[:cnt2].min() == ov2.min() and m1[cnt1:].max() == m2[cnt2:].max() == ov2.max(), "min / max partials of the output"
    dy2, dga2, dbe2, dsc2 = be.empty(shape), be.empty(Cc), be.empty(Cc), be.empty(shape)
    be.call("mn_iao_qadd_bn_bwd", be.ptr(dG), be.ptr(dYr), be.ptr(dSr), be.ptr(dGr), be.ptr(dBr), N_, Cc, HW, be.ptr(dqp), be.ptr(bits_r), None if scbn else be.ptr(bits_s),
            be.ptr(dy2), None if scbn else be.ptr(dsc2), be.ptr(dga2), be.ptr(dbe2), be.ptr(ws), be.stream)
    assert np.array_equal(be.to_host(dy2), be.to_host(dy1)) and np.array_equal(be.to_host(dga2), be.to_host(dga1)) and np.array_equal(be.to_host(dbe2), be.to_host(dbe1)), "res side"
    if scbn:
        dys2, dgas2, dbes2 = be.empty(shape), be.empty(Cc), be.empty(Cc)
        be.call("mn_iao_qadd_bn_bwd", be.ptr(dG), be.ptr(dXs), be.ptr(dSs), be.ptr(dGs), be.ptr(dBs), N_, Cc, HW, be.ptr(dqp), be.ptr(bits_s), None, be.ptr(dys2), None,
                be.ptr(dgas2), be.ptr(dbes2), be.ptr(ws), be.stream)
        assert np.array_equal(be.to_host(dys2), be.to_host(dys1)) and np.array_equal(be.to_host(dgas2), be.to_host(dgas1)) and np.array_equal(be.to_host(dbes2), be.to_host(dbes1)), \
            "shortcut side"
    else:
        assert np.array_equal(be.to_host(dsc2), be.to_host(db1)), "identity shortcut's gradient"
    d = be.to_host(da1)
    assert (0.02 if relu else 0.0005) * n < np.count_nonzero(d == 0) < 0.95 * n          # (masks are exercised)


def check_iao_qadd(be, n=4096 + 8, bits=8, q_type=0, obs_kind=1, first=(True, False), update=True, seed=0, relu=False):
    """mn_iao_qadd_observe / _fwd / _bwd (QuantAdd, wqaq/iao/quantize.py:1484-1498, in three launches) == the separate entry points it replaces
    (mn_iao_observe x 2, mn_iao_union_range, mn_iao_qparams, mn_iao_fq_fwd x 2 + add, mn_iao_fq_bwd x 2), bit for bit: outputs, gradients, every buffer."""
    r = np.random.default_rng(seed)
    a, b = (r.standard_normal(n) * 3).astype(F), (r.standard_normal(n) * 2 + 0.5).astype(F)
    g = r.standard_normal(n).astype(F)
    st0 = dict(min_a=F(-1.5), max_a=F(2.5), min_b=F(-0.7), max_b=F(3.1), min_o=F(-9), max_o=F(9), scale=F(0.031), zp=F(3.0 if q_type else 0.0))
    mom = 0.1

    def fresh():
        return {k: be.to_dev(np.array([v], dtype=F)) for k, v in st0.items()}
    dA, dB, dG = be.to_dev(a), be.to_dev(b), be.to_dev(g)
    # ---- reference sequence
    s1 = fresh()
    ws1 = be.empty(int(be.lib.mn_iao_observe_ws_floats(1, n)))
    be.call("mn_iao_observe", be.ptr(dA), 1, n, obs_kind, int(first[0]), mom, be.ptr(s1["min_a"]), be.ptr(s1["max_a"]), be.ptr(ws1), be.stream)
    be.call("mn_iao_observe", be.ptr(dB), 1, n, obs_kind, int(first[1]), mom, be.ptr(s1["min_b"]), be.ptr(s1["max_b"]), be.ptr(ws1), be.stream)
    be.call("mn_iao_union_range", be.ptr(s1["min_a"]), be.ptr(s1["max_a"]), be.ptr(s1["min_b"]), be.ptr(s1["max_b"]), be.ptr(s1["min_o"]), be.ptr(s1["max_o"]), be.stream)
    qp1 = be.empty((1, 4))
    be.call("mn_iao_qparams", be.ptr(s1["min_o"]), be.ptr(s1["max_o"]), 1, bits, q_type, 1, int(update), be.ptr(s1["scale"]), be.ptr(s1["zp"]), be.ptr(qp1), be.stream)
    ya, yb = be.empty(n), be.empty(n)
    be.call("mn_iao_fq_fwd", be.ptr(dA), be.ptr(ya), 1, n, be.ptr(qp1), bits, q_type, 1, be.stream)
    be.call("mn_iao_fq_fwd", be.ptr(dB), be.ptr(yb), 1, n, be.ptr(qp1), bits, q_type, 1, be.stream)
    y_ref = (be.to_host(ya) + be.to_host(yb)).astype(F)
    dG_eff = dG
    if relu:            # the block's ReLU on the sum: forward max(s, 0), backward g * [s > 0]
        dG_eff = be.to_dev(np.where(y_ref > 0, g, F(0)).astype(F))
        y_ref = np.maximum(y_ref, F(0))
    da1, db1 = be.empty(n), be.empty(n)
    be.call("mn_iao_fq_bwd", be.ptr(dG_eff), be.ptr(dA), be.ptr(da1), 1, n, be.ptr(qp1), bits, q_type, 1, be.stream)
    be.call("mn_iao_fq_bwd", be.ptr(dG_eff), be.ptr(dB), be.ptr(db1), 1, n, be.ptr(qp1), bits, q_type, 1, be.stream)
    # ---- fused
    s2 = fresh()
    ws2 = be.empty(int(be.lib.mn_iao_qadd_ws_floats()))
    qp2 = be.empty((1, 4))
    be.call("mn_iao_qadd_observe", be.ptr(dA), be.ptr(dB), n, obs_kind, int(first[0]), int(first[1]), mom, be.ptr(s2["min_a"]), be.ptr(s2["max_a"]), be.ptr(s2["min_b"]),
            be.ptr(s2["max_b"]), be.ptr(s2["min_o"]), be.ptr(s2["max_o"]), bits, q_type, int(update), be.ptr(s2["scale"]), be.ptr(s2["zp"]), be.ptr(qp2), be.ptr(ws2), be.stream)
    y2, da2, db2 = be.empty(n), be.empty(n), be.empty(n)
    be.call("mn_iao_qadd_fwd", be.ptr(dA), be.ptr(dB), be.ptr(y2), n, be.ptr(qp2), bits, q_type, int(relu), be.stream)
    be.call("mn_iao_qadd_bwd", be.ptr(dG), be.ptr(dA), be.ptr(dB), be.ptr(da2), be.ptr(db2), n, be.ptr(qp2), bits, q_type, int(relu), be.stream)
    # + per-block (min, max) of the output for the observers of the layers that read it
    cnt = int(be.lib.mn_iao_qadd_mm_count(n))
    y3, mm = be.empty(n), be.empty(2 * cnt)
    be.call("mn_iao_qadd_fwd_mm", be.ptr(dA), be.ptr(dB), be.ptr(y3), n, be.ptr(qp2), bits, q_type, int(relu), be.ptr(mm), be.stream)
    assert np.array_equal(be.to_host(y3), be.to_host(y2))
    m1, M1, m2, M2 = be.to_dev(np.array([-0.25], dtype=F)), be.to_dev(np.array([0.75], dtype=F)), be.to_dev(np.array([-0.25], dtype=F)), be.to_dev(np.array([0.75], dtype=F))
    wso = be.empty(int(be.lib.mn_iao_observe_ws_floats(1, n)))
    be.call("mn_iao_observe", be.ptr(y3), 1, n, 1, 0, 0.1, be.ptr(m1), be.ptr(M1), be.ptr(wso), be.stream)
    be.call("mn_iao_observe_partials", be.ptr(mm), cnt, 1, 0, 0.1, be.ptr(m2), be.ptr(M2), be.stream)
    assert np.array_equal(be.to_host(m1), be.to_host(m2)) and np.array_equal(be.to_host(M1), be.to_host(M2))
    # ---- the same bookkeeping from producer partials: a as the output of mn_bn2d_fwd_mm (identity BatchNorm: eval statistics 0 / 1, gamma 1, beta 0, eps 0 -- the
    # kernel stores y itself) and b as (min, max) partials made by hand in another block structure
    if n % 64 == 0:
        Cc, HW = 4, 16
        Nn = n // (Cc * HW)
        cnt_a = int(be.lib.mn_bnrelu_mm_count(Nn, Cc, HW))
        assert cnt_a > 0
        one, zero = be.to_dev(np.ones(Cc, dtype=F)), be.to_dev(np.zeros(Cc, dtype=F))
        save, a_out, mm_a = be.empty(2 * Cc), be.empty(n), be.empty(2 * cnt_a)
        wsb = be.empty(int(be.lib.mn_bnsign_ws_floats(Cc)))
        be.call("mn_bn2d_fwd_mm", be.ptr(dA), Nn, Cc, HW, be.ptr(one), be.ptr(zero), 0.0, 0.1, 0, be.ptr(zero), be.ptr(one), be.ptr(save), be.ptr(a_out), be.ptr(wsb),
                be.ptr(mm_a), be.stream)
        assert np.array_equal(be.to_host(a_out), a)
        nbk = 1024 if n % 1024 == 0 and n >= 2048 else 8          # (more than 768 partials: the unrolled pass of k_qadd_final_p runs too)
        bb = b.reshape(nbk, -1)
        mm_b = be.to_dev(np.concatenate([bb.min(axis=1), bb.max(axis=1)]).astype(F))
        s3 = fresh()
        qp3 = be.empty((1, 4))
        be.call("mn_iao_qadd_observe_partials", be.ptr(mm_a), cnt_a, be.ptr(mm_b), nbk, obs_kind, int(first[0]), int(first[1]), mom, be.ptr(s3["min_a"]),
                be.ptr(s3["max_a"]), be.ptr(s3["min_b"]), be.ptr(s3["max_b"]), be.ptr(s3["min_o"]), be.ptr(s3["max_o"]), bits, q_type, int(update), be.ptr(s3["scale"]),
                be.ptr(s3["zp"]), be.ptr(qp3), be.stream)
        for k in st0:
            assert np.array_equal(be.to_host(s1[k]), be.to_host(s3[k])), ("partials", k)
        assert np.array_equal(be.to_host(qp1), be.to_host(qp3)), "qp from partials"
    for k in st0:
        assert np.array_equal(be.to_host(s1[k]), be.to_host(s2[k])), k
    assert np.array_equal(be.to_host(qp1), be.to_host(qp2)), "qp"
    assert np.array_equal(be.to_host(y2), y_ref), "out"
    assert np.array_equal(be.to_host(da1), be.to_host(da2)) and np.array_equal(be.to_host(db1), be.to_host(db2)), "gradients"


def check_iao_w_multi(be, bits=4, q_type=0, obs_kind=0, seed=0):
    """mn_iao_w_fwd_multi / _bwd_multi (per-channel IAO weight quantizers of several layers in one launch) == mn_iao_observe + mn_iao_qparams + mn_iao_fq_fwd /
    _bwd per tensor, bit for bit: quantised weights, gradients, observer / scale / zero_point buffers, the qp snapshots."""
    r = np.random.default_rng(seed)
    shapes = [(24, 16, 3, 3), (8, 4, 1, 1), (10, 33), (5, 700)]
    first = [True, False, False, True]
    ws = [(r.standard_normal(s_) * 0.3).astype(F) for s_ in shapes]
    gs = [r.standard_normal(s_).astype(F) for s_ in shapes]
    n = len(shapes)

    def fresh():
        out = []
        for s_ in shapes:
            O_ = s_[0]
            out.append(dict(mn=be.to_dev((r2.standard_normal(O_) * 0.1 - 0.5).astype(F)), mx=be.to_dev((r2.standard_normal(O_) * 0.1 + 0.5).astype(F)),
                            sc=be.to_dev(np.full(O_, 0.01, dtype=F)), zp=be.to_dev(np.zeros(O_, dtype=F)), qp=be.empty((O_, 4))))
        return out
    r2 = np.random.default_rng(seed + 1); st1 = fresh()
    r2 = np.random.default_rng(seed + 1); st2 = fresh()
    dW, dG = [be.to_dev(w) for w in ws], [be.to_dev(g) for g in gs]
    q1, d1 = [be.empty(s_) for s_ in shapes], [be.empty(s_) for s_ in shapes]
    for i, s_ in enumerate(shapes):
        rows, cols = s_[0], int(np.prod(s_[1:]))
        be.call("mn_iao_observe", be.ptr(dW[i]), rows, cols, obs_kind, int(first[i]), 0.1, be.ptr(st1[i]["mn"]), be.ptr(st1[i]["mx"]), None, be.stream)
        be.call("mn_iao_qparams", be.ptr(st1[i]["mn"]), be.ptr(st1[i]["mx"]), rows, bits, q_type, 0, 1, be.ptr(st1[i]["sc"]), be.ptr(st1[i]["zp"]), be.ptr(st1[i]["qp"]), be.stream)
        be.call("mn_iao_fq_fwd", be.ptr(dW[i]), be.ptr(q1[i]), rows, cols, be.ptr(st1[i]["qp"]), bits, q_type, 0, be.stream)
        be.call("mn_iao_fq_bwd", be.ptr(dG[i]), be.ptr(dW[i]), be.ptr(d1[i]), rows, cols, be.ptr(st1[i]["qp"]), bits, q_type, 0, be.stream)
    q2, d2 = [be.empty(s_) for s_ in shapes], [be.empty(s_) for s_ in shapes]
    PA, LA, IA = C.c_void_p * n, C.c_int64 * n, C.c_int32 * n
    pa = lambda arrs: PA(*[be.ptr(a).value for a in arrs])
    rows_a, cols_a = LA(*[s_[0] for s_ in shapes]), LA(*[int(np.prod(s_[1:])) for s_ in shapes])
    be.call("mn_iao_w_fwd_multi", pa(dW), pa(q2), pa([t["mn"] for t in st2]), pa([t["mx"] for t in st2]), pa([t["sc"] for t in st2]), pa([t["zp"] for t in st2]),
            pa([t["qp"] for t in st2]), rows_a, cols_a, IA(*[int(f) for f in first]), n, obs_kind, 0.1, bits, q_type, be.stream)
    be.call("mn_iao_w_bwd_multi", pa(dG), pa(dW), pa(d2), pa([t["qp"] for t in st2]), rows_a, cols_a, n, bits, q_type, be.stream)
    for i in range(n):
        for k in ("mn", "mx", "sc", "zp", "qp"):
            assert np.array_equal(be.to_host(st1[i][k]), be.to_host(st2[i][k])), (i, k)
        assert np.array_equal(be.to_host(q1[i]), be.to_host(q2[i])), ("qw", i)
        assert np.array_equal(be.to_host(d1[i]), be.to_host(d2[i])), ("dw", i)


def check_gap(be, planes=37, HW=64, seed=0):
    """mn_avgpool_global_fwd / _bwd (nn.AvgPool2d over the whole image) vs numpy."""
    r = np.random.default_rng(seed)
    x = r.standard_normal((planes, HW)).astype(F)
    gy = r.standard_normal(planes).astype(F)
    y, dx = be.empty(planes), be.empty((planes, HW))
    dX, dG = be.to_dev(x), be.to_dev(gy)
    be.call("mn_avgpool_global_fwd", be.ptr(dX), planes, HW, be.ptr(y), be.stream)
    be.call("mn_avgpool_global_bwd", be.ptr(dG), planes, HW, be.ptr(dx), be.stream)
    assert close(be.to_host(y), x.astype(np.float64).mean(axis=1), 1e-6)
    assert np.array_equal(be.to_host(dx), np.repeat((gy / F(HW)).astype(F)[:, None], HW, axis=1))


def check_qd_pack_multi(be, w_bits=2, iao=False, seed=0):
    """mn_qd_pack_multi over several tensors (mixed shapes, 3 x 3 and 1 x 1) == the same call per tensor: the table / tile bookkeeping of the one-launch pack."""
    r = np.random.default_rng(seed)
    shapes = [(64, 64, 3, 3), (128, 64, 1, 1), (64, 128, 3, 3), (128, 128, 3, 3)]
    n = len(shapes)
    nw = 2 ** w_bits - 1
    ws, scs = [], []
    for s_ in shapes:
        if iao:
            qmax = 2 ** (w_bits - 1) - 1
            sc = (np.abs(r.standard_normal(s_[0])) * 0.01 + 0.002).astype(F)
            ws.append((r.integers(-qmax, qmax + 1, size=s_).astype(F) * sc.reshape(-1, 1, 1, 1)).astype(F)); scs.append(sc)
        else:
            k = r.integers(0, nw + 1, size=s_).astype(F)
            ws.append((F(2) * (k * F(1.0 / nw)) - F(1)).astype(F))
    dW = [be.to_dev(w) for w in ws]
    dS = [be.to_dev(sc) for sc in scs]
    nbytes = [s_[0] * s_[1] * s_[2] * s_[3] * 2 for s_ in shapes]
    PA1, LA1, IA1 = C.c_void_p * 1, C.c_int64 * 1, C.c_int32 * 1
    ref = []
    for i, s_ in enumerate(shapes):
        f, b = be.empty_i8((nbytes[i],)), be.empty_i8((nbytes[i],))
        be.call("mn_qd_pack_multi", PA1(be.ptr(dW[i]).value), PA1(be.ptr(f).value), PA1(be.ptr(b).value), LA1(s_[0]), LA1(s_[1]), LA1(s_[2] * s_[3]),
                PA1(be.ptr(dS[i]).value) if iao else None, IA1(1) if iao else None, 1, w_bits, be.stream)
        ref.append((be.to_host(f), be.to_host(b)))
    PA, LA, IA = C.c_void_p * n, C.c_int64 * n, C.c_int32 * n
    fs, bs = [be.empty_i8((nb,)) for nb in nbytes], [be.empty_i8((nb,)) for nb in nbytes]
    be.call("mn_qd_pack_multi", PA(*[be.ptr(t).value for t in dW]), PA(*[be.ptr(t).value for t in fs]), PA(*[be.ptr(t).value for t in bs]),
            LA(*[s_[0] for s_ in shapes]), LA(*[s_[1] for s_ in shapes]), LA(*[s_[2] * s_[3] for s_ in shapes]),
            PA(*[be.ptr(t).value for t in dS]) if iao else None, IA(*[1] * n) if iao else None, n, w_bits, be.stream)
    fwd_bytes = lambda i: nbytes[i] // 2 if (w_bits <= 7 or iao) else nbytes[i]          # the int8 forward image is half the size
    for i in range(n):
        assert np.array_equal(be.to_host(fs[i])[:fwd_bytes(i)], ref[i][0][:fwd_bytes(i)]), ("forward image", i)
        assert np.array_equal(be.to_host(bs[i]), ref[i][1]), ("backward-data image", i)
