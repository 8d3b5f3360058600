"""torch.autograd bindings of the libmicronet_hip C ABI -- the only compute path of the product.

PyTorch supplies device memory, the current HIP stream and the autograd tape; every forward/backward below is one or
a few launches of hand-written gfx950 kernels (``micronet_amd/csrc``).  There is NO fallback: a CPU tensor or a
missing library raises ``MicronetHipError``.

Reference call sites each op replaces (micronet/compression/quantization/...):
  round_half_away      wqaq/dorefa/quantize.py:11-21 (Round)
  dorefa_act           wqaq/dorefa/quantize.py:36-46
  dorefa_weight        wqaq/dorefa/quantize.py:61-73
  binary_act           wbwtab/quantize.py:11-36
  ternary_weight       wbwtab/quantize.py:55-75 + 132-146
  binary_weight        wbwtab/quantize.py:40-51 + 98-102 + 121-130
  iao_*                wqaq/iao/quantize.py:15-113 (observers), 214-240 (fake-quant), 293-321 (qparams)
  qconv2d / qlinear    F.conv2d / F.linear call sites (dorefa 113-121/198, wbwtab 186-194, iao 498-506/843-851/947-993/1156)
  bn_batch_stats       wqaq/iao/quantize.py:853-855
"""
import ctypes as C
import threading
import weakref

import torch
from torch.autograd import Function

from . import _lib
from ._lib import ActQ, ConvGeom, MicronetHipError, WQ
from .sign_tensor import LazyBNAct, LazyBNGrad, LazyConvOut, LazyPoolGrad, LazyQConvOut, LazyReluConvOut, LazyReluGrad, QActTensor, QGrad, SignTensor

ACTQ_NONE, ACTQ_DOREFA, ACTQ_IAO, ACTQ_SIGN8, ACTQ_CODE8 = _lib.MN_ACTQ_NONE, _lib.MN_ACTQ_DOREFA, _lib.MN_ACTQ_IAO, _lib.MN_ACTQ_SIGN8, _lib.MN_ACTQ_CODE8
WQ_REAL, WQ_TERNARY, WQ_DOREFA, WQ_IAO = _lib.MN_WQ_REAL, _lib.MN_WQ_TERNARY, _lib.MN_WQ_DOREFA, _lib.MN_WQ_IAO

# algorithm used by the conv entry points; tests flip it to compare kernels (0 auto, 1 direct VALU, 2 fp32-MFMA only, 3 code-domain bf16-MFMA only)
CONV_ALGO = _lib.MN_ALGO_AUTO


# optional per-launch timing (bench.py): an object with .arm(which, nbytes) -> token (arms mn_profile_next with two raw HIP
# events that the library records right around the main kernel) and .done(token, kernel_name)
PROFILER = None


class _Span:
    __slots__ = ("tok",)

    def __init__(self, tok):
        self.tok = tok

    def __enter__(self):
        return self

    def __exit__(self, *a):
        if self.tok is not None:
            PROFILER.done(self.tok, last_kernel())
        return False


_NOSPAN = _Span(None)


def _span(g, which, nbytes):
    if PROFILER is None:
        return _NOSPAN
    tok = PROFILER.arm(which, nbytes)
    return _Span(tok) if tok is not None else _NOSPAN


import os as _os0


# Stock-operator fall-throughs.  A few modules of this package hand geometries their gfx950 kernels do not cover to the stock torch operator (MIOpen / ATen) --
# the reference's own behaviour, so never wrong, but not the hot path this package exists for.  Every such fall-through is counted here by site name, so that a
# benchmark or a test can ASSERT that a model runs entirely on the library's kernels (bench.py reports ``stock_fallbacks``; 0 for every benched workload).
_FALLBACKS = {}


def note_fallback(site):
    _FALLBACKS[site] = _FALLBACKS.get(site, 0) + 1


def fallback_counts(reset=False):
    out = dict(_FALLBACKS)
    if reset:
        _FALLBACKS.clear()
    return out


def last_kernel():
    """Name of the main kernel the last conv call launched (for the profiler's per-kernel aggregation)."""
    k = _lib_().mn_last_kernel()
    return k.decode() if k else "?"


def _lib_():
    return _lib.get_lib()


def _chk(t, name="tensor"):
    if t is None:
        return None
    if isinstance(t, (LazyBNGrad, LazyPoolGrad, LazyConvOut, LazyQConvOut, LazyReluConvOut, LazyReluGrad, QActTensor, QGrad)):      # a lazy tensor reaching a kernel that wants plain memory
        t = t.materialize()
    elif isinstance(t, SignTensor):
        t = t.to_float()
    if not t.is_cuda:
        raise MicronetHipError("%s is on %s: micronet_amd runs on MI355X only (no CPU fallback)" % (name, t.device))
    if t.dtype != torch.float32:
        raise MicronetHipError("%s must be float32, got %s" % (name, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _call(name, *args):
    lib = _lib_()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        lib.check(rc, name)


# ------------------------------------------------------------------------------------------------ DoReFa
class RoundHalfAway(Function):
    @staticmethod
    def forward(ctx, v):
        v = _chk(v, "input")
        out = torch.empty_like(v)
        with torch.cuda.device_of(v):
            _call("mn_round_half_away", _p(v), _p(out), v.numel(), _s())
        return out

    @staticmethod
    def backward(ctx, g):
        return g.clone()


class DorefaAct(Function):
    @staticmethod
    def backward_raw(g, x, bits):
        """the clip-STE of the quantizer without autograd bookkeeping: dx = ((g*s)/s) * [0 <= 0.1x <= 1] * 0.1"""
        g, x = _chk(g, "grad"), _chk(x, "input")
        dx = torch.empty_like(x)
        with torch.cuda.device_of(x):
            _call("mn_dorefa_act_bwd", _p(g), _p(x), _p(dx), x.numel(), bits, _s())
        return dx

    @staticmethod
    def forward(ctx, x, bits):
        x = _chk(x, "input")
        y = torch.empty_like(x)
        with torch.cuda.device_of(x):
            _call("mn_dorefa_act_fwd", _p(x), _p(y), x.numel(), bits, _s())
        ctx.save_for_backward(x)
        ctx.bits = bits
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = _chk(g, "grad")
        dx = torch.empty_like(x)
        with torch.cuda.device_of(x):
            _call("mn_dorefa_act_bwd", _p(g), _p(x), _p(dx), x.numel(), ctx.bits, _s())
        return dx, None


class DorefaWeight(Function):
    @staticmethod
    def forward(ctx, w, bits):
        w = _chk(w, "weight")
        lib = _lib_()
        ws = torch.empty(int(lib.mn_dorefa_w_ws_floats(w.numel())), dtype=torch.float32, device=w.device)
        qw = torch.empty_like(w)
        with torch.cuda.device_of(w):
            _call("mn_dorefa_w_fwd", _p(w), _p(qw), w.numel(), bits, _p(ws), _s())
        ctx.save_for_backward(w)
        ctx.bits = bits
        return qw

    @staticmethod
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        g = _chk(g, "grad")
        lib = _lib_()
        ws = torch.empty(int(lib.mn_dorefa_w_ws_floats(w.numel())), dtype=torch.float32, device=w.device)
        dw = torch.empty_like(w)
        with torch.cuda.device_of(w):
            _call("mn_dorefa_w_bwd", _p(g), _p(w), _p(dw), w.numel(), ctx.bits, _p(ws), _s())
        return dw, None


class MultiDorefaWeight(Function):
    """DorefaWeight over several weight tensors in ONE launch per phase (mn_dorefa_w_fwd/bwd_multi_cached: 2 launches forward, 2 backward for the whole net
    instead of 5 per layer), with tanh(w) -- an fp64 evaluation, see quant_kernels.hip -- computed once per step and cached.  Same arithmetic per tensor:
    bit-identical to the per-layer calls."""

    @staticmethod
    def forward(ctx, bits, *ws):
        ws = [_chk(w, "weight") for w in ws]
        n = len(ws)
        lib = _lib_()
        qws = [torch.empty_like(w) for w in ws]
        scratch = [torch.empty(int(lib.mn_dorefa_w_ws_floats(w.numel())), dtype=torch.float32, device=w.device) for w in ws]
        th = [torch.empty(w.numel(), dtype=torch.float32, device=w.device) for w in ws]
        PA, LA = C.c_void_p * n, C.c_int64 * n
        with torch.cuda.device_of(ws[0]):
            _call("mn_dorefa_w_fwd_multi_cached", PA(*[w.data_ptr() for w in ws]), PA(*[q.data_ptr() for q in qws]), PA(*[t.data_ptr() for t in scratch]),
                  PA(*[t.data_ptr() for t in th]), LA(*[w.numel() for w in ws]), n, bits, _s())
        ctx.save_for_backward(*ws, *scratch, *th)
        ctx.bits, ctx.n = bits, n
        return tuple(qws)

    @staticmethod
    def backward(ctx, *gs):
        n, bits = ctx.n, ctx.bits
        ws, scratch, th = ctx.saved_tensors[:n], ctx.saved_tensors[n:2 * n], ctx.saved_tensors[2 * n:]
        idx = [i for i in range(n) if gs[i] is not None]
        dws = [None] * n
        if idx:
            g = [_chk(gs[i], "grad") for i in idx]
            out = [torch.empty_like(ws[i]) for i in idx]
            m = len(idx)
            PA, LA = C.c_void_p * m, C.c_int64 * m
            with torch.cuda.device_of(ws[0]):
                _call("mn_dorefa_w_bwd_multi_cached", PA(*[t.data_ptr() for t in g]), PA(*[ws[i].data_ptr() for i in idx]), PA(*[t.data_ptr() for t in out]),
                      PA(*[scratch[i].data_ptr() for i in idx]), PA(*[th[i].data_ptr() for i in idx]), LA(*[ws[i].numel() for i in idx]), m, bits, _s())
            for k, i in enumerate(idx):
                dws[i] = out[k]
        return (None,) + tuple(dws)


# ------------------------------------------------------------------------------------------------ WbWtAb
class BinaryAct(Function):
    @staticmethod
    def forward(ctx, x):
        x = _chk(x, "input")
        y = torch.empty_like(x)
        with torch.cuda.device_of(x):
            _call("mn_binact_fwd", _p(x), _p(y), x.numel(), _s())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = _chk(g, "grad")
        dx = torch.empty_like(x)
        with torch.cuda.device_of(x):
            _call("mn_binact_bwd", _p(g), _p(x), _p(dx), x.numel(), _s())
        return dx


class TernaryWeight(Function):
    """out = ternary(w) * alpha, backward = STE + the autograd path through alpha."""

    @staticmethod
    def forward(ctx, w):
        w = _chk(w, "weight")
        O, K = w.shape[0], w[0].numel()
        qw = torch.empty_like(w)
        stats = torch.empty((O, 4), dtype=torch.float32, device=w.device)
        with torch.cuda.device_of(w):
            _call("mn_ternary_w_fwd", _p(w), _p(qw), _p(stats), O, K, _s())
        ctx.save_for_backward(w, stats)
        return qw

    @staticmethod
    def backward(ctx, g):
        w, stats = ctx.saved_tensors
        g = _chk(g, "grad")
        dw = torch.empty_like(w)
        with torch.cuda.device_of(w):
            _call("mn_ternary_w_bwd", _p(g), _p(w), _p(stats), _p(dw), w.shape[0], w[0].numel(), _s())
        return dw


class MultiTernaryWeight(Function):
    """TernaryWeight over several weight tensors in ONE launch each way (mn_ternary_w_fwd_multi / _bwd_multi): a training step quantizes
    the weights of every conv; as separate autograd nodes those are 2 x (#layers) launches of ~5 us.  Same arithmetic per tensor."""

    @staticmethod
    def forward(ctx, *ws):
        ws = [_chk(w, "weight") for w in ws]
        n = len(ws)
        qws = [torch.empty_like(w) for w in ws]
        stats = [torch.empty((w.shape[0], 4), dtype=torch.float32, device=w.device) for w in ws]
        PA, LA = C.c_void_p * n, C.c_int64 * n
        Os, Ks = LA(*[w.shape[0] for w in ws]), LA(*[w[0].numel() for w in ws])
        with torch.cuda.device_of(ws[0]):
            _call("mn_ternary_w_fwd_multi", PA(*[w.data_ptr() for w in ws]), PA(*[q.data_ptr() for q in qws]), PA(*[t.data_ptr() for t in stats]),
                  Os, Ks, n, _s())
        ctx.save_for_backward(*ws, *stats)
        ctx.n = n
        return tuple(qws)

    @staticmethod
    def backward(ctx, *gs):
        n = ctx.n
        ws, stats = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        idx = [i for i in range(n) if gs[i] is not None]
        dws = [None] * n
        if idx:
            g = [_chk(gs[i], "grad") for i in idx]
            out = [torch.empty_like(ws[i]) for i in idx]
            m = len(idx)
            PA, LA = C.c_void_p * m, C.c_int64 * m
            with torch.cuda.device_of(ws[0]):
                _call("mn_ternary_w_bwd_multi", PA(*[t.data_ptr() for t in g]), PA(*[ws[i].data_ptr() for i in idx]), PA(*[stats[i].data_ptr() for i in idx]),
                      PA(*[t.data_ptr() for t in out]), LA(*[ws[i].shape[0] for i in idx]), LA(*[ws[i][0].numel() for i in idx]), m, _s())
            for k, i in enumerate(idx):
                dws[i] = out[k]
        return tuple(dws)


def ternary_stats(w):
    """(qw, stats[O,4] = alpha, thr, cnt, sum) without autograd."""
    w = _chk(w.detach(), "weight")
    qw = torch.empty_like(w)
    stats = torch.empty((w.shape[0], 4), dtype=torch.float32, device=w.device)
    with torch.cuda.device_of(w):
        _call("mn_ternary_w_fwd", _p(w), _p(qw), _p(stats), w.shape[0], w[0].numel(), _s())
    return qw, stats


class BinaryWeight(Function):
    """Mean-centre + clamp ``w`` IN PLACE (as the reference mutates weight.data), then sign(w) * mean|w|."""

    @staticmethod
    def forward(ctx, w):
        if not w.is_contiguous():
            raise MicronetHipError("binary weight quantizer mutates the weight in place and needs it contiguous")
        _chk(w, "weight")
        if w.dim() != 4:
            raise MicronetHipError("binary weight quantizer expects a 4-D conv weight")
        O, Cc, R = w.shape[0], w.shape[1], w.shape[2] * w.shape[3]
        qw = torch.empty_like(w)
        alpha = torch.empty(O, dtype=torch.float32, device=w.device)
        with torch.cuda.device_of(w):
            _call("mn_binary_w_fwd", _p(w), _p(qw), _p(alpha), O, Cc, R, _s())
        ctx.save_for_backward(w, alpha)
        return qw

    @staticmethod
    def backward(ctx, g):
        w, alpha = ctx.saved_tensors
        g = _chk(g, "grad")
        dw = torch.empty_like(w)
        with torch.cuda.device_of(w):
            _call("mn_binary_w_bwd", _p(g), _p(w), _p(alpha), _p(dw), w.shape[0], w[0].numel(), _s())
        return dw


class MultiBinaryWeight(Function):
    """BinaryWeight over several conv weights in ONE launch each way (mn_binary_w_fwd_multi / _bwd_multi): every tensor is mean-centred and clamped IN PLACE like the
    single-tensor op (the reference mutates weight.data, wbwtab/quantize.py:98-102, 123), then sign(w) * mean|w|.  Same arithmetic per tensor."""

    @staticmethod
    def forward(ctx, *ws):
        for w in ws:
            if not w.is_contiguous() or w.dim() != 4:
                raise MicronetHipError("binary weight quantizer mutates 4-D conv weights in place and needs them contiguous")
            _chk(w, "weight")
        n = len(ws)
        qws = [torch.empty_like(w) for w in ws]
        alphas = [torch.empty(w.shape[0], dtype=torch.float32, device=w.device) for w in ws]
        PA, LA = C.c_void_p * n, C.c_int64 * n
        with torch.cuda.device_of(ws[0]):
            _call("mn_binary_w_fwd_multi", PA(*[w.data_ptr() for w in ws]), PA(*[q.data_ptr() for q in qws]), PA(*[a.data_ptr() for a in alphas]),
                  LA(*[w.shape[0] for w in ws]), LA(*[w.shape[1] for w in ws]), LA(*[w.shape[2] * w.shape[3] for w in ws]), n, _s())
        ctx.save_for_backward(*ws, *alphas)
        ctx.n = n
        return tuple(qws)

    @staticmethod
    def backward(ctx, *gs):
        n = ctx.n
        ws, alphas = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        idx = [i for i in range(n) if gs[i] is not None]
        dws = [None] * n
        if idx:
            g = [_chk(gs[i], "grad") for i in idx]
            out = [torch.empty_like(ws[i]) for i in idx]
            m = len(idx)
            PA, LA = C.c_void_p * m, C.c_int64 * m
            with torch.cuda.device_of(ws[0]):
                _call("mn_binary_w_bwd_multi", PA(*[t.data_ptr() for t in g]), PA(*[ws[i].data_ptr() for i in idx]), PA(*[alphas[i].data_ptr() for i in idx]),
                      PA(*[t.data_ptr() for t in out]), LA(*[ws[i].shape[0] for i in idx]), LA(*[ws[i].shape[1] for i in idx]),
                      LA(*[ws[i].shape[2] * ws[i].shape[3] for i in idx]), m, _s())
            for k, i in enumerate(idx):
                dws[i] = out[k]
        return tuple(dws)


# ------------------------------------------------------------------------------------------------ IAO
def iao_observe(x, rows, obs_kind, first, momentum, min_val, max_val):
    x = _chk(x.detach(), "input")
    cols = x.numel() // rows
    lib = _lib_()
    nws = int(lib.mn_iao_observe_ws_floats(rows, cols))
    ws = torch.empty(max(nws, 1), dtype=torch.float32, device=x.device)
    with torch.cuda.device_of(x):
        _call("mn_iao_observe", _p(x), rows, cols, obs_kind, int(first), float(momentum), _p(min_val), _p(max_val), _p(ws), _s())


def iao_qparams(min_val, max_val, bits, q_type, is_act, update, scale, zero_point):
    """Returns the {scale, zp, lo, hi} snapshot ([rows, 4]) the kernels read; updates scale/zero_point if ``update``."""
    rows = min_val.numel()
    qp = torch.empty((rows, 4), dtype=torch.float32, device=min_val.device)
    with torch.cuda.device_of(min_val):
        _call("mn_iao_qparams", _p(min_val), _p(max_val), rows, bits, q_type, int(is_act), int(update), _p(scale),
              _p(zero_point), _p(qp), _s())
    return qp


def iao_union_range(a_min, a_max, b_min, b_max, out_min, out_max):
    with torch.cuda.device_of(a_min):
        _call("mn_iao_union_range", _p(a_min), _p(a_max), _p(b_min), _p(b_max), _p(out_min), _p(out_max), _s())


class MultiIaoWeight(Function):
    """The per-channel IAO weight quantizers of several layers in ONE launch (mn_iao_w_fwd_multi: observer update, qparams and fake-quant of every output
    channel) and one in backward, instead of four launches per layer.  ``state`` = per tensor (min_val, max_val, scale, zero_point, qp [rows, 4], first-call flag):
    the quantizers' own buffers, updated in place exactly as ``Quantizer.forward`` would.  Bit-identical to the per-layer path."""

    @staticmethod
    def forward(ctx, cfg, state, *ws):
        bits, q_type, obs_kind, momentum = cfg
        ws = [_chk(w, "weight") for w in ws]
        n = len(ws)
        qws = [torch.empty_like(w) for w in ws]
        rows = [w.shape[0] for w in ws]
        cols = [w.numel() // w.shape[0] for w in ws]
        PA, LA, IA = C.c_void_p * n, C.c_int64 * n, C.c_int32 * n
        with torch.cuda.device_of(ws[0]):
            _call("mn_iao_w_fwd_multi", PA(*[w.data_ptr() for w in ws]), PA(*[q.data_ptr() for q in qws]), PA(*[st[0].data_ptr() for st in state]),
                  PA(*[st[1].data_ptr() for st in state]), PA(*[st[2].data_ptr() for st in state]), PA(*[st[3].data_ptr() for st in state]),
                  PA(*[st[4].data_ptr() for st in state]), LA(*rows), LA(*cols), IA(*[int(st[5]) for st in state]), n, obs_kind, float(momentum), bits, q_type, _s())
        ctx.save_for_backward(*ws)
        ctx.qps = [st[4] for st in state]
        ctx.cfg = (bits, q_type, rows, cols)
        return tuple(qws)

    @staticmethod
    def backward(ctx, *gs):
        ws = ctx.saved_tensors
        bits, q_type, rows, cols = ctx.cfg
        idx = [i for i in range(len(ws)) if gs[i] is not None]
        dws = [None] * len(ws)
        if idx:
            g = [_chk(gs[i], "grad") for i in idx]
            out = [torch.empty_like(ws[i]) for i in idx]
            m = len(idx)
            PA, LA = C.c_void_p * m, C.c_int64 * m
            with torch.cuda.device_of(ws[0]):
                _call("mn_iao_w_bwd_multi", PA(*[t.data_ptr() for t in g]), PA(*[ws[i].data_ptr() for i in idx]), PA(*[t.data_ptr() for t in out]),
                      PA(*[ctx.qps[i].data_ptr() for i in idx]), LA(*[rows[i] for i in idx]), LA(*[cols[i] for i in idx]), m, bits, q_type, _s())
            for k, i in enumerate(idx):
                dws[i] = out[k]
        return (None, None) + tuple(dws)


def iao_qadd_observe(res, shortcut, obs_res, obs_sc, quantizer, update):
    """QuantAdd's bookkeeping in two launches (mn_iao_qadd_observe): both input observers, the union range into the shared quantizer's observer, its qparams.
    Returns the {scale, zero_point, lo, hi} snapshot."""
    lib = _lib_()
    res, shortcut = _chk(res.detach(), "res"), _chk(shortcut.detach(), "shortcut")
    obs = quantizer.observer
    ws = torch.empty(int(lib.mn_iao_qadd_ws_floats()), dtype=torch.float32, device=res.device)
    qp = torch.empty((1, 4), dtype=torch.float32, device=res.device)
    with torch.cuda.device_of(res):
        _call("mn_iao_qadd_observe", _p(res), _p(shortcut), res.numel(), obs_res._kind, int(obs_res.num_flag == 0), int(obs_sc.num_flag == 0),
              float(getattr(obs_res, "momentum", 0.1)), _p(obs_res.min_val), _p(obs_res.max_val), _p(obs_sc.min_val), _p(obs_sc.max_val), _p(obs.min_val), _p(obs.max_val),
              quantizer.bits, quantizer._q_type_static if update else quantizer.q_type, int(update), _p(quantizer.scale), _p(quantizer.zero_point), _p(qp), _p(ws), _s())
    return qp


def _valid_minmax(t):
    """(mm, count) the producer of ``t`` left on it (per-block minima / maxima of exactly this tensor), or None"""
    mm = getattr(t, "_mn_minmax", None)
    if mm is None or mm[2] != t._version or mm[0] is None or mm[1] <= 0:
        return None
    return mm[0], mm[1]


def iao_qadd_observe_partials(pr, ps, obs_res, obs_sc, quantizer, update):
    """``iao_qadd_observe`` from the producers' (min, max) partials: one launch, neither tensor is read."""
    obs = quantizer.observer
    dev = pr[0].device
    qp = torch.empty((1, 4), dtype=torch.float32, device=dev)
    with torch.cuda.device_of(pr[0]):
        _call("mn_iao_qadd_observe_partials", _p(pr[0]), pr[1], _p(ps[0]), ps[1], obs_res._kind, int(obs_res.num_flag == 0), int(obs_sc.num_flag == 0),
              float(getattr(obs_res, "momentum", 0.1)), _p(obs_res.min_val), _p(obs_res.max_val), _p(obs_sc.min_val), _p(obs_sc.max_val), _p(obs.min_val), _p(obs.max_val),
              quantizer.bits, quantizer._q_type_static if update else quantizer.q_type, int(update), _p(quantizer.scale), _p(quantizer.zero_point), _p(qp), _s())
    return qp


RES_ADD_FOLD = True          # the identity shortcut's gradient is added in the store of the conv's dx (round 5: -1.1 % step time against autograd's add kernel per block)


class ResidualToken:
    """Left on a tensor x by the dense IAO conv that reads it (QConv2d.forward, when its backward-data can add a tensor in its store: mn_actq.dx_add).  A QuantAdd whose
    IDENTITY shortcut is that same x -- and whose other input descends from that conv -- parks the shortcut's gradient here instead of returning it to autograd; the
    conv's backward-data, which autograd runs later, adds it while it stores dx: d loss / d x arrives in ONE tensor and the engine's accumulate kernel (18 us per
    residual block of the IAO resnet18 at batch 256) disappears."""
    __slots__ = ("node", "d_sc", "claimed", "consumed")

    def __init__(self):
        self.consumed = False          # set when the conv's backward-data has run: a gradient parked after that would be lost, so nobody parks any more
        self.node, self.d_sc, self.claimed = None, None, False          # node: a WEAK reference to the conv's autograd node (the node saved x, x carries this token:
        #                                                                  a strong one would close a cycle through C++ that a forward without backward never breaks)


def _descends_from(t, node, limit=96):
    """True when `node` is among the first `limit` autograd nodes above t (breadth first): t is computed from that node's output."""
    seen, frontier, n = set(), [t.grad_fn] if t.grad_fn is not None else [], 0
    while frontier and n < limit:
        nxt = []
        for f in frontier:
            if f is node:
                return True
            if f is None or f in seen:
                continue
            seen.add(f)          # (the node objects themselves: ids of short-lived wrappers of C++ nodes are reused)
            n += 1
            nxt.extend(fn for fn, _ in f.next_functions if fn is not None)
        frontier = nxt
    return False


class IaoQuantAdd(Function):
    """out = Q(res) + Q(shortcut) with one shared per-tensor quantizer (QuantAdd, wqaq/iao/quantize.py:1484-1498): one pass forward, one backward."""

    @staticmethod
    def forward(ctx, res, shortcut, qp, bits, q_type, relu=False, want_minmax=False, res_tok=None):
        res, shortcut = _chk(res, "res"), _chk(shortcut, "shortcut")
        out = torch.empty_like(res)
        with torch.cuda.device_of(res):
            if want_minmax:
                count = int(_lib_().mn_iao_qadd_mm_count(res.numel()))
                mm = torch.empty(2 * count, dtype=torch.float32, device=res.device)
                _call("mn_iao_qadd_fwd_mm", _p(res), _p(shortcut), _p(out), res.numel(), _p(qp), bits, q_type, int(relu), _p(mm), _s())
                _PENDING_MINMAX[0] = (mm, count)
            else:
                _call("mn_iao_qadd_fwd", _p(res), _p(shortcut), _p(out), res.numel(), _p(qp), bits, q_type, int(relu), _s())
        ctx.save_for_backward(res, shortcut, qp)
        ctx.cfg = (bits, q_type, int(relu))
        ctx.res_tok = res_tok
        return out

    @staticmethod
    def backward(ctx, g):
        res, shortcut, qp = ctx.saved_tensors
        bits, q_type, relu = ctx.cfg
        g = _chk(g, "grad")
        da, db = torch.empty_like(res), torch.empty_like(shortcut)
        with torch.cuda.device_of(res):
            _call("mn_iao_qadd_bwd", _p(g), _p(res), _p(shortcut), _p(da), _p(db), res.numel(), _p(qp), bits, q_type, relu, _s())
        tok = ctx.res_tok
        if tok is not None and ctx.needs_input_grad[0]:
            # the shortcut is the input of a conv that `res` descends from: that conv's backward-data (still to run) adds this gradient in its store
            tok.d_sc = db
            return da, None, None, None, None, None, None, None
        return da, db, None, None, None, None, None, None


def _ste_from_bits(g, bits, qp):
    """(g * sc) / sc where the bit passes, else 0 -- the gradient of one QuantAdd input from the fused forward's pass bits (torch ops: only a foreign consumer of the
    un-computed gradient ever gets here)"""
    sc = qp.reshape(-1)[0]
    t = (g * sc) / sc
    m = ((bits.reshape(-1, 1) >> torch.arange(8, device=bits.device, dtype=torch.uint8)) & 1).reshape(-1)[:g.numel()].reshape(g.shape).bool()
    return torch.where(m, t, torch.zeros((), dtype=g.dtype, device=g.device))


class IaoQuantAddBN(Function):
    """The END of an IAO residual block (models/resnet.py:21-29, 60-65 with QuantAdd, wqaq/iao/quantize.py:1484-1498) in one pass: ``res`` -- and, in a down-sampling
    block, ``shortcut`` -- arrive as ``LazyBNAct`` (the un-computed output of the BatchNorm behind a dense conv; ``prep()`` has run: the QuantAdd's observers took
    their ranges from it) and out = [relu] (Q(bn(y_res)) + Q(shortcut)) is written straight from the convs' outputs (mn_iao_qadd_bn_fwd).  Backward: each BatchNorm
    side in two passes over (g, y) with the quantizer's clip-STE and the ReLU mask read as bits (mn_iao_qadd_bn_bwd) -- handed to that BatchNorm's autograd node as a
    finished ``LazyBNGrad(kind="bn_done")``; the identity shortcut's gradient comes out of the res side's apply pass."""

    @staticmethod
    def forward(ctx, res, shortcut, qp, bits, q_type, relu, want_minmax, res_tok):
        r = res.recipe
        y = r["y"]
        N, Cc, H, W = y.shape
        dev = y.device
        sb = isinstance(shortcut, LazyBNAct)
        q = shortcut.recipe if sb else None
        sc_x = q["y"] if sb else _chk(shortcut, "shortcut")
        out = torch.empty_like(y)
        bits_r = torch.empty(y.numel() // 8, dtype=torch.uint8, device=dev)
        bits_s = torch.empty(y.numel() // 8, dtype=torch.uint8, device=dev)
        mm, count = None, 0
        with torch.cuda.device(dev):
            if want_minmax:
                count = int(_lib_().mn_bnrelu_mm_count(N, Cc, H * W))
                mm = torch.empty(2 * count, dtype=torch.float32, device=dev)
            with _span(None, 3, 12.25 * y.numel()):
                _call("mn_iao_qadd_bn_fwd", _p(y), _p(r["save"]), _p(r["gamma"]), _p(r["beta"]), _p(sc_x), _p(q["save"]) if sb else None, _p(q["gamma"]) if sb else None,
                      _p(q["beta"]) if sb else None, N, Cc, H * W, _p(qp), bits, q_type, int(relu), _p(out), _p(mm), _p(bits_r), _p(bits_s), _s())
        if mm is not None:
            _PENDING_MINMAX[0] = (mm, count)
        saved = [y, r["save"], r["gamma"], r["beta"], qp, bits_r, bits_s]
        if sb:
            saved += [sc_x, q["save"], q["gamma"], q["beta"]]
        ctx.save_for_backward(*saved)
        ctx.sb = sb
        ctx.res_tok = res_tok
        return out

    @staticmethod
    def backward(ctx, g):
        t = ctx.saved_tensors
        y, save, gamma, beta, qp, bits_r, bits_s = t[:7]
        g = _chk(g, "grad")
        N, Cc, H, W = y.shape
        dev = y.device
        lib = _lib_()

        def side(yy, sv, ga, be, bits_this, bits_other, want_other):
            dy, dgamma, dbeta = torch.empty_like(yy), torch.empty_like(ga), torch.empty_like(be)
            d_other = torch.empty_like(yy) if want_other else None
            ws = torch.empty(int(lib.mn_bnsign_ws_floats(Cc)), dtype=torch.float32, device=dev)
            with _span(None, 3, (24.25 if want_other else 20.25) * yy.numel()):
                _call("mn_iao_qadd_bn_bwd", _p(g), _p(yy), _p(sv), _p(ga), _p(be), N, Cc, H * W, _p(qp), _p(bits_this), _p(bits_other) if want_other else None, _p(dy),
                      _p(d_other), _p(dgamma), _p(dbeta), _p(ws), _s())
            grad = LazyBNGrad(tuple(yy.shape), dev, dict(kind="bn_done", y=yy, dy=dy, dgamma=dgamma, dbeta=dbeta, g=g, bits=bits_this, qp=qp),
                              lambda rr: _ste_from_bits(rr["g"], rr["bits"], rr["qp"]))
            return grad, d_other
        with torch.cuda.device(dev):
            want_sc = ctx.needs_input_grad[1]
            d_res, d_sc = side(y, save, gamma, beta, bits_r, bits_s, want_sc and not ctx.sb)
            if ctx.sb and want_sc:
                d_sc, _ = side(t[7], t[8], t[9], t[10], bits_s, None, False)
        tok = ctx.res_tok
        if tok is not None and not ctx.sb and d_sc is not None:
            tok.d_sc = d_sc          # the identity shortcut is the input of a conv that `res` descends from: that conv's backward-data adds it in its store
            return d_res, None, None, None, None, None, None, None
        return d_res, d_sc, None, None, None, None, None, None


class IaoFakeQuant(Function):
    @staticmethod
    def forward(ctx, x, qp, bits, q_type, is_act):
        x = _chk(x, "input")
        rows = qp.shape[0]
        cols = x.numel() // rows
        y = torch.empty_like(x)
        with torch.cuda.device_of(x):
            _call("mn_iao_fq_fwd", _p(x), _p(y), rows, cols, _p(qp), bits, q_type, int(is_act), _s())
        ctx.save_for_backward(x, qp)
        ctx.cfg = (bits, q_type, int(is_act), rows, cols)
        return y

    @staticmethod
    def backward(ctx, g):
        x, qp = ctx.saved_tensors
        bits, q_type, is_act, rows, cols = ctx.cfg
        g = _chk(g, "grad")
        dx = torch.empty_like(x)
        with torch.cuda.device_of(x):
            _call("mn_iao_fq_bwd", _p(g), _p(x), _p(dx), rows, cols, _p(qp), bits, q_type, is_act, _s())
        return dx, None, None, None, None


ACT_RELU, ACT_LEAKY, ACT_SIGMOID = 1, 2, 3


class IaoFakeQuantAct(Function):
    """act(Q(x)) in one pass (QuantReLU / QuantLeakyReLU / QuantSigmoid, wqaq/iao/quantize.py:1196-1199, 1240-1243, 1279-1282); per-tensor quantizer."""

    @staticmethod
    def forward(ctx, x, qp, bits, q_type, act, slope):
        x = _chk(x, "input")
        y = torch.empty_like(x)
        with torch.cuda.device_of(x):
            _call("mn_iao_fq_act_fwd", _p(x), _p(y), x.numel(), _p(qp), bits, q_type, act, float(slope), _s())
        ctx.save_for_backward(x, qp)
        ctx.cfg = (bits, q_type, act, float(slope))
        return y

    @staticmethod
    def backward(ctx, g):
        x, qp = ctx.saved_tensors
        bits, q_type, act, slope = ctx.cfg
        g = _chk(g, "grad")
        dx = torch.empty_like(x)
        with torch.cuda.device_of(x):
            _call("mn_iao_fq_act_bwd", _p(g), _p(x), _p(dx), x.numel(), _p(qp), bits, q_type, act, slope, _s())
        return dx, None, None, None, None, None


def iao_avgpool_supported(x, k):
    return torch.is_tensor(x) and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.numel() > 0 and \
        bool(_lib_().mn_iao_fq_avgpool_supported(x.shape[2], x.shape[3], k))


class IaoFakeQuantAvgPool(Function):
    """avg_pool2d(Q(x), k, k) (QuantAvgPool2d / QuantAdaptiveAvgPool2d((1, 1)), wqaq/iao/quantize.py:1401-1436) without materialising Q(x)."""

    @staticmethod
    def forward(ctx, x, qp, bits, q_type, k):
        x = _chk(x, "input")
        N, Cc, H, W = x.shape
        y = torch.empty((N, Cc, H // k, W // k), dtype=torch.float32, device=x.device)
        with torch.cuda.device_of(x):
            _call("mn_iao_fq_avgpool_fwd", _p(x), _p(y), N * Cc, H, W, k, _p(qp), bits, q_type, _s())
        ctx.save_for_backward(x, qp)
        ctx.cfg = (bits, q_type, k)
        return y

    @staticmethod
    def backward(ctx, g):
        x, qp = ctx.saved_tensors
        bits, q_type, k = ctx.cfg
        g = _chk(g, "grad")
        N, Cc, H, W = x.shape
        dx = torch.empty_like(x)
        with torch.cuda.device_of(x):
            _call("mn_iao_fq_avgpool_bwd", _p(g), _p(x), _p(dx), N * Cc, H, W, k, _p(qp), bits, q_type, _s())
        return dx, None, None, None, None


class IaoBNFold(Function):
    """(weight_fused, bias_fused) of QuantBNFuseConv2d (wqaq/iao/quantize.py:900-956): w * (gamma / sqrt(var_w + eps)) and beta + (bias - mean) * (gamma /
    sqrt(var_b + eps)) in one launch, the analytic backward in one launch (the reference's ~8 + ~25 element-wise kernels per layer)."""

    @staticmethod
    def forward(ctx, weight, bias, gamma, beta, mean, var_b, var_w, eps):
        weight, gamma, beta, mean, var_b, var_w = (_chk(t, "tensor") for t in (weight, gamma, beta, mean, var_b, var_w))
        bias = _chk(bias, "bias")
        O = weight.shape[0]
        K = weight.numel() // O
        wf, bf = torch.empty_like(weight), torch.empty(O, dtype=torch.float32, device=weight.device)
        with torch.cuda.device_of(weight):
            _call("mn_iao_bnfold_fwd", _p(weight), _p(bias), _p(gamma), _p(beta), _p(mean), _p(var_b), _p(var_w), float(eps), O, K, _p(wf), _p(bf), _s())
        ctx.save_for_backward(weight, bias, gamma, mean, var_b, var_w)
        ctx.eps = float(eps)
        return wf, bf

    @staticmethod
    def backward(ctx, dwf, dbf):
        weight, bias, gamma, mean, var_b, var_w = ctx.saved_tensors
        O = weight.shape[0]
        K = weight.numel() // O
        dev = weight.device
        dwf = _chk(dwf, "grad") if dwf is not None else torch.zeros_like(weight)
        dbf = _chk(dbf.reshape(-1), "grad") if dbf is not None else torch.zeros(O, dtype=torch.float32, device=dev)
        need = ctx.needs_input_grad
        new = lambda: torch.empty(O, dtype=torch.float32, device=dev)
        dw = torch.empty_like(weight) if need[0] else None
        dbias = new() if (bias is not None and need[1]) else None
        dgamma, dbeta = (new() if need[2] else None), (new() if need[3] else None)
        dmean, dvb, dvw = (new() if need[4] else None), (new() if need[5] else None), (new() if need[6] else None)
        with torch.cuda.device_of(weight):
            _call("mn_iao_bnfold_bwd", _p(dwf), _p(dbf), _p(weight), _p(bias), _p(gamma), _p(mean), _p(var_b), _p(var_w), ctx.eps, O, K, _p(dw), _p(dbias),
                  _p(dgamma), _p(dbeta), _p(dmean), _p(dvb), _p(dvw), _s())
        return dw, dbias, dgamma, dbeta, dmean, dvb, dvw, None


class ReluToken:
    """Hand-shake between a block that ends in a ReLU and the ONE consumer of its output: the consumer's backward-data kernel reads that activation anyway (the
    clip-STE of its activation quantizer), so it applies the ReLU's backward mask [a > 0] to the gradient it returns and leaves the tensor here; the producer
    recognises its incoming gradient as exactly that tensor (same storage, same version -- autograd passes a single contribution through untouched, and adds
    two contributions OUT of place while this reference keeps the first one alive) and skips its own masking pass.  Masking is idempotent, so any other path
    (a second consumer, a hook, a foreign op) simply masks again: always correct, one streaming pass slower."""
    __slots__ = ("dx",)

    def __init__(self):
        self.dx = None

    def premasked(self, g):
        d = self.dx
        self.dx = None
        return d is not None and torch.is_tensor(g) and type(g) is torch.Tensor and g.data_ptr() == d.data_ptr() and g.shape == d.shape and g._version == d._version


def relu_premask_ok(x):
    """May the consumer of ``x`` (the output of a fused conv + ReLU) return its input gradient already masked by [x > 0]?  Only when nobody else can observe the
    un-masked gradient of ``x``: no tensor hooks, no retain_grad.  NOT detectable: ``torch.autograd.grad(loss, x)`` on such an intermediate (saliency / Grad-CAM in train
    mode) -- it would receive the masked gradient; MN_NO_RELU_PREMASK=1 switches the pre-masking off for such uses (tests/test_gpu_bnfuse_block.py)."""
    if _NO_RELU_PREMASK:
        return False
    return getattr(x, "_mn_relu_token", None) is not None and not getattr(x, "_backward_hooks", None) and not x.retains_grad


_NO_RELU_PREMASK = False


def iao_bnfuse_pw_supported(x, weight, stride, padding, dilation, groups, in_shuffle):
    if not (torch.is_tensor(x) and type(x) is torch.Tensor and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous() and x.numel() > 0
            and weight.is_cuda and weight.dtype == torch.float32 and weight.is_contiguous()):
        return False
    if x.shape[1] != weight.shape[1] * groups:
        return False
    g = _geom(x.shape, weight.shape, stride, padding, dilation, groups, in_shuffle)
    lib = _lib_()
    return bool(lib.mn_iaobf_gram_supported(C.byref(g))) and bool(lib.mn_iaobf_bwd_data_supported(C.byref(g)))


class IaoBNFusePW(Function):
    """The whole training-mode ``QuantBNFuseConv2d.forward`` (wqaq/iao/quantize.py:837-994, not qaft, not bn_fuse_calib) of a POINTWISE grouped layer, optionally
    with the ReLU the block applies to its output, WITHOUT the statistics convolution (csrc/iao_bnfuse.hip): Gram data of the input -> one preparation launch
    (batch / running statistics, fold, per-channel weight observer + qparams + fake-quant) -> the quantised convolution with ReLU and the (min, max) partials of its
    output in the epilogue.  Backward: quantised backward-weight -> one preparation launch (weight clip-STE, fold backward, dmean / dvar, the raw convolution's
    weight gradient from the Gram data) -> ONE backward-data kernel for the quantised and the raw path.  ``st``: the module (buffers are updated in place exactly as
    the reference does)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, st, aqp, relu, want_mm):
        lib = _lib_()
        x, weight, gamma, beta = _chk(x, "input"), _chk(weight, "weight"), _chk(gamma, "gamma"), _chk(beta, "beta")
        bias = _chk(bias, "bias")
        wq_, aq_ = st.weight_quantizer, st.activation_quantizer
        wobs = wq_.observer
        g = _geom(x.shape, weight.shape, st.stride, st.padding, st.dilation, st.groups, int(getattr(st, "in_shuffle_groups", 0) or 0))
        N, O, H, W = g.N, g.O, g.H, g.W
        Cg = weight.shape[1]
        dev = x.device
        n = float(N * H * W)
        with torch.cuda.device_of(x):
            nb = int(lib.mn_iaobf_gram_ws_bytes(C.byref(g)))
            ws = torch.empty(nb // 4 + 4, dtype=torch.float32, device=dev)
            gram = torch.empty((g.groups, Cg, Cg), dtype=torch.float64, device=dev)
            sx = torch.empty(g.C, dtype=torch.float64, device=dev)
            _call("mn_iaobf_gram", C.byref(g), _p(x), _p(gram), _p(sx), _p(ws), nb, _s())
            first_bn = (not st.pretrained_model) and st.num_flag == 0
            if first_bn:
                st.num_flag += 1
            first_w = wobs.num_flag == 0
            stats = torch.empty((2, O), dtype=torch.float32, device=dev)
            kfold, bias_f = torch.empty(O, dtype=torch.float32, device=dev), torch.empty(O, dtype=torch.float32, device=dev)
            qw, wqp = torch.empty_like(weight), torch.empty((O, 4), dtype=torch.float32, device=dev)
            stats_raw = torch.empty((2, O), dtype=torch.float32, device=dev)
            vc = torch.empty((O, Cg), dtype=torch.float32, device=dev)
            _call("mn_iaobf_gram_stats", _p(weight), _p(bias), _p(gram), _p(sx), O, Cg, g.groups, n, _p(stats_raw), _p(vc), _s())
            _call("mn_iaobf_prep_fwd", _p(weight), _p(bias), _p(gamma), _p(beta), O, Cg, _p(stats_raw), float(st.eps), float(st.momentum),
                  int(first_bn), _p(st.running_mean), _p(st.running_var), wq_.bits, wq_._q_type_static, wobs._kind, int(first_w), float(getattr(wobs, "momentum", 0.1)),
                  _p(wobs.min_val), _p(wobs.max_val), _p(wq_.scale), _p(wq_.zero_point), _p(stats), _p(kfold), _p(bias_f), _p(qw), _p(wqp), _s())
            if first_w:
                wobs.num_flag += 1
            wq_.q_type = wq_._q_type_static
            wq_._last_qp = wqp
            st.__dict__["_mn_last_qw"] = qw          # (tests: the quantised folded weights of this forward)
            st.__dict__["_mn_path"] = "pw"
            aq = ActQ(ACTQ_IAO, aq_.bits, aq_.q_type, 0, aqp.data_ptr())
            wd = WQ(WQ_IAO, wq_.bits, 0, 4, wqp.data_ptr())
            a = torch.empty((N, O, H, W), dtype=torch.float32, device=dev)
            mm, count = None, 0
            if want_mm:
                count = int(lib.mn_conv2d_fwd_act_mm_count(C.byref(g), C.byref(aq), C.byref(wd)))
                mm = torch.empty(2 * count, dtype=torch.float32, device=dev)
            wsf, nbf = _ws(g, 0, dev)
            _call("mn_conv2d_fwd_act", C.byref(g), C.byref(aq), C.byref(wd), _p(x), _p(qw), _p(bias_f), _p(a), int(relu), _p(mm), _p(wsf), nbf, _s())
        ctx.save_for_backward(x, weight, bias, gamma, a if relu else None, stats, qw, wqp, aqp, vc, sx)
        ctx.cfg = (g, aq_.bits, aq_.q_type, wq_.bits, wq_._q_type_static, float(st.eps), n, bool(relu))
        ctx.tok_in = getattr(x, "_mn_relu_token", None)          # the ReLU in FRONT of this block (its producer's token)
        ctx.x_obj = x
        if not relu:
            return a

        def compute():          # the un-rectified convolution output for a consumer other than the block's ReLU: the same kernel without the epilogue
            out = torch.empty_like(a)
            aq2 = ActQ(ACTQ_IAO, aq.bits, aq.q_type, 0, aqp.data_ptr())
            wd2 = WQ(WQ_IAO, wd.bits, 0, 4, wqp.data_ptr())
            with torch.cuda.device_of(x):
                ws2, nb2 = _ws(g, 0, dev)
                _call("mn_conv2d_fwd_act", C.byref(g), C.byref(aq2), C.byref(wd2), _p(x), _p(qw), _p(bias_f), _p(out), 0, None, _p(ws2), nb2, _s())
            return out
        return LazyReluConvOut(a, dict(compute=compute, mm=(mm, count) if want_mm else None))

    @staticmethod
    def backward(ctx, gin):
        x, weight, bias, gamma, a, stats, qw, wqp, aqp, vc, sx = ctx.saved_tensors
        g, a_bits, a_qtype, w_bits, w_qtype, eps, n, relu = ctx.cfg
        lib = _lib_()
        dev = x.device
        if relu and isinstance(gin, LazyReluGrad) and gin._mn_value is None:
            gy = _chk(gin._mn_g, "grad") if gin._mn_premasked else relu_mask(_chk(gin._mn_g, "grad"), a)          # from the block's fused ReLU
        else:
            gy = _chk(gin, "grad")          # a gradient w.r.t. the un-rectified output (a foreign consumer of the conv module): nothing to mask
        O, Cg = weight.shape[0], weight.shape[1]
        aq = ActQ(ACTQ_IAO, a_bits, a_qtype, 0, aqp.data_ptr())
        dx = None
        with torch.cuda.device_of(x):
            dwq, dbf = torch.empty_like(weight), torch.empty(O, dtype=torch.float32, device=dev)
            ws, nb = _ws(g, 2, dev)
            _call("mn_conv2d_bwd_weight", C.byref(g), C.byref(aq), _p(gy), _p(x), _p(dwq), _p(dbf), _p(ws), nb, CONV_ALGO, _s())
            dw = torch.empty_like(weight)
            dbias = torch.empty(O, dtype=torch.float32, device=dev) if bias is not None else None
            dgamma, dbeta = torch.empty(O, dtype=torch.float32, device=dev), torch.empty(O, dtype=torch.float32, device=dev)
            coef = torch.empty((4, O), dtype=torch.float32, device=dev)
            _call("mn_iaobf_prep_bwd", _p(dwq), _p(dbf), _p(weight), _p(bias), _p(gamma), _p(stats), _p(wqp), O, Cg, g.groups, _p(vc), _p(sx), n, eps, w_bits, w_qtype,
                  _p(dw), _p(dbias), _p(dgamma), _p(dbeta), _p(coef), _s())
            if ctx.needs_input_grad[0]:
                pre = ctx.tok_in is not None and relu_premask_ok(ctx.x_obj)
                nbd = int(lib.mn_iaobf_bwd_data_ws_bytes(C.byref(g)))
                wsd = torch.empty(nbd // 4 + 4, dtype=torch.float32, device=dev)
                dx = torch.empty_like(x)
                _call("mn_iaobf_bwd_data", C.byref(g), C.byref(aq), _p(gy), _p(x), _p(weight), _p(qw), _p(wqp), _p(coef), _p(sx), int(pre), _p(dx), _p(wsd), nbd, _s())
                if pre:
                    ctx.tok_in.dx = dx
        ctx.x_obj = None
        return dx, dw, dbias, dgamma, dbeta, None, None, None, None


class ReluOfFusedConv(Function):
    """The block's ``nn.ReLU`` behind a ``LazyReluConvOut``: forward takes the rectified tensor out of the wrapper (no kernel); backward hands the gradient back as a
    ``LazyReluGrad`` -- already masked when the ONE consumer of the activation did it inside its backward-data kernel (``ReluToken``), else masked by the conv."""

    @staticmethod
    def forward(ctx, lazy):
        a = lazy._mn_a
        # The conv node saved THIS tensor object for its own backward (the ReLU mask).  Returning it would hang this node on it as grad_fn: conv node -> saved a ->
        # a.grad_fn (this node) -> next edge -> conv node, a C++ reference cycle only backward() breaks (every fused block of a forward that is never
        # backpropagated leaked x, a, qw).  The output is therefore a second tensor on the same storage and version counter; `a` itself stays without grad_fn.
        ctx.save_for_backward(a)
        ctx.tok = ReluToken()
        return a.detach()

    @staticmethod
    def backward(ctx, g):
        (a,), tok = ctx.saved_tensors, ctx.tok
        return LazyReluGrad(g if type(g) is torch.Tensor else _chk(g, "grad"), a, tok.premasked(g))


def relu_of_fused(lazy):
    """``relu`` of a ``LazyReluConvOut``: the plain rectified tensor, tagged with the ReLU's token and the (min, max) partials the conv's epilogue left."""
    if torch.is_grad_enabled() and lazy.requires_grad:
        out = ReluOfFusedConv.apply(lazy)
        out._mn_relu_token = out.grad_fn.tok if out.grad_fn is not None else None
    else:
        out = lazy._mn_a
    mm = lazy._mn_recipe.get("mm")
    if mm is not None:
        out._mn_minmax = mm + (out._version,)
    return out


def relu_mask(g, a):
    """g * [a > 0]: the backward of a ReLU whose output is ``a`` (only on paths where no consumer pre-masked the gradient)."""
    return add_relu_mask(g, None, a)


def add_relu_mask(a, b, x):
    """(a [+ b]) * [x > 0] in one pass (``mn_add_relu_mask``); b / x may be None."""
    if a.numel() % 4 == 0 and a.is_contiguous() and (b is None or b.is_contiguous()) and (x is None or x.is_contiguous()):
        out = torch.empty_like(a)
        with torch.cuda.device_of(a):
            _call("mn_add_relu_mask", _p(a), _p(b), _p(x), _p(out), a.numel(), _s())
        return out
    out = a if b is None else a + b
    return out if x is None else torch.where(x > 0, out, torch.zeros((), dtype=out.dtype, device=out.device))


def iao_bnfuse_generic_supported(x, weight):
    return (torch.is_tensor(x) and type(x) is torch.Tensor and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous() and x.numel() > 0
            and weight.is_cuda and weight.dtype == torch.float32 and weight.is_contiguous())


class IaoBNFuseGeneric(Function):
    """Training-mode ``QuantBNFuseConv2d.forward`` (wqaq/iao/quantize.py:837-994, not qaft, not bn_fuse_calib) for geometries the pointwise kernels do not cover
    (k x k, > 128 channels per group): the reference's own dataflow -- raw convolution (843-851) -> batch statistics (853-855) -> fold + weight quantizer ->
    quantised convolution (947-955) [-> the block's ReLU] -- as ONE autograd node: the bookkeeping between the convolutions is one launch per direction
    (``mn_iaobf_prep_fwd`` / ``_bwd`` with the statistics given), the two input gradients are summed (and masked for the ReLU in front) in one pass, and the first
    layer (an image with <= 76 taps per output: ``mn_conv2d_first_supported``) runs its quantised convolution and backward-weight on the exact-fp32 first-layer
    kernels over the fake-quantised image.  The raw output y_raw is kept for the backward (d y_raw = dmean / n + 2 dvar (y_raw - mean) / (n - 1))."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, st, aqp, relu, want_mm):
        lib = _lib_()
        x, weight, gamma, beta = _chk(x, "input"), _chk(weight, "weight"), _chk(gamma, "gamma"), _chk(beta, "beta")
        bias = _chk(bias, "bias")
        wq_, aq_ = st.weight_quantizer, st.activation_quantizer
        wobs = wq_.observer
        g = _geom(x.shape, weight.shape, st.stride, st.padding, st.dilation, st.groups, 0)
        O = g.O
        K = weight[0].numel()
        Ho, Wo = _out_hw(g)
        dev = x.device
        n = float(g.N * Ho * Wo)
        none = ActQ(ACTQ_NONE, 0, 0, 0, None)
        # the first layer of a net (an image, <= 128 patch elements per output): the statistics of the raw convolution from the Gram matrix of the im2col matrix
        # (mn_iaobf_gram in patch mode) -- no raw convolution, no y_raw, and in the backward no raw backward-weight (same algebra as the pointwise layers)
        gram_first = (not x.requires_grad) and CONV_ALGO == _lib.MN_ALGO_AUTO and g.KH > 1 and bool(lib.mn_iaobf_gram_supported(C.byref(g))) and \
            bool(lib.mn_conv2d_first_supported(C.byref(g), 0)) and bool(lib.mn_conv2d_first_supported(C.byref(g), 2))
        # a pointwise layer with <= 16 outputs (the classifier conv of nin_gc): HBM-bound streaming kernels on the vector units (csrc/iao_thin.hip)
        thin = CONV_ALGO == _lib.MN_ALGO_AUTO and bool(lib.mn_iaobf_thin_supported(C.byref(g))) and x.requires_grad
        wt = qwt = None
        y_raw = vc = sx = None
        with torch.cuda.device_of(x):
            ws, nb = _ws(g, 0, dev)
            stats_raw = torch.empty((2, O), dtype=torch.float32, device=dev)
            if thin:
                wt = torch.empty((g.C, 16), dtype=torch.float32, device=dev)
                _call("mn_iaobf_thin_pack", _p(weight), O, g.C, _p(wt), _s())
                y_raw = torch.empty((g.N, O, Ho, Wo), dtype=torch.float32, device=dev)
                _call("mn_iaobf_thin_fwd", C.byref(g), _p(x), None, 8, _p(wt), _p(bias), 0, _p(y_raw), None, _s())
                wss = torch.empty(int(lib.mn_bn_stats_ws_floats(g.N, O, Ho * Wo)) + 2, dtype=torch.float32, device=dev)
                _call("mn_bn_stats_fwd", _p(y_raw), g.N, O, Ho * Wo, _p(stats_raw), _p(wss), _s())
            elif gram_first:
                nbg = int(lib.mn_iaobf_gram_ws_bytes(C.byref(g)))
                wsg = torch.empty(nbg // 4 + 4, dtype=torch.float32, device=dev)
                gram = torch.empty((K, K), dtype=torch.float64, device=dev)
                sx = torch.empty(K, dtype=torch.float64, device=dev)
                _call("mn_iaobf_gram", C.byref(g), _p(x), _p(gram), _p(sx), _p(wsg), nbg, _s())
                vc = torch.empty((O, K), dtype=torch.float32, device=dev)
                _call("mn_iaobf_gram_stats", _p(weight), _p(bias), _p(gram), _p(sx), O, K, 1, n, _p(stats_raw), _p(vc), _s())
            else:
                y_raw = torch.empty((g.N, O, Ho, Wo), dtype=torch.float32, device=dev)
                _call("mn_conv2d_fwd", C.byref(g), C.byref(none), None, _p(x), _p(weight), _p(bias), _p(y_raw), _p(ws), nb, CONV_ALGO, _s())
                wss = torch.empty(int(lib.mn_bn_stats_ws_floats(g.N, O, Ho * Wo)) + 2, dtype=torch.float32, device=dev)
                _call("mn_bn_stats_fwd", _p(y_raw), g.N, O, Ho * Wo, _p(stats_raw), _p(wss), _s())
            first_bn = (not st.pretrained_model) and st.num_flag == 0
            if first_bn:
                st.num_flag += 1
            first_w = wobs.num_flag == 0
            stats = torch.empty((2, O), dtype=torch.float32, device=dev)
            kfold, bias_f = torch.empty(O, dtype=torch.float32, device=dev), torch.empty(O, dtype=torch.float32, device=dev)
            qw, wqp = torch.empty_like(weight), torch.empty((O, 4), dtype=torch.float32, device=dev)
            _call("mn_iaobf_prep_fwd", _p(weight), _p(bias), _p(gamma), _p(beta), O, K, _p(stats_raw), float(st.eps), float(st.momentum),
                  int(first_bn), _p(st.running_mean), _p(st.running_var), wq_.bits, wq_._q_type_static, wobs._kind, int(first_w), float(getattr(wobs, "momentum", 0.1)),
                  _p(wobs.min_val), _p(wobs.max_val), _p(wq_.scale), _p(wq_.zero_point), _p(stats), _p(kfold), _p(bias_f), _p(qw), _p(wqp), _s())
            if first_w:
                wobs.num_flag += 1
            wq_.q_type = wq_._q_type_static
            wq_._last_qp = wqp
            st.__dict__["_mn_last_qw"] = qw
            st.__dict__["_mn_path"] = "thin" if thin else "generic"
            aq = ActQ(ACTQ_IAO, aq_.bits, aq_.q_type, 0, aqp.data_ptr())
            wd = WQ(WQ_IAO, wq_.bits, 0, 4, wqp.data_ptr())
            first_layer = (not x.requires_grad) and CONV_ALGO == _lib.MN_ALGO_AUTO and bool(lib.mn_conv2d_first_supported(C.byref(g), 0)) and \
                bool(lib.mn_conv2d_first_supported(C.byref(g), 2))
            out = torch.empty((g.N, O, Ho, Wo), dtype=torch.float32, device=dev)
            xq, mm, count, relu_done = None, None, 0, False
            if thin:
                qwt = torch.empty((g.C, 16), dtype=torch.float32, device=dev)
                _call("mn_iaobf_thin_pack", _p(qw), O, g.C, _p(qwt), _s())
                if relu and want_mm:
                    count = int(lib.mn_iaobf_thin_mm_count(C.byref(g)))
                    mm = torch.empty(2 * count, dtype=torch.float32, device=dev)
                _call("mn_iaobf_thin_fwd", C.byref(g), _p(x), _p(aqp), aq_.bits, _p(qwt), _p(bias_f), int(bool(relu)), _p(out), _p(mm), _s())
                relu_done = True
            elif first_layer:
                xq = torch.empty_like(x)          # the fake-quantised image (tiny): exact fp32 products on the first-layer kernels
                _call("mn_iao_fq_fwd", _p(x), _p(xq), 1, x.numel(), _p(aqp), aq_.bits, aq_.q_type, 1, _s())
                cnt = int(lib.mn_conv2d_fwd_act_mm_count(C.byref(g), C.byref(none), None)) if relu else 0
                if cnt > 0:          # the block's ReLU and the (min, max) partials in the first-layer kernel's epilogue
                    if want_mm:
                        mm, count = torch.empty(2 * cnt, dtype=torch.float32, device=dev), cnt
                    _call("mn_conv2d_fwd_act", C.byref(g), C.byref(none), None, _p(xq), _p(qw), _p(bias_f), _p(out), 1, _p(mm), _p(ws), nb, _s())
                    relu_done = True
                else:
                    _call("mn_conv2d_fwd", C.byref(g), C.byref(none), None, _p(xq), _p(qw), _p(bias_f), _p(out), _p(ws), nb, CONV_ALGO, _s())
            else:
                cnt = int(lib.mn_conv2d_fwd_act_mm_count(C.byref(g), C.byref(aq), C.byref(wd))) if relu else 0
                if cnt > 0:
                    if want_mm:
                        mm, count = torch.empty(2 * cnt, dtype=torch.float32, device=dev), cnt
                    _call("mn_conv2d_fwd_act", C.byref(g), C.byref(aq), C.byref(wd), _p(x), _p(qw), _p(bias_f), _p(out), 1, _p(mm), _p(ws), nb, _s())
                    relu_done = True
                else:
                    _call("mn_conv2d_fwd", C.byref(g), C.byref(aq), C.byref(wd), _p(x), _p(qw), _p(bias_f), _p(out), _p(ws), nb, CONV_ALGO, _s())
            pre = None
            if relu and not relu_done:
                # the conv kernel has no ReLU epilogue: one streaming pass (in place would lose the un-rectified output a foreign consumer may ask for -- it is
                # recomputed in that case, so in place it is) + the (min, max) partials for the next layer's observer
                if want_mm:
                    count = int(lib.mn_relu_mm_count(out.numel()))
                    mm = torch.empty(2 * count, dtype=torch.float32, device=dev) if count > 0 else None
                if out.numel() % 4 == 0:
                    _call("mn_relu_mm", _p(out), _p(out), out.numel(), _p(mm), _s())
                else:
                    out.clamp_(min=0)
                    mm, count = None, 0
        # an input that lies on a quantizer grid (the output of QuantMaxPool2d: value = code * scale, |code| <= 128): the raw convolution's backward-weight may
        # read it as exact codes -- one bf16 term instead of the three-term split of arbitrary fp32 values (a third of the passes of the k x k kernel)
        grid = getattr(x, "_mn_qgrid", None)
        if grid is not None and (grid[3] != x._version or not (2 <= grid[1] <= 8) or grid[2] != 0):
            grid = None
        ctx.save_for_backward(x, weight, bias, gamma, out if relu else None, stats, qw, wqp, aqp, y_raw, xq, grid[0] if grid is not None else None, vc, sx, wt, qwt)
        ctx.xgrid = (grid[1], grid[2]) if grid is not None else None
        ctx.cfg = (g, aq_.bits, aq_.q_type, wq_.bits, wq_._q_type_static, float(st.eps), n, bool(relu), bool(first_layer))
        ctx.tok_in = getattr(x, "_mn_relu_token", None)
        ctx.x_obj = x
        if not relu:
            return out

        def compute():          # the un-rectified output for a consumer other than the block's ReLU
            o2 = torch.empty_like(out)
            with torch.cuda.device_of(x):
                ws2, nb2 = _ws(g, 0, dev)
                if thin:
                    _call("mn_iaobf_thin_fwd", C.byref(g), _p(x), _p(aqp), aq_.bits, _p(qwt), _p(bias_f), 0, _p(o2), None, _s())
                elif first_layer:
                    _call("mn_conv2d_fwd", C.byref(g), C.byref(ActQ(ACTQ_NONE, 0, 0, 0, None)), None, _p(xq), _p(qw), _p(bias_f), _p(o2), _p(ws2), nb2, CONV_ALGO, _s())
                else:
                    aq2 = ActQ(ACTQ_IAO, aq.bits, aq.q_type, 0, aqp.data_ptr())
                    wd2 = WQ(WQ_IAO, wd.bits, 0, 4, wqp.data_ptr())
                    _call("mn_conv2d_fwd", C.byref(g), C.byref(aq2), C.byref(wd2), _p(x), _p(qw), _p(bias_f), _p(o2), _p(ws2), nb2, CONV_ALGO, _s())
            return o2
        return LazyReluConvOut(out, dict(compute=compute, mm=(mm, count) if (want_mm and mm is not None) else None))

    @staticmethod
    def backward(ctx, gin):
        x, weight, bias, gamma, a, stats, qw, wqp, aqp, y_raw, xq, gridqp, vc, sx, wt, qwt = ctx.saved_tensors
        g, a_bits, a_qtype, w_bits, w_qtype, eps, n, relu, first_layer = ctx.cfg
        thin = wt is not None
        dev = x.device
        if relu and isinstance(gin, LazyReluGrad) and gin._mn_value is None:
            gy = _chk(gin._mn_g, "grad") if gin._mn_premasked else relu_mask(_chk(gin._mn_g, "grad"), a)
        else:
            gy = _chk(gin, "grad")
        O = weight.shape[0]
        K = weight[0].numel()
        aq = ActQ(ACTQ_IAO, a_bits, a_qtype, 0, aqp.data_ptr())
        none = ActQ(ACTQ_NONE, 0, 0, 0, None)
        wd = WQ(WQ_IAO, w_bits, 0, 4, wqp.data_ptr())
        dx = None
        with torch.cuda.device_of(x):
            dwq, dbf = torch.empty_like(weight), torch.empty(O, dtype=torch.float32, device=dev)
            ws, nb = _ws(g, 2, dev)
            if thin:
                _call("mn_iaobf_thin_bwd_weight", C.byref(g), _p(gy), _p(x), _p(aqp), a_bits, 0, _p(dwq), _p(dbf), _s())
            elif first_layer:
                _call("mn_conv2d_bwd_weight", C.byref(g), C.byref(none), _p(gy), _p(xq), _p(dwq), _p(dbf), _p(ws), nb, CONV_ALGO, _s())
            else:
                _call("mn_conv2d_bwd_weight", C.byref(g), C.byref(aq), _p(gy), _p(x), _p(dwq), _p(dbf), _p(ws), nb, CONV_ALGO, _s())
            dw = torch.empty_like(weight)
            dbias = torch.empty(O, dtype=torch.float32, device=dev) if bias is not None else None
            dgamma, dbeta = torch.empty(O, dtype=torch.float32, device=dev), torch.empty(O, dtype=torch.float32, device=dev)
            coef = torch.empty((4, O), dtype=torch.float32, device=dev)
            _call("mn_iaobf_prep_bwd", _p(dwq), _p(dbf), _p(weight), _p(bias), _p(gamma), _p(stats), _p(wqp), O, K, g.groups, _p(vc), _p(sx), n, eps, w_bits, w_qtype,
                  _p(dw), _p(dbias), _p(dgamma), _p(dbeta), _p(coef), _s())
            if vc is not None:          # first layer on the Gram data: dw is complete, there is no input gradient
                ctx.x_obj = None
                return None, dw, dbias, dgamma, dbeta, None, None, None, None
            # the statistics path: d y_raw from (dmean, dvar), the raw convolution's backward-weight (and backward-data)
            d_o = torch.empty_like(y_raw)
            _call("mn_bn_stats_bwd", _p(y_raw), _p(stats), _p(coef[2]), _p(coef[3]), _p(d_o), y_raw.shape[0], O, y_raw.shape[2] * y_raw.shape[3], _s())
            if thin:          # the raw backward-weight accumulated into dw, both input gradients (+ the ReLU mask of the block in front) in one pass
                _call("mn_iaobf_thin_bwd_weight", C.byref(g), _p(d_o), _p(x), None, 8, 1, _p(dw), None, _s())
                if ctx.needs_input_grad[0]:
                    pre = ctx.tok_in is not None and relu_premask_ok(ctx.x_obj)
                    dx = torch.empty_like(x)
                    _call("mn_iaobf_thin_bwd_data", C.byref(g), _p(gy), _p(d_o), _p(x), _p(aqp), a_bits, _p(qwt), _p(wt), int(pre), _p(dx), _s())
                    if pre:
                        ctx.tok_in.dx = dx
                ctx.x_obj = None
                return dx, dw, dbias, dgamma, dbeta, None, None, None, None
            dw_raw = torch.empty_like(weight)
            if gridqp is not None:
                aqg = ActQ(ACTQ_IAO, ctx.xgrid[0], ctx.xgrid[1], 0, gridqp.data_ptr())          # x = code * scale exactly: the codes are recovered in the kernel's prologue
                _call("mn_conv2d_bwd_weight", C.byref(g), C.byref(aqg), _p(d_o), _p(x), _p(dw_raw), None, _p(ws), nb, CONV_ALGO, _s())
            else:
                _call("mn_conv2d_bwd_weight", C.byref(g), C.byref(none), _p(d_o), _p(x), _p(dw_raw), None, _p(ws), nb, CONV_ALGO, _s())
            dw.add_(dw_raw)
            if ctx.needs_input_grad[0]:
                dxq, dxr = torch.empty_like(x), torch.empty_like(x)
                ws1, nb1 = _ws(g, 1, dev)
                _call("mn_conv2d_bwd_data", C.byref(g), C.byref(aq), C.byref(wd), _p(gy), _p(qw), _p(x), _p(dxq), _p(ws1), nb1, CONV_ALGO, _s())
                _call("mn_conv2d_bwd_data", C.byref(g), C.byref(none), None, _p(d_o), _p(weight), None, _p(dxr), _p(ws1), nb1, CONV_ALGO, _s())
                pre = ctx.tok_in is not None and relu_premask_ok(ctx.x_obj)
                dx = add_relu_mask(dxq, dxr, x if pre else None)
                if pre:
                    ctx.tok_in.dx = dx
        ctx.x_obj = None
        return dx, dw, dbias, dgamma, dbeta, None, None, None, None




def _valid_qgrid(x):
    """(qp, bits, q_type) of the quantizer grid a tensor is tagged to lie on (the output of the fused QuantMaxPool2d: every value = code * qp[0]), or None."""
    grid = getattr(x, "_mn_qgrid", None)
    if grid is None or grid[3] != x._version or not (2 <= grid[1] <= 8) or grid[2] != 0:
        return None
    return grid[:3]


def iao_bnfuse_g3_supported(x, weight, stride, padding, dilation, groups, in_shuffle):
    """The grouped 3 x 3 BN-fused IAO block on the persistent kernels of csrc/iao_g3.hip: nin_gc's geometry (16 -> 32 channels per group, 8 x 8 / 16 x 16 maps) and
    an input tagged as lying on a symmetric <= 8-bit quantizer grid."""
    if not (iao_bnfuse_generic_supported(x, weight) and weight.dim() == 4 and tuple(weight.shape[2:]) == (3, 3) and _valid_qgrid(x) is not None):
        return False
    g = _geom(x.shape, weight.shape, stride, padding, dilation, groups, int(in_shuffle) if in_shuffle and in_shuffle > 1 else 0)
    return bool(_lib_().mn_iaobf_g3_supported(C.byref(g)))


class IaoBNFuseG3(Function):
    """Training-mode ``QuantBNFuseConv2d.forward`` (wqaq/iao/quantize.py:837-994, not qaft, not bn_fuse_calib) [+ the block's ReLU] for the grouped 3 x 3 layers of
    nin_gc behind a QuantMaxPool2d: the reference's dataflow -- raw convolution (843-851) -> batch statistics (853-855) -> fold + weight quantizer -> quantised
    convolution (947-955) -- on the persistent image-resident kernels of csrc/iao_g3.hip.  The raw output is never written in the forward (statistics from the
    accumulators) and recomputed once in the backward (d y_raw); x is read through the channel shuffle in front of the conv (``st.in_shuffle_groups``)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, st, aqp, relu, want_mm):
        lib = _lib_()
        x, weight, gamma, beta = _chk(x, "input"), _chk(weight, "weight"), _chk(gamma, "gamma"), _chk(beta, "beta")
        bias = _chk(bias, "bias")
        wq_, aq_ = st.weight_quantizer, st.activation_quantizer
        wobs = wq_.observer
        gridqp, grid_bits, _ = _valid_qgrid(x)
        sg = int(st.in_shuffle_groups) if st.in_shuffle_groups > 1 else 0
        g = _geom(x.shape, weight.shape, st.stride, st.padding, st.dilation, st.groups, sg)
        O = g.O
        K = weight[0].numel()
        dev = x.device
        n = float(g.N * g.H * g.W)
        with torch.cuda.device_of(x):
            nb = int(lib.mn_iaobf_g3_ws_bytes(C.byref(g)))
            ws = torch.empty(nb // 4 + 4, dtype=torch.float32, device=dev)
            stats_raw = torch.empty((2, O), dtype=torch.float32, device=dev)
            _call("mn_iaobf_g3_stats", C.byref(g), _p(x), _p(gridqp), grid_bits, _p(weight), _p(bias), _p(stats_raw), _p(ws), nb, _s())
            first_bn = (not st.pretrained_model) and st.num_flag == 0
            if first_bn:
                st.num_flag += 1
            first_w = wobs.num_flag == 0
            stats = torch.empty((2, O), dtype=torch.float32, device=dev)
            kfold, bias_f = torch.empty(O, dtype=torch.float32, device=dev), torch.empty(O, dtype=torch.float32, device=dev)
            qw, wqp = torch.empty_like(weight), torch.empty((O, 4), dtype=torch.float32, device=dev)
            _call("mn_iaobf_prep_fwd", _p(weight), _p(bias), _p(gamma), _p(beta), O, K, _p(stats_raw), float(st.eps), float(st.momentum),
                  int(first_bn), _p(st.running_mean), _p(st.running_var), wq_.bits, wq_._q_type_static, wobs._kind, int(first_w), float(getattr(wobs, "momentum", 0.1)),
                  _p(wobs.min_val), _p(wobs.max_val), _p(wq_.scale), _p(wq_.zero_point), _p(stats), _p(kfold), _p(bias_f), _p(qw), _p(wqp), _s())
            if first_w:
                wobs.num_flag += 1
            wq_.q_type = wq_._q_type_static
            wq_._last_qp = wqp
            st.__dict__["_mn_last_qw"] = qw
            st.__dict__["_mn_path"] = "g3"          # (tests: which kernel family ran this forward)
            out = torch.empty((g.N, O, g.H, g.W), dtype=torch.float32, device=dev)
            mm, count = None, 0
            if relu and want_mm:
                count = int(lib.mn_iaobf_g3_mm_count(C.byref(g)))
                mm = torch.empty(2 * count, dtype=torch.float32, device=dev)
            _call("mn_iaobf_g3_fwd", C.byref(g), _p(x), _p(aqp), aq_.bits, _p(qw), _p(wqp), _p(bias_f), int(bool(relu)), _p(out), _p(mm), _s())
        ctx.save_for_backward(x, weight, bias, gamma, out if relu else None, stats, qw, wqp, aqp, gridqp)
        ctx.cfg = (g, aq_.bits, wq_.bits, wq_._q_type_static, float(st.eps), n, bool(relu), grid_bits, nb)
        ctx.tok_in = getattr(x, "_mn_relu_token", None)
        ctx.x_obj = x
        if not relu:
            return out
        a_bits = aq_.bits

        def compute():          # the un-rectified output for a consumer other than the block's ReLU
            o2 = torch.empty_like(out)
            with torch.cuda.device_of(x):
                _call("mn_iaobf_g3_fwd", C.byref(g), _p(x), _p(aqp), a_bits, _p(qw), _p(wqp), _p(bias_f), 0, _p(o2), None, _s())
            return o2
        return LazyReluConvOut(out, dict(compute=compute, mm=(mm, count) if mm is not None else None))

    @staticmethod
    def backward(ctx, gin):
        x, weight, bias, gamma, a, stats, qw, wqp, aqp, gridqp = ctx.saved_tensors
        g, a_bits, w_bits, w_qtype, eps, n, relu, grid_bits, nb = ctx.cfg
        dev = x.device
        mask = None
        if relu and isinstance(gin, LazyReluGrad) and gin._mn_value is None:
            gy = _chk(gin._mn_g, "grad")
            mask = None if gin._mn_premasked else a          # the block's own ReLU: applied while the gradient is staged (no pass of its own)
        else:
            gy = _chk(gin, "grad")
        O = weight.shape[0]
        K = weight[0].numel()
        dx = None
        with torch.cuda.device_of(x):
            ws = torch.empty(nb // 4 + 4, dtype=torch.float32, device=dev)
            dwq, dbf = torch.empty_like(weight), torch.empty(O, dtype=torch.float32, device=dev)
            _call("mn_iaobf_g3_bwd_weight", C.byref(g), _p(gy), _p(mask), _p(x), _p(aqp), a_bits, 0, _p(dwq), _p(dbf), _p(ws), nb, _s())
            dw = torch.empty_like(weight)
            dbias = torch.empty(O, dtype=torch.float32, device=dev) if bias is not None else None
            dgamma, dbeta = torch.empty(O, dtype=torch.float32, device=dev), torch.empty(O, dtype=torch.float32, device=dev)
            coef = torch.empty((4, O), dtype=torch.float32, device=dev)
            _call("mn_iaobf_prep_bwd", _p(dwq), _p(dbf), _p(weight), _p(bias), _p(gamma), _p(stats), _p(wqp), O, K, g.groups, None, None, n, eps, w_bits, w_qtype,
                  _p(dw), _p(dbias), _p(dgamma), _p(dbeta), _p(coef), _s())
            # the statistics path: d y_raw from (dmean, dvar) on a recomputed raw convolution, its backward-weight accumulated into dw
            dy = torch.empty((g.N, O, g.H, g.W), dtype=torch.float32, device=dev)
            _call("mn_iaobf_g3_dyraw", C.byref(g), _p(x), _p(gridqp), grid_bits, _p(weight), _p(bias), _p(stats), _p(coef), _p(dy), _s())
            _call("mn_iaobf_g3_bwd_weight", C.byref(g), _p(dy), None, _p(x), _p(gridqp), grid_bits, 1, _p(dw), None, _p(ws), nb, _s())
            if ctx.needs_input_grad[0]:
                pre = ctx.tok_in is not None and relu_premask_ok(ctx.x_obj)
                dx = torch.empty_like(x)
                _call("mn_iaobf_g3_bwd_data", C.byref(g), _p(gy), _p(mask), _p(dy), _p(x), _p(aqp), a_bits, _p(qw), _p(wqp), _p(weight), int(pre), _p(dx), _s())
                if pre:
                    ctx.tok_in.dx = dx
        ctx.x_obj = None
        return dx, dw, dbias, dgamma, dbeta, None, None, None, None


def iao_fq_maxpool_supported(x, kernel_size, stride, padding, dilation, ceil_mode):
    two = lambda v: v in (2, (2, 2), [2, 2])
    return (torch.is_tensor(x) and type(x) is torch.Tensor and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous() and x.numel() > 0
            and two(kernel_size) and two(stride if stride is not None else kernel_size) and padding in (0, (0, 0)) and dilation in (1, (1, 1)) and not ceil_mode
            and bool(_lib_().mn_iao_fq_maxpool2x2_supported(x.shape[2], x.shape[3])))


class IaoFakeQuantMaxPool2x2(Function):
    """max_pool2d(Q(x), 2, 2) (QuantMaxPool2d, wqaq/iao/quantize.py:1347-1359) in one pass per direction: Q(x) is never written, the backward applies pool scatter,
    the quantizer's clip-STE and -- when x is the output of a fused conv + ReLU block -- that ReLU's mask in the same pass (``ReluToken``).  ``st``: the module
    (receives the (min, max) partials of the output for the next layer's observer)."""

    @staticmethod
    def forward(ctx, x, qp, bits, q_type, want_mm, st):
        x = _chk(x, "input")
        N, Cc, H, W = x.shape
        y = torch.empty((N, Cc, H // 2, W // 2), dtype=torch.float32, device=x.device)
        idx = torch.empty((N, Cc, H // 2, W // 2), dtype=torch.uint8, device=x.device)
        mm, count = None, 0
        with torch.cuda.device_of(x):
            if want_mm:
                count = int(_lib_().mn_iao_fq_maxpool2x2_mm_count(N * Cc, H, W))
                mm = torch.empty(2 * count, dtype=torch.float32, device=x.device)
            _call("mn_iao_fq_maxpool2x2_fwd", _p(x), N * Cc, H, W, _p(qp), bits, q_type, _p(y), _p(idx), _p(mm), _s())
        ctx.save_for_backward(x, qp, idx)
        ctx.cfg = (bits, q_type)
        ctx.tok_in = getattr(x, "_mn_relu_token", None)
        ctx.x_obj = x
        st.__dict__["_mn_fwd_out"] = (mm, count) if want_mm else None
        st.__dict__["_mn_fwd_grid"] = (qp, bits, q_type)          # every value of y is code * scale of THIS quantizer (the maximum of quantised values)
        return y

    @staticmethod
    def backward(ctx, g):
        x, qp, idx = ctx.saved_tensors
        bits, q_type = ctx.cfg
        g = _chk(g, "grad")
        N, Cc, H, W = x.shape
        dx = torch.empty_like(x)
        pre = ctx.tok_in is not None and relu_premask_ok(ctx.x_obj)
        with torch.cuda.device_of(x):
            _call("mn_iao_fq_maxpool2x2_bwd", _p(g), _p(idx), _p(x), N * Cc, H, W, _p(qp), bits, q_type, int(pre), _p(dx), _s())
        if pre:
            ctx.tok_in.dx = dx
        ctx.x_obj = None
        return dx, None, None, None, None, None


def hist_observe(x, percentile, first, momentum, max_val):
    """HistogramObserver.forward (ref 126-139) on the device: exact k-th smallest |x| + first-call / EMA update of ``max_val``."""
    x = _chk(x.detach(), "input")
    n = x.numel()
    ws = torch.empty(int(_lib_().mn_kth_abs_ws_bytes()) // 4, dtype=torch.int32, device=x.device)
    with torch.cuda.device_of(x):
        _call("mn_hist_observe", _p(x), n, int(percentile * n), int(first), float(momentum), _p(max_val), None, _p(ws), _s())


class BnBatchStats(Function):
    """(mean, unbiased var) over (N, H, W) of a conv output; differentiable (the BN-fuse fold keeps them in the graph)."""

    @staticmethod
    def forward(ctx, o):
        o = _chk(o, "conv output")
        N, Cc, HW = o.shape[0], o.shape[1], o.shape[2] * o.shape[3]
        lib = _lib_()
        stats = torch.empty((2, Cc), dtype=torch.float32, device=o.device)
        ws = torch.empty(int(lib.mn_bn_stats_ws_floats(N, Cc, HW)) + 2, dtype=torch.float32, device=o.device)
        with torch.cuda.device_of(o):
            _call("mn_bn_stats_fwd", _p(o), N, Cc, HW, _p(stats), _p(ws), _s())
        ctx.save_for_backward(o, stats)
        return stats[0], stats[1]

    @staticmethod
    def backward(ctx, dmean, dvar):
        o, stats = ctx.saved_tensors
        dmean, dvar = _chk(dmean, "dmean"), _chk(dvar, "dvar")
        d_o = torch.empty_like(o)
        with torch.cuda.device_of(o):
            _call("mn_bn_stats_bwd", _p(o), _p(stats), _p(dmean), _p(dvar), _p(d_o), o.shape[0], o.shape[1],
                  o.shape[2] * o.shape[3], _s())
        return d_o


LAZY_BN_GRAD = True
FIRST_GRAM = True          # one-pass backward of the first block (round 5 A/B against the sums pass + fold in the backward-weight: c2 107.3k -> 111.9k img/s)


class FirstConvRecord:
    """What ``nn.Conv2dFirst`` leaves on its output for the BatchNorm block behind it: the convolution's operands.  With them that block's backward runs the
    one-pass first-block backward (mn_conv2d_first_xgram + mn_conv2d_bwd_first_*_gram) and hands the finished dw / dbias to the conv node."""

    def __init__(self, y, x, w, bias, stride, padding, dilation, groups):
        self.x, self.w, self.bias = x, w, bias
        self.conv = (stride, padding, dilation, groups)
        self.node = weakref.ref(y.grad_fn) if y.grad_fn is not None else None          # (weak: the record hangs on y, the node is y's own)
        self.version = y._version          # the record is valid for exactly this tensor as the conv wrote it

    def valid_for(self, y):
        return self.node is not None and y.grad_fn is not None and self.node() is y.grad_fn and y._version == self.version


def _first_record(y, training):
    rec = getattr(y, "_mn_first_conv", None)
    if rec is None or not (FIRST_GRAM and LAZY_BN_GRAD and training and type(y) is torch.Tensor and rec.valid_for(y)):
        return None
    return rec


def _first_gram(rec, eps, momentum, running_mean, running_var):
    """Gram data of the first conv's input (mn_conv2d_first_xgram) and, from them, the BatchNorm's batch statistics save = {mean, invstd} + running update -- no pass over
    the conv output.  -> (gram, save)"""
    lib = _lib_()
    x, w, b = _chk(rec.x, "input"), _chk(rec.w, "weight"), _chk(rec.bias, "bias")
    g = _geom(x.shape, w.shape, *rec.conv)
    dev = x.device
    with torch.cuda.device_of(x):
        nbg = int(lib.mn_conv2d_first_xgram_ws_bytes(C.byref(g)))
        wsg = torch.empty(nbg // 4 + 4, dtype=torch.float32, device=dev)
        gram = torch.empty(80 * 80, dtype=torch.float64, device=dev)
        _call("mn_conv2d_first_xgram", C.byref(g), _p(x), _p(gram), _p(wsg), nbg, _s())
        save = torch.empty((2, g.O), dtype=torch.float32, device=dev)
        _call("mn_conv2d_first_gram_bnstats", C.byref(g), _p(w), _p(b), _p(gram), float(eps), float(momentum), _p(running_mean), _p(running_var), _p(save), _s())
    return gram, save


def _first_mask_backward(rec, gram, mask4, da, quant, chan, gamma):
    """The one-pass backward of the first block on (da, pass nibbles): -> (dw, dbias, dgamma, dbeta).  chan rows 2, 3 = the BatchNorm's saved mean, invstd."""
    x, w, b = _chk(rec.x, "input"), _chk(rec.w, "weight"), _chk(rec.bias, "bias")
    g = _geom(x.shape, w.shape, *rec.conv)
    dev = da.device
    save = chan[2:4]
    with torch.cuda.device_of(da):
        dw = torch.empty_like(w)
        db = torch.empty_like(b) if b is not None else None
        dgamma, dbeta = torch.empty(g.O, dtype=torch.float32, device=dev), torch.empty(g.O, dtype=torch.float32, device=dev)
        ws, nb = _ws(g, 2, dev)
        with _span(g, 2, 4.25 * da.numel() + 4 * x.numel()):
            _call("mn_conv2d_bwd_first_mask_gram", C.byref(g), _p(da), _p(mask4), int(quant), _p(save), _p(gamma), _p(w), _p(b), _p(gram), _p(x), _p(dw), _p(db),
                  _p(dgamma), _p(dbeta), _p(ws), nb, _s())
    return dw, db, dgamma, dbeta


def _first_gram_backward(rec, gram, y, kind, da, save, gamma, beta, chan, bits, quant):
    """-> (dw, dbias, dgamma, dbeta) of the first block in ONE pass over (da, y); kind "bn": BatchNorm + sign (save, gamma, beta); "qa": the DoReFa block (chan)."""
    lib = _lib_()
    x, w, b = _chk(rec.x, "input"), _chk(rec.w, "weight"), _chk(rec.bias, "bias")
    g = _geom(x.shape, w.shape, *rec.conv)
    dev = y.device
    with torch.cuda.device_of(y):
        dw = torch.empty_like(w)
        db = torch.empty_like(b) if b is not None else None
        dgamma, dbeta = torch.empty(g.O, dtype=torch.float32, device=dev), torch.empty(g.O, dtype=torch.float32, device=dev)
        ws, nb = _ws(g, 2, dev)
        with _span(g, 2, 8 * y.numel() + 4 * x.numel()):
            if kind == "bn":
                _call("mn_conv2d_bwd_first_bn_gram", C.byref(g), _p(da), _p(y), _p(save), _p(gamma), _p(beta), _p(w), _p(b), _p(gram), _p(x), _p(dw), _p(db),
                      _p(dgamma), _p(dbeta), _p(ws), nb, _s())
            else:
                _call("mn_conv2d_bwd_first_qa_gram", C.byref(g), _p(da), _p(y), _p(chan), bits, quant, _p(w), _p(b), _p(gram), _p(x), _p(dw), _p(db),
                      _p(dgamma), _p(dbeta), _p(ws), nb, _s())
    return dw, db, dgamma, dbeta


class BNSign(Function):
    """a = sign(batch_norm(y)) in one fused op (training or eval statistics); backward = clip-STE of the sign through the
    BatchNorm backward.  The normalised tensor is never materialised (it is recomputed from y in the backward)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, eps, momentum, training, packed=False, lazy_grad=False):
        y, gamma, beta = _chk(y, "input"), _chk(gamma, "weight"), _chk(beta, "bias")
        N, Cc, HW = y.shape[0], y.shape[1], y.shape[2] * y.shape[3]
        a = torch.empty(y.shape, dtype=torch.int8 if packed else torch.float32, device=y.device)
        ctx.first = rec = _first_record(y, training) if lazy_grad else None
        ctx.gram = None
        if rec is not None:
            # y = the first conv's output: batch statistics from the Gram data of its INPUT (kept for the one-pass backward), then the apply pass alone
            ctx.gram, save = _first_gram(rec, eps, momentum, running_mean, running_var)
            with torch.cuda.device_of(y):
                _call("mn_bnsign_apply", _p(y), N, Cc, HW, _p(gamma), _p(beta), _p(save), _p(a), int(bool(packed)), _s())
        else:
            save = torch.empty((2, Cc), dtype=torch.float32, device=y.device)
            ws = torch.empty(int(_lib_().mn_bnsign_ws_floats(Cc)), dtype=torch.float32, device=y.device)
            with torch.cuda.device_of(y):
                _call("mn_bnsign_fwd_i8" if packed else "mn_bnsign_fwd", _p(y), N, Cc, HW, _p(gamma), _p(beta), float(eps), float(momentum),
                      int(training), _p(running_mean), _p(running_var), _p(save), _p(a), _p(ws), _s())
        ctx.save_for_backward(y, gamma, beta, save)
        ctx.training = int(training)
        ctx.lazy_grad = bool(lazy_grad)
        return SignTensor(a) if packed else a

    @staticmethod
    def backward(ctx, da):
        y, gamma, beta, save = ctx.saved_tensors
        da = _chk(da, "grad")
        N, Cc, HW = y.shape[0], y.shape[1], y.shape[2] * y.shape[3]
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
        ws = torch.empty(int(_lib_().mn_bnsign_ws_floats(Cc)), dtype=torch.float32, device=y.device)
        training = ctx.training
        if ctx.lazy_grad and LAZY_BN_GRAD:
            # y comes from the first conv (no backward-data): only the sums are computed here; the conv's backward-weight forms dy itself
            rec = ctx.first
            sums = None
            if rec is not None:
                # ... or everything at once: the BatchNorm backward is linear in dz, so ONE pass over (da, y) gives dw, dgamma, dbeta (csrc/conv_first.hip)
                dw1, db1, dgamma, dbeta = _first_gram_backward(rec, ctx.gram, y, "bn", da, save, gamma, beta, None, 0, 0)
            else:
                sums = torch.empty((2, Cc), dtype=torch.float32, device=y.device)
                with torch.cuda.device_of(y):
                    _call("mn_bnsign_bwd_sums", _p(da), _p(y), _p(save), _p(gamma), _p(beta), N, Cc, HW, _p(dgamma), _p(dbeta), _p(sums), _p(ws), _s())

            def expand(r):
                dy_ = torch.empty_like(r["y"])
                ws_ = torch.empty(int(_lib_().mn_bnsign_ws_floats(Cc)), dtype=torch.float32, device=dy_.device)
                with torch.cuda.device_of(dy_):
                    _call("mn_bnsign_bwd", _p(r["da"]), _p(r["y"]), _p(r["save"]), _p(r["gamma"]), _p(r["beta"]), N, Cc, HW, r["training"], _p(dy_),
                          None, None, _p(ws_), _s())
                return dy_
            recipe = dict(da=da, y=y, save=save, gamma=gamma, beta=beta, sums=sums, training=training)
            if rec is not None:
                recipe.update(kind="first_done", dw=dw1, db=db1, x=rec.x, w=rec.w)
            return LazyBNGrad(y.shape, y.device, recipe, expand), dgamma, dbeta, None, None, None, None, None, None, None
        dy = torch.empty_like(y)
        with torch.cuda.device_of(y):
            _call("mn_bnsign_bwd", _p(da), _p(y), _p(save), _p(gamma), _p(beta), N, Cc, HW, training, _p(dy), _p(dgamma), _p(dbeta),
                  _p(ws), _s())
        return dy, dgamma, dbeta, None, None, None, None, None, None, None


class _PendingMinMax(threading.local):
    """One slot per THREAD (nn.DataParallel-style replicas run their forwards on separate threads; the GIL is released inside the ctypes / HIP calls)."""

    def __init__(self):
        self.v = None

    def __setitem__(self, i, v):
        self.v = v

    def __getitem__(self, i):
        return self.v


_PENDING_MINMAX = _PendingMinMax()


class _PendingAccStats(threading.local):
    """Hand-over of a dense IAO conv's epilogue statistics (exact per-channel sums of its integer accumulator: mn_actq.stats) from inside ``QConv2d.forward`` to the
    module that called it, which attaches them to the conv's output for the BatchNorm behind it (``BNReLU`` -> mn_bn_fwd_acc: no statistics pass over y)."""

    def __init__(self):
        self.slot = [None]

    def __getitem__(self, i):
        return self.slot[i]

    def __setitem__(self, i, v):
        self.slot[i] = v


_PENDING_ACCSTATS = _PendingAccStats()
WANT_ACCSTATS = 0x100          # Python-side bit of ``aq_flags`` (never passed to the library): request mn_actq.stats from the forward
DONATE_DX = 0x200              # Python-side bit: this conv is the shortcut conv of a residual block whose first conv reads the same tensor (set by the IAO prepare)


def take_accstats():
    st, _PENDING_ACCSTATS[0] = _PENDING_ACCSTATS[0], None
    return st



def take_minmax():
    """(mm, count) left by the last forward OF THIS THREAD that was asked for per-block (min, max) partials of its output, or None; cleared by the call."""
    v = _PENDING_MINMAX[0]
    _PENDING_MINMAX[0] = None
    return v


def iao_observe_partials(mm, obs_kind, first, momentum, min_val, max_val):
    """Observer update from the producer's partials (``tensor._mn_minmax``): no pass over the tensor."""
    buf, count = mm
    with torch.cuda.device_of(buf):
        _call("mn_iao_observe_partials", _p(buf), count, obs_kind, int(first), float(momentum), _p(min_val), _p(max_val), _s())


def iao_observe_partials_qparams(mm, obs_kind, first, momentum, min_val, max_val, bits, q_type, is_act, scale, zero_point):
    """Observer update from the producer's partials + the quantizer's update_qparams, one launch; returns the {scale, zero_point, lo, hi} snapshot."""
    buf, count = mm
    qp = torch.empty((1, 4), dtype=torch.float32, device=buf.device)
    with torch.cuda.device_of(buf):
        _call("mn_iao_observe_partials_qparams", _p(buf), count, obs_kind, int(first), float(momentum), _p(min_val), _p(max_val), bits, q_type, int(is_act),
              _p(scale), _p(zero_point), _p(qp), _s())
    return qp


class BNReLU(Function):
    """relu(batch_norm(y)) in one fused op (training or eval statistics): the three / five streaming passes of BNSign with max(z, 0) and
    the ReLU mask -- the ConvBNReLU blocks of the DoReFa / IAO nets (models/nin_gc.py:53-59) otherwise run MIOpen's BatchNorm kernels
    plus separate ReLU forward / backward kernels.  z is recomputed from y in the backward, never stored.  ``_fn``: "mn_bn2d" = the same passes without
    the activation (plain nn.BatchNorm2d: ``BatchNorm2dPlain``)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, eps, momentum, training, fn="mn_bnrelu", want_minmax=False, accstats=None):
        y, gamma, beta = _chk(y, "input"), _chk(gamma, "weight"), _chk(beta, "bias")
        N, Cc, HW = y.shape[0], y.shape[1], y.shape[2] * y.shape[3]
        a = torch.empty_like(y)
        save = torch.empty((2, Cc), dtype=torch.float32, device=y.device)
        ws = torch.empty(int(_lib_().mn_bnsign_ws_floats(Cc)), dtype=torch.float32, device=y.device)
        with torch.cuda.device_of(y):
            if accstats is not None and training and fn in ("mn_bnrelu", "mn_bn2d") and accstats[0].shape[1] == Cc:
                # y came out of a dense IAO conv that left the exact sums of its integer accumulator: batch statistics from those, ONE pass over y
                stats, rows, qp, wscale, sw_stride, cbias = accstats[:6]
                mm = None
                if want_minmax:
                    count = int(_lib_().mn_bnrelu_mm_count(N, Cc, HW))
                    mm = torch.empty(2 * count, dtype=torch.float32, device=y.device)
                with _span(None, 3, 8 * y.numel()):
                    _call("mn_bn_fwd_acc", _p(y), N, Cc, HW, _p(gamma), _p(beta), float(eps), float(momentum), _p(running_mean), _p(running_var), _p(save), _p(a), _p(mm),
                          1 if fn == "mn_bnrelu" else 2, _p(stats), int(rows), _p(qp), _p(wscale), int(sw_stride), _p(cbias), _s())
                if mm is not None:
                    _PENDING_MINMAX[0] = (mm, count)
            elif want_minmax and fn in ("mn_bnrelu", "mn_bn2d"):
                count = int(_lib_().mn_bnrelu_mm_count(N, Cc, HW))
                mm = torch.empty(2 * count, dtype=torch.float32, device=y.device)
                _call(fn + "_fwd_mm", _p(y), N, Cc, HW, _p(gamma), _p(beta), float(eps), float(momentum), int(training), _p(running_mean),
                      _p(running_var), _p(save), _p(a), _p(ws), _p(mm), _s())
                _PENDING_MINMAX[0] = (mm, count)
            else:
                _call(fn + "_fwd", _p(y), N, Cc, HW, _p(gamma), _p(beta), float(eps), float(momentum), int(training), _p(running_mean),
                      _p(running_var), _p(save), _p(a), _p(ws), _s())
        ctx.save_for_backward(y, gamma, beta, save)
        ctx.training = int(training)
        ctx.fn = fn
        return a

    @staticmethod
    def backward(ctx, da):
        y, gamma, beta, save = ctx.saved_tensors
        da = _chk(da, "grad")
        N, Cc, HW = y.shape[0], y.shape[1], y.shape[2] * y.shape[3]
        dgamma, dbeta, dy = torch.empty_like(gamma), torch.empty_like(beta), torch.empty_like(y)
        ws = torch.empty(int(_lib_().mn_bnsign_ws_floats(Cc)), dtype=torch.float32, device=y.device)
        with torch.cuda.device_of(y):
            _call(ctx.fn + "_bwd", _p(da), _p(y), _p(save), _p(gamma), _p(beta), N, Cc, HW, ctx.training, _p(dy), _p(dgamma), _p(dbeta), _p(ws), _s())
        return dy, dgamma, dbeta, None, None, None, None, None, None, None, None


class BNActLazy(Function):
    """Training-mode BatchNorm2d [+ ReLU] behind a dense IAO conv (models/resnet.py:17-29) whose only consumer is the next dense IAO ``QuantConv2d`` or the block's
    ``QuantAdd``: returns a ``LazyBNAct`` -- nothing is computed here.  The consumer pulls (``prep`` -> its observer / qparams -> ``iao_bn_apply_codes`` or the fused
    QuantAdd kernel); the backward is BNReLU's (z recomputed from y)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, eps, momentum, act, accstats):
        y, gamma, beta = _chk(y, "input"), _chk(gamma, "weight"), _chk(beta, "bias")
        N, Cc, H, W = y.shape
        dev = y.device
        stats, rows, qp_in, wscale, sw_stride, cbias, accmm = accstats
        save = torch.empty((2, Cc), dtype=torch.float32, device=dev)
        state = {"mm": None}

        def prep():
            if state["mm"] is None:
                mm = torch.empty(2 * Cc, dtype=torch.float32, device=dev)
                with torch.cuda.device(dev):
                    _call("mn_bn_acc_prep", N, Cc, H * W, _p(gamma), _p(beta), float(eps), float(momentum), _p(running_mean), _p(running_var), _p(save), int(act), _p(stats),
                          _p(accmm), int(rows), _p(qp_in), _p(wscale), int(sw_stride), _p(cbias), _p(mm), _s())
                state["mm"] = mm
            return state["mm"], Cc

        def compute():
            prep()
            a = torch.empty_like(y)
            with torch.cuda.device(dev):
                _call("mn_bn_apply", _p(y), N, Cc, H * W, _p(gamma), _p(beta), _p(save), int(act), _p(a), _s())
            return a
        ctx.save_for_backward(y, gamma, beta, save)
        ctx.act = int(act)
        return LazyBNAct(y.shape, dev, dict(kind="bn_act", y=y, gamma=gamma, beta=beta, save=save, act=int(act), prep=prep, compute=compute))

    @staticmethod
    def backward(ctx, da):
        y, gamma, beta, save = ctx.saved_tensors
        if isinstance(da, LazyBNGrad) and da._mn_value is None and da._mn_recipe.get("kind") == "bn_done" and da._mn_recipe["y"].data_ptr() == y.data_ptr():
            r = da._mn_recipe          # the fused QuantAdd behind this BatchNorm ran its backward too (IaoQuantAddBN: mn_iao_qadd_bn_bwd)
            return r["dy"], r["dgamma"], r["dbeta"], None, None, None, None, None, None
        da = _chk(da, "grad")
        N, Cc, HW = y.shape[0], y.shape[1], y.shape[2] * y.shape[3]
        dgamma, dbeta, dy = torch.empty_like(gamma), torch.empty_like(beta), torch.empty_like(y)
        ws = torch.empty(int(_lib_().mn_bnsign_ws_floats(Cc)), dtype=torch.float32, device=y.device)
        with torch.cuda.device_of(y):
            _call("mn_bnrelu_bwd" if ctx.act == 1 else "mn_bn2d_bwd", _p(da), _p(y), _p(save), _p(gamma), _p(beta), N, Cc, HW, 1, _p(dy), _p(dgamma), _p(dbeta), _p(ws), _s())
        return dy, dgamma, dbeta, None, None, None, None, None, None


def iao_bn_lazy_supported(y, accstats):
    return (torch.is_tensor(y) and type(y) is torch.Tensor and y.is_cuda and y.dim() == 4 and y.dtype == torch.float32 and (y.shape[2] * y.shape[3]) % 8 == 0 and
            accstats is not None and len(accstats) >= 7 and accstats[6] is not None and accstats[0].shape[1] == y.shape[1] and y.is_contiguous() and y.data_ptr() % 16 == 0)


def iao_codes_bytes(x_shape, w_shape, stride, padding, dilation, groups, a_bits, w_bits, dummy):
    """bytes of the signed-code buffer the dense IAO kernels keep for a layer of this geometry (0: they do not cover it); ``dummy``: any device tensor (the descriptors
    only have to be non-NULL for the query)"""
    if CONV_ALGO != _lib.MN_ALGO_AUTO:
        return 0
    g = _geom(x_shape, w_shape, stride, padding, dilation, groups, 0)
    aq = ActQ(ACTQ_IAO, a_bits, 0, 0, dummy.data_ptr())
    wd = WQ(WQ_IAO, w_bits, 0, 4, dummy.data_ptr())
    return int(_lib_().mn_conv2d_iao_codes_bytes(C.byref(g), C.byref(aq), C.byref(wd)))


def iao_bn_apply_codes(lazy, qp, bits, nc):
    """The pulled half of a ``LazyBNAct`` for a dense IAO conv: act(bn(y)) -> that conv's signed activation codes + clip-STE bits, one pass (``lazy.prep()`` ran)."""
    r = lazy.recipe
    y = r["y"]
    N, Cc, H, W = y.shape
    codes = torch.empty(nc, dtype=torch.int8, device=y.device)
    mask = torch.empty(nc // 8, dtype=torch.uint8, device=y.device)
    with torch.cuda.device_of(y):
        with _span(None, 3, 5.125 * y.numel()):
            _call("mn_bn_apply_codes", _p(y), N, Cc, H * W, _p(r["gamma"]), _p(r["beta"]), _p(r["save"]), r["act"], _p(qp), int(bits), _p(codes), _p(mask), _s())
    return codes, mask


class LazyBNActToFloat(Function):
    """LazyBNAct -> the float32 activation, WITH an autograd link (identity backward)."""

    @staticmethod
    def forward(ctx, a):
        return a.materialize()

    @staticmethod
    def backward(ctx, g):
        return g


class BNReLUTailLazy(Function):
    """The BatchNorm2d + ReLU in front of the net's global average pool (models/nin_gc.py:136-147), training mode: returns a ``LazyBNAct`` (kind "bn_tail") -- nothing is
    computed; ``AvgPool2dGlobal`` pulls it through ONE kernel (``BNReLUGapPull``: statistics, normalise, rectify, pool) and hands this node its finished gradients.
    Any other consumer gets relu(batch_norm(y)) from ATen (forward and backward)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, eps, momentum):
        y, gamma, beta = _chk(y, "input"), _chk(gamma, "weight"), _chk(beta, "bias")
        save = torch.empty((2, y.shape[1]), dtype=torch.float32, device=y.device)

        def compute():
            return torch.relu(torch.nn.functional.batch_norm(y, running_mean, running_var, gamma, beta, True, momentum, eps))
        ctx.save_for_backward(y, gamma, beta)
        ctx.eps = eps
        return LazyBNAct(y.shape, y.device, dict(kind="bn_tail", y=y, gamma=gamma, beta=beta, save=save, running_mean=running_mean, running_var=running_var, eps=float(eps),
                                                 momentum=float(momentum), compute=compute))

    @staticmethod
    def backward(ctx, da):
        y, gamma, beta = ctx.saved_tensors
        if isinstance(da, LazyBNGrad) and da._mn_value is None and da._mn_recipe.get("kind") == "bn_done" and da._mn_recipe["y"].data_ptr() == y.data_ptr():
            r = da._mn_recipe
            return r["dy"], r["dgamma"], r["dbeta"], None, None, None, None
        with torch.enable_grad():          # a foreign consumer took the float32 activation: ATen's backward of the same function
            y_, g_, b_ = y.detach().requires_grad_(True), gamma.detach().requires_grad_(True), beta.detach().requires_grad_(True)
            a = torch.relu(torch.nn.functional.batch_norm(y_, None, None, g_, b_, True, 0.0, ctx.eps))
            dy, dg, db = torch.autograd.grad(a, (y_, g_, b_), materialize(da))
        return dy, dg, db, None, None, None, None


class BNReLUGapPull(Function):
    """pooled = mean over the image of relu(batch_norm(y)) from a ``LazyBNAct`` of kind "bn_tail": mn_bnrelu_gap_fwd / _bwd, one launch per direction."""

    @staticmethod
    def forward(ctx, lazy):
        r = lazy.recipe
        y = r["y"]
        N, Cc, H, W = y.shape
        pooled = torch.empty((N, Cc, 1, 1), dtype=torch.float32, device=y.device)
        with torch.cuda.device_of(y):
            _call("mn_bnrelu_gap_fwd", _p(y), N, Cc, H * W, _p(r["gamma"]), _p(r["beta"]), r["eps"], r["momentum"], _p(r["running_mean"]), _p(r["running_var"]), _p(r["save"]),
                  _p(pooled), _s())
        ctx.save_for_backward(y, r["gamma"], r["beta"], r["save"])
        return pooled

    @staticmethod
    def backward(ctx, g):
        y, gamma, beta, save = ctx.saved_tensors
        g = _chk(g, "grad")
        N, Cc, H, W = y.shape
        dy, dgamma, dbeta = torch.empty_like(y), torch.empty_like(gamma), torch.empty_like(beta)
        with torch.cuda.device_of(y):
            _call("mn_bnrelu_gap_bwd", _p(g), _p(y), _p(save), _p(gamma), _p(beta), N, Cc, H * W, _p(dy), _p(dgamma), _p(dbeta), _s())
        return LazyBNGrad((N, Cc, H, W), y.device, dict(kind="bn_done", y=y, dy=dy, dgamma=dgamma, dbeta=dbeta, g=g),
                          lambda rr: (rr["g"].reshape(N, Cc, 1, 1) / float(H * W)).expand(N, Cc, H, W).contiguous())


def bn_tail_supported(y):
    return (type(y) is torch.Tensor and y.is_cuda and y.dtype == torch.float32 and y.dim() == 4 and (y.shape[2] * y.shape[3]) % 4 == 0 and y.is_contiguous() and
            y.data_ptr() % 16 == 0 and y.shape[1] <= 64 and bool(_lib_().mn_bnrelu_gap_supported(y.shape[0], y.shape[1], y.shape[2] * y.shape[3])))


class CrossEntropy(Function):
    """nn.CrossEntropyLoss() (mean reduction; the criterion of the reference's main.py, wqaq/dorefa/main.py:87-92) on [N, K] logits: the loss and its gradient in one
    launch (mn_cross_entropy_fwd); the backward scales that gradient by the incoming scalar (mn_scale_by)."""

    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        logits = _chk(logits, "logits")
        N, K = logits.shape
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        dl = torch.empty_like(logits)
        with torch.cuda.device_of(logits):
            _call("mn_cross_entropy_fwd", _p(logits), _p(target), N, K, int(ignore_index), _p(loss), _p(dl), _s())
        ctx.save_for_backward(dl)
        return loss

    @staticmethod
    def backward(ctx, g):
        dl, = ctx.saved_tensors
        g = _chk(g, "grad")
        out = torch.empty_like(dl)
        with torch.cuda.device_of(dl):
            _call("mn_scale_by", _p(dl), _p(g), _p(out), dl.numel(), _s())
        return out, None, None


def cross_entropy(logits, target, ignore_index=-100):
    """F.cross_entropy(logits, target) (mean reduction, no weights / smoothing) -- on the gfx950 kernel for float32 CUDA logits [N, K] with int64 targets"""
    if (type(logits) is torch.Tensor and logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2 and logits.shape[1] <= 4096 and logits.shape[0] > 0 and
            type(target) is torch.Tensor and target.is_cuda and target.dtype == torch.int64 and target.dim() == 1 and target.shape[0] == logits.shape[0] and
            target.is_contiguous()):
        return CrossEntropy.apply(logits, target, ignore_index)
    return torch.nn.functional.cross_entropy(logits, target, ignore_index=ignore_index)


class GlobalAvgPool(Function):
    """nn.AvgPool2d over the whole image (the tail of nin / nin_gc): [N, C, H, W] -> [N, C, 1, 1]."""

    @staticmethod
    def forward(ctx, x):
        x = _chk(x, "input")
        N, Cc, H, W = x.shape
        y = torch.empty((N, Cc, 1, 1), dtype=torch.float32, device=x.device)
        with torch.cuda.device_of(x):
            _call("mn_avgpool_global_fwd", _p(x), N * Cc, H * W, _p(y), _s())
        ctx.shape = (N, Cc, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = _chk(gy, "grad")
        N, Cc, H, W = ctx.shape
        dx = torch.empty(ctx.shape, dtype=torch.float32, device=gy.device)
        with torch.cuda.device_of(gy):
            _call("mn_avgpool_global_bwd", _p(gy), N * Cc, H * W, _p(dx), _s())
        return dx


def bnrelu_supported(x):
    return torch.is_tensor(x) and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and (x.shape[2] * x.shape[3]) % 4 == 0


class MaxPool2x2F32(Function):
    """nn.MaxPool2d(2, 2) on fp32 activations: the forward keeps the argmax of every window in one byte, the backward is a scatter
    (ATen keeps int64 indices and its backward kernel is several times slower)."""

    @staticmethod
    def forward(ctx, x):
        x = _chk(x, "input")
        N, Cc, H, W = x.shape
        y = torch.empty((N, Cc, H // 2, W // 2), dtype=torch.float32, device=x.device)
        idx = torch.empty((N, Cc, H // 2, W // 2), dtype=torch.uint8, device=x.device)
        with torch.cuda.device_of(x):
            _call("mn_maxpool2x2_f32_fwd", _p(x), N * Cc, H, W, _p(y), _p(idx), _s())
        ctx.save_for_backward(idx)
        ctx.shape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        g = _chk(g, "grad")
        N, Cc, H, W = ctx.shape
        dx = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        with torch.cuda.device_of(g):
            _call("mn_maxpool2x2_f32_bwd", _p(g), _p(idx), N * Cc, H, W, _p(dx), _s())
        return dx


def f32_pool_supported(x, kernel_size, stride, padding, dilation, ceil_mode):
    two = lambda v: v in (2, (2, 2), [2, 2])
    return (torch.is_tensor(x) and type(x) is torch.Tensor and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and two(kernel_size) and two(stride)
            and padding in (0, (0, 0)) and dilation in (1, (1, 1)) and not ceil_mode and bool(_lib_().mn_maxpool2x2_f32_supported(x.shape[2], x.shape[3])))


class SignToFloat(Function):
    """SignTensor -> the float32 +-1 tensor, with an identity backward (for consumers our kernels do not cover)."""

    @staticmethod
    def forward(ctx, a):
        return a.codes.to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


def sign_to_float(a):
    return SignToFloat.apply(a) if isinstance(a, SignTensor) else a


def sign_pool_supported(a, kernel_size, stride, padding, dilation, ceil_mode):
    two = lambda v: v in (2, (2, 2), [2, 2])
    return (isinstance(a, SignTensor) and a.dim() == 4 and two(kernel_size) and two(stride) and padding in (0, (0, 0)) and
            dilation in (1, (1, 1)) and not ceil_mode and a.shape[2] % 2 == 0 and a.shape[3] % 8 == 0)


LAZY_POOL_GRAD = True
POOL_IN_SIGN_PASS = _os0.environ.get("MN_POOL_IN_SIGN", "1") != "0"          # (A/B: pooled sign codes from the producing block's sign pass instead of a pool launch)


def _expand_pool_grad(g, codes):
    N, Cc, H, W = codes.shape
    din = torch.empty(codes.shape, dtype=torch.float32, device=codes.device)
    with torch.cuda.device_of(codes):
        _call("mn_maxpool2x2_sign8_bwd", _p(g), _p(codes), N * Cc, H, W, _p(din), _s())
    return din


class SignMaxPool2x2(Function):
    """nn.MaxPool2d(2, 2) on packed sign activations (models/nin_gc.py:88,119): int8 in, int8 out; the backward routes each
    output gradient to the first maximum of its window, as ATen's max_pool2d does."""

    @staticmethod
    def forward(ctx, a):
        codes = a.codes
        N, Cc, H, W = codes.shape
        out = getattr(a, "_mn_pooled", None)          # the producing block's sign pass already wrote them (ConvBNSign, pool_next)
        if out is None or tuple(out.shape) != (N, Cc, H // 2, W // 2):
            out = torch.empty((N, Cc, H // 2, W // 2), dtype=torch.int8, device=codes.device)
            with torch.cuda.device_of(codes):
                _call("mn_maxpool2x2_sign8_fwd", _p(codes), N * Cc, H, W, _p(out), _s())
        ctx.save_for_backward(codes)
        return SignTensor(out)

    @staticmethod
    def backward(ctx, g):
        (codes,) = ctx.saved_tensors
        g = _chk(g, "grad")
        if LAZY_POOL_GRAD and codes.shape[3] % 4 == 0:
            # hand the POOLED gradient on: the fused block in front expands it inside its kernels (anything else materialises it)
            return LazyPoolGrad(codes.shape, codes.device, g, codes, _expand_pool_grad)
        return _expand_pool_grad(g, codes)


# ------------------------------------------------------------------------------------------------ convolution
def _pair(v):
    return (v, v) if isinstance(v, int) else (int(v[0]), int(v[1]))


def _geom(x_shape, w_shape, stride, padding, dilation, groups, in_shuffle=0):
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    N, Cc, H, W = x_shape
    O, _, KH, KW = w_shape
    return ConvGeom(N, Cc, H, W, O, KH, KW, sh, sw, ph, pw, dh, dw, groups, in_shuffle)


def _out_hw(g):
    return ((g.H + 2 * g.pad_h - g.dil_h * (g.KH - 1) - 1) // g.stride_h + 1,
            (g.W + 2 * g.pad_w - g.dil_w * (g.KW - 1) - 1) // g.stride_w + 1)


def _ws(g, which, device):
    nb = int(_lib_().mn_conv2d_ws_bytes(C.byref(g), which, CONV_ALGO))
    if nb < 0:
        raise MicronetHipError("invalid convolution geometry")
    return torch.empty(max(nb // 4, 4), dtype=torch.float32, device=device), nb


# backward-data and backward-weight of a pointwise binary block in one kernel that reads (da, h) once (k_pwb); MN_PWB=0 restores the two-kernel backward (A/B)
FUSE_PW_BWD = _os0.environ.get("MN_PWB", "1") != "0"


def _wq_desc(wdesc):
    """wdesc = None | (mode, bits, q_type, per_channel, scale_tensor) -> (ctypes struct or None, keep-alive)"""
    if wdesc is None:
        return None
    mode, bits, q_type, per_channel, scale = wdesc
    return WQ(mode, bits, q_type, per_channel, scale.data_ptr() if scale is not None else None)


def _ref(d):
    return None if d is None else C.byref(d)


class UpSums:
    """Hand-over between two consecutive BatchNorm+sign blocks on pointwise convs (round 6): block k leaves its byte stash ``h`` and channel constants ``chan`` on its
    output SignTensor; the ONE-launch backward of block k + 1 (k_pwb) -- whose dx IS block k's d a -- also accumulates the per-channel sums of block k's BatchNorm
    backward into ``ready = (dx, version, partials, splits)``; block k's backward recognises its incoming gradient as exactly that tensor (same storage, shape,
    version: autograd hands a single contribution through untouched) and only finishes the partials (mn_bnh_bwd_sums_final) instead of a pass over (d a, h).  Any other
    path (a second consumer, a hook, a pooled gradient) fails the identity test and takes the ordinary pass: always correct, one streaming pass slower."""
    __slots__ = ("h", "chan", "k", "ready", "kind")

    def __init__(self, h, chan, k, kind=1):          # kind 1: byte stash of a wbwtab block (3: behind a 3x3 conv -- nnz per border class); 2: 16-bit stash of a k-bit (DoReFa) block (its `ready[0]` is the raw dq of a QGrad)
        self.h, self.chan, self.k, self.ready, self.kind = h, chan, int(k), None, kind


UP_SUMS_PLAIN = False          # (the k-bit hand-over also behind a plain gradient: tests only)
UP_SUMS_3X3 = _os0.environ.get("MN_UP_SUMS_3X3", "1") != "0"          # (A/B: the hand-over behind a 3x3 block, k_pwb<1, 0, 0, 3>)
UP_SUMS_FOLD = _os0.environ.get("MN_UP_SUMS", "1") != "0"          # (A/B and the equality test: MN_UP_SUMS=0 restores k_bnh_partial for every block)


class QConv2d(Function):
    """y = conv2d(actq(x), wq, bias): the activation quantizer runs inside the conv kernels' prologue, its clip-STE in the
    backward-data epilogue; ``wq`` is the already fake-quantised weight (its own Function supplies d wq / d w);
    ``wdesc`` tells the code-domain kernels how wq factors into integer codes x scale (None: arbitrary fp32 weights)."""

    @staticmethod
    def forward(ctx, x, wq, bias, stride, padding, dilation, groups, aq_mode, aq_bits, aq_qtype, qp, wdesc, aq_flags, in_shuffle=0, given=None):
        ctx.x_shape = tuple(x.shape)
        if given is not None:               # x is a LazyBNAct whose consumer-side codes + clip-STE bits are already written (iao_bn_apply_codes): fp32 x never exists and
            if aq_mode != ACTQ_IAO or qp is None or wdesc is None or CONV_ALGO != _lib.MN_ALGO_AUTO:          # no kernel reads it; the codes stand in wherever a pointer is due
                raise MicronetHipError("activation codes handed over without the dense IAO path")
            xl, x = x, given[0]
        elif isinstance(x, SignTensor):     # packed +-1 activations: the kernels read the int8 codes (MN_ACTQ_SIGN8)
            if aq_mode != ACTQ_NONE:
                raise MicronetHipError("a SignTensor input cannot be combined with a fused activation quantizer")
            x, aq_mode = x.codes, ACTQ_SIGN8
        else:
            x = _chk(x, "input")
        wq, bias = _chk(wq, "weight"), _chk(bias, "bias")
        xs = ctx.x_shape
        if len(xs) != 4 or wq.dim() != 4 or xs[1] != wq.shape[1] * groups:
            raise MicronetHipError("conv2d shape mismatch: input %s weight %s groups %d" % (xs, tuple(wq.shape), groups))
        g = _geom(xs, wq.shape, stride, padding, dilation, groups, in_shuffle)
        Ho, Wo = _out_hw(g)
        y = torch.empty((g.N, g.O, Ho, Wo), dtype=torch.float32, device=x.device)
        want_stats, donate, aq_flags = bool(aq_flags & WANT_ACCSTATS), bool(aq_flags & DONATE_DX), aq_flags & ~(WANT_ACCSTATS | DONATE_DX)
        aq = ActQ(aq_mode, aq_bits, aq_qtype, aq_flags, qp.data_ptr() if qp is not None else None)
        wd = _wq_desc(wdesc)
        ctx.packed = packed = getattr(wq, "_mn_packed", None) if wd is not None else None
        stats = None
        if want_stats and aq_mode == ACTQ_IAO and wd is not None and qp is not None and CONV_ALGO == _lib.MN_ALGO_AUTO and wdesc[4] is not None:
            rows = int(_lib_().mn_conv2d_iao_stats_rows(C.byref(g), C.byref(aq), C.byref(wd)))
            if rows > 0 and x.data_ptr() % 16 == 0 and y.data_ptr() % 16 == 0:          # dense layer on the int8 matrix cores (16-byte aligned operands, as mn_conv2d_fwd asks):
                stats = torch.empty((rows, g.O, 2), dtype=torch.float64, device=x.device)          # exact sums of acc / acc^2 per channel from the epilogue, for the BatchNorm behind the conv
                accmm = torch.empty((rows, g.O, 2), dtype=torch.int32, device=x.device)            # ... and the accumulator's extrema: the range of that BatchNorm's output without a pass (mn_bn_acc_prep)
                aq.stats, aq.acc_mm = stats.data_ptr(), accmm.data_ptr()
        if packed is not None and packed[0] is not None:
            wd.packed_fwd = packed[0].data_ptr()
        codes = ste_mask = None
        if given is not None:
            codes, ste_mask = given
            nc = int(_lib_().mn_conv2d_iao_codes_bytes(C.byref(g), C.byref(aq), C.byref(wd)))
            if nc <= 0 or nc != codes.numel() or ste_mask.numel() * 8 != nc:
                raise MicronetHipError("activation codes handed over to a layer the dense IAO kernels do not cover")
            aq.codes, aq.ste_mask, aq.flags = codes.data_ptr(), ste_mask.data_ptr(), aq.flags | _lib.MN_ACTQ_CODES_GIVEN
        elif aq_mode == ACTQ_IAO and wd is not None and qp is not None and CONV_ALGO == _lib.MN_ALGO_AUTO and ctx.needs_input_grad[1]:
            nc = int(_lib_().mn_conv2d_iao_codes_bytes(C.byref(g), C.byref(aq), C.byref(wd)))
            if nc > 0:          # dense IAO layer: the forward's signed activation codes are kept for backward-weight (1 byte per element)
                codes = torch.empty(nc, dtype=torch.int8, device=x.device)
                aq.codes = codes.data_ptr()
                if ctx.needs_input_grad[0]:          # ... and the quantizer's clip-STE decisions for backward-data (1 bit per element instead of a second read of x)
                    ste_mask = torch.empty(nc // 8, dtype=torch.uint8, device=x.device)
                    aq.ste_mask = ste_mask.data_ptr()
        with torch.cuda.device_of(x):
            ws, nb = _ws(g, 0, x.device)
            with _span(g, 0, 4 * (x.numel() + y.numel() + wq.numel())):
                _call("mn_conv2d_fwd", C.byref(g), C.byref(aq), _ref(wd), _p(x), _p(wq), _p(bias), _p(y), _p(ws), nb, CONV_ALGO, _s())
        wscale = wdesc[4] if wdesc is not None else None
        if (stats is not None or given is not None) and not _lib_().mn_last_kernel().decode().startswith("k_qd_fwd"):
            if given is not None:
                raise MicronetHipError("activation codes handed over, but the library did not take the dense IAO kernel (%s)" % _lib_().mn_last_kernel().decode())
            stats = None          # the library took another kernel (workspace / alignment): nothing wrote the sums -- the BatchNorm computes its own statistics
        if stats is not None:          # (stats, rows, activation qparams, per-channel weight scale, its stride, conv bias, extrema of acc): what mn_bn_fwd_acc / mn_bn_acc_prep read
            _PENDING_ACCSTATS[0] = (stats, stats.shape[0], qp, wscale, int(wdesc[3]), bias, accmm)
        ctx.iao_codes, ctx.iao_mask = codes, ste_mask
        ctx.res_tok = ctx.donate_tok = None
        if (RES_ADD_FOLD and given is None and aq_mode == ACTQ_IAO and wd is not None and qp is not None and CONV_ALGO == _lib.MN_ALGO_AUTO and ctx.needs_input_grad[0]
                and type(x) is torch.Tensor):
            prev = getattr(x, "_mn_res_token", None)
            if donate and prev is not None and not prev.claimed and not prev.consumed and prev.node is not None and prev.node() is not None:
                # a second conv on the same tensor (the 1 x 1 shortcut conv of a down-sampling block, models/resnet.py:21-29: x feeds the residual function's first conv
                # AND this one): its d x is parked on the first conv's token and added in THAT conv's backward-data store -- autograd's accumulate kernel (12 B per
                # element) disappears.  This node is younger, so its backward runs first; if it ever does not, `consumed` says so and d x goes back to autograd.
                prev.claimed = True
                ctx.donate_tok = prev
            elif _lib_().mn_conv2d_bwd_data_add_supported(C.byref(g), C.byref(aq), C.byref(wd)):
                ctx.res_tok = x._mn_res_token = ResidualToken()          # (x is the caller's tensor object; the node is filled in by qconv2d() below)
        ctx.save_for_backward(x, wq, qp, wscale)
        ctx.cfg = (g, aq_mode, aq_bits, aq_qtype, bias is not None, wdesc[:4] if wdesc is not None else None, aq_flags)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, wq, qp, wscale = ctx.saved_tensors
        g, aq_mode, aq_bits, aq_qtype, has_bias, wd4, aq_flags = ctx.cfg
        if isinstance(gy, LazyBNGrad) and gy._mn_value is None and gy._mn_recipe.get("kind") == "first_done":
            r = gy._mn_recipe              # the block behind this (first) conv already ran the one-pass backward on this conv's operands: dw, dbias are finished
            if aq_mode == ACTQ_NONE and not ctx.needs_input_grad[0] and r["x"].data_ptr() == x.data_ptr() and r["w"].data_ptr() == wq.data_ptr() and \
                    tuple(r["dw"].shape) == tuple(wq.shape) and (not has_bias or r["db"] is not None):
                return None, r["dw"], (r["db"] if has_bias else None), None, None, None, None, None, None, None, None, None, None, None, None
        if isinstance(gy, LazyBNGrad) and gy._mn_value is None and gy._mn_recipe.get("kind") in ("bnh", "bnh_pool") and aq_mode == ACTQ_SIGN8 and wd4 is not None and \
                CONV_ALGO == _lib.MN_ALGO_AUTO:
            r = gy._mn_recipe              # the fused BatchNorm+sign (+ max-pool) behind this conv: dy is formed inside backward-data / backward-weight
            pool = r.get("kind") == "bnh_pool"
            wd = _wq_desc(wd4 + (wscale,))
            if getattr(ctx, "packed", None) is not None and ctx.packed[1] is not None:
                wd.packed_bwd = ctx.packed[1].data_ptr()
            dx = dw = db = None
            want_w = ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2])
            if ctx.needs_input_grad[0] and want_w and FUSE_PW_BWD and _lib_().mn_conv2d_bwd_bnh_supported(C.byref(g), _ref(wd), 1 if pool else 0) and \
                    r["da"].data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0:
                # both gradients from ONE read of (da, h): k_pwb (qgemm_pwb.hip)
                with torch.cuda.device_of(x):
                    dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
                    dw = torch.empty_like(wq)
                    db = torch.empty(g.O, dtype=torch.float32, device=x.device) if has_bias else None
                    nb = int(_lib_().mn_conv2d_bwd_bnh_ws_bytes(C.byref(g)))
                    ws = torch.empty(max(nb // 4, 4), dtype=torch.float32, device=x.device)
                    up = getattr(ctx, "up_rec", None)
                    splits, up9 = 0, False
                    if up is not None and up.kind == 1 and UP_SUMS_FOLD and tuple(up.h.shape) == tuple(x.shape) and up.h.data_ptr() % 16 == 0 and up.chan.shape[0] == 8:
                        splits = int(_lib_().mn_conv2d_bwd_bnh_up_splits(C.byref(g), _ref(wd), 1 if pool else 0, up.k))
                    elif up is not None and up.kind == 3 and UP_SUMS_FOLD and UP_SUMS_3X3 and not pool and tuple(up.h.shape) == tuple(x.shape) and up.h.data_ptr() % 16 == 0 and \
                            up.chan.shape[0] == 17:          # a 3x3 block in front: the stash offset per pixel class (k_pwb<1, 0, 0, 3>)
                        splits = int(_lib_().mn_conv2d_bwd_bnh_up9_splits(C.byref(g), _ref(wd), up.k))
                        up9 = splits > 0
                    if splits > 0:          # ... and the sums of the BatchNorm backward of the block in front (this dx is its d a)
                        part = torch.empty(x.shape[1] * splits * 2, dtype=torch.float64, device=x.device)
                        with _span(g, 1, (3 if pool else 5) * r["h"].numel() + 6 * dx.numel()):
                            if up9:
                                _call("mn_conv2d_bwd_bnh_up9", C.byref(g), _ref(wd), _p(r["da"]), _p(r["h"]), _p(r["chan"]), _p(r["sums"]), r["training"], _p(wq), _p(x),
                                      _p(dx), _p(dw), _p(db), _p(ws), nb, _p(up.h), _p(up.chan), _p(part), _s())
                            else:
                                _call("mn_conv2d_bwd_bnh_up", C.byref(g), _ref(wd), _p(r["da"]), _p(r["h"]), _p(r["own"]) if pool else None, _p(r["chan"]), _p(r["sums"]),
                                      r["training"], _p(wq), _p(x), _p(dx), _p(dw), _p(db), _p(ws), nb, _p(up.h), _p(up.chan), _p(part), _s())
                        up.ready = (dx, dx._version, part, splits)
                    else:
                        with _span(g, 1, (3 if pool else 5) * r["h"].numel() + 5 * dx.numel()):
                            _call("mn_conv2d_bwd_bnh", C.byref(g), _ref(wd), _p(r["da"]), _p(r["h"]), _p(r["own"]) if pool else None, _p(r["chan"]), _p(r["sums"]),
                                  r["training"], _p(wq), _p(x), _p(dx), _p(dw), _p(db), _p(ws), nb, _s())
                return dx, dw, db, None, None, None, None, None, None, None, None, None, None, None, None
            with torch.cuda.device_of(x):
                if ctx.needs_input_grad[0]:
                    dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
                    ws, nb = _ws(g, 1, x.device)
                    with _span(g, 1, (3 if pool else 5) * r["h"].numel() + 4 * dx.numel()):
                        if pool:
                            _call("mn_conv2d_bwd_data_bnh_pool", C.byref(g), _ref(wd), _p(r["da"]), _p(r["h"]), _p(r["own"]), _p(r["chan"]), _p(r["sums"]), r["training"],
                                  _p(wq), _p(dx), _p(ws), nb, _s())
                        else:
                            _call("mn_conv2d_bwd_data_bnh", C.byref(g), _ref(wd), _p(r["da"]), _p(r["h"]), _p(r["chan"]), _p(r["sums"]), r["training"], _p(wq),
                                  _p(dx), _p(ws), nb, _s())
                if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
                    dw = torch.empty_like(wq)
                    db = torch.empty(g.O, dtype=torch.float32, device=x.device) if has_bias else None
                    ws, nb = _ws(g, 2, x.device)
                    with _span(g, 2, (3 if pool else 5) * r["h"].numel() + x.numel()):
                        if pool:
                            _call("mn_conv2d_bwd_weight_bnh_pool", C.byref(g), _p(r["da"]), _p(r["h"]), _p(r["own"]), _p(r["chan"]), _p(r["sums"]), r["training"], _p(x),
                                  _p(dw), _p(db), _p(ws), nb, _s())
                        else:
                            _call("mn_conv2d_bwd_weight_bnh", C.byref(g), _p(r["da"]), _p(r["h"]), _p(r["chan"]), _p(r["sums"]), r["training"], _p(x), _p(dw),
                                  _p(db), _p(ws), nb, _s())
            return dx, dw, db, None, None, None, None, None, None, None, None, None, None, None, None
        if isinstance(gy, LazyBNGrad) and gy._mn_value is None and gy._mn_recipe.get("kind") == "qa" and aq_mode == ACTQ_NONE and not ctx.needs_input_grad[0] and \
                CONV_ALGO == _lib.MN_ALGO_AUTO and _lib_().mn_conv2d_first_supported(C.byref(g), 2):
            r = gy._mn_recipe              # the DoReFa block (BatchNorm + ReLU + next quantizer) behind the first conv: dy is formed inside the backward-weight kernel
            dw = torch.empty_like(wq)
            db = torch.empty(g.O, dtype=torch.float32, device=x.device) if has_bias else None
            with torch.cuda.device_of(x):
                ws, nb = _ws(g, 2, x.device)
                _call("mn_conv2d_bwd_weight_first_qa", C.byref(g), _p(r["dq"]), _p(r["y"]), _p(r["chan"]), _p(r["sums"]), r["bits"], r["quant"], r["training"],
                      _p(x), _p(dw), _p(db), _p(ws), nb, _s())
            return None, dw, db, None, None, None, None, None, None, None, None, None, None, None, None
        if isinstance(gy, LazyBNGrad) and gy._mn_value is None and gy._mn_recipe.get("kind") not in ("bnh", "bnh_pool", "qa", "first_done") and aq_mode == ACTQ_NONE and not ctx.needs_input_grad[0] and \
                CONV_ALGO == _lib.MN_ALGO_AUTO and _lib_().mn_conv2d_first_supported(C.byref(g), 2):
            r = gy._mn_recipe              # the BatchNorm+sign behind the first conv: dy is formed inside the backward-weight kernel
            dw = torch.empty_like(wq)
            db = torch.empty(g.O, dtype=torch.float32, device=x.device) if has_bias else None
            with torch.cuda.device_of(x):
                ws, nb = _ws(g, 2, x.device)
                _call("mn_conv2d_bwd_weight_first_bn", C.byref(g), _p(r["da"]), _p(r["y"]), _p(r["save"]), _p(r["gamma"]), _p(r["beta"]), _p(r["sums"]),
                      r["training"], _p(x), _p(dw), _p(db), _p(ws), nb, _s())
            return None, dw, db, None, None, None, None, None, None, None, None, None, None, None, None
        gy = _chk(gy, "grad")
        x_shape = getattr(ctx, "x_shape", None) or tuple(x.shape)          # (x is the int8 code buffer when the forward was handed the codes: LazyBNAct)
        x_numel = x_shape[0] * x_shape[1] * x_shape[2] * x_shape[3]
        aq = ActQ(aq_mode, aq_bits, aq_qtype, aq_flags, qp.data_ptr() if qp is not None else None)
        if getattr(ctx, "iao_codes", None) is not None:
            aq.codes = ctx.iao_codes.data_ptr()
            if getattr(ctx, "iao_mask", None) is not None:
                aq.ste_mask = ctx.iao_mask.data_ptr()
        wd = _wq_desc(wd4 + (wscale,)) if wd4 is not None else None
        if wd is not None and getattr(ctx, "packed", None) is not None and ctx.packed[1] is not None:
            wd.packed_bwd = ctx.packed[1].data_ptr()
        dx = dw = db = None
        tok = getattr(ctx, "res_tok", None)
        d_sc = None
        if tok is not None:
            tok.consumed = True
            if tok.d_sc is not None:          # a QuantAdd (or the block's shortcut conv) parked a gradient w.r.t. x here: it is added in the store of dx
                d_sc, tok.d_sc = tok.d_sc, None
        with torch.cuda.device_of(x):
            if ctx.needs_input_grad[0]:
                dx = torch.empty(x_shape, dtype=torch.float32, device=x.device)
                ws, nb = _ws(g, 1, x.device)
                # (every operand the dense kernel insists on is checked here: a misaligned view must take the dx.add_ fallback below, not raise inside backward)
                fold = d_sc is not None and d_sc.shape == dx.shape and d_sc.is_contiguous() and d_sc.data_ptr() % 16 == 0 and wd is not None and \
                    gy.data_ptr() % 16 == 0 and (x.data_ptr() % 16 == 0 or getattr(ctx, "iao_mask", None) is not None) and \
                    bool(_lib_().mn_conv2d_bwd_data_add_supported(C.byref(g), C.byref(aq), C.byref(wd)))
                if fold:
                    aq.dx_add = d_sc.data_ptr()
                with _span(g, 1, 4 * (gy.numel() + dx.numel() + wq.numel() + (x_numel if aq_mode not in (ACTQ_NONE, ACTQ_SIGN8) else 0))):
                    _call("mn_conv2d_bwd_data", C.byref(g), C.byref(aq), _ref(wd), _p(gy), _p(wq), _p(x), _p(dx), _p(ws), nb, CONV_ALGO, _s())
                aq.dx_add = None
                if d_sc is not None and not fold:
                    dx.add_(d_sc)
            elif d_sc is not None:
                dx = d_sc
            if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
                dw = torch.empty_like(wq)
                db = torch.empty(g.O, dtype=torch.float32, device=x.device) if has_bias else None
                ws, nb = _ws(g, 2, x.device)
                with _span(g, 2, 4 * (gy.numel() + x_numel + dw.numel())):
                    _call("mn_conv2d_bwd_weight", C.byref(g), C.byref(aq), _p(gy), _p(x), _p(dw), _p(db), _p(ws), nb, CONV_ALGO, _s())
        dtok = getattr(ctx, "donate_tok", None)
        if dtok is not None and dx is not None and not dtok.consumed and dtok.d_sc is None and dtok.node is not None and dtok.node() is not None:
            dtok.d_sc, dx = dx, None          # (the other conv on x adds it while it stores its own d x)
        return dx, dw, db, None, None, None, None, None, None, None, None, None, None, None, None


class QConv2dLazy(Function):
    """QConv2d on packed sign activations whose result is NOT computed: returns a ``LazyConvOut`` carrying the recipe (see
    micronet_amd/sign_tensor.py).  Backward is QConv2d's (mn_conv2d_bwd_data / _bwd_weight on the int8 codes)."""

    @staticmethod
    def forward(ctx, x, wq, bias, stride, padding, dilation, groups, wdesc, in_shuffle):
        codes = x.codes
        wq, bias = _chk(wq, "weight"), _chk(bias, "bias")
        g = _geom(codes.shape, wq.shape, stride, padding, dilation, groups, in_shuffle)
        Ho, Wo = _out_hw(g)
        wscale = wdesc[4] if wdesc is not None else None
        ctx.save_for_backward(codes, wq, None, wscale)
        ctx.cfg = (g, ACTQ_SIGN8, 8, 0, bias is not None, wdesc[:4] if wdesc is not None else None, 0)
        ctx.packed = packed = getattr(wq, "_mn_packed", None)          # (the step's pre-packed code images: pack_pointwise_weights)
        ctx.up_rec = getattr(x, "_mn_up", None)          # the BatchNorm+sign block that produced x (UpSums)

        def compute():
            y = torch.empty((g.N, g.O, Ho, Wo), dtype=torch.float32, device=codes.device)
            aq = ActQ(ACTQ_SIGN8, 8, 0, 0, None)
            wd = _wq_desc(wdesc)
            with torch.cuda.device_of(codes):
                ws, nb = _ws(g, 0, codes.device)
                _call("mn_conv2d_fwd", C.byref(g), C.byref(aq), _ref(wd), _p(codes), _p(wq), _p(bias), _p(y), _p(ws), nb, CONV_ALGO, _s())
            return y
        recipe = dict(codes=codes, wq=wq, bias=bias, geom=g, wdesc=wdesc, compute=compute, packed=packed)
        return LazyConvOut((g.N, g.O, Ho, Wo), codes.device, recipe)

    @staticmethod
    def backward(ctx, gy):
        return QConv2d.backward(ctx, gy)[:9]


def qconv_bnsign_supported(x, wq, stride, padding, dilation, groups, wdesc, in_shuffle):
    if not isinstance(x, SignTensor) or wdesc is None or CONV_ALGO != _lib.MN_ALGO_AUTO:
        return False
    g = _geom(x.shape, wq.shape, stride, padding, dilation, groups, in_shuffle or 0)
    return bool(_lib_().mn_qconv_bnsign_stash_supported(C.byref(g), _ref(_wq_desc(wdesc))))


import os as _os
# Fold the BatchNorm+sign backward into the block's own conv backward (k_pwd / k_pws_wgrad_s form dy from (da, h) in registers / in
# the LDS staging pass, dy is never written).  With the LDS-staged backward-weight kernel (the fold costs one pass per BLOCK there)
# this is +3.5 % step throughput on c2 (3.25 -> 3.14 ms): ON by default, MN_BNH_FOLD=0 switches it off.
FOLD_BN_INTO_CONV_BWD = True
# ... and the 2x2 max-pool behind a block as well (k_pwd<.., 2> / k_pws_wgrad_s<.., 2, ..>): MN_BNH_POOL_FOLD=0 restores mn_bnh_bwd_apply's full-size dy (A/B)
FOLD_POOL_INTO_CONV_BWD = True


class ConvBNSign(Function):
    """a = sign(batch_norm(y)) for a LazyConvOut y: conv, batch statistics, normalisation and sign in the fused kernels of
    qgemm_sign.hip -- y is never written; the forward stashes the integer conv result in ONE byte per element (h) and the
    per-channel integer thresholds (chan), so the backward -- clip-STE of the sign through the BatchNorm backward -- is two
    streaming passes over (da, h) with no convolution recompute (mn_bnh_bwd_sums / mn_bnh_bwd_apply)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, eps, momentum, training, nbt=None, pool_next=False):
        r = y.recipe
        codes, wq, bias, g, wdesc = r["codes"], r["wq"], r["bias"], r["geom"], r["wdesc"]
        gamma, beta = _chk(gamma, "weight"), _chk(beta, "bias")
        a = torch.empty(y.shape, dtype=torch.int8, device=codes.device)
        h = torch.empty(y.shape, dtype=torch.uint8, device=codes.device)
        save = torch.empty((2, g.O), dtype=torch.float32, device=codes.device)
        chan = torch.empty((int(_lib_().mn_qconv_bnsign_stash_chan_rows(C.byref(g))), g.O), dtype=torch.float32, device=codes.device)
        wd = _wq_desc(wdesc)
        pk = r.get("packed")
        if pk is not None and pk[0] is not None:
            wd.packed_fwd = pk[0].data_ptr()
        with torch.cuda.device_of(codes):
            nb = int(_lib_().mn_qconv_bnsign_stash_ws_bytes(C.byref(g)))
            ws = torch.empty(nb // 4 + 4, dtype=torch.float32, device=codes.device)
            # a 2x2 / stride-2 max-pool behind the block (prepare() marked it): the sign pass writes the pooled codes too, the pool module hands them on (SignMaxPool2x2)
            ap = None
            if pool_next and training and POOL_IN_SIGN_PASS and a.data_ptr() % 16 == 0 and h.data_ptr() % 16 == 0 and \
                    _lib_().mn_qconv_bnsign_fwd_stash_pool_supported(C.byref(g), _ref(wd)):
                ap = torch.empty((y.shape[0], y.shape[1], y.shape[2] // 2, y.shape[3] // 2), dtype=torch.int8, device=codes.device)
                _call("mn_qconv_bnsign_fwd_stash_pool", C.byref(g), _ref(wd), _p(codes), _p(wq), _p(bias), _p(gamma), _p(beta), float(eps), float(momentum),
                      int(training), _p(running_mean), _p(running_var), _p(nbt), _p(save), _p(a), _p(ap), _p(h), _p(chan), _p(ws), nb, _s())
            else:
                _call("mn_qconv_bnsign_fwd_stash", C.byref(g), _ref(wd), _p(codes), _p(wq), _p(bias), _p(gamma), _p(beta), float(eps), float(momentum),
                      int(training), _p(running_mean), _p(running_var), _p(nbt), _p(save), _p(a), _p(h), _p(chan), _p(ws), nb, _s())
        ctx.save_for_backward(h, chan, gamma, beta)
        ctx.training = int(training)
        ctx.fold_ok = FOLD_BN_INTO_CONV_BWD and bool(_lib_().mn_conv2d_bnh_supported(C.byref(g), _ref(wd)))     # the conv's own backward can form dy from (da, h)
        ctx.fold_pool_ok = FOLD_POOL_INTO_CONV_BWD and ctx.fold_ok and bool(_lib_().mn_conv2d_bnh_pool_supported(C.byref(g), _ref(wd)))      # ... and from the POOLED gradient
        out = SignTensor(a)
        if ap is not None:
            out._mn_pooled = ap
        ctx.up_rec = None
        if chan.shape[0] in (8, 17) and UP_SUMS_FOLD:          # (8: pointwise block, one nnz per channel; 17: 3x3 block, nnz per border class) -- the next block's
            ctx.up_rec = out._mn_up = UpSums(h, chan, wq.shape[1] * wq.shape[2] * wq.shape[3], kind=1 if chan.shape[0] == 8 else 3)          # backward may form this block's sums
        return out

    @staticmethod
    def backward(ctx, da):
        h, chan, gamma, beta = ctx.saved_tensors
        training = ctx.training
        pooled = isinstance(da, LazyPoolGrad) and da._mn_value is None
        if pooled:
            grad, own = da._mn_pg, da._mn_codes           # the 2x2 max-pool behind this block: gradient still pooled
        else:
            grad, own = _chk(da, "grad"), None
        N, Cc, H, W = h.shape
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
        sums = torch.empty((2, Cc), dtype=torch.float32, device=h.device)
        ws = torch.empty(int(_lib_().mn_bnsign_ws_floats(Cc)), dtype=torch.float32, device=h.device)
        rec = getattr(ctx, "up_rec", None)
        ready = None
        if rec is not None:
            ready, rec.ready = rec.ready, None
        with torch.cuda.device_of(h):
            if ready is not None and not pooled and type(da) is torch.Tensor and da.data_ptr() == ready[0].data_ptr() and tuple(da.shape) == tuple(ready[0].shape) and \
                    da._version == ready[1] and da.is_contiguous():
                # the producer of d a (the next block's one-launch backward) already summed dz and dz zhat per channel: only the fixed-order finish is left
                _call("mn_bnh_bwd_sums_final", _p(ready[2]), ready[3], N, Cc, H, W, _p(dgamma), _p(dbeta), _p(sums), _s())
            else:
                _call("mn_bnh_bwd_sums", _p(grad), _p(h), _p(own), _p(chan), N, Cc, H, W, _p(dgamma), _p(dbeta), _p(sums), _p(ws), _s())
            if pooled and getattr(ctx, "fold_pool_ok", False) and LAZY_BN_GRAD:
                # a 2x2 max-pool behind the block: the conv's backward-data / backward-weight route the pooled gradient to each window's first +1 and apply the
                # BatchNorm+sign backward while (dpool, own codes, h) stream in -- the full-size dy (4 B per element written, then read twice) never exists
                def expand_p(r):
                    dy_ = torch.empty(r["h"].shape, dtype=torch.float32, device=r["h"].device)
                    with torch.cuda.device_of(dy_):
                        _call("mn_bnh_bwd_apply", _p(r["da"]), _p(r["h"]), _p(r["own"]), _p(r["chan"]), _p(r["sums"]), N, Cc, H, W, r["training"], _p(dy_), _s())
                    return dy_
                recipe = dict(kind="bnh_pool", da=grad, own=own, h=h, chan=chan, sums=sums, training=training)
                return LazyBNGrad(h.shape, h.device, recipe, expand_p), dgamma, dbeta, None, None, None, None, None, None, None
            if not pooled and ctx.fold_ok and LAZY_BN_GRAD:
                # d loss / d y is not written: the convolution's backward-data / backward-weight form it from (da, h) while they stream in
                def expand(r):
                    dy_ = torch.empty(r["h"].shape, dtype=torch.float32, device=r["h"].device)
                    with torch.cuda.device_of(dy_):
                        _call("mn_bnh_bwd_apply", _p(r["da"]), _p(r["h"]), None, _p(r["chan"]), _p(r["sums"]), N, Cc, H, W, r["training"], _p(dy_), _s())
                    return dy_
                recipe = dict(kind="bnh", da=grad, h=h, chan=chan, sums=sums, training=training)
                return LazyBNGrad(h.shape, h.device, recipe, expand), dgamma, dbeta, None, None, None, None, None, None, None
            dy = torch.empty(h.shape, dtype=torch.float32, device=h.device)
            _call("mn_bnh_bwd_apply", _p(grad), _p(h), _p(own), _p(chan), _p(sums), N, Cc, H, W, training, _p(dy), _s())
        return dy, dgamma, dbeta, None, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------------ k-bit (DoReFa) fused block
def qconv_bnq_supported(x, wq, stride, padding, dilation, groups, w_bits, in_shuffle):
    """True when conv(x) for a ``QActTensor`` x and DoReFa weights can stay un-computed: the fused kernels (16 / 32-bit stash forward, code-reading
    backward-weight, STE-free backward-data) cover this geometry (grouped 1x1 / 3x3 stride 1: qgemm_sign / qgemm_k3s; dense layers with C, O multiples of
    64, 3x3 stride 1 / 2 and 1x1 stride 2: qgemm_dense)."""
    if not isinstance(x, QActTensor) or CONV_ALGO != _lib.MN_ALGO_AUTO or x.dim() != 4 or not (2 <= w_bits <= 8):
        return False
    one = lambda v: v in (1, (1, 1), [1, 1])
    if not one(dilation):
        return False
    g = _geom(x.shape, wq.shape, stride, padding, dilation, groups, in_shuffle or 0)
    wd = WQ(WQ_DOREFA, w_bits, 0, 0, None)
    return bool(_lib_().mn_qconv_bnq_supported(C.byref(g), C.byref(wd), x.bits))


def _wq_dorefa(w_bits, packed=None, which=0):
    """mn_wq of DoReFa weights; ``packed`` = the (forward, backward-data) code images mn_qd_pack_multi wrote for this step's weights (or None)."""
    wd = WQ(WQ_DOREFA, w_bits, 0, 0, None)
    if packed is not None:
        if which == 0 and packed[0] is not None:
            wd.packed_fwd = packed[0].data_ptr()
        if which == 1 and packed[1] is not None:
            wd.packed_bwd = packed[1].data_ptr()
    return wd


def pack_dense_weights(mods_wq, w_bits, qps=None):
    """One launch writing the weight codes of every dense-family conv (both fragment orders) for this step: ``mods_wq`` = [(conv module, quantised weight)].
    The images ride on the quantised weight tensor (``_mn_packed``) to the convs' forward and backward.  ``qps`` (IAO): the per-channel {scale, ...} snapshots
    [O, 4] of the same tensors -- codes = rint(w / scale[o]); None (DoReFa): codes = rint(w (2^bits - 1))."""
    lib = _lib_()
    items, scales = [], []
    for k, (m, wq) in enumerate(mods_wq):
        if wq.dim() != 4 or not wq.is_cuda or getattr(m, "groups", 1) != 1 or wq.shape[0] % 64 or wq.shape[1] % 64:
            continue
        g = _geom((1, wq.shape[1], 8, 8), wq.shape, m.stride, m.padding, m.dilation, 1, 0)
        nb = int(lib.mn_qd_packed_bytes(C.byref(g)))
        if nb <= 0:
            continue
        items.append((wq, nb))
        if qps is not None:
            scales.append(qps[k])
    if not items:
        return
    n = len(items)
    dev = items[0][0].device
    total = sum(2 * nb for _, nb in items)
    buf = torch.empty(total, dtype=torch.uint8, device=dev)
    outs, off = [], 0
    for wq, nb in items:
        outs.append((buf[off:off + nb], buf[off + nb:off + 2 * nb]))
        off += 2 * nb
    PA, LA, IA = C.c_void_p * n, C.c_int64 * n, C.c_int32 * n
    with torch.cuda.device_of(buf):
        _call("mn_qd_pack_multi", PA(*[wq.data_ptr() for wq, _ in items]), PA(*[o[0].data_ptr() for o in outs]), PA(*[o[1].data_ptr() for o in outs]),
              LA(*[wq.shape[0] for wq, _ in items]), LA(*[wq.shape[1] for wq, _ in items]), LA(*[wq.shape[2] * wq.shape[3] for wq, _ in items]),
              PA(*[q.data_ptr() for q in scales]) if qps is not None else None, IA(*[4] * n) if qps is not None else None, n, w_bits, _s())
    for (wq, _), o in zip(items, outs):
        wq._mn_packed = o


PACK_PW_MULTI = _os.environ.get("MN_NO_PACKED_PW") is None          # (the library's own A/B knob, csrc/common.h: every conv call packs its own weight codes)


def pack_pointwise_weights(mods_wq, wdesc):
    """The pointwise (1x1, stride 1) counterpart of ``pack_dense_weights``: the forward and backward-data weight-code images of every such conv of the step in ONE
    launch (mn_qg_pack_multi) instead of one 5 us launch per conv call and direction.  ``wdesc`` = (mode, bits, q_type, per_channel, None): ternary / binary or
    DoReFa codes.  The images ride on the quantised weight tensor (``_mn_packed`` = (forward image, backward image)); layers the code kernels do not cover are
    skipped (their calls keep packing for themselves)."""
    if not PACK_PW_MULTI:
        return
    lib = _lib_()
    items = []
    for m, wq in mods_wq:
        if wq.dim() != 4 or not wq.is_cuda or wq.shape[2] != 1 or wq.shape[3] != 1 or getattr(wq, "_mn_packed", None) is not None:
            continue
        one = lambda v: v in (1, (1, 1), [1, 1])
        if not (one(m.stride) and one(m.dilation) and m.padding in (0, (0, 0), [0, 0])):
            continue
        g = _geom((1, wq.shape[1] * m.groups, 8, 8), wq.shape, 1, 0, 1, m.groups, 0)
        nb = [int(lib.mn_qg_packed_bytes(C.byref(g), k)) for k in (0, 1)]
        if nb[0] <= 0 and nb[1] <= 0:
            continue
        items.append((wq, g, nb))
    if not items:
        return
    dev = items[0][0].device
    total = sum(nb[0] + nb[1] for _, _, nb in items)
    buf = torch.empty(total, dtype=torch.uint8, device=dev)
    gs, wds, wps, whs, outs, off = [], [], [], [], [], 0
    for wq, g, nb in items:
        views = [None, None]
        for k in (0, 1):
            if nb[k] > 0:
                views[k] = buf[off:off + nb[k]]
                off += nb[k]
                gs.append(g); wds.append(_wq_desc(wdesc)); wps.append(wq.data_ptr()); whs.append(k); outs.append(views[k].data_ptr())
        wq._mn_packed = (views[0], views[1])
    n = len(gs)
    GP, WP, PA, IA = C.POINTER(ConvGeom) * n, C.POINTER(WQ) * n, C.c_void_p * n, C.c_int32 * n
    with torch.cuda.device(dev):
        _call("mn_qg_pack_multi", n, GP(*[C.pointer(g) for g in gs]), WP(*[C.pointer(w) for w in wds]), PA(*wps), IA(*whs), PA(*outs), _s())


def _code_conv_backward(g, a_bits, w_bits, codes, wq, gy, need_dx, need_dw, packed=None):
    """(dq, dw) of a conv on activation codes: mn_conv2d_bwd_data without clip-STE, mn_conv2d_bwd_weight on the codes."""
    gy = _chk(gy, "grad")
    aq = ActQ(ACTQ_CODE8, a_bits, 0, 0, None)
    wd = _wq_dorefa(w_bits, packed, 1)
    dq = dw = None
    if need_dx:
        dq = torch.empty(codes.shape, dtype=torch.float32, device=codes.device)
        ws, nb = _ws(g, 1, codes.device)
        with _span(g, 1, 4 * (gy.numel() + dq.numel())):
            _call("mn_conv2d_bwd_data", C.byref(g), C.byref(aq), C.byref(wd), _p(gy), _p(wq), None, _p(dq), _p(ws), nb, CONV_ALGO, _s())
    if need_dw:
        dw = torch.empty_like(wq)
        ws, nb = _ws(g, 2, codes.device)
        with _span(g, 2, 4 * gy.numel() + codes.numel() + 4 * dw.numel()):
            _call("mn_conv2d_bwd_weight", C.byref(g), C.byref(aq), _p(gy), _p(codes), _p(dw), None, _p(ws), nb, CONV_ALGO, _s())
    return dq, dw


class QConvCodeLazy(Function):
    """DoReFa QuantConv2d (wqaq/dorefa/quantize.py:107-122) on a ``QActTensor``: the activation codes ARE the quantizer's output, so nothing is
    quantised here, and the result is NOT computed -- a ``LazyQConvOut`` goes to the fused BatchNorm+ReLU+quantizer (``BNReLUQ``).  Backward:
    mn_conv2d_bwd_weight on the codes (dw = s * sum gy * j), mn_conv2d_bwd_data without the clip-STE (returned as a ``QGrad``: the producing block
    applies the STE where it recomputes the activation)."""

    @staticmethod
    def forward(ctx, x, wq, bias, stride, padding, dilation, groups, w_bits, in_shuffle):
        codes, a_bits = x.codes, x.bits
        wq, bias = _chk(wq, "weight"), _chk(bias, "bias")
        g = _geom(codes.shape, wq.shape, stride, padding, dilation, groups, in_shuffle)
        ctx.save_for_backward(codes, wq)
        ctx.cfg = (g, a_bits, w_bits, bias is not None)
        ctx.x_ref = x
        ctx.up_rec = getattr(x, "_mn_up", None)          # the k-bit block that produced x (UpSums, kind 2)
        ctx.packed = packed = getattr(wq, "_mn_packed", None)

        def compute():          # a foreign consumer: the ordinary conv kernels on the materialised activation, quantizer in their prologue
            xa = x.materialize()
            if in_shuffle and in_shuffle > 1:
                xa = channel_shuffle(xa, in_shuffle)
            return QConv2d.apply(xa, wq, bias, stride, padding, dilation, groups, ACTQ_DOREFA, a_bits, 0, None, (WQ_DOREFA, w_bits, 0, 0, None), 0, 0)
        Ho, Wo = _out_hw(g)
        recipe = dict(codes=codes, a_bits=a_bits, wq=wq, bias=bias, geom=g, w_bits=w_bits, compute=compute, out_hw=(Ho, Wo), packed=packed,
                      stash_bits=int(_lib_().mn_qconv_bnq_stash_bits(C.byref(g), C.byref(WQ(WQ_DOREFA, w_bits, 0, 0, None)), a_bits)))
        return LazyQConvOut((g.N, g.O, Ho, Wo), codes.device, recipe)

    @staticmethod
    def backward(ctx, gy):
        codes, wq = ctx.saved_tensors
        g, a_bits, w_bits, has_bias = ctx.cfg
        x = ctx.x_ref
        wd = _wq_dorefa(w_bits, ctx.packed, 1)
        want_w = ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2])
        fused = ctx.needs_input_grad[0] and want_w and FUSE_PW_BWD and codes.data_ptr() % 16 == 0 and \
            bool(_lib_().mn_conv2d_bwd_bnh_supported(C.byref(g), C.byref(wd), 0))
        fold = fused and isinstance(gy, LazyBNGrad) and gy._mn_value is None and gy._mn_recipe.get("kind") == "qa_pw"
        if not fold:
            gy = _chk(gy, "grad")
            fused = fused and gy.data_ptr() % 16 == 0
        if fused:
            # both gradients in one launch (k_pwb, qgemm_pwb.hip); with the block's BatchNorm + ReLU + quantizer backward formed inside from (dq, stash) when the
            # block left a lazy gradient
            with torch.cuda.device_of(codes):
                dq = torch.empty(codes.shape, dtype=torch.float32, device=codes.device)
                dw = torch.empty_like(wq)
                db = torch.empty(g.O, dtype=torch.float32, device=codes.device) if has_bias else None
                nb = int(_lib_().mn_conv2d_bwd_bnh_ws_bytes(C.byref(g)))
                ws = torch.empty(max(nb // 4, 4), dtype=torch.float32, device=codes.device)
                up = getattr(ctx, "up_rec", None)
                splits, part = 0, None
                # (only on the variant that forms its own dy from (dq, stash): there the producer waves set the pace and the sums ride for +18 us where k_qa_partial
                #  takes 25-50; behind a plain dy -- the pooled blocks -- the dx waves set it and the same arithmetic costs +84 us per launch: measured, UP_SUMS_PLAIN)
                if up is not None and up.kind == 2 and UP_SUMS_FOLD and tuple(up.h.shape) == tuple(codes.shape) and up.h.dtype == torch.int16 and up.h.data_ptr() % 16 == 0 \
                        and ((fold and gy._mn_recipe["kind_in"] == 0) or (not fold and UP_SUMS_PLAIN)):
                    splits = int(_lib_().mn_conv2d_bwd_bnh_up_splits(C.byref(g), C.byref(wd), 0, 1))
                    if splits > 0:          # ... and the sums of the BatchNorm backward of the k-bit block in front (this dq is its gradient): UpSums
                        part = torch.empty(codes.shape[1] * splits * 2, dtype=torch.float64, device=codes.device)
                if fold:
                    r = gy._mn_recipe
                    with _span(g, 1, (8 if r["kind_in"] == 2 else 6) * r["dq"].numel() + (7 if part is not None else 5) * dq.numel()):
                        if part is not None:
                            _call("mn_conv2d_bwd_qa_up", C.byref(g), C.byref(wd), _p(r["dq"]), _p(r["stash"]), _p(r["chan"]), _p(r["sums"]), r["bits"], r["quant"], r["training"],
                                  _p(wq), _p(codes), a_bits, _p(dq), _p(dw), _p(db), _p(ws), nb, _p(up.h), _p(up.chan), 1, _p(part), _s())
                        else:
                            _call("mn_conv2d_bwd_qa", C.byref(g), C.byref(wd), _p(r["dq"]), _p(r["stash"]), 32 if r["kind_in"] == 2 else 16, _p(r["chan"]), _p(r["sums"]),
                                  r["bits"], r["quant"], r["training"], _p(wq), _p(codes), a_bits, _p(dq), _p(dw), _p(db), _p(ws), nb, _s())
                else:
                    with _span(g, 1, 4 * gy.numel() + (7 if part is not None else 5) * dq.numel()):
                        if part is not None:
                            _call("mn_conv2d_bwd_codes_up", C.byref(g), C.byref(wd), _p(gy), _p(wq), _p(codes), a_bits, _p(dq), _p(dw), _p(db), _p(ws), nb, _p(up.h), _p(up.chan),
                                  1, _p(part), _s())
                        else:
                            _call("mn_conv2d_bwd_codes", C.byref(g), C.byref(wd), _p(gy), _p(wq), _p(codes), a_bits, _p(dq), _p(dw), _p(db), _p(ws), nb, _s())
                if part is not None:
                    up.ready = (dq, dq._version, part, splits)

            def expand_f(dq_):
                return DorefaAct.backward_raw(dq_, x.materialize(), a_bits)
            ctx.x_ref = None
            return QGrad(dq, expand_f), dw, db, None, None, None, None, None, None
        aq = ActQ(ACTQ_CODE8, a_bits, 0, 0, None)
        dx = dw = db = None
        with torch.cuda.device_of(codes):
            if ctx.needs_input_grad[0]:
                dq = torch.empty(codes.shape, dtype=torch.float32, device=codes.device)
                ws, nb = _ws(g, 1, codes.device)
                _call("mn_conv2d_bwd_data", C.byref(g), C.byref(aq), C.byref(wd), _p(gy), _p(wq), None, _p(dq), _p(ws), nb, CONV_ALGO, _s())

                def expand(dq_):
                    return DorefaAct.backward_raw(dq_, x.materialize(), a_bits)
                dx = QGrad(dq, expand)
            if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
                dw = torch.empty_like(wq)
                db = torch.empty(g.O, dtype=torch.float32, device=codes.device) if has_bias else None
                ws, nb = _ws(g, 2, codes.device)
                _call("mn_conv2d_bwd_weight", C.byref(g), C.byref(aq), _p(gy), _p(codes), _p(dw), _p(db), _p(ws), nb, CONV_ALGO, _s())
        ctx.x_ref = None
        return dx, dw, db, None, None, None, None, None, None


class QConvCodeLazy2(Function):
    """Two DoReFa QuantConv2d reading the SAME ``QActTensor`` (the first conv of a down-sampling residual block and its 1x1 shortcut conv,
    models/resnet.py:24-44): two ``LazyQConvOut``; the backward returns ONE ``QGrad`` carrying both raw gradients -- the producing block applies the
    quantizer's clip-STE to each and adds them, exactly what autograd would do with two separate consumers."""

    @staticmethod
    def forward(ctx, x, wq1, wq2, cfg1, cfg2, w_bits):
        codes, a_bits = x.codes, x.bits
        wq1, wq2 = _chk(wq1, "weight"), _chk(wq2, "weight")
        outs, geoms = [], []
        ctx.packed = (getattr(wq1, "_mn_packed", None), getattr(wq2, "_mn_packed", None))
        for wq, (stride, padding, dilation, groups) in ((wq1, cfg1), (wq2, cfg2)):
            g = _geom(codes.shape, wq.shape, stride, padding, dilation, groups, 0)
            Ho, Wo = _out_hw(g)

            def compute(wq=wq, stride=stride, padding=padding, dilation=dilation, groups=groups):
                return QConv2d.apply(x.materialize(), wq, None, stride, padding, dilation, groups, ACTQ_DOREFA, a_bits, 0, None, (WQ_DOREFA, w_bits, 0, 0, None), 0, 0)
            recipe = dict(codes=codes, a_bits=a_bits, wq=wq, bias=None, geom=g, w_bits=w_bits, compute=compute, out_hw=(Ho, Wo), packed=getattr(wq, "_mn_packed", None),
                          stash_bits=int(_lib_().mn_qconv_bnq_stash_bits(C.byref(g), C.byref(WQ(WQ_DOREFA, w_bits, 0, 0, None)), a_bits)))
            outs.append(LazyQConvOut((g.N, g.O, Ho, Wo), codes.device, recipe))
            geoms.append(g)
        ctx.save_for_backward(codes, wq1, wq2)
        ctx.cfg = (geoms, a_bits, w_bits)
        ctx.x_ref = x
        return outs[0], outs[1]

    @staticmethod
    def backward(ctx, gy1, gy2):
        codes, wq1, wq2 = ctx.saved_tensors
        geoms, a_bits, w_bits = ctx.cfg
        x = ctx.x_ref
        with torch.cuda.device_of(codes):
            dq1, dw1 = _code_conv_backward(geoms[0], a_bits, w_bits, codes, wq1, gy1, ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.packed[0])
            dq2, dw2 = _code_conv_backward(geoms[1], a_bits, w_bits, codes, wq2, gy2, ctx.needs_input_grad[0], ctx.needs_input_grad[2], ctx.packed[1])
        dx = None
        if ctx.needs_input_grad[0]:
            def expand(dq_, dq2_=dq2):
                xa = x.materialize()
                return DorefaAct.backward_raw(dq_, xa, a_bits) + DorefaAct.backward_raw(dq2_, xa, a_bits)
            dx = QGrad(dq1, expand)
            dx._mn_dq2 = dq2
        ctx.x_ref = None
        return dx, dw1, dw2, None, None, None


def materialize(t):
    """The plain float32 tensor behind any lazy / packed wrapper of this package (identity for ordinary tensors); NO autograd link."""
    return t.materialize() if hasattr(t, "materialize") else (t.to_float() if isinstance(t, SignTensor) else t)


class QActToFloat(Function):
    """QActTensor -> the float32 activation it stands for, WITH an autograd link (identity backward: the consumer's gradient is w.r.t. the
    activation itself, so the producing block applies no quantizer STE)."""

    @staticmethod
    def forward(ctx, a):
        return a.materialize()

    @staticmethod
    def backward(ctx, g):
        return g


def qa_supported(shape, pool):
    return len(shape) == 4 and bool(_lib_().mn_qa_supported(shape[2], shape[3], int(bool(pool))))


def _bn_front(y, gamma, beta, running_mean, running_var, eps, momentum, training, nbt, first=None):
    """The statistics half of a fused k-bit block.  ``y`` = a ``LazyQConvOut``: the conv runs HERE on activation codes (mn_qconv_bnq_fwd_stash: 16 / 32-bit stash
    + exact integer statistics; fp32 y never exists); a plain fp32 tensor (the block behind the un-quantised first conv): mn_bn_save_stats.  Returns
    (src, chan [9][C], in_kind 0 int16 / 1 fp32 / 2 int32, (N, C, H, W), device)."""
    lib = _lib_()
    if isinstance(y, LazyQConvOut) and y._mn_value is None:
        r = y.recipe
        g, codes_in, wq = r["geom"], r["codes"], r["wq"]
        H, W = r.get("out_hw", (g.H, g.W))
        N, Cc = g.N, g.O
        dev = codes_in.device
        wide = r.get("stash_bits", 16) == 32
        src = torch.empty((N, Cc, H, W), dtype=torch.int32 if wide else torch.int16, device=dev)
        save = torch.empty((2, Cc), dtype=torch.float32, device=dev)
        chan = torch.empty((9, Cc), dtype=torch.float32, device=dev)
        wd = _wq_dorefa(r["w_bits"], r.get("packed"), 0)
        with torch.cuda.device(dev):
            nb = int(lib.mn_qconv_bnq_ws_bytes(C.byref(g)))
            ws = torch.empty(nb // 4 + 4, dtype=torch.float32, device=dev)
            with _span(g, 0, codes_in.numel() + (4 if wide else 2) * src.numel()):
                _call("mn_qconv_bnq_fwd_stash", C.byref(g), C.byref(wd), _p(codes_in), r["a_bits"], _p(wq), _p(r["bias"]), _p(gamma), _p(beta), float(eps),
                      float(momentum), int(training), _p(running_mean), _p(running_var), _p(nbt), _p(save), _p(src), _p(chan), _p(ws), nb, _s())
        return src, chan, (2 if wide else 0), (N, Cc, H, W), dev
    src = _chk(y, "input")
    N, Cc, H, W = src.shape
    dev = src.device
    chan = torch.empty((9, Cc), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        if first is not None:          # y = the first conv's output: statistics from the Gram data of its input, no pass over y (the Gram data stay on the record holder)
            first[1], save = _first_gram(first[0], eps, momentum, running_mean, running_var)
        else:
            save = torch.empty((2, Cc), dtype=torch.float32, device=dev)
            ws = torch.empty(int(lib.mn_bnsign_ws_floats(Cc)), dtype=torch.float32, device=dev)
            _call("mn_bn_save_stats", _p(src), N, Cc, H * W, float(eps), float(momentum), int(training), _p(running_mean), _p(running_var), _p(save), _p(ws), _s())
        _call("mn_qa_chan_from_save", _p(save), _p(gamma), _p(beta), Cc, _p(chan), _s())
    return src, chan, 1, (N, Cc, H, W), dev


class BNReLUQ(Function):
    """relu(batch_norm(y)) [-> 2x2 max-pool] -> the k-bit activation quantizer of the NEXT QuantConv2d, fused (qact_kernels.hip).
    y is a ``LazyQConvOut`` (the conv runs here, on codes: 16-bit stash + exact integer statistics; fp32 y never exists) or a plain fp32 tensor (the
    block behind the un-quantised first conv).  Output: a ``QActTensor`` (codes of the ``out_bits`` quantizer; ``out_bits`` > 0) or the fp32 activation
    (``out_bits`` == 0: the consumer is not a quantised conv of ours).  Backward: two streaming passes over (gradient, stash)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, eps, momentum, training, nbt, out_bits, pool):
        gamma, beta = _chk(gamma, "weight"), _chk(beta, "bias")
        lazy = isinstance(y, LazyQConvOut) and y._mn_value is None
        rec = _first_record(y, training) if (not lazy and not pool and FOLD_BN_INTO_CONV_BWD) else None
        holder = [rec, None] if rec is not None else None
        src, chan, in_f32, (N, Cc, H, W), dev = _bn_front(y, gamma, beta, running_mean, running_var, eps, momentum, training, nbt, holder)
        Ho, Wo = (H // 2, W // 2) if pool else (H, W)
        qbits = out_bits if out_bits else 2          # the kernels want a valid width even when no code is produced
        ctx.save_for_backward(src, chan, gamma, beta)
        ctx.cfg = (in_f32, N, Cc, H, W, qbits, int(bool(pool)), int(training), bool(out_bits))
        ctx.first, ctx.gram = (rec, holder[1]) if rec is not None else (None, None)
        # the conv in front is a pointwise layer k_pwb covers: its backward forms this block's dy itself from (dq, stash) -- the apply pass and its fp32 dy disappear
        ctx.pwb_fold = bool(lazy and not pool and FUSE_PW_BWD and LAZY_BN_GRAD and _lib_().mn_conv2d_bwd_bnh_supported(
            C.byref(y.recipe["geom"]), C.byref(WQ(WQ_DOREFA, y.recipe["w_bits"], 0, 0, None)), 0))

        def materialize():
            act = torch.empty((N, Cc, Ho, Wo), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _call("mn_qa_fwd", in_f32, _p(src), _p(chan), N, Cc, H, W, qbits, int(bool(pool)), None, _p(act), _s())
            return act
        ctx.mask4 = None
        if not out_bits:
            return materialize()
        codes = torch.empty((N, Cc, Ho, Wo), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            if rec is not None:          # the first block: the pass also leaves the backward's pass nibbles (1 byte per 4 elements) -- the backward then reads them, not y
                ctx.mask4 = torch.empty((N, Cc, (H * W) // 4), dtype=torch.uint8, device=dev)
                _call("mn_qa_fwd_f32_mask", _p(src), _p(chan), N, Cc, H, W, qbits, _p(codes), _p(ctx.mask4), _s())
            else:
                _call("mn_qa_fwd", in_f32, _p(src), _p(chan), N, Cc, H, W, qbits, int(bool(pool)), _p(codes), None, _s())
        out = QActTensor(codes, out_bits, materialize, pooled=bool(pool))
        ctx.up_rec = None
        if lazy and not pool and in_f32 == 0 and UP_SUMS_FOLD:          # (16-bit stash, no pool: the next block's one-launch backward may form this block's sums)
            ctx.up_rec = out._mn_up = UpSums(src, chan, 1, kind=2)
        return out

    @staticmethod
    def backward(ctx, g):
        src, chan, gamma, beta = ctx.saved_tensors
        in_f32, N, Cc, H, W, qbits, pool, training, coded = ctx.cfg
        if isinstance(g, QGrad) and g._mn_value is None and g._mn_dq2 is None and coded:
            dq, quant = g._mn_dq, 1            # gradient w.r.t. the quantised activation: the clip-STE is applied by the kernels below
        else:
            dq, quant = _chk(g, "grad"), 0
        dev = src.device
        if ctx.first is not None:
            # the block behind the un-quantised first conv, one pass over (dq, y): dw, dgamma, dbeta at once (csrc/conv_first.hip, the Gram data of x)
            dq = _chk(dq, "grad")
            if ctx.mask4 is not None:
                dw1, db1, dgamma, dbeta = _first_mask_backward(ctx.first, ctx.gram, ctx.mask4, dq, quant, chan, gamma)
            else:
                dw1, db1, dgamma, dbeta = _first_gram_backward(ctx.first, ctx.gram, src, "qa", dq, None, None, None, chan, qbits, quant)

            def expand1(r):
                dy_ = torch.empty((N, Cc, H, W), dtype=torch.float32, device=dev)
                with torch.cuda.device(dev):
                    sums_ = torch.empty((2, Cc), dtype=torch.float32, device=dev)
                    ws_ = torch.empty(int(_lib_().mn_qa_ws_floats(Cc)), dtype=torch.float32, device=dev)
                    _call("mn_qa_bwd_sums", 1, _p(r["y"]), _p(r["chan"]), _p(r["dq"]), N, Cc, H, W, r["bits"], 0, r["quant"], None, None, _p(sums_), _p(ws_), _s())
                    _call("mn_qa_bwd_apply", 1, _p(r["y"]), _p(r["chan"]), _p(sums_), _p(r["dq"]), N, Cc, H, W, r["bits"], 0, r["quant"], r["training"], _p(dy_), _s())
                return dy_
            recipe = dict(kind="first_done", dq=dq, y=src, chan=chan, bits=qbits, quant=quant, training=training, dw=dw1, db=db1, x=ctx.first.x, w=ctx.first.w)
            return LazyBNGrad((N, Cc, H, W), dev, recipe, expand1), dgamma, dbeta, None, None, None, None, None, None, None, None
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
        sums = torch.empty((2, Cc), dtype=torch.float32, device=dev)
        rec = getattr(ctx, "up_rec", None)
        ready = None
        if rec is not None:
            ready, rec.ready = rec.ready, None
        # the producer of dq (the next block's one-launch backward) already summed dz and dz zhat per channel (UpSums): only the fixed-order finish is left
        presummed = ready is not None and quant == 1 and not pool and type(dq) is torch.Tensor and dq.data_ptr() == ready[0].data_ptr() and \
            tuple(dq.shape) == tuple(ready[0].shape) and dq._version == ready[1] and dq.is_contiguous()
        with torch.cuda.device(dev):
            ws = torch.empty(int(_lib_().mn_qa_ws_floats(Cc)), dtype=torch.float32, device=dev)
            lazy_first = in_f32 == 1 and not pool and LAZY_BN_GRAD and FOLD_BN_INTO_CONV_BWD
            if presummed:
                _call("mn_qa_bwd_sums_final", _p(ready[2]), ready[3], Cc, _p(dgamma), _p(dbeta), _p(sums), _s())
            if getattr(ctx, "pwb_fold", False) and dq.data_ptr() % 16 == 0:
                if not presummed:
                    with _span(None, 3, (2 if in_f32 == 0 else 4) * src.numel() + 4 * dq.numel()):
                        _call("mn_qa_bwd_sums", in_f32, _p(src), _p(chan), _p(dq), N, Cc, H, W, qbits, pool, quant, _p(dgamma), _p(dbeta), _p(sums), _p(ws), _s())

                def expand_pw(r):
                    dy_ = torch.empty((N, Cc, H, W), dtype=torch.float32, device=dev)
                    with torch.cuda.device(dev):
                        _call("mn_qa_bwd_apply", r["kind_in"], _p(r["stash"]), _p(r["chan"]), _p(r["sums"]), _p(r["dq"]), N, Cc, H, W, r["bits"], 0, r["quant"], r["training"],
                              _p(dy_), _s())
                    return dy_
                recipe = dict(kind="qa_pw", dq=dq, stash=src, kind_in=in_f32, chan=chan, sums=sums, bits=qbits, quant=quant, training=training)
                return LazyBNGrad((N, Cc, H, W), dev, recipe, expand_pw), dgamma, dbeta, None, None, None, None, None, None, None, None
            if presummed:
                dy = torch.empty((N, Cc, H, W), dtype=torch.float32, device=dev)
                with _span(None, 3, (2 if in_f32 == 0 else 4) * src.numel() + 4 * dq.numel() + 4 * dy.numel()):
                    _call("mn_qa_bwd_apply", in_f32, _p(src), _p(chan), _p(sums), _p(dq), N, Cc, H, W, qbits, pool, quant, training, _p(dy), _s())
                return dy, dgamma, dbeta, None, None, None, None, None, None, None, None
            if QA_BWD_TWO_LAUNCHES and not lazy_first:
                # partial sums, then the apply pass whose blocks finish the sums themselves (mn_qa_bwd: one launch less per block, bit-identical)
                dy = torch.empty((N, Cc, H, W), dtype=torch.float32, device=dev)
                with _span(None, 3, 2 * ((2 if in_f32 == 0 else 4) * src.numel() + 4 * dq.numel()) + 4 * dy.numel()):
                    _call("mn_qa_bwd", in_f32, _p(src), _p(chan), _p(dq), N, Cc, H, W, qbits, pool, quant, training, _p(dgamma), _p(dbeta), _p(sums), _p(dy), _p(ws), _s())
                return dy, dgamma, dbeta, None, None, None, None, None, None, None, None
            with _span(None, 3, (2 if in_f32 == 0 else 4) * src.numel() + 4 * dq.numel()):
                _call("mn_qa_bwd_sums", in_f32, _p(src), _p(chan), _p(dq), N, Cc, H, W, qbits, pool, quant, _p(dgamma), _p(dbeta), _p(sums), _p(ws), _s())
            if lazy_first:
                # the block behind the un-quantised first conv: d loss / d y has ONE consumer, that conv's backward-weight, which forms it from
                # (dq, y) while they stream in (mn_conv2d_bwd_weight_first_qa) -- dy is neither written nor re-read
                def expand(r):
                    dy_ = torch.empty((N, Cc, H, W), dtype=torch.float32, device=dev)
                    with torch.cuda.device(dev):
                        _call("mn_qa_bwd_apply", 1, _p(r["y"]), _p(r["chan"]), _p(r["sums"]), _p(r["dq"]), N, Cc, H, W, r["bits"], 0, r["quant"], r["training"], _p(dy_), _s())
                    return dy_
                recipe = dict(kind="qa", dq=dq, y=src, chan=chan, sums=sums, bits=qbits, quant=quant, training=training)
                return LazyBNGrad((N, Cc, H, W), dev, recipe, expand), dgamma, dbeta, None, None, None, None, None, None, None, None
            dy = torch.empty((N, Cc, H, W), dtype=torch.float32, device=dev)
            with _span(None, 3, (2 if in_f32 == 0 else 4) * src.numel() + 4 * dq.numel() + 4 * dy.numel()):
                _call("mn_qa_bwd_apply", in_f32, _p(src), _p(chan), _p(sums), _p(dq), N, Cc, H, W, qbits, pool, quant, training, _p(dy), _s())
        return dy, dgamma, dbeta, None, None, None, None, None, None, None, None


class BNAddReLUQ(Function):
    """The END of a residual block (models/resnet.py:60-65 under the k-bit DoReFa scheme), fused (mn_qr_*, qact_kernels.hip):
        u = batch_norm(y) + res,  a = relu(u)  ->  (codes of the next convs' ``out_bits`` quantizer as a ``QActTensor``, fp32 a)
    ``y``: the ``LazyQConvOut`` of the branch's last QuantConv2d (the conv runs here: 16 / 32-bit stash) or a plain fp32 tensor; ``res``: None (no residual:
    a plain block that has to emit codes AND fp32 -- the stem), the block's fp32 input (identity shortcut) or the ``LazyQConvOut`` of the 1x1 shortcut conv,
    whose BatchNorm (``bn_s`` = its gamma, beta, running statistics ...) is evaluated in the same kernels.  Two autograd outputs, so the gradients of the code
    readers (``QGrad``: raw, the clip-STE is applied here) and of the fp32 reader (the next block's identity shortcut) arrive separately and are added in
    the streaming backward pass.  Either output may be switched off (``out_bits`` = 0 / ``want_f32`` = False: an empty tensor is returned in its place)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, eps, momentum, training, nbt, res, gamma_s, beta_s, rm_s, rv_s, eps_s, momentum_s, nbt_s,
                out_bits, want_f32):
        gamma, beta = _chk(gamma, "weight"), _chk(beta, "bias")
        src, chan, in_kind, (N, Cc, H, W), dev = _bn_front(y, gamma, beta, running_mean, running_var, eps, momentum, training, nbt)
        res_kind, rsrc, rchan = 0, None, None
        if isinstance(res, LazyQConvOut) and res._mn_value is None:
            gamma_s, beta_s = _chk(gamma_s, "weight"), _chk(beta_s, "bias")
            rsrc, rchan, rk, rshape, _ = _bn_front(res, gamma_s, beta_s, rm_s, rv_s, eps_s, momentum_s, training, nbt_s)
            if rshape != (N, Cc, H, W):
                raise MicronetHipError("residual block: the shortcut's shape %s differs from the branch's %s" % (rshape, (N, Cc, H, W)))
            res_kind = 3 if rk == 2 else 2
        elif res is not None:
            if gamma_s is not None:
                raise MicronetHipError("residual block: a shortcut BatchNorm needs the shortcut conv's lazy output")
            rsrc = _chk(res, "residual")
            if tuple(rsrc.shape) != (N, Cc, H, W):
                raise MicronetHipError("residual block: the shortcut's shape %s differs from the branch's %s" % (tuple(rsrc.shape), (N, Cc, H, W)))
            res_kind = 1
        qbits = out_bits if out_bits else 2
        codes = torch.empty((N, Cc, H, W), dtype=torch.uint8, device=dev) if out_bits else None
        act = torch.empty((N, Cc, H, W), dtype=torch.float32, device=dev) if (want_f32 or not out_bits) else None
        with torch.cuda.device(dev):
            with _span(None, 3, src.numel() * src.element_size() + (rsrc.numel() * rsrc.element_size() if rsrc is not None else 0) + (N * Cc * H * W) * ((1 if out_bits else 0) + (4 if act is not None else 0))):
                _call("mn_qr_fwd", in_kind, _p(src), _p(chan), res_kind, _p(rsrc), _p(rchan), N, Cc, H, W, qbits, _p(codes), _p(act), _s())
        ctx.save_for_backward(src, chan, rsrc, rchan, gamma, beta, gamma_s if res_kind >= 2 else None, beta_s if res_kind >= 2 else None)
        ctx.cfg = (in_kind, res_kind, N, Cc, H, W, qbits, int(training))

        def materialize():
            a_ = torch.empty((N, Cc, H, W), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _call("mn_qr_fwd", in_kind, _p(src), _p(chan), res_kind, _p(rsrc), _p(rchan), N, Cc, H, W, qbits, None, _p(a_), _s())
            return a_
        q = QActTensor(codes, out_bits, (lambda: act) if act is not None else materialize) if out_bits else torch.empty(0, device=dev)
        a = act if act is not None else torch.empty(0, device=dev)
        ctx.mark_non_differentiable(*([q] if not out_bits else []), *([a] if act is None else []))
        return q, a

    @staticmethod
    def backward(ctx, gq, ga):
        src, chan, rsrc, rchan, gamma, beta, gamma_s, beta_s = ctx.saved_tensors
        in_kind, res_kind, N, Cc, H, W, qbits, training = ctx.cfg
        dev = src.device
        dq = dq2 = gf = None
        if isinstance(gq, QGrad) and gq._mn_value is None:
            dq, dq2 = gq._mn_dq, gq._mn_dq2       # raw gradients w.r.t. the QUANTISED activation: STE applied per consumer by the kernel
        elif gq is not None and gq.numel():
            gf = _chk(gq, "grad")                 # a foreign reader of the codes tensor saw the fp32 activation: its gradient is w.r.t. the activation
        if ga is not None and ga.numel():
            ga = _chk(ga, "grad")
            gf = ga if gf is None else gf + ga
        if dq is None and gf is None:
            gf = torch.zeros((N, Cc, H, W), dtype=torch.float32, device=dev)
        new = lambda: torch.empty(Cc, dtype=torch.float32, device=dev)
        dgamma, dbeta, sums = new(), new(), torch.empty((2, Cc), dtype=torch.float32, device=dev)
        dgamma_s = dbeta_s = sums_s = dy_s = None
        if res_kind >= 2:
            dgamma_s, dbeta_s, sums_s = new(), new(), torch.empty((2, Cc), dtype=torch.float32, device=dev)
            dy_s = torch.empty((N, Cc, H, W), dtype=torch.float32, device=dev)
        du = torch.empty((N, Cc, H, W), dtype=torch.float32, device=dev)
        dy = torch.empty((N, Cc, H, W), dtype=torch.float32, device=dev)
        nel = N * Cc * H * W
        with torch.cuda.device(dev):
            ws = torch.empty(int(_lib_().mn_qr_ws_floats(Cc)), dtype=torch.float32, device=dev)
            if QA_BWD_TWO_LAUNCHES:
                with _span(None, 3, 2 * src.numel() * src.element_size() + 12 * nel):
                    _call("mn_qr_bwd", in_kind, _p(src), _p(chan), res_kind, _p(rsrc), _p(rchan), _p(dq), _p(dq2), _p(gf), N, Cc, H, W, qbits, training, _p(du), _p(dgamma),
                          _p(dbeta), _p(sums), _p(dgamma_s), _p(dbeta_s), _p(sums_s), _p(dy), _p(dy_s), _p(ws), _s())
                dres = du if res_kind == 1 else dy_s
                return (dy, dgamma, dbeta, None, None, None, None, None, None, dres, dgamma_s, dbeta_s, None, None, None, None, None, None, None)
            with _span(None, 3, src.numel() * src.element_size() + (rsrc.numel() * rsrc.element_size() if rsrc is not None else 0) + 4 * nel * (1 + sum(t is not None for t in (dq, dq2, gf)))):
                _call("mn_qr_bwd_sums", in_kind, _p(src), _p(chan), res_kind, _p(rsrc), _p(rchan), _p(dq), _p(dq2), _p(gf), N, Cc, H, W, qbits, _p(du), _p(dgamma), _p(dbeta),
                      _p(sums), _p(dgamma_s), _p(dbeta_s), _p(sums_s), _p(ws), _s())
            with _span(None, 3, src.numel() * src.element_size() + 8 * nel + ((rsrc.numel() * rsrc.element_size() + 4 * nel) if res_kind >= 2 else 0)):
                _call("mn_qr_bwd_apply", in_kind, _p(src), _p(chan), _p(sums), res_kind, _p(rsrc), _p(rchan), _p(sums_s), _p(du), N, Cc, H, W, training, _p(dy), _p(dy_s), _s())
        dres = du if res_kind == 1 else dy_s
        return (dy, dgamma, dbeta, None, None, None, None, None, None, dres, dgamma_s, dbeta_s, None, None, None, None, None, None, None)


def channel_shuffle(x, groups):
    """(N, g*c, H, W) -> interleave the g groups (models/nin_gc.py:4-15), materialised with torch."""
    n, ch, h, w = x.size()
    return x.view(n, groups, ch // groups, h, w).transpose(1, 2).contiguous().view(n, ch, h, w)


def qconv2d(x, wq, bias, stride=1, padding=0, dilation=1, groups=1, aq_mode=ACTQ_NONE, aq_bits=8, aq_qtype=0, qp=None,
            wdesc=None, x_is_code=False, in_shuffle=0, lazy_for_bn=False, want_accstats=False, given=None, donate_dx=False):
    """``in_shuffle`` > 1: the convolution of ``channel_shuffle(x, in_shuffle)``; the permutation is folded into the kernels'
    channel addressing when the code-domain kernels cover all three passes, else materialised."""
    packed = isinstance(x, SignTensor)
    if lazy_for_bn and packed and aq_mode == ACTQ_NONE and qconv_bnsign_supported(x, wq, stride, padding, dilation, groups, wdesc, in_shuffle):
        return QConv2dLazy.apply(x, wq, bias, stride, padding, dilation, groups, wdesc, in_shuffle or 0)
    if packed or (in_shuffle and in_shuffle > 1):
        g = _geom(x.shape, wq.shape, stride, padding, dilation, groups, in_shuffle or 0)
        aq = ActQ(ACTQ_SIGN8 if packed else aq_mode, aq_bits, aq_qtype, 0, qp.data_ptr() if qp is not None else None)
        wd = _wq_desc(wdesc)
        lib = _lib_()
        ok = CONV_ALGO in (_lib.MN_ALGO_AUTO, _lib.MN_ALGO_QGEMM) and all(
            lib.mn_conv2d_qgemm_supported(C.byref(g), C.byref(aq), _ref(wd), k) for k in range(3))
        if not ok:                          # kernels that read neither int8 codes nor shuffled channels: hand them the plain tensor
            x = sign_to_float(x)
            if in_shuffle and in_shuffle > 1:
                x, in_shuffle = channel_shuffle(x, in_shuffle), 0
    y = QConv2d.apply(x, wq, bias, stride, padding, dilation, groups, aq_mode, aq_bits, aq_qtype, qp, wdesc,
                      (_lib.MN_ACTQ_X_IS_CODE if x_is_code else 0) | (WANT_ACCSTATS if want_accstats else 0) | (DONATE_DX if donate_dx else 0), in_shuffle or 0, given)
    tok = getattr(x, "_mn_res_token", None) if aq_mode == ACTQ_IAO else None
    if tok is not None and tok.node is None and y.grad_fn is not None:
        tok.node = weakref.ref(y.grad_fn)          # (the autograd node whose backward-data will consume a parked shortcut gradient)
    return y


class QLinearSmall(Function):
    """F.linear(Q_a(x), wq, bias) for a layer with few outputs (the 512 -> 10 classifier of the ResNets): mn_qlinear_* -- the activation quantizer in registers,
    its clip-STE in the backward-data launch; three launches of a few microseconds instead of the 1x1-conv route's direct VALU kernels."""

    @staticmethod
    def forward(ctx, x, wq, bias, aq_mode, aq_bits, aq_qtype, qp):
        x, wq, bias = _chk(x, "input"), _chk(wq, "weight"), _chk(bias, "bias")
        N, Cc = x.shape
        O = wq.shape[0]
        y = torch.empty((N, O), dtype=torch.float32, device=x.device)
        aq = ActQ(aq_mode, aq_bits, aq_qtype, 0, qp.data_ptr() if qp is not None else None)
        with torch.cuda.device_of(x):
            _call("mn_qlinear_fwd", C.byref(aq), _p(x), _p(wq), _p(bias), _p(y), N, Cc, O, _s())
        ctx.save_for_backward(x, wq, qp)
        ctx.cfg = (aq_mode, aq_bits, aq_qtype, bias is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, wq, qp = ctx.saved_tensors
        aq_mode, aq_bits, aq_qtype, has_bias = ctx.cfg
        gy = _chk(gy, "grad")
        N, Cc = x.shape
        O = wq.shape[0]
        aq = ActQ(aq_mode, aq_bits, aq_qtype, 0, qp.data_ptr() if qp is not None else None)
        dx = dw = db = None
        with torch.cuda.device_of(x):
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                _call("mn_qlinear_bwd_data", C.byref(aq), _p(gy), _p(wq), _p(x), _p(dx), N, Cc, O, _s())
            if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
                dw = torch.empty_like(wq)
                db = torch.empty(O, dtype=torch.float32, device=x.device) if has_bias else None
                _call("mn_qlinear_bwd_weight", C.byref(aq), _p(gy), _p(x), _p(dw), _p(db), N, Cc, O, _s())
        return dx, dw, db, None, None, None, None


def qlinear(x, wq, bias, aq_mode=ACTQ_NONE, aq_bits=8, aq_qtype=0, qp=None, wdesc=None):
    """F.linear as a 1x1 convolution over 1x1 'images' (same kernels, same fused quantizer); few outputs: the dedicated small-linear kernels."""
    if (CONV_ALGO == _lib.MN_ALGO_AUTO and x.dim() == 2 and type(x) is torch.Tensor and x.is_cuda and x.dtype == torch.float32 and
            aq_mode in (ACTQ_NONE, ACTQ_DOREFA, ACTQ_IAO) and _lib_().mn_qlinear_supported(x.shape[0], x.shape[1], wq.shape[0])):
        return QLinearSmall.apply(x, wq, bias, aq_mode, aq_bits, aq_qtype, qp)
    lead = x.shape[:-1]
    x4 = x.reshape(-1, x.shape[-1], 1, 1)
    y = QConv2d.apply(x4, wq.reshape(wq.shape[0], wq.shape[1], 1, 1), bias, 1, 0, 1, 1, aq_mode, aq_bits, aq_qtype, qp, wdesc, 0)
    return y.reshape(*lead, wq.shape[0])


class ConvTranspose2d(Function):
    """conv_transpose2d(x, w) = backward-data of the convolution whose weight is w (OIHW with O = in_channels)."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, padding, output_padding, groups, dilation):
        x, w, bias = _chk(x, "input"), _chk(w, "weight"), _chk(bias, "bias")
        (sh, sw), (ph, pw), (dh, dw_), (oph, opw) = _pair(stride), _pair(padding), _pair(dilation), _pair(output_padding)
        N, Cin, H, W = x.shape
        KH, KW = w.shape[2], w.shape[3]
        Cout = w.shape[1] * groups
        Hout = (H - 1) * sh - 2 * ph + dh * (KH - 1) + oph + 1
        Wout = (W - 1) * sw - 2 * pw + dw_ * (KW - 1) + opw + 1
        g = ConvGeom(N, Cout, Hout, Wout, Cin, KH, KW, sh, sw, ph, pw, dh, dw_, groups)   # the "forward conv": y -> x
        y = torch.empty((N, Cout, Hout, Wout), dtype=torch.float32, device=x.device)
        none = ActQ(ACTQ_NONE, 0, 0, 0, None)
        with torch.cuda.device_of(x):
            ws, nb = _ws(g, 1, x.device)
            _call("mn_conv2d_bwd_data", C.byref(g), C.byref(none), None, _p(x), _p(w), None, _p(y), _p(ws), nb, CONV_ALGO, _s())
        if bias is not None:
            y += bias.view(1, -1, 1, 1)
        ctx.save_for_backward(x, w)
        ctx.cfg = (g, bias is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        g, has_bias = ctx.cfg
        gy = _chk(gy, "grad")
        none = ActQ(ACTQ_NONE, 0, 0, 0, None)
        dx = dw = db = None
        with torch.cuda.device_of(x):
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                ws, nb = _ws(g, 0, x.device)
                _call("mn_conv2d_fwd", C.byref(g), C.byref(none), None, _p(gy), _p(w), None, _p(dx), _p(ws), nb, CONV_ALGO, _s())
            if ctx.needs_input_grad[1]:
                dw = torch.empty_like(w)
                ws, nb = _ws(g, 2, x.device)
                _call("mn_conv2d_bwd_weight", C.byref(g), C.byref(none), _p(x), _p(gy), _p(dw), None, _p(ws), nb, CONV_ALGO, _s())
            if has_bias and ctx.needs_input_grad[2]:
                db = gy.sum(dim=(0, 2, 3))
        return dx, dw, db, None, None, None, None, None


QA_BWD_TWO_LAUNCHES = True          # the k-bit blocks' backward: partial sums + apply (which finishes the sums) instead of partial + final + apply
FIRST_FUSED = True          # the fused first block (round 5 A/B against conv + the BatchNorm block's own kernels: c2 111.9k -> 115.8k img/s)
FIRST_FUSED_QA = _os0.environ.get("MN_FIRST_FUSED_QA", "0") == "1"     # ... for the DoReFa block too (off: its epilogue -- the quantizer's rounding -- makes the fused
#                                                                         forward VALU-bound, 209 us against 110 + 58 us for conv + mn_qa_fwd_f32_mask on nin_gc at batch 256)


class FirstConvLazy(Function):
    """The un-quantised first convolution whose result is NOT computed: a ``LazyConvOut`` (kind "first") carrying the operands.  The BatchNorm block that prepare()
    found behind it runs conv + BatchNorm + activation in ONE kernel (``FirstConvBNSign`` / ``FirstConvBNReLUQ``: the statistics come from the Gram data of the
    image, so the conv's epilogue can normalise) and y -- the largest tensor of the step -- is neither written nor read.  Any other consumer materialises y."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, padding, dilation, groups):
        x, w, bias = _chk(x, "input"), _chk(w, "weight"), _chk(bias, "bias")
        g = _geom(x.shape, w.shape, stride, padding, dilation, groups)
        Ho, Wo = _out_hw(g)
        ctx.save_for_backward(x, w, None, None)
        ctx.cfg = (g, ACTQ_NONE, 8, 0, bias is not None, None, 0)

        def compute():
            y = torch.empty((g.N, g.O, Ho, Wo), dtype=torch.float32, device=x.device)
            aq = ActQ(ACTQ_NONE, 8, 0, 0, None)
            with torch.cuda.device_of(x):
                ws, nb = _ws(g, 0, x.device)
                _call("mn_conv2d_fwd", C.byref(g), C.byref(aq), None, _p(x), _p(w), _p(bias), _p(y), _p(ws), nb, CONV_ALGO, _s())
            return y
        recipe = dict(kind="first", x=x, w=w, bias=bias, geom=g, conv=(stride, padding, dilation, groups), compute=compute)
        return LazyConvOut((g.N, g.O, Ho, Wo), x.device, recipe)

    @staticmethod
    def backward(ctx, gy):
        return QConv2d.backward(ctx, gy)[:7]


class _FirstRec:
    def __init__(self, r):
        self.x, self.w, self.bias, self.conv = r["x"], r["w"], r["bias"], r["conv"]


def _first_fused_backward(ctx, da, quant, expand):
    mask4, gamma, beta, save, x, w, b = ctx.saved_tensors
    g = ctx.geom
    dev = x.device
    with torch.cuda.device_of(x):
        dw = torch.empty_like(w)
        db = torch.empty_like(b) if b is not None else None
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
        ws, nb = _ws(g, 2, dev)
        with _span(g, 2, 4.25 * da.numel() + 4 * x.numel()):
            _call("mn_conv2d_bwd_first_mask_gram", C.byref(g), _p(da), _p(mask4), int(quant), _p(save), _p(gamma), _p(w), _p(b), _p(ctx.gram), _p(x), _p(dw), _p(db),
                  _p(dgamma), _p(dbeta), _p(ws), nb, _s())
    recipe = dict(kind="first_done", dw=dw, db=db, x=x, w=w, da=da, save=save, gamma=gamma, beta=beta, compute=ctx.compute, quant=quant)
    return LazyBNGrad((g.N, g.O, g.H, g.W), dev, recipe, expand), dgamma, dbeta


class FirstConvBNSign(Function):
    """sign(batch_norm(conv(x))) of the first block, training mode, for a ``LazyConvOut`` of kind "first": Gram data of x -> batch statistics -> ONE kernel that
    convolves, normalises and writes int8 sign codes + the backward's pass bits (mn_conv2d_first_bnact_fwd).  Backward: the one-pass first-block backward on
    (da, pass bits) -- dw, dbias, dgamma, dbeta at once (mn_conv2d_bwd_first_mask_gram); the conv node receives the finished dw / dbias."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, eps, momentum):
        r = y.recipe
        x, w, b, g = r["x"], r["w"], r["bias"], r["geom"]
        gamma, beta = _chk(gamma, "weight"), _chk(beta, "bias")
        ctx.gram, save = _first_gram(_FirstRec(r), eps, momentum, running_mean, running_var)
        a = torch.empty((g.N, g.O, g.H, g.W), dtype=torch.int8, device=x.device)
        mask4 = torch.empty((g.N, g.O, (g.H * g.W) // 4), dtype=torch.uint8, device=x.device)
        with torch.cuda.device_of(x):
            with _span(g, 0, 4 * x.numel() + 1.25 * a.numel()):
                _call("mn_conv2d_first_bnact_fwd", C.byref(g), _p(x), _p(w), _p(b), _p(save), _p(gamma), _p(beta), 1, 0, _p(a), _p(mask4), _s())
        ctx.save_for_backward(mask4, gamma, beta, save, x, w, b)
        ctx.geom, ctx.compute = g, r["compute"]
        return SignTensor(a)

    @staticmethod
    def backward(ctx, da):
        da = _chk(da, "grad")

        def expand(r):
            yv = r["compute"]()
            N, Cc, HW = yv.shape[0], yv.shape[1], yv.shape[2] * yv.shape[3]
            dy_ = torch.empty_like(yv)
            with torch.cuda.device_of(dy_):
                ws_ = torch.empty(int(_lib_().mn_bnsign_ws_floats(Cc)), dtype=torch.float32, device=dy_.device)
                _call("mn_bnsign_bwd", _p(r["da"]), _p(yv), _p(r["save"]), _p(r["gamma"]), _p(r["beta"]), N, Cc, HW, 1, _p(dy_), None, None, _p(ws_), _s())
            return dy_
        gyl, dgamma, dbeta = _first_fused_backward(ctx, da, 0, expand)
        return gyl, dgamma, dbeta, None, None, None, None


class FirstConvBNReLUQ(Function):
    """relu(batch_norm(conv(x))) -> the a-bit activation quantizer of the next QuantConv2d, for the first block of a DoReFa net (``LazyConvOut`` of kind "first"):
    as ``FirstConvBNSign`` with uint8 quantizer codes (a ``QActTensor``) and two pass nibbles (ReLU; ReLU and the quantizer's clamp)."""

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, eps, momentum, out_bits):
        r = y.recipe
        x, w, b, g = r["x"], r["w"], r["bias"], r["geom"]
        gamma, beta = _chk(gamma, "weight"), _chk(beta, "bias")
        ctx.gram, save = _first_gram(_FirstRec(r), eps, momentum, running_mean, running_var)
        N, Cc, H, W = g.N, g.O, g.H, g.W
        dev = x.device
        codes = torch.empty((N, Cc, H, W), dtype=torch.uint8, device=dev)
        mask4 = torch.empty((N, Cc, (H * W) // 4), dtype=torch.uint8, device=dev)
        with torch.cuda.device_of(x):
            with _span(g, 0, 4 * x.numel() + 1.25 * codes.numel()):
                _call("mn_conv2d_first_bnact_fwd", C.byref(g), _p(x), _p(w), _p(b), _p(save), _p(gamma), _p(beta), 2, int(out_bits), _p(codes), _p(mask4), _s())
        ctx.save_for_backward(mask4, gamma, beta, save, x, w, b)
        ctx.geom, ctx.compute, ctx.bits = g, r["compute"], int(out_bits)
        compute = r["compute"]

        def chan_of():
            chan = torch.empty((9, Cc), dtype=torch.float32, device=dev)
            _call("mn_qa_chan_from_save", _p(save), _p(gamma), _p(beta), Cc, _p(chan), _s())
            return chan

        def materialize():
            with torch.cuda.device(dev):
                yv, act = compute(), torch.empty((N, Cc, H, W), dtype=torch.float32, device=dev)
                _call("mn_qa_fwd", 1, _p(yv), _p(chan_of()), N, Cc, H, W, int(out_bits), 0, None, _p(act), _s())
            return act
        ctx.chan_of = chan_of
        return QActTensor(codes, int(out_bits), materialize, pooled=False)

    @staticmethod
    def backward(ctx, g_):
        if isinstance(g_, QGrad) and g_._mn_value is None and g_._mn_dq2 is None:
            dq, quant = _chk(g_._mn_dq, "grad"), 1          # gradient w.r.t. the quantised activation: the clip-STE = the high pass nibble and the factor 0.1
        else:
            dq, quant = _chk(g_, "grad"), 0
        bits, chan_of = ctx.bits, ctx.chan_of

        def expand(r):
            yv = r["compute"]()
            N, Cc, H, W = yv.shape
            dy_ = torch.empty_like(yv)
            with torch.cuda.device_of(dy_):
                chan = chan_of()
                sums_ = torch.empty((2, Cc), dtype=torch.float32, device=dy_.device)
                ws_ = torch.empty(int(_lib_().mn_qa_ws_floats(Cc)), dtype=torch.float32, device=dy_.device)
                _call("mn_qa_bwd_sums", 1, _p(yv), _p(chan), _p(r["da"]), N, Cc, H, W, bits, 0, r["quant"], None, None, _p(sums_), _p(ws_), _s())
                _call("mn_qa_bwd_apply", 1, _p(yv), _p(chan), _p(sums_), _p(r["da"]), N, Cc, H, W, bits, 0, r["quant"], 1, _p(dy_), _s())
            return dy_
        gyl, dgamma, dbeta = _first_fused_backward(ctx, dq, quant, expand)
        return gyl, dgamma, dbeta, None, None, None, None, None


def first_conv_supported(x_shape, w_shape, stride, padding, dilation, groups):
    """True when the first-layer kernels (conv_first.hip: real fp32 operands, Cin*KH*KW <= 76) cover forward and backward-weight."""
    if CONV_ALGO != _lib.MN_ALGO_AUTO:
        return False
    g = _geom(x_shape, w_shape, stride, padding, dilation, groups)
    lib = _lib_()
    return bool(lib.mn_conv2d_first_supported(C.byref(g), 0)) and bool(lib.mn_conv2d_first_supported(C.byref(g), 2))


def sign_classifier_supported(x, weight, stride, padding, dilation, groups):
    if not isinstance(x, SignTensor) or x.dim() != 4 or weight.dim() != 4 or CONV_ALGO != _lib.MN_ALGO_AUTO:
        return False
    one = lambda v, k: v in (k, (k, k), [k, k])
    if not (weight.shape[2] == 1 and weight.shape[3] == 1 and one(stride, 1) and one(padding, 0) and one(dilation, 1) and groups == 1):
        return False
    g = _geom(x.shape, weight.shape, 1, 0, 1, 1)
    aq = ActQ(ACTQ_SIGN8, 8, 0, 0, None)
    lib = _lib_()
    return bool(lib.mn_signconv1x1_small_supported(x.shape[1], x.shape[2] * x.shape[3], weight.shape[0])) and \
        bool(lib.mn_conv2d_qgemm_supported(C.byref(g), C.byref(aq), None, 2))


class SignClassifierConv(Function):
    """1x1 convolution with few outputs and full-precision weights on packed sign activations -- the last conv of a WbWtAb net
    (models/nin_gc.py 1024 -> 10; wbwtab/quantize.py:251 leaves it un-quantised): the codes are read directly, nothing is unpacked."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        codes = x.codes
        weight, bias = _chk(weight, "weight"), _chk(bias, "bias")
        N, Cc, H, W = codes.shape
        Oc = weight.shape[0]
        y = torch.empty((N, Oc, H, W), dtype=torch.float32, device=codes.device)
        with torch.cuda.device_of(codes):
            _call("mn_signconv1x1_small_fwd", _p(codes), _p(weight), _p(bias), _p(y), N, Cc, H * W, Oc, _s())
        ctx.save_for_backward(codes, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        codes, weight = ctx.saved_tensors
        gy = _chk(gy, "grad")
        N, Cc, H, W = codes.shape
        Oc = weight.shape[0]
        dx = dw = db = None
        with torch.cuda.device_of(codes):
            if ctx.needs_input_grad[0]:
                dx = torch.empty(codes.shape, dtype=torch.float32, device=codes.device)
                _call("mn_conv1x1_small_bwd_data", _p(gy), _p(weight), _p(dx), N, Cc, H * W, Oc, _s())
            if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
                g = _geom(codes.shape, weight.shape, 1, 0, 1, 1)
                aq = ActQ(ACTQ_SIGN8, 8, 0, 0, None)
                dw = torch.empty_like(weight)
                db = torch.empty(Oc, dtype=torch.float32, device=codes.device) if ctx.has_bias else None
                ws, nb = _ws(g, 2, codes.device)
                _call("mn_conv2d_bwd_weight", C.byref(g), C.byref(aq), _p(gy), _p(codes), _p(dw), _p(db), _p(ws), nb, CONV_ALGO, _s())
        return dx, dw, db


def code_classifier_supported(x, weight, stride, padding, dilation, groups):
    """True when a DoReFa QuantConv2d on a ``QActTensor`` is the small 1x1 classifier the dedicated kernels cover (O <= 16)."""
    if not isinstance(x, QActTensor) or x.dim() != 4 or weight.dim() != 4 or CONV_ALGO != _lib.MN_ALGO_AUTO or not (2 <= x.bits <= 8):
        return False
    one = lambda v, k: v in (k, (k, k), [k, k])
    if not (weight.shape[2] == 1 and weight.shape[3] == 1 and one(stride, 1) and one(padding, 0) and one(dilation, 1) and groups == 1):
        return False
    g = _geom(x.shape, weight.shape, 1, 0, 1, 1)
    aq = ActQ(ACTQ_CODE8, x.bits, 0, 0, None)
    lib = _lib_()
    return bool(lib.mn_signconv1x1_small_supported(x.shape[1], x.shape[2] * x.shape[3], weight.shape[0])) and \
        bool(lib.mn_conv2d_qgemm_supported(C.byref(g), C.byref(aq), None, 2))


class CodeClassifierConv(Function):
    """The classifier conv of a DoReFa net (models/nin_gc.py 1024 -> 10, 1x1; wqaq/dorefa/quantize.py:107-122) on a ``QActTensor``: the byte codes
    are read directly (mn_codeconv1x1_small_fwd); backward-data returns the gradient w.r.t. the QUANTISED activation as a ``QGrad`` (the producing
    block applies the clip-STE), backward-weight contracts gy with the codes (mn_conv2d_bwd_weight, MN_ACTQ_CODE8)."""

    @staticmethod
    def forward(ctx, x, wq, bias):
        codes, a_bits = x.codes, x.bits
        wq, bias = _chk(wq, "weight"), _chk(bias, "bias")
        N, Cc, H, W = codes.shape
        Oc = wq.shape[0]
        y = torch.empty((N, Oc, H, W), dtype=torch.float32, device=codes.device)
        with torch.cuda.device_of(codes):
            _call("mn_codeconv1x1_small_fwd", _p(codes), a_bits, _p(wq), _p(bias), _p(y), N, Cc, H * W, Oc, _s())
        ctx.save_for_backward(codes, wq)
        ctx.cfg = (a_bits, bias is not None)
        ctx.x_ref = x
        return y

    @staticmethod
    def backward(ctx, gy):
        codes, wq = ctx.saved_tensors
        a_bits, has_bias = ctx.cfg
        x = ctx.x_ref
        gy = _chk(gy, "grad")
        N, Cc, H, W = codes.shape
        Oc = wq.shape[0]
        dx = dw = db = None
        with torch.cuda.device_of(codes):
            if ctx.needs_input_grad[0]:
                dq = torch.empty(codes.shape, dtype=torch.float32, device=codes.device)
                _call("mn_conv1x1_small_bwd_data", _p(gy), _p(wq), _p(dq), N, Cc, H * W, Oc, _s())

                def expand(dq_):
                    return DorefaAct.backward_raw(dq_, x.materialize(), a_bits)
                dx = QGrad(dq, expand)
            if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
                g = _geom(codes.shape, wq.shape, 1, 0, 1, 1)
                aq = ActQ(ACTQ_CODE8, a_bits, 0, 0, None)
                dw = torch.empty_like(wq)
                db = torch.empty(Oc, dtype=torch.float32, device=codes.device) if has_bias else None
                ws, nb = _ws(g, 2, codes.device)
                _call("mn_conv2d_bwd_weight", C.byref(g), C.byref(aq), _p(gy), _p(codes), _p(dw), _p(db), _p(ws), nb, CONV_ALGO, _s())
        ctx.x_ref = None
        return dx, dw, db
