"""IAO (integer-arithmetic-only, Jacob et al.) fake-quantised layers on MI355X -- same module surface as the
reference's ``micronet/compression/quantization/wqaq/iao/quantize.py`` (class names, constructor signatures,
buffer names/shapes hence ``state_dict`` keys, ``prepare`` rewrite rules), computed by gfx950 kernels:

  * observers (ref 15-113): min/max by wavefront reductions, the running-extreme / EMA update and
    ``update_qparams`` (ref 293-321) happen ON DEVICE -- no host synchronisation inside ``forward``;
  * ``Quantizer.forward`` (ref 214-240): one fused fake-quant pass; for conv / linear inputs it is not even a pass:
    the scale / round / clamp runs in the prologue of the implicit-GEMM kernel and its clip-STE (ref 163-168) in the
    epilogue of the backward-data kernel;
  * ``QuantBNFuseConv2d`` (ref 652-994): raw conv -> batch mean / unbiased var (kept in the autograd graph) -> fold
    -> per-channel weight quantizer -> quantised conv, all on the same kernels.
Python attributes that the reference keeps off the ``state_dict`` (``num_flag``) are kept the same way.
"""
import copy

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function
from torch.nn import init
from torch.nn.parameter import Parameter

import os

from micronet_amd import ops
from micronet_amd.base_module.op import Add

_FUSE_G3 = os.environ.get("MN_NO_G3") is None          # A/B knob: the grouped 3 x 3 layers on the generic kernels
_PRODUCER_MINMAX = os.environ.get("MN_NO_PRODUCER_MINMAX") is None          # A/B knob: observers read the tensor themselves
_PRODUCER_ACCSTATS = os.environ.get("MN_NO_PRODUCER_ACCSTATS") is None      # A/B knob: the BatchNorm behind a dense conv makes its own statistics pass
_FUSE_BNFUSE = os.environ.get("MN_NO_BNFUSE_BLOCK") is None                 # A/B knob: QuantBNFuseConv2d on the generic kernels (raw conv + statistics passes)
_FUSE_BN_CODES = os.environ.get("MN_IAO_BN_CODES", "1") != "0"              # A/B knob (round 6): BatchNorm + ReLU + the next dense conv's activation codes in one pass (LazyBNAct)
_FUSE_BN_ADD = os.environ.get("MN_IAO_BN_ADD", "1") != "0"                  # A/B knob (round 6): the BatchNorm(s) in front of a residual block's QuantAdd folded into its pass, both directions

__all__ = ["ObserverBase", "MinMaxObserver", "MovingAverageMinMaxObserver", "HistogramObserver", "Round", "Quantizer",
           "SignedQuantizer", "UnsignedQuantizer", "SymmetricQuantizer", "AsymmetricQuantizer", "QuantConv2d",
           "QuantConvTranspose2d", "QuantBNFuseConv2d", "QuantLinear", "QuantReLU", "QuantLeakyReLU", "QuantSigmoid",
           "QuantMaxPool2d", "QuantAvgPool2d", "QuantAdaptiveAvgPool2d", "QuantAdd", "add_quant_op", "prepare",
           "reshape_to_activation", "reshape_to_weight", "reshape_to_bias"]


# ------------------------------------------------------------------------------------------------ observers
def _producer_minmax(t):
    """(partials, count) the producing kernel left on ``t`` (``_mn_minmax``), or None -- also None once the tensor has been written in place since."""
    mm = getattr(t, "_mn_minmax", None)
    if mm is None or not torch.is_tensor(t) or t._version != mm[2]:
        return None
    return mm[0], mm[1]


def _range_shape(q_level, out_channels):
    return {"L": (1,), "C": (out_channels, 1, 1, 1), "FC": (out_channels, 1)}[q_level]


def _synced(obs):
    """the observer reduces the current batch's range over the data-parallel ranks first (micronet_amd.dp.sync_observers) and data parallelism is on"""
    if not getattr(obs, "_mn_sync", False):
        return False
    from micronet_amd import dp
    return dp.active(obs._mn_sync_group)


def _global_ranges(items, group):
    """[(tensor, its producer's partials or None), ...] -> a device buffer [min_0, max_0, min_1, max_1, ...] holding each tensor's range over the GLOBAL batch:
    the local reductions (from the partials where the producer left them: no pass over the tensor), then ONE MAX collective for all of them.  Slices
    ``(buf[2 * i:2 * i + 2], 1)`` have the layout of a one-block partials buffer, so the fused observer kernels consume them unchanged."""
    from micronet_amd import dp
    cur = torch.empty(2 * len(items), dtype=torch.float32, device=items[0][0].device)
    for i, (t, mm) in enumerate(items):
        if mm is not None:
            ops.iao_observe_partials(mm, 0, True, 0.0, cur[2 * i:2 * i + 1], cur[2 * i + 1:2 * i + 2])
        else:
            ops.iao_observe(t, 1, 0, True, 0.0, cur[2 * i:2 * i + 1], cur[2 * i + 1:2 * i + 2])
    dp.allreduce_range(cur, group)
    return cur


class ObserverBase(nn.Module):
    """min/max at level 'L' (whole tensor), 'C' (conv out-channel) or 'FC' (linear row) (ref 15-36)."""
    _kind = None  # 0 running min/max, 1 EMA

    def __init__(self, q_level):
        super().__init__()
        self.q_level = q_level

    def update_range(self, min_val, max_val):
        raise NotImplementedError

    _mn_sync = False          # micronet_amd.dp.sync_observers(): reduce the current batch's range over the data-parallel ranks first
    _mn_sync_group = None

    @torch.no_grad()
    def forward(self, input):
        rows = 1 if self.q_level == "L" else input.shape[0]
        if rows == 1 and _synced(self) and self._kind in (0, 1):
            # local (min, max) of this rank's shard -> global over the ranks -> the ordinary update on the two global extremes
            cur = _global_ranges([(input, _producer_minmax(input))], self._mn_sync_group)
            ops.iao_observe_partials((cur, 1), self._kind, self.num_flag == 0, getattr(self, "momentum", 0.1), self.min_val, self.max_val)
        elif rows == 1 and _producer_minmax(input) is not None and self._kind in (0, 1):
            # the kernel that produced this activation left per-block (min, max): the same update without a pass over the tensor
            ops.iao_observe_partials(_producer_minmax(input), self._kind, self.num_flag == 0, getattr(self, "momentum", 0.1), self.min_val, self.max_val)
        else:
            ops.iao_observe(input, rows, self._kind, self.num_flag == 0, getattr(self, "momentum", 0.1),
                            self.min_val, self.max_val)
        if self.num_flag == 0:
            self.num_flag += 1


class MinMaxObserver(ObserverBase):
    _kind = 0

    def __init__(self, q_level, out_channels):
        super().__init__(q_level)
        self.num_flag = 0
        self.out_channels = out_channels
        shape = _range_shape(q_level, out_channels)
        self.register_buffer("min_val", torch.zeros(shape, dtype=torch.float32))
        self.register_buffer("max_val", torch.zeros(shape, dtype=torch.float32))


class MovingAverageMinMaxObserver(ObserverBase):
    _kind = 1

    def __init__(self, q_level, out_channels, momentum=0.1):
        super().__init__(q_level)
        self.momentum = momentum
        self.num_flag = 0
        self.out_channels = out_channels
        shape = _range_shape(q_level, out_channels)
        self.register_buffer("min_val", torch.zeros(shape, dtype=torch.float32))
        self.register_buffer("max_val", torch.zeros(shape, dtype=torch.float32))


class HistogramObserver(nn.Module):
    """Percentile calibrator of the PTQ mode (ref 116-139): max_val = moving average of the k-th smallest |x|, k = int(percentile * n).
    On the GPU the selection is an exact three-pass radix select over the fp32 bit patterns (``mn_hist_observe``: bit-identical to
    ``torch.kthvalue``, no sort, no host synchronisation -- the first-call / EMA update happens in the last launch)."""

    def __init__(self, q_level, momentum=0.1, percentile=0.9999):
        super().__init__()
        self.q_level = q_level
        self.momentum = momentum
        self.percentile = percentile
        self.num_flag = 0
        self.register_buffer("min_val", torch.zeros((1), dtype=torch.float32))
        self.register_buffer("max_val", torch.zeros((1), dtype=torch.float32))

    @torch.no_grad()
    def forward(self, input):
        ops.hist_observe(input, self.percentile, self.num_flag == 0, self.momentum, self.max_val)
        if self.num_flag == 0:
            self.num_flag += 1


# ------------------------------------------------------------------------------------------------ quantizers
class Round(Function):
    """round-half-away with the clip-STE of ref 144-168 (kept for API parity; the modules use the fused kernels)."""

    @staticmethod
    def forward(self, input, observer_min_val, observer_max_val, q_type):
        if q_type == 0:
            max_val = torch.max(torch.abs(observer_min_val), torch.abs(observer_max_val))
            min_val = -max_val
        else:
            max_val, min_val = observer_max_val, observer_min_val
        self.save_for_backward(input, min_val, max_val)
        return ops.RoundHalfAway.forward(self, input)

    @staticmethod
    def backward(self, grad_output):
        input, min_val, max_val = self.saved_tensors
        grad_input = grad_output.clone()
        grad_input[input.gt(max_val)] = 0
        grad_input[input.lt(min_val)] = 0
        return grad_input, None, None, None


class Quantizer(nn.Module):
    def __init__(self, bits, observer, activation_weight_flag, qaft=False, union=False):
        super().__init__()
        self.bits = bits
        self.observer = observer
        self.activation_weight_flag = activation_weight_flag
        self.qaft = qaft
        self.union = union
        self.q_type = 0
        shape = _range_shape(observer.q_level, getattr(observer, "out_channels", None))
        self.register_buffer("scale", torch.ones(shape, dtype=torch.float32))
        self.register_buffer("zero_point", torch.zeros(shape, dtype=torch.float32))
        self.register_buffer("eps", torch.tensor((torch.finfo(torch.float32).eps), dtype=torch.float32))

    _q_type_static = 0

    def update_qparams(self):
        """scale / zero_point from the observer range, on device (ref 293-305 / 310-321)."""
        self.q_type = self._q_type_static
        return ops.iao_qparams(self.observer.min_val, self.observer.max_val, self.bits, self._q_type_static,
                               self.activation_weight_flag == 1, True, self.scale, self.zero_point)

    def round(self, input, observer_min_val, observer_max_val, q_type):
        return Round.apply(input, observer_min_val, observer_max_val, q_type)

    def qparams(self, input):
        """Observer + qparams bookkeeping of ``forward`` (ref 221-226); returns the device snapshot
        {scale, zero_point, lo, hi} that the fused kernels read, or None for 32 bit."""
        if self.bits == 32:
            return None
        if self.bits == 1:
            print("！Binary quantization is not supported ！")
            assert self.bits != 1
        if not self.qaft and self.training:
            obs = self.observer
            mm = _producer_minmax(input)
            if (not self.union and isinstance(obs, ObserverBase) and obs.q_level == "L" and obs._kind in (0, 1) and 2 <= self.bits <= 24 and _synced(obs)
                    and torch.is_tensor(input) and input.is_cuda):
                mm = (_global_ranges([(input, mm)], obs._mn_sync_group), 1)          # data parallel: the global batch's range as a one-block partials buffer
            if (not self.union and mm is not None and isinstance(obs, ObserverBase) and obs.q_level == "L" and obs._kind in (0, 1) and 2 <= self.bits <= 24):
                # the producing kernel left per-block (min, max): observer update + update_qparams in one launch, no pass over the activation
                self.q_type = self._q_type_static
                qp = ops.iao_observe_partials_qparams(mm, obs._kind, obs.num_flag == 0, getattr(obs, "momentum", 0.1), obs.min_val, obs.max_val, self.bits,
                                                      self._q_type_static, self.activation_weight_flag == 1, self.scale, self.zero_point)
                if obs.num_flag == 0:
                    obs.num_flag += 1
                return qp
            if not self.union:
                self.observer(input)
            return self.update_qparams()
        # the reference sets q_type only inside update_qparams (default 0 until the first training step)
        return ops.iao_qparams(self.observer.min_val, self.observer.max_val, self.bits, self.q_type,
                               self.activation_weight_flag == 1, False, self.scale, self.zero_point)

    def forward(self, input):
        pre = self.__dict__.pop("_mn_pre", None)
        if pre is not None and pre[0] is input:
            # computed ahead for all layers in one launch (micronet_amd.train.prefetch_weight_path -> ops.MultiIaoWeight): observer, qparams and the
            # fake-quantised weights of THIS step; only the python-side bookkeeping of the ordinary path is left
            self.q_type = self._q_type_static
            self._last_qp = pre[2]
            return pre[1]
        qp = self.qparams(input)
        self._last_qp = qp        # snapshot {scale, zp, lo, hi} of THIS call (python attribute, not in the state_dict)
        if qp is None:
            return input
        return ops.IaoFakeQuant.apply(input, qp, self.bits, self.q_type, self.activation_weight_flag == 1)


def _register_range(q, lo, hi):
    q.register_buffer("quant_min_val", torch.tensor(lo, dtype=torch.float32))
    q.register_buffer("quant_max_val", torch.tensor(hi, dtype=torch.float32))


class SignedQuantizer(Quantizer):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if self.activation_weight_flag == 0:
            _register_range(self, -((1 << (self.bits - 1)) - 1), (1 << (self.bits - 1)) - 1)
        elif self.activation_weight_flag == 1:
            _register_range(self, -(1 << (self.bits - 1)), (1 << (self.bits - 1)) - 1)
        else:
            print("activation_weight_flag error")


class UnsignedQuantizer(Quantizer):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if self.activation_weight_flag == 0:
            _register_range(self, 0, (1 << self.bits) - 2)
        elif self.activation_weight_flag == 1:
            _register_range(self, 0, (1 << self.bits) - 1)
        else:
            print("activation_weight_flag error")


class SymmetricQuantizer(SignedQuantizer):
    _q_type_static = 0


class AsymmetricQuantizer(UnsignedQuantizer):
    _q_type_static = 1


def _activation_quantizer(a_bits, q_type, qaft, ptq, percentile, union=False):
    if ptq:
        return SymmetricQuantizer(bits=a_bits, observer=HistogramObserver(q_level="L", percentile=percentile),
                                  activation_weight_flag=1, qaft=qaft, union=union)
    cls = SymmetricQuantizer if q_type == 0 else AsymmetricQuantizer
    return cls(bits=a_bits, observer=MovingAverageMinMaxObserver(q_level="L", out_channels=None),
               activation_weight_flag=1, qaft=qaft, union=union)


def _weight_quantizer(w_bits, q_type, q_level, weight_observer, out_channels, channel_level, qaft, ptq):
    """ref 369-490: per-channel ('C'/'FC') when q_level == 0 else per-layer; MinMax or EMA observer."""
    cls = SymmetricQuantizer if (q_type == 0 or ptq) else AsymmetricQuantizer
    level = channel_level if q_level == 0 else "L"
    obs_cls = MinMaxObserver if weight_observer == 0 else MovingAverageMinMaxObserver
    observer = obs_cls(q_level=level, out_channels=out_channels if level != "L" else None)
    return cls(bits=w_bits, observer=observer, activation_weight_flag=0, qaft=qaft)


def _wdesc(weight_quantizer, quantized):
    """weight-code descriptor for the code-domain conv kernels: w = code * scale[o], scale = column 0 of the qparams snapshot
    the weight quantizer used in this forward (stride 4 floats per channel, 0 for per-layer)."""
    qp = getattr(weight_quantizer, "_last_qp", None)
    if not quantized or qp is None or not (2 <= weight_quantizer.bits <= 8) or weight_quantizer.q_type != 0:
        return None
    return (ops.WQ_IAO, weight_quantizer.bits, 0, 4 if qp.shape[0] > 1 else 0, qp)


def _fused_aq(quantizer, input):
    """(mode, bits, q_type, qp) for the activation quantizer fused into the conv kernels."""
    qp = quantizer.qparams(input)
    if qp is None:
        return ops.ACTQ_NONE, 0, 0, None
    return ops.ACTQ_IAO, quantizer.bits, quantizer.q_type, qp


# ------------------------------------------------------------------------------------------------ conv / linear
class QuantConv2d(nn.Conv2d):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 padding_mode="zeros", a_bits=8, w_bits=8, q_type=0, q_level=0, weight_observer=0,
                 quant_inference=False, qaft=False, ptq=False, percentile=0.9999):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, padding_mode)
        self.quant_inference = quant_inference
        self.activation_quantizer = _activation_quantizer(a_bits, q_type, qaft, ptq, percentile)
        self.weight_quantizer = _weight_quantizer(w_bits, q_type, q_level, weight_observer, out_channels, "C", qaft, ptq)

    donate_dx = False          # set by prepare(): the 1 x 1 shortcut conv of a down-sampling residual block -- the block's first conv reads the same tensor, and this conv's
                               # d x is added in that conv's backward-data store (ops.ResidualToken) instead of by autograd's accumulate kernel
    emit_accstats = False      # set by prepare(): a BatchNorm2dReLU / BatchNorm2dPlain of ours reads this conv's output next -- in training the forward leaves the exact
                               # sums of its integer accumulator (dense layers: mn_actq.stats) on the output tensor, and that BatchNorm needs no statistics pass

    def _pull_codes(self, lazy, quantized):
        """``lazy`` = the un-computed BatchNorm [+ ReLU] output in front of this conv (``LazyBNAct``): its per-channel extrema -> this conv's observer / qparams (the
        ordinary bookkeeping of ``Quantizer.qparams`` on a partials buffer) -> ONE pass over the BatchNorm's input writes this conv's activation codes and clip-STE
        bits.  Returns (codes, mask, qp), or None when this layer / quantizer is not one the dense kernels and the partials path cover."""
        q, wq = self.activation_quantizer, self.weight_quantizer
        obs = q.observer
        if not (quantized and self.training and not q.qaft and not q.union and isinstance(obs, ObserverBase) and obs.q_level == "L" and obs._kind in (0, 1)
                and 2 <= q.bits <= 8 and q._q_type_static == 0 and _wdesc(wq, quantized) is not None and not isinstance(self.padding, str)):
            return None
        nc = ops.iao_codes_bytes(lazy.shape, self.weight.shape, self.stride, self.padding, self.dilation, self.groups, q.bits, wq.bits, q.scale)
        if nc <= 0:
            return None
        mm, count = lazy.prep()
        mm._mn_minmax = (mm, count, mm._version)          # the partials stand for the activation as far as the observer is concerned
        qp = q.qparams(mm)
        if qp is None or qp.shape[0] != 1:
            return None
        codes, mask = ops.iao_bn_apply_codes(lazy, qp, q.bits, nc)
        return codes, mask, qp

    def _qconv(self, input, weight, bias, quantized=True):
        from micronet_amd.sign_tensor import LazyBNAct
        want = bool(self.emit_accstats and self.training and quantized and _PRODUCER_ACCSTATS)
        if isinstance(input, LazyBNAct) and input._mn_value is None:
            pulled = self._pull_codes(input, quantized)
            if pulled is not None:
                codes, mask, qp = pulled
                q = self.activation_quantizer
                out = ops.qconv2d(input, weight, bias, self.stride, self.padding, self.dilation, self.groups, aq_mode=ops.ACTQ_IAO, aq_bits=q.bits, aq_qtype=q.q_type,
                                  qp=qp, wdesc=_wdesc(self.weight_quantizer, quantized), want_accstats=want, given=(codes, mask))
                return self._take_accstats(out, want)
            input = ops.LazyBNActToFloat.apply(input)
        mode, bits, q_type, qp = _fused_aq(self.activation_quantizer, input)
        out = ops.qconv2d(input, weight, bias, self.stride, self.padding, self.dilation, self.groups,
                          aq_mode=mode, aq_bits=bits, aq_qtype=q_type, qp=qp, wdesc=_wdesc(self.weight_quantizer, quantized), want_accstats=want, donate_dx=self.donate_dx)
        return self._take_accstats(out, want)

    @staticmethod
    def _take_accstats(out, want):
        if want:
            st = ops.take_accstats()
            if st is not None and type(out) is torch.Tensor:
                out._mn_accstats = st + (out._version,)          # (valid only while nothing writes into the tensor in place)
        return out

    def forward(self, input):
        # (the reference quantises the input first; the two quantizers are independent, order is immaterial)
        quant_weight = self.weight if self.quant_inference else self.weight_quantizer(self.weight)
        return self._qconv(input, quant_weight, self.bias, quantized=not self.quant_inference)


class QuantConvTranspose2d(nn.ConvTranspose2d):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, output_padding=0, groups=1,
                 bias=True, dilation=1, padding_mode="zeros", a_bits=8, w_bits=8, q_type=0, weight_observer=0,
                 quant_inference=False, qaft=False, ptq=False, percentile=0.9999):
        # positional order as the reference's (iao/quantize.py:511-531: groups, bias, dilation -- nn.ConvTranspose2d's own order; its dorefa / wbwtab
        # siblings declare dilation, groups, bias)
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, output_padding, groups, bias,
                         dilation, padding_mode)
        self.quant_inference = quant_inference
        self.activation_quantizer = _activation_quantizer(a_bits, q_type, qaft, ptq, percentile)
        # conv-transpose weights are quantised per layer only (ref 555-570)
        self.weight_quantizer = _weight_quantizer(w_bits, q_type, 1, weight_observer, None, "C", qaft, ptq)

    def forward(self, input):
        quant_input = self.activation_quantizer(input)
        quant_weight = self.weight if self.quant_inference else self.weight_quantizer(self.weight)
        return ops.ConvTranspose2d.apply(quant_input, quant_weight, self.bias, self.stride, self.padding,
                                         self.output_padding, self.groups, self.dilation)


def reshape_to_activation(input):
    return input.reshape(1, -1, 1, 1)


def reshape_to_weight(input):
    return input.reshape(-1, 1, 1, 1)


def reshape_to_bias(input):
    return input.reshape(-1)


class QuantBNFuseConv2d(QuantConv2d):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=False,
                 padding_mode="zeros", eps=1e-5, momentum=0.1, a_bits=8, w_bits=8, q_type=0, q_level=0,
                 weight_observer=0, pretrained_model=False, qaft=False, ptq=False, percentile=0.9999,
                 bn_fuse_calib=False):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, padding_mode,
                         a_bits=a_bits, w_bits=w_bits, q_type=q_type, q_level=q_level, weight_observer=weight_observer,
                         qaft=qaft, ptq=ptq, percentile=percentile)
        self.num_flag = 0
        self.pretrained_model = pretrained_model
        self.qaft = qaft
        self.bn_fuse_calib = bn_fuse_calib
        self.eps = eps
        self.momentum = momentum
        self.gamma = Parameter(torch.Tensor(out_channels))
        self.beta = Parameter(torch.Tensor(out_channels))
        self.register_buffer("running_mean", torch.zeros((out_channels), dtype=torch.float32))
        self.register_buffer("running_var", torch.ones((out_channels), dtype=torch.float32))
        init.uniform_(self.gamma)
        init.zeros_(self.beta)

    def _fold(self, mean, var_for_bias, var_for_weight):
        if self.weight.is_cuda and self.weight.dtype == torch.float32 and self.weight.is_contiguous():
            wf, bf = ops.IaoBNFold.apply(self.weight, self.bias, self.gamma, self.beta, mean, var_for_bias, var_for_weight, self.eps)
            return wf, reshape_to_bias(bf)
        ops.note_fallback("QuantBNFuseConv2d._fold -> torch ops")
        k_b = self.gamma / torch.sqrt(var_for_bias + self.eps)
        if self.bias is not None:
            bias_fused = reshape_to_bias(self.beta + (self.bias - mean) * k_b)
        else:
            bias_fused = reshape_to_bias(self.beta - mean * k_b)
        weight_fused = self.weight * reshape_to_weight(self.gamma / torch.sqrt(var_for_weight + self.eps))
        return weight_fused, bias_fused

    relu_fused = False          # set by prepare(fuse_blocks=True): the block applies nn.ReLU to this conv's output (ConvBNReLU with bn = Identity) -> fused
    in_shuffle_groups = 0       # > 1: this conv reads channel_shuffle(input, groups) (the block's shuffle folded into the kernels' addressing)

    def _fused_pw_ok(self, input):
        """Training-mode forward on the kernels of csrc/iao_bnfuse.hip: pointwise grouped layer, symmetric <= 8-bit quantizers, per-channel weight observer."""
        return self._fused_quantizers_ok() and ops.iao_bnfuse_pw_supported(input, self.weight, self.stride, self.padding, self.dilation, self.groups, self.in_shuffle_groups)

    def _forward_fused_pw(self, input):
        aq = self.activation_quantizer
        qp = aq.qparams(input)          # observer (from the producer's partials when it left them) + update_qparams: the snapshot {scale, zp, lo, hi}
        aq._last_qp = qp
        relu = bool(self.relu_fused)
        # relu: the result is a LazyReluConvOut -- logically the conv's output (the reference module's contract), physically the rectified tensor the block's
        # ReLUAfterFusedConv takes out of it
        return ops.IaoBNFusePW.apply(input, self.weight, self.bias, self.gamma, self.beta, self, qp, relu, relu and _PRODUCER_MINMAX)

    def _fused_quantizers_ok(self):
        wq, aq = self.weight_quantizer, self.activation_quantizer
        wobs, aobs = wq.observer, aq.observer
        return (_FUSE_BNFUSE and not self.bn_fuse_calib and self.padding_mode == "zeros" and not isinstance(self.padding, str)
                and isinstance(wobs, ObserverBase) and wobs.q_level == "C" and wobs._kind in (0, 1) and not wobs._mn_sync and wq._q_type_static == 0 and 2 <= wq.bits <= 8
                and not wq.qaft and isinstance(aobs, ObserverBase) and aobs.q_level == "L" and aq._q_type_static == 0 and 2 <= aq.bits <= 8 and not aq.qaft and not aq.union)

    def _forward_fused_generic(self, input):
        aq = self.activation_quantizer
        qp = aq.qparams(input)
        aq._last_qp = qp
        relu = bool(self.relu_fused)
        return ops.IaoBNFuseGeneric.apply(input, self.weight, self.bias, self.gamma, self.beta, self, qp, relu, relu and _PRODUCER_MINMAX)

    def _forward_fused_g3(self, input):
        aq = self.activation_quantizer
        qp = aq.qparams(input)          # (the pool in front left its (min, max) partials on the tensor: no pass over it)
        aq._last_qp = qp
        relu = bool(self.relu_fused)
        return ops.IaoBNFuseG3.apply(input, self.weight, self.bias, self.gamma, self.beta, self, qp, relu, relu and _PRODUCER_MINMAX)

    def forward(self, input):
        training_stats = (not self.qaft) and self.training
        if training_stats and self._fused_pw_ok(input):
            return self._forward_fused_pw(input)
        if training_stats and _FUSE_G3 and self._fused_quantizers_ok() and ops.CONV_ALGO == 0 and \
                ops.iao_bnfuse_g3_supported(input, self.weight, self.stride, self.padding, self.dilation, self.groups, self.in_shuffle_groups):
            return self._forward_fused_g3(input)          # (the channel shuffle in front stays folded into the kernels' addressing)
        if self.in_shuffle_groups > 1:
            grid = getattr(input, "_mn_qgrid", None)
            grid = grid if (grid is not None and grid[3] == input._version) else None
            input = ops.channel_shuffle(input, self.in_shuffle_groups)
            if grid is not None:
                input._mn_qgrid = grid[:3] + (input._version,)          # a permutation of channels keeps every value on the grid
        if training_stats and self._fused_quantizers_ok() and ops.iao_bnfuse_generic_supported(input, self.weight) and ops.CONV_ALGO == 0:
            return self._forward_fused_generic(input)
        if training_stats:
            # raw conv for the batch statistics (ref 843-855); the statistics stay in the autograd graph
            output = ops.qconv2d(input, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
            batch_mean, batch_var = ops.BnBatchStats.apply(output)
            with torch.no_grad():
                if not self.pretrained_model and self.num_flag == 0:
                    self.num_flag += 1
                    running_mean, running_var = batch_mean, batch_var
                else:
                    running_mean = (1 - self.momentum) * self.running_mean + self.momentum * batch_mean
                    running_var = (1 - self.momentum) * self.running_var + self.momentum * batch_var
                self.running_mean.copy_(running_mean)
                self.running_var.copy_(running_var)
            weight_fused, bias_fused = self._fold(batch_mean, batch_var,
                                                  self.running_var if self.bn_fuse_calib else batch_var)
        else:
            weight_fused, bias_fused = self._fold(self.running_mean, self.running_var, self.running_var)

        quant_weight = self.weight_quantizer(weight_fused)
        self.__dict__["_mn_last_qw"] = quant_weight.detach()          # (tests: the quantised folded weights of this forward)
        if training_stats and self.bn_fuse_calib:
            # weights folded with the running sigma, output rescaled to the batch sigma (ref 957-972)
            output = self._qconv(input, quant_weight, None)
            output = output * reshape_to_activation(torch.sqrt(self.running_var + self.eps) / torch.sqrt(batch_var + self.eps))
            return output + reshape_to_activation(bias_fused)
        return self._qconv(input, quant_weight, bias_fused)


class QuantLinear(nn.Linear):
    def __init__(self, in_features, out_features, bias=True, a_bits=8, w_bits=8, q_type=0, q_level=0,
                 weight_observer=0, quant_inference=False, qaft=False, ptq=False, percentile=0.9999):
        super().__init__(in_features, out_features, bias)
        self.quant_inference = quant_inference
        self.activation_quantizer = _activation_quantizer(a_bits, q_type, qaft, ptq, percentile)
        self.weight_quantizer = _weight_quantizer(w_bits, q_type, q_level, weight_observer, out_features, "FC", qaft, ptq)

    def forward(self, input):
        quant_weight = self.weight if self.quant_inference else self.weight_quantizer(self.weight)
        mode, bits, q_type, qp = _fused_aq(self.activation_quantizer, input)
        return ops.qlinear(input, quant_weight, self.bias, aq_mode=mode, aq_bits=bits, aq_qtype=q_type, qp=qp,
                           wdesc=_wdesc(self.weight_quantizer, not self.quant_inference))


# ------------------------------------------------------------------------------------------------ quantised non-conv ops
def _fused_act(quantizer, input, act, slope, fallback):
    """act(Q(input)): one fused gfx950 pass (ops.IaoFakeQuantAct) for a per-tensor quantizer on a CUDA fp32 tensor; ``fallback`` (the
    reference's two-step form on our fake-quant kernel) otherwise."""
    if torch.is_tensor(input) and input.is_cuda and input.dtype == torch.float32 and input.numel() > 0:
        qp = quantizer.qparams(input)
        quantizer._last_qp = qp
        if qp is None:
            return fallback(input)
        if qp.shape[0] == 1:
            return ops.IaoFakeQuantAct.apply(input, qp, quantizer.bits, quantizer.q_type, act, slope)
        return fallback(ops.IaoFakeQuant.apply(input, qp, quantizer.bits, quantizer.q_type, quantizer.activation_weight_flag == 1))
    return fallback(quantizer(input))


class QuantReLU(nn.ReLU):
    def __init__(self, inplace=False, a_bits=8, q_type=0, qaft=False, ptq=False, percentile=0.9999):
        super().__init__(inplace)
        self.activation_quantizer = _activation_quantizer(a_bits, q_type, qaft, ptq, percentile)

    def forward(self, input):
        return _fused_act(self.activation_quantizer, input, ops.ACT_RELU, 0.0, lambda q: F.relu(q, self.inplace))


class QuantLeakyReLU(nn.LeakyReLU):
    def __init__(self, negative_slope=0.01, inplace=False, a_bits=8, q_type=0, qaft=False, ptq=False, percentile=0.9999):
        super().__init__(negative_slope, inplace)
        self.activation_quantizer = _activation_quantizer(a_bits, q_type, qaft, ptq, percentile)

    def forward(self, input):
        return _fused_act(self.activation_quantizer, input, ops.ACT_LEAKY, self.negative_slope,
                          lambda q: F.leaky_relu(q, self.negative_slope, self.inplace))


class QuantSigmoid(nn.Sigmoid):
    def __init__(self, a_bits=8, q_type=0, qaft=False, ptq=False, percentile=0.9999):
        super().__init__()
        self.activation_quantizer = _activation_quantizer(a_bits, q_type, qaft, ptq, percentile)

    def forward(self, input):
        return _fused_act(self.activation_quantizer, input, ops.ACT_SIGMOID, 0.0, torch.sigmoid)


class QuantMaxPool2d(nn.MaxPool2d):
    def __init__(self, kernel_size, stride=None, padding=0, dilation=1, return_indices=False, ceil_mode=False, a_bits=8,
                 q_type=0, qaft=False, ptq=False, percentile=0.9999):
        super().__init__(kernel_size, stride, padding, dilation, return_indices, ceil_mode)
        self.activation_quantizer = _activation_quantizer(a_bits, q_type, qaft, ptq, percentile)

    def forward(self, input):
        aq = self.activation_quantizer
        if (_FUSE_BNFUSE and not self.return_indices and aq.bits != 32 and 2 <= aq.bits <= 24 and isinstance(aq.observer, (ObserverBase, HistogramObserver))
                and getattr(aq.observer, "q_level", "L") == "L" and ops.iao_fq_maxpool_supported(input, self.kernel_size, self.stride, self.padding, self.dilation, self.ceil_mode)):
            # quantizer + 2 x 2 max-pool in one pass (and one in backward); leaves (min, max) partials of its output for the next layer's observer
            qp = aq.qparams(input)
            aq._last_qp = qp
            if qp is not None and qp.shape[0] == 1:
                want_mm = self.training and _PRODUCER_MINMAX
                out = ops.IaoFakeQuantMaxPool2x2.apply(input, qp, aq.bits, aq.q_type, want_mm, self)
                mm = self.__dict__.pop("_mn_fwd_out", None)
                if mm is not None:
                    out._mn_minmax = mm + (out._version,)
                grid = self.__dict__.pop("_mn_fwd_grid", None)
                if grid is not None:
                    out._mn_qgrid = grid + (out._version,)          # (qp, bits, q_type, version): every value is code * scale of this quantizer
                return out
            q = input if qp is None else ops.IaoFakeQuant.apply(input, qp, aq.bits, aq.q_type, True)
        else:
            q = aq(input)
        if not self.return_indices and ops.f32_pool_supported(q, self.kernel_size, self.stride, self.padding, self.dilation, self.ceil_mode):
            return ops.MaxPool2x2F32.apply(q)        # 2x2 / stride 2: byte argmax, scatter backward (same values and gradient routing as ATen)
        ops.note_fallback("QuantMaxPool2d -> F.max_pool2d")
        return F.max_pool2d(q, self.kernel_size, self.stride, self.padding, self.dilation, self.ceil_mode, self.return_indices)


class QuantAvgPool2d(nn.AvgPool2d):
    def __init__(self, kernel_size, stride=None, padding=0, ceil_mode=False, count_include_pad=True,
                 divisor_override=None, a_bits=8, q_type=0, qaft=False, ptq=False, percentile=0.9999):
        super().__init__(kernel_size, stride, padding, ceil_mode, count_include_pad, divisor_override)
        self.activation_quantizer = _activation_quantizer(a_bits, q_type, qaft, ptq, percentile)

    def forward(self, input):
        k = self.kernel_size if isinstance(self.kernel_size, int) else (self.kernel_size[0] if self.kernel_size[0] == self.kernel_size[1] else 0)
        st = self.stride if self.stride is not None else self.kernel_size
        whole = torch.is_tensor(input) and input.dim() == 4 and k and input.shape[2] == k and input.shape[3] == k          # one window = the image: the stride is moot
        if k and (st in (k, (k, k)) or whole) and self.padding in (0, (0, 0)) and not self.ceil_mode and self.divisor_override is None and ops.iao_avgpool_supported(input, k):
            q = self.activation_quantizer
            qp = q.qparams(input)
            if qp is not None and qp.shape[0] == 1:
                return ops.IaoFakeQuantAvgPool.apply(input, qp, q.bits, q.q_type, k)     # the quantised tensor is never written
            return F.avg_pool2d(input if qp is None else ops.IaoFakeQuant.apply(input, qp, q.bits, q.q_type, True), self.kernel_size, self.stride,
                                self.padding, self.ceil_mode, self.count_include_pad, self.divisor_override)
        ops.note_fallback("QuantAvgPool2d -> F.avg_pool2d")
        return F.avg_pool2d(self.activation_quantizer(input), self.kernel_size, self.stride, self.padding, self.ceil_mode,
                            self.count_include_pad, self.divisor_override)


class QuantAdaptiveAvgPool2d(nn.AdaptiveAvgPool2d):
    def __init__(self, output_size, a_bits=8, q_type=0, qaft=False, ptq=False, percentile=0.9999):
        super().__init__(output_size)
        self.activation_quantizer = _activation_quantizer(a_bits, q_type, qaft, ptq, percentile)

    def forward(self, input):
        if self.output_size in (1, (1, 1)) and input.dim() == 4 and input.shape[2] == input.shape[3] and ops.iao_avgpool_supported(input, input.shape[2]):
            q = self.activation_quantizer
            qp = q.qparams(input)
            if qp is not None and qp.shape[0] == 1:
                return ops.IaoFakeQuantAvgPool.apply(input, qp, q.bits, q.q_type, int(input.shape[2]))
            return F.adaptive_avg_pool2d(input if qp is None else ops.IaoFakeQuant.apply(input, qp, q.bits, q.q_type, True), self.output_size)
        ops.note_fallback("QuantAdaptiveAvgPool2d -> F.adaptive_avg_pool2d")
        return F.adaptive_avg_pool2d(self.activation_quantizer(input), self.output_size)


class QuantAdd(nn.Module):
    def __init__(self, a_bits=8, q_type=0, qaft=False, ptq=False, percentile=0.9999):
        super().__init__()
        if not ptq:
            self.observer_res = MovingAverageMinMaxObserver(q_level="L", out_channels=None)
            self.observer_shortcut = MovingAverageMinMaxObserver(q_level="L", out_channels=None)
        else:
            self.observer_res = HistogramObserver(q_level="L", percentile=percentile)
            self.observer_shortcut = HistogramObserver(q_level="L", percentile=percentile)
        self.activation_quantizer = _activation_quantizer(a_bits, q_type, qaft, ptq, percentile, union=True)

    def forward(self, res, shortcut, relu=False):
        """``relu`` (ours): also apply the ReLU the ResNet block puts on the sum (models/resnet.py:63) -- in the same pass on the fused path."""
        if relu:
            out = self._forward(res, shortcut, True)
            return out if getattr(out, "_mn_relu_done", False) else F.relu(out)
        return self._forward(res, shortcut, False)

    def _forward(self, res, shortcut, relu):
        from micronet_amd.sign_tensor import LazyBNAct
        q = self.activation_quantizer
        obs_r, obs_s = self.observer_res, self.observer_shortcut
        lazy_r = isinstance(res, LazyBNAct) and res._mn_value is None
        lazy_s = isinstance(shortcut, LazyBNAct) and shortcut._mn_value is None
        fused = (lazy_r and self.training and not q.qaft and torch.is_tensor(shortcut) and tuple(res.shape) == tuple(shortcut.shape) and 2 <= q.bits <= 24
                 and type(obs_r) is type(obs_s) and getattr(obs_r, "_kind", None) in (0, 1) and obs_r.q_level == "L" and obs_s.q_level == "L"
                 and getattr(q.observer, "q_level", None) == "L" and hasattr(q.observer, "min_val") and _synced(obs_r) == _synced(obs_s)
                 and getattr(obs_r, "momentum", 0.1) == getattr(obs_s, "momentum", 0.1)
                 and (lazy_s or (type(shortcut) is torch.Tensor and shortcut.is_cuda and shortcut.dtype == torch.float32 and shortcut.is_contiguous()
                                 and shortcut.data_ptr() % 16 == 0 and ops._valid_minmax(shortcut) is not None)))
        if fused:
            # the BatchNorm(s) in front stayed un-computed (LazyBNAct): their per-channel extrema are the two input observers' partials, then ONE pass normalises,
            # quantises, adds [and rectifies] straight from the convs' outputs
            q.q_type = q._q_type_static
            pr = res.prep()
            ps = shortcut.prep() if lazy_s else ops._valid_minmax(shortcut)
            if _synced(obs_r):
                cur = _global_ranges([(res, pr), (shortcut, ps)], obs_r._mn_sync_group)
                pr, ps = (cur[0:2], 1), (cur[2:4], 1)
            qp = ops.iao_qadd_observe_partials(pr, ps, obs_r, obs_s, q, True)
            for o in (obs_r, obs_s):
                if o.num_flag == 0:
                    o.num_flag += 1
            q._last_qp = qp
            want_mm = bool(relu) and _PRODUCER_MINMAX
            tok = None if lazy_s else getattr(shortcut, "_mn_res_token", None)
            node = tok.node() if (tok is not None and tok.node is not None) else None
            if tok is not None and (tok.claimed or node is None or not torch.is_grad_enabled() or not shortcut.requires_grad or not res.requires_grad
                                    or not ops._descends_from(res, node)):
                tok = None
            if tok is not None:
                tok.claimed = True
            out = ops.IaoQuantAddBN.apply(res, shortcut, qp, q.bits, q.q_type, bool(relu), want_mm, tok)
            if relu:
                out._mn_relu_done = True
            if want_mm:
                mm = ops.take_minmax()
                if mm is not None:
                    out._mn_minmax = mm + (out._version,)
            return out
        if lazy_r:
            res = ops.LazyBNActToFloat.apply(res)
        if lazy_s:
            shortcut = ops.LazyBNActToFloat.apply(shortcut)
        if (torch.is_tensor(res) and torch.is_tensor(shortcut) and res.is_cuda and shortcut.is_cuda and res.dtype == torch.float32 and shortcut.dtype == torch.float32
                and res.shape == shortcut.shape and res.is_contiguous() and shortcut.is_contiguous() and res.numel() % 4 == 0 and res.numel() > 0
                and 2 <= q.bits <= 24 and type(obs_r) is type(obs_s) and getattr(obs_r, "_kind", None) in (0, 1) and obs_r.q_level == "L" and obs_s.q_level == "L"
                and getattr(q.observer, "q_level", None) == "L" and hasattr(q.observer, "min_val") and _synced(obs_r) == _synced(obs_s)
                and getattr(obs_r, "momentum", 0.1) == getattr(obs_s, "momentum", 0.1)):
            # the same bookkeeping and arithmetic in three launches instead of nine (+ one instead of two in backward)
            update = (not q.qaft) and q.training
            if update:
                q.q_type = q._q_type_static
            pr, ps = (ops._valid_minmax(res), ops._valid_minmax(shortcut)) if (_PRODUCER_MINMAX and self.training) else (None, None)
            if _synced(obs_r):                             # data parallel: both inputs' ranges over the global batch, one collective for the two
                cur = _global_ranges([(res, pr), (shortcut, ps)], obs_r._mn_sync_group)
                pr, ps = (cur[0:2], 1), (cur[2:4], 1)
            if pr is not None and ps is not None:          # both producers left (min, max) partials: the two input observers need no pass over the tensors
                qp = ops.iao_qadd_observe_partials(pr, ps, obs_r, obs_s, q, update)
            else:
                qp = ops.iao_qadd_observe(res, shortcut, obs_r, obs_s, q, update)
            for o in (obs_r, obs_s):
                if o.num_flag == 0:
                    o.num_flag += 1
            q._last_qp = qp
            want_mm = bool(relu) and self.training and _PRODUCER_MINMAX          # (a ResNet block's output: the next block's convs observe it)
            tok = getattr(shortcut, "_mn_res_token", None)
            node = tok.node() if (tok is not None and tok.node is not None) else None
            if tok is not None and (tok.claimed or node is None or not torch.is_grad_enabled() or not shortcut.requires_grad or not res.requires_grad
                                    or not ops._descends_from(res, node)):
                tok = None          # (only an identity shortcut into the conv that res descends from, and only one QuantAdd per token)
            if tok is not None:
                tok.claimed = True
            out = ops.IaoQuantAdd.apply(res, shortcut, qp, q.bits, q.q_type, bool(relu), want_mm, tok)
            if relu:
                out._mn_relu_done = True
            if want_mm:
                mm = ops.take_minmax()
                if mm is not None:
                    out._mn_minmax = mm + (out._version,)          # (valid only while nothing writes into the tensor in place)
            return out
        # both observers run unconditionally, also in eval (ref 1485-1486); the union range feeds ONE shared quantizer
        self.observer_res(res)
        self.observer_shortcut(shortcut)
        obs = self.activation_quantizer.observer
        ops.iao_union_range(self.observer_res.min_val, self.observer_res.max_val, self.observer_shortcut.min_val,
                            self.observer_shortcut.max_val, obs.min_val, obs.max_val)
        q = self.activation_quantizer
        qp = q.qparams(res)
        if qp is None:
            return res + shortcut
        is_act = q.activation_weight_flag == 1
        return ops.IaoFakeQuant.apply(res, qp, q.bits, q.q_type, is_act) + ops.IaoFakeQuant.apply(shortcut, qp, q.bits, q.q_type, is_act)


# ------------------------------------------------------------------------------------------------ graph rewrite
def _copy_params(new, child):
    if child.bias is not None:
        new.bias.data = child.bias
    new.weight.data = child.weight


def add_quant_op(module, a_bits=8, w_bits=8, q_type=0, q_level=0, weight_observer=0, bn_fuse=False, bn_fuse_calib=False,
                 quant_inference=False, pretrained_model=False, qaft=False, ptq=False, percentile=0.9999, fuse_bn_act=True):
    """ref 1501-1788: every conv / linear is quantised (no first/last skip); with ``bn_fuse`` a conv is replaced when
    the BatchNorm2d that follows it among the same parent's children is met, and that BN becomes ``nn.Identity``."""
    common = dict(a_bits=a_bits, q_type=q_type, qaft=qaft, ptq=ptq, percentile=percentile)
    kw_all = dict(a_bits=a_bits, w_bits=w_bits, q_type=q_type, q_level=q_level, weight_observer=weight_observer,
                  bn_fuse=bn_fuse, bn_fuse_calib=bn_fuse_calib, quant_inference=quant_inference,
                  pretrained_model=pretrained_model, qaft=qaft, ptq=ptq, percentile=percentile, fuse_bn_act=fuse_bn_act)
    conv_name_temp = conv_child_temp = None
    prev = None
    for name, child in module.named_children():
        if (fuse_bn_act and not bn_fuse and type(child) is nn.ReLU and type(prev) is nn.BatchNorm2d and prev.affine and prev.track_running_stats
                and isinstance(module, nn.Sequential)):
            # ours (numerically the same function): BatchNorm2d directly in front of a ReLU the Sequential calls right after it -> one fused gfx950 op
            # (ops.BNReLU) instead of MIOpen's BatchNorm kernels + ReLU forward / backward; the ReLU stays in place as a no-op subclass.  Same objects,
            # same parameters / buffers / state_dict keys, isinstance contracts intact (the reference leaves nn.ReLU alone: ref 1705-1709).
            from micronet_amd.quantization.wqaq.dorefa.quantize import BatchNorm2dReLU, ReLUAfterFusedBN
            prev.__class__ = BatchNorm2dReLU
            prev.emit_minmax = _PRODUCER_MINMAX          # the IAO conv behind it observes this activation: the fused op hands over per-block (min, max)
            child.__class__ = ReLUAfterFusedBN
            prev = child
            continue
        prev = child
        if isinstance(child, nn.Conv2d):
            if bn_fuse:
                conv_name_temp, conv_child_temp = name, child
            else:
                new = QuantConv2d(child.in_channels, child.out_channels, child.kernel_size, stride=child.stride,
                                  padding=child.padding, dilation=child.dilation, groups=child.groups,
                                  bias=child.bias is not None, padding_mode=child.padding_mode, w_bits=w_bits,
                                  q_level=q_level, weight_observer=weight_observer, quant_inference=quant_inference,
                                  **common)
                _copy_params(new, child)
                module._modules[name] = new
        elif isinstance(child, nn.BatchNorm2d):
            if bn_fuse:
                c = conv_child_temp
                new = QuantBNFuseConv2d(c.in_channels, c.out_channels, c.kernel_size, stride=c.stride, padding=c.padding,
                                        dilation=c.dilation, groups=c.groups, bias=c.bias is not None,
                                        padding_mode=c.padding_mode, eps=child.eps, momentum=child.momentum,
                                        w_bits=w_bits, q_level=q_level, weight_observer=weight_observer,
                                        pretrained_model=pretrained_model, bn_fuse_calib=bn_fuse_calib, **common)
                _copy_params(new, c)
                new.gamma.data = child.weight
                new.beta.data = child.bias
                new.running_mean.copy_(child.running_mean)
                new.running_var.copy_(child.running_var)
                module._modules[conv_name_temp] = new
                module._modules[name] = nn.Identity()
        elif isinstance(child, nn.ConvTranspose2d):
            new = QuantConvTranspose2d(child.in_channels, child.out_channels, child.kernel_size, stride=child.stride,
                                       padding=child.padding, output_padding=child.output_padding, groups=child.groups,
                                       bias=child.bias is not None, dilation=child.dilation,
                                       padding_mode=child.padding_mode, w_bits=w_bits, weight_observer=weight_observer,
                                       quant_inference=quant_inference, **common)
            _copy_params(new, child)
            module._modules[name] = new
        elif isinstance(child, nn.Linear):
            new = QuantLinear(child.in_features, child.out_features, bias=child.bias is not None, w_bits=w_bits,
                              q_level=q_level, weight_observer=weight_observer, quant_inference=quant_inference, **common)
            _copy_params(new, child)
            module._modules[name] = new
        # nn.ReLU is deliberately left alone: it is fused at inference (ref 1705-1709)
        elif isinstance(child, nn.LeakyReLU):
            module._modules[name] = QuantLeakyReLU(negative_slope=child.negative_slope, inplace=child.inplace, **common)
        elif isinstance(child, nn.Sigmoid):
            module._modules[name] = QuantSigmoid(**common)
        elif isinstance(child, nn.MaxPool2d):
            module._modules[name] = QuantMaxPool2d(kernel_size=child.kernel_size, stride=child.stride,
                                                   padding=child.padding, **common)
        elif isinstance(child, nn.AvgPool2d):
            module._modules[name] = QuantAvgPool2d(kernel_size=child.kernel_size, stride=child.stride,
                                                   padding=child.padding, **common)
        elif isinstance(child, nn.AdaptiveAvgPool2d):
            module._modules[name] = QuantAdaptiveAvgPool2d(output_size=child.output_size, **common)
        elif isinstance(child, Add):
            module._modules[name] = QuantAdd(**common)
        else:
            add_quant_op(child, **kw_all)


class _ResidualAddReLUMixin:
    """forward of the reference's residual blocks (models/resnet.py:60-65: ``relu(add(residual_function(x), shortcut(x)))``) with the trailing ReLU folded into
    the QuantAdd pass; installed by ``prepare(fuse_bn_act=True)`` as a subclass of the block's own class (same children, parameters, ``state_dict``)."""

    def forward(self, x):
        return self.add(self.residual_function(x), self.shortcut(x), relu=True)


def _fuse_residual_tails(model):
    from micronet_amd.quantization.wqaq.dorefa.quantize import BatchNorm2dPlain
    for m in model.modules():
        # the BatchNorms that are NOT followed by a ReLU inside a Sequential (the last one of a residual function, the shortcut's): our streaming kernels
        if isinstance(m, nn.Sequential):
            for child in m.children():
                if type(child) is nn.BatchNorm2d and child.affine and child.track_running_stats:
                    child.__class__ = BatchNorm2dPlain
                    child.emit_minmax = _PRODUCER_MINMAX          # (an IAO QuantAdd observes this output: its two input observers then read partials only)
    # conv -> BatchNorm (ours) adjacency inside a Sequential: the conv's forward hands the exact sums of its integer accumulator to that BatchNorm (dense layers)
    from micronet_amd.quantization.wqaq.dorefa.quantize import BatchNorm2dReLU
    for m in model.modules():
        if isinstance(m, nn.Sequential):
            kids = list(m.children())
            for a_, b_ in zip(kids, kids[1:]):
                if type(a_) is QuantConv2d and isinstance(b_, (BatchNorm2dReLU, BatchNorm2dPlain)) and b_.affine and b_.track_running_stats:
                    a_.emit_accstats = True
            # conv -> BatchNorm2dReLU -> (its no-op ReLU) -> conv: the activation between the two convs has ONE consumer, the second conv's quantizer -- it stays
            # un-computed and that conv pulls its codes from the first conv's output in one pass (LazyBNAct)
            from micronet_amd.quantization.wqaq.dorefa.quantize import ReLUAfterFusedBN
            for a_, b_, r_, c_ in zip(kids, kids[1:], kids[2:], kids[3:]):
                if _FUSE_BN_CODES and type(a_) is QuantConv2d and a_.emit_accstats and type(b_) is BatchNorm2dReLU and type(r_) is ReLUAfterFusedBN and type(c_) is QuantConv2d \
                        and b_.momentum is not None:
                    b_.iao_lazy_out = True
    for m in model.modules():
        t = type(m)
        if t.__name__ in ("BasicBlock", "BottleNeck") and t.__module__.split(".")[-1] == "resnet" and isinstance(getattr(m, "add", None), QuantAdd) \
                and isinstance(getattr(m, "residual_function", None), nn.Sequential) and isinstance(getattr(m, "shortcut", None), nn.Sequential):
            from micronet_amd.nn import derive_class
            m.__class__ = derive_class("AddReLU", _ResidualAddReLUMixin, t)
            # the BatchNorms whose only consumer is this block's QuantAdd (the last module of the residual function, of the shortcut) stay un-computed behind a dense
            # conv: the QuantAdd normalises, quantises and adds in one pass (LazyBNAct -> ops.IaoQuantAddBN)
            sc0, rf0 = next(iter(m.shortcut.children()), None), next(iter(m.residual_function.children()), None)
            if type(sc0) is QuantConv2d and type(rf0) is QuantConv2d:
                sc0.donate_dx = True
            if _FUSE_BN_ADD:
                for seq in (m.residual_function, m.shortcut):
                    kids = list(seq.children())
                    if len(kids) >= 2 and type(kids[-1]) is BatchNorm2dPlain and type(kids[-2]) is QuantConv2d and kids[-2].emit_accstats and kids[-1].momentum is not None:
                        kids[-1].iao_lazy_out = True


class ReLUAfterFusedConv(nn.ReLU):
    """The ``nn.ReLU`` of a ``ConvBNReLU`` block whose conv is a BN-fused ``QuantBNFuseConv2d``: when the conv handed over a ``LazyReluConvOut`` (its kernel already
    wrote relu(out)) this takes the rectified tensor out of the wrapper -- no kernel -- else the ordinary ReLU.  Same module object, name and ``isinstance``."""

    def forward(self, input):
        from micronet_amd.sign_tensor import LazyReluConvOut
        if isinstance(input, LazyReluConvOut) and input._mn_value is None:
            return ops.relu_of_fused(input)
        return super().forward(input)


def _fuse_bnfuse_blocks(model):
    """``prepare(bn_fuse=True, fuse_blocks=True)``: in every block KNOWN to run shuffle -> conv -> bn -> relu (the reference's ``ConvBNReLU``, models/nin_gc.py:18-59,
    or a block that declares ``_mn_ordered_forward``) whose conv became a ``QuantBNFuseConv2d`` and whose bn became ``nn.Identity``: the block's ReLU moves into the
    conv's epilogue (``relu_fused``), the channel shuffle into the conv's addressing (``in_shuffle_groups``).  Same objects, parameters, buffers, ``state_dict``."""
    from micronet_amd.quantization.wqaq.dorefa.quantize import _is_ref_block
    for blk in model.modules():
        if not _is_ref_block(blk):
            continue
        conv, bn, relu = getattr(blk, "conv", None), getattr(blk, "bn", None), getattr(blk, "relu", None)
        if not (type(conv) is QuantBNFuseConv2d and type(bn) is nn.Identity):
            continue
        if type(relu) is nn.ReLU:
            relu.__class__ = ReLUAfterFusedConv
            conv.relu_fused = True
        if getattr(blk, "channel_shuffle_flag", 0) and getattr(blk, "shuffle_groups", 1) > 1 and conv.in_channels % blk.shuffle_groups == 0:
            conv.in_shuffle_groups = int(blk.shuffle_groups)
            blk.channel_shuffle_flag = 0


def prepare(model, inplace=False, a_bits=8, w_bits=8, q_type=0, q_level=0, weight_observer=0, bn_fuse=False,
            bn_fuse_calib=False, quant_inference=False, pretrained_model=False, qaft=False, ptq=False, percentile=0.9999, fuse_bn_act=True, fuse_blocks=True):
    """Same rewrite as the reference (ref 1791-1830).  ``fuse_bn_act`` (ours, default on): see add_quant_op; off = exactly the reference's module classes.
    ``fuse_blocks`` (ours, default on, only with ``bn_fuse``): see ``_fuse_bnfuse_blocks``."""
    if not inplace:
        model = copy.deepcopy(model)
    add_quant_op(model, a_bits=a_bits, w_bits=w_bits, q_type=q_type, q_level=q_level, weight_observer=weight_observer,
                 bn_fuse=bn_fuse, bn_fuse_calib=bn_fuse_calib, quant_inference=quant_inference,
                 pretrained_model=pretrained_model, qaft=qaft, ptq=ptq, percentile=percentile, fuse_bn_act=fuse_bn_act)
    if fuse_bn_act:
        _fuse_residual_tails(model)
    if bn_fuse and fuse_blocks:
        _fuse_bnfuse_blocks(model)
    return model
