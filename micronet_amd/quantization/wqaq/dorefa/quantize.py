"""DoReFa k-bit fake-quantised layers on MI355X -- same module surface as the reference's
``micronet/compression/quantization/wqaq/dorefa/quantize.py`` (class names, constructor signatures, attributes,
``state_dict`` keys, ``prepare`` rewrite rule), with the arithmetic done by hand-written gfx950 kernels:

  * ``QuantConv2d`` / ``QuantLinear`` forward (ref 107-122 / 192-199): one weight-quantizer launch pair + ONE fused
    kernel that clamps/rounds the activations in the implicit-GEMM prologue and contracts on the MFMA units;
    backward: the clip-STE of the activation quantizer is the epilogue of the backward-data kernel.
  * ``ActivationQuantizer`` / ``WeightQuantizer`` / ``Round`` (ref 11-73) remain callable on their own
    (``m.weight_quantizer(m.weight)`` is used by the reference's quant_model_test scripts).
"""
import copy

import torch
import torch.nn as nn
from torch.autograd import Function

from micronet_amd import ops

__all__ = ["Round", "ActivationQuantizer", "WeightQuantizer", "QuantConv2d", "QuantConvTranspose2d", "QuantLinear",
           "add_quant_op", "prepare"]


class Round(Function):
    """sign(v) * floor(|v| + 0.5) with a straight-through gradient (ref 11-21)."""

    @staticmethod
    def forward(self, input):
        return ops.RoundHalfAway.forward(self, input)

    @staticmethod
    def backward(self, grad_output):
        return grad_output.clone()


def _check_bits(bits):
    if bits == 1:
        print("！Binary quantization is not supported ！")
        assert bits != 1


class ActivationQuantizer(nn.Module):
    def __init__(self, a_bits):
        super().__init__()
        self.a_bits = a_bits

    def round(self, input):
        return Round.apply(input)

    def forward(self, input):
        if self.a_bits == 32:
            return input
        _check_bits(self.a_bits)
        return ops.DorefaAct.apply(input, self.a_bits)


class WeightQuantizer(nn.Module):
    def __init__(self, w_bits):
        super().__init__()
        self.w_bits = w_bits

    def round(self, input):
        return Round.apply(input)

    def forward(self, input):
        if self.w_bits == 32:
            return input
        _check_bits(self.w_bits)
        pre = self.__dict__.pop("_mn_pre", None)
        if pre is not None and pre[0] is input:
            return pre[1]            # computed ahead for all layers in one launch (micronet_amd.train.prefetch_weight_path)
        return ops.DorefaWeight.apply(input, self.w_bits)


def _weight_is_coded(module):
    """May the code-domain kernels read ``module``'s effective weights as integer codes (2k - n) / n?  Always for the training graph (the weight quantizer
    produced them).  ``quant_inference=True`` uses the STORED weights as they are (ref 107-122): only when they lie on the quantizer's grid -- true after
    ``inference.prequantize_weights`` / quant_model_test.py:189-191 -- which is checked once per weight version (one host sync); anything else keeps the
    fp32-weight kernels, as the reference would convolve raw weights."""
    b = module.weight_quantizer.w_bits
    if not (2 <= b <= 8):
        return False
    if not module.quant_inference:
        return True
    w = module.weight
    key = (w.data_ptr(), w._version, tuple(w.shape))
    c = module.__dict__.get("_mn_grid")
    if c is None or c[0] != key:
        if w.is_cuda and torch.cuda.is_current_stream_capturing():
            # the check reads a device value on the host: not possible while a HIP graph is being captured, and the verdict must not be guessed
            raise RuntimeError("quant_inference forward under stream capture before the stored weights were checked against the quantizer grid: run one eager "
                               "forward first (or micronet_amd.inference.prequantize_weights(model), which records the verdict)")
        ok = False
        if w.numel() and w.is_cuda and w.dtype == torch.float32:
            n = float(2 ** b - 1)
            with torch.no_grad():
                k = (w.detach() * n + n) * 0.5
                ok = bool(((k - k.round()).abs().max() <= 1e-4) & (k.min() >= -1e-4) & (k.max() <= n + 1e-4))
        # (the checked storage is kept alive with the verdict: a later `w.data = other` can then never land on the same address with a stale `ok`)
        c = (key, ok, w.detach())
        module.__dict__["_mn_grid"] = c
    return c[1]


def _forget_weight_grid(module, *args, **kwargs):
    """load_state_dict pre-hook of the quantised layers: new stored weights, new verdict"""
    module.__dict__.pop("_mn_grid", None)


def _wdesc(module):
    """weight-code descriptor for the code-domain conv kernels: w = (2k - n)/n, n = 2^w_bits - 1"""
    b = module.weight_quantizer.w_bits
    return (ops.WQ_DOREFA, b, 0, 0, None) if (2 <= b <= 8 and _weight_is_coded(module)) else None


def _aq_args(quantizer):
    """(mode, bits) of the activation quantizer fused into the conv kernels."""
    if quantizer.a_bits == 32:
        return ops.ACTQ_NONE, 0
    _check_bits(quantizer.a_bits)
    return ops.ACTQ_DOREFA, quantizer.a_bits


class QuantConv2d(nn.Conv2d):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 padding_mode="zeros", a_bits=8, w_bits=8, quant_inference=False):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, padding_mode)
        self.quant_inference = quant_inference
        self.activation_quantizer = ActivationQuantizer(a_bits=a_bits)
        self.weight_quantizer = WeightQuantizer(w_bits=w_bits)
        self.in_shuffle_groups = 0     # > 1: this conv reads channel_shuffle(input, groups) (set by prepare(fold_shuffle=True))
        self.lazy_for_bn = False       # True (set by prepare(fuse_blocks=True)): a BatchNorm2dReLU of ours consumes the output -> it may stay un-computed
        self._register_load_state_dict_pre_hook(_forget_weight_grid, with_module=True)          # (a module-level function: the module must stay picklable)

    def forward(self, input):
        from micronet_amd.sign_tensor import QActTensor
        quant_weight = self.weight if self.quant_inference else self.weight_quantizer(self.weight)
        mode, bits = _aq_args(self.activation_quantizer)
        if isinstance(input, QActTensor):
            # the block in front already evaluated THIS conv's activation quantizer (codes, one byte per element)
            w_bits = self.weight_quantizer.w_bits
            if (self.lazy_for_bn and _weight_is_coded(self) and input.bits == bits and mode == ops.ACTQ_DOREFA and
                    ops.qconv_bnq_supported(input, quant_weight, self.stride, self.padding, self.dilation, self.groups, w_bits, self.in_shuffle_groups)):
                return ops.QConvCodeLazy.apply(input, quant_weight, self.bias, self.stride, self.padding, self.dilation, self.groups, w_bits,
                                               self.in_shuffle_groups or 0)
            if (_weight_is_coded(self) and input.bits == bits and mode == ops.ACTQ_DOREFA and not self.in_shuffle_groups and
                    ops.code_classifier_supported(input, quant_weight, self.stride, self.padding, self.dilation, self.groups)):
                return ops.CodeClassifierConv.apply(input, quant_weight, self.bias)       # the 1024 -> 10 classifier: reads the codes directly
            input = ops.QActToFloat.apply(input)       # anything else: the fp32 activation the reference holds here (one streaming kernel)
        # like the reference, the forward always zero-pads whatever padding_mode says (ref 113-121)
        return ops.qconv2d(input, quant_weight, self.bias, self.stride, self.padding, self.dilation, self.groups,
                           aq_mode=mode, aq_bits=bits, wdesc=_wdesc(self), in_shuffle=self.in_shuffle_groups)


class QuantConvTranspose2d(nn.ConvTranspose2d):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, output_padding=0, dilation=1,
                 groups=1, bias=True, padding_mode="zeros", a_bits=8, w_bits=8, quant_inference=False):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, output_padding, groups, bias,
                         dilation, padding_mode)
        self.quant_inference = quant_inference
        self.activation_quantizer = ActivationQuantizer(a_bits=a_bits)
        self.weight_quantizer = WeightQuantizer(w_bits=w_bits)

    def forward(self, input):
        quant_input = self.activation_quantizer(input)
        quant_weight = self.weight if self.quant_inference else self.weight_quantizer(self.weight)
        return ops.ConvTranspose2d.apply(quant_input, quant_weight, self.bias, self.stride, self.padding,
                                         self.output_padding, self.groups, self.dilation)


class QuantLinear(nn.Linear):
    def __init__(self, in_features, out_features, bias=True, a_bits=8, w_bits=8, quant_inference=False):
        super().__init__(in_features, out_features, bias)
        self.quant_inference = quant_inference
        self.activation_quantizer = ActivationQuantizer(a_bits=a_bits)
        self.weight_quantizer = WeightQuantizer(w_bits=w_bits)

    def forward(self, input):
        quant_weight = self.weight if self.quant_inference else self.weight_quantizer(self.weight)
        mode, bits = _aq_args(self.activation_quantizer)
        return ops.qlinear(input, quant_weight, self.bias, aq_mode=mode, aq_bits=bits, wdesc=_wdesc(self))


def _swap_conv(child, cls, **kw):
    new = cls(child.in_channels, child.out_channels, child.kernel_size, stride=child.stride, padding=child.padding,
              dilation=child.dilation, groups=child.groups, bias=child.bias is not None,
              padding_mode=child.padding_mode, **kw)
    if child.bias is not None:
        new.bias.data = child.bias
    new.weight.data = child.weight      # shares the original storage, as the reference does
    return new


def _accstats_of(t):
    """the exact accumulator sums a dense IAO conv left on its output ``t`` (``_mn_accstats``), or None -- also None once the tensor was written in place since"""
    st = getattr(t, "_mn_accstats", None)
    if st is None or st[-1] != t._version:
        return None
    return st[:-1]


class BatchNorm2dReLU(nn.BatchNorm2d):
    """``nn.BatchNorm2d`` whose forward also applies the ReLU behind it (same parameters, buffers and ``state_dict`` keys): one fused
    gfx950 op (ops.BNReLU: three streaming passes forward, five backward, z never stored) instead of MIOpen's BatchNorm kernels plus
    separate ReLU forward / backward kernels.  Installed by ``prepare(fuse_bn_act=True)`` in front of a ReLU that the parent calls right
    after it; that ReLU becomes a ``ReLUAfterFusedBN`` (a no-op ``nn.ReLU``)."""

    q_out_bits = 0          # > 0 (set by prepare(fuse_blocks=True)): the only consumer is the a-bit activation quantizer of the next QuantConv2d ->
    q_pool = False          # emit its codes (QActTensor), through the 2x2 max-pool behind the block when q_pool
    q_also_f32 = False      # the consumer is a fused residual block with an identity shortcut: emit the codes AND the fp32 activation (one pass, two autograd outputs)
    emit_minmax = False     # (set by the IAO prepare) leave per-block (min, max) of the output for the observer of the IAO layer that reads it
    iao_lazy_out = False    # (set by the IAO prepare) the only consumer is the next dense IAO QuantConv2d of the same Sequential: behind a conv that left its epilogue
    #                         statistics the output stays un-computed (ops.BNActLazy -> LazyBNAct) and that conv pulls its activation codes from y in one pass

    def forward(self, input):
        from micronet_amd import ops
        from micronet_amd.sign_tensor import LazyQConvOut
        import torch.nn.functional as F
        lazy = isinstance(input, LazyQConvOut)
        from micronet_amd.sign_tensor import LazyConvOut
        if ops.FIRST_FUSED_QA and isinstance(input, LazyConvOut) and input.recipe.get("kind") == "first" and input._mn_value is None and self.affine and self.momentum is not None and \
                self.training and self.q_out_bits and not self.q_pool and not self.q_also_f32:
            # the un-computed output of the first (un-quantised) conv: conv + batch statistics (Gram data of the image) + BatchNorm + ReLU + the next conv's
            # quantizer in one kernel
            if self.track_running_stats and self.num_batches_tracked is not None:
                self.num_batches_tracked.add_(1)
            return ops.FirstConvBNReLUQ.apply(input, self.weight, self.bias, self.running_mean if self.track_running_stats else None,
                                              self.running_var if self.track_running_stats else None, self.eps, self.momentum, int(self.q_out_bits))
        plain_ok = self.affine and ops.bnrelu_supported(input)
        use_batch = self.training or self.running_mean is None
        stats_ok = self.momentum is not None and (use_batch or self.track_running_stats)
        pool = bool(self.q_pool and self.q_out_bits)
        if (lazy or (plain_ok and self.q_out_bits)) and self.affine and stats_ok and ops.qa_supported(input.shape, pool):
            nbt = None
            if self.training and self.track_running_stats and self.num_batches_tracked is not None:
                if lazy and self.num_batches_tracked.is_cuda and self.num_batches_tracked.dtype == torch.int64:
                    nbt = self.num_batches_tracked           # incremented by the launch that forms the statistics
                else:
                    self.num_batches_tracked.add_(1)
            if self.q_also_f32 and self.q_out_bits and not pool:
                q, a = ops.BNAddReLUQ.apply(input, self.weight, self.bias, self.running_mean if self.track_running_stats else None,
                                            self.running_var if self.track_running_stats else None, self.eps, self.momentum, use_batch, nbt,
                                            None, None, None, None, None, 0.0, 0.0, None, int(self.q_out_bits), True)
                q._mn_f32 = a
                return q
            return ops.BNReLUQ.apply(input, self.weight, self.bias, self.running_mean if self.track_running_stats else None,
                                     self.running_var if self.track_running_stats else None, self.eps, self.momentum, use_batch, nbt,
                                     int(self.q_out_bits), pool)
        if not plain_ok:
            ops.note_fallback("BatchNorm2dReLU -> nn.BatchNorm2d + relu")
            return F.relu(super().forward(input))
        momentum = 0.0 if self.momentum is None else self.momentum
        if self.training and self.track_running_stats and self.num_batches_tracked is not None:
            if not self.__dict__.pop("_mn_nbt_pre", False):          # (already incremented for this forward by micronet_amd.train.bump_bn_counters: one launch for the net)
                self.num_batches_tracked.add_(1)
            if self.momentum is None:
                momentum = 1.0 / float(self.num_batches_tracked)
        acc = _accstats_of(input) if (self.training and self.track_running_stats) else None
        if self.iao_lazy_out and acc is not None and ops.iao_bn_lazy_supported(input, acc):
            return ops.BNActLazy.apply(input, self.weight, self.bias, self.running_mean, self.running_var, self.eps, momentum, 1, acc)
        if self.emit_minmax and self.training:
            out = ops.BNReLU.apply(input, self.weight, self.bias, self.running_mean if self.track_running_stats else None,
                                   self.running_var if self.track_running_stats else None, self.eps, momentum, use_batch, "mn_bnrelu", True, acc)
            mm = ops.take_minmax()
            if mm is not None:
                out._mn_minmax = mm + (out._version,)          # (valid only while nothing writes into the tensor in place)
            return out
        return ops.BNReLU.apply(input, self.weight, self.bias, self.running_mean if self.track_running_stats else None,
                                self.running_var if self.track_running_stats else None, self.eps, momentum, use_batch, "mn_bnrelu", False, acc)


class BatchNorm2dPlain(nn.BatchNorm2d):
    """``nn.BatchNorm2d`` (no activation behind it -- the BatchNorms in front of a residual add, models/resnet.py:21-29) on the streaming kernels of
    ``BatchNorm2dReLU`` instead of MIOpen's: same parameters, buffers and ``state_dict`` keys; anything the kernels do not cover runs the stock forward."""

    emit_minmax = False     # (set by the IAO prepare) leave per-block (min, max) of the output for the observers of the QuantAdd that reads it
    iao_lazy_out = False    # (set by the IAO prepare) the only consumer is the residual block's QuantAdd: behind a conv that left its epilogue statistics the output
    #                         stays un-computed (ops.BNActLazy -> LazyBNAct) and the QuantAdd normalises, quantises and adds in one pass (ops.IaoQuantAddBN)

    def forward(self, input):
        from micronet_amd import ops
        use_batch = self.training or self.running_mean is None
        if not (self.affine and ops.bnrelu_supported(input) and self.momentum is not None and (use_batch or self.track_running_stats)):
            ops.note_fallback("BatchNorm2dPlain -> nn.BatchNorm2d")
            return super().forward(input)
        if self.training and self.track_running_stats and self.num_batches_tracked is not None:
            if not self.__dict__.pop("_mn_nbt_pre", False):
                self.num_batches_tracked.add_(1)
        acc = _accstats_of(input) if (self.training and self.track_running_stats) else None
        if self.iao_lazy_out and acc is not None and ops.iao_bn_lazy_supported(input, acc):
            return ops.BNActLazy.apply(input, self.weight, self.bias, self.running_mean, self.running_var, self.eps, self.momentum, 2, acc)
        if self.emit_minmax and self.training:          # (set by the IAO prepare: an IAO QuantAdd observes this output)
            out = ops.BNReLU.apply(input, self.weight, self.bias, self.running_mean if self.track_running_stats else None,
                                   self.running_var if self.track_running_stats else None, self.eps, self.momentum, use_batch, "mn_bn2d", True, acc)
            mm = ops.take_minmax()
            if mm is not None:
                out._mn_minmax = mm + (out._version,)
            return out
        return ops.BNReLU.apply(input, self.weight, self.bias, self.running_mean if self.track_running_stats else None,
                                self.running_var if self.track_running_stats else None, self.eps, self.momentum, use_batch, "mn_bn2d", False, acc)


class MaxPool2dF32(nn.MaxPool2d):
    """``nn.MaxPool2d`` whose 2x2 / stride-2 case runs on the gfx950 kernels (byte argmax, scatter backward); anything else is the stock module."""

    def forward(self, input):
        from micronet_amd import ops
        from micronet_amd.sign_tensor import QActTensor
        if isinstance(input, QActTensor) and getattr(input, "_mn_pooled", False) and getattr(self, "_mn_fused_pool", False):
            return input          # the fused block in front already pooled (max of the activation = max of its codes: the quantizer is monotone)
        if not self.return_indices and ops.f32_pool_supported(input, self.kernel_size, self.stride, self.padding, self.dilation, self.ceil_mode):
            return ops.MaxPool2x2F32.apply(input)
        ops.note_fallback("MaxPool2dF32 -> nn.MaxPool2d")
        return super().forward(input)


class ReLUAfterFusedBN(nn.ReLU):
    """The ``nn.ReLU`` behind a ``BatchNorm2dReLU``: the rectification already happened in the fused op (relu is idempotent, so this is the
    same function), the module stays in place (``isinstance(m, nn.ReLU)``, module names) and costs no kernel."""

    def forward(self, input):
        return input


def _ordered_parent(module):
    """True when the parent is KNOWN to call its children in definition order, so bn -> relu adjacency means bn feeds relu: ``nn.Sequential``, the
    ``ConvBNReLU`` block of the reference's own model files (models/nin.py, models/nin_gc.py -- identified by class AND defining module, not by
    name alone), or a user block that opts in with ``_mn_ordered_forward = True``."""
    return isinstance(module, nn.Sequential) or _is_ref_block(module)


def _is_ref_block(module):
    """The reference's ``ConvBNReLU`` (shuffle -> conv -> bn -> relu, models/nin_gc.py:18-59) or a block that declares the same call order."""
    t = type(module)
    return bool(getattr(module, "_mn_ordered_forward", False)) or (t.__name__ == "ConvBNReLU" and t.__module__.split(".")[-1] in ("nin", "nin_gc"))


def _fuse_blocks(model, fold_shuffle=True):
    """Second pass of ``prepare(fuse_blocks=True)``: wherever a ``ConvBNReLU`` block's output feeds ONLY the activation quantizer of the next block's
    QuantConv2d (directly, or through a 2x2 / stride-2 max-pool) inside an ``nn.Sequential``, the producing BatchNorm2dReLU emits that quantizer's
    codes (``q_out_bits`` / ``q_pool``), and every QuantConv2d followed by a BatchNorm2dReLU in its block may leave its output un-computed
    (``lazy_for_bn``); the block's channel shuffle moves into the conv's addressing (``in_shuffle_groups``)."""
    def is_block(m):
        return _is_ref_block(m) and isinstance(getattr(m, "bn", None), BatchNorm2dReLU) and isinstance(getattr(m, "conv", None), nn.Conv2d)

    def two(v):
        return v in (2, (2, 2), [2, 2])
    for parent in model.modules():
        if not isinstance(parent, nn.Sequential):
            continue
        kids = list(parent.children())
        for i, blk in enumerate(kids):
            if not is_block(blk):
                continue
            if isinstance(blk.conv, QuantConv2d):
                blk.conv.lazy_for_bn = True
                if fold_shuffle and getattr(blk, "channel_shuffle_flag", 0) and getattr(blk, "shuffle_groups", 1) > 1 and \
                        blk.conv.in_channels % blk.shuffle_groups == 0:
                    blk.conv.in_shuffle_groups = int(blk.shuffle_groups)
                    blk.channel_shuffle_flag = 0
            nxt = kids[i + 1] if i + 1 < len(kids) else None
            pool = None
            if isinstance(nxt, MaxPool2dF32) and two(nxt.kernel_size) and two(nxt.stride) and nxt.padding in (0, (0, 0)) and nxt.dilation in (1, (1, 1)) \
                    and not nxt.ceil_mode and not nxt.return_indices:
                pool, nxt = nxt, (kids[i + 2] if i + 2 < len(kids) else None)
            # (the streaming kernels read either stash width, pooled or not: mn_qa_* with in_kind 0 / 2)
            if nxt is not None and is_block(nxt) and isinstance(nxt.conv, QuantConv2d):
                bits = nxt.conv.activation_quantizer.a_bits
                if 2 <= bits <= 8 and 2 <= nxt.conv.weight_quantizer.w_bits <= 8:          # (8-bit codes: the wide kernels, 32-bit stash)
                    blk.bn.q_out_bits = int(bits)
                    blk.bn.q_pool = pool is not None
                    from micronet_amd.nn import Conv2dFirst
                    if isinstance(blk.conv, Conv2dFirst) and pool is None and not getattr(blk, "channel_shuffle_flag", 0):
                        blk.conv.lazy_for_bn = "qa"          # the first block: conv + BatchNorm + ReLU + quantizer in one kernel (ops.FirstConvBNReLUQ, knob MN_FIRST_FUSED_QA)
                    if pool is not None:
                        pool._mn_fused_pool = True


def _bn_call_args(bn, lazy):
    """(running_mean, running_var, eps, momentum, use_batch, nbt) of a BatchNorm2d for the fused ops, with nn.BatchNorm2d's bookkeeping; None when the
    module's configuration is one the fused kernels do not cover (no affine parameters, cumulative moving average)."""
    if not bn.affine or bn.momentum is None:
        return None
    use_batch = bn.training or bn.running_mean is None
    if not (use_batch or bn.track_running_stats):
        return None
    nbt = None
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        if lazy and bn.num_batches_tracked.is_cuda and bn.num_batches_tracked.dtype == torch.int64:
            nbt = bn.num_batches_tracked                 # incremented by the launch that forms the statistics
        else:
            bn.num_batches_tracked.add_(1)
    return (bn.running_mean if bn.track_running_stats else None, bn.running_var if bn.track_running_stats else None, bn.eps, bn.momentum, use_batch, nbt)


class _FusedBasicBlockMixin:
    """forward of the reference's ``BasicBlock`` (models/resnet.py:7-65: relu(add(residual_function(x), shortcut(x)))) on the fused k-bit kernels, installed by
    ``prepare(fuse_blocks=True)`` as a subclass of the block's own class (same children, parameters, ``state_dict``).  The block's input arrives as a ``QActTensor``
    (codes of the first conv's quantizer + the fp32 activation for an identity shortcut); conv -> bn -> relu -> quantizer of the second conv is the fused
    block of ``BatchNorm2dReLU``; the block's end -- bn + shortcut (identity, or 1x1 conv + bn) + relu + the NEXT block's quantizer -- is ``ops.BNAddReLUQ``.
    Any other input (or a configuration the kernels do not cover) runs the reference's forward."""

    _mn_out_bits = 0         # codes of the next block's activation quantizer (0: the consumer is not a fused block)
    _mn_out_f32 = True       # the fp32 activation as well (the next block's identity shortcut / a foreign consumer)
    _mn_mid_hook = None      # optional callable applied to the activation between the two convs (a QActTensor): the parity tests teacher-force it

    def forward(self, x):
        from micronet_amd import ops
        from micronet_amd.sign_tensor import LazyQConvOut, QActTensor
        rf = self.residual_function
        conv_a, bn_a, conv_b, bn_b = rf[0], rf[1], rf[3], rf[4]
        has_sc = len(self.shortcut) > 0
        ok = isinstance(x, QActTensor) and x.bits == conv_a.activation_quantizer.a_bits and _weight_is_coded(conv_a) and _weight_is_coded(conv_b)
        if ok and not has_sc and x._mn_f32 is None:
            ok = False
        w_bits = conv_a.weight_quantizer.w_bits
        if ok and has_sc:
            conv_s = self.shortcut[0]
            ok = (_weight_is_coded(conv_s) and conv_s.weight_quantizer.w_bits == w_bits and conv_s.activation_quantizer.a_bits == x.bits and
                  conv_a.bias is None and conv_s.bias is None and
                  ops.qconv_bnq_supported(x, conv_a.weight, conv_a.stride, conv_a.padding, conv_a.dilation, conv_a.groups, w_bits, 0) and
                  ops.qconv_bnq_supported(x, conv_s.weight, conv_s.stride, conv_s.padding, conv_s.dilation, conv_s.groups, w_bits, 0))
        if not ok:
            return super().forward(ops.QActToFloat.apply(x) if isinstance(x, QActTensor) else x)
        if has_sc:
            wq = lambda c: c.weight if c.quant_inference else c.weight_quantizer(c.weight)
            ya, ys = ops.QConvCodeLazy2.apply(x, wq(conv_a), wq(conv_s),
                                              (conv_a.stride, conv_a.padding, conv_a.dilation, conv_a.groups),
                                              (conv_s.stride, conv_s.padding, conv_s.dilation, conv_s.groups), w_bits)
        else:
            ya, ys = conv_a(x), None
        h = rf[2](bn_a(ya))                       # BatchNorm2dReLU -> codes of conv_b's quantizer; rf[2] is the no-op ReLUAfterFusedBN
        if self._mn_mid_hook is not None:
            h = self._mn_mid_hook(h)
        yb = conv_b(h)
        args_b = _bn_call_args(bn_b, isinstance(yb, LazyQConvOut)) if isinstance(yb, LazyQConvOut) else None
        args_s = _bn_call_args(self.shortcut[1], True) if (has_sc and args_b is not None) else None
        if args_b is None or (has_sc and args_s is None):
            # a configuration outside the fused kernels: the reference's tail on materialised tensors
            sc = self.shortcut[1](ys) if has_sc else ops.QActToFloat.apply(x) if x._mn_f32 is None else x._mn_f32
            return nn.ReLU(inplace=True)(self.add(bn_b(yb), sc))
        bn_s = self.shortcut[1] if has_sc else None
        q, a = ops.BNAddReLUQ.apply(yb, bn_b.weight, bn_b.bias, args_b[0], args_b[1], args_b[2], args_b[3], args_b[4], args_b[5],
                                    ys if has_sc else x._mn_f32,
                                    bn_s.weight if has_sc else None, bn_s.bias if has_sc else None, args_s[0] if has_sc else None, args_s[1] if has_sc else None,
                                    args_s[2] if has_sc else 0.0, args_s[3] if has_sc else 0.0, args_s[5] if has_sc else None,
                                    int(self._mn_out_bits), bool(self._mn_out_f32 or not self._mn_out_bits))
        if not self._mn_out_bits:
            return a
        q._mn_f32 = a if a.numel() else None
        return q


def _is_ref_basic_block(m):
    t = type(m)
    return (t.__name__ == "BasicBlock" and t.__module__.split(".")[-1] == "resnet") or bool(getattr(m, "_mn_basic_block", False))


def _fuse_residual_blocks(model):
    """Third pass of ``prepare(fuse_blocks=True)``: every ``BasicBlock`` of the reference's ResNets (models/resnet.py:7-65) whose convs became QuantConv2d
    with 2..7-bit activations becomes a ``_FusedBasicBlockMixin`` subclass of its class.  Where the call order is known -- the reference's ``ResNet``
    (stem ``conv1``, stages ``conv2_x .. conv5_x`` of blocks in ``nn.Sequential``) -- each block is told which outputs its consumer wants: the next block's
    quantizer codes, and the fp32 activation only when that block's shortcut is the identity; the stem's BatchNorm2dReLU emits both."""
    from micronet_amd.base_module.op import Add

    def fusable(b):
        rf, sc = getattr(b, "residual_function", None), getattr(b, "shortcut", None)
        if not (isinstance(rf, nn.Sequential) and isinstance(sc, nn.Sequential) and len(rf) == 5 and len(sc) in (0, 2) and type(getattr(b, "add", None)) is Add):
            return False
        if not (isinstance(rf[0], QuantConv2d) and isinstance(rf[1], BatchNorm2dReLU) and isinstance(rf[2], ReLUAfterFusedBN) and isinstance(rf[3], QuantConv2d)
                and type(rf[4]) is nn.BatchNorm2d):
            return False
        if len(sc) and not (isinstance(sc[0], QuantConv2d) and type(sc[1]) is nn.BatchNorm2d):
            return False
        convs = [rf[0], rf[3]] + ([sc[0]] if len(sc) else [])
        return all(2 <= c.activation_quantizer.a_bits <= 7 and 2 <= c.weight_quantizer.w_bits <= 8 and c.bias is None for c in convs)

    blocks = []
    for m in model.modules():
        if _is_ref_basic_block(m) and fusable(m):
            cls = type(m)
            from micronet_amd.nn import derive_class
            m.__class__ = derive_class("Fused", _FusedBasicBlockMixin, cls)
            rf = m.residual_function
            rf[0].lazy_for_bn = True
            rf[3].lazy_for_bn = True
            rf[1].q_out_bits = int(rf[3].activation_quantizer.a_bits)
            rf[1].q_pool = False
            blocks.append(m)
    if not blocks:
        return
    # call order: only for the reference's ResNet layout
    t = type(model)
    stages = [getattr(model, "conv%d_x" % i, None) for i in range(2, 6)]
    if not (t.__name__ == "ResNet" and t.__module__.split(".")[-1] == "resnet" and all(isinstance(s, nn.Sequential) for s in stages)):
        return          # unknown call order: every fused block emits fp32 only (``_mn_out_bits`` 0) and re-quantises its input itself -- still correct
    chain = [b for s in stages for b in s.children()]
    for i, b in enumerate(chain):
        nxt = chain[i + 1] if i + 1 < len(chain) else None
        if isinstance(b, _FusedBasicBlockMixin) and isinstance(nxt, _FusedBasicBlockMixin):
            b._mn_out_bits = int(nxt.residual_function[0].activation_quantizer.a_bits)
            b._mn_out_f32 = len(nxt.shortcut) == 0
        elif isinstance(b, _FusedBasicBlockMixin):
            b._mn_out_bits, b._mn_out_f32 = 0, True
    stem = getattr(model, "conv1", None)
    if isinstance(stem, nn.Sequential) and len(stem) == 3 and isinstance(stem[1], BatchNorm2dReLU) and isinstance(stem[2], ReLUAfterFusedBN) and chain and \
            isinstance(chain[0], _FusedBasicBlockMixin):
        stem[1].q_out_bits = int(chain[0].residual_function[0].activation_quantizer.a_bits)
        stem[1].q_pool = False
        stem[1].q_also_f32 = len(chain[0].shortcut) == 0


def add_quant_op(module, layer_counter, a_bits=8, w_bits=8, quant_inference=False, fuse_bn_act=True):
    """Swap every conv / conv-transpose / linear EXCEPT the first one met (ref 202-309: ``layer_counter[0] > 1``)."""
    kw = dict(a_bits=a_bits, w_bits=w_bits, quant_inference=quant_inference)
    prev = None
    for name, child in module.named_children():
        if (fuse_bn_act and type(child) is nn.ReLU and type(prev) is nn.BatchNorm2d and prev.affine and prev.track_running_stats
                and _ordered_parent(module)):
            prev.__class__ = BatchNorm2dReLU          # same object and state: only its class changes
            child.__class__ = ReLUAfterFusedBN
            prev = module._modules[name]
            continue
        prev = child
        if fuse_bn_act and type(child) is nn.MaxPool2d:
            child.__class__ = MaxPool2dF32            # same object and state
            continue
        if fuse_bn_act and type(child) is nn.AvgPool2d:
            from micronet_amd.nn import AvgPool2dGlobal
            child.__class__ = AvgPool2dGlobal         # same object and state; its own kernel only when the window is the whole image
            continue
        if isinstance(child, nn.Conv2d):
            layer_counter[0] += 1
            if layer_counter[0] > 1:
                module._modules[name] = _swap_conv(child, QuantConv2d, **kw)
            elif type(child) is nn.Conv2d:
                from micronet_amd.nn import Conv2dFirst
                child.__class__ = Conv2dFirst      # the un-quantised first conv: same object and state, gfx950 kernels when covered
        elif isinstance(child, nn.ConvTranspose2d):
            layer_counter[0] += 1
            if layer_counter[0] > 1:
                new = QuantConvTranspose2d(child.in_channels, child.out_channels, child.kernel_size, stride=child.stride,
                                           padding=child.padding, output_padding=child.output_padding,
                                           dilation=child.dilation, groups=child.groups, bias=child.bias is not None,
                                           padding_mode=child.padding_mode, **kw)
                if child.bias is not None:
                    new.bias.data = child.bias
                new.weight.data = child.weight
                module._modules[name] = new
        elif isinstance(child, nn.Linear):
            layer_counter[0] += 1
            if layer_counter[0] > 1:
                new = QuantLinear(child.in_features, child.out_features, bias=child.bias is not None, **kw)
                if child.bias is not None:
                    new.bias.data = child.bias
                new.weight.data = child.weight
                module._modules[name] = new
        else:
            add_quant_op(child, layer_counter, fuse_bn_act=fuse_bn_act, **kw)


def prepare(model, inplace=False, a_bits=8, w_bits=8, quant_inference=False, fuse_bn_act=True, fuse_blocks=True, fold_shuffle=True):
    """Same rewrite as the reference (ref 312-320).  ``fuse_bn_act`` (ours, default on): a ``BatchNorm2d`` directly in front of a ``ReLU`` in a
    block that calls them in that order becomes ``BatchNorm2dReLU`` (one fused op) and the ReLU a no-op subclass; with it off the module
    graph is exactly the reference's.  ``fuse_blocks`` (ours, default on, needs ``fuse_bn_act``): conv + BatchNorm + ReLU (+ 2x2 max-pool) + the next
    layer's activation quantizer run as the fused k-bit block of ``_fuse_blocks`` -- activations travel as one-byte codes, the conv output as a
    16-bit integer stash; numerically the same function (tests compare with it off).  ``fold_shuffle``: see ``_fuse_blocks``."""
    if not inplace:
        model = copy.deepcopy(model)
    add_quant_op(model, [0], a_bits=a_bits, w_bits=w_bits, quant_inference=quant_inference, fuse_bn_act=fuse_bn_act)
    if fuse_bn_act and fuse_blocks:          # (quant_inference graphs too: their convs take the code kernels once the stored weights are on the grid)
        _fuse_blocks(model, fold_shuffle=fold_shuffle)
        _fuse_residual_blocks(model)
    if fuse_bn_act:
        from micronet_amd.nn import fuse_tail
        fuse_tail(model)          # bn -> relu -> global average pool of the last block: one kernel per direction (TailBNMixin)
    return model
