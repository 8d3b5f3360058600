"""Plain (un-quantised) layers of the reference nets that sit ON the QAT step and are worth a gfx950 kernel.

``Conv2dFirst``: the first convolution of a net is skipped by the DoReFa and WbWtAb rewrites (wqaq/dorefa/quantize.py:206,
wbwtab/quantize.py:251) and stays an ``nn.Conv2d`` -- but its output is the largest tensor of the step.  ``prepare()`` switches
that module (same object, same parameters, same ``state_dict``) to this subclass, whose forward / backward-weight run on
``conv_first.hip`` (exact fp32 products on v_mfma_f32_16x16x4_f32, bias and dbias fused) whenever the geometry is covered;
anything else falls through to ``nn.Conv2d``."""
import torch
import torch.nn as nn

from . import ops
from .sign_tensor import LazyConvOut, SignTensor


class Conv2dFirst(nn.Conv2d):
    lazy_for_bn = False        # True (set by prepare()): a fused BatchNorm block of ours consumes the output in training mode -> it may stay un-computed (ops.FirstConvLazy)

    def forward(self, input):
        if (input.is_cuda and input.dtype == torch.float32 and input.dim() == 4 and self.padding_mode == "zeros" and
                not isinstance(input, (SignTensor, LazyConvOut)) and not isinstance(self.padding, str) and
                ops.first_conv_supported(input.shape, self.weight.shape, self.stride, self.padding, self.dilation, self.groups)):
            if (self.lazy_for_bn and self.training and ops.FIRST_FUSED and (self.lazy_for_bn != "qa" or ops.FIRST_FUSED_QA) and ops.FIRST_GRAM and ops.LAZY_BN_GRAD and torch.is_grad_enabled() and
                    not input.requires_grad and self.weight.requires_grad and type(input) is torch.Tensor):
                return ops.FirstConvLazy.apply(input, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
            out = ops.qconv2d(input, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
            if not input.requires_grad:
                out._mn_first_conv_out = True      # no backward-data: a BatchNorm2dBinAct behind it may hand its gradient over lazily
                if not isinstance(out, LazyConvOut):
                    out._mn_first_conv = ops.FirstConvRecord(out, input, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups)
            return out
        ops.note_fallback("Conv2dFirst -> nn.Conv2d")
        return super().forward(input)


class Conv2dSignIn(nn.Conv2d):
    """The LAST conv of a WbWtAb net: full-precision weights (the rewrite skips it, wbwtab/quantize.py:251) but a +-1 input.  When
    that input arrives packed (``SignTensor``) and the layer is a 1x1 classifier with few outputs it runs on the sign-code kernels
    (``mn_signconv1x1_small_*``); otherwise it is an ordinary ``nn.Conv2d`` (a SignTensor is then unpacked on the way in)."""

    def forward(self, input):
        if self.padding_mode == "zeros" and not isinstance(self.padding, str) and \
                ops.sign_classifier_supported(input, self.weight, self.stride, self.padding, self.dilation, self.groups):
            return ops.SignClassifierConv.apply(input, self.weight, self.bias)
        ops.note_fallback("Conv2dSignIn -> nn.Conv2d")
        return super().forward(input)


class AvgPool2dGlobal(nn.AvgPool2d):
    """``nn.AvgPool2d`` whose window is the whole image (the tail of the reference's nin / nin_gc, models/nin_gc.py:139: AvgPool2d(8) on 8 x 8 maps): one small
    gfx950 kernel instead of ATen's generic pooling kernel (32 us for a 160 KB tensor).  Any other geometry is the stock module."""

    def forward(self, input):
        pair = lambda v: (v, v) if isinstance(v, int) else tuple(v)
        from .sign_tensor import LazyBNAct
        if isinstance(input, LazyBNAct) and input._mn_value is None and input.recipe.get("kind") == "bn_tail":
            # the un-computed BatchNorm + ReLU of the last block (TailBNMixin): statistics, normalise, rectify and pool in ONE kernel
            if pair(self.kernel_size) == tuple(input.shape[2:]) and pair(self.padding) == (0, 0) and not self.ceil_mode and self.divisor_override is None:
                return ops.BNReLUGapPull.apply(input)
            input = ops.LazyBNActToFloat.apply(input)
        if (torch.is_tensor(input) and input.is_cuda and input.dtype == torch.float32 and input.dim() == 4 and type(input) is torch.Tensor and input.is_contiguous()
                and pair(self.kernel_size) == tuple(input.shape[2:]) and pair(self.padding) == (0, 0) and not self.ceil_mode and self.divisor_override is None):
            return ops.GlobalAvgPool.apply(input)
        ops.note_fallback("AvgPool2dGlobal -> nn.AvgPool2d")
        return super().forward(input)


class TailBNMixin:
    """The BatchNorm2d of the net's LAST block -- bn -> relu -> AvgPool2d over the whole map (models/nin_gc.py:136-147) -- in training mode: its output (with the ReLU
    behind it) stays un-computed (``ops.BNReLUTailLazy``) and the pool computes bn + relu + pool in one kernel per direction.  Installed by ``fuse_tail`` as a subclass
    of the module's own class (same parameters, buffers, ``state_dict``); eval mode and anything the kernel does not cover run the base class."""

    def forward(self, input):
        if (TAIL_FUSED and self.training and self.affine and self.track_running_stats and self.momentum is not None and self.running_mean is not None and
                ops.bn_tail_supported(input)):
            if self.num_batches_tracked is not None and not self.__dict__.pop("_mn_nbt_pre", False):
                self.num_batches_tracked.add_(1)
            return ops.BNReLUTailLazy.apply(input, self.weight, self.bias, self.running_mean, self.running_var, self.eps, self.momentum)
        return super().forward(input)


class ReLUTail(nn.ReLU):
    """The ``nn.ReLU`` behind a ``TailBNMixin`` BatchNorm: an un-computed ``LazyBNAct`` of kind "bn_tail" already stands for relu(bn(y)) (relu is idempotent) and passes
    through; anything else is the ordinary ReLU."""

    def forward(self, input):
        from .sign_tensor import LazyBNAct
        if isinstance(input, LazyBNAct) and input.recipe.get("kind") == "bn_tail":
            return input
        return super().forward(input)


import os as _os
TAIL_FUSED = _os.environ.get("MN_TAIL_FUSED", "1") != "0"          # A/B knob (round 6): bn + relu + global pool of the last block in one kernel per direction


def fuse_tail(model):
    """``prepare(fuse_bn_act=True)`` of the DoReFa / WbWtAb rewrites: a reference block (conv -> bn -> relu in definition order) directly in front of an
    ``AvgPool2dGlobal`` in the same ``nn.Sequential`` gets ``TailBNMixin`` / ``ReLUTail``.  Same objects, parameters, buffers and ``state_dict``."""
    from micronet_amd.quantization.wqaq.dorefa.quantize import _is_ref_block
    for seq in model.modules():
        if not isinstance(seq, nn.Sequential):
            continue
        kids = list(seq.children())
        for blk, pool in zip(kids, kids[1:]):
            if type(pool) is not AvgPool2dGlobal or not _is_ref_block(blk):
                continue
            bn, relu = getattr(blk, "bn", None), getattr(blk, "relu", None)
            if (type(bn).__name__ in ("BatchNorm2d", "BatchNorm2dReLU") and isinstance(bn, nn.BatchNorm2d) and bn.affine and bn.track_running_stats and
                    not getattr(bn, "q_out_bits", 0) and not getattr(bn, "emit_minmax", False) and
                    (type(relu) is nn.ReLU or type(relu).__name__ == "ReLUAfterFusedBN")):
                bn.__class__ = derive_class("Tail", TailBNMixin, type(bn))
                if type(relu) is nn.ReLU:
                    relu.__class__ = ReLUTail


# ------------------------------------------------------------------------------------------------ class swaps that survive pickling
_DERIVED = {}


def _mixin_by_name(mixin_module, mixin_name):
    import importlib
    return getattr(importlib.import_module(mixin_module), mixin_name)


def _rebuild_derived(prefix, mixin_module, mixin_name, base):
    """Unpickling hook: an EMPTY instance of the derived class (pickle restores its ``__dict__`` through ``nn.Module.__setstate__``)."""
    cls = derive_class(prefix, _mixin_by_name(mixin_module, mixin_name), base)
    return cls.__new__(cls)


def derive_class(prefix, mixin, base):
    """``type(prefix + base.__name__, (mixin, base))`` -- the subclass ``prepare()`` swaps a reference block's class for -- created once per (mixin, base) and
    picklable: ``torch.save(model)`` of a prepared net (the reference saves whole models: wqaq/dorefa/quant_model_test/quant_model_para.py:67,84) stores
    (prefix, mixin, base class) and rebuilds the class on load, so neither side needs the generated class to be importable by name."""
    key = (prefix, mixin, base)
    cls = _DERIVED.get(key)
    if cls is None:
        def __reduce_ex__(self, protocol, _a=(prefix, mixin.__module__, mixin.__name__, base)):
            return _rebuild_derived, _a, self.__dict__
        cls = type(prefix + base.__name__, (mixin, base), {"__module__": base.__module__, "__reduce_ex__": __reduce_ex__})
        _DERIVED[key] = cls
    return cls
