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
        return ops.DorefaWeight.apply(input, self.w_bits)


def _wdesc(module):
    """weight-code descriptor for the code-domain conv kernels: w = (2k - n)/n, n = 2^w_bits - 1"""
    b = module.weight_quantizer.w_bits
    return (ops.WQ_DOREFA, b, 0, 0, None) if (not module.quant_inference and 2 <= b <= 8) else None


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

    def forward(self, input):
        quant_weight = self.weight if self.quant_inference else self.weight_quantizer(self.weight)
        mode, bits = _aq_args(self.activation_quantizer)
        # like the reference, the forward always zero-pads whatever padding_mode says (ref 113-121)
        return ops.qconv2d(input, quant_weight, self.bias, self.stride, self.padding, self.dilation, self.groups,
                           aq_mode=mode, aq_bits=bits, wdesc=_wdesc(self))


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


class BatchNorm2dReLU(nn.BatchNorm2d):
    """``nn.BatchNorm2d`` whose forward also applies the ReLU behind it (same parameters, buffers and ``state_dict`` keys): one fused
    gfx950 op (ops.BNReLU: three streaming passes forward, five backward, z never stored) instead of MIOpen's BatchNorm kernels plus
    separate ReLU forward / backward kernels.  Installed by ``prepare(fuse_bn_act=True)`` in front of a ReLU that the parent calls right
    after it; that ReLU becomes a ``ReLUAfterFusedBN`` (a no-op ``nn.ReLU``)."""

    def forward(self, input):
        from micronet_amd import ops
        import torch.nn.functional as F
        if not (self.affine and ops.bnrelu_supported(input)):
            return F.relu(super().forward(input))
        use_batch = self.training or self.running_mean is None
        momentum = 0.0 if self.momentum is None else self.momentum
        if self.training and self.track_running_stats and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(1)
            if self.momentum is None:
                momentum = 1.0 / float(self.num_batches_tracked)
        return ops.BNReLU.apply(input, self.weight, self.bias, self.running_mean if self.track_running_stats else None,
                                self.running_var if self.track_running_stats else None, self.eps, momentum, use_batch)


class MaxPool2dF32(nn.MaxPool2d):
    """``nn.MaxPool2d`` whose 2x2 / stride-2 case runs on the gfx950 kernels (byte argmax, scatter backward); anything else is the stock module."""

    def forward(self, input):
        from micronet_amd import ops
        if not self.return_indices and ops.f32_pool_supported(input, self.kernel_size, self.stride, self.padding, self.dilation, self.ceil_mode):
            return ops.MaxPool2x2F32.apply(input)
        return super().forward(input)


class ReLUAfterFusedBN(nn.ReLU):
    """The ``nn.ReLU`` behind a ``BatchNorm2dReLU``: the rectification already happened in the fused op (relu is idempotent, so this is the
    same function), the module stays in place (``isinstance(m, nn.ReLU)``, module names) and costs no kernel."""

    def forward(self, input):
        return input


def _ordered_parent(module):
    """True when the parent calls its children in definition order, so bn -> relu adjacency means bn feeds relu."""
    return isinstance(module, nn.Sequential) or type(module).__name__ == "ConvBNReLU"


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


def prepare(model, inplace=False, a_bits=8, w_bits=8, quant_inference=False, fuse_bn_act=True):
    """Same rewrite as the reference (ref 312-320).  ``fuse_bn_act`` (ours, default on): a ``BatchNorm2d`` directly in front of a ``ReLU`` in a
    block that calls them in that order becomes ``BatchNorm2dReLU`` (one fused op) and the ReLU a no-op subclass; with it off the module
    graph is exactly the reference's."""
    if not inplace:
        model = copy.deepcopy(model)
    add_quant_op(model, [0], a_bits=a_bits, w_bits=w_bits, quant_inference=quant_inference, fuse_bn_act=fuse_bn_act)
    return model
