"""Binary / ternary weights and binary activations on MI355X -- same module surface as the reference's
``micronet/compression/quantization/wbwtab/quantize.py``.

  * ``WeightQuantizer`` (ref 105-149): W==3 ternary (TWN threshold 0.7*E|w|, per-channel alpha) and W==2 binary
    (in-place mean-centre + clamp of ``weight.data``, per-channel alpha) are ONE gfx950 launch each, one workgroup
    per output channel; their backward includes the autograd path through alpha.
  * ``ActivationQuantizer`` (ref 79-94) replaces ``nn.ReLU``: sign with saturating STE, one fused pass.
  * ``QuantConv2d`` (ref 152-195): the +-1 activations are contracted against the quantised weights by the MFMA
    implicit-GEMM kernels.
"""
import copy

import torch
import torch.nn as nn
from torch.autograd import Function

from micronet_amd import ops
from micronet_amd.nn import Conv2dFirst, Conv2dSignIn
from micronet_amd.sign_tensor import LazyConvOut, SignTensor

__all__ = ["BinaryActivation", "BinaryWeight", "Ternary", "ActivationQuantizer", "meancenter_clamp_convparams",
           "WeightQuantizer", "QuantConv2d", "QuantConvTranspose2d", "BatchNorm2dBinAct", "MaxPool2dSign", "SignTensor",
           "add_quant_op", "prepare"]


class BinaryActivation(Function):
    """sign(x) with 0 -> +1; backward zeroes the gradient where |x| >= 1 (ref 11-36)."""

    @staticmethod
    def forward(self, input):
        return ops.BinaryAct.forward(self, input)

    @staticmethod
    def backward(self, grad_output):
        return ops.BinaryAct.backward(self, grad_output)


class BinaryWeight(Function):
    """sign(w) with 0 -> +1, identity STE (ref 40-51)."""

    @staticmethod
    def forward(self, input):
        return ops.BinaryAct.forward(self, input)

    @staticmethod
    def backward(self, grad_output):
        return grad_output.clone()


class Ternary(Function):
    """(t, threshold) with t in {-1, 0, +1}, threshold = 0.7 * mean|w| per output channel; identity STE (ref 55-75)."""

    @staticmethod
    def forward(self, input):
        _, stats = ops.ternary_stats(input)
        thr = stats[:, 1].reshape(-1, 1, 1, 1)
        w = input.detach()
        t = torch.sign(torch.sign(w + thr) + torch.sign(w - thr))
        return t, thr

    @staticmethod
    def backward(self, grad_output, grad_threshold):
        return grad_output.clone()


class ActivationQuantizer(nn.Module):
    deploy_packed = False      # set by micronet_amd.inference.wbwtab_model_bn_fuse: this sign sits directly behind a BN-folded conv (conv -> Identity -> sign,
                               # wbwtab/bn_fuse/bn_fuse.py:36-55) of a block that ran packed in training -- the deployed graph keeps one byte per activation

    def __init__(self, A=2):
        super().__init__()
        self.A = A
        self.relu = nn.ReLU(inplace=True)

    def binary(self, input):
        return BinaryActivation.apply(input)

    def _identity_bn(self, C, device):
        """(gamma = 1, beta = 0, mean = 0, var = 1) of the fused BatchNorm+sign kernels: with eps = 0 they evaluate ((y - 0) * 1) * 1 + 0 = y in fp32, so
        ``sign(bn(y))`` IS the reference's ``sign(y)`` (0 -> +1).  Cached per (C, device); not part of the ``state_dict``."""
        cache = self.__dict__.setdefault("_mn_ident", {})
        key = (int(C), str(device))
        if key not in cache:
            one, zero = torch.ones(C, dtype=torch.float32, device=device), torch.zeros(C, dtype=torch.float32, device=device)
            cache[key] = (one, zero, zero.clone(), one.clone())
        return cache[key]

    def forward(self, input):
        if self.A == 2 and (isinstance(input, SignTensor) or getattr(input, "_mn_binarized", False)):
            return input              # the BatchNorm2dBinAct in front already produced sign(bn(x)) in its fused kernel
        if self.A == 2 and self.deploy_packed and input.is_cuda and input.dim() == 4:
            # the deployed (BN-folded) graph on the packed kernels: sign(conv(a) + b) straight from the +-1 codes (the folded conv left its output un-computed), or
            # -- behind the fp32 first conv -- sign(y) written as one byte per element; same kernels as the training graph's eval mode, identity statistics
            hw = input.shape[2] * input.shape[3]
            if isinstance(input, LazyConvOut) and input._mn_value is None:
                g_, b_, m_, v_ = self._identity_bn(input.shape[1], input.device)
                return ops.ConvBNSign.apply(input, g_, b_, m_, v_, 0.0, 0.0, False, None)
            if type(input) is torch.Tensor and input.dtype == torch.float32 and hw % 4 == 0 and input.is_contiguous():
                g_, b_, m_, v_ = self._identity_bn(input.shape[1], input.device)
                return ops.BNSign.apply(input, g_, b_, m_, v_, 0.0, 0.0, False, True, False)
        return self.binary(input) if self.A == 2 else self.relu(input)


class BatchNorm2dBinAct(nn.BatchNorm2d):
    """``nn.BatchNorm2d`` (same parameters, buffers and ``state_dict`` keys) that is immediately followed by a binary
    ``ActivationQuantizer``: on the GPU it computes ``sign(bn(x))`` in ONE fused op -- the normalised tensor is never written,
    the backward recomputes it -- and tags the result so that the ``ActivationQuantizer`` passes it through.  ``prepare()``
    installs it only where the module order guarantees that hand-off (``nn.Sequential`` parents and the reference's
    ``ConvBNReLU`` blocks, models/nin_gc.py:53-59); everywhere else the plain modules run.

    ``packed`` (set by ``prepare(packed_activations=True)``, the default): the result is a ``SignTensor`` -- logically the
    same float32 +-1 tensor, physically one byte per element, read directly by the next QuantConv2d / MaxPool2dSign; any
    other consumer sees the float32 values (micronet_amd/sign_tensor.py)."""

    packed = False

    def forward(self, input):
        hw = input.shape[2] * input.shape[3] if input.dim() == 4 else 0
        if not (input.is_cuda and input.dim() == 4 and hw % 4 == 0 and self.affine and input.dtype == torch.float32):
            ops.note_fallback("BatchNorm2dBinAct -> nn.BatchNorm2d")
            return super().forward(input)
        use_batch = self.training or self.running_mean is None
        momentum = 0.0 if self.momentum is None else self.momentum
        first = isinstance(input, LazyConvOut) and input.recipe.get("kind") == "first"          # the un-computed output of the first (un-quantised) conv
        fused = isinstance(input, LazyConvOut) and not first and (use_batch or self.track_running_stats)
        nbt = None
        if self.training and self.track_running_stats and self.num_batches_tracked is not None:
            if fused and self.momentum is not None and self.num_batches_tracked.dtype == torch.int64 and self.num_batches_tracked.is_cuda:
                nbt = self.num_batches_tracked           # incremented by the launch that forms the statistics (one tiny kernel less)
            else:
                self.num_batches_tracked.add_(1)
                if self.momentum is None:
                    momentum = 1.0 / float(self.num_batches_tracked)
        if fused:
            # the conv in front did not compute its output: conv + statistics + normalisation + sign in the fused kernels
            return ops.ConvBNSign.apply(input, self.weight, self.bias, self.running_mean if self.track_running_stats else None,
                                        self.running_var if self.track_running_stats else None, self.eps, momentum, use_batch, nbt,
                                        bool(getattr(self, "pool_next", False)))
        if first and use_batch and self.packed and input._mn_value is None:
            # conv + batch statistics (from the Gram data of the image) + normalisation + sign in one kernel: the conv output is never written
            return ops.FirstConvBNSign.apply(input, self.weight, self.bias, self.running_mean if self.track_running_stats else None,
                                             self.running_var if self.track_running_stats else None, self.eps, momentum)
        out = ops.BNSign.apply(input, self.weight, self.bias, self.running_mean if self.track_running_stats else None,
                               self.running_var if self.track_running_stats else None, self.eps, momentum, use_batch, bool(self.packed),
                               bool(getattr(input, "_mn_first_conv_out", False)))
        if not self.packed:
            out._mn_binarized = True
        return out


class MaxPool2dSign(nn.MaxPool2d):
    """``nn.MaxPool2d`` that pools packed sign activations in their int8 form (2x2 / stride 2); anything else takes the
    stock path (a SignTensor is materialised as float32 on the way)."""

    def forward(self, input):
        if not self.return_indices and ops.sign_pool_supported(input, self.kernel_size, self.stride, self.padding, self.dilation, self.ceil_mode):
            return ops.SignMaxPool2x2.apply(input)
        ops.note_fallback("MaxPool2dSign -> nn.MaxPool2d")
        return super().forward(input)


def meancenter_clamp_convparams(w):
    """In place on ``w.data``: subtract the mean over the input-channel axis, clamp to [-1, 1] (ref 98-102)."""
    ops.BinaryWeight.apply(w.data)
    return w


class WeightQuantizer(nn.Module):
    def __init__(self, W=2):
        super().__init__()
        self.W = W

    def binary(self, input):
        return BinaryWeight.apply(input)

    def ternary(self, input):
        return Ternary.apply(input)

    def forward(self, input):
        pre = self.__dict__.pop("_mn_pre", None)
        if pre is not None and pre[0] is input:
            # computed ahead (micronet_amd.train.prefetch_weight_path: all layers in one launch, or on a side stream -- then wait for it here
            # and keep its memory alive for this stream)
            if pre[2] is not None:
                cur = torch.cuda.current_stream()
                cur.wait_event(pre[2])
                pre[1].record_stream(cur)
            return pre[1]
        if self.W == 2:
            return ops.BinaryWeight.apply(input)     # mutates input.data like the reference (ref 123)
        if self.W == 3:
            return ops.TernaryWeight.apply(input)
        return input


def _forget_stored_codes(module, *args, **kwargs):
    """load_state_dict pre-hook: newly loaded weights are not known to be codes x alpha[o] any more (micronet_amd.inference re-establishes the verdict when it checks them)"""
    module.stored_codes = False          # (lazy_for_bn is structural -- set by prepare() -- and gated by `coded` in forward: it stays)


class QuantConv2d(nn.Conv2d):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 padding_mode="zeros", W=2, quant_inference=False):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias, padding_mode)
        self.quant_inference = quant_inference
        self.weight_quantizer = WeightQuantizer(W=W)
        self.in_shuffle_groups = 0     # > 1: this conv reads channel_shuffle(input, groups) (set by prepare(), see add_quant_op)
        self.lazy_for_bn = False       # True (set by prepare()): a packed BatchNorm2dBinAct consumes the output -> it may stay uncomputed
        self.stored_codes = False      # quant_inference only: the STORED weights are known to be codes x alpha[o] (set by micronet_amd.inference, which checks
                                       # them): the layer then contracts integer codes on the matrix cores like the training graph does
        self._register_load_state_dict_pre_hook(_forget_stored_codes, with_module=True)          # (a module-level function: the module stays picklable)

    def _codes_valid(self):
        if not getattr(self, "stored_codes", False):
            return False
        if self.__dict__.pop("_mn_codes_carry", False):          # a copy (deepcopy / pickle / DataParallel replica) of a module whose verdict held: same values in a
            self._mn_codes_key = (self.weight.data_ptr(), self.weight._version)          # new tensor -- re-keyed at first use
            return True
        ok = getattr(self, "_mn_codes_key", None) == (self.weight.data_ptr(), self.weight._version)
        if not ok and not getattr(self, "_mn_codes_noted", False):
            self._mn_codes_noted = True
            ops.note_fallback("wbwtab.QuantConv2d: stored_codes verdict no longer matches the weight tensor (float path)")
        return ok

    def __getstate__(self):          # copy.deepcopy / pickle / torch.save(model): the raw-pointer key does not travel, the verdict about the VALUES does
        st = dict(self.__dict__)
        key = st.pop("_mn_codes_key", None)
        st.pop("_mn_codes_noted", None)
        st["_mn_codes_carry"] = bool(st.get("stored_codes", False)) and (st.get("_mn_codes_carry", False) or key == (self.weight.data_ptr(), self.weight._version))
        return st

    def _replicate_for_data_parallel(self):          # nn.DataParallel: the replica's weight is a broadcast copy of the same values
        r = super()._replicate_for_data_parallel()
        valid = bool(getattr(self, "stored_codes", False)) and (self.__dict__.get("_mn_codes_carry", False) or
                                                                  getattr(self, "_mn_codes_key", None) == (self.weight.data_ptr(), self.weight._version))
        r.__dict__.pop("_mn_codes_key", None)
        r.__dict__["_mn_codes_carry"] = valid
        return r

    def _apply(self, fn, *args, **kwargs):          # .cuda() / .to(): same values in a new tensor -- the verdict moves with them
        was = self._codes_valid()
        out = super()._apply(fn, *args, **kwargs)
        if was:
            self._mn_codes_key = (self.weight.data_ptr(), self.weight._version)
        return out

    def forward(self, input):
        tnn_bin_weight = self.weight if self.quant_inference else self.weight_quantizer(self.weight)
        # binary / ternary weights are t * alpha[o]: the conv contracts the integer codes t on the bf16 matrix cores
        # (stored_codes is a verdict about ONE weight tensor state: data pointer + version, recorded by micronet_amd.inference.mark_stored_codes; a later
        #  `weight.data = ...` or in-place update silently invalidates it)
        stored = self._codes_valid()
        coded = self.weight_quantizer.W in (2, 3) and ((not self.quant_inference) or stored)
        return ops.qconv2d(input, tnn_bin_weight, self.bias, self.stride, self.padding, self.dilation, self.groups,
                           wdesc=(ops.WQ_TERNARY, 0, 0, 0, None) if coded else None, in_shuffle=self.in_shuffle_groups,
                           lazy_for_bn=self.lazy_for_bn and coded)


class QuantConvTranspose2d(nn.ConvTranspose2d):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, output_padding=0, dilation=1,
                 groups=1, bias=True, padding_mode="zeros", W=2, quant_inference=False):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, output_padding, groups, bias,
                         dilation, padding_mode)
        self.quant_inference = quant_inference
        self.weight_quantizer = WeightQuantizer(W=W)

    def forward(self, input):
        tnn_bin_weight = self.weight if self.quant_inference else self.weight_quantizer(self.weight)
        return ops.ConvTranspose2d.apply(input, tnn_bin_weight, self.bias, self.stride, self.padding,
                                         self.output_padding, self.groups, self.dilation)


def _ordered_parent(module):
    """True when the parent is KNOWN to call its children in definition order, so bn -> relu adjacency means bn feeds relu: ``nn.Sequential``, the
    ``ConvBNReLU`` block of the reference's own model files (models/nin.py, models/nin_gc.py -- identified by class AND defining module, not by
    name alone), or a user block that opts in with ``_mn_ordered_forward = True``."""
    return isinstance(module, nn.Sequential) or _is_ref_block(module)


def _is_ref_block(module):
    """The reference's ``ConvBNReLU`` (shuffle -> conv -> bn -> relu, models/nin_gc.py:18-59) or a block that declares the same call order."""
    t = type(module)
    return bool(getattr(module, "_mn_ordered_forward", False)) or (t.__name__ == "ConvBNReLU" and t.__module__.split(".")[-1] in ("nin", "nin_gc"))


def add_quant_op(module, layer_counter, layer_num, A=2, W=2, quant_inference=False, fuse_bn_act=True, fold_shuffle=True,
                 packed_activations=True, fuse_conv_bn=True):
    """Quantise conv k iff 1 < k < layer_num; every ReLU met while 0 < k < layer_num becomes the binary activation
    (ref 247-331).  With ``fuse_bn_act`` a plain BatchNorm2d directly in front of such a binary activation is switched to
    ``BatchNorm2dBinAct`` (same object, same state: only its class changes)."""
    prev = prev2 = None
    for name, child in module.named_children():
        if isinstance(child, nn.Conv2d):
            layer_counter[0] += 1
            if 1 < layer_counter[0] < layer_num:
                new = QuantConv2d(child.in_channels, child.out_channels, child.kernel_size, stride=child.stride,
                                  padding=child.padding, dilation=child.dilation, groups=child.groups,
                                  bias=child.bias is not None, padding_mode=child.padding_mode, W=W,
                                  quant_inference=quant_inference)
                if child.bias is not None:
                    new.bias.data = child.bias
                new.weight.data = child.weight
                module._modules[name] = new
                # the reference's ConvBNReLU block shuffles its input in front of the conv (models/nin_gc.py:53-56):
                # hand the permutation to the conv, which folds it into its channel addressing (no copy of the tensor)
                if fold_shuffle and _is_ref_block(module) and getattr(module, "channel_shuffle_flag", 0) \
                        and getattr(module, "shuffle_groups", 1) > 1 and child.in_channels % module.shuffle_groups == 0:
                    new.in_shuffle_groups = int(module.shuffle_groups)
                    module.channel_shuffle_flag = 0
            elif layer_counter[0] == 1 and type(child) is nn.Conv2d:
                child.__class__ = Conv2dFirst       # the un-quantised first conv: same object and state, gfx950 kernels when covered
            elif layer_counter[0] == layer_num and type(child) is nn.Conv2d and packed_activations:
                child.__class__ = Conv2dSignIn      # the un-quantised last conv reads the packed +-1 output of the block in front
        elif isinstance(child, nn.ConvTranspose2d):
            layer_counter[0] += 1
            if 1 < layer_counter[0] < layer_num:
                new = QuantConvTranspose2d(child.in_channels, child.out_channels, child.kernel_size, stride=child.stride,
                                           padding=child.padding, output_padding=child.output_padding,
                                           dilation=child.dilation, groups=child.groups, bias=child.bias is not None,
                                           padding_mode=child.padding_mode, W=W, quant_inference=quant_inference)
                if child.bias is not None:
                    new.bias.data = child.bias
                new.weight.data = child.weight
                module._modules[name] = new
        elif isinstance(child, nn.ReLU):
            if 0 < layer_counter[0] < layer_num:
                module._modules[name] = ActivationQuantizer(A=A)
                if fuse_bn_act and A == 2 and type(prev) is nn.BatchNorm2d and prev.affine and _ordered_parent(module):
                    prev.__class__ = BatchNorm2dBinAct
                    prev.packed = bool(packed_activations)
                    if packed_activations and fuse_conv_bn and isinstance(prev2, (QuantConv2d, Conv2dFirst)):
                        prev2.lazy_for_bn = True        # conv -> bn -> sign in definition order: the conv output need never be stored
        elif type(child) is nn.MaxPool2d and packed_activations and A == 2:
            child.__class__ = MaxPool2dSign       # same object and state; pools SignTensors without unpacking them
            two = lambda v: v in (2, (2, 2), [2, 2])
            if prev is not None and _is_ref_block(prev) and type(getattr(prev, "bn", None)) is BatchNorm2dBinAct and _ordered_parent(module) and two(child.kernel_size) and \
                    two(child.stride) and child.padding in (0, (0, 0)) and child.dilation in (1, (1, 1)) and not child.ceil_mode:
                prev.bn.pool_next = True          # the block's sign pass writes the pooled codes as well (ops.ConvBNSign): this pool only hands them on
        elif type(child) is nn.AvgPool2d and fuse_bn_act:
            from micronet_amd.nn import AvgPool2dGlobal
            child.__class__ = AvgPool2dGlobal     # same object and state; its own kernel only when the window is the whole image
        else:
            add_quant_op(child, layer_counter, layer_num, A=A, W=W, quant_inference=quant_inference, fuse_bn_act=fuse_bn_act,
                         fold_shuffle=fold_shuffle, packed_activations=packed_activations, fuse_conv_bn=fuse_conv_bn)
        prev2, prev = prev, module._modules[name]


def prepare(model, inplace=False, A=2, W=2, quant_inference=False, fuse_bn_act=True, fold_shuffle=True, packed_activations=True,
            fuse_conv_bn=True):
    """Same rewrite as the reference (ref 334-347); ``fuse_bn_act`` (ours, default on) additionally fuses BatchNorm2d with
    the binary activation that follows it (see ``BatchNorm2dBinAct``), and ``fold_shuffle`` (ours, default on) moves the
    channel shuffle of a ``ConvBNReLU`` block into its quantised conv's addressing -- numerically the same function;
    ``packed_activations`` (ours, default on) lets the fused BN+sign hand its +-1 output to the next conv / max-pool as one
    byte per element (``SignTensor``), float32 for everybody else; ``fuse_conv_bn`` (ours, default on) lets a quantised
    conv whose packed input and BatchNorm2dBinAct consumer are both ours skip writing its output (``LazyConvOut``): the fused
    kernels recompute it on the matrix cores."""
    if not inplace:
        model = copy.deepcopy(model)
    layer_num = sum(isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)) for m in model.modules())
    add_quant_op(model, [0], layer_num, A=A, W=W, quant_inference=quant_inference, fuse_bn_act=fuse_bn_act, fold_shuffle=fold_shuffle,
                 packed_activations=packed_activations, fuse_conv_bn=fuse_conv_bn)
    if fuse_bn_act:
        from micronet_amd.nn import fuse_tail
        fuse_tail(model)          # bn -> relu -> global average pool of the last block: one kernel per direction (TailBNMixin)
    return model
