"""Packed binary activations.

In the reference's W/A-binary nets (wbwtab) every tensor between ``BinaryActivation`` and the next convolution holds only
+-1 (wbwtab/quantize.py:13-19), yet travels as fp32: 4 bytes per element written by the activation and read again by the
convolution's forward and by its backward-weight.  ``SignTensor`` is that tensor with a one-byte physical representation:

  * logically a float32 NCHW tensor (shape / dtype / device / autograd behave as such),
  * physically an int8 tensor of codes in {-1, +1} (``.codes``) produced by ``mn_bnsign_fwd_i8`` and consumed directly by
    the code-domain conv kernels (``MN_ACTQ_SIGN8``) and by the sign max-pool -- a quarter of the HBM traffic.

It is a wrapper subclass: operators of this package read ``.codes`` and never touch fp32; ANY other torch operator
that receives a SignTensor (a hook, a plain ``nn.Conv2d`` such as the un-quantised last layer, ``.cpu()`` ...) goes through
``__torch_dispatch__``, which materialises the float32 values first -- so foreign consumers see exactly the tensor the
reference produces, gradients included.  Safe by construction; fast only between our own modules.
"""
import torch
from torch.utils._pytree import tree_map

_ALIAS_OPS = None


def _alias_ops():
    global _ALIAS_OPS
    if _ALIAS_OPS is None:
        a = torch.ops.aten
        _ALIAS_OPS = {a.detach.default, a.alias.default}
    return _ALIAS_OPS


class SignTensor(torch.Tensor):
    @staticmethod
    def __new__(cls, codes):
        if codes.dtype != torch.int8 or not codes.is_contiguous():
            raise TypeError("SignTensor wraps a contiguous int8 tensor of +-1 codes")
        r = torch.Tensor._make_wrapper_subclass(cls, codes.shape, dtype=torch.float32, device=codes.device, requires_grad=False)
        r._mn_codes = codes
        return r

    def __init__(self, codes):
        pass

    @property
    def codes(self):
        return self._mn_codes

    def to_float(self):
        """The float32 tensor the reference holds at this point (no autograd link; use ops.sign_to_float for one)."""
        return self._mn_codes.to(torch.float32)

    def __repr__(self):
        return "SignTensor(shape=%s, device=%s)" % (tuple(self.shape), self.device)

    __torch_function__ = torch._C._disabled_torch_function_impl

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _alias_ops() and isinstance(args[0], SignTensor):
            return SignTensor(args[0]._mn_codes)
        un = lambda t: t._mn_codes.to(torch.float32) if isinstance(t, SignTensor) else t
        return func(*tree_map(un, args), **tree_map(un, kwargs))


class LazyConvOut(torch.Tensor):
    """The output of a binary-weight convolution on a SignTensor that has NOT been computed.

    Between ``QuantConv2d`` and the ``BatchNorm2dBinAct`` that follows it the convolution result y (fp32, the largest tensor
    of the block) would be written once and read back three times.  Its input is one byte per element and its weights are
    ternary codes, so y is cheaper to RECOMPUTE on the matrix cores than to move: the conv module returns this wrapper (shape /
    dtype / device / autograd of y, plus the recipe: input codes, fake-quantised weights, bias, geometry) and the fused
    BatchNorm+sign kernels (``mn_qconv_bnsign_*``) consume the recipe directly.  Any other consumer triggers
    ``__torch_dispatch__``, which runs the convolution kernel first -- it then sees exactly the tensor the reference produces."""

    @staticmethod
    def __new__(cls, shape, device, recipe):
        r = torch.Tensor._make_wrapper_subclass(cls, shape, dtype=torch.float32, device=device, requires_grad=False)
        r._mn_recipe = recipe
        r._mn_value = None
        return r

    def __init__(self, shape, device, recipe):
        pass

    @property
    def recipe(self):
        return self._mn_recipe

    def materialize(self):
        if self._mn_value is None:
            self._mn_value = self._mn_recipe["compute"]()
        return self._mn_value

    def __repr__(self):
        return "LazyConvOut(shape=%s, device=%s)" % (tuple(self.shape), self.device)

    __torch_function__ = torch._C._disabled_torch_function_impl

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _alias_ops() and isinstance(args[0], LazyConvOut):
            r = LazyConvOut(args[0].shape, args[0].device, args[0]._mn_recipe)
            r._mn_value = args[0]._mn_value
            return r
        un = lambda t: t.materialize() if isinstance(t, LazyConvOut) else (t._mn_codes.to(torch.float32) if isinstance(t, SignTensor) else t)
        return func(*tree_map(un, args), **tree_map(un, kwargs))


class LazyPoolGrad(torch.Tensor):
    """The gradient of a 2x2 max-pool's INPUT that has not been expanded: logically the full-size float32 tensor, physically the
    pooled gradient plus the pool's input codes.  ``ConvBNSign.backward`` (the block in front of the pool) reads it in this
    form -- a quarter of the bytes, no full-size write -- and routes it through the pool inside its kernels
    (``mn_qconv_bnsign_bwd_pooled``); any other consumer materialises it with the sign max-pool backward kernel."""

    @staticmethod
    def __new__(cls, shape, device, pooled_grad, codes, expand):
        r = torch.Tensor._make_wrapper_subclass(cls, shape, dtype=torch.float32, device=device, requires_grad=False)
        r._mn_pg, r._mn_codes, r._mn_expand, r._mn_value = pooled_grad, codes, expand, None
        return r

    def __init__(self, shape, device, pooled_grad, codes, expand):
        pass

    def materialize(self):
        if self._mn_value is None:
            self._mn_value = self._mn_expand(self._mn_pg, self._mn_codes)
        return self._mn_value

    def __repr__(self):
        return "LazyPoolGrad(shape=%s, device=%s)" % (tuple(self.shape), self.device)

    __torch_function__ = torch._C._disabled_torch_function_impl

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _alias_ops() and isinstance(args[0], LazyPoolGrad):
            a = args[0]
            r = LazyPoolGrad(a.shape, a.device, a._mn_pg, a._mn_codes, a._mn_expand)
            r._mn_value = a._mn_value
            return r
        un = lambda t: t.materialize() if isinstance(t, (LazyPoolGrad, LazyConvOut)) else (t._mn_codes.to(torch.float32) if isinstance(t, SignTensor) else t)
        return func(*tree_map(un, args), **tree_map(un, kwargs))


class LazyBNGrad(torch.Tensor):
    """d loss / d y of a BatchNorm2d + BinaryActivation whose input y is the output of the FIRST convolution: logically the full-size
    float32 gradient, physically (da, y, statistics, sums).  The first layer has no backward-data, so its backward-weight is the only
    consumer: it forms dy in registers while da and y stream in (``mn_conv2d_bwd_weight_first_bn``); any other consumer
    materialises it with the BatchNorm+sign backward kernel."""

    @staticmethod
    def __new__(cls, shape, device, recipe, expand):
        r = torch.Tensor._make_wrapper_subclass(cls, shape, dtype=torch.float32, device=device, requires_grad=False)
        r._mn_recipe, r._mn_expand, r._mn_value = recipe, expand, None
        return r

    def __init__(self, shape, device, recipe, expand):
        pass

    def materialize(self):
        if self._mn_value is None:
            self._mn_value = self._mn_expand(self._mn_recipe)
        return self._mn_value

    def __repr__(self):
        return "LazyBNGrad(shape=%s, device=%s)" % (tuple(self.shape), self.device)

    __torch_function__ = torch._C._disabled_torch_function_impl

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _alias_ops() and isinstance(args[0], LazyBNGrad):
            a = args[0]
            r = LazyBNGrad(a.shape, a.device, a._mn_recipe, a._mn_expand)
            r._mn_value = a._mn_value
            return r
        un = lambda t: t.materialize() if isinstance(t, (LazyBNGrad, LazyPoolGrad, LazyConvOut)) else (t._mn_codes.to(torch.float32) if isinstance(t, SignTensor) else t)
        return func(*tree_map(un, args), **tree_map(un, kwargs))


class QActTensor(torch.Tensor):
    """The output of a k-bit (DoReFa) block -- relu(bn(conv(x))), possibly max-pooled -- whose only consumer is the activation quantizer of the
    next QuantConv2d: logically the float32 activation a (shape / dtype / device / autograd behave as such), physically the CODES j of that
    quantizer (uint8, one byte per element: what the next conv's kernels read) plus a recipe that re-creates a exactly from the producer's 16-bit
    stash when anybody else asks (``materialize``: one streaming kernel).  Any torch operator outside this package goes through
    ``__torch_dispatch__`` and sees the float32 activation the reference holds at this point."""

    @staticmethod
    def __new__(cls, codes, bits, materialize, pooled=False, f32=None):
        if codes.dtype != torch.uint8 or not codes.is_contiguous():
            raise TypeError("QActTensor wraps a contiguous uint8 tensor of activation codes")
        r = torch.Tensor._make_wrapper_subclass(cls, codes.shape, dtype=torch.float32, device=codes.device, requires_grad=False)
        r._mn_codes, r._mn_bits, r._mn_mat, r._mn_value = codes, int(bits), materialize, None
        r._mn_pooled = bool(pooled)       # the producing block already applied the 2x2 max-pool behind it (the pool module then passes the codes through)
        r._mn_f32 = f32                   # the fp32 activation as a SECOND autograd output of the producer (the next residual block's identity shortcut)
        return r

    def __init__(self, codes, bits, materialize, pooled=False, f32=None):
        pass

    @property
    def codes(self):
        return self._mn_codes

    @property
    def bits(self):
        return self._mn_bits

    def materialize(self):
        if self._mn_value is None:
            self._mn_value = self._mn_mat()
        return self._mn_value

    def __repr__(self):
        return "QActTensor(shape=%s, bits=%d, device=%s)" % (tuple(self.shape), self._mn_bits, self.device)

    __torch_function__ = torch._C._disabled_torch_function_impl

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _alias_ops() and isinstance(args[0], QActTensor):
            a = args[0]
            r = QActTensor(a._mn_codes, a._mn_bits, a._mn_mat, a._mn_pooled, a._mn_f32)
            r._mn_value = a._mn_value
            return r
        return func(*tree_map(_unwrap_any, args), **tree_map(_unwrap_any, kwargs))


class QGrad(torch.Tensor):
    """d loss / d (QUANTISED activation) handed back by a conv that read a ``QActTensor``'s codes: logically the gradient w.r.t. the activation itself
    (i.e. after the quantizer's clip-STE, wqaq/dorefa/quantize.py:36-46), physically the raw gradient dq -- the producing block applies the STE inside
    its streaming backward kernels, where it recomputes the activation anyway.  Any other consumer (e.g. autograd adding a second gradient)
    materialises STE(dq) first."""

    @staticmethod
    def __new__(cls, dq, expand):
        r = torch.Tensor._make_wrapper_subclass(cls, dq.shape, dtype=torch.float32, device=dq.device, requires_grad=False)
        r._mn_dq, r._mn_expand, r._mn_value = dq, expand, None
        r._mn_dq2 = None                  # a second raw gradient from another conv that read the same codes (QConvCodeLazy2); ``expand`` accounts for it
        return r

    def __init__(self, dq, expand):
        pass

    def materialize(self):
        if self._mn_value is None:
            self._mn_value = self._mn_expand(self._mn_dq)
        return self._mn_value

    def __repr__(self):
        return "QGrad(shape=%s, device=%s)" % (tuple(self.shape), self.device)

    __torch_function__ = torch._C._disabled_torch_function_impl

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _alias_ops() and isinstance(args[0], QGrad):
            a = args[0]
            r = QGrad(a._mn_dq, a._mn_expand)
            r._mn_value, r._mn_dq2 = a._mn_value, a._mn_dq2
            return r
        return func(*tree_map(_unwrap_any, args), **tree_map(_unwrap_any, kwargs))


class LazyQConvOut(torch.Tensor):
    """The output of a DoReFa QuantConv2d on a ``QActTensor`` that has NOT been computed (the k-bit analogue of ``LazyConvOut``): the fused
    BatchNorm+ReLU+quantizer behind it consumes the recipe (input codes, fake-quantised weights, bias, geometry) and never needs fp32 y.  Any other
    consumer gets the convolution computed by the ordinary kernels on the materialised activation."""

    @staticmethod
    def __new__(cls, shape, device, recipe):
        r = torch.Tensor._make_wrapper_subclass(cls, shape, dtype=torch.float32, device=device, requires_grad=False)
        r._mn_recipe, r._mn_value = recipe, None
        return r

    def __init__(self, shape, device, recipe):
        pass

    @property
    def recipe(self):
        return self._mn_recipe

    def materialize(self):
        if self._mn_value is None:
            self._mn_value = self._mn_recipe["compute"]()
        return self._mn_value

    def __repr__(self):
        return "LazyQConvOut(shape=%s, device=%s)" % (tuple(self.shape), self.device)

    __torch_function__ = torch._C._disabled_torch_function_impl

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _alias_ops() and isinstance(args[0], LazyQConvOut):
            a = args[0]
            r = LazyQConvOut(a.shape, a.device, a._mn_recipe)
            r._mn_value = a._mn_value
            return r
        return func(*tree_map(_unwrap_any, args), **tree_map(_unwrap_any, kwargs))


class LazyReluConvOut(torch.Tensor):
    """The output of a BN-fused IAO convolution whose block applies a ReLU right behind it (``ConvBNReLU`` with ``bn = nn.Identity``, models/nin_gc.py:53-59 after
    the rewrite of wqaq/iao/quantize.py:1567-1624): LOGICALLY the convolution's output (what the reference's ``QuantBNFuseConv2d.forward`` returns), PHYSICALLY the
    already rectified tensor ``a = relu(out)`` the kernel's epilogue wrote.  The block's ReLU module (``ReLUAfterFusedConv``) takes ``a`` out of the wrapper; any other
    consumer -- a hook, a direct call of the conv module -- gets the un-rectified output, recomputed by the same kernel without the epilogue (``recipe['compute']``)."""

    @staticmethod
    def __new__(cls, a, recipe):
        r = torch.Tensor._make_wrapper_subclass(cls, a.shape, dtype=torch.float32, device=a.device, requires_grad=False)
        r._mn_a, r._mn_recipe, r._mn_value = a, recipe, None
        return r

    def __init__(self, a, recipe):
        pass

    def materialize(self):
        if self._mn_value is None:
            self._mn_value = self._mn_recipe["compute"]()
        return self._mn_value

    def __repr__(self):
        return "LazyReluConvOut(shape=%s, device=%s)" % (tuple(self.shape), self.device)

    __torch_function__ = torch._C._disabled_torch_function_impl

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _alias_ops() and isinstance(args[0], LazyReluConvOut):
            a = args[0]
            r = LazyReluConvOut(a._mn_a, a._mn_recipe)
            r._mn_value = a._mn_value
            return r
        return func(*tree_map(_unwrap_any, args), **tree_map(_unwrap_any, kwargs))


class LazyReluGrad(torch.Tensor):
    """The gradient a fused ReLU hands back to the convolution in front of it: logically ``g * [a > 0]``, physically ``g`` (+ whether the consumer of ``a`` already
    applied that mask in its own backward-data kernel).  The convolution's backward unpacks it; any other consumer materialises the masked gradient."""

    @staticmethod
    def __new__(cls, g, a, premasked):
        r = torch.Tensor._make_wrapper_subclass(cls, g.shape, dtype=torch.float32, device=g.device, requires_grad=False)
        r._mn_g, r._mn_a, r._mn_premasked, r._mn_value = g, a, bool(premasked), None
        return r

    def __init__(self, g, a, premasked):
        pass

    def materialize(self):
        if self._mn_value is None:
            self._mn_value = self._mn_g if self._mn_premasked else torch.where(self._mn_a > 0, self._mn_g, torch.zeros((), dtype=self._mn_g.dtype, device=self._mn_g.device))
        return self._mn_value

    def __repr__(self):
        return "LazyReluGrad(shape=%s, device=%s)" % (tuple(self.shape), self.device)

    __torch_function__ = torch._C._disabled_torch_function_impl

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _alias_ops() and isinstance(args[0], LazyReluGrad):
            a = args[0]
            r = LazyReluGrad(a._mn_g, a._mn_a, a._mn_premasked)
            r._mn_value = a._mn_value
            return r
        return func(*tree_map(_unwrap_any, args), **tree_map(_unwrap_any, kwargs))


class LazyBNAct(torch.Tensor):
    """The output of a training-mode BatchNorm2d [+ ReLU] behind a dense IAO convolution (models/resnet.py:17-29) that has NOT been computed: logically the float32
    activation a = act(bn(y)), physically a recipe (y, the BatchNorm's parameters, the conv's epilogue statistics).  Its consumer -- the next dense IAO ``QuantConv2d``
    (wqaq/iao/quantize.py:492-507) or the block's ``QuantAdd`` (:1484-1498) -- PULLS: ``prep()`` finishes the batch statistics and hands back the per-channel (min,
    max) of ``a`` for the consumer's observer (mn_bn_acc_prep: no pass over y), then ONE kernel normalises, rectifies and applies the consumer's quantizer
    (mn_bn_apply_codes / mn_iao_qadd_bn_fwd); fp32 ``a`` is never written.  Any other consumer goes through ``__torch_dispatch__`` and sees the float32 activation."""

    @staticmethod
    def __new__(cls, shape, device, recipe):
        r = torch.Tensor._make_wrapper_subclass(cls, shape, dtype=torch.float32, device=device, requires_grad=False)
        r._mn_recipe, r._mn_value = recipe, None
        return r

    def __init__(self, shape, device, recipe):
        pass

    @property
    def recipe(self):
        return self._mn_recipe

    def prep(self):
        """(mm, count): per-channel minima / maxima of the activation (a partials buffer); the first call also finishes save = {mean, invstd} and the running statistics"""
        return self._mn_recipe["prep"]()

    def materialize(self):
        if self._mn_value is None:
            self._mn_value = self._mn_recipe["compute"]()
        return self._mn_value

    def __repr__(self):
        return "LazyBNAct(shape=%s, device=%s)" % (tuple(self.shape), self.device)

    __torch_function__ = torch._C._disabled_torch_function_impl

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func in _alias_ops() and isinstance(args[0], LazyBNAct):
            a = args[0]
            r = LazyBNAct(a.shape, a.device, a._mn_recipe)
            r._mn_value = a._mn_value
            return r
        return func(*tree_map(_unwrap_any, args), **tree_map(_unwrap_any, kwargs))


def _unwrap_any(t):
    """The plain tensor a foreign operator should see for any wrapper of this module."""
    if isinstance(t, (QActTensor, QGrad, LazyQConvOut, LazyBNGrad, LazyPoolGrad, LazyConvOut, LazyReluConvOut, LazyReluGrad, LazyBNAct)):
        return t.materialize()
    if isinstance(t, SignTensor):
        return t._mn_codes.to(torch.float32)
    return t
