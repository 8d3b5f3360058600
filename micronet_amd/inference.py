"""Inference-graph tooling: turning a QAT model into the graph that is deployed -- the MI355X counterpart of the reference's
``wbwtab/bn_fuse/bn_fuse.py:20-107``, ``wqaq/iao/bn_fuse/bn_fuse.py:20-80`` and the pre-quantisation loop of
``wqaq/dorefa/quant_model_test/quant_model_test.py:189-191``.

  * ``prequantize_weights``: ``m.weight.data = m.weight_quantizer(m.weight)`` for every ``quant_inference=True`` layer (the stored weights ARE the
    fake-quantised ones; the forward then skips the weight quantizer);
  * ``wbwtab_model_bn_fuse``: BatchNorm folded into the convolution in front of it.  In front of a BINARY activation the fold needs no multiplier --
    sign(gamma * (y - mean) / std + beta) = sign(+-(y - mean + beta * std / gamma)) -- so the folded weights stay ternary / binary codes x alpha and
    only a bias (and the weights' sign where gamma < 0) changes (ref 36-55); elsewhere the ordinary w * gamma / std fold (ref 56-59);
  * ``iao_model_bn_fuse``: ``QuantBNFuseConv2d`` -> ``QuantConv2d(quant_inference=True)`` with w * gamma / std, beta + (b - mean) * gamma / std and the
    trained quantizer scales / zero points copied over (ref 20-66).
The results are ordinary modules of this package: in eval mode they run on the same gfx950 kernels (activation codes, integer accumulators).
``tests/test_gpu_inference.py`` checks train-graph == inference-graph on the same batch, as the reference's ``*_test.py`` scripts do."""
import copy

import torch
import torch.nn as nn


def mark_stored_codes(m):
    """Record that the weights m.weight holds RIGHT NOW are codes x alpha[o] (wbwtab QuantConv2d with quant_inference=True): the verdict is tied to the tensor's data
    pointer and version, so `weight.data = ...`, an in-place update or load_state_dict (pre-hook) drop it; Module._apply (.cuda(), .to()) carries it over."""
    m.stored_codes = True
    m._mn_codes_key = (m.weight.data_ptr(), m.weight._version)


@torch.no_grad()
def prequantize_weights(model):
    """For every layer built with ``quant_inference=True``: store the fake-quantised weights (what ``quant_model_test.py:189-191`` does after loading)."""
    n = 0
    for m in model.modules():
        if getattr(m, "quant_inference", False) and hasattr(m, "weight_quantizer") and hasattr(m, "weight"):
            was = m.weight_quantizer.training
            m.weight_quantizer.eval()                  # IAO: do not move the trained observer / scale
            m.weight.data = m.weight_quantizer(m.weight).detach()
            m.weight_quantizer.train(was)
            n += 1
            if type(m).__module__.endswith("wbwtab.quantize") and hasattr(m, "stored_codes") and getattr(m.weight_quantizer, "W", 0) in (2, 3):
                mark_stored_codes(m)           # the stored weights ARE the quantizer's output t * alpha[o]: the layer keeps contracting integer codes
            # DoReFa layers: record the "stored weights lie on the quantizer grid" verdict now (one host sync per layer, here instead of inside the first forward --
            # which may be a captured one)
            if type(m).__module__.endswith("dorefa.quantize") and m.weight.is_cuda and not torch.cuda.is_current_stream_capturing():
                from micronet_amd.quantization.wqaq.dorefa.quantize import _weight_is_coded
                m.__dict__.pop("_mn_grid", None)
                _weight_is_coded(m)
    return n


def _conv_like(conv, cls, **kw):
    return cls(conv.in_channels, conv.out_channels, conv.kernel_size, stride=conv.stride, padding=conv.padding, dilation=conv.dilation,
               groups=conv.groups, bias=True, padding_mode=conv.padding_mode, **kw)


@torch.no_grad()
def wbwtab_model_bn_fuse(model, W=2, inplace=False):
    """ref wbwtab/bn_fuse/bn_fuse.py:20-107.  ``model``: prepared with ``quant_inference=True`` and its weights pre-quantised or not (the fold acts on
    whatever ``conv.weight`` holds, as the reference does).  BN layers are counted in module order; the first ``bin_bn_fuse_num`` of them (= the
    number of binary ``ActivationQuantizer``s) sit in front of a binary activation."""
    from micronet_amd.quantization.wbwtab import quantize
    if not inplace:
        model = copy.deepcopy(model)
    bin_bn_fuse_num = sum(isinstance(m, quantize.ActivationQuantizer) for m in model.modules())
    counter = [0]

    def fuse(conv, bn):
        counter[0] += 1
        k = counter[0]
        mean, std, gamma, beta = bn.running_mean, torch.sqrt(bn.running_var + bn.eps), bn.weight, bn.bias
        w = conv.weight
        b = conv.bias if conv.bias is not None else mean.new_zeros(mean.shape)
        if 1 <= k <= bin_bn_fuse_num:
            w_f, b_f = w.clone(), b.clone()
            pos, neg = gamma.gt(0), gamma.lt(0)
            b_f[pos] = b[pos] - mean[pos] + beta[pos] * (std[pos] / gamma[pos])
            w_f[neg] = w[neg] * -1
            b_f[neg] = mean[neg] - b[neg] - beta[neg] * (std[neg] / gamma[neg])
        else:
            w_f = w * (gamma / std).reshape([conv.out_channels, 1, 1, 1])
            b_f = beta + (b - mean) * (gamma / std)
        if 2 <= k <= bin_bn_fuse_num:
            new = _conv_like(conv, quantize.QuantConv2d, W=W, quant_inference=True)
            new.in_shuffle_groups = getattr(conv, "in_shuffle_groups", 0)
            # The low-bit deployed path: in front of a binary activation the fold leaves the weights codes x alpha[o] (only signs and the bias change, ref 36-55).
            # Checked here, once: every non-zero |w| of an output channel equals the channel's maximum -- then the layer contracts the +-1 input codes against
            # integer weight codes on the matrix cores and (where prepare() had established conv -> bn -> sign in this order: ``lazy_for_bn``) hands its
            # un-computed result to the sign behind it, exactly like the training graph in eval mode: one byte per activation end to end.
            mag = w_f.detach().abs().flatten(1)
            coded = bool(((mag == 0) | (mag == mag.amax(1, keepdim=True))).all()) and W in (2, 3)
            new.lazy_for_bn = bool(coded and getattr(conv, "lazy_for_bn", False))
        else:
            new = _conv_like(conv, nn.Conv2d)
            from micronet_amd.nn import Conv2dFirst, Conv2dSignIn
            if type(conv) in (Conv2dFirst, Conv2dSignIn):
                new.__class__ = type(conv)          # the fp32 first / last conv keep their gfx950 kernels (same parameters: only the forward differs)
        new = new.to(w.device)
        new.weight.data, new.bias.data = w_f, b_f
        if 2 <= k <= bin_bn_fuse_num and coded:
            mark_stored_codes(new)
        return new

    def walk(module):
        last, packed_bn = None, False
        for name, child in module.named_children():
            if isinstance(child, nn.Conv2d):
                last = (name, child)
            elif isinstance(child, nn.BatchNorm2d):
                module._modules[last[0]] = fuse(last[1], child)
                module._modules[name] = nn.Identity()
                packed_bn = isinstance(child, quantize.BatchNorm2dBinAct) and bool(child.packed)      # (prepare() established bn -> sign adjacency for this block)
            elif isinstance(child, quantize.ActivationQuantizer):
                if packed_bn and child.A == 2:
                    child.deploy_packed = True      # conv -> Identity -> sign on the packed kernels (ActivationQuantizer.forward)
                packed_bn = False
            else:
                packed_bn = False
                walk(child)
    walk(model)
    return model


@torch.no_grad()
def iao_model_bn_fuse(model, inplace=False):
    """ref wqaq/iao/bn_fuse/bn_fuse.py:20-80: every ``QuantBNFuseConv2d`` becomes a ``QuantConv2d(quant_inference=True)`` holding the folded (not yet
    quantised) weights and the trained quantizer state; follow with ``prequantize_weights``."""
    from micronet_amd.quantization.wqaq.iao import quantize
    if not inplace:
        model = copy.deepcopy(model)

    def fuse(m):
        mean, std = m.running_mean, torch.sqrt(m.running_var + m.eps)
        b = m.bias if m.bias is not None else mean.new_zeros(mean.shape)
        aq, wq = m.activation_quantizer, m.weight_quantizer
        q_level = 0 if getattr(wq.observer, "q_level", "L") != "L" else 1          # per-channel iff the weight observer is (an out_channels == 1 conv has ONE scale either way)
        new = _conv_like(m, quantize.QuantConv2d, a_bits=aq.bits, w_bits=wq.bits, q_type=wq._q_type_static, q_level=q_level, quant_inference=True).to(m.weight.device)
        new.weight.data = m.weight * (m.gamma / std).reshape([m.out_channels, 1, 1, 1])
        new.bias.data = m.beta + (b - mean) * (m.gamma / std)
        for src, dst in ((aq, new.activation_quantizer), (wq, new.weight_quantizer)):
            dst.scale.copy_(src.scale)
            dst.zero_point.copy_(src.zero_point)
            dst.eps = src.eps
            dst.q_type = src.q_type
            dst.observer.min_val.copy_(src.observer.min_val)
            dst.observer.max_val.copy_(src.observer.max_val)
            dst.observer.num_flag = 1
        return new

    def walk(module):
        for name, child in module.named_children():
            if isinstance(child, quantize.QuantBNFuseConv2d):
                module._modules[name] = fuse(child)
                if getattr(child, "in_shuffle_groups", 0) > 1 and getattr(module, "shuffle_groups", 0) == child.in_shuffle_groups and hasattr(module, "channel_shuffle_flag"):
                    module.channel_shuffle_flag = 1          # prepare(fuse_blocks=True) had folded the block's shuffle into the BN-fused conv: hand it back to the block
            else:
                walk(child)
    walk(model)
    return model
