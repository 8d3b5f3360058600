"""CPU oracle, module/model level (torch CPU ops) -- TEST INFRASTRUCTURE ONLY.

``np_oracle.py`` restates the arithmetic of each primitive; this file restates
how the reference *composes* them into ``nn.Module``s and rewrites a model
(``prepare``), using the same ATen CPU ops in the same order so that on CPU it
is bit-identical to the reference (checked against the fixtures that
``tests/golden/make_golden.py`` generated from the imported reference:
``tests/test_oracle_golden.py``).  It is also the ``cpu_baseline`` ("port")
that ``bench.py`` times on the host cores.

Never imported by ``micronet_amd/`` (the product).  Reference lines followed
(relative to micronet/compression/quantization/):
  dorefa:  wqaq/dorefa/quantize.py 11-73 (quantizers), 107-122/192-199 (modules), 202-323 (prepare)
  wbwtab:  wbwtab/quantize.py 11-149, 181-195, 247-347
  iao:     wqaq/iao/quantize.py 15-113 (observers), 144-321 (quantizers), 492-507, 837-994 (bn-fuse),
           1150-1157 (linear), 1330-1438 (pools), 1484-1498 (add), 1501-1824 (prepare)
One class per role instead of the reference's per-scheme copies; the scheme is data.
"""
import copy

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function


class _RoundSTE(Function):
    @staticmethod
    def forward(ctx, v):
        return torch.sign(v) * torch.floor(torch.abs(v) + 0.5)

    @staticmethod
    def backward(ctx, g):
        return g.clone()


class _RoundClipSTE(Function):
    """iao Round: clip-STE against the observer range expressed in the q-domain."""

    @staticmethod
    def forward(ctx, v, lo, hi, symmetric):
        if symmetric:
            hi = torch.max(torch.abs(lo), torch.abs(hi))
            lo = -hi
        ctx.save_for_backward(v, lo, hi)
        return torch.sign(v) * torch.floor(torch.abs(v) + 0.5)

    @staticmethod
    def backward(ctx, g):
        v, lo, hi = ctx.saved_tensors
        g = g.clone()
        g[v.gt(hi)] = 0
        g[v.lt(lo)] = 0
        return g, None, None, None


class _SignSatSTE(Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        y = torch.sign(x)
        y[y == 0] = 1
        return y

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        g = g.clone()
        g[x.ge(1.0)] = 0
        g[x.le(-1.0)] = 0
        return g


class _SignSTE(Function):
    @staticmethod
    def forward(ctx, x):
        y = torch.sign(x)
        y[y == 0] = 1
        return y

    @staticmethod
    def backward(ctx, g):
        return g.clone()


class _TernarySTE(Function):
    @staticmethod
    def forward(ctx, w):
        thr = torch.mean(torch.abs(w), (3, 2, 1), keepdim=True) * 0.7
        t = torch.sign(torch.add(torch.sign(torch.add(w, thr)), torch.sign(torch.add(w, -thr))))
        return t, thr

    @staticmethod
    def backward(ctx, g, g_thr):
        return g.clone()


# ------------------------------------------------------------------ quantizers
def dorefa_act(x, bits):
    if bits == 32:
        return x
    assert bits != 1
    y = torch.clamp(x * 0.1, 0, 1)
    s = 1 / float(2 ** bits - 1)
    return _RoundSTE.apply(y / s) * s


def dorefa_weight(w, bits):
    if bits == 32:
        return w
    assert bits != 1
    t = torch.tanh(w)
    t = t / 2 / torch.max(torch.abs(t)) + 0.5
    s = 1 / float(2 ** bits - 1)
    q = _RoundSTE.apply(t / s) * s
    return 2 * q - 1


def wbwtab_weight(w, W):
    """w is the Parameter; W==2 mutates w.data in place like the reference."""
    if W == 2:
        w.data.sub_(w.data.mean(1, keepdim=True))
        w.data.clamp_(-1.0, 1.0)
        alpha = torch.mean(torch.abs(w), (3, 2, 1), keepdim=True)
        return _SignSTE.apply(w) * alpha
    if W == 3:
        fp = w.clone()
        t, thr = _TernarySTE.apply(w)
        a = torch.abs(fp)
        le, gt = a.le(thr), a.gt(thr)
        a[le] = 0
        a_th = a.clone()
        alpha = torch.sum(a_th, (3, 2, 1), keepdim=True) / torch.sum(gt, (3, 2, 1), keepdim=True).float()
        return t * alpha
    return w


class Observer(nn.Module):
    def __init__(self, level, channels=None, ema=True, momentum=0.1):
        super().__init__()
        self.level, self.ema, self.momentum, self.first = level, ema, momentum, True
        shape = {"L": (1,), "C": (channels, 1, 1, 1), "FC": (channels, 1)}[level]
        self.register_buffer("min_val", torch.zeros(shape))
        self.register_buffer("max_val", torch.zeros(shape))

    @torch.no_grad()
    def forward(self, x):
        if self.level == "L":
            lo, hi = torch.min(x), torch.max(x)
        elif self.level == "C":
            f = torch.flatten(x, start_dim=1)
            lo, hi = torch.min(f, 1)[0].reshape(self.min_val.shape), torch.max(f, 1)[0].reshape(self.max_val.shape)
        else:
            lo, hi = torch.min(x, 1, keepdim=True)[0], torch.max(x, 1, keepdim=True)[0]
        if self.first:
            self.first = False
        elif self.ema:
            lo = (1 - self.momentum) * self.min_val + self.momentum * lo
            hi = (1 - self.momentum) * self.max_val + self.momentum * hi
        else:
            lo, hi = torch.min(lo, self.min_val), torch.max(hi, self.max_val)
        self.min_val.copy_(lo)
        self.max_val.copy_(hi)


class HistObserver(nn.Module):
    """wqaq/iao/quantize.py:116-139 (PTQ percentile calibrator): max_val = EMA of the k-th smallest |x|, k = int(percentile * n);
    min_val stays 0."""

    def __init__(self, percentile=0.9999, momentum=0.1):
        super().__init__()
        self.level, self.percentile, self.momentum, self.first = "L", percentile, momentum, True
        self.register_buffer("min_val", torch.zeros(1))
        self.register_buffer("max_val", torch.zeros(1))

    @torch.no_grad()
    def forward(self, x):
        cur = torch.kthvalue(x.abs().view(-1), int(self.percentile * x.view(-1).size(0)), dim=0)[0]
        if self.first:
            self.first = False
        else:
            cur = (1 - self.momentum) * self.max_val + self.momentum * cur
        self.max_val.copy_(cur)


class IaoQuantizer(nn.Module):
    def __init__(self, bits, observer, is_act, q_type, union=False, qaft=False):
        super().__init__()
        self.bits, self.observer, self.is_act, self.q_type, self.union, self.qaft = bits, observer, is_act, q_type, union, qaft
        self.register_buffer("scale", torch.ones_like(observer.min_val))
        self.register_buffer("zero_point", torch.zeros_like(observer.min_val))
        self.register_buffer("eps", torch.tensor(torch.finfo(torch.float32).eps))
        if q_type == 0:
            lo = -(1 << (bits - 1)) if is_act else -((1 << (bits - 1)) - 1)
            hi = (1 << (bits - 1)) - 1
        else:
            lo, hi = 0, ((1 << bits) - 1 if is_act else (1 << bits) - 2)
        self.register_buffer("qmin", torch.tensor(float(lo)))
        self.register_buffer("qmax", torch.tensor(float(hi)))

    def update_qparams(self):
        o = self.observer
        if self.q_type == 0:
            qr = float(self.qmax - self.qmin) / 2
            fr = torch.max(torch.abs(o.min_val), torch.abs(o.max_val))
            scale = torch.max(fr / qr, self.eps)
            zp = torch.zeros_like(scale)
        else:
            qr = float(self.qmax - self.qmin)
            scale = torch.max((o.max_val - o.min_val) / qr, self.eps)
            zp = torch.sign(o.min_val) * torch.floor(torch.abs(o.min_val / scale) + 0.5)
        self.scale.copy_(scale)
        self.zero_point.copy_(zp)

    def forward(self, x):
        if self.bits == 32:
            return x
        assert self.bits != 1
        if not self.qaft and self.training:
            if not self.union:
                self.observer(x)
            self.update_qparams()
        o = self.observer
        r = _RoundClipSTE.apply(x / self.scale.clone() - self.zero_point,
                                o.min_val / self.scale - self.zero_point,
                                o.max_val / self.scale - self.zero_point, self.q_type == 0)
        out = (torch.clamp(r, self.qmin, self.qmax) + self.zero_point) * self.scale.clone()
        fd = getattr(self, "force_codes", None)
        if fd is not None:
            # TEST TOOL (teacher forcing of a knife-edge decision, like the masked activation ties of the parity tests): a quantised value whose pre-image sits
            # on a rounding boundary to within the round-off of the float accumulate in front of it may legitimately land on the neighbouring code in another
            # summation order; the parity test then re-evaluates the oracle with THAT value at the tied elements: force_codes = (mask, target).  Gradients flow
            # exactly as before (the forced difference is a constant).
            mask, target = fd
            out = out + ((target.to(out.dtype) - out).detach() * mask.to(out.dtype))
        return out


def _iao_act_quantizer(a_bits, q_type, qaft=False, ptq=False, percentile=0.9999, union=False):
    """ref 361-368 and the `ptq` branches of every Quant* constructor: PTQ is always symmetric with the percentile observer."""
    if ptq:
        return IaoQuantizer(a_bits, HistObserver(percentile), True, 0, union=union, qaft=qaft)
    return IaoQuantizer(a_bits, Observer("L"), True, q_type, union=union, qaft=qaft)


def _iao_pair(a_bits, w_bits, q_type, q_level, weight_observer, out_ch, w_level_c, qaft=False, ptq=False, percentile=0.9999):
    aq = _iao_act_quantizer(a_bits, q_type, qaft, ptq, percentile)
    lvl = w_level_c if q_level == 0 else "L"
    wq = IaoQuantizer(w_bits, Observer(lvl, out_ch if lvl != "L" else None, ema=(weight_observer != 0)), False, 0 if ptq else q_type, qaft=qaft)
    return aq, wq


# --------------------------------------------------------------------- modules
class OConv2d(nn.Conv2d):
    """scheme in {'dorefa','wbwtab','iao'}; cfg is the scheme's keyword dict."""

    def __init__(self, src, scheme, **cfg):
        super().__init__(src.in_channels, src.out_channels, src.kernel_size, src.stride, src.padding, src.dilation,
                         src.groups, src.bias is not None, src.padding_mode)
        self.scheme, self.cfg = scheme, cfg
        self.weight = src.weight
        if src.bias is not None:
            self.bias = src.bias
        if scheme == "iao":
            self.aq, self.wq = _iao_pair(cfg["a_bits"], cfg["w_bits"], cfg.get("q_type", 0), cfg.get("q_level", 0),
                                         cfg.get("weight_observer", 0), src.out_channels, "C", cfg.get("qaft", False),
                                         cfg.get("ptq", False), cfg.get("percentile", 0.9999))

    def _conv(self, x, w, b):
        return F.conv2d(x, w, b, self.stride, self.padding, self.dilation, self.groups)

    quant_inference = False          # True: the stored weights are convolved as they are (every scheme's `if not self.quant_inference` branch: dorefa 107-112,
                                      # wbwtab 181-185, iao 492-497)

    def forward(self, x):
        if self.scheme == "dorefa":
            return self._conv(dorefa_act(x, self.cfg["a_bits"]), self.weight if self.quant_inference else dorefa_weight(self.weight, self.cfg["w_bits"]), self.bias)
        if self.scheme == "wbwtab":
            return self._conv(x, self.weight if self.quant_inference else wbwtab_weight(self.weight, self.cfg["W"]), self.bias)
        return self._conv(self.aq(x), self.weight if self.quant_inference else self.wq(self.weight), self.bias)


class OBNFuseConv2d(OConv2d):
    """wqaq/iao/quantize.py:837-994 incl. the bn_fuse_calib (889-899, 957-972), qaft (918-935, 983-993) and pretrained_model (856-879) branches."""

    def __init__(self, src, bn, **cfg):
        super().__init__(src, "iao", **cfg)
        self.eps, self.momentum, self.first = bn.eps, bn.momentum, True
        self.calib, self.qaft, self.pretrained = cfg.get("bn_fuse_calib", False), cfg.get("qaft", False), cfg.get("pretrained_model", False)
        self.gamma, self.beta = bn.weight, bn.bias
        self.register_buffer("running_mean", bn.running_mean.clone())
        self.register_buffer("running_var", bn.running_var.clone())

    def forward(self, x):
        batch = self.training and not self.qaft
        if batch:
            o = self._conv(x, self.weight, self.bias)
            mean, var = torch.mean(o, dim=[0, 2, 3]), torch.var(o, dim=[0, 2, 3])
            with torch.no_grad():
                if self.first and not self.pretrained:
                    self.first = False
                    rm, rv = mean, var
                else:
                    rm = (1 - self.momentum) * self.running_mean + self.momentum * mean
                    rv = (1 - self.momentum) * self.running_var + self.momentum * var
                self.running_mean.copy_(rm)
                self.running_var.copy_(rv)
        else:
            mean, var = self.running_mean, self.running_var
        if self.bias is not None:
            b_f = (self.beta + (self.bias - mean) * (self.gamma / torch.sqrt(var + self.eps))).reshape(-1)
        else:
            b_f = (self.beta - mean * (self.gamma / torch.sqrt(var + self.eps))).reshape(-1)
        calib = batch and self.calib
        var_w = self.running_var if calib else var          # calib: the weights are folded with the (already updated) running sigma
        w_f = self.weight * (self.gamma / torch.sqrt(var_w + self.eps)).reshape(-1, 1, 1, 1)
        qx, qw = self.aq(x), self.wq(w_f)
        if not calib:
            return self._conv(qx, qw, b_f)
        out = self._conv(qx, qw, None)
        out = out * (torch.sqrt(self.running_var + self.eps) / torch.sqrt(var + self.eps)).reshape(1, -1, 1, 1)
        return out + b_f.reshape(1, -1, 1, 1)


class OConvTranspose2d(nn.ConvTranspose2d):
    """QuantConvTranspose2d of the three schemes (dorefa/quantize.py:126-174, wbwtab/quantize.py:198-244, iao/quantize.py:510-636): the scheme's quantizers in front of
    F.conv_transpose2d; IAO weights are quantised per LAYER only (555-570).  `src`: an nn.ConvTranspose2d whose parameters are shared."""

    def __init__(self, src, scheme, **cfg):
        super().__init__(src.in_channels, src.out_channels, src.kernel_size, src.stride, src.padding, src.output_padding, src.groups, src.bias is not None,
                         src.dilation, src.padding_mode)
        self.scheme, self.cfg = scheme, cfg
        self.weight = src.weight
        if src.bias is not None:
            self.bias = src.bias
        if scheme == "iao":
            self.aq, self.wq = _iao_pair(cfg["a_bits"], cfg["w_bits"], cfg.get("q_type", 0), 1, cfg.get("weight_observer", 0), None, "C", cfg.get("qaft", False),
                                         cfg.get("ptq", False), cfg.get("percentile", 0.9999))

    quant_inference = False

    def forward(self, x):
        if self.scheme == "dorefa":
            qx, qw = dorefa_act(x, self.cfg["a_bits"]), self.weight if self.quant_inference else dorefa_weight(self.weight, self.cfg["w_bits"])
        elif self.scheme == "wbwtab":
            qx, qw = x, self.weight if self.quant_inference else wbwtab_weight(self.weight, self.cfg["W"])
        else:
            qx, qw = self.aq(x), self.weight if self.quant_inference else self.wq(self.weight)
        return F.conv_transpose2d(qx, qw, self.bias, self.stride, self.padding, self.output_padding, self.groups, self.dilation)


class OLinear(nn.Linear):
    def __init__(self, src, scheme, **cfg):
        super().__init__(src.in_features, src.out_features, src.bias is not None)
        self.scheme, self.cfg = scheme, cfg
        self.weight = src.weight
        if src.bias is not None:
            self.bias = src.bias
        if scheme == "iao":
            self.aq, self.wq = _iao_pair(cfg["a_bits"], cfg["w_bits"], cfg.get("q_type", 0), cfg.get("q_level", 0),
                                         cfg.get("weight_observer", 0), src.out_features, "FC")

    def forward(self, x):
        if self.scheme == "dorefa":
            return F.linear(dorefa_act(x, self.cfg["a_bits"]), dorefa_weight(self.weight, self.cfg["w_bits"]), self.bias)
        return F.linear(self.aq(x), self.wq(self.weight), self.bias)


class OBinAct(nn.Module):
    def forward(self, x):
        return _SignSatSTE.apply(x)


class OQuantWrap(nn.Module):
    """iao Quant{ReLU,LeakyReLU,Sigmoid,MaxPool2d,AvgPool2d,AdaptiveAvgPool2d} (ref 1160-1438): quantise the input, then the op."""

    def __init__(self, op, a_bits, q_type, qaft=False, ptq=False, percentile=0.9999):
        super().__init__()
        self.op = op
        self.aq = _iao_act_quantizer(a_bits, q_type, qaft, ptq, percentile)

    def forward(self, x):
        return self.op(self.aq(x))


class OQuantAdd(nn.Module):
    def __init__(self, a_bits, q_type):
        super().__init__()
        self.obs_res, self.obs_short = Observer("L"), Observer("L")
        self.aq = IaoQuantizer(a_bits, Observer("L"), True, q_type, union=True)

    def forward(self, res, shortcut):
        self.obs_res(res)
        self.obs_short(shortcut)
        self.aq.observer.min_val = torch.min(self.obs_res.min_val, self.obs_short.min_val)
        self.aq.observer.max_val = torch.max(self.obs_res.max_val, self.obs_short.max_val)
        return self.aq(res) + self.aq(shortcut)


# --------------------------------------------------------------------- prepare
def _is_add(m):
    return type(m).__name__ == "Add"


def prepare(model, scheme, inplace=False, **cfg):
    """Graph rewrite with the reference's per-scheme skip rules."""
    if not inplace:
        model = copy.deepcopy(model)
    if scheme == "dorefa":
        counter = [0]

        def walk(mod):
            for name, ch in mod.named_children():
                if isinstance(ch, (nn.Conv2d, nn.Linear)):
                    counter[0] += 1
                    if counter[0] > 1:
                        mod._modules[name] = (OConv2d if isinstance(ch, nn.Conv2d) else OLinear)(ch, "dorefa", **cfg)
                else:
                    walk(ch)
        walk(model)
    elif scheme == "wbwtab":
        total = sum(isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)) for m in model.modules())
        counter = [0]

        def walk(mod):
            for name, ch in mod.named_children():
                if isinstance(ch, nn.Conv2d):
                    counter[0] += 1
                    if 1 < counter[0] < total:
                        mod._modules[name] = OConv2d(ch, "wbwtab", W=cfg.get("W", 2))
                elif isinstance(ch, nn.ReLU):
                    if 0 < counter[0] < total:
                        mod._modules[name] = OBinAct() if cfg.get("A", 2) == 2 else nn.ReLU(inplace=True)
                else:
                    walk(ch)
        walk(model)
    elif scheme == "iao":
        bn_fuse = cfg.get("bn_fuse", False)
        a_bits, q_type = cfg["a_bits"], cfg.get("q_type", 0)

        def walk(mod):
            pending = None
            for name, ch in mod.named_children():
                if isinstance(ch, nn.Conv2d):
                    if bn_fuse:
                        pending = (name, ch)
                    else:
                        mod._modules[name] = OConv2d(ch, "iao", **cfg)
                elif isinstance(ch, nn.BatchNorm2d):
                    if bn_fuse:
                        mod._modules[pending[0]] = OBNFuseConv2d(pending[1], ch, **cfg)
                        mod._modules[name] = nn.Identity()
                elif isinstance(ch, nn.Linear):
                    mod._modules[name] = OLinear(ch, "iao", **cfg)
                elif isinstance(ch, nn.MaxPool2d):   # the rewrite keeps only k/stride/padding (1727-1737)
                    mod._modules[name] = OQuantWrap(nn.MaxPool2d(ch.kernel_size, ch.stride, ch.padding), a_bits, q_type)
                elif isinstance(ch, nn.AvgPool2d):
                    mod._modules[name] = OQuantWrap(nn.AvgPool2d(ch.kernel_size, ch.stride, ch.padding), a_bits, q_type)
                elif isinstance(ch, nn.AdaptiveAvgPool2d):
                    mod._modules[name] = OQuantWrap(nn.AdaptiveAvgPool2d(ch.output_size), a_bits, q_type)
                elif _is_add(ch):
                    mod._modules[name] = OQuantAdd(a_bits, q_type)
                else:
                    walk(ch)
        walk(model)
    else:
        raise ValueError(scheme)
    return model


# ---------------------------------------------------------------- inference graphs
def _plain_conv_like(conv):
    return nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, stride=conv.stride, padding=conv.padding, dilation=conv.dilation, groups=conv.groups,
                     bias=True, padding_mode=conv.padding_mode)


@torch.no_grad()
def bn_fuse_wbwtab(model, W, inplace=False):
    """wbwtab/bn_fuse/bn_fuse.py:20-107 on an oracle-prepared net (``prepare(model, "wbwtab", ...)``): every Conv2d -> BatchNorm2d pair (in child order, 86-95) becomes
    one conv with a bias and the BatchNorm an Identity.  The first ``bin_bn_fuse_num`` BatchNorms (= the number of binary activations, ref __main__ 172-177) sit in
    front of a sign: sign(gamma (y - mean) / std + beta) = sign(+-(y - mean + beta std / gamma)), so only the bias -- and the weights' sign where gamma < 0 -- changes
    (36-55); the others get the ordinary w gamma / std fold (56-59).  BatchNorm 2 .. bin_bn_fuse_num yield a quantised conv that convolves its stored weights
    (quant_inference=True, 60-73), the first and the ones past the binary part a plain nn.Conv2d (74-85)."""
    if not inplace:
        model = copy.deepcopy(model)
    bin_bn_fuse_num = sum(isinstance(m, OBinAct) for m in model.modules())
    counter = [0]

    def fuse(conv, bn):
        counter[0] += 1
        k = counter[0]
        mean, std, gamma, beta = bn.running_mean, torch.sqrt(bn.running_var + bn.eps), bn.weight, bn.bias
        w = conv.weight
        b = conv.bias if conv.bias is not None else mean.new_zeros(mean.shape)
        w_f, b_f = w.clone(), b.clone()
        if 1 <= k <= bin_bn_fuse_num:
            pos, neg = gamma.data.gt(0), gamma.data.lt(0)
            w_f[pos] = w[pos]
            b_f[pos] = b[pos] - mean[pos] + beta[pos] * (std[pos] / gamma[pos])
            w_f[neg] = w[neg] * -1
            b_f[neg] = mean[neg] - b[neg] - beta[neg] * (std[neg] / gamma[neg])
        else:
            w_f = w * (gamma / std).reshape([conv.out_channels, 1, 1, 1])
            b_f = beta + (b - mean) * (gamma / std)
        if 2 <= k <= bin_bn_fuse_num:
            new = OConv2d(_plain_conv_like(conv), "wbwtab", W=W)
            new.quant_inference = True
        else:
            new = _plain_conv_like(conv)
        new.weight = nn.Parameter(w_f.detach().clone())
        new.bias = nn.Parameter(b_f.detach().clone())
        return new

    def walk(mod):
        last = None
        for name, ch in mod.named_children():
            if isinstance(ch, nn.Conv2d):
                last = (name, ch)
            elif isinstance(ch, nn.BatchNorm2d):
                mod._modules[last[0]] = fuse(last[1], ch)
                mod._modules[name] = nn.Identity()
            else:
                walk(ch)
    walk(model)
    return model


@torch.no_grad()
def bn_fuse_iao(model, inplace=False):
    """wqaq/iao/bn_fuse/bn_fuse.py:20-80 on an oracle-prepared net: every BN-fused conv (``OBNFuseConv2d``) becomes a quantised conv that convolves its stored
    weights (quant_inference=True) -- w gamma / std and beta + (b - mean) gamma / std from the running statistics (34-35) -- with the trained scale / zero point of
    both quantizers copied over (55-62; the observers' ranges are NOT copied, exactly as in the reference)."""
    if not inplace:
        model = copy.deepcopy(model)

    def fuse(m):
        mean, std = m.running_mean, torch.sqrt(m.running_var + m.eps)
        b = m.bias if m.bias is not None else mean.new_zeros(mean.shape)
        new = OConv2d(_plain_conv_like(m), "iao", **m.cfg)
        new.quant_inference = True
        new.weight = nn.Parameter((m.weight * (m.gamma / std).reshape([m.out_channels, 1, 1, 1])).detach().clone())
        new.bias = nn.Parameter((m.beta + (b - mean) * (m.gamma / std)).detach().clone())
        for src, dst in ((m.aq, new.aq), (m.wq, new.wq)):
            dst.scale.copy_(src.scale)
            dst.zero_point.copy_(src.zero_point)
            dst.eps = src.eps
        return new

    def walk(mod):
        for name, ch in mod.named_children():
            if isinstance(ch, OBNFuseConv2d):
                mod._modules[name] = fuse(ch)
            else:
                walk(ch)
    walk(model)
    return model


# ---------------------------------------------------------------- train step
def make_optimizer(model, lr=0.01, wd=1e-5):
    """dorefa/main.py:308-315 -- Adam, one param group per tensor."""
    groups = [{"params": [p], "lr": lr, "weight_decay": wd} for _, p in model.named_parameters()]
    return torch.optim.Adam(groups, lr=lr, weight_decay=wd)


def train_step(model, opt, x, y):
    """dorefa/main.py:77-82."""
    out = model(x)
    loss = F.cross_entropy(out, y)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss, out
