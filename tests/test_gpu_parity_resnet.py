"""Parity AT THE BENCHED CONFIGURATION for the ResNet workloads (BASELINE configs[3]: resnet18 DoReFa W2A2, 256 images per GPU): the module graph ``prepare()``
really builds -- fused residual blocks on activation codes (qgemm_dense.hip, mn_qa_* / mn_qr_*) -- against the torch-CPU oracle of the reference.

The oracle runs ONE full forward + backward at batch 256 and records, per stage (the stem ``conv1``, each ``BasicBlock``, the classifier tail), input, output,
incoming gradient, input gradient and parameter gradients, plus the two pre-activations of a block (the BatchNorm output in front of the inner ReLU, the sum in
front of the final ReLU).  Each product stage is then TEACHER-FORCED: it gets the oracle's input in the pipeline's physical form (a ``QActTensor``: codes of the
first conv's quantizer + the fp32 activation for an identity shortcut) and the oracle's incoming gradient.  Inside a block the activation between the two convs is
teacher-forced as well (``_mn_mid_hook``): the product's own value is compared with the oracle's and then replaced by it, so that one code flip at a rounding tie
does not hide everything behind it.

Ties: where the ORACLE's own pre-activation lies within TIE_EPS of a ReLU kink (or the quantizer's clamp edge), a 1-ulp difference legitimately flips a mask and a
whole gradient term with it.  The incoming gradient is zeroed at those positions ON BOTH SIDES (the oracle block is re-run with the same masks), so both
back-propagate through identical decisions; the fraction of masked positions is recorded (it is ~1e-5).
Tolerance: 1e-5 relative (max-norm) on every float result; integer codes: flips only at rounding ties (fraction <= 1e-5).  d weight of a DoReFa conv: the single
arg-max |w| element carries a cancelling sum over the whole tensor (see tests/test_gpu_parity_full.py) and is judged against an fp64 evaluation.
Results: ``gpurun_out/parity_r06/`` (merged into ``profiles/parity_r06.json``), once per gradient-term setting (MN_GRAD_TERMS unset = 2, or 3)."""
import copy
import importlib
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BATCH = 256
TIE_EPS = 4e-6
RES = {
    "c4_resnet18_dorefa_w2a2": ("resnet18", "wqaq.dorefa", dict(a_bits=2, w_bits=2)),
}


def _grad_terms():
    """bf16 terms the dense backward kernels carry the fp32 gradient in: 2 (default) or 3 (MN_GRAD_TERMS=3, the exact split) -- micronet_amd/csrc/common.h:mn_grad_terms"""
    e = os.environ.get("MN_GRAD_TERMS") or os.environ.get("MN_QD_TERMS") or ""
    return 3 if e[:1] == "3" else 2


def _record(path, key, value):
    """One file per config AND gradient-term setting under gpurun_out/parity_r06/ (a later subset run on a fresh GPU box can then never overwrite another config's
    record); scripts/merge_parity.py folds them into profiles/parity_r06.json.  The default (two-term) run keeps the plain key, MN_GRAD_TERMS=3 adds "_terms3"."""
    d = os.path.join(os.path.dirname(path), "parity_r06")
    os.makedirs(d, exist_ok=True)
    if _grad_terms() == 3:
        key = key + "_terms3"
    if isinstance(value, dict):
        value = dict(value, _grad_terms=_grad_terms())
    json.dump({key: value}, open(os.path.join(d, "%s.json" % key.replace("/", "_").replace(" ", "_")), "w"), indent=1, sort_keys=True)


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def _codes_of(x, bits):
    s = torch.tensor(1.0 / (2 ** bits - 1), dtype=torch.float32, device=x.device)
    v = torch.clamp(x * 0.1, 0, 1) / s                             # wqaq/dorefa/quantize.py:43-45, every op IEEE-exact
    return (torch.sign(v) * torch.floor(torch.abs(v) + 0.5)).to(torch.uint8).contiguous()


class _FloatToQActF32(torch.autograd.Function):
    """fp32 activation leaf -> what a fused block hands to the next one: (QActTensor of the consuming convs' quantizer codes, the fp32 activation).  Backward:
    the clip-STE of the quantizer applied to the code readers' gradient (a QGrad materialises as exactly that) + the fp32 reader's gradient."""

    @staticmethod
    def forward(ctx, x, bits):
        from micronet_amd.sign_tensor import QActTensor
        xd = x.detach()
        return QActTensor(_codes_of(xd, bits), bits, lambda: xd), xd.clone()

    @staticmethod
    def backward(ctx, gq, ga):
        from micronet_amd import ops
        g = None
        if gq is not None:
            g = ops._chk(gq, "grad")
        if ga is not None:
            g = ga if g is None else g + ga
        return g, None


class _ForceMid(torch.autograd.Function):
    """Teacher-forces the activation between the two convs of a fused block: records the product's own codes / values, hands on the ORACLE's; the gradient coming
    back (a raw QGrad) is zeroed where the oracle's pre-activation is a tie."""

    @staticmethod
    def forward(ctx, h, a_ref, keep, rec):
        from micronet_amd.sign_tensor import QActTensor
        rec["h_codes"], rec["h_f32"], rec["bits"] = h.codes.clone(), h.materialize().clone(), h.bits
        ctx.keep = keep
        return QActTensor(_codes_of(a_ref, h.bits), h.bits, lambda: a_ref)

    @staticmethod
    def backward(ctx, g):
        from micronet_amd.sign_tensor import QGrad
        assert isinstance(g, QGrad) and g._mn_value is None and g._mn_dq2 is None
        return QGrad(g._mn_dq * ctx.keep, g._mn_expand), None, None, None


def _oracle_pass(arch, scheme, kw):
    from oracle import torch_oracle as TO
    from micronet_amd.train import build_model, synth_batch
    torch.set_num_threads(min(32, os.cpu_count()))
    orc = TO.prepare(build_model(arch), scheme.split(".")[-1], inplace=True, **kw).train()
    pristine = copy.deepcopy(orc)
    rec = {}
    blocks = [("conv%d_x.%d" % (i, j), b) for i in range(2, 6) for j, b in enumerate(getattr(orc, "conv%d_x" % i))]
    stages = [("conv1", orc.conv1)] + blocks + [("fc", orc.fc)]

    def hook(name):
        def fn(mod, inputs, output):
            r = rec.setdefault(name, {})
            r["in"] = inputs[0].detach().clone()
            r["out"] = output.detach().clone()
            if inputs[0].requires_grad:
                inputs[0].register_hook(lambda g, r=r: r.__setitem__("gin", g.detach().clone()))
            output.register_hook(lambda g, r=r: r.__setitem__("gout", g.detach().clone()))
        return fn
    for n, m in stages:
        m.register_forward_hook(hook(n))
    orc.conv1[2].register_forward_pre_hook(lambda mod, i: rec.setdefault("conv1", {}).__setitem__("z", i[0].detach().clone()))
    for n, b in blocks:
        b.residual_function[2].register_forward_pre_hook(lambda mod, i, n=n: rec.setdefault(n, {}).__setitem__("z_mid", i[0].detach().clone()))
        b.residual_function[2].register_forward_hook(lambda mod, i, o, n=n: rec.setdefault(n, {}).__setitem__("a_mid", o.detach().clone()))
        b.add.register_forward_hook(lambda mod, i, o, n=n: rec.setdefault(n, {}).__setitem__("u", o.detach().clone()))
    x, y = synth_batch(BATCH)
    out = orc(x)
    loss = torch.nn.functional.cross_entropy(out, y)
    loss.backward()
    return orc, pristine, rec, x, float(loss), [n for n, _ in stages]


def _get(model, name):
    m = model
    for part in name.split("."):
        m = m[int(part)] if part.isdigit() else getattr(m, part)
    return m


def _oracle_rerun(stage, x_in, gout, keep_out, keep_mid, double=False):
    """The oracle stage again (fp32, or fp64 for the cancelling sums) with the tie masks applied to the incoming gradient and to the gradient of the mid activation."""
    st = copy.deepcopy(stage).train()
    if double:
        st = st.double()
    if keep_mid is not None:
        km = keep_mid.double() if double else keep_mid.float()
        def mid_hook(mod, i, o):
            o.register_hook(lambda g: g * km)
        st.residual_function[2].register_forward_hook(mid_hook)
    xi = (x_in.double() if double else x_in.clone()).requires_grad_(True)
    out = st(xi)
    g = gout * keep_out if keep_out is not None else gout
    out.backward(g.double() if double else g)
    return xi.grad, {pn: p.grad.detach().clone() for pn, p in st.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("key", list(RES))
def test_full_batch_teacher_forced_resnet(key):
    from micronet_amd import ops
    from micronet_amd.sign_tensor import QActTensor
    from micronet_amd.train import build_model
    from oracle import np_oracle as NO
    arch, scheme, kw = RES[key]
    quantize = importlib.import_module("micronet.compression.quantization.%s.quantize" % scheme)
    orc, pristine, rec, x, loss0, names = _oracle_pass(arch, scheme, kw)
    prod = quantize.prepare(build_model(arch), inplace=True, **kw).cuda().train()
    bits = kw["a_bits"]
    report, failures, worst = {}, [], 0.0

    def check(tag, errs, k_, v, lim=1e-5):
        nonlocal worst
        errs[k_] = v
        worst = max(worst, v)
        if not v <= lim:
            failures.append((tag, k_, v, lim))

    def check_pgrads(tag, errs, pmod, pg_ref, stage_pristine, x_in, gout, keep_out, keep_mid):
        pn = dict(pmod.named_parameters())
        p64 = None
        for name, g_ref in pg_ref.items():
            if name not in pn or pn[name].grad is None:
                failures.append((tag, "missing gradient", name))
                continue
            g = pn[name].grad
            if name.endswith("weight") and g.dim() == 4 and g.shape[1] > 3:
                # DoReFa weight quantizer: the arg-max |w| element carries the cancelling sum (tests/test_gpu_parity_full.py): fp64 judge for that element
                k_arg = int(pn[name].detach().abs().flatten().argmax())
                d = (g.detach().double().cpu().flatten() - g_ref.double().flatten()).abs() / g_ref.double().abs().max().clamp_min(1e-30)
                e_arg = float(d[k_arg])
                d[k_arg] = 0.0
                check(tag, errs, "d" + name, float(d.max()))
                errs["d" + name + "_argmax_element_vs_reference_fp32"] = e_arg
                if e_arg > 1e-5:
                    if p64 is None:
                        _, p64 = _oracle_rerun(stage_pristine, x_in, gout, keep_out, keep_mid, double=True)
                    g64 = p64[name]
                    sc = g64.abs().max().clamp_min(1e-300)
                    e_ours = float((g.double().cpu().flatten()[k_arg] - g64.flatten()[k_arg]).abs() / sc)
                    e_ref = float((g_ref.double().flatten()[k_arg] - g64.flatten()[k_arg]).abs() / sc)
                    errs["d" + name + "_argmax_element"], errs["d" + name + "_argmax_element_reference_fp32_vs_fp64"] = e_ours, e_ref
                    if e_ours > max(2e-4, 2.0 * e_ref):
                        failures.append((tag, "d" + name + " at the arg-max element vs fp64", e_ours, e_ref))
            else:
                e = _rel(g, g_ref)
                if e > 1e-5:          # a cancelling per-channel sum (d gamma / d beta): ours must be as close to fp64 as the reference's own fp32 result
                    if p64 is None:
                        _, p64 = _oracle_rerun(stage_pristine, x_in, gout, keep_out, keep_mid, double=True)
                    g64 = p64[name]
                    sc = g64.abs().max().clamp_min(1e-300)
                    e_ours, e_ref = float((g.double().cpu() - g64).abs().max() / sc), float((g_ref.double() - g64).abs().max() / sc)
                    errs["d" + name + "_reference_fp32_vs_fp64"] = e_ref
                    check(tag, errs, "d" + name, e_ours, max(1e-5, 2.0 * e_ref))
                else:
                    check(tag, errs, "d" + name, e)

    for name in names:
        r = rec[name]
        pst, ost = _get(prod, name), _get(pristine, name)
        for p in pst.parameters():
            p.grad = None
        errs = {}
        if name == "conv1":
            # ---- the stem: image -> conv (fp32) -> BatchNorm + ReLU -> codes of the first block's quantizer + fp32 (one pass, two autograd outputs)
            out = pst(r["in"].cuda())
            assert isinstance(out, QActTensor) and out._mn_f32 is not None, "the stem does not emit codes + fp32"
            check(name, errs, "y", _rel(out._mn_f32, r["out"]))
            _, cref = NO.dorefa_act_fwd(r["out"].numpy(), out.bits)
            errs["codes_flipped_frac"] = float((out.codes.cpu().numpy().astype("float32") != cref).sum()) / cref.size
            if errs["codes_flipped_frac"] > 1e-5:
                failures.append((name, "codes differ beyond rounding ties", errs["codes_flipped_frac"]))
            keep = ~(r["z"].abs() <= TIE_EPS)
            errs["ties_masked_frac"] = float((~keep).sum()) / keep.numel()
            st = copy.deepcopy(ost).train()
            st(r["in"]).backward(r["gout"] * keep)
            pg_ref = {pn: p.grad.detach().clone() for pn, p in st.named_parameters() if p.grad is not None}
            torch.autograd.backward([out._mn_f32], [(r["gout"] * keep).cuda()])
            for pn_, g_ref in pg_ref.items():
                check(name, errs, "d" + pn_, _rel(dict(pst.named_parameters())[pn_].grad, g_ref))
        elif name == "fc":
            # ---- classifier tail: QuantLinear on the pooled fp32 activation (quantizer in the kernel prologue)
            leaf = r["in"].cuda().requires_grad_(True)
            out = pst(leaf)
            check(name, errs, "y", _rel(out, r["out"]))
            out.backward(r["gout"].cuda())
            st = copy.deepcopy(ost).train()
            xi = r["in"].clone().requires_grad_(True)
            st(xi).backward(r["gout"])
            check(name, errs, "dx", _rel(leaf.grad, xi.grad))
            st64 = copy.deepcopy(ost).double().train()
            st64(r["in"].double()).backward(r["gout"].double())
            g64s = {pn_: p.grad for pn_, p in st64.named_parameters()}
            for pn_, p in st.named_parameters():
                g = dict(pst.named_parameters())[pn_].grad
                if pn_ == "weight":          # against the fp64 evaluation, next to the reference's own fp32 result against it
                    sc = g64s[pn_].abs().max().clamp_min(1e-300)
                    e_ours, e_ref = float((g.double().cpu() - g64s[pn_]).abs().max() / sc), float((p.grad.double() - g64s[pn_]).abs().max() / sc)
                    errs["d" + pn_ + "_vs_reference_fp32"] = _rel(g, p.grad)
                    errs["d" + pn_ + "_reference_fp32_vs_fp64"] = e_ref
                    check(name, errs, "d" + pn_ + "_vs_fp64", e_ours, max(1e-5, 2.0 * e_ref))
                else:
                    check(name, errs, "d" + pn_, _rel(g, p.grad))
        else:
            # ---- a residual block
            assert type(pst).__name__.startswith("Fused"), "prepare() did not fuse %s" % name
            leaf = r["in"].cuda().requires_grad_(True)
            q, a = _FloatToQActF32.apply(leaf, bits)
            q._mn_f32 = a if len(pst.shortcut) == 0 else None
            z_mid, u = r["z_mid"], r["u"]
            keep_mid = ~((z_mid.abs() <= TIE_EPS) | ((0.1 * z_mid - 1.0).abs() <= TIE_EPS))
            keep_out = ~(u.abs() <= TIE_EPS)
            mid = {}
            pst._mn_mid_hook = lambda h, r=r, keep_mid=keep_mid, mid=mid: _ForceMid.apply(h, r["a_mid"].cuda(), keep_mid.cuda(), mid)
            try:
                out = pst(q)
            finally:
                pst._mn_mid_hook = None
            # the product's own mid activation (before it was replaced) and the block output
            check(name, errs, "a_mid", _rel(mid["h_f32"], r["a_mid"]))
            _, cref = NO.dorefa_act_fwd(r["a_mid"].numpy(), mid["bits"])
            errs["mid_codes_flipped_frac"] = float((mid["h_codes"].cpu().numpy().astype("float32") != cref).sum()) / cref.size
            if errs["mid_codes_flipped_frac"] > 1e-5:
                failures.append((name, "mid codes differ beyond rounding ties", errs["mid_codes_flipped_frac"]))
            out_f32 = out._mn_f32 if isinstance(out, QActTensor) and out._mn_f32 is not None else (out.materialize() if isinstance(out, QActTensor) else out)
            check(name, errs, "y", _rel(out_f32, r["out"]))
            if isinstance(out, QActTensor):
                _, cref = NO.dorefa_act_fwd(r["out"].numpy(), out.bits)
                errs["codes_flipped_frac"] = float((out.codes.cpu().numpy().astype("float32") != cref).sum()) / cref.size
                if errs["codes_flipped_frac"] > 1e-5:
                    failures.append((name, "codes differ beyond rounding ties", errs["codes_flipped_frac"]))
            errs["ties_masked_frac"] = (float((~keep_mid).sum()) + float((~keep_out).sum())) / (keep_mid.numel() + keep_out.numel())
            gin_ref, pg_ref = _oracle_rerun(ost, r["in"], r["gout"], keep_out, keep_mid)
            gout = (r["gout"] * keep_out).cuda()
            if isinstance(out, QActTensor):
                # the oracle's incoming gradient is w.r.t. the block's fp32 output: feed it to the fp32 output when there is one, else through the code
                # tensor as a plain (already STE-treated) gradient
                torch.autograd.backward([out._mn_f32 if out._mn_f32 is not None else out], [gout])
            else:
                torch.autograd.backward([out], [gout])
            check(name, errs, "dx", _rel(leaf.grad, gin_ref))
            check_pgrads(name, errs, pst, pg_ref, ost, r["in"], r["gout"], keep_out, keep_mid)
        report[name] = {k: float("%.2e" % v) for k, v in errs.items()}
    report["_oracle_loss0"], report["_batch"], report["_failures"] = loss0, BATCH, [list(map(str, f)) for f in failures]
    _record(os.path.join(ROOT, "gpurun_out", "parity_r06.json"), key, report)
    print(key, "worst rel err over all stages:", worst)
    assert not failures, (key, failures, report)


# ------------------------------------------------------------------------------------------------ BASELINE configs[4]: resnet18 IAO W4A4 (+ QuantAdd)
IAO_RES = {
    "c5_resnet18_iao_w4a4": ("resnet18", "wqaq.iao", dict(a_bits=4, w_bits=4, q_type=0, q_level=0)),
}


def _flip_aware(out, ref, step_hint=None):
    """A quantised activation (QuantReLU / QuantAdd output): (max-norm relative error over the elements that agree, fraction of elements that sit on ANOTHER
    quantisation level).  A 4-bit level is 1/15 of the range, so one element whose pre-rounding value is a tie in the oracle flips by a whole step; such
    flips are counted (bounded at 2e-5 of the elements), everything else must agree to 1e-5."""
    o, r = out.detach().double().cpu(), ref.detach().double().cpu()
    sc = r.abs().max().clamp_min(1e-30)
    d = (o - r).abs() / sc
    flipped = d > 1e-3          # (a level step is >= 1/255 of the range at 8 bits, 1/15 at 4)
    return float(d[~flipped].max()) if bool((~flipped).any()) else 0.0, float(flipped.double().mean())


@pytest.mark.parametrize("key", list(IAO_RES))
def test_full_batch_teacher_forced_resnet_iao(key):
    """c5 at the benched batch: every stage of the IAO ResNet -- stem, the four pieces of each BasicBlock (conv-bn-relu, conv-bn, the 1x1 shortcut,
    QuantAdd + relu: models/resnet.py:7-65 under wqaq/iao/quantize.py) and the classifier tail -- teacher-forced with the oracle's input and incoming
    gradient (first training step: every observer takes its first range from this batch, on both sides).  The dense convs run on qgemm_dense.hip
    (k_qd_fwd8 on signed codes, k_qd_dgrad + STE, k_qd_wgrad)."""
    from micronet_amd.train import build_model, synth_batch
    from oracle import torch_oracle as TO
    arch, scheme, kw = IAO_RES[key]
    quantize = importlib.import_module("micronet.compression.quantization.%s.quantize" % scheme)
    torch.set_num_threads(min(32, os.cpu_count()))
    orc = TO.prepare(build_model(arch), "iao", inplace=True, **kw).train()
    pristine = copy.deepcopy(orc)
    prod = quantize.prepare(build_model(arch), inplace=True, **kw).cuda().train()
    rec = {}
    blocks = [("conv%d_x.%d" % (i, j)) for i in range(2, 6) for j in range(len(getattr(orc, "conv%d_x" % i)))]

    def hook(name):
        def fn(mod, inputs, output):
            r = rec.setdefault(name, {})
            r["in"], r["out"] = inputs[0].detach().clone(), output.detach().clone()
            output.register_hook(lambda g, r=r: r.__setitem__("gout", g.detach().clone()))
        return fn
    for n in ["conv1"] + blocks + ["avg_pool", "fc"]:
        _get(orc, n).register_forward_hook(hook(n))
    x, y = synth_batch(BATCH)
    out = orc(x)
    loss0 = float(torch.nn.functional.cross_entropy(out, y).detach())
    torch.nn.functional.cross_entropy(out, y).backward()

    report, failures, worst = {}, [], 0.0

    def check(tag, errs, k_, v, lim=1e-5):
        nonlocal worst
        errs[k_] = v
        worst = max(worst, v)
        if not v <= lim:
            failures.append((tag, k_, v, lim))

    def run_piece(tag, errs, omods, pmods, ins, gout, quantised_out, combine=None):
        """One piece: oracle modules (CPU, pristine copies) vs product modules (GPU) on the SAME inputs / incoming gradient.  ins: list of CPU tensors (None grads for
        the image).  combine(mods, *inputs) -> output (default: the modules in sequence on the single input).  Returns the oracle's output and input gradients."""
        oms = [copy.deepcopy(m).train() for m in omods]
        seq = combine or (lambda mods, t: _seq(mods, t))
        o_in = [t.clone().requires_grad_(t.is_floating_point() and tag != "conv1") for t in ins]
        o_out = seq(oms, *o_in)
        o_out.backward(gout)
        p_in = [t.cuda().requires_grad_(t_.requires_grad) for t, t_ in zip(ins, o_in)]
        for m in pmods:
            for p in m.parameters():
                p.grad = None
        p_out = seq(pmods, *p_in)
        if quantised_out:
            e, ff = _flip_aware(p_out, o_out)
            check(tag, errs, "y", e)
            errs["y_level_flips_frac"] = ff
            if ff > 2e-5:
                failures.append((tag, "quantised output on other levels beyond rounding ties", ff))
        else:
            check(tag, errs, "y", _rel(p_out, o_out))
        p_out.backward(gout.cuda())
        for k_, (a, b) in enumerate(zip(p_in, o_in)):
            if b.grad is not None:
                e = _rel(a.grad, b.grad)
                check(tag, errs, "dx%d" % k_, e)
        need64 = []
        for j, (om, pm) in enumerate(zip(oms, pmods)):
            pn = dict(pm.named_parameters())
            for name, p in om.named_parameters():
                if p.grad is None:
                    continue
                e = _rel(pn[name].grad, p.grad)
                if e > 1e-5:
                    need64.append((j, name, pn[name].grad, p.grad))
                else:
                    check(tag, errs, "d%d.%s" % (j, name), e)
        if need64:          # cancelling sums: both sides against the fp64 evaluation of the same oracle piece
            dms = [copy.deepcopy(m).double().train() for m in omods]
            d_in = [t.double().clone().requires_grad_(b.requires_grad) for t, b in zip(ins, o_in)]
            seq(dms, *d_in).backward(gout.double())
            for j, name, g, g_ref in need64:
                g64 = dict(dms[j].named_parameters())[name].grad
                sc = g64.abs().max().clamp_min(1e-300)
                e_ours, e_ref = float((g.double().cpu() - g64).abs().max() / sc), float((g_ref.double() - g64).abs().max() / sc)
                errs["d%d.%s_reference_fp32_vs_fp64" % (j, name)] = e_ref
                check(tag, errs, "d%d.%s" % (j, name), e_ours, max(1e-5, 2.0 * e_ref))
        return o_out.detach(), [t.grad for t in o_in]

    def _seq(mods, t):
        for m in mods:
            t = m(t)
        return t

    # ---- stem
    # (the ReLU of the reference's IAO graph stays a plain nn.ReLU -- wqaq/iao/quantize.py:1706-1709 is commented out -- so the only knife-edge decision inside a
    #  conv-bn-relu piece is the kink: where the ORACLE's pre-activation is within TIE_EPS of 0 the incoming gradient is zeroed on both sides, as for c4)
    errs = {}
    with torch.no_grad():
        st0 = copy.deepcopy(pristine.conv1).train()
        z0 = st0[1](st0[0](rec["conv1"]["in"]))
    keep0 = ~(z0.abs() <= TIE_EPS)
    errs["ties_masked_frac"] = float((~keep0).sum()) / keep0.numel()
    run_piece("conv1", errs, [pristine.conv1], [prod.conv1], [rec["conv1"]["in"]], rec["conv1"]["gout"] * keep0, False)
    report["conv1"] = {k: float("%.2e" % v) for k, v in errs.items()}
    # ---- blocks, in four pieces each
    for n in blocks:
        ob, pb = _get(pristine, n), _get(prod, n)
        pb_whole = copy.deepcopy(pb)          # (the whole-block stage at the end needs the block's quantizers at their FIRST call too: the pieces below consume pb's)
        xin, gout = rec[n]["in"], rec[n]["gout"]
        has_sc = len(ob.shortcut) > 0
        # oracle intermediates of THIS block on the oracle's own input (fresh copy: first observer call, as in the full pass)
        oc = copy.deepcopy(ob).train()
        xa = xin.clone().requires_grad_(True)
        z_mid = oc.residual_function[1](oc.residual_function[0](xa))
        keep_a = ~(z_mid.detach().abs() <= TIE_EPS)
        a_mid = oc.residual_function[2](z_mid)
        a_mid.retain_grad()
        zb = oc.residual_function[4](oc.residual_function[3](a_mid))
        zb.retain_grad()
        xs = xin.clone().requires_grad_(True)
        zs = oc.shortcut(xs)
        if has_sc:
            zs.retain_grad()
        u_sum = oc.add(zb, zs)
        # u is a sum of two values of ONE quantizer grid (k1 + k2) * s, bit-identical on both sides: an exact zero is not a tie (relu'(0) = 0 for the oracle and for
        # the kernel alike); only a NON-zero |u| within TIE_EPS could flip -- VERDICT r3 weak 1(a)
        keep_t = ~((u_sum.detach().abs() <= TIE_EPS) & (u_sum.detach() != 0))
        o_out = oc.relu(u_sum) if hasattr(oc, "relu") else torch.nn.functional.relu(u_sum)
        o_out.backward(gout)
        g_mid, g_zb, g_zs = a_mid.grad.clone(), zb.grad.clone(), (zs.grad.clone() if has_sc else xs.grad.clone())
        rf_o, rf_p = ob.residual_function, pb.residual_function
        errs = {"ties_masked_frac": float((~keep_a).sum()) / keep_a.numel()}
        run_piece(n + ":a", errs, [rf_o[0], rf_o[1], rf_o[2]], [rf_p[0], rf_p[1], rf_p[2]], [xin], g_mid * keep_a, False)
        report[n + ":conv-bn-relu"] = {k: float("%.2e" % v) for k, v in errs.items()}
        errs = {}
        run_piece(n + ":b", errs, [rf_o[3], rf_o[4]], [rf_p[3], rf_p[4]], [a_mid.detach()], g_zb, False)
        report[n + ":conv-bn"] = {k: float("%.2e" % v) for k, v in errs.items()}
        if has_sc:
            errs = {}
            run_piece(n + ":s", errs, [ob.shortcut[0], ob.shortcut[1]], [pb.shortcut[0], pb.shortcut[1]], [xin], g_zs, False)
            report[n + ":shortcut"] = {k: float("%.2e" % v) for k, v in errs.items()}
        errs = {}
        tail_o = [ob.add] + ([ob.relu] if hasattr(ob, "relu") else [])
        tail_p = [pb.add] + ([pb.relu] if hasattr(pb, "relu") else [])
        def comb(mods, u, v):          # the product's QuantAdd takes the block's ReLU into its own pass (relu=True); the oracle's is the reference's
            import inspect
            if len(mods) > 1:
                return mods[1](mods[0](u, v))
            if "relu" in inspect.signature(mods[0].forward).parameters:
                return mods[0](u, v, relu=True)
            return torch.nn.functional.relu(mods[0](u, v))
        errs["ties_masked_frac"] = float((~keep_t).sum()) / keep_t.numel()
        run_piece(n + ":t", errs, tail_o, tail_p, [zb.detach(), zs.detach()], gout * keep_t, True, combine=comb)
        report[n + ":qadd-relu"] = {k: float("%.2e" % v) for k, v in errs.items()}
        # ---- the WHOLE block with every hand-over live: conv -> fused BN+ReLU (its (min, max) partials feed the next conv's observer) -> conv ->
        # BN -> [shortcut conv -> BN] -> QuantAdd + ReLU (its partials feed the next block), on the oracle's input and incoming gradient; the product's intermediate
        # tensors are its own (1 ulp from the oracle's), nothing is re-fed.  Two decisions INSIDE the block sit on a knife edge BY CONSTRUCTION and are neutralised the
        # same way as everywhere else in this file -- the gradient AT those elements is zeroed on both sides (tensor hooks on the three intermediate activations; the
        # hooks change no value and no hand-over):
        #   * the ReLU kink of the mid activation (keep_a, ~1e-5 of the elements);
        #   * the LARGEST element of every tensor a symmetric IAO quantizer observes: scale = max / 7.5, so that element's code is round(7.5 -+ 1 ulp) = 7 (gradient
        #     passes) or 8 -> clamped to 7 (torch.clamp's gradient is 0), decided by the last bit of max / (max / 7.5) (measured, scripts/dbg_c5_whole.py: in conv2_x.1
        #     the two sides' mid maxima differ in the last bit, the oracle clamps and the product does not; ONE element of g_mid then moves BatchNorm-1's d gamma by
        #     1.6 % because the per-channel sums cancel to ~1e-2 of their terms).  Every element within 4e-6 of the tensor's positive maximum is masked (1-3 elements).
        # With those two classes out, output, dx and every parameter gradient are held to the ordinary 1e-5 (2e-5 for the cancelling parameter sums).
        #   * (third class, forward) a ROUNDING tie of one of the block's 4-bit quantizers -- an element whose value / scale sits within 1 ulp of k + 1/2 lands on the
        #     neighbouring level on one side; at 4 bits that moves 9 x Cout outputs of the next conv by a whole weight step and the two forward passes are no longer
        #     the same function (measured: conv5_x.1, 38 of 2 M outputs on another level, every gradient then 0.3-3 % apart).  Such a realisation cannot be compared
        #     and cannot be forced from outside; the stage then re-runs BOTH sides on the same input scaled by (1 + 2^-9) -- another realisation of the rounding
        #     decisions, every observer again at its first call -- and requires one attempt of three without a forward flip (flipped attempts are recorded).
        def _top(t):
            t = t.detach()
            return (t >= t.abs().max() * (1.0 - TIE_EPS)) & (t > 0)

        def _mask_grad(mod, keep, cuda):
            keep = keep.cuda() if cuda else keep
            def fn(m_, i_, o_):
                if o_.requires_grad:
                    o_.register_hook(lambda g_: g_ * keep)
            return mod.register_forward_hook(fn)

        def whole(x_in, attempt):
            errs_, fails_ = {}, []
            def chk(name, err, tol=1e-5):
                errs_[name] = err
                if not err <= tol:
                    fails_.append((n + ":whole", name, err, tol))
            # the oracle's own intermediates on THIS input: where the ties are
            oc_ = copy.deepcopy(ob).train()
            with torch.no_grad():
                z_ = oc_.residual_function[1](oc_.residual_function[0](x_in))
                kink = z_.abs() <= TIE_EPS          # (before the ReLU: the reference's is in place)
                a_ = oc_.residual_function[2](z_)
                zb_ = oc_.residual_function[4](oc_.residual_function[3](a_))
                zs_ = oc_.shortcut(x_in)
                u_ = oc_.add(zb_, zs_)
            k_mid = ~kink & ~_top(a_)
            k_zb, k_zs = ~_top(zb_), (~_top(zs_) if has_sc else None)
            k_t = ~((u_.abs() <= TIE_EPS) & (u_ != 0))
            errs_.update(mid_ties=float((~k_mid).sum()), zb_top_ties=float((~k_zb).sum()))
            ob2, pb2 = copy.deepcopy(ob).train(), copy.deepcopy(pb_whole).train()
            handles = []
            for blk_, cuda in ((ob2, False), (pb2, True)):
                handles.append(_mask_grad(blk_.residual_function[2], k_mid, cuda))
                handles.append(_mask_grad(blk_.residual_function[4], k_zb, cuda))
                if has_sc:
                    handles.append(_mask_grad(blk_.shortcut[1], k_zs, cuda))
            xo = x_in.clone().requires_grad_(True)
            yo = ob2(xo)
            yo.backward(gout * k_t)
            for p_ in pb2.parameters():
                p_.grad = None
            xp = x_in.cuda().requires_grad_(True)
            yp = pb2(xp)
            e, ff = _flip_aware(yp, yo)
            chk("y", e)
            errs_["y_level_flips_frac"] = ff
            yp.backward((gout * k_t).cuda())
            for h_ in handles:
                h_.remove()
            if ff > 0:
                return errs_, fails_, True
            # (dx: the input quantizer's own top element and -- identity shortcut -- the QuantAdd's second operand are the block INPUT's knife edges: one element each)
            d = (xp.grad.detach().double().cpu() - xo.grad.double()).abs() / xo.grad.double().abs().max().clamp_min(1e-30)
            errs_["dx_max"] = float(d.max())
            errs_["dx_outliers"] = float((d > 1e-5).sum())
            kth = max(1, int(0.9999 * d.numel()))
            chk("dx_p9999", float(d.flatten().kthvalue(kth).values))
            if errs_["dx_outliers"] > 16:
                fails_.append((n + ":whole", "dx outliers beyond the reach of the input's top-element tie", errs_["dx_outliers"]))
            # parameter gradients: against the fp64 evaluation of the same oracle block (same masks), with the reference's own fp32 result against it alongside --
            # ours must be within 1e-5 of fp64, or as close to it as twice the reference's own fp32 rounding
            ob64 = copy.deepcopy(ob).double().train()
            h64 = [_mask_grad(ob64.residual_function[2], k_mid, False), _mask_grad(ob64.residual_function[4], k_zb, False)]
            if has_sc:
                h64.append(_mask_grad(ob64.shortcut[1], k_zs, False))
            x64 = x_in.double().clone().requires_grad_(True)
            ob64(x64).backward((gout * k_t).double())
            for h_ in h64:
                h_.remove()
            g64s = {name: p_.grad for name, p_ in ob64.named_parameters() if p_.grad is not None}
            pn2 = dict(pb2.named_parameters())
            for name, p_ in ob2.named_parameters():
                if p_.grad is not None and name in pn2 and pn2[name].grad is not None:
                    errs_["d" + name + "_vs_reference_fp32"] = _rel(pn2[name].grad, p_.grad)
                    if name in g64s:
                        sc = g64s[name].abs().max().clamp_min(1e-300)
                        e_ours = float((pn2[name].grad.double().cpu() - g64s[name]).abs().max() / sc)
                        e_ref = float((p_.grad.double() - g64s[name]).abs().max() / sc)
                        errs_["d" + name + "_reference_fp32_vs_fp64"] = e_ref
                        chk("d" + name, e_ours, max(1e-5, 2.0 * e_ref))
                    else:
                        chk("d" + name, errs_["d" + name + "_vs_reference_fp32"])
            return errs_, fails_, False

        errs, flipped_attempts = {}, []
        for attempt in range(3):
            errs, fails_, flipped = whole(xin * (1.0 + attempt * 2.0 ** -9), attempt)
            if not flipped:
                break
            flipped_attempts.append(errs["y_level_flips_frac"])
        errs["attempts_with_forward_level_flips"] = float(len(flipped_attempts))
        if flipped:
            failures.append((n + ":whole", "three realisations in a row with forward level flips", flipped_attempts))
        else:
            for f_ in fails_:
                failures.append(f_)
                worst = max(worst, f_[2]) if isinstance(f_[2], float) else worst
            for k_, v_ in errs.items():
                if k_ in ("y", "dx_p9999") or k_.startswith("dresidual") or k_.startswith("dshortcut"):
                    worst = max(worst, v_)
        report[n + ":whole-block"] = {k: float("%.2e" % v) for k, v in errs.items()}
    # ---- classifier tail: average pool -> flatten -> QuantLinear
    errs = {}
    comb = lambda mods, t: mods[1](mods[0](t).view(t.size(0), -1))
    run_piece("tail", errs, [pristine.avg_pool, pristine.fc], [prod.avg_pool, prod.fc], [rec["avg_pool"]["in"]], rec["fc"]["gout"], False, combine=comb)
    report["tail"] = {k: float("%.2e" % v) for k, v in errs.items()}
    report["_oracle_loss0"], report["_batch"], report["_failures"] = loss0, BATCH, [list(map(str, f)) for f in failures]
    _record(os.path.join(ROOT, "gpurun_out", "parity_r06.json"), key, report)
    print(key, "worst rel err over all stages:", worst)
    assert not failures, (key, failures, report)
