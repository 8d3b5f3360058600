"""The Python boundary (module surface) without running any kernel: graph rewrite, class names, state_dict keys and
shapes must equal the reference's (recorded in tests/golden/meta.json by make_golden.py)."""
import importlib

import pytest
import torch
import torch.nn as nn

from micronet_amd.train import build_model

CFG = {
    "c1_nin_gc_dorefa_w8a8": ("nin_gc", "wqaq.dorefa", dict(a_bits=8, w_bits=8)),
    "c2_nin_gc_wbwtab_w3a2": ("nin_gc", "wbwtab", dict(A=2, W=3)),
    "c2b_nin_gc_wbwtab_w2a2": ("nin_gc", "wbwtab", dict(A=2, W=2)),
    "c3_nin_gc_iao_w8a8_bnfuse": ("nin_gc", "wqaq.iao", dict(a_bits=8, w_bits=8, q_type=0, q_level=0, weight_observer=0, bn_fuse=True)),
    "c4_resnet18_dorefa_w2a2": ("resnet18", "wqaq.dorefa", dict(a_bits=2, w_bits=2)),
    "c5_resnet18_iao_w4a4": ("resnet18", "wqaq.iao", dict(a_bits=4, w_bits=4, q_type=0, q_level=0)),
    "nin_dorefa_w4a4": ("nin", "wqaq.dorefa", dict(a_bits=4, w_bits=4)),
}


@pytest.mark.parametrize("key", list(CFG))
def test_prepare_matches_reference_surface(golden, key):
    arch, scheme, kw = CFG[key]
    quantize = importlib.import_module("micronet.compression.quantization.%s.quantize" % scheme)
    model = build_model(arch)
    before = {k: v.clone() for k, v in model.state_dict().items()}
    q = quantize.prepare(model, inplace=True, **kw)
    surf = golden.meta["surface"][key]
    # same names in the same order; each module is the reference's class or a subclass of it (isinstance contract:
    # wbwtab's fused BatchNorm2dBinAct is an nn.BatchNorm2d with identical parameters, buffers and state_dict keys)
    mods = list(q.named_modules())
    assert [n for n, _ in mods] == [n for n, _ in surf["modules"]]
    for (n, m), (_, ref_cls) in zip(mods, surf["modules"]):
        assert ref_cls in [c.__name__ for c in type(m).__mro__], (n, type(m).__name__, ref_cls)
    assert [[k, list(v.shape)] for k, v in q.state_dict().items()] == surf["state"]
    # parameters keep their values and the quantised modules share the original storage
    for k, v in q.state_dict().items():
        if k in before:
            assert torch.equal(v, before[k])


def test_not_inplace_deepcopies():
    from micronet.compression.quantization.wqaq.dorefa import quantize
    m = build_model("nin")
    q = quantize.prepare(m, inplace=False, a_bits=4, w_bits=4)
    assert isinstance(m.model[1].conv, nn.Conv2d) and not isinstance(m.model[1].conv, quantize.QuantConv2d)
    assert isinstance(q.model[1].conv, quantize.QuantConv2d)
    assert isinstance(q.model[0].conv, nn.Conv2d) and not isinstance(q.model[0].conv, quantize.QuantConv2d)   # first conv skipped


def test_smoke_constructors(capsys):
    import micronet
    out = micronet.quant_test_auto()
    assert set(out) == {"wbwtab", "dorefa", "iao"}
    micronet.quant_test_manual()
    assert "quant_model is ready" in capsys.readouterr().out


def test_binary_bits_rejected():
    from micronet.compression.quantization.wqaq.dorefa.quantize import ActivationQuantizer
    with pytest.raises(AssertionError):
        ActivationQuantizer(a_bits=1)(torch.zeros(4))


def test_no_cpu_fallback():
    """The product refuses CPU tensors instead of silently computing somewhere else."""
    from micronet_amd import ops
    from micronet_amd._lib import MicronetHipError
    with pytest.raises(MicronetHipError):
        ops.DorefaAct.apply(torch.zeros(8), 4)
    from micronet.compression.quantization.wbwtab.quantize import QuantConv2d
    with pytest.raises(MicronetHipError):
        QuantConv2d(4, 4, 1, W=3)(torch.zeros(1, 4, 4, 4))


def test_abi_exports_every_declared_symbol():
    """libmicronet_hip.so loads without a GPU and exports exactly what include/micronet_hip.h declares."""
    import os, re
    from micronet_amd import _lib, build
    build.build(verbose=False)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "micronet_hip.h")).read()
    declared = set(re.findall(r"\b(mn_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    lib = _lib.Lib(_lib.LIB_PATH)           # AttributeError if a symbol is missing
    assert lib.mn_version() >= 100 and lib.mn_is_emulation() == 0


def test_sign_tensor_is_a_float_tensor_to_everyone_else():
    """SignTensor: logically float32 +-1, physically int8.  Foreign operators see the float values and gradients flow through
    them; our Functions get the codes without any dispatch (checked on CPU with toy Functions: no kernel involved)."""
    from micronet_amd.sign_tensor import SignTensor

    class Producer(torch.autograd.Function):
        @staticmethod
        def forward(ctx, y):
            ctx.save_for_backward(y)
            return SignTensor(torch.where(y < 0, -1, 1).to(torch.int8))

        @staticmethod
        def backward(ctx, da):
            (y,) = ctx.saved_tensors
            assert type(da) is torch.Tensor and da.dtype == torch.float32
            return da * (y.abs() < 1)

    y = torch.randn(4, 6, 2, 8, requires_grad=True)
    a = Producer.apply(y)
    assert isinstance(a, SignTensor) and a.dtype == torch.float32 and a.shape == y.shape and a.requires_grad
    assert a.codes.dtype == torch.int8 and a.detach().codes is a.codes
    ref = torch.where(y.detach() < 0, -1.0, 1.0)
    assert torch.equal(a.to_float(), ref)
    # a foreign consumer (here a stock conv, like the un-quantised last layer of nin_gc) sees float32 and back-propagates
    w = torch.randn(3, 6, 1, 1, requires_grad=True)
    out = torch.nn.functional.conv2d(a, w)
    assert type(out) is torch.Tensor and torch.allclose(out, torch.nn.functional.conv2d(ref, w))
    out.sum().backward()
    y2 = y.detach().clone().requires_grad_(True)
    g_ref = torch.autograd.grad(torch.nn.functional.conv2d(ref.requires_grad_(True), w).sum(), ref)[0] * (y2.abs() < 1)
    assert torch.allclose(y.grad, g_ref)
    assert torch.equal(a.cpu().numpy() if False else (a + 0), ref)
    with pytest.raises(TypeError):
        SignTensor(torch.zeros(3))


def test_wbwtab_prepare_packs_activations_by_default():
    from micronet.compression.quantization.wbwtab import quantize
    q = quantize.prepare(build_model("nin_gc"), inplace=True, A=2, W=3)
    bns = [m for m in q.modules() if isinstance(m, quantize.BatchNorm2dBinAct)]
    assert len(bns) == 8 and all(m.packed for m in bns)
    assert sum(isinstance(m, quantize.MaxPool2dSign) for m in q.modules()) == 2
    assert sum(bool(getattr(m, "lazy_for_bn", False)) for m in q.modules()) == 8       # every quantised conv -- and the un-quantised first one -- feeds a packed BN+sign
    q2 = quantize.prepare(build_model("nin_gc"), inplace=True, A=2, W=3, packed_activations=False)
    assert not any(m.packed for m in q2.modules() if isinstance(m, quantize.BatchNorm2dBinAct))
    assert not any(isinstance(m, quantize.MaxPool2dSign) for m in q2.modules())


def test_dorefa_prepare_fuses_resnet_basic_blocks():
    """prepare() on the reference's resnet18 (models/resnet.py:7-65, 69-112): every BasicBlock becomes a fused subclass of ITS class (isinstance and state_dict
    unchanged), told what its consumer reads -- codes of the next block's quantizer, fp32 only for an identity shortcut -- and fuse_blocks=False leaves them alone."""
    from micronet.compression.quantization.wqaq.dorefa import quantize
    from micronet_amd.models.resnet import BasicBlock
    q = quantize.prepare(build_model("resnet18"), inplace=True, a_bits=2, w_bits=2)
    blocks = [m for m in q.modules() if isinstance(m, BasicBlock)]
    assert len(blocks) == 8 and all(type(b).__name__ == "FusedBasicBlock" and isinstance(b, quantize._FusedBasicBlockMixin) for b in blocks)
    assert [b._mn_out_bits for b in blocks] == [2] * 7 + [0]
    # fp32 copy wanted exactly when the NEXT block adds its input back (identity shortcut); the last block feeds the pool / classifier in fp32
    assert [b._mn_out_f32 for b in blocks] == [True, False, True, False, True, False, True, True]
    assert q.conv1[1].q_out_bits == 2 and q.conv1[1].q_also_f32 is True
    plain = quantize.prepare(build_model("resnet18"), inplace=True, a_bits=2, w_bits=2, fuse_blocks=False)
    assert not any(isinstance(m, quantize._FusedBasicBlockMixin) for m in plain.modules())
    assert list(plain.state_dict()) == list(q.state_dict())
    # 8-bit activations do not fit the code path's exact-integer range for these layer widths: blocks stay unfused
    q8 = quantize.prepare(build_model("resnet18"), inplace=True, a_bits=8, w_bits=8)
    assert not any(isinstance(m, quantize._FusedBasicBlockMixin) for m in q8.modules())


def test_bench_cli_contract_without_gpu():
    """bench.py (the driver's contract): refuses a line whose n_gpus would differ from --gpus, fails loudly without an MI355X (no CPU fallback), and its
    self-launch for --gpus N > 1 goes through torch.distributed.run on 127.0.0.1 (checked on the command it would run, not by running it)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing to report a line whose n_gpus differs" in (r.stderr + r.stdout)
    env.pop("WORLD_SIZE")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)
    src = open(os.path.join(root, "bench.py")).read()
    assert "torch.distributed.run" in src and '"--master-addr", "127.0.0.1"' in src and "--nproc-per-node" in src


def test_bench_final_line_stays_parsable():
    """Round 3's line was 34 KB (six workloads x per-kernel tables) and the driver could not parse it.  Feed the sections of that very run
    (profiles/r03_c_bench.json) through bench.compose_line: the final line must stay under bench.MAX_LINE_BYTES, carry the contract's keys with the
    headline roofline and cpu_baseline, and push every per-kernel table to the detail record instead."""
    import argparse
    import importlib.util
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("mn_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    r3 = json.load(open(os.path.join(root, "profiles", "r03_c_bench.json")))
    sec = {k: r3[k] for k in ("value", "ms_per_step", "ms_per_step_min", "value_best_window", "repeats", "window_ms", "roofline", "kernels", "step_level")}
    sec.update(hip_graph=True, final_loss=1.6, unit="images/s", workload="x")
    args = argparse.Namespace(steps=20, warmup=5, batch=256)
    also_err = {"cX": "RuntimeError: " + "x" * 1000}
    out, detail = bench.compose_line("c2", sec, dict(r3["also"]), also_err, "PMC time budget exhausted " * 20, r3["cpu_baseline"], args, 1)
    line = json.dumps(out)
    assert len(line) + 64 <= bench.MAX_LINE_BYTES, len(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "values"):
        assert k in out, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in out["roofline"], k
    assert "kernels" not in out and set(out["values"]) == {"c2", "c1_w2a2", "c1", "c3", "c4", "c5"}
    assert "workload" in out["config"] and "model" not in out["config"]
    assert set(detail["sections"]) >= {"c2", "c1", "c3", "c4", "c5"} and "kernels" in detail["sections"]["c4"]


def test_bench_line_is_printed_behind_what_libraries_buffered_in_c_stdio():
    """Round 5: RCCL prints a version banner through C stdio when a communicator is created; on a pipe that buffer is flushed at process exit, i.e. BEHIND a line
    printed from Python -- and the driver parses the LAST stdout line.  bench.drain_c_stdio() empties the C streams first."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import ctypes, importlib.util, json\n"
            "spec = importlib.util.spec_from_file_location('mn_bench', %r); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
            "ctypes.CDLL(None).printf(b'RCCL version : banner\\n')\n"
            "b.drain_c_stdio()\n"
            "print(json.dumps({'metric': 'x'}), flush=True)\n") % os.path.join(root, "bench.py")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    lines = r.stdout.strip().splitlines()
    assert r.returncode == 0 and lines[-1] == '{"metric": "x"}' and lines[0].startswith("RCCL version"), (r.stdout, r.stderr[-300:])


@pytest.mark.parametrize("scheme,kw", [("wqaq.dorefa", dict(a_bits=2, w_bits=2)), ("wqaq.iao", dict(a_bits=4, w_bits=4, q_type=0, q_level=0))])
def test_prepared_resnet_pickles(scheme, kw):
    """The reference saves whole prepared models (wqaq/dorefa/quant_model_test/quant_model_para.py:67,84): torch.save / torch.load of a prepared resnet18 must
    round-trip although prepare() swapped its BasicBlocks for generated subclasses (micronet_amd.nn.derive_class)."""
    import importlib
    import io
    import torch
    from micronet_amd.train import build_model
    quantize = importlib.import_module("micronet.compression.quantization.%s.quantize" % scheme)
    m = quantize.prepare(build_model("resnet18"), inplace=True, **kw)
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    m2 = torch.load(buf, weights_only=False)
    assert [type(a).__name__ for a in m.modules()] == [type(a).__name__ for a in m2.modules()]
    assert [type(a).__mro__[1:] for a in m.modules()] == [type(a).__mro__[1:] for a in m2.modules()]
    sd, sd2 = m.state_dict(), m2.state_dict()
    assert list(sd) == list(sd2) and all(torch.equal(sd[k], sd2[k]) for k in sd)


def test_dorefa_weight_grid_verdict_is_forgotten_on_load_state_dict():
    """the cached "stored weights lie on the quantizer grid" verdict of a quant_inference layer dies with a load_state_dict (new stored weights)."""
    import importlib
    Q = importlib.import_module("micronet.compression.quantization.wqaq.dorefa.quantize")
    m = Q.QuantConv2d(4, 4, 3, a_bits=4, w_bits=4, quant_inference=True)
    m.__dict__["_mn_grid"] = (("stale",), True, None)
    m.load_state_dict(m.state_dict())
    assert "_mn_grid" not in m.__dict__


def test_iao_prepare_marks_convs_that_feed_our_batchnorms():
    """prepare() of an IAO ResNet: every QuantConv2d that a BatchNorm2dReLU / BatchNorm2dPlain of ours reads next (same nn.Sequential, definition order) is told to
    leave the exact sums of its integer accumulator on its output (``emit_accstats``: the BatchNorm then normalises in one pass) -- and nothing else changes:
    module names and ``state_dict`` are the reference's (checked by test_prepare_matches_reference_surface)."""
    from micronet.compression.quantization.wqaq.iao import quantize
    from micronet_amd.quantization.wqaq.dorefa.quantize import BatchNorm2dPlain, BatchNorm2dReLU
    q = quantize.prepare(build_model("resnet18"), inplace=True, a_bits=4, w_bits=4, q_type=0, q_level=0)
    marked = [m for m in q.modules() if getattr(m, "emit_accstats", False)]
    bns = [m for m in q.modules() if isinstance(m, (BatchNorm2dReLU, BatchNorm2dPlain))]
    assert len(marked) == len(bns) == 20 and all(type(m) is quantize.QuantConv2d for m in marked)
    # without the fused BatchNorms there is nobody to hand the sums to
    q2 = quantize.prepare(build_model("resnet18"), inplace=True, a_bits=4, w_bits=4, q_type=0, q_level=0, fuse_bn_act=False)
    assert not any(getattr(m, "emit_accstats", False) for m in q2.modules())


@pytest.mark.parametrize("W", [3, 2])
def test_wbwtab_bn_fuse_keeps_code_weights_on_the_packed_path(W):
    """micronet_amd.inference.wbwtab_model_bn_fuse (ref wbwtab/bn_fuse/bn_fuse.py:20-107) on the host: folded from PRE-QUANTISED weights (codes x alpha per output
    channel) the quantised convs in front of a sign are marked ``stored_codes`` / ``lazy_for_bn`` and the signs behind them ``deploy_packed`` -- the deployed graph then
    stays on one byte per activation; folded from raw weights nothing is marked (the reference convolves them as they are, so do we).  Fold arithmetic itself: the
    golden tests (tests/test_oracle_golden.py pins the oracle, tests/test_gpu_inference.py the product)."""
    from micronet.compression.quantization.wbwtab import quantize
    from micronet_amd import inference
    from micronet_amd.nn import Conv2dFirst, Conv2dSignIn
    torch.manual_seed(3)
    I = quantize.prepare(build_model("nin_gc"), inplace=True, A=2, W=W, quant_inference=True)
    raw = inference.wbwtab_model_bn_fuse(I, W=W)
    qc = [m for m in raw.modules() if isinstance(m, quantize.QuantConv2d)]
    assert len(qc) == 7 and not any(m.stored_codes or m.lazy_for_bn for m in qc)
    with torch.no_grad():
        for m in I.modules():
            if isinstance(m, quantize.QuantConv2d):          # what quant_model_para.py stores: t * alpha[o] (t ternary / binary)
                t = torch.randint(-1 if W == 3 else 0, 2, m.weight.shape).float()
                if W == 2:
                    t = t * 2 - 1
                t.view(t.shape[0], -1)[:, 0] = 1
                m.weight.copy_(t * (torch.rand(t.shape[0], 1, 1, 1) * 0.2 + 0.05))
        for m in I.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.weight.copy_(torch.randn_like(m.weight))          # both signs of gamma: the fold flips the weights' sign where gamma < 0
    F = inference.wbwtab_model_bn_fuse(I, W=W)
    qc = [m for m in F.modules() if isinstance(m, quantize.QuantConv2d)]
    assert len(qc) == 7 and all(m.stored_codes and m.lazy_for_bn and m.quant_inference for m in qc)
    acts = [m for m in F.modules() if isinstance(m, quantize.ActivationQuantizer)]
    assert len(acts) == 8 and all(a.deploy_packed for a in acts)
    convs = [m for m in F.modules() if isinstance(m, nn.Conv2d)]
    assert type(convs[0]) is Conv2dFirst and type(convs[-1]) is Conv2dSignIn          # the fp32 ends keep their kernels
    assert list(F.state_dict().keys()) == list(raw.state_dict().keys())
    assert not any(isinstance(m, nn.BatchNorm2d) for m in F.modules())
    # the training graph is untouched by the conversion (deep copy) and its signs do not take the deployed route
    assert not any(getattr(m, "deploy_packed", False) for m in I.modules() if isinstance(m, quantize.ActivationQuantizer))


def test_first_conv_is_marked_for_the_fused_first_block():
    """prepare() marks the un-quantised first conv whose output feeds only our BatchNorm block: wbwtab -> the fused conv + BatchNorm + sign kernel (True), DoReFa ->
    "qa" (fused forward behind MN_FIRST_FUSED_QA; the one-pass backward on pass nibbles is the default either way).  Structural, CPU-checkable."""
    from micronet.compression.quantization.wbwtab import quantize as wb
    from micronet.compression.quantization.wqaq.dorefa import quantize as dr
    from micronet_amd.nn import Conv2dFirst
    q = wb.prepare(build_model("nin_gc"), inplace=True, A=2, W=3)
    assert isinstance(q.model[0].conv, Conv2dFirst) and q.model[0].conv.lazy_for_bn is True
    q = wb.prepare(build_model("nin_gc"), inplace=True, A=2, W=3, fuse_conv_bn=False)
    assert q.model[0].conv.lazy_for_bn is False
    q = dr.prepare(build_model("nin_gc"), inplace=True, a_bits=2, w_bits=2)
    assert isinstance(q.model[0].conv, Conv2dFirst) and q.model[0].conv.lazy_for_bn == "qa"
    q = dr.prepare(build_model("nin_gc"), inplace=True, a_bits=2, w_bits=2, fuse_blocks=False)
    assert not q.model[0].conv.lazy_for_bn


def test_residual_token_only_for_descendants():
    """ops._descends_from (the guard of the shortcut-gradient fold, ops.ResidualToken): true only when the tensor is computed from the node's output."""
    from micronet_amd import ops
    x = torch.randn(4, 3, requires_grad=True)
    a = x * 2.0
    b = (a + 1.0).relu() * 3.0
    c = x.exp()
    assert ops._descends_from(b, a.grad_fn) and not ops._descends_from(c, a.grad_fn) and not ops._descends_from(x, a.grad_fn)
    deep = a
    for _ in range(200):
        deep = deep + 1.0
    assert not ops._descends_from(deep, a.grad_fn, limit=96)          # (beyond the search budget: the fold is simply not taken)


def test_stored_codes_verdict_survives_copies_of_the_module():
    """the deployed wbwtab layer's ``stored_codes`` verdict is keyed on (data pointer, version) of ONE weight tensor.  copy.deepcopy, a pickle round trip and
    nn.DataParallel's replicas hold the same VALUES in a new tensor: the verdict travels with them (re-keyed at first use, no raw pointer in the pickle); a copy of a
    module whose verdict no longer held does not get one, and new values in the copy drop it again."""
    import copy
    import io
    import pickle
    from micronet.compression.quantization.wbwtab import quantize
    from micronet_amd import inference, ops
    torch.manual_seed(5)
    m = quantize.QuantConv2d(8, 16, 1, W=3, quant_inference=True)
    inference.mark_stored_codes(m)
    assert m._codes_valid()
    d = copy.deepcopy(m)
    assert d.weight.data_ptr() != m.weight.data_ptr() and "_mn_codes_key" not in d.__dict__ and d._codes_valid() and d._codes_valid()
    p = pickle.loads(pickle.dumps(m))
    assert p._codes_valid()
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    t = torch.load(buf, weights_only=False)
    assert t._codes_valid()
    r = m._replicate_for_data_parallel()
    r.weight = m.weight.detach().clone()          # what replicate() hands a replica: a broadcast copy, not a Parameter
    assert r._codes_valid()
    assert m._codes_valid()                       # the original is untouched by all of this
    # new values: the verdict goes, visibly
    ops.fallback_counts(reset=True)
    d.weight.data = torch.randn_like(d.weight)
    assert not d._codes_valid() and any("stored_codes" in k for k in ops.fallback_counts(reset=True))
    assert not copy.deepcopy(d)._codes_valid()    # ... and a copy of THAT module has none either
    with torch.no_grad():
        m.weight.mul_(1.0)                        # an in-place update of the original (what an optimizer step does): the version moves on
    assert not m._codes_valid() and not copy.deepcopy(m)._codes_valid()
