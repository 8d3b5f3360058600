"""Run-to-run determinism at the benched batch size (256).

Every kernel of this library is designed to be deterministic (no float atomics; partial sums are reduced in a fixed order).  A kernel
with a latent data race can still pass every parity test at small batch and fail intermittently under load: the round-2 full-batch
parity test caught exactly that in the k x k backward-data kernel (its run-time "skip an all-zero term plane" flags), one 256-pixel
tile wrong in roughly one launch out of six at batch 256.  These tests repeat the same computation and demand BIT-IDENTICAL results:
  * every conv entry point of the C ABI on the nin_gc hot shapes, per scheme;
  * whole-net forward + backward (all fused blocks, lazy gradients, weight quantizers) from identical state."""
import ctypes as C
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu

N = 256
SHAPES = {
    "L2 1x1 g2": dict(x_shape=(N, 256, 32, 32), w_shape=(256, 128, 1, 1), groups=2),
    "L4 3x3 g16": dict(x_shape=(N, 256, 16, 16), w_shape=(512, 16, 3, 3), padding=1, groups=16),
    "L5 1x1 g4": dict(x_shape=(N, 512, 16, 16), w_shape=(512, 128, 1, 1), groups=4),
    "L7 3x3 g32": dict(x_shape=(N, 512, 8, 8), w_shape=(1024, 16, 3, 3), padding=1, groups=32),
    "L8 1x1 g8": dict(x_shape=(N, 1024, 8, 8), w_shape=(1024, 128, 1, 1), groups=8),
    "L9 1x1 ->10": dict(x_shape=(N, 1024, 8, 8), w_shape=(10, 1024, 1, 1)),
}
import os
REPS = int(os.environ.get("MN_DET_REPS", "8"))


@pytest.fixture(scope="module")
def be():
    from abi_driver import Backend
    return Backend("gpu")


def _weights(scheme, w_shape, gen):
    if scheme in ("ternary_sign8", "ternary_real"):
        t = torch.randint(-1, 2, w_shape, device="cuda", generator=gen).float()
        alpha = torch.rand((w_shape[0], 1, 1, 1), device="cuda", generator=gen) * 0.2 + 0.05
        return t * alpha
    if scheme == "dorefa2":
        k = torch.randint(0, 4, w_shape, device="cuda", generator=gen).float()
        return (2 * k - 3) / 3
    if scheme == "dorefa8":
        k = torch.randint(0, 256, w_shape, device="cuda", generator=gen).float()
        s = torch.tensor(1.0 / 255.0)
        return 2 * (k * s) - 1
    raise KeyError(scheme)


@pytest.mark.parametrize("scheme", ["ternary_sign8", "ternary_real", "dorefa2", "dorefa8"])
@pytest.mark.parametrize("name", list(SHAPES))
def test_conv_entry_points_bitwise_repeatable(be, name, scheme):
    kw = SHAPES[name]
    g = be.geom(kw["x_shape"], kw["w_shape"], padding=kw.get("padding", 0), groups=kw.get("groups", 1))
    gen = torch.Generator(device="cuda").manual_seed(11)
    w = _weights(scheme, kw["w_shape"], gen)
    if scheme == "ternary_sign8":
        x = ((torch.rand(kw["x_shape"], device="cuda", generator=gen) > 0.5).to(torch.int8) * 2 - 1).contiguous()
        aq, wq = be.actq(3), be.wq(mode=1)
    elif scheme == "ternary_real":
        x = torch.randn(kw["x_shape"], device="cuda", generator=gen)
        aq, wq = be.actq(0), be.wq(mode=1)
    else:
        bits = 2 if scheme == "dorefa2" else 8
        x = torch.rand(kw["x_shape"], device="cuda", generator=gen) * 12 - 1
        aq, wq = be.actq(1, bits), be.wq(mode=2, bits=bits)
    Ho, Wo = kw["x_shape"][2], kw["x_shape"][3]
    gy = torch.randn((N, kw["w_shape"][0], Ho, Wo), device="cuda", generator=gen)
    lib = be.lib
    for which in (0, 1, 2):
        if not lib.mn_conv2d_qgemm_supported(C.byref(g), C.byref(aq), C.byref(wq), which):
            continue
        outs = []
        for _ in range(REPS):
            if which == 0:
                o = be.conv_fwd(g, aq, x, w, None, 3, wq=wq)
            elif which == 1:
                o = be.conv_bwd_data(g, aq, gy, w, x if scheme != "ternary_sign8" else None, 3, wq=wq)
            else:
                o = be.conv_bwd_weight(g, aq, gy, x, 3, bias=False)[0]
            outs.append(o)
        torch.cuda.synchronize()
        kern = lib.mn_last_kernel().decode()
        for i in range(1, REPS):
            same = torch.equal(outs[i], outs[0])
            if not same:
                d = (outs[i] - outs[0]).abs()
                raise AssertionError("%s %s pass %d (%s): launch %d differs from launch 0 in %d elements (max %.3g)" %
                                     (name, scheme, which, kern, i, int((d > 0).sum()), float(d.max())))


NETS = {
    "c2": ("nin_gc", "wbwtab", dict(A=2, W=3)),
    "c1_w2a2": ("nin_gc", "wqaq.dorefa", dict(a_bits=2, w_bits=2)),
    "c3": ("nin_gc", "wqaq.iao", dict(a_bits=8, w_bits=8, q_type=0, q_level=0, weight_observer=0, bn_fuse=True)),
}


@pytest.mark.parametrize("key", list(NETS))
def test_whole_net_gradients_bitwise_repeatable(key):
    """The same model state and batch -> bit-identical logits and gradients, 6 times (the model is re-created from the same seed every time:
    observers and running statistics start from the same state)."""
    from micronet_amd.train import build_model, synth_batch
    arch, scheme, kw = NETS[key]
    quantize = importlib.import_module("micronet.compression.quantization.%s.quantize" % scheme)
    x, y = synth_batch(N, device="cuda")
    ref = None
    for rep in range(6):
        model = quantize.prepare(build_model(arch), inplace=True, **kw).cuda().train()
        out = model(x)
        torch.nn.functional.cross_entropy(out, y).backward()
        torch.cuda.synchronize()
        cur = [out.detach().clone()] + [p.grad.detach().clone() for p in model.parameters()]
        if ref is None:
            ref = cur
            continue
        names = ["logits"] + [n for n, _ in model.named_parameters()]
        for n_, a, b in zip(names, cur, ref):
            assert torch.equal(a, b), "%s: %s differs between identical runs (rep %d): max %.3g" % (key, n_, rep, float((a - b).abs().max()))
