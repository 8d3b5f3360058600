"""Parity tests proper: the gfx950 kernels through the C ABI on a real MI355X vs the oracle and the golden fixtures
generated from the reference.  Same checks as tests/test_kernels_emulated.py plus hot-path shapes and full-size
properties."""
import ctypes as C

import numpy as np
import pytest

import abi_driver
import kernel_cases as K

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def be():
    return abi_driver.Backend("gpu")


@pytest.mark.parametrize("bits", [2, 3, 4, 8])
def test_dorefa_act(be, golden, bits):
    K.check_dorefa_act(be, golden.q, bits)


@pytest.mark.parametrize("bits", [2, 4, 8])
def test_dorefa_w(be, golden, bits):
    K.check_dorefa_w(be, golden.q, bits)


def test_wbwtab(be, golden):
    K.check_wbwtab(be, golden.q)


def test_iao(be, golden):
    K.check_iao(be, golden.q, golden.meta["iao"])


def test_bn_stats(be):
    K.check_bn_stats(be)
    K.check_bn_stats(be, shape=(3, 5, 3, 3), seed=1)
    K.check_bn_stats(be, shape=(64, 32, 16, 16), seed=2)


@pytest.mark.parametrize("case", range(len(K.SMALL_CONV_CASES)))
def test_conv_plain(be, case):
    K.check_conv(be, seed=case, **K.SMALL_CONV_CASES[case])


@pytest.mark.parametrize("case", [1, 2, 3, 5, 8])
def test_conv_dorefa_fused(be, case):
    K.check_conv(be, seed=10 + case, mode=1, bits=3, **K.SMALL_CONV_CASES[case])


@pytest.mark.parametrize("case,q_type", [(1, 0), (2, 1), (3, 0), (8, 1), (5, 0)])
def test_conv_iao_fused(be, case, q_type):
    K.check_conv(be, seed=20 + case, mode=2, bits=4, q_type=q_type, **K.SMALL_CONV_CASES[case])


# the real layer shapes of the benchmark nets at a batch the numpy oracle finishes in seconds
HOT_SHAPES = [
    ("nin_gc L1 5x5 (iao)", dict(x_shape=(4, 3, 32, 32), w_shape=(256, 3, 5, 5), padding=2)),
    ("nin_gc L2 1x1 g2", dict(x_shape=(4, 256, 32, 32), w_shape=(256, 128, 1, 1), groups=2)),
    ("nin_gc L4 3x3 g16", dict(x_shape=(4, 256, 16, 16), w_shape=(512, 16, 3, 3), padding=1, groups=16)),
    ("nin_gc L5 1x1 g4", dict(x_shape=(4, 512, 16, 16), w_shape=(512, 128, 1, 1), groups=4)),
    ("nin_gc L7 3x3 g32", dict(x_shape=(6, 512, 8, 8), w_shape=(1024, 16, 3, 3), padding=1, groups=32)),
    ("nin_gc L8 1x1 g8", dict(x_shape=(6, 1024, 8, 8), w_shape=(1024, 128, 1, 1), groups=8)),
    ("nin_gc L9 1x1 ->10", dict(x_shape=(6, 1024, 8, 8), w_shape=(10, 1024, 1, 1))),
    ("resnet 3x3 64", dict(x_shape=(2, 64, 32, 32), w_shape=(64, 64, 3, 3), padding=1, bias=False)),
    ("resnet 3x3 s2 64->128", dict(x_shape=(2, 64, 32, 32), w_shape=(128, 64, 3, 3), stride=2, padding=1, bias=False)),
    ("resnet 1x1 s2 shortcut", dict(x_shape=(2, 64, 32, 32), w_shape=(128, 64, 1, 1), stride=2, bias=False)),
    ("resnet 3x3 512 @4x4", dict(x_shape=(9, 512, 4, 4), w_shape=(512, 512, 3, 3), padding=1, bias=False)),
]


@pytest.mark.parametrize("name,kw", HOT_SHAPES, ids=[n for n, _ in HOT_SHAPES])
def test_hot_shapes(be, name, kw):
    K.check_conv(be, seed=77, expect_mfma=True, **kw)
    K.check_conv(be, seed=78, mode=1, bits=2, algos=(2,), **kw)


# ---- code-domain (bf16 MFMA) kernels, algo 3: same cases as the emulated run + the hot pointwise shapes
@pytest.mark.parametrize("case", range(len(K.QGEMM_PW_CASES)))
@pytest.mark.parametrize("wmode", [1, 2, 3])
def test_qgemm_pointwise_binary_x(be, case, wmode):
    K.check_conv(be, seed=40 + case, wmode=wmode, wbits=4, binary_x=True, algos=(3,), expect_qgemm=True, **K.QGEMM_PW_CASES[case])


@pytest.mark.parametrize("case", [0, 1, 2])
def test_qgemm_pointwise_real_x(be, case):
    K.check_conv(be, seed=50 + case, wmode=1, algos=(3,), expect_qgemm=True, **K.QGEMM_PW_CASES[case])


@pytest.mark.parametrize("case,mode,q_type", [(1, 1, 0), (2, 1, 0), (1, 2, 0), (3, 2, 0)])
def test_qgemm_pointwise_fused_actq(be, case, mode, q_type):
    K.check_conv(be, seed=60 + case, mode=mode, bits=4, q_type=q_type, wmode=2 if mode == 1 else 3, wbits=4, algos=(3,),
                 expect_qgemm=True, **K.QGEMM_PW_CASES[case])


KXK_SUP = [True] * 5 + [(True, False, False)]


@pytest.mark.parametrize("case", range(len(K.QGEMM_KXK_CASES)))
def test_qgemm_kxk_binary_x(be, case):
    K.check_conv(be, seed=70 + case, wmode=1, binary_x=True, algos=(3,), expect_qgemm=KXK_SUP[case],
                 **K.QGEMM_KXK_CASES[case])


@pytest.mark.parametrize("case", [0, 4])
def test_qgemm_kxk_real_x(be, case):
    K.check_conv(be, seed=75 + case, wmode=3, wbits=8, algos=(3,), expect_qgemm=KXK_SUP[case], **K.QGEMM_KXK_CASES[case])


@pytest.mark.parametrize("case,mode", [(0, 1), (1, 2), (2, 2), (4, 1), (5, 1)])
def test_qgemm_kxk_fused_actq(be, case, mode):
    K.check_conv(be, seed=80 + case, mode=mode, bits=4, wmode=2 if mode == 1 else 3, wbits=4, algos=(3,),
                 expect_qgemm=KXK_SUP[case], **K.QGEMM_KXK_CASES[case])


QGEMM_HOT = [(n, kw) for n, kw in HOT_SHAPES if kw.get("stride", 1) == 1]


@pytest.mark.parametrize("name,kw", QGEMM_HOT, ids=[n for n, _ in QGEMM_HOT])
def test_qgemm_hot_shapes(be, name, kw):
    # 4x4 images: rows are shorter than one 8-pixel fragment, backward-weight stays on the fp32-MFMA kernel
    exp = (True, True, False) if kw["x_shape"][3] < 8 and kw["w_shape"][2] > 1 else True
    K.check_conv(be, seed=81, wmode=1, binary_x=True, algos=(3, 0), expect_qgemm=exp, **kw)          # wbwtab W3/A2
    K.check_conv(be, seed=82, wmode=2, wbits=8, mode=1, bits=8, algos=(3,), expect_qgemm=exp, **kw)   # DoReFa W8A8
    dense = kw.get("groups", 1) == 1 and kw["w_shape"][0] % 64 == 0 and kw["w_shape"][1] % 64 == 0          # the ResNet layers: qgemm_dense.hip covers all three directions
    K.check_conv(be, seed=83, wmode=3, wbits=8, mode=2, bits=8, algos=(3,), expect_qgemm=True if dense else exp, **kw)   # IAO W8A8 sym per-channel
    K.check_conv(be, seed=84, wmode=1, algos=(3,), expect_qgemm=exp, **kw)                           # real x (W-only quantization)


def test_full_size_properties(be):
    """BASELINE config 2 layer L2 at batch 256 (268 MB activations): properties that need no CPU-sized oracle.
    (1) MFMA kernel == direct kernel on a strided sample of outputs; (2) linearity in the weights;
    (3) backward-weight of ones == per-channel input sums (a checksum of checksums)."""
    torch = be.torch
    N = 256
    g = be.geom((N, 256, 32, 32), (256, 128, 1, 1), groups=2)
    gen = torch.Generator(device="cuda").manual_seed(5)
    x = (torch.rand((N, 256, 32, 32), device="cuda", generator=gen) > 0.5).float() * 2 - 1      # +-1 activations
    w = torch.randn((256, 128, 1, 1), device="cuda", generator=gen) * 0.1
    aq = be.actq(0)
    _full_size_checks(be, g, aq, x, w, N, 2, None)
    # the same through the code-domain kernels: ternary weights t * alpha[o]
    t = torch.randint(-1, 2, (256, 128, 1, 1), device="cuda", generator=gen).float()
    t[:, 0] = 1
    alpha = torch.rand((256, 1, 1, 1), device="cuda", generator=gen) * 0.2 + 0.05
    _full_size_checks(be, g, aq, x, t * alpha, N, 3, be.wq(mode=1))


def _full_size_checks(be, g, aq, x, w, N, algo, wq):
    torch = be.torch
    y = be.conv_fwd(g, aq, x, w, None, algo, wq=wq)
    y2 = be.conv_fwd(g, aq, x, 2 * w, None, algo, wq=wq)
    assert torch.equal(y2, 2 * y)                                # scaling by 2 is exact in fp32
    # sampled comparison with an fp64 einsum on 3 images
    for n in (0, 100, 255):
        ref = torch.einsum("gchw,goc->gohw", x[n].double().view(2, 128, 32, 32), w.double().view(2, 128, 128)).reshape(256, 32, 32)
        assert (y[n].double() - ref).abs().max() <= 1e-5 * ref.abs().max()
    gy = torch.ones_like(y)
    dw, db = be.conv_bwd_weight(g, aq, gy, x, algo)
    colsum = x.double().sum(dim=(0, 2, 3))                       # [256]
    ref = colsum.view(2, 1, 128).expand(2, 128, 128).reshape(256, 128)
    assert (dw.view(256, 128).double() - ref).abs().max() <= 1e-5 * ref.abs().max().clamp_min(1.0)
    assert torch.equal(db, torch.full_like(db, float(N * 32 * 32)))
    dx = be.conv_bwd_data(g, aq, gy, w, None, algo, wq=wq)
    ref = w.double().view(2, 128, 128).sum(dim=1).reshape(256)    # sum over out-channels of each group
    assert (dx[17, :, 5, 9].double() - ref).abs().max() <= 1e-5 * ref.abs().max()
    assert torch.equal(dx[0, :, 0, 0], dx[255, :, 31, 31])


def test_errors_are_reported(be):
    g = be.geom((1, 4, 8, 8), (4, 3, 1, 1))      # C not divisible consistently -> invalid
    g.groups = 3
    rc = be.lib.mn_conv2d_fwd(C.byref(g), C.byref(be.actq(0)), None, None, None, None, None, None, 0, 0, be.stream)
    assert rc == -22 and b"invalid" in be.lib.mn_last_error()
    g2 = be.geom((2, 8, 6, 6), (12, 4, 3, 3), padding=1, groups=2)
    x = be.to_dev(np.zeros((2, 8, 6, 6)))
    w = be.to_dev(np.zeros((12, 4, 3, 3)))
    y = be.empty((2, 12, 6, 6))
    rc = be.lib.mn_conv2d_fwd(C.byref(g2), C.byref(be.actq(0)), None, be.ptr(x), be.ptr(w), None, be.ptr(y), None, 0, 2, be.stream)
    assert rc == -95       # MFMA requested for a shape the tiler rejects


def test_adam_step(be):
    K.check_adam(be)
    K.check_adam(be, sizes=tuple(range(1, 41)), steps=2, seed=1)
    K.check_adam(be, sizes=(591390,), steps=2, seed=2)


def test_adam_optimizer_matches_torch():
    """micronet_amd.optim.Adam vs torch.optim.Adam on the GPU over 3 steps of a small model (per-tensor groups as main.py builds them)."""
    import torch
    from micronet_amd.optim import Adam
    torch.manual_seed(0)
    def mk():
        torch.manual_seed(1)
        return torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8), torch.nn.Conv2d(8, 4, 1)).cuda()
    a, b = mk(), mk()
    oa = Adam([{"params": [p], "lr": 0.01, "weight_decay": 1e-5} for p in a.parameters()], lr=0.01, weight_decay=1e-5)
    ob = torch.optim.Adam([{"params": [p], "lr": 0.01, "weight_decay": 1e-5} for p in b.parameters()], lr=0.01, weight_decay=1e-5)
    x = torch.randn(4, 3, 8, 8, device="cuda")
    for _ in range(3):
        for m_, o in ((a, oa), (b, ob)):
            o.zero_grad()
            m_(x).square().mean().backward()
            o.step()
    for pa, pb in zip(a.parameters(), b.parameters()):
        # conv biases in front of a BatchNorm see gradients ~1e-9: m/(sqrt(v)+eps) amplifies last-bit differences there
        assert (pa - pb).abs().max() <= 1e-5 * pb.abs().max().clamp_min(1.0)
    assert set(oa.state_dict()["state"][0].keys()) == set(ob.state_dict()["state"][0].keys())


@pytest.mark.parametrize("training", [True, False])
def test_bnsign(be, training):
    K.check_bnsign(be, training=training)
    K.check_bnsign(be, shape=(3, 7, 2, 2), seed=3, training=training)
    K.check_bnsign(be, shape=(32, 64, 16, 16), seed=4, training=training)


@pytest.mark.parametrize("case,sg", [(1, 2), (1, 5), (2, 4)])
def test_qgemm_pointwise_in_shuffle(be, case, sg):
    K.check_conv(be, seed=90 + case, wmode=1, binary_x=True, algos=(3,), expect_qgemm=True, in_shuffle=sg, **K.QGEMM_PW_CASES[case])
    K.check_conv(be, seed=95 + case, mode=1, bits=4, wmode=2, wbits=4, algos=(3,), expect_qgemm=True, in_shuffle=sg, **K.QGEMM_PW_CASES[case])


@pytest.mark.parametrize("case,sg", [(0, 2), (0, 8), (4, 5)])
def test_qgemm_kxk_in_shuffle(be, case, sg):
    K.check_conv(be, seed=97 + case, wmode=1, binary_x=True, algos=(3,), expect_qgemm=True, in_shuffle=sg, **K.QGEMM_KXK_CASES[case])


# ---- packed (int8) sign activations
@pytest.mark.parametrize("case", range(len(K.QGEMM_PW_CASES)))
def test_qgemm_pointwise_sign8(be, case):
    K.check_conv(be, seed=140 + case, wmode=1, sign8=True, algos=(3, 0), expect_qgemm=True, **K.QGEMM_PW_CASES[case])


@pytest.mark.parametrize("case", [0, 1, 3, 4])
def test_qgemm_kxk_sign8(be, case):
    K.check_conv(be, seed=150 + case, wmode=1, sign8=True, algos=(3,), expect_qgemm=True, **K.QGEMM_KXK_CASES[case])


def test_qgemm_sign8_shuffle(be):
    K.check_conv(be, seed=160, wmode=1, sign8=True, algos=(3,), in_shuffle=2, **K.QGEMM_PW_CASES[1])
    K.check_conv(be, seed=161, wmode=1, sign8=True, algos=(3,), in_shuffle=2, **K.QGEMM_KXK_CASES[0])


# backward-weight on sign codes with Mg, Cg > 32: the LDS-staged kernel (k_pws_wgrad_s) when H*W % 16 == 0 -- odd step count, clamped
# rows, shuffled input, bias gradient -- and the direct-load kernel (k_pws_wgrad) otherwise
SIGN8_WGRAD_CASES = [
    dict(x_shape=(3, 96, 4, 8), w_shape=(80, 48, 1, 1), groups=2),                    # MW = 2 (40 x 48 per group), 3 steps in one block
    dict(x_shape=(5, 256, 4, 8), w_shape=(200, 128, 1, 1), groups=2, in_shuffle=2),   # MW = 4 (100 x 128), 5 steps over 2 blocks
    dict(x_shape=(2, 70, 4, 4), w_shape=(66, 70, 1, 1), bias=False),                  # one step, rows 66..127 clamped
    dict(x_shape=(8, 80, 2, 2), w_shape=(96, 40, 1, 1), groups=2),                    # H*W = 4: direct-load kernel
]


@pytest.mark.parametrize("case", range(len(SIGN8_WGRAD_CASES)))
def test_sign8_wgrad_large_tiles(be, case):
    K.check_conv(be, seed=165 + case, wmode=1, sign8=True, algos=(3,), expect_qgemm=True, **SIGN8_WGRAD_CASES[case])


# 3 x 3 backward-weight on sign codes (k_k3s_wgrad): several blocks per group with uneven step ranges, W = 8 / 16 / 32
K3S_CASES = [
    dict(x_shape=(9, 32, 4, 16), w_shape=(64, 16, 3, 3), padding=1, groups=2),                 # 18 steps over 4 blocks (5, 5, 5, 3)
    dict(x_shape=(5, 48, 8, 8), w_shape=(96, 24, 3, 3), padding=1, groups=2, bias=False),      # Cg = 24: two c-tiles, the second half empty
    # W = 32, H = 3: every step touches top or bottom padding (the generic k x k forward / backward-data tilers reject H = 3)
    dict(x_shape=(3, 8, 3, 32), w_shape=(40, 8, 3, 3), padding=1, in_shuffle=2, expect_qgemm=(False, False, True)),
]


@pytest.mark.parametrize("case", range(len(K3S_CASES)))
def test_sign8_wgrad_3x3(be, case):
    K.check_conv(be, seed=175 + case, wmode=1, sign8=True, algos=(3,), **{"expect_qgemm": True, **K3S_CASES[case]})


# 3 x 3 backward-data of a ternary-weight layer (k_k3s_dgrad): staged image with zero frame, k-permuted transposition
K3D_CASES = [
    dict(x_shape=(3, 32, 16, 16), w_shape=(64, 16, 3, 3), padding=1, groups=2),                  # the nin_gc L4 pattern: one image per stage
    dict(x_shape=(5, 48, 8, 8), w_shape=(48, 24, 3, 3), padding=1, groups=2, bias=False),        # Mg = 24 (padded k), Cg = 24 (two c-tiles), odd N with two images per stage
    dict(x_shape=(2, 16, 4, 8), w_shape=(32, 16, 3, 3), padding=1, in_shuffle=2),                # 32-pixel images: four per stage; shuffled dx channels
]


@pytest.mark.parametrize("case", range(len(K3D_CASES)))
def test_sign8_dgrad_3x3(be, case):
    K.check_conv(be, seed=185 + case, wmode=1, sign8=True, algos=(3,), **{"expect_qgemm": True, **K3D_CASES[case]})


@pytest.mark.parametrize("name,kw", [(n, kw) for n, kw in QGEMM_HOT if "nin_gc" in n], ids=[n for n, _ in QGEMM_HOT if "nin_gc" in n])
def test_qgemm_hot_shapes_sign8(be, name, kw):
    if kw["w_shape"][1] * kw.get("groups", 1) == 3:
        pytest.skip("first layer reads the image, not sign codes")
    K.check_conv(be, seed=181, wmode=1, sign8=True, algos=(3,), expect_qgemm=True, **kw)


def test_pool_sign8(be):
    K.check_pool_sign8(be)
    K.check_pool_sign8(be, shape=(2, 3, 2, 8), seed=1)
    K.check_pool_sign8(be, shape=(16, 64, 32, 32), seed=2)


@pytest.mark.parametrize("case", range(len(K.QGEMM_PW_CASES)))
@pytest.mark.parametrize("training", [True, False])
def test_qconv_bnsign_fused(be, case, training):
    K.check_qconv_bnsign(be, seed=170 + case, training=training, **K.QGEMM_PW_CASES[case])


def test_qconv_bnsign_fused_hot_shapes(be):
    K.check_qconv_bnsign(be, seed=180, in_shuffle=2, **K.QGEMM_PW_CASES[1])
    K.check_qconv_bnsign(be, seed=181, x_shape=(8, 256, 32, 32), w_shape=(256, 128, 1, 1), groups=2, in_shuffle=2)      # nin_gc L3
    K.check_qconv_bnsign(be, seed=182, x_shape=(8, 512, 16, 16), w_shape=(512, 128, 1, 1), groups=4, in_shuffle=16)     # L5
    K.check_qconv_bnsign(be, seed=183, x_shape=(16, 1024, 8, 8), w_shape=(1024, 128, 1, 1), groups=8, in_shuffle=32)    # L8


FIRST_CASES = [
    dict(x_shape=(3, 3, 8, 8), w_shape=(24, 3, 5, 5), padding=2),
    dict(x_shape=(2, 3, 16, 16), w_shape=(160, 3, 3, 3), padding=1),
    dict(x_shape=(2, 1, 8, 16), w_shape=(70, 1, 3, 3), padding=1, bias=False),
    dict(x_shape=(8, 3, 32, 32), w_shape=(256, 3, 5, 5), padding=2),                 # nin_gc L1
    dict(x_shape=(8, 3, 32, 32), w_shape=(64, 3, 3, 3), padding=1, bias=False),      # resnet conv1
]


@pytest.mark.parametrize("case", range(len(FIRST_CASES)))
def test_conv_first_layer(be, case):
    kw = FIRST_CASES[case]
    g = be.geom(kw["x_shape"], kw["w_shape"], padding=kw["padding"])
    assert be.lib.mn_conv2d_first_supported(C.byref(g), 0) == 1 and be.lib.mn_conv2d_first_supported(C.byref(g), 2) == 1
    K.check_conv(be, seed=200 + case, algos=(0,), rel=2e-6 if case < 3 else 1e-5, **kw)     # fp32 accumulation over 8k pixels per partial


WG2_CASES = [
    dict(x_shape=(2, 128, 4, 8), w_shape=(128, 64, 1, 1), groups=2),
    dict(x_shape=(3, 256, 4, 8), w_shape=(256, 128, 1, 1), groups=2, bias=False),
    dict(x_shape=(2, 80, 4, 4), w_shape=(100, 40, 1, 1), groups=2),
    dict(x_shape=(8, 512, 16, 16), w_shape=(512, 128, 1, 1), groups=4),              # nin_gc L5 at batch 8
]


@pytest.mark.parametrize("case", range(len(WG2_CASES)))
def test_qgemm_sign8_wgrad_direct(be, case):
    K.check_conv(be, seed=210 + case, wmode=1, sign8=True, algos=(3,), **WG2_CASES[case])
    K.check_conv(be, seed=215 + case, wmode=1, sign8=True, algos=(3,), in_shuffle=2, **WG2_CASES[case])


@pytest.mark.parametrize("case", [1, 2])
def test_qconv_bnsign_fused_pooled_gradient(be, case):
    K.check_qconv_bnsign(be, seed=190 + case, pooled=True, **K.QGEMM_PW_CASES[case])
    K.check_qconv_bnsign(be, seed=195 + case, pooled=True, training=False, in_shuffle=2 if case == 1 else 0, **K.QGEMM_PW_CASES[case])
    K.check_qconv_bnsign(be, seed=199, pooled=True, x_shape=(8, 256, 32, 32), w_shape=(256, 128, 1, 1), groups=2, in_shuffle=2)      # nin_gc L3


def test_sign_classifier(be):
    K.check_sign_classifier(be)
    K.check_sign_classifier(be, N=2, Cc=130, H=2, W=2, Oc=16, bias=False, seed=1)
    K.check_sign_classifier(be, N=16, Cc=1024, H=8, W=8, Oc=10, seed=2)                 # nin_gc L9
    K.check_sign_classifier(be, N=5, Cc=330, H=8, W=8, Oc=10, seed=3)


@pytest.mark.parametrize("case", range(len(K.DEPLOYED_CASES)))
def test_deployed_sign_block_identity_statistics(be, case):
    """sign(conv(a) + b) of the BN-folded deployed graph (wbwtab/bn_fuse/bn_fuse.py:36-55) on the fused code kernels, bit for bit."""
    K.check_deployed_sign_block(be, seed=500 + case, **K.DEPLOYED_CASES[case])


@pytest.mark.parametrize("in_kind,quant,bits", [(0, 1, 2), (0, 0, 4), (2, 1, 8), (2, 0, 8), (1, 1, 2), (1, 1, 8), (1, 0, 3)])
def test_qa_backward_masks_as_intervals_bit_exact(be, in_kind, quant, bits):
    K.check_qa_interval_masks(be, in_kind=in_kind, quant=quant, bits=bits, seed=10 * in_kind + quant)


def test_qg_pack_multi_images_bit_identical(be):
    K.check_qg_pack_multi(be)



def test_code_classifier(be):
    K.check_code_classifier(be)
    K.check_code_classifier(be, N=2, Cc=130, H=2, W=2, Oc=16, bits=3, bias=False, seed=1)
    K.check_code_classifier(be, N=16, Cc=1024, H=8, W=8, Oc=10, bits=2, seed=2)         # nin_gc L9 under DoReFa W2A2
    K.check_code_classifier(be, N=5, Cc=330, H=8, W=8, Oc=10, bits=4, seed=3)
    K.check_code_classifier(be, N=16, Cc=1024, H=8, W=8, Oc=10, bits=8, seed=4)         # nin_gc L9 under DoReFa W8A8


@pytest.mark.parametrize("training", [True, False])
def test_first_conv_qa_wgrad(be, training):
    K.check_first_conv_qa_wgrad(be, training=training)
    K.check_first_conv_qa_wgrad(be, x_shape=(2, 3, 16, 16), Oc=160, k=3, training=training, quant=0, bits=4, seed=1)
    K.check_first_conv_qa_wgrad(be, x_shape=(8, 3, 32, 32), Oc=256, k=5, training=training, quant=1, seed=2)


def test_first_conv_fused_block(be):
    K.check_first_conv_fused(be, act=1)
    K.check_first_conv_fused(be, act=2, bits=2, seed=1)
    K.check_first_conv_fused(be, x_shape=(2, 3, 16, 16), Oc=72, k=3, act=1, bias=False, seed=2)
    K.check_first_conv_fused(be, x_shape=(2, 3, 16, 16), Oc=40, k=3, act=2, bits=4, seed=3)
    K.check_first_conv_fused(be, x_shape=(32, 3, 32, 32), Oc=256, k=5, act=1, seed=4)          # nin_gc's first block
    K.check_first_conv_fused(be, x_shape=(32, 3, 32, 32), Oc=256, k=5, act=2, bits=8, seed=5)


def test_first_conv_block_edge_geometries(be):
    """two channel blocks (O > 256), more image tiles than Gram blocks (grid-stride tile loop), a single input channel (K = 9), batch 600"""
    K.check_first_conv_gram_bwd(be, x_shape=(2, 3, 8, 8), Oc=320, k=5, kind="bn", seed=11)
    K.check_first_conv_fused(be, x_shape=(2, 3, 8, 8), Oc=320, k=5, act=1, seed=12)
    K.check_first_conv_xgram(be, (520, 1, 4, 8), 3, seed=14)
    K.check_first_conv_fused(be, x_shape=(600, 3, 8, 8), Oc=40, k=3, act=2, bits=3, seed=15)
    K.check_first_conv_gram_bwd(be, x_shape=(600, 3, 8, 8), Oc=40, k=3, kind="bn", seed=17)
    K.check_first_conv_gram_bwd(be, x_shape=(2, 1, 8, 8), Oc=24, k=3, kind="bn", seed=16)


def test_first_conv_gram_backward(be):
    K.check_first_conv_gram_bwd(be, kind="bn")
    K.check_first_conv_gram_bwd(be, kind="qa", quant=1, bits=2, seed=1)
    K.check_first_conv_gram_bwd(be, x_shape=(2, 3, 16, 16), Oc=72, k=3, kind="bn", bias=False, seed=2)
    K.check_first_conv_gram_bwd(be, x_shape=(2, 3, 16, 16), Oc=40, k=3, kind="qa", quant=0, bits=4, seed=3)
    K.check_first_conv_gram_bwd(be, x_shape=(64, 3, 32, 32), Oc=256, k=5, kind="bn", seed=4)          # nin_gc's first block
    K.check_first_conv_gram_bwd(be, x_shape=(64, 3, 32, 32), Oc=256, k=5, kind="qa", quant=1, bits=8, seed=5)
    K.check_first_conv_gram_bwd(be, x_shape=(32, 3, 32, 32), Oc=64, k=3, kind="qa", quant=1, bits=4, seed=6)          # resnet's first block


@pytest.mark.parametrize("training", [True, False])
def test_first_conv_bn_wgrad(be, training):
    K.check_first_conv_bn_wgrad(be, training=training)
    K.check_first_conv_bn_wgrad(be, x_shape=(2, 3, 16, 16), Oc=160, k=3, training=training, seed=1)
    K.check_first_conv_bn_wgrad(be, x_shape=(8, 3, 32, 32), Oc=256, k=5, training=training, seed=2)


@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_qconv_bnsign_byte_stash(be, case):
    K.check_qconv_bnsign(be, seed=220 + case, stash=True, **K.QGEMM_PW_CASES[case])
    K.check_qconv_bnsign(be, seed=225 + case, stash=True, training=False, **K.QGEMM_PW_CASES[case])
    if case in (1, 2):
        K.check_qconv_bnsign(be, seed=230 + case, stash=True, pooled=True, **K.QGEMM_PW_CASES[case])
    if case == 0:
        K.check_qconv_bnsign(be, seed=240, stash=True, x_shape=(8, 256, 32, 32), w_shape=(256, 128, 1, 1), groups=2, in_shuffle=2)
        K.check_qconv_bnsign(be, seed=241, stash=True, pooled=True, x_shape=(8, 256, 32, 32), w_shape=(256, 128, 1, 1), groups=2, in_shuffle=2)


def test_conv_backward_with_bn_and_maxpool_folded_in(be):
    """mn_conv2d_bwd_data_bnh_pool / mn_conv2d_bwd_weight_bnh_pool: the conv's backward forms dy from (pooled gradient, the block's own sign codes, h) -- the pool's
    first-maximum routing and the BatchNorm+sign backward in the operand load -- against the two-step path through mn_bnh_bwd_apply's full-size dy."""
    before = getattr(K.check_qconv_bnsign, "pool_fold_checked", 0)
    for i, case in enumerate(K.WGRAD_SPEC_CASES):
        K.check_qconv_bnsign(be, seed=320 + i, stash=True, pooled=True, **case)
        K.check_qconv_bnsign(be, seed=330 + i, stash=True, pooled=True, training=False, **case)
    K.check_qconv_bnsign(be, seed=340, stash=True, pooled=True, x_shape=(3, 256, 32, 32), w_shape=(256, 128, 1, 1), groups=2, in_shuffle=2)       # nin_gc L3
    K.check_qconv_bnsign(be, seed=341, stash=True, pooled=True, x_shape=(4, 512, 16, 16), w_shape=(512, 128, 1, 1), groups=4, in_shuffle=4)      # nin_gc L6
    assert getattr(K.check_qconv_bnsign, "pool_fold_checked", 0) - before == 8


def test_conv_backward_with_bn_folded_in(be):
    K.check_qconv_bnsign(be, seed=250, stash=True, x_shape=(2, 128, 4, 8), w_shape=(128, 64, 1, 1), groups=2)
    K.check_qconv_bnsign(be, seed=251, stash=True, x_shape=(3, 256, 4, 8), w_shape=(256, 128, 1, 1), groups=2, in_shuffle=2, bias=False)
    K.check_qconv_bnsign(be, seed=252, stash=True, training=False, x_shape=(2, 80, 4, 4), w_shape=(100, 40, 1, 1), groups=2)
    K.check_qconv_bnsign(be, seed=253, stash=True, x_shape=(8, 512, 16, 16), w_shape=(512, 128, 1, 1), groups=4, in_shuffle=16)


def test_pointwise_block_backward_in_one_kernel(be):
    """mn_conv2d_bwd_bnh (k_pwb): the emulated run's cases + nin_gc's pointwise layers L2 / L5 / L8 (unpooled) and L3 / L6 (pooled) at batch 8."""
    K.check_pwb(be)
    before = getattr(K._check_pwb, "count", 0)
    K.check_qconv_bnsign(be, seed=430, stash=True, x_shape=(8, 256, 32, 32), w_shape=(256, 128, 1, 1), groups=2)                                   # L2
    K.check_qconv_bnsign(be, seed=431, stash=True, pooled=True, x_shape=(8, 256, 32, 32), w_shape=(256, 128, 1, 1), groups=2, in_shuffle=2)       # L3
    K.check_qconv_bnsign(be, seed=432, stash=True, x_shape=(8, 512, 16, 16), w_shape=(512, 128, 1, 1), groups=4, in_shuffle=16)                   # L5
    K.check_qconv_bnsign(be, seed=433, stash=True, pooled=True, x_shape=(8, 512, 16, 16), w_shape=(512, 128, 1, 1), groups=4, in_shuffle=4)       # L6
    K.check_qconv_bnsign(be, seed=434, stash=True, x_shape=(8, 1024, 8, 8), w_shape=(1024, 128, 1, 1), groups=8, in_shuffle=32)                   # L8
    assert getattr(K._check_pwb, "count", 0) - before == 5


def test_kbit_block_backward_in_one_kernel(be):
    """mn_conv2d_bwd_codes / mn_conv2d_bwd_qa: the emulated run's cases + nin_gc's DoReFa layers L2 (W2A2, 16-bit stash), L5 (W8A8, 32-bit stash) and L3 (pooled) at batch 8."""
    K.check_pwb_bnq(be)
    before = getattr(K.check_qconv_bnq, "pwb_checked", 0)
    K.check_qconv_bnq(be, seed=450, x_shape=(8, 256, 32, 32), w_shape=(256, 128, 1, 1), groups=2)
    K.check_qconv_bnq(be, seed=451, x_shape=(8, 512, 16, 16), w_shape=(512, 128, 1, 1), groups=4, in_shuffle=16, a_bits=8, w_bits=8)
    K.check_qconv_bnq(be, seed=452, x_shape=(8, 256, 32, 32), w_shape=(256, 128, 1, 1), groups=2, in_shuffle=2, pooled=True)
    assert getattr(K.check_qconv_bnq, "pwb_checked", 0) - before == 3


# stashed block around a 3 x 3 convolution: h written by the k x k kernel, statistics / sign streamed from h (k_h_stats, k_h_sign)
KXK_STASH_CASES = [
    dict(x_shape=(3, 32, 8, 8), w_shape=(64, 16, 3, 3), padding=1, groups=2),                    # the nin_gc L7 pattern
    dict(x_shape=(2, 32, 16, 16), w_shape=(64, 16, 3, 3), padding=1, groups=2, in_shuffle=2, bias=False),
    dict(x_shape=(2, 6, 16, 16), w_shape=(40, 6, 3, 3), padding=1),
]


@pytest.mark.parametrize("case", range(len(KXK_STASH_CASES)))
@pytest.mark.parametrize("training", [True, False])
def test_qconv_kxk_bnsign_stash(be, case, training):
    K.check_qconv_bnsign(be, seed=260 + case, stash=True, training=training, **KXK_STASH_CASES[case])


def test_sign_pass_with_the_statistics_finals_folded_in(be):
    """k_h_sign_prep (the default since round 5): k_pws_stats_prep's work inside the streaming sign pass."""
    K.check_hsign_fold(be, KXK_STASH_CASES, full=True)


def test_ternary_weight_quantizer_multi(be):
    K.check_ternary_multi(be)
    K.check_binary_multi(be)


@pytest.mark.parametrize("training", [True, False])
def test_bnrelu(be, training):
    K.check_bnrelu(be, training=training)
    K.check_bnrelu(be, shape=(3, 7, 2, 2), seed=3, training=training)
    K.check_bnrelu(be, shape=(32, 64, 16, 16), seed=4, training=training)


def test_pool_f32(be):
    K.check_pool_f32(be)
    K.check_pool_f32(be, shape=(2, 3, 2, 8), seed=1)


# ---- k-bit (DoReFa) fused block on activation codes (16-bit stash): the small cases of the emulated run + nin_gc's layers L2..L8
@pytest.mark.parametrize("case", range(len(K.BNQ_CASES)))
def test_qconv_bnq_block(be, case):
    K.check_qconv_bnq(be, seed=300 + case, **K.BNQ_CASES[case])


BNQ_HOT = [
    ("nin_gc L2", dict(x_shape=(3, 256, 32, 32), w_shape=(256, 128, 1, 1), groups=2)),
    ("nin_gc L3 + pool", dict(x_shape=(3, 256, 32, 32), w_shape=(256, 128, 1, 1), groups=2, in_shuffle=2, pooled=True)),
    ("nin_gc L4 3x3", dict(x_shape=(4, 256, 16, 16), w_shape=(512, 16, 3, 3), groups=16, padding=1, in_shuffle=2)),
    ("nin_gc L5", dict(x_shape=(4, 512, 16, 16), w_shape=(512, 128, 1, 1), groups=4, in_shuffle=16)),
    ("nin_gc L6 + pool", dict(x_shape=(4, 512, 16, 16), w_shape=(512, 128, 1, 1), groups=4, in_shuffle=4, pooled=True)),
    ("nin_gc L7 3x3", dict(x_shape=(6, 512, 8, 8), w_shape=(1024, 16, 3, 3), groups=32, padding=1, in_shuffle=4)),
    ("nin_gc L8 (fp32 consumer)", dict(x_shape=(6, 1024, 8, 8), w_shape=(1024, 128, 1, 1), groups=8, in_shuffle=32, quant=0)),
]


@pytest.mark.parametrize("name,kw", BNQ_HOT, ids=[n for n, _ in BNQ_HOT])
@pytest.mark.parametrize("training", [True, False])
def test_qconv_bnq_hot_shapes(be, name, kw, training):
    K.check_qconv_bnq(be, seed=77, training=training, **kw)


@pytest.mark.parametrize("name,kw", BNQ_HOT, ids=[n for n, _ in BNQ_HOT])
def test_qconv_bnq_hot_shapes_w8a8(be, name, kw):
    """The same layers at W8A8 (the reference's CPU configuration, wqaq/dorefa/main.py:135,189-190): the wide variants of the grouped kernels -- bf16 codes up to
    255, |acc| < 2^24 exact in the fp32 accumulators, 32-bit stash, pooled and unpooled streaming passes on it."""
    K.check_qconv_bnq(be, seed=78, training=True, a_bits=8, w_bits=8, **kw)


def test_dorefa_weight_quantizer_multi_bit_identical(be):
    """mn_dorefa_w_fwd_multi / _bwd_multi over several tensors == the per-tensor entry points, bit for bit (same partial blocks, same reduction order)."""
    import ctypes as C
    torch = be.torch
    gen = torch.Generator(device="cuda").manual_seed(9)
    shapes = [(256, 128, 1, 1), (512, 16, 3, 3), (10, 1024, 1, 1), (7,), (1024, 128, 1, 1)]
    ws = [torch.randn(s, device="cuda", generator=gen) * 0.7 for s in shapes]
    ws[3][2] = ws[3][5] = ws[3].abs().max() + 1.0                      # a tie at the maximum
    gs = [torch.randn(s, device="cuda", generator=gen) for s in shapes]
    lib = be.lib
    for bits in (2, 8):
        single_q, single_d = [], []
        for w, g in zip(ws, gs):
            sc = torch.empty(int(lib.mn_dorefa_w_ws_floats(w.numel())), device="cuda")
            q, d = torch.empty_like(w), torch.empty_like(w)
            be.call("mn_dorefa_w_fwd", be.ptr(w), be.ptr(q), w.numel(), bits, be.ptr(sc), be.stream)
            be.call("mn_dorefa_w_bwd", be.ptr(g), be.ptr(w), be.ptr(d), w.numel(), bits, be.ptr(sc), be.stream)
            single_q.append(q); single_d.append(d)
        n = len(ws)
        PA, LA = C.c_void_p * n, C.c_int64 * n
        qs, ds = [torch.empty_like(w) for w in ws], [torch.empty_like(w) for w in ws]
        scs = [torch.empty(int(lib.mn_dorefa_w_ws_floats(w.numel())), device="cuda") for w in ws]
        be.call("mn_dorefa_w_fwd_multi", PA(*[w.data_ptr() for w in ws]), PA(*[q.data_ptr() for q in qs]), PA(*[t.data_ptr() for t in scs]), LA(*[w.numel() for w in ws]), n, bits, be.stream)
        be.call("mn_dorefa_w_bwd_multi", PA(*[g.data_ptr() for g in gs]), PA(*[w.data_ptr() for w in ws]), PA(*[d.data_ptr() for d in ds]), PA(*[t.data_ptr() for t in scs]),
                LA(*[w.numel() for w in ws]), n, bits, be.stream)
        torch.cuda.synchronize()
        for a, b in zip(single_q + single_d, qs + ds):
            assert torch.equal(a, b)
        # the variant the modules use: the forward keeps tanh(w) and its scratch, the backward reuses them (two launches instead of four)
        qc, dc = [torch.empty_like(w) for w in ws], [torch.empty_like(w) for w in ws]
        ths = [torch.empty_like(w) for w in ws]
        sc2 = [torch.empty(int(lib.mn_dorefa_w_ws_floats(w.numel())), device="cuda") for w in ws]
        be.call("mn_dorefa_w_fwd_multi_cached", PA(*[w.data_ptr() for w in ws]), PA(*[q.data_ptr() for q in qc]), PA(*[t.data_ptr() for t in sc2]),
                PA(*[t.data_ptr() for t in ths]), LA(*[w.numel() for w in ws]), n, bits, be.stream)
        be.call("mn_dorefa_w_bwd_multi_cached", PA(*[g.data_ptr() for g in gs]), PA(*[w.data_ptr() for w in ws]), PA(*[d.data_ptr() for d in dc]),
                PA(*[t.data_ptr() for t in sc2]), PA(*[t.data_ptr() for t in ths]), LA(*[w.numel() for w in ws]), n, bits, be.stream)
        torch.cuda.synchronize()
        for a, b in zip(single_q + single_d, qc + dc):
            assert torch.equal(a, b)


def test_dorefa_tanh_pinned(be):
    """The one documented deviation of the DoReFa weight quantizer: tanh.  The reference evaluates torch.tanh on the CPU, which in a MKL build of torch is Intel
    MKL VML vsTanh (HA mode) -- a closed-source kernel whose last bit even depends on the host CPU (MKL dispatches per micro-architecture): it cannot be restated.
    The kernels therefore evaluate the CORRECTLY ROUNDED fp32 tanh (fp64 evaluation, one rounding).  Pinned three ways:
    (1) mn_tanh_f32 == float32(tanh(float64(x))) bit for bit on 2^20 seeded inputs and on the committed inputs of tests/golden/tanh_device_vs_cpu.json (inputs where
        the reference host's torch.tanh and the kernels differed when the fixture was harvested by scripts/make_tanh_fixture.py), whose recorded kernel bits must
        still come out -- machine independent;
    (2) torch-CPU tanh of THIS host never differs from it by more than one ulp, and in less than 2 % of the inputs;
    (3) weight codes differ from the torch-CPU evaluation of the reference formula (wqaq/dorefa/quantize.py:61-73) only where the two tanh differ across a rounding
        boundary (or next to one through the shared maximum): at most 2 per 10^5, by one step."""
    import json
    import os
    torch = be.torch
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "tanh_device_vs_cpu.json")))
    x = np.array(fx["x_bits"], dtype=np.int32).view(np.float32)
    y = torch.empty(x.size, device="cuda")
    be.call("mn_tanh_f32", be.ptr(torch.from_numpy(x).cuda()), be.ptr(y), x.size, be.stream)
    assert np.array_equal(y.cpu().numpy().view(np.int32), np.array(fx["dev_bits"], dtype=np.int32))
    assert np.array_equal(np.tanh(x.astype(np.float64)).astype(np.float32).view(np.int32), np.array(fx["dev_bits"], dtype=np.int32))
    assert np.all(np.abs(np.array(fx["dev_bits"], dtype=np.int64) - np.array(fx["cpu_bits"], dtype=np.int64)) == 1)
    rng = np.random.default_rng(7)
    w = (rng.standard_normal(1 << 20) * 0.4).astype(np.float32)
    wt = torch.from_numpy(w)
    yd = torch.empty(w.size, device="cuda")
    be.call("mn_tanh_f32", be.ptr(wt.cuda()), be.ptr(yd), w.size, be.stream)
    dev = yd.cpu().numpy().view(np.int32)
    assert np.array_equal(dev, np.tanh(w.astype(np.float64)).astype(np.float32).view(np.int32)), "the kernels' tanh is not the correctly rounded value"
    d = dev.astype(np.int64) - torch.tanh(wt).numpy().view(np.int32).astype(np.int64)
    assert np.abs(d).max() <= 1 and np.count_nonzero(d) <= 0.02 * w.size, (int(np.abs(d).max()), int(np.count_nonzero(d)))
    for bits in (2, 4, 8):
        n = float(2 ** bits - 1)
        t = torch.tanh(wt)
        u = t / 2 / t.abs().max() + 0.5
        s_ = 1.0 / n
        v = u / s_
        k_ref = (torch.sign(v) * torch.floor(v.abs() + 0.5)).numpy()
        q = torch.empty(w.size, device="cuda")
        ws = torch.empty(int(be.lib.mn_dorefa_w_ws_floats(w.size)), device="cuda")
        be.call("mn_dorefa_w_fwd", be.ptr(wt.cuda()), be.ptr(q), w.size, bits, be.ptr(ws), be.stream)
        k_dev = np.round((q.cpu().numpy() + 1) / 2 * n)
        flips = np.nonzero(k_dev != k_ref)[0]
        assert flips.size <= 2 * (w.size // 100000 + 1), (bits, flips.size)
        assert np.all(np.abs(k_dev[flips] - k_ref[flips]) == 1)


def test_pointwise_wgrad_specialised_edge_tiles(be):
    K.check_wgrad_spec(be)


def test_iao_bnfold(be):
    K.check_iao_bnfold(be)
    K.check_iao_bnfold(be, O_=70, K_=288, bias=False, shared_var=False, seed=1)
    K.check_iao_bnfold(be, O_=5, K_=1300, seed=2)


@pytest.mark.parametrize("bits", [2, 3, 4, 7])
def test_qa_activation_code_bit_exact_at_boundaries(be, bits):
    K.check_qa_code_exact(be, bits=bits)


@pytest.mark.parametrize("bits,pool", [(2, False), (2, True), (3, False), (3, True), (4, False)])
def test_qa_forward_integer_thresholds(be, bits, pool):
    K.check_qa_thresholds(be, bits=bits, pool=pool, seed=bits)


# ---- dense layers on activation codes (the ResNet family): qgemm_dense.hip
@pytest.mark.parametrize("case", range(len(K.QDENSE_CASES)))
def test_qdense_layer(be, case):
    xs, Oc, k, s = K.QDENSE_CASES[case]
    K.check_qdense(be, xs, Oc, k, s, seed=300 + case)


@pytest.mark.parametrize("case", [1, 4, 6])
def test_qdense_layer_prepacked_weights(be, case):
    xs, Oc, k, s = K.QDENSE_CASES[case]
    K.check_qdense(be, xs, Oc, k, s, seed=380 + case, prepack=True)


def test_qdense_layer_bf16_forward(be):
    """8-bit DoReFa weight codes (+-255) do not fit signed bytes: the bf16 forward."""
    K.check_qdense(be, (2, 64, 8, 8), 64, 3, 1, a_bits=2, w_bits=8, seed=390)
    K.check_qdense(be, (3, 128, 8, 8), 64, 3, 2, a_bits=3, w_bits=8, seed=391, prepack=True)


def test_qdense_layer_wide_codes(be):
    K.check_qdense(be, (2, 64, 8, 8), 64, 3, 1, a_bits=4, w_bits=4, seed=320)
    K.check_qdense(be, (3, 128, 16, 16), 128, 3, 2, a_bits=4, w_bits=4, seed=321)     # 32-bit stash at stride 2


# every layer shape of the reference's resnet18 on CIFAR (models/resnet.py:69-112) at a batch that leaves partial tiles
@pytest.mark.parametrize("xs,Oc,k,s", [
    ((37, 64, 32, 32), 64, 3, 1), ((37, 64, 32, 32), 128, 3, 2), ((37, 64, 32, 32), 128, 1, 2), ((37, 128, 16, 16), 128, 3, 1), ((37, 128, 16, 16), 256, 3, 2),
    ((37, 128, 16, 16), 256, 1, 2), ((37, 256, 8, 8), 256, 3, 1), ((37, 256, 8, 8), 512, 3, 2), ((37, 256, 8, 8), 512, 1, 2), ((37, 512, 4, 4), 512, 3, 1)])
def test_qdense_resnet18_shapes(be, xs, Oc, k, s):
    K.check_qdense(be, xs, Oc, k, s, seed=hash((xs, Oc, k, s)) % 1000)


def test_qdense_hot_shapes_with_exact_three_term_gradient():
    """MN_GRAD_TERMS=3: the exact three-term bf16 split of the fp32 gradient in the dense backward kernels (the default since round 5 is the two-term split), on the
    resnet18 shapes that dominate the c4 / c5 steps -- a child process, because the library reads its knobs once."""
    body = ("assert be.lib.mn_dense_grad_terms() == 3; "
            "K.check_qdense(be, (37, 64, 32, 32), 64, 3, 1, seed=701); K.check_qdense(be, (37, 128, 16, 16), 128, 3, 1, seed=702); "
            "K.check_qdense(be, (37, 64, 32, 32), 128, 3, 2, seed=703); K.check_qdense(be, (37, 64, 32, 32), 128, 1, 2, seed=704); "
            "K.check_qdense(be, (37, 256, 8, 8), 256, 3, 1, seed=705, prepack=True)")
    K.run_child(body, "gpu", {"MN_GRAD_TERMS": "3"}, 900)


@pytest.mark.parametrize("in_kind,res_kind", [(0, 1), (0, 2), (2, 3), (1, 1), (0, 0), (2, 1), (0, 3), (2, 2)])
@pytest.mark.parametrize("training", [True, False])
def test_residual_block_end(be, in_kind, res_kind, training):
    K.check_qr(be, in_kind=in_kind, res_kind=res_kind, training=training, with_dq2=(res_kind != 1), with_gf=(res_kind != 0), seed=330 + in_kind * 4 + res_kind)
    K.check_qr(be, shape=(33, 64, 16, 16), in_kind=in_kind, res_kind=res_kind, bits=4, training=training, with_dq2=True, with_gf=False, seed=350)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_qlinear(be, mode):
    K.check_qlinear(be, mode=mode, seed=360 + mode)
    K.check_qlinear(be, N=256, Cc=512, Oc=10, mode=mode, bits=4, bias=False, seed=365 + mode)
    K.check_qlinear(be, N=3, Cc=64, Oc=64, mode=mode, seed=370 + mode)


@pytest.mark.parametrize("case", range(len(K.QDENSE_CASES)))
def test_qdense_layer_iao(be, case):
    xs, Oc, k, s = K.QDENSE_CASES[case]
    K.check_qdense_iao(be, xs, Oc, k, s, seed=400 + case)


@pytest.mark.parametrize("shape,bits,q_type,relu,scbn", [((3, 6, 4, 8), 4, 0, True, False), ((2, 5, 8, 8), 8, 1, True, True), ((3, 4, 2, 4), 4, 0, False, True),
                                                         ((9, 3, 4, 4), 6, 0, False, False), ((64, 128, 16, 16), 4, 0, True, False), ((64, 128, 16, 16), 4, 0, True, True)])
def test_iao_qadd_bn_fused(be, shape, bits, q_type, relu, scbn):
    K.check_iao_qadd_bn(be, shape, bits, q_type, relu, scbn, seed=430 + bits)


def test_tail_bn_relu_pool_and_loss(be):
    K.check_tail(be, (37, 10, 8, 8), seed=450)
    K.check_tail(be, (300, 3, 2, 4), seed=452)
    K.check_tail(be, (256, 10, 8, 8), seed=451)


def test_qdense_layer_iao_w8a8_bias(be):
    K.check_qdense_iao(be, (2, 64, 8, 8), 64, 3, 1, a_bits=8, w_bits=8, bias=True, seed=410)
    K.check_qdense_iao(be, (37, 256, 8, 8), 512, 3, 2, seed=411)
    K.check_qdense_iao(be, (37, 128, 16, 16), 256, 1, 2, seed=412)


@pytest.mark.parametrize("bits,q_type,obs_kind,first,update", [(8, 0, 1, (True, True), True), (4, 0, 1, (False, False), True), (8, 1, 1, (False, True), True),
                                                              (8, 0, 0, (False, False), True), (4, 0, 1, (False, False), False)])
def test_iao_quant_add_fused(be, bits, q_type, obs_kind, first, update):
    K.check_iao_qadd(be, bits=bits, q_type=q_type, obs_kind=obs_kind, first=first, update=update, seed=bits + q_type)
    K.check_iao_qadd(be, bits=bits, q_type=q_type, obs_kind=obs_kind, first=first, update=update, seed=bits + q_type + 7, relu=True)
    K.check_iao_qadd(be, n=256 * 64 * 32 * 32, bits=bits, q_type=q_type, obs_kind=obs_kind, first=first, update=update, seed=1)


@pytest.mark.parametrize("bits,q_type,obs_kind", [(4, 0, 0), (8, 0, 1), (8, 1, 0)])
def test_iao_weight_quantizers_multi(be, bits, q_type, obs_kind):
    K.check_iao_w_multi(be, bits=bits, q_type=q_type, obs_kind=obs_kind, seed=bits)


@pytest.mark.parametrize("training", [True, False])
def test_bn2d_plain(be, training):
    K.check_bnrelu(be, training=training, plain=True)
    K.check_bnrelu(be, shape=(33, 64, 16, 16), seed=3, training=training, plain=True)


def test_global_avgpool(be):
    K.check_gap(be)
    K.check_gap(be, planes=2560, HW=64, seed=1)


@pytest.mark.parametrize("w_bits,iao", [(2, False), (8, False), (4, True)])
def test_qd_pack_multi_tables(be, w_bits, iao):
    K.check_qd_pack_multi(be, w_bits=w_bits, iao=iao, seed=w_bits)


# ---- the BN-fused IAO block without the statistics convolution (iao_bnfuse.hip)
@pytest.mark.parametrize("case", range(5))
def test_iaobf_pointwise(be, case):
    import iaobf_cases as B
    B.check_iaobf_pointwise(be, B.CASES[case], seed=case)


def test_iaobf_pointwise_nin_gc_layer(be):
    """the 256 -> 256, groups 2, shuffle 2 layer of nin_gc (models/nin_gc.py:78-87) at 16 images of 32 x 32: full-size tiles, several slabs per block, all four
    waves of the backward-data kernel busy"""
    import iaobf_cases as B
    B.check_iaobf_pointwise(be, dict(N=16, C=256, O=256, H=32, W=32, groups=2, shuffle=2, bias=True), seed=11, nsteps=1)


@pytest.mark.parametrize("bits,q_type,relu_mask", [(8, 0, False), (4, 1, False), (8, 0, True)])
def test_iao_fq_maxpool(be, bits, q_type, relu_mask):
    import iaobf_cases as B
    B.check_fq_maxpool(be, bits=bits, q_type=q_type, relu_mask=relu_mask, seed=bits + q_type)
    B.check_fq_maxpool(be, shape=(8, 64, 32, 32), bits=bits, q_type=q_type, relu_mask=relu_mask, seed=100 + bits)


def test_bnfuse_stream_helpers(be):
    import iaobf_cases as B
    B.check_stream_helpers(be)
    B.check_stream_helpers(be, n=4 * 3000017, seed=3)


def test_iao_codes_at_rounding_boundaries(be):
    import iaobf_cases as B
    B.check_iao_codes_at_boundaries(be)
    B.check_iao_codes_at_boundaries(be, seed=5)


@pytest.mark.parametrize("k,Cin,W", [(5, 3, 12), (3, 7, 8), (5, 5, 4)])
def test_iaobf_gram_of_first_layer_patches(be, k, Cin, W):
    import iaobf_cases as B
    B.check_gram_patch(be, Cin=Cin, W=W, k=k, seed=k + Cin)
    B.check_gram_patch(be, N=64, Cin=3, H=32, W=32, k=5, seed=1)


@pytest.mark.parametrize("ci", range(3))
def test_iaobf_grouped_3x3_family(be, ci):
    import iaobf_cases as B
    B.check_g3(be, B.G3_CASES[ci], seed=ci)


def test_iaobf_grouped_3x3_family_at_nin_gc_shapes(be):
    """the two layers of nin_gc at a batch that fills the persistent grids (several tiles per block, every group)"""
    import iaobf_cases as B
    B.check_g3(be, dict(N=64, G=16, HW=16, shuffle=2, bias=True, blocks=0), seed=7)
    B.check_g3(be, dict(N=64, G=32, HW=8, shuffle=4, bias=True, blocks=0), seed=8)


def test_first_layer_forward_with_relu_and_minmax_epilogue(be):
    import iaobf_cases as B
    B.check_first_layer_act(be)
    B.check_first_layer_act(be, N=64, Cin=3, H=32, W=32, O=256, k=5, seed=3)


@pytest.mark.parametrize("shuffle,bias", [(0, True), (2, False)])
def test_iaobf_thin_output_family(be, shuffle, bias):
    import iaobf_cases as B
    B.check_thin(be, shuffle=shuffle, bias=bias, seed=shuffle)
    B.check_thin(be, N=64, Cc=1024, O=10, HW=(8, 8), shuffle=shuffle, bias=bias, seed=5 + shuffle)


def test_iaobf_gram_statistics_never_negative_variance(be):
    import iaobf_cases as B
    B.check_gram_stats_variance_clamp(be)


def test_first_conv_gram_statistics_on_unnormalised_images_and_difference_filters(be):
    K.check_first_conv_gram_conditioning(be)
    K.check_first_conv_gram_conditioning(be, x_shape=(8, 3, 32, 32), Oc=24, k=5, seed=3)
    K.check_first_conv_gram_conditioning(be, x_shape=(1024, 3, 32, 32), Oc=32, k=5, seed=4)          # 2048 pixels per block in fp32: 1e-3 ... 1e-1 without the shift
