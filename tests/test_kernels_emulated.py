"""The HIP kernel sources, compiled for the CPU SIMT emulator (tests/emu), driven through the real C ABI and checked
against the oracle + golden fixtures.  This validates index math / tiling / MFMA fragment maps without a GPU; the same
checks run on the MI355X in tests/test_gpu_kernels.py."""
import numpy as np
import pytest

import abi_driver
import kernel_cases as K


@pytest.fixture(scope="module")
def be():
    return abi_driver.Backend("emu")


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
    K.check_bn_stats(be, shape=(3, 5, 3, 3), seed=1)   # HW % 4 != 0 path


@pytest.mark.parametrize("case", range(len(K.SMALL_CONV_CASES)))
def test_conv_plain(be, case):
    K.check_conv(be, seed=case, **K.SMALL_CONV_CASES[case])


@pytest.mark.parametrize("case", [1, 2, 3, 5])
def test_conv_dorefa_fused(be, case):
    K.check_conv(be, seed=10 + case, mode=1, bits=3, **K.SMALL_CONV_CASES[case])


@pytest.mark.parametrize("case,q_type", [(1, 0), (2, 1), (3, 0), (8, 1)])
def test_conv_iao_fused(be, case, q_type):
    K.check_conv(be, seed=20 + case, mode=2, bits=4, q_type=q_type, **K.SMALL_CONV_CASES[case])


# ---- code-domain (bf16 MFMA) kernels: algo 3
@pytest.mark.parametrize("case", range(len(K.QGEMM_PW_CASES)))
@pytest.mark.parametrize("wmode", [1, 2, 3])
def test_qgemm_pointwise_binary_x(be, case, wmode):
    """wbwtab-style: +-1 activations (no fused quantizer), coded weights; fwd / bwd-data / bwd-weight on algo 3."""
    K.check_conv(be, seed=40 + case, wmode=wmode, wbits=4, binary_x=True, algos=(3,), expect_qgemm=True, **K.QGEMM_PW_CASES[case])


@pytest.mark.parametrize("case", [0, 1])
def test_qgemm_pointwise_real_x(be, case):
    """real-valued activations: the three-term split (zero terms skipped) keeps all three passes exact."""
    K.check_conv(be, seed=50 + case, wmode=1, algos=(3,), expect_qgemm=True, **K.QGEMM_PW_CASES[case])


@pytest.mark.parametrize("case,mode,q_type", [(1, 1, 0), (2, 1, 0), (1, 2, 0), (3, 2, 0)])
def test_qgemm_pointwise_fused_actq(be, case, mode, q_type):
    """DoReFa / IAO activation quantizer fused: codes in the prologue, scale in the epilogue, clip-STE in bwd-data."""
    K.check_conv(be, seed=60 + case, mode=mode, bits=4, q_type=q_type, wmode=2 if mode == 1 else 3, wbits=4, algos=(3,),
                 expect_qgemm=True, **K.QGEMM_PW_CASES[case])


KXK_SUP = [True] * 5 + [(True, False, False)]   # per case: (fwd, bwd_data, bwd_weight); stride 2: forward only


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


def test_adam_step(be):
    K.check_adam(be)
    K.check_adam(be, sizes=tuple(range(1, 41)), steps=2, seed=1)      # more tensors than one launch table holds


@pytest.mark.parametrize("training", [True, False])
def test_bnsign(be, training):
    K.check_bnsign(be, training=training)
    K.check_bnsign(be, shape=(3, 7, 2, 2), seed=3, training=training)


@pytest.mark.parametrize("case,sg", [(1, 2), (1, 5), (2, 4)])
def test_qgemm_pointwise_in_shuffle(be, case, sg):
    """channel shuffle folded into the conv's addressing: fwd reads, bwd-data writes and bwd-weight reads through the map."""
    K.check_conv(be, seed=90 + case, wmode=1, binary_x=True, algos=(3,), expect_qgemm=True, in_shuffle=sg, **K.QGEMM_PW_CASES[case])
    K.check_conv(be, seed=95 + case, mode=1, bits=4, wmode=2, wbits=4, algos=(3,), expect_qgemm=True, in_shuffle=sg, **K.QGEMM_PW_CASES[case])


@pytest.mark.parametrize("case,sg", [(0, 2), (0, 8), (4, 5)])
def test_qgemm_kxk_in_shuffle(be, case, sg):
    K.check_conv(be, seed=97 + case, wmode=1, binary_x=True, algos=(3,), expect_qgemm=True, in_shuffle=sg, **K.QGEMM_KXK_CASES[case])


# ---- packed (int8) sign activations: MN_ACTQ_SIGN8 input of the code-domain kernels, int8 BN-sign output, sign max-pool
@pytest.mark.parametrize("case", range(len(K.QGEMM_PW_CASES)))
def test_qgemm_pointwise_sign8(be, case):
    K.check_conv(be, seed=140 + case, wmode=1, sign8=True, algos=(3,), expect_qgemm=True, **K.QGEMM_PW_CASES[case])


@pytest.mark.parametrize("case", [0, 1, 4])
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


def test_pool_sign8(be):
    K.check_pool_sign8(be)
    K.check_pool_sign8(be, shape=(2, 3, 2, 8), seed=1)


@pytest.mark.parametrize("case", range(len(K.QGEMM_PW_CASES)))
@pytest.mark.parametrize("training", [True, False])
def test_qconv_bnsign_fused(be, case, training):
    K.check_qconv_bnsign(be, seed=170 + case, training=training, **K.QGEMM_PW_CASES[case])


def test_qconv_bnsign_fused_shuffle_and_wide(be):
    K.check_qconv_bnsign(be, seed=180, in_shuffle=2, **K.QGEMM_PW_CASES[1])
    K.check_qconv_bnsign(be, seed=181, x_shape=(2, 256, 4, 8), w_shape=(256, 128, 1, 1), groups=2, in_shuffle=2)   # KS = 4, two m-blocks: the nin_gc pattern


# ---- first-layer kernels (real fp32 operands, K = Cin*KH*KW <= 76): algo 0 (auto) must pick them and match fp64
FIRST_CASES = [
    dict(x_shape=(3, 3, 8, 8), w_shape=(24, 3, 5, 5), padding=2),                 # nin_gc L1 pattern, MT = 1, masked channels
    dict(x_shape=(2, 3, 16, 16), w_shape=(160, 3, 3, 3), padding=1),              # resnet conv1 pattern, MT = 4, two strips
    dict(x_shape=(2, 1, 8, 16), w_shape=(70, 1, 3, 3), padding=1, bias=False),    # one input channel, MT = 2
]


@pytest.mark.parametrize("case", range(len(FIRST_CASES)))
def test_conv_first_layer(be, case):
    import ctypes as C
    kw = FIRST_CASES[case]
    g = be.geom(kw["x_shape"], kw["w_shape"], padding=kw["padding"])
    assert be.lib.mn_conv2d_first_supported(C.byref(g), 0) == 1 and be.lib.mn_conv2d_first_supported(C.byref(g), 2) == 1
    K.check_conv(be, seed=200 + case, algos=(0,), rel=2e-6, **kw)


WG2_CASES = [
    dict(x_shape=(2, 128, 4, 8), w_shape=(128, 64, 1, 1), groups=2),                 # 64 x 64 tiles (MW = 2), two steps
    dict(x_shape=(3, 256, 4, 8), w_shape=(256, 128, 1, 1), groups=2, bias=False),    # 128 x 128 tiles (MW = 4): the nin_gc pattern
    dict(x_shape=(2, 80, 4, 4), w_shape=(100, 40, 1, 1), groups=2),                  # clamped rows / columns (Mg = 50, Cg = 40), 16-pixel images
]


@pytest.mark.parametrize("case", range(len(WG2_CASES)))
def test_qgemm_sign8_wgrad_direct(be, case):
    """k_pws_wgrad (LDS-free backward-weight on sign codes)."""
    K.check_conv(be, seed=210 + case, wmode=1, sign8=True, algos=(3,), **WG2_CASES[case])
    K.check_conv(be, seed=215 + case, wmode=1, sign8=True, algos=(3,), in_shuffle=2, **WG2_CASES[case])


@pytest.mark.parametrize("case", [1, 2])
def test_qconv_bnsign_fused_pooled_gradient(be, case):
    K.check_qconv_bnsign(be, seed=190 + case, pooled=True, **K.QGEMM_PW_CASES[case])
    K.check_qconv_bnsign(be, seed=195 + case, pooled=True, training=False, in_shuffle=2 if case == 1 else 0, **K.QGEMM_PW_CASES[case])


def test_sign_classifier(be):
    K.check_sign_classifier(be)
    K.check_sign_classifier(be, N=2, Cc=130, H=2, W=2, Oc=16, bias=False, seed=1)      # partial pixel chunk, C not a multiple of 64
    K.check_sign_classifier(be, N=5, Cc=330, H=8, W=8, Oc=10, seed=3)                  # two blocks (320 pixels), 21 channels per wave: unrolled loads + tail


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
    K.check_code_classifier(be, N=5, Cc=330, H=8, W=8, Oc=10, bits=4, seed=3)
    K.check_code_classifier(be, N=3, Cc=200, H=4, W=8, Oc=10, bits=8, seed=4)          # 8-bit codes (W8A8)


@pytest.mark.parametrize("training", [True, False])
def test_first_conv_qa_wgrad(be, training):
    K.check_first_conv_qa_wgrad(be, training=training)
    K.check_first_conv_qa_wgrad(be, x_shape=(2, 3, 16, 16), Oc=160, k=3, training=training, quant=0, bits=4, seed=1)
    if training:          # 10 partial rows per channel: the batched stage of k_qa_final_bwd
        K.check_first_conv_qa_wgrad(be, x_shape=(20, 3, 32, 32), Oc=8, k=3, seed=8)


def test_first_conv_fused_block(be):
    K.check_first_conv_fused(be, act=1)
    K.check_first_conv_fused(be, act=2, bits=2, seed=1)
    K.check_first_conv_fused(be, x_shape=(2, 3, 16, 16), Oc=72, k=3, act=1, bias=False, seed=2)
    K.check_first_conv_fused(be, x_shape=(2, 3, 16, 16), Oc=40, k=3, act=2, bits=4, seed=3)


def test_first_conv_block_edge_geometries(be):
    """two channel blocks (O > 256), more image tiles than Gram blocks (grid-stride tile loop), a single input channel (K = 9)"""
    K.check_first_conv_gram_bwd(be, x_shape=(2, 3, 8, 8), Oc=320, k=5, kind="bn", seed=11)
    K.check_first_conv_fused(be, x_shape=(2, 3, 8, 8), Oc=320, k=5, act=1, seed=12)
    K.check_first_conv_xgram(be, (520, 1, 4, 8), 3, seed=14)
    K.check_first_conv_gram_bwd(be, x_shape=(2, 1, 8, 8), Oc=24, k=3, kind="bn", seed=16)


def test_first_conv_gram_backward(be):
    K.check_first_conv_gram_bwd(be, kind="bn")
    K.check_first_conv_gram_bwd(be, kind="qa", quant=1, bits=2, seed=1)
    K.check_first_conv_gram_bwd(be, x_shape=(2, 3, 16, 16), Oc=72, k=3, kind="bn", bias=False, seed=2)
    K.check_first_conv_gram_bwd(be, x_shape=(2, 3, 16, 16), Oc=40, k=3, kind="qa", quant=0, bits=4, seed=3)


@pytest.mark.parametrize("training", [True, False])
def test_first_conv_bn_wgrad(be, training):
    K.check_first_conv_bn_wgrad(be, training=training)
    K.check_first_conv_bn_wgrad(be, x_shape=(2, 3, 16, 16), Oc=160, k=3, training=training, seed=1)
    if training:          # many partial tiles / 20 partial rows per channel: the batched (8 / 4 loads in flight) stages of k_c1_reduce and k_bns_final_bwd
        K.check_first_conv_bn_wgrad(be, x_shape=(20, 3, 32, 32), Oc=8, k=3, training=True, seed=5)


@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_qconv_bnsign_byte_stash(be, case):
    """forward stash h = (acc + nnz) / 2 + the streaming BatchNorm+sign backward on (da, h)."""
    K.check_qconv_bnsign(be, seed=220 + case, stash=True, **K.QGEMM_PW_CASES[case])
    K.check_qconv_bnsign(be, seed=225 + case, stash=True, training=False, **K.QGEMM_PW_CASES[case])
    if case in (1, 2):
        K.check_qconv_bnsign(be, seed=230 + case, stash=True, pooled=True, **K.QGEMM_PW_CASES[case])


def test_conv_backward_with_bn_and_maxpool_folded_in(be):
    """mn_conv2d_bwd_data_bnh_pool / mn_conv2d_bwd_weight_bnh_pool: the conv's backward forms dy from (pooled gradient, the block's own sign codes, h) -- the pool's
    first-maximum routing and the BatchNorm+sign backward in the operand load -- against the two-step path through mn_bnh_bwd_apply's full-size dy."""
    before = getattr(K.check_qconv_bnsign, "pool_fold_checked", 0)
    for i, case in enumerate(K.WGRAD_SPEC_CASES):          # (the emulator runs every geometry once, eval mode on one of them; the GPU twin runs the cross product)
        K.check_qconv_bnsign(be, seed=320 + i, stash=True, pooled=True, **case)
        if i == 1:
            K.check_qconv_bnsign(be, seed=330 + i, stash=True, pooled=True, training=False, **case)
    assert getattr(K.check_qconv_bnsign, "pool_fold_checked", 0) - before == 4


def test_pointwise_block_backward_in_one_kernel(be):
    """mn_conv2d_bwd_bnh (k_pwb): backward-data + backward-weight of the binary block from ONE read of (da, h), pooled and unpooled, against the two-kernel path."""
    K.check_pwb(be, light=True)


def test_kbit_block_backward_in_one_kernel(be):
    """mn_conv2d_bwd_codes / mn_conv2d_bwd_qa (k_pwb on k-bit activation codes): both gradients of a DoReFa block in one launch, from the plain gradient and with
    the BatchNorm + ReLU + quantizer backward formed from (dq, stash) inside, against the two-kernel path."""
    K.check_pwb_bnq(be)


def test_conv_backward_with_bn_folded_in(be):
    """mn_conv2d_bwd_data_bnh / mn_conv2d_bwd_weight_bnh (dy formed in registers from (da, h)) on shapes the direct kernels cover."""
    K.check_qconv_bnsign(be, seed=250, stash=True, x_shape=(2, 128, 4, 8), w_shape=(128, 64, 1, 1), groups=2)
    K.check_qconv_bnsign(be, seed=251, stash=True, x_shape=(3, 256, 4, 8), w_shape=(256, 128, 1, 1), groups=2, in_shuffle=2, bias=False)
    K.check_qconv_bnsign(be, seed=252, stash=True, training=False, x_shape=(2, 80, 4, 4), w_shape=(100, 40, 1, 1), groups=2)


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
    """k_h_sign_prep (the default) instead of k_pws_stats_prep + k_h_sign."""
    K.check_hsign_fold(be, KXK_STASH_CASES)


def test_sign_pass_two_launch_path_opt_out():
    """MN_HSIGN_FOLD=0 (child process: the knob is read once): k_pws_stats_prep + k_h_sign, the path the generic k x k forward still takes."""
    K.run_child("K.check_qconv_bnsign(be, seed=221, stash=True, **K.QGEMM_PW_CASES[1]); assert K.check_qconv_bnsign.last_fwd_kernel == 'k_h_sign', K.check_qconv_bnsign.last_fwd_kernel",
                "emu", {"MN_HSIGN_FOLD": "0"}, 900)


def test_ternary_weight_quantizer_multi(be):
    K.check_ternary_multi(be)
    K.check_binary_multi(be)


@pytest.mark.parametrize("training", [True, False])
def test_bnrelu(be, training):
    K.check_bnrelu(be, training=training)
    K.check_bnrelu(be, shape=(3, 7, 2, 2), seed=3, training=training)


def test_pool_f32(be):
    K.check_pool_f32(be)
    K.check_pool_f32(be, shape=(2, 3, 2, 8), seed=1)


@pytest.mark.parametrize("case", range(len(K.BNQ_CASES)))
def test_qconv_bnq_block(be, case):
    """k-bit (DoReFa) conv + BatchNorm + ReLU + next-layer quantizer on activation codes: 16-bit stash, streaming forward / backward, conv backward."""
    K.check_qconv_bnq(be, seed=300 + case, **K.BNQ_CASES[case])


def test_cifar_augment_kernel_vs_oracle(be):
    """mn_cifar_augment (RandomCrop(32, 4) + HFlip + ToTensor + Normalize for given draws) == the numpy restatement of wqaq/dorefa/main.py:203-210, bit for bit."""
    import ctypes as C
    from oracle import np_oracle as O
    r = np.random.default_rng(0)
    imgs = r.integers(0, 256, size=(10, 32, 32, 3), dtype=np.uint8)
    B = 9
    idx, ox, oy = r.integers(0, 10, size=B).astype(np.int32), r.integers(0, 9, size=B).astype(np.int32), r.integers(0, 9, size=B).astype(np.int32)
    ox[:4], oy[:4] = [0, 8, 0, 8], [0, 0, 8, 8]
    flip = (r.random(B) < 0.5).astype(np.uint8)
    out = np.zeros((B, 3, 32, 32), dtype=np.float32)
    FA = C.c_float * 3
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    be.call("mn_cifar_augment", p(imgs), 10, p(idx), p(ox), p(oy), p(flip), B, 32, 32, 3, 4, FA(0.4914, 0.4822, 0.4465), FA(0.2023, 0.1994, 0.2010), p(out), None)
    assert np.array_equal(out, O.cifar_augment(imgs, idx, ox, oy, flip))


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
    """4-bit activations x 4-bit weights: the accumulator leaves int16 -> 32-bit stash."""
    K.check_qdense(be, (2, 64, 8, 8), 64, 3, 1, a_bits=4, w_bits=4, seed=320)


@pytest.mark.parametrize("in_kind,res_kind", [(0, 1), (0, 2), (2, 3), (1, 1), (0, 0), (2, 1)])
@pytest.mark.parametrize("training", [True, False])
def test_residual_block_end(be, in_kind, res_kind, training):
    K.check_qr(be, in_kind=in_kind, res_kind=res_kind, training=training, with_dq2=(res_kind != 1), with_gf=(res_kind != 0), seed=330 + in_kind * 4 + res_kind)
    K.check_qr(be, shape=(2, 3, 2, 4), in_kind=in_kind, res_kind=res_kind, bits=4, training=training, with_dq2=True, with_gf=False, seed=350)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_qlinear(be, mode):
    K.check_qlinear(be, mode=mode, seed=360 + mode)
    K.check_qlinear(be, N=19, Cc=512, Oc=10, mode=mode, bits=4, bias=False, seed=365 + mode)
    K.check_qlinear(be, N=3, Cc=64, Oc=64, mode=mode, seed=370 + mode)


def test_dorefa_weight_quantizer_multi_cached(be):
    """mn_dorefa_w_fwd_multi_cached / _bwd_multi_cached (tanh(w) kept between forward and backward) == the per-tensor entry points, bit for bit."""
    import ctypes as C
    r = np.random.default_rng(11)
    shapes = [(24, 16, 1, 1), (8, 4, 3, 3), (7,), (3000,)]
    ws = [be.to_dev(r.standard_normal(s) * 0.7) for s in shapes]
    gs = [be.to_dev(r.standard_normal(s)) for s in shapes]
    lib, n = be.lib, len(shapes)
    PA, LA = C.c_void_p * n, C.c_int64 * n
    pa = lambda arrs: PA(*[be.ptr(a).value for a in arrs])
    for bits in (2, 8):
        ref = []
        for w, g in zip(ws, gs):
            sc = be.empty(int(lib.mn_dorefa_w_ws_floats(w.size)))
            q, d = be.empty(w.shape), be.empty(w.shape)
            be.call("mn_dorefa_w_fwd", be.ptr(w), be.ptr(q), w.size, bits, be.ptr(sc), be.stream)
            be.call("mn_dorefa_w_bwd", be.ptr(g), be.ptr(w), be.ptr(d), w.size, bits, be.ptr(sc), be.stream)
            ref.append((q, d))
        qs, ds, ths = [be.empty(w.shape) for w in ws], [be.empty(w.shape) for w in ws], [be.empty(w.shape) for w in ws]
        scs = [be.empty(int(lib.mn_dorefa_w_ws_floats(w.size))) for w in ws]
        sizes = LA(*[w.size for w in ws])
        be.call("mn_dorefa_w_fwd_multi_cached", pa(ws), pa(qs), pa(scs), pa(ths), sizes, n, bits, be.stream)
        be.call("mn_dorefa_w_bwd_multi_cached", pa(gs), pa(ws), pa(ds), pa(scs), pa(ths), sizes, n, bits, be.stream)
        for (q, d), q2, d2 in zip(ref, qs, ds):
            assert np.array_equal(q, q2) and np.array_equal(d, d2)


@pytest.mark.parametrize("case", [0, 1, 4, 6, 9])
def test_qdense_layer_iao(be, case):
    xs, Oc, k, s = K.QDENSE_CASES[case]
    K.check_qdense_iao(be, xs, Oc, k, s, seed=400 + case)


@pytest.mark.parametrize("shape,bits,q_type,relu,scbn", [((3, 6, 4, 8), 4, 0, True, False), ((2, 5, 8, 8), 8, 1, True, True), ((3, 4, 2, 4), 4, 0, False, True),
                                                         ((9, 3, 4, 4), 6, 0, False, False)])
def test_iao_qadd_bn_fused(be, shape, bits, q_type, relu, scbn):
    K.check_iao_qadd_bn(be, shape, bits, q_type, relu, scbn, seed=430 + bits)


def test_tail_bn_relu_pool_and_loss(be):
    K.check_tail(be, (37, 10, 8, 8), seed=450)
    K.check_tail(be, (300, 3, 2, 4), seed=452)


def test_qdense_layer_iao_w8a8_bias(be):
    K.check_qdense_iao(be, (2, 64, 8, 8), 64, 3, 1, a_bits=8, w_bits=8, bias=True, seed=410)


@pytest.mark.parametrize("bits,q_type,obs_kind,first,update", [(8, 0, 1, (True, True), True), (4, 0, 1, (False, False), True), (8, 1, 1, (False, True), True),
                                                              (8, 0, 0, (False, False), True), (4, 0, 1, (False, False), False)])
def test_iao_quant_add_fused(be, bits, q_type, obs_kind, first, update):
    K.check_iao_qadd(be, bits=bits, q_type=q_type, obs_kind=obs_kind, first=first, update=update, seed=bits + q_type)
    K.check_iao_qadd(be, bits=bits, q_type=q_type, obs_kind=obs_kind, first=first, update=update, seed=bits + q_type + 7, relu=True)
    K.check_iao_qadd(be, n=12, bits=bits, q_type=q_type, obs_kind=obs_kind, first=first, update=update, seed=1)
    K.check_iao_qadd(be, n=4096, bits=bits, q_type=q_type, obs_kind=obs_kind, first=first, update=update, seed=2)          # (+ the observers from producer partials)


@pytest.mark.parametrize("bits,q_type,obs_kind", [(4, 0, 0), (8, 0, 1), (8, 1, 0)])
def test_iao_weight_quantizers_multi(be, bits, q_type, obs_kind):
    K.check_iao_w_multi(be, bits=bits, q_type=q_type, obs_kind=obs_kind, seed=bits)


@pytest.mark.parametrize("training", [True, False])
def test_bn2d_plain(be, training):
    K.check_bnrelu(be, training=training, plain=True)
    K.check_bnrelu(be, shape=(3, 7, 2, 6), seed=3, training=training, plain=True)


def test_global_avgpool(be):
    K.check_gap(be)
    K.check_gap(be, planes=5, HW=9, seed=1)


@pytest.mark.parametrize("w_bits,iao", [(2, False), (8, False), (4, True)])
def test_qd_pack_multi_tables(be, w_bits, iao):
    K.check_qd_pack_multi(be, w_bits=w_bits, iao=iao, seed=w_bits)


# ---- the BN-fused IAO block without the statistics convolution (iao_bnfuse.hip)
@pytest.mark.parametrize("case", range(5))
def test_iaobf_pointwise(be, case):
    """Gram data -> batch statistics / fold / weight quantizer in one launch -> conv + ReLU + (min, max) -> backward-weight -> one-launch backward preparation ->
    backward-data with the raw path folded in, two training steps, against the oracle's QuantBNFuseConv2d + ReLU in fp32 and fp64."""
    import iaobf_cases as B
    B.check_iaobf_pointwise(be, B.CASES[case], seed=case)


@pytest.mark.parametrize("bits,q_type,relu_mask", [(8, 0, False), (4, 1, False), (8, 0, True)])
def test_iao_fq_maxpool(be, bits, q_type, relu_mask):
    import iaobf_cases as B
    B.check_fq_maxpool(be, bits=bits, q_type=q_type, relu_mask=relu_mask, seed=bits + q_type)


def test_bnfuse_stream_helpers(be):
    import iaobf_cases as B
    B.check_stream_helpers(be)


def test_iao_codes_at_rounding_boundaries(be):
    import iaobf_cases as B
    B.check_iao_codes_at_boundaries(be)


@pytest.mark.parametrize("k,Cin,W", [(5, 3, 12), (3, 7, 8), (5, 5, 4)])
def test_iaobf_gram_of_first_layer_patches(be, k, Cin, W):
    import iaobf_cases as B
    B.check_gram_patch(be, Cin=Cin, W=W, k=k, seed=k + Cin)


@pytest.mark.parametrize("ci", range(3))
def test_iaobf_grouped_3x3_family(be, ci):
    """csrc/iao_g3.hip: statistics of the raw conv, quantised conv + ReLU + (min, max), d y_raw, both backward-weights, the two-path backward-data -- each
    against an fp64 evaluation (<= 2e-6), through the channel shuffle, one and several tiles per persistent block."""
    import iaobf_cases as B
    B.check_g3(be, B.G3_CASES[ci], seed=ci)


def test_first_layer_forward_with_relu_and_minmax_epilogue(be):
    import iaobf_cases as B
    B.check_first_layer_act(be)
    B.check_first_layer_act(be, N=2, Cin=3, H=4, W=8, O=70, k=3, seed=3)


@pytest.mark.parametrize("shuffle,bias", [(0, True), (2, False)])
def test_iaobf_thin_output_family(be, shuffle, bias):
    """csrc/iao_thin.hip: raw / quantised pointwise conv with <= 16 outputs, both backward-weights, the two-path backward-data against fp64"""
    import iaobf_cases as B
    B.check_thin(be, shuffle=shuffle, bias=bias, seed=shuffle)


def test_iaobf_gram_statistics_never_negative_variance(be):
    import iaobf_cases as B
    B.check_gram_stats_variance_clamp(be)


def test_first_conv_gram_statistics_on_unnormalised_images_and_difference_filters(be):
    K.check_first_conv_gram_conditioning(be)
    K.check_first_conv_gram_conditioning(be, x_shape=(8, 3, 32, 32), Oc=24, k=5, seed=3)
