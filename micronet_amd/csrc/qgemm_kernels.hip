// Code-domain convolution kernels for gfx950 (CDNA4): the contraction runs on v_mfma_f32_16x16x32_bf16.
//
// Why this is exact.  Every fake-quantised operand of the reference is `integer code x scale` with |code| <= 255 at
// <= 8 bit: DoReFa act j*s, weight (2k-n)/n; ternary/binary t*alpha[o]; IAO (clamp(r)+zp)*s[o].
// Integers up to 256 are exact in bf16, their products are exact in fp32 and the fp32 accumulation is exact while
// K*max|a|*max|w| < 2^24 -- so the forward is ONE bf16 MFMA pass over codes and an epilogue `* s_a*s_w[o] + bias`.
// The backward passes have one real-valued operand (gy): it is written as gy = t0 + t1 + t2 with t_i the bf16 head of the
// running remainder (exact: 3 x 8 significant bits = the 24 of an fp32), so three MFMA passes form exactly the fp32
// products the reference forms, at 3/16 of the fp32-MFMA cost.  Real-valued activations (wbwtab feeds the +-1 of
// BinaryActivation as ordinary floats) go through the same split and the passes of an all-zero term are skipped
// wave-uniformly, so +-1 inputs cost one pass and arbitrary floats stay exact.
//
// Pointwise (1x1, stride 1) convolutions -- 72 % of nin_gc's activation bytes -- need no LDS for the activations:
//   lane (j = lane&15, kg = lane>>4) loads float4 = 4 consecutive pixels of each of 8 channels (16 lanes -> 256-B runs);
//   pixel column q of those loads is the B fragment (k = channel, column = pixel 4j+q) of MFMA q, q = 0..3;
//   the A fragments are weight codes read from LDS (row = out-channel); D[q] leaves lane (j, kg) with out-channels
//   4kg..4kg+3 of pixel 4j+q, i.e. across q a float4 of 4 consecutive pixels per out-channel: stores are again 256-B runs.
// The pixel <-> MFMA-column permutation costs nothing.  Backward-data is the same kernel on gy with transposed codes, the
// per-channel weight scale applied to gy before the split, and the clip-STE of the activation quantizer in the epilogue.
// Backward-weight contracts over pixels: both operands are pixel-contiguous in NCHW, so 64-pixel slabs of gy (3 terms)
// and of the activation codes are staged in LDS as bf16 rows and read back as b128 fragments; the grid splits the pixel
// range (Z) and a second kernel reduces the partial tiles in a fixed order with fp64 accumulation (deterministic).
#include "qgemm.h"

#include "qgemm_dev.h"

#include <stdlib.h>

__device__ __forceinline__ float wq_code(float w, int mode, float sc, float n) {
    if (mode == MN_WQ_TERNARY) return (w > 0.f) ? 1.f : ((w < 0.f) ? -1.f : w);   // +-0 -> 0, NaN stays NaN
    if (mode == MN_WQ_DOREFA) {
        const float s = 1.0f / n;                 // fp32(1/n): the same scale the quantizer divides by
        const float k = mn_rha(((w + 1.0f) * 0.5f) / s);
        return 2.0f * k - n;
    }
    return mn_rha(w / sc);                        // IAO: (clamp(r) + zp)
}
__device__ __forceinline__ void qg_pack_row(const PackParams& p, int blk) {
    const int rows = p.transpose ? p.Mgp : p.Mpad;
    const int g = blk / rows, m = blk % rows;
    const int lane = threadIdx.x;
    const bool mv = m < p.Mg;
    const int K = p.Cg * p.T;
    const float* wr = p.w + ((int64_t)g * p.Mg + (mv ? m : 0)) * K;
    float sc = 0.f;
    const float n = (float)((1ll << p.bits) - 1);
    if (mv) {
        if (p.mode == MN_WQ_TERNARY) {
            float mx = 0.f;
            for (int i = lane; i < K; i += 64) mx = OpMaxF()(mx, fabsf(wr[i]));
            mx = wave_reduce(mx, OpMaxF());
            sc = __shfl(mx, 0, 64);
        } else if (p.mode == MN_WQ_DOREFA) {
            sc = 1.0f / n;
        } else {
            sc = p.scale_in[(int64_t)(g * p.Mg + m) * p.per_channel];   // per_channel = stride in floats (0: one scale)
        }
    }
    if (lane == 0) p.scale_out[g * rows + m] = sc;
    if (!p.transpose) {
        uint16_t* dst = p.codes + ((int64_t)g * p.Mpad + m) * p.T * p.Cgp;
        for (int i = lane; i < p.T * p.Cgp; i += 64) {
            const int tap = i / p.Cgp, c = i - tap * p.Cgp;
            float code = 0.f;
            if (mv && c < p.Cg) code = wq_code(wr[c * p.T + tap], p.mode, sc, n);
            dst[i] = (uint16_t)(mn_f2u(code) >> 16);
        }
    } else {
        for (int i = lane; i < p.T * p.Cpad; i += 64) {
            const int c = i / p.T, tap = i - c * p.T;
            float code = 0.f;
            if (mv && c < p.Cg) code = wq_code(wr[c * p.T + tap], p.mode, sc, n);
            const int tapflip = p.T - 1 - tap;
            p.codes[(((int64_t)g * p.Cpad + c) * p.T + tapflip) * p.Mgp + m] = (uint16_t)(mn_f2u(code) >> 16);
        }
    }
}

__global__ __launch_bounds__(64) void k_qg_pack(const PackParams p) { qg_pack_row(p, (int)blockIdx.x); }
// the packs of SEVERAL pointwise layers (forward and backward-data code images) in one launch: mn_qg_pack_multi -- once per training step, right after the weight
// quantizers, instead of one 5 us launch per conv call and direction (10 per nin_gc step)
#define QG_PACKM_MAX 20
struct PackTable { PackParams e[QG_PACKM_MAX]; int blk0[QG_PACKM_MAX + 1]; int count; };
__global__ __launch_bounds__(64) void k_qg_pack_multi(const PackTable t) {
    int i = 0;
    while (i + 1 < t.count && (int)blockIdx.x >= t.blk0[i + 1]) ++i;
    qg_pack_row(t.e[i], (int)blockIdx.x - t.blk0[i]);
}
void qg_launch_pack(const PackParams& p, int grid, hipStream_t s) { hipLaunchKernelGGL(k_qg_pack, dim3(grid), dim3(64), 0, s, p); }

// ------------------------------------------------------------------------------------------------
// pointwise forward / backward-data
struct PwParams {
    const float* x;          // streamed operand: x (fwd) or gy (bwd-data)   [N][G*Kc][HW]
    float* y;                // y (fwd) or dx (bwd-data)                      [N][G*Mr][HW]
    const uint16_t* wc;      // codes [G][Mpad][Kp]
    const float* rowscale;   // [G][Mpad] scale of output row (fwd) or null
    const float* kscale;     // [G][Kp]   scale of contraction channel (bwd-data) or null
    const float* bias;       // fwd
    const float* aux;        // bwd-data STE: x
    Pro pro, ste;
    int N, HW, Cin_total, Cout_total, Kc, Mr, G;
    int Kp, KS, Mpad, num_mblk, nchunks, CB, epi;
    uint32_t NP;
    FastDiv fd_hw, fd_ks;
    float ascale;
    ChanMap in_map, out_map;   // channel shuffle folded into the input (fwd) / output (bwd-data) addressing
    int relu;                  // QG_EPI_SCALE_BIAS only: y = relu(y) (the nn.ReLU behind a BN-fused IAO conv, models/nin_gc.py:53-59 with bn = Identity)
    float* mm;                 // nullable: per-wave (min, max) of what this launch stores -> mm[4 b + wave], mm[4 gridDim + 4 b + wave]: the NEXT layer's observer
};

template <int NT, int XMODE>
__global__ __launch_bounds__(256, 2) void k_pw(const PwParams p) {
    HIP_DYNAMIC_SHARED(float, smem)
    constexpr int MB = 16 * NT;
    const int LDW = p.Kp + 8;                      // u16 per weight row: +16 B spreads the 16 rows of a fragment over all banks
    uint16_t* wsm = reinterpret_cast<uint16_t*>(smem);
    float* rs = smem + (MB * LDW) / 2;
    float* bs = rs + MB;
    float* ks = bs + MB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kg = lane >> 4;

    // block -> (group, m-block, chunk-block); blocks that share a chunk set differ only in m-block and sit on one XCD (b % 8)
    uint32_t b = blockIdx.x;
    const uint32_t xcd = b & 7u; b >>= 3;
    const int mblk = b % p.num_mblk; b /= p.num_mblk;
    const uint32_t idx = b * 8u + xcd;
    if (idx >= (uint32_t)(p.G * p.CB)) {
        if (p.mm && lane == 0) { p.mm[4 * blockIdx.x + wave] = INFINITY; p.mm[4 * gridDim.x + 4 * blockIdx.x + wave] = -INFINITY; }
        return;
    }
    const int cb = idx % p.CB, g = idx / p.CB;
    float mlo = INFINITY, mhi = -INFINITY;
    int mnan = 0;

    {   // stage weight codes, scales, bias
        const uint16_t* wg = p.wc + ((int64_t)g * p.Mpad + mblk * MB) * p.Kp;
        const int k8 = p.Kp >> 3;
        for (int q = tid; q < MB * k8; q += 256) {
            const int row = q / k8, c8 = q - row * k8;
            *reinterpret_cast<u32x4*>(wsm + row * LDW + c8 * 8) = *reinterpret_cast<const u32x4*>(wg + (int64_t)row * p.Kp + c8 * 8);
        }
        float as = p.ascale;
        if (XMODE == MN_ACTQ_IAO) as = p.pro.qp[0];
        for (int i = tid; i < MB; i += 256) {
            const int m = mblk * MB + i;
            rs[i] = p.rowscale ? p.rowscale[g * p.Mpad + m] * as : 1.f;
            bs[i] = (p.bias && m < p.Mr) ? p.bias[g * p.Mr + m] : 0.f;
        }
        for (int i = tid; i < p.Kp; i += 256) ks[i] = p.kscale ? p.kscale[g * p.Kp + i] : 1.f;
    }
    __syncthreads();

    float sc = 1.f, zp = 0.f;
    if (XMODE == MN_ACTQ_IAO) { sc = p.pro.qp[0]; zp = p.pro.qp[1]; }
    const float inv_sc = 1.0f / sc;
    float ste_sc = 1.f, ste_zp = 0.f, ste_lo = 0.f, ste_hi = 0.f;
    if (p.epi == QG_EPI_STE && p.ste.mode == MN_ACTQ_IAO) { ste_sc = p.ste.qp[0]; ste_zp = p.ste.qp[1]; ste_lo = p.ste.qp[2]; ste_hi = p.ste.qp[3]; }

    const int chunk0 = cb * 4 + wave, cstride = p.CB * 4;
    const int my_chunks = chunk0 < p.nchunks ? (p.nchunks - chunk0 + cstride - 1) / cstride : 0;
    const int total = my_chunks * p.KS;
    const uint16_t* wl = wsm + j * LDW + kg * 8;
    const int64_t HW = p.HW;

    f32x4 acc[4][NT];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    // MN_ACTQ_SIGN8: the stream is int8 sign codes -- one dword (4 pixels) per channel lands in raw[jj][0] as raw bits
    auto issue = [&](float (&raw)[8][4], int it) {
        const uint32_t ci = fd_div((uint32_t)it, p.fd_ks);
        const int s = it - (int)ci * p.KS;
        const uint32_t P = (uint32_t)(chunk0 + (int)ci * cstride) * 64u + 4u * j;
        const bool pv = P < p.NP;
        const uint32_t n = fd_div(P, p.fd_hw);
        const int pp = (int)(P - n * (uint32_t)p.HW);
        const int64_t boff = (int64_t)n * p.Cin_total * HW + pp;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int c = s * 32 + kg * 8 + jj;
            if (XMODE == MN_ACTQ_SIGN8) {
                uint32_t u = 0u;
                if (pv && c < p.Kc) u = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(p.x) + boff + (int64_t)chan_phys(p.in_map, g * p.Kc + c) * HW);
                raw[jj][0] = mn_u2f(u);
            } else {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pv && c < p.Kc) v = *reinterpret_cast<const float4*>(p.x + boff + (int64_t)chan_phys(p.in_map, g * p.Kc + c) * HW);
                raw[jj][0] = v.x; raw[jj][1] = v.y; raw[jj][2] = v.z; raw[jj][3] = v.w;
            }
        }
    };

    // one K-step: build the B fragments from the landed loads, re-issue the loads of the next step into the same registers
    // (they fly during this step's LDS reads + MFMAs), contract, store at the end of a chunk
    auto compute = [&](float (&raw)[8][4], int it) {
        const uint32_t ci = fd_div((uint32_t)it, p.fd_ks);
        const int s = it - (int)ci * p.KS;
        u32x4 b0[4], b1[4], b2[4];
        int use1 = 0, use2 = 0;
        if (XMODE == MN_ACTQ_NONE) {
            unsigned any1 = 0u, any2 = 0u;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float t0[8], t1[8], t2[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = raw[e][q];
                    if (p.kscale) v = v * ks[s * 32 + kg * 8 + e];
                    t0[e] = mn_bf16_head(v);
                    const float r1 = v - t0[e];
                    t1[e] = mn_bf16_head(r1);
                    const float r2 = r1 - t1[e];
                    t2[e] = r2;                            // <= 8 significant bits: its bf16 head is r2 itself
                    any1 |= mn_f2u(r1) << 1;               // ignore the sign of a zero
                    any2 |= mn_f2u(r2) << 1;
                }
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    b0[q][d] = mn_pack_bf16x2(t0[2 * d], t0[2 * d + 1]);
                    b1[q][d] = mn_pack_bf16x2(t1[2 * d], t1[2 * d + 1]);
                    b2[q][d] = mn_pack_bf16x2(t2[2 * d], t2[2 * d + 1]);
                }
            }
            use1 = mn_wave_any(any1 != 0u);
            use2 = mn_wave_any(any2 != 0u);
        } else if (XMODE == MN_ACTQ_SIGN8) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int d = 0; d < 4; ++d) b0[q][d] = mn_sign8_pair(mn_f2u(raw[2 * d][0]), mn_f2u(raw[2 * d + 1][0]), q);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float c8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    c8[e] = XMODE == MN_ACTQ_IAO ? iao_code_m(raw[e][q], sc, inv_sc, zp, p.pro.qmin, p.pro.qmax) : act_code<XMODE>(raw[e][q], p.pro, sc, zp);
#pragma unroll
                for (int d = 0; d < 4; ++d) b0[q][d] = mn_pack_bf16x2(c8[2 * d], c8[2 * d + 1]);
            }
        }
        if (it + 1 < total) issue(raw, it + 1);
        const uint16_t* wk = wl + s * 32;
        // term-outer: 4*NT independent accumulators between two MFMAs on the same one (a dependent MFMA stalls the issue)
        u32x4 av[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) av[t] = *reinterpret_cast<const u32x4*>(wk + t * 16 * LDW);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q][t] = mn_mfma_bf16(av[t], b0[q], acc[q][t]);
        if (XMODE == MN_ACTQ_NONE) {
            if (use1) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q][t] = mn_mfma_bf16(av[t], b1[q], acc[q][t]);
            }
            if (use2) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q][t] = mn_mfma_bf16(av[t], b2[q], acc[q][t]);
            }
        }
        if (s == p.KS - 1) {
            const uint32_t P = (uint32_t)(chunk0 + (int)ci * cstride) * 64u + 4u * j;
            const uint32_t n = fd_div(P, p.fd_hw);
            const int pp = (int)(P - n * (uint32_t)p.HW);
            if (P < p.NP) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ml = t * 16 + kg * 4 + r;
                        const int m = mblk * MB + ml;
                        if (m < p.Mr) {
                            const int64_t off = ((int64_t)n * p.Cout_total + chan_phys(p.out_map, g * p.Mr + m)) * HW + pp;
                            float o0 = acc[0][t][r], o1 = acc[1][t][r], o2 = acc[2][t][r], o3 = acc[3][t][r];
                            if (p.epi == QG_EPI_SCALE_BIAS) {
                                const float a_ = rs[ml], b_ = bs[ml];
                                o0 = o0 * a_ + b_; o1 = o1 * a_ + b_; o2 = o2 * a_ + b_; o3 = o3 * a_ + b_;
                                if (p.relu) { o0 = qa_relu(o0); o1 = qa_relu(o1); o2 = qa_relu(o2); o3 = qa_relu(o3); }
                                if (p.mm) {          // plain min / max (NaN-ignoring, 8 instructions) + a NaN flag: torch.min / max propagate a NaN
                                    mlo = fminf(mlo, fminf(fminf(o0, o1), fminf(o2, o3)));
                                    mhi = fmaxf(mhi, fmaxf(fmaxf(o0, o1), fmaxf(o2, o3)));
                                    mnan |= (int)((o0 != o0) | (o1 != o1) | (o2 != o2) | (o3 != o3));
                                }
                            } else if (p.epi == QG_EPI_STE) {
                                const float4 xv = *reinterpret_cast<const float4*>(p.aux + off);
                                if (p.ste.mode == MN_ACTQ_DOREFA) {
                                    o0 = dorefa_act_grad(o0, xv.x, p.ste.s); o1 = dorefa_act_grad(o1, xv.y, p.ste.s);
                                    o2 = dorefa_act_grad(o2, xv.z, p.ste.s); o3 = dorefa_act_grad(o3, xv.w, p.ste.s);
                                } else if (p.ste.mode == MN_ACTQ_IAO) {
                                    o0 = iao_fq_grad(o0, xv.x, ste_sc, ste_zp, ste_lo, ste_hi, p.ste.qmin, p.ste.qmax);
                                    o1 = iao_fq_grad(o1, xv.y, ste_sc, ste_zp, ste_lo, ste_hi, p.ste.qmin, p.ste.qmax);
                                    o2 = iao_fq_grad(o2, xv.z, ste_sc, ste_zp, ste_lo, ste_hi, p.ste.qmin, p.ste.qmax);
                                    o3 = iao_fq_grad(o3, xv.w, ste_sc, ste_zp, ste_lo, ste_hi, p.ste.qmin, p.ste.qmax);
                                }
                            }
                            *reinterpret_cast<float4*>(p.y + off) = make_float4(o0, o1, o2, o3);
                        }
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };

    float ra[8][4];
    if (total > 0) issue(ra, 0);
    for (int it = 0; it < total; ++it) compute(ra, it);
    if (p.mm) {
        if (mnan) mlo = mhi = NAN;
        mlo = wave_reduce(mlo, OpMinF());
        mhi = wave_reduce(mhi, OpMaxF());
        if (lane == 0) { p.mm[4 * blockIdx.x + wave] = mlo; p.mm[4 * gridDim.x + 4 * blockIdx.x + wave] = mhi; }
    }
}

// ------------------------------------------------------------------------------------------------
// pointwise backward-data without a clip-STE epilogue (binary / ternary nets: the STE of the sign lives in the BatchNorm+sign
// backward), tuned for instruction issue and memory-level parallelism -- the structure of k_pws (qgemm_sign.hip):
//   * gy is streamed with UNCONDITIONAL float4 loads (indices clamped, never masked) from 32-bit offsets; the channel offsets
//     of a K-step come from an LDS table; the registers of a K-step are re-loaded with the data of TWO steps ahead as soon as
//     its fragments are built, so two steps (16 KB per wave) are always in flight;
//   * the three exact bf16 terms are built per pixel column q and contracted term-outer: NT independent accumulators
//     between two MFMAs on the same one;
//   * dx leaves as float4 = 4 consecutive pixels per channel through the output channel map (folded shuffle).
struct PwdParams {
    const float* gy;          // [N][G*Kc][HW]     (Kc = out-channels of the group: the contraction index)
    float* dx;                // [N][G*Mr][HW]     (Mr = in-channels of the group)
    const uint16_t* wc;       // transposed codes [G][Mpad][Kp]
    const float* kscale;      // [G][Kp] weight scale of contraction channel k
    int N, HW, Cin_total, Cout_total, Kc, Mr, G, Kp, KS, Mpad, num_mblk, nchunks, CB;
    uint32_t NP;
    FastDiv fd_hw;
    ChanMap out_map;
    // BNH variant: the streamed operand is not given, it is the BatchNorm+sign backward of (da, h) -- see k_bnh_apply (norm_kernels.hip):
    // gy := gi*dz - gi*k1 - gi*k2*zhat with dz = da*[hlo <= h <= hhi], zhat = (2h - nnz)*A + B, folded per channel into G*dz + E1*h + E0
    const unsigned char* h;   // [N][G*Kc][HW] one-byte conv stash
    const float* chan;        // [8][G*Kc]
    const float* sums;        // [2][G*Kc]
    int training;
    float n_f;
    // BNH 2: a 2x2 / stride-2 max-pool sits behind the block: gy is the POOLED gradient [N][G*Kc][H/2][W/2] and `own` the block's own sign output [N][G*Kc][H][W];
    // the gradient of a window goes to its first +1 in scan order (ATen's max_pool2d backward), as in k_bnh_apply<1> -- whose full-size dy is then never written
    const char* own;
    int W;
    FastDiv fd_w;
};
template <int NT, int KS, int BNH>
__global__ __launch_bounds__(256, 2) void k_pwd(const PwdParams p) {
    HIP_DYNAMIC_SHARED(float, smem)
    constexpr int MB = 16 * NT;
    const int LDW = p.Kp + 8;
    uint16_t* wsm = reinterpret_cast<uint16_t*>(smem);
    float* ks = smem + (MB * LDW) / 2;                            // [Kp]
    uint32_t* koff = reinterpret_cast<uint32_t*>(ks + p.Kp);      // [Kp] element offset of gy channel k (clamped)
    uint32_t* ooff = koff + p.Kp;                                  // [MB] element offset of the (shuffled) dx channel
    float* fhlo = reinterpret_cast<float*>(ooff + MB);            // BNH: [Kp] each
    float* fhhi = fhlo + p.Kp;
    float* fG = fhhi + p.Kp;
    float* fE1 = fG + p.Kp;
    float* fE0 = fE1 + p.Kp;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kg = lane >> 4;
    const uint32_t HW = (uint32_t)p.HW;

    uint32_t b = blockIdx.x;
    const uint32_t xcd = b & 7u; b >>= 3;
    const int mblk = b % p.num_mblk; b /= p.num_mblk;
    const uint32_t idx = b * 8u + xcd;
    if (idx >= (uint32_t)(p.G * p.CB)) return;
    const int cb = idx % p.CB, g = idx / p.CB;
    {
        const uint16_t* wg = p.wc + ((int64_t)g * p.Mpad + mblk * MB) * p.Kp;
        const int k8 = p.Kp >> 3;
        for (int q = tid; q < MB * k8; q += 256) {
            const int row = q / k8, c8 = q - row * k8;
            *reinterpret_cast<u32x4*>(wsm + row * LDW + c8 * 8) = *reinterpret_cast<const u32x4*>(wg + (int64_t)row * p.Kp + c8 * 8);
        }
        for (int k = tid; k < p.Kp; k += 256) {
            ks[k] = p.kscale[g * p.Kp + k];
            koff[k] = (uint32_t)(g * p.Kc + (k < p.Kc ? k : p.Kc - 1)) * HW;
            if (BNH) {
                float hlo = 1.f, hhi = 0.f, G = 0.f, E1 = 0.f, E0 = 0.f;      // padded channels: contribute exactly 0 (their weight codes are 0 too)
                if (k < p.Kc) bnh_fold(p.chan, p.sums, p.Cin_total, g * p.Kc + k, p.training, p.n_f, ks[k], hlo, hhi, G, E1, E0);
                fhlo[k] = hlo; fhhi[k] = hhi; fG[k] = G; fE1[k] = E1; fE0[k] = E0;
            }
        }
        for (int i = tid; i < MB; i += 256) {
            const int m = mblk * MB + i;
            ooff[i] = (uint32_t)chan_phys(p.out_map, g * p.Mr + (m < p.Mr ? m : p.Mr - 1)) * HW;
        }
    }
    __syncthreads();

    const int chunk0 = cb * 4 + wave, cstride = p.CB * 4;
    const int my_chunks = chunk0 < p.nchunks ? (p.nchunks - chunk0 + cstride - 1) / cstride : 0;
    const int total = my_chunks * KS;
    const uint16_t* wl = wsm + j * LDW + kg * 8;
    const uint32_t Pmax = p.NP - 4u;

    // K-step `it` (chunk it / KS, step it % KS) of this wave: 8 channels x 4 pixels per lane
    auto issue = [&](float4 (&raw)[8], uint32_t (&hb)[8], int it) {
        const int ci = it / KS, s = it - ci * KS;
        uint32_t P = (uint32_t)(chunk0 + ci * cstride) * 64u + 4u * j;
        P = P < Pmax ? P : Pmax;
        const uint32_t n = fd_div(P, p.fd_hw);
        const uint32_t go = n * (uint32_t)p.Cin_total * HW + (P - n * HW);
        const u32x4 o0 = *reinterpret_cast<const u32x4*>(koff + s * 32 + kg * 8), o1 = *reinterpret_cast<const u32x4*>(koff + s * 32 + kg * 8 + 4);
        if (BNH == 2) {
            // the lane's pixel quad (row hr, columns w .. w + 3) covers half of two pooling windows: raw = {g[win 0], g[win 1], own codes of row hr & ~1, of row hr | 1}
            const uint32_t pp = P - n * HW;
            const uint32_t hr = fd_div(pp, p.fd_w), w = pp - hr * (uint32_t)p.W;
            const uint32_t gbase = n * (uint32_t)p.Cin_total * (HW >> 2) + (hr >> 1) * ((uint32_t)p.W >> 1) + (w >> 1);      // + channel * HW / 4
            const uint32_t cbase = n * (uint32_t)p.Cin_total * HW + (hr & ~1u) * (uint32_t)p.W + w;                          // + channel * HW
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint32_t ko = e < 4 ? o0[e & 3] : o1[e & 3];
                const float2 g2 = *reinterpret_cast<const float2*>(p.gy + (gbase + (ko >> 2)));
                const uint32_t r0 = *reinterpret_cast<const uint32_t*>(p.own + (cbase + ko));
                const uint32_t r1 = *reinterpret_cast<const uint32_t*>(p.own + (cbase + ko + (uint32_t)p.W));
                raw[e] = make_float4(g2.x, g2.y, mn_u2f(r0), mn_u2f(r1));
                hb[e] = *reinterpret_cast<const uint32_t*>(p.h + (go + ko));
            }
            return;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            raw[e] = *reinterpret_cast<const float4*>(p.gy + (go + o0[e]));
            raw[4 + e] = *reinterpret_cast<const float4*>(p.gy + (go + o1[e]));
            if (BNH) {
                hb[e] = *reinterpret_cast<const uint32_t*>(p.h + (go + o0[e]));
                hb[4 + e] = *reinterpret_cast<const uint32_t*>(p.h + (go + o1[e]));
            }
        }
    };

    f32x4 acc[4][NT];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    float4 ra[8], rb[8];
    uint32_t ha[8], hbb[8];
    if (total > 0) issue(ra, ha, 0);
    if (total > 1) issue(rb, hbb, 1);
    auto step = [&](float4 (&raw)[8], uint32_t (&hb)[8], int it) {
        const int ci = it / KS, s = it - ci * KS;
        float v[8][4];
        uint32_t hbit = 0u;          // BNH 2: the parity of this lane's image row (which half of its pooling windows it holds)
        if (BNH == 2) {
            uint32_t P = (uint32_t)(chunk0 + ci * cstride) * 64u + 4u * j;
            P = P < Pmax ? P : Pmax;
            const uint32_t n = fd_div(P, p.fd_hw);
            hbit = fd_div(P - n * HW, p.fd_w) & 1u;
        }
        if (BNH) {
            // operand = BatchNorm+sign backward of (da, h), already times the weight scale: G*dz + E1*h + E0
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int kb = s * 32 + kg * 8 + half * 4;
                const f32x4 lo4 = *reinterpret_cast<const f32x4*>(fhlo + kb), hi4 = *reinterpret_cast<const f32x4*>(fhhi + kb);
                const f32x4 g4 = *reinterpret_cast<const f32x4*>(fG + kb), e14 = *reinterpret_cast<const f32x4*>(fE1 + kb), e04 = *reinterpret_cast<const f32x4*>(fE0 + kb);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float4 d4 = raw[half * 4 + e];
                    float dv[4] = {d4.x, d4.y, d4.z, d4.w};
                    if (BNH == 2) {          // route the two windows' gradients to their first +1 (row-major), else to element 0 -- only if that element is in THIS row
                        const uint32_t r0 = mn_f2u(d4.z), r1 = mn_f2u(d4.w);
#pragma unroll
                        for (int e2 = 0; e2 < 2; ++e2) {
                            const bool p00 = !((r0 >> (16 * e2)) & 0x80u), p01 = !((r0 >> (16 * e2 + 8)) & 0x80u);
                            const bool p10 = !((r1 >> (16 * e2)) & 0x80u), p11 = !((r1 >> (16 * e2 + 8)) & 0x80u);
                            const uint32_t win = p00 ? 0u : (p01 ? 1u : (p10 ? 2u : (p11 ? 3u : 0u)));
                            const float ge = e2 ? d4.y : d4.x;
                            dv[2 * e2] = win == hbit * 2u ? ge : 0.f;
                            dv[2 * e2 + 1] = win == hbit * 2u + 1u ? ge : 0.f;
                        }
                    }
                    const uint32_t hw = hb[half * 4 + e];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float hf = (float)((hw >> (8 * q)) & 0xffu);
                        const float dz = (hf >= lo4[e] && hf <= hi4[e]) ? dv[q] : 0.f;
                        v[half * 4 + e][q] = fmaf(g4[e], dz, fmaf(e14[e], hf, e04[e]));
                    }
                }
            }
        } else {
            // weight scale of the 8 contraction channels of this lane
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(ks + s * 32 + kg * 8), s1 = *reinterpret_cast<const f32x4*>(ks + s * 32 + kg * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e][0] = raw[e].x * s0[e]; v[e][1] = raw[e].y * s0[e]; v[e][2] = raw[e].z * s0[e]; v[e][3] = raw[e].w * s0[e];
                v[4 + e][0] = raw[4 + e].x * s1[e]; v[4 + e][1] = raw[4 + e].y * s1[e]; v[4 + e][2] = raw[4 + e].z * s1[e]; v[4 + e][3] = raw[4 + e].w * s1[e];
            }
        }
        if (it + 2 < total) issue(raw, hb, it + 2);           // the registers are free: two steps ahead
        u32x4 av[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) av[t] = *reinterpret_cast<const u32x4*>(wl + s * 32 + t * 16 * LDW);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u32x4 b0, b1, b2;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const float x0 = v[2 * d][q], x1 = v[2 * d + 1][q];
                const float h0 = mn_bf16_head(x0), h1 = mn_bf16_head(x1);
                const float r0 = x0 - h0, r1 = x1 - h1;
                const float m0 = mn_bf16_head(r0), m1 = mn_bf16_head(r1);
                b0[d] = mn_pack_bf16x2(h0, h1);
                b1[d] = mn_pack_bf16x2(m0, m1);
                b2[d] = mn_pack_bf16x2(r0 - m0, r1 - m1);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[q][t] = mn_mfma_bf16(av[t], b0, acc[q][t]);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[q][t] = mn_mfma_bf16(av[t], b1, acc[q][t]);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[q][t] = mn_mfma_bf16(av[t], b2, acc[q][t]);
        }
        if (s == KS - 1) {
            const uint32_t P = (uint32_t)(chunk0 + ci * cstride) * 64u + 4u * j;
            if (P < p.NP) {
                const uint32_t n = fd_div(P, p.fd_hw);
                const uint32_t ob = n * (uint32_t)p.Cout_total * HW + (P - n * HW);
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ml = t * 16 + kg * 4 + r;
                        if (mblk * MB + ml < p.Mr)
                            *reinterpret_cast<float4*>(p.dx + (ob + ooff[ml])) = make_float4(acc[0][t][r], acc[1][t][r], acc[2][t][r], acc[3][t][r]);
                    }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    for (int it = 0; it < total; it += 2) {
        step(ra, ha, it);
        if (it + 1 < total) step(rb, hbb, it + 1);
    }
}

// ------------------------------------------------------------------------------------------------
// pointwise backward-weight: dwq[g][m][c] = sum_{n,p} gy[n][g*Mg+m][p] * code_a[n][g*Cg+c][p]   (times the activation scale)
struct PwWgParams {
    const float* gy;
    const float* x;
    float* part;     // [Z][G][Mgw][Cgw]
    float* dbpart;   // [Z][G][Mgw]
    Pro pro;
    int N, HW, Cin_total, Cout_total, Cg, Mg, G;
    int nmb, ncb, Z, nchunks, Mgw, Cgw, want_db;
    uint32_t NP;
    FastDiv fd_hw;
    ChanMap in_map;
};
#define WG_LDP 72   // u16 per LDS row: 64 pixels + 8 pad -> 144 B rows, the 16 rows of a b128 fragment read hit all 64 banks

template <int MW, int CW, int WGC, int XMODE>
__global__ __launch_bounds__(256, 2) void k_pw_wgrad(const PwWgParams p) {
    HIP_DYNAMIC_SHARED(float, smem)
    constexpr int WGM = 4 / WGC, TM = 16 * MW * WGM, TC = 16 * CW * WGC, RM = TM / 16, RC = TC / 16;
    uint16_t* gt = reinterpret_cast<uint16_t*>(smem);        // [3][TM][WG_LDP]
    uint16_t* xq = gt + 3 * TM * WG_LDP;                     // [TC][WG_LDP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kg = lane >> 4;
    const int r0 = tid >> 4, qd = tid & 15;
    uint32_t b = blockIdx.x;
    const int z = b % p.Z; b /= p.Z;
    const int cb = b % p.ncb; b /= p.ncb;
    const int mb = b % p.nmb;
    const int g = b / p.nmb;
    const int wm = wave / WGC, wc = wave % WGC;
    const int64_t HW = p.HW;
    float sc = 1.f, zp = 0.f;
    if (XMODE == MN_ACTQ_IAO) { sc = p.pro.qp[0]; zp = p.pro.qp[1]; }
    const float inv_sc = 1.0f / sc;

    f32x4 acc[MW][CW];
#pragma unroll
    for (int mi = 0; mi < MW; ++mi)
#pragma unroll
        for (int ci = 0; ci < CW; ++ci) acc[mi][ci] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbacc[RM];
#pragma unroll
    for (int i = 0; i < RM; ++i) dbacc[i] = 0.f;

    float4 rg[RM], rx[RC];
    auto fetch = [&](int chunk) {
        const uint32_t P = (uint32_t)chunk * 64u + 4u * qd;
        const bool pv = P < p.NP;
        const uint32_t n = fd_div(P, p.fd_hw);
        const int pp = (int)(P - n * (uint32_t)p.HW);
#pragma unroll
        for (int i = 0; i < RM; ++i) {
            const int m = mb * TM + r0 + 16 * i;
            rg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pv && m < p.Mg) rg[i] = *reinterpret_cast<const float4*>(p.gy + ((int64_t)n * p.Cout_total + (int64_t)g * p.Mg + m) * HW + pp);
        }
#pragma unroll
        for (int i = 0; i < RC; ++i) {
            const int c = cb * TC + r0 + 16 * i;
            rx[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int64_t xoff = ((int64_t)n * p.Cin_total + chan_phys(p.in_map, g * p.Cg + c)) * HW + pp;
            if (XMODE == MN_ACTQ_SIGN8 || XMODE == MN_ACTQ_CODE8) {       // byte codes: one dword = the 4 pixels, kept as raw bits in rx[i].x
                if (pv && c < p.Cg) rx[i].x = mn_u2f(*reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(p.x) + xoff));
            } else if (pv && c < p.Cg) rx[i] = *reinterpret_cast<const float4*>(p.x + xoff);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < RM; ++i) {
            const float v[4] = {rg[i].x, rg[i].y, rg[i].z, rg[i].w};
            float t0[4], t1[4], t2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                t0[e] = mn_bf16_head(v[e]);
                const float r1 = v[e] - t0[e];
                t1[e] = mn_bf16_head(r1);
                t2[e] = r1 - t1[e];
            }
            dbacc[i] += (v[0] + v[1]) + (v[2] + v[3]);
            uint16_t* d = gt + (r0 + 16 * i) * WG_LDP + qd * 4;
            *reinterpret_cast<u32x2*>(d) = u32x2{mn_pack_bf16x2(t0[0], t0[1]), mn_pack_bf16x2(t0[2], t0[3])};
            *reinterpret_cast<u32x2*>(d + TM * WG_LDP) = u32x2{mn_pack_bf16x2(t1[0], t1[1]), mn_pack_bf16x2(t1[2], t1[3])};
            *reinterpret_cast<u32x2*>(d + 2 * TM * WG_LDP) = u32x2{mn_pack_bf16x2(t2[0], t2[1]), mn_pack_bf16x2(t2[2], t2[3])};
        }
#pragma unroll
        for (int i = 0; i < RC; ++i) {
            if (XMODE == MN_ACTQ_SIGN8) {
                const unsigned u = mn_f2u(rx[i].x);
                *reinterpret_cast<u32x2*>(xq + (r0 + 16 * i) * WG_LDP + qd * 4) =
                    u32x2{0x3F803F80u | ((u & 0x80u) << 8) | ((u & 0x8000u) << 16), 0x3F803F80u | ((u & 0x800000u) >> 8) | (u & 0x80000000u)};
                continue;
            }
            if (XMODE == MN_ACTQ_CODE8) {       // k-bit activation codes j (bytes): bf16 j, exact (rows beyond Cg / pixels beyond the tensor hold 0)
                const unsigned u = mn_f2u(rx[i].x);
                *reinterpret_cast<u32x2*>(xq + (r0 + 16 * i) * WG_LDP + qd * 4) =
                    u32x2{mn_pack_bf16x2((float)(u & 0xffu), (float)((u >> 8) & 0xffu)), mn_pack_bf16x2((float)((u >> 16) & 0xffu), (float)(u >> 24))};
                continue;
            }
            float c0, c1, c2, c3;
            if (XMODE == MN_ACTQ_IAO) {
                c0 = iao_code_m(rx[i].x, sc, inv_sc, zp, p.pro.qmin, p.pro.qmax); c1 = iao_code_m(rx[i].y, sc, inv_sc, zp, p.pro.qmin, p.pro.qmax);
                c2 = iao_code_m(rx[i].z, sc, inv_sc, zp, p.pro.qmin, p.pro.qmax); c3 = iao_code_m(rx[i].w, sc, inv_sc, zp, p.pro.qmin, p.pro.qmax);
            } else {
                c0 = act_code<XMODE>(rx[i].x, p.pro, sc, zp); c1 = act_code<XMODE>(rx[i].y, p.pro, sc, zp);
                c2 = act_code<XMODE>(rx[i].z, p.pro, sc, zp); c3 = act_code<XMODE>(rx[i].w, p.pro, sc, zp);
            }
            *reinterpret_cast<u32x2*>(xq + (r0 + 16 * i) * WG_LDP + qd * 4) = u32x2{mn_pack_bf16x2(c0, c1), mn_pack_bf16x2(c2, c3)};
        }
    };

    // MFMA phase over the slab currently in LDS: 3 gy terms x the x plane
    auto contract = [&]() {
#pragma unroll
        for (int ksx = 0; ksx < 2; ++ksx) {
            const int ko = ksx * 32 + kg * 8;
            u32x4 bf[CW];
#pragma unroll
            for (int ci = 0; ci < CW; ++ci) bf[ci] = *reinterpret_cast<const u32x4*>(xq + ((wc * CW + ci) * 16 + j) * WG_LDP + ko);
#pragma unroll
            for (int term = 0; term < 3; ++term) {          // term-outer: MW*CW independent accumulators between dependent MFMAs
#pragma unroll
                for (int mi = 0; mi < MW; ++mi) {
                    const u32x4 a = *reinterpret_cast<const u32x4*>(gt + (term * TM + (wm * MW + mi) * 16 + j) * WG_LDP + ko);
#pragma unroll
                    for (int ci = 0; ci < CW; ++ci) acc[mi][ci] = mn_mfma_bf16(a, bf[ci], acc[mi][ci]);
                }
            }
        }
    };
    // XMODE NONE only: x is an arbitrary float tensor.  The slab is contracted with the bf16 head of x (exact when x holds
    // small integers such as the +-1 of BinaryActivation); if any element of the slab has a remainder, the x plane is
    // re-staged with the second and third term of the split and contracted again -- exact for any fp32 input.
    int* xflag = reinterpret_cast<int*>(xq + TC * WG_LDP);   // [2], alternating per slab
    auto restage_x_term = [&](int chunk, int term) {
        const uint32_t P = (uint32_t)chunk * 64u + 4u * qd;
        const bool pv = P < p.NP;
        const uint32_t n = fd_div(P, p.fd_hw);
        const int pp = (int)(P - n * (uint32_t)p.HW);
#pragma unroll
        for (int i = 0; i < RC; ++i) {
            const int c = cb * TC + r0 + 16 * i;
            float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pv && c < p.Cg) v4 = *reinterpret_cast<const float4*>(p.x + ((int64_t)n * p.Cin_total + chan_phys(p.in_map, g * p.Cg + c)) * HW + pp);
            float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float r = v[e] - mn_bf16_head(v[e]);
                if (term == 2) r = r - mn_bf16_head(r);
                v[e] = r;
            }
            *reinterpret_cast<u32x2*>(xq + (r0 + 16 * i) * WG_LDP + qd * 4) = u32x2{mn_pack_bf16x2(v[0], v[1]), mn_pack_bf16x2(v[2], v[3])};
        }
    };

    int chunk = z, par = 0;
    if (XMODE == MN_ACTQ_NONE && tid < 2) xflag[tid] = 0;
    if (chunk < p.nchunks) fetch(chunk);
    for (; chunk < p.nchunks; chunk += p.Z, par ^= 1) {
        __syncthreads();          // previous slab fully consumed
        if (XMODE == MN_ACTQ_NONE) {
            unsigned inx = 0u;
#pragma unroll
            for (int i = 0; i < RC; ++i)
                inx |= (mn_f2u(rx[i].x - mn_bf16_head(rx[i].x)) | mn_f2u(rx[i].y - mn_bf16_head(rx[i].y)) |
                        mn_f2u(rx[i].z - mn_bf16_head(rx[i].z)) | mn_f2u(rx[i].w - mn_bf16_head(rx[i].w))) << 1;
            if (inx) xflag[par] = 1;
        }
        commit();
        __syncthreads();
        const int inexact = (XMODE == MN_ACTQ_NONE) ? xflag[par] : 0;
        if (XMODE == MN_ACTQ_NONE && tid == 0) xflag[par ^ 1] = 0;
        if (chunk + p.Z < p.nchunks) fetch(chunk + p.Z);     // in flight during the MFMA phase
        contract();
        if (inexact) {            // block-uniform
            for (int term = 1; term <= 2; ++term) {
                __syncthreads();
                restage_x_term(chunk, term);
                __syncthreads();
                contract();
            }
        }
    }
    // partial tile: lane (j, kg) holds rows m = 4kg + r, column c = j
#pragma unroll
    for (int mi = 0; mi < MW; ++mi)
#pragma unroll
        for (int ci = 0; ci < CW; ++ci) {
            const int mrow = mb * TM + (wm * MW + mi) * 16 + kg * 4;
            const int ccol = cb * TC + (wc * CW + ci) * 16 + j;
            float* dst = p.part + (((int64_t)z * p.G + g) * p.Mgw + mrow) * p.Cgw + ccol;
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(int64_t)r * p.Cgw] = acc[mi][ci][r];
        }
    if (p.want_db && cb == 0) {
#pragma unroll
        for (int i = 0; i < RM; ++i) {
            float v = dbacc[i];
            v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
            if (qd == 0) p.dbpart[((int64_t)z * p.G + g) * p.Mgw + mb * TM + r0 + 16 * i] = v;
        }
    }
}
// fixed-order reduction of the Z partial tiles (fp64 accumulate, one rounding), times the activation scale.
// Eight lanes share an output: lane s sums partials s, s+8, ... and the eight sums are combined by a butterfly -- the same
// order on every run (deterministic), 8x the parallelism of one thread per output.
__global__ __launch_bounds__(256) void k_pw_wgrad_reduce(const float* __restrict__ part, const float* __restrict__ dbpart, float* __restrict__ dw,
                                                         float* __restrict__ db, int Z, int G, int Mg, int Cg, int Mgw, int Cgw,
                                                         float ascale, const float* __restrict__ qp, const float* __restrict__ rowdiv) {
    const float as = qp ? qp[0] : ascale;
    const int64_t nw = (int64_t)G * Mg * Cg, total = nw + (db ? (int64_t)G * Mg : 0);
    const int sub = threadIdx.x & 7;
    const int64_t nslots = ((total + 31) / 32) * 32;           // whole waves take part in the shuffles
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3; i < nslots; i += ((int64_t)gridDim.x * blockDim.x) >> 3) {
        double s = 0.0;
        if (i < nw) {
            const int c = (int)(i % Cg);
            const int64_t o = i / Cg;
            const int g = (int)(o / Mg), m = (int)(o % Mg);
            for (int z = sub; z < Z; z += 8) s += (double)part[(((int64_t)z * G + g) * Mgw + m) * Cgw + c];
        } else if (i < total) {
            const int64_t o = i - nw;
            const int g = (int)(o / Mg), m = (int)(o % Mg);
            for (int z = sub; z < Z; z += 8) s += (double)dbpart[((int64_t)z * G + g) * Mgw + m];
        }
        s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 1, 64);
        if (sub == 0) {
            if (rowdiv) {          // the partials were formed on a row-scaled operand (k_pwb): one fp64 division by the row's scale (0 -> 1: an unscaled row)
                const int64_t o = i < nw ? i / Cg : i - nw;
                const float rd = i < total ? rowdiv[(o / Mg) * Mgw + (o % Mg)] : 1.f;
                s /= (double)(rd == 0.f ? 1.f : rd);
            }
            if (i < nw) dw[i] = (float)s * as;
            else if (i < total) db[i - nw] = (float)s;
        }
    }
}

// The same for many partial tiles (Z >= 16): a block owns 64 consecutive floats of the padded [G][Mgw][Cgw] tile; its 16 groups of 16 lanes each sum a
// z subset (z = q, q + 16, ...) with 16-byte loads -- 256 contiguous bytes per group and z, against 32-byte pieces in the kernel above (10-12 us for
// 17 MB of partials) -- and the 16 sums are combined through LDS in a fixed order.
__global__ __launch_bounds__(256) void k_pw_wgrad_reduce_v(const float* __restrict__ part, const float* __restrict__ dbpart, float* __restrict__ dw,
                                                           float* __restrict__ db, int Z, int G, int Mg, int Cg, int Mgw, int Cgw,
                                                           float ascale, const float* __restrict__ qp, int nblk_w, const float* __restrict__ rowdiv) {
    __shared__ double sm[16][16][4];
    const float as = qp ? qp[0] : ascale;
    const int q = threadIdx.x >> 4, l = threadIdx.x & 15;
    if ((int)blockIdx.x < nblk_w) {
        const int64_t tile = (int64_t)G * Mgw * Cgw;
        const int64_t e0 = (int64_t)blockIdx.x * 64 + 4 * l;                  // first of this lane's four floats in the padded tile
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll 4
        for (int z = q; z < Z; z += 16) {
            const float4 v = *reinterpret_cast<const float4*>(part + (int64_t)z * tile + e0);
            s0 += (double)v.x; s1 += (double)v.y; s2 += (double)v.z; s3 += (double)v.w;
        }
        sm[q][l][0] = s0; sm[q][l][1] = s1; sm[q][l][2] = s2; sm[q][l][3] = s3;
        __syncthreads();
        if (threadIdx.x < 64) {
            const int ll = threadIdx.x >> 2, e = threadIdx.x & 3;
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += sm[k][ll][e];
            const int64_t ei = (int64_t)blockIdx.x * 64 + 4 * ll + e;
            const int c = (int)(ei % Cgw);
            const int64_t o = ei / Cgw;
            const int m = (int)(o % Mgw), g = (int)(o / Mgw);
            if (rowdiv && m < Mg) { const float rd = rowdiv[g * Mgw + m]; t /= (double)(rd == 0.f ? 1.f : rd); }
            if (m < Mg && c < Cg) dw[((int64_t)g * Mg + m) * Cg + c] = (float)t * as;
        }
    } else if (db) {
        // bias gradient: the 8-lane scheme of the kernel above
        const int sub = threadIdx.x & 7;
        const int64_t total = (int64_t)G * Mg, nslots = ((total + 31) / 32) * 32;
        const int64_t nthr = (int64_t)(gridDim.x - nblk_w) * blockDim.x;
        for (int64_t i = ((int64_t)(blockIdx.x - nblk_w) * blockDim.x + threadIdx.x) >> 3; i < nslots; i += nthr >> 3) {
            double sd = 0.0;
            if (i < total) {
                const int g = (int)(i / Mg), m = (int)(i % Mg);
                for (int z = sub; z < Z; z += 8) sd += (double)dbpart[((int64_t)z * G + g) * Mgw + m];
            }
            sd += __shfl_xor(sd, 4, 64); sd += __shfl_xor(sd, 2, 64); sd += __shfl_xor(sd, 1, 64);
            if (rowdiv && sub == 0 && i < total) { const float rd = rowdiv[(i / Mg) * Mgw + (i % Mg)]; sd /= (double)(rd == 0.f ? 1.f : rd); }
            if (sub == 0 && i < total) db[i] = (float)sd;
        }
    }
}

static void qg_launch_wgrad_reduce_(const float* part, const float* dbpart, float* dw, float* db, int Z, int G, int Mg, int Cg, int Mgw, int Cgw,
                                    float ascale, const float* qp, const float* rowdiv, hipStream_t s);
void qg_launch_wgrad_reduce(const float* part, const float* dbpart, float* dw, float* db, int Z, int G, int Mg, int Cg, int Mgw, int Cgw,
                            float ascale, const float* qp, hipStream_t s) {
    qg_launch_wgrad_reduce_(part, dbpart, dw, db, Z, G, Mg, Cg, Mgw, Cgw, ascale, qp, nullptr, s);
}
// ... of partials formed on an operand whose row m of group g was multiplied by rowdiv[g * Mgw + m] (k_pwb, qgemm_pwb.hip): dw, db = sum / rowdiv
void qg_launch_wgrad_reduce_div(const float* part, const float* dbpart, float* dw, float* db, int Z, int G, int Mg, int Cg, int Mgw, int Cgw, float ascale,
                                const float* rowdiv, hipStream_t s) {
    qg_launch_wgrad_reduce_(part, dbpart, dw, db, Z, G, Mg, Cg, Mgw, Cgw, ascale, nullptr, rowdiv, s);
}
static void qg_launch_wgrad_reduce_(const float* part, const float* dbpart, float* dw, float* db, int Z, int G, int Mg, int Cg, int Mgw, int Cgw,
                                    float ascale, const float* qp, const float* rowdiv, hipStream_t s) {
    const int64_t tile = (int64_t)G * Mgw * Cgw;
    if (Z >= 16 && tile % 64 == 0 && tile / 64 < (1 << 22) && !(((uintptr_t)part) & 15)) {
        const int nblk_w = (int)(tile / 64);
        const int nblk_b = db ? mn_grid_for((int64_t)G * Mg * 8, 256, 64) : 0;
        hipLaunchKernelGGL(k_pw_wgrad_reduce_v, dim3(nblk_w + nblk_b), dim3(256), 0, s, part, dbpart, dw, db, Z, G, Mg, Cg, Mgw, Cgw, ascale, qp, nblk_w, rowdiv);
        return;
    }
    const int64_t total = (int64_t)G * Mg * Cg + (db ? (int64_t)G * Mg : 0);
    hipLaunchKernelGGL(k_pw_wgrad_reduce, dim3(mn_grid_for(total * 8, 256, 4096)), dim3(256), 0, s, part, dbpart, dw, db, Z, G, Mg, Cg, Mgw, Cgw, ascale, qp, rowdiv);
}

// ------------------------------------------------------------------------------------------------
// host side

static int pw_geom_ok(const mn_conv_geom* g) {
    if (g->KH != 1 || g->KW != 1 || g->stride_h != 1 || g->stride_w != 1 || g->pad_h != 0 || g->pad_w != 0) return 0;
    const int64_t HW = (int64_t)g->H * g->W, NP = (int64_t)g->N * HW;
    if (HW % 4) return 0;
    if (NP * HW >= ((int64_t)1 << 32) || NP + 256 >= ((int64_t)1 << 31)) return 0;   // FastDiv range
    return 1;
}
struct PwPlan {
    PwParams p;
    PackParams pk;
    int NT;
    size_t lds;
    int grid, pack_grid;
    int64_t off_codes, off_scale, ws_bytes;
};
static int plan_pw(const mn_conv_geom* g, int which, int xmode, PwPlan* pl) {
    if (!pw_geom_ok(g)) return 0;
    const int Cg = g->C / g->groups, Mg = g->O / g->groups;
    PwParams& p = pl->p;
    p.N = g->N; p.HW = g->H * g->W; p.G = g->groups;
    p.NP = (uint32_t)((int64_t)g->N * p.HW);
    if (g->in_shuffle > 1 && g->C % g->in_shuffle) return 0;
    if (which == 0) { p.Cin_total = g->C; p.Cout_total = g->O; p.Kc = Cg; p.Mr = Mg; p.in_map = make_chanmap(g->in_shuffle, g->C); p.out_map = make_chanmap(0, 0); }
    else { p.Cin_total = g->O; p.Cout_total = g->C; p.Kc = Mg; p.Mr = Cg; p.in_map = make_chanmap(0, 0); p.out_map = make_chanmap(g->in_shuffle, g->C); }
    p.Kp = qg_roundup(p.Kc, 32);
    p.KS = p.Kp / 32;
    int NT = p.Mr > 64 ? 8 : (p.Mr > 32 ? 4 : (p.Mr > 16 ? 2 : 1));
    (void)xmode;
    if (NT > 4) NT = 4;   // 64 accumulator VGPRs: two waves per SIMD without spills (NT = 8 spills at 256 VGPRs); the second m-block
                          // of a 128-channel group re-reads x through the XCD's L2 (same-XCD block placement, see k_pw)
    size_t lds;
    for (;;) {
        lds = (size_t)16 * NT * (p.Kp + 8) * 2 + ((size_t)2 * 16 * NT + p.Kp) * 4;
        if (lds <= QG_LDS_CAP) break;
        if (NT == 1) return 0;
        NT /= 2;
    }
    pl->NT = NT; pl->lds = lds;
    p.num_mblk = (p.Mr + 16 * NT - 1) / (16 * NT);
    p.Mpad = p.num_mblk * 16 * NT;
    p.nchunks = (int)((p.NP + 63) / 64);
    int CB = (p.nchunks + 3) / 4;
    int capb = 512;           // one round of 2 blocks per CU (against 1024: -2 ... -3 % on the DoReFa layers)
    const int cap = capb / (p.G * p.num_mblk) > 0 ? capb / (p.G * p.num_mblk) : 1;
    if (CB > cap) CB = cap;
    p.CB = CB;
    p.fd_hw = make_fastdiv((uint32_t)p.HW);
    p.fd_ks = make_fastdiv((uint32_t)p.KS);
    const int64_t nb = (int64_t)qg_roundup(p.G * CB, 8) * p.num_mblk;
    if (nb > 0x7fffffff) return 0;
    pl->grid = (int)nb;
    // workspace: codes [G][Mpad][Kp] u16, then per-channel scale floats
    pl->off_codes = 0;
    const int64_t code_bytes = (int64_t)p.G * p.Mpad * p.Kp * 2;
    pl->off_scale = (code_bytes + 255) / 256 * 256;
    const int64_t nscale = which == 0 ? (int64_t)p.G * p.Mpad : (int64_t)p.G * p.Kp;
    pl->ws_bytes = pl->off_scale + nscale * 4;
    PackParams& k = pl->pk;
    k.G = g->groups; k.Mg = Mg; k.Cg = Cg; k.T = 1; k.KW = 1; k.transpose = which == 1;
    k.Mpad = which == 0 ? p.Mpad : 0; k.Cgp = which == 0 ? p.Kp : 0;
    k.Cpad = which == 1 ? p.Mpad : 0; k.Mgp = which == 1 ? p.Kp : 0;
    pl->pack_grid = k.G * (which == 0 ? k.Mpad : k.Mgp);
    return 1;
}

struct WgPlan {
    PwWgParams p;
    int cfg;          // 0: 128x128  1: 16x128  2: 128x16  3: 64x64
    size_t lds;
    int grid;
    int64_t off_db, ws_bytes;
};
static int plan_pw_wgrad(const mn_conv_geom* g, WgPlan* pl) {
    if (!pw_geom_ok(g)) return 0;
    const int Cg = g->C / g->groups, Mg = g->O / g->groups;
    PwWgParams& p = pl->p;
    if (g->in_shuffle > 1 && g->C % g->in_shuffle) return 0;
    p.in_map = make_chanmap(g->in_shuffle, g->C);
    p.N = g->N; p.HW = g->H * g->W; p.G = g->groups; p.Cin_total = g->C; p.Cout_total = g->O; p.Cg = Cg; p.Mg = Mg;
    p.NP = (uint32_t)((int64_t)g->N * p.HW);
    int TM, TC;
    if (Mg <= 16) { pl->cfg = 1; TM = 16; TC = 128; }
    else if (Cg <= 16) { pl->cfg = 2; TM = 128; TC = 16; }
    else if (Mg <= 64 && Cg <= 64) { pl->cfg = 3; TM = 64; TC = 64; }
    else { pl->cfg = 0; TM = 128; TC = 128; }
    pl->lds = (size_t)(3 * TM + TC) * WG_LDP * 2 + 16;   // + the two exactness flags of the real-x path
    p.nmb = (Mg + TM - 1) / TM; p.ncb = (Cg + TC - 1) / TC;
    p.Mgw = p.nmb * TM; p.Cgw = p.ncb * TC;
    p.nchunks = (int)((p.NP + 63) / 64);
    const int base = p.G * p.nmb * p.ncb;
    int Z = 512 / base;
    if (Z > p.nchunks / 4) Z = p.nchunks / 4;
    if (Z < 1) Z = 1;
    p.Z = Z;
    p.fd_hw = make_fastdiv((uint32_t)p.HW);
    const int64_t nb = (int64_t)base * Z;
    if (nb > 0x7fffffff) return 0;
    pl->grid = (int)nb;
    const int64_t part_bytes = (int64_t)Z * p.G * p.Mgw * p.Cgw * 4;
    pl->off_db = (part_bytes + 255) / 256 * 256;
    pl->ws_bytes = pl->off_db + (int64_t)Z * p.G * p.Mgw * 4;
    return 1;
}

struct PwdPlan { PwdParams p; PackParams pk; int NT; size_t lds; int grid, pack_grid; int64_t off_scale, ws_bytes; };
static int plan_pwd(const mn_conv_geom* g, PwdPlan* pl) {
    if (!pw_geom_ok(g)) return 0;
    const int Cg = g->C / g->groups, Mg = g->O / g->groups;
    const int64_t NP = (int64_t)g->N * g->H * g->W;
    if (4 * NP * (g->C > g->O ? g->C : g->O) >= ((int64_t)1 << 32)) return 0;        // 32-bit element offsets
    if (g->in_shuffle > 1 && g->C % g->in_shuffle) return 0;
    PwdParams& p = pl->p;
    p.N = g->N; p.HW = g->H * g->W; p.G = g->groups; p.NP = (uint32_t)NP;
    p.Cin_total = g->O; p.Cout_total = g->C; p.Kc = Mg; p.Mr = Cg;
    p.out_map = make_chanmap(g->in_shuffle, g->C);
    p.Kp = qg_roundup(Mg, 32); p.KS = p.Kp / 32;
    if (p.KS < 1 || p.KS > 4) return 0;
    int NT = Cg > 32 ? 4 : (Cg > 16 ? 2 : 1);
    pl->NT = NT;
    const int MB = 16 * NT;
    p.num_mblk = (Cg + MB - 1) / MB; p.Mpad = p.num_mblk * MB;
    pl->lds = (size_t)MB * (p.Kp + 8) * 2 + (size_t)2 * p.Kp * 4 + (size_t)MB * 4 + (size_t)5 * p.Kp * 4;
    p.nchunks = (int)((NP + 63) / 64);
    int CB = (p.nchunks + 3) / 4;
    int capb = 512;           // one round of 2 blocks per CU (against 1024: -2 ... -5 %)
    const int cap = capb / (p.G * p.num_mblk) > 0 ? capb / (p.G * p.num_mblk) : 1;
    if (CB > cap) CB = cap;
    p.CB = CB;
    p.fd_hw = make_fastdiv((uint32_t)p.HW);
    const int64_t nb = (int64_t)qg_roundup(p.G * CB, 8) * p.num_mblk;
    if (nb > 0x7fffffff) return 0;
    pl->grid = (int)nb;
    const int64_t code_bytes = (int64_t)p.G * p.Mpad * p.Kp * 2;
    pl->off_scale = (code_bytes + 255) / 256 * 256;
    pl->ws_bytes = pl->off_scale + (int64_t)p.G * p.Kp * 4;
    PackParams& k = pl->pk;
    k.G = g->groups; k.Mg = Mg; k.Cg = Cg; k.T = 1; k.KW = 1; k.transpose = 1;
    k.Mpad = 0; k.Cgp = 0; k.Cpad = p.Mpad; k.Mgp = p.Kp;
    pl->pack_grid = k.G * k.Mgp;
    return 1;
}
// the transposed weight codes + contraction-channel scales of this call: the step's pre-packed image (mn_wq.packed_bwd, written by mn_qg_pack_multi in the layout
// [codes | scales at off_scale] of this very plan) or packed here into the call's workspace
static void pwd_codes(PwdPlan& pd, const mn_wq* wq, const float* w, void* ws, hipStream_t s) {
    if (wq->packed_bwd && mn_use_packed()) {
        pd.pk.codes = (uint16_t*)const_cast<void*>(wq->packed_bwd);
        pd.pk.scale_out = (float*)((char*)const_cast<void*>(wq->packed_bwd) + pd.off_scale);
        return;
    }
    fill_pack(pd.pk, wq, w, ws, 0, pd.off_scale);
    qg_launch_pack(pd.pk, pd.pack_grid, s);
}
int pwd_pack_plan(const mn_conv_geom* g, PackParams* pk, int* grid, int64_t* off_scale, int64_t* bytes) {
    PwdPlan pd;
    if (!plan_pwd(g, &pd)) return 0;
    *pk = pd.pk; *grid = pd.pack_grid; *off_scale = pd.off_scale; *bytes = pd.ws_bytes;
    return 1;
}
template <int NT, int BNH>
static void launch_pwd2(const PwdPlan& pl, hipStream_t s) {
    switch (pl.p.KS) {
        case 1: hipLaunchKernelGGL((k_pwd<NT, 1, BNH>), dim3(pl.grid), dim3(256), pl.lds, s, pl.p); break;
        case 2: hipLaunchKernelGGL((k_pwd<NT, 2, BNH>), dim3(pl.grid), dim3(256), pl.lds, s, pl.p); break;
        case 3: hipLaunchKernelGGL((k_pwd<NT, 3, BNH>), dim3(pl.grid), dim3(256), pl.lds, s, pl.p); break;
        default: hipLaunchKernelGGL((k_pwd<NT, 4, BNH>), dim3(pl.grid), dim3(256), pl.lds, s, pl.p); break;
    }
}
template <int NT>
static void launch_pwd(const PwdPlan& pl, hipStream_t s) {
    if (pl.p.h && pl.p.own) launch_pwd2<NT, 2>(pl, s); else if (pl.p.h) launch_pwd2<NT, 1>(pl, s); else launch_pwd2<NT, 0>(pl, s);
}
int pwd_supported(const mn_conv_geom* g, const mn_wq* wq) { PwdPlan pd; return wq_codeable(wq) && plan_pwd(g, &pd); }
int64_t pwd_ws_bytes(const mn_conv_geom* g) { PwdPlan pd; return plan_pwd(g, &pd) ? pd.ws_bytes : 0; }
// backward-data whose incoming gradient is the BatchNorm+sign backward of (da, h): formed inside the kernel
int pwd_bwd_data_bnh(const mn_conv_geom* g, const mn_wq* wq, const float* da, const uint8_t* h, const float* chan, const float* sums, int training,
                     const float* w, float* dx, void* ws, int64_t ws_bytes, hipStream_t s, const int8_t* own) {
    PwdPlan pd;
    if (!wq_codeable(wq) || !plan_pwd(g, &pd) || !aligned16(dx) || (((uintptr_t)h) & 3) || (own ? ((((uintptr_t)da) & 7) || (((uintptr_t)own) & 3) || (g->H & 1) || (g->W & 3)) : !aligned16(da)))
        MN_FAIL(MN_ENOTSUP, "mn_conv2d_bwd_data_bnh: geometry / quantizer combination not covered");
    if (!ws || ws_bytes < pd.ws_bytes || !aligned16(ws)) MN_FAIL(MN_ENOSPC, "mn_conv2d_bwd_data_bnh: workspace too small");
    pwd_codes(pd, wq, w, ws, s);
    pd.p.gy = da; pd.p.dx = dx; pd.p.wc = pd.pk.codes; pd.p.kscale = pd.pk.scale_out;
    pd.p.h = h; pd.p.chan = chan; pd.p.sums = sums; pd.p.training = training; pd.p.n_f = (float)g->N * (float)(g->H * g->W);
    pd.p.own = (const char*)own; pd.p.W = (int)g->W; pd.p.fd_w = make_fastdiv((uint32_t)g->W);
    mn_set_last_kernel(own ? "k_pwd<%d, %d, 2>" : "k_pwd<%d, %d, 1>", pd.NT, pd.p.KS);
    { const double nx = (double)g->N * g->C * g->H * g->W, ny = (double)g->N * g->O * g->H * g->W; mn_prof_bytes((own ? 3.0 : 5.0) * ny + 4.0 * nx); }
    mn_prof_begin(s);
    if (pd.NT == 4) launch_pwd<4>(pd, s); else if (pd.NT == 2) launch_pwd<2>(pd, s); else launch_pwd<1>(pd, s);
    mn_prof_end(s);
    MN_CHECK_LAUNCH("mn_conv2d_bwd_data_bnh");
    return MN_OK;
}
int qg_supported(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, int which) {
    if (!pw_geom_ok(g)) return kk_supported(g, aq, wq, which);
    if (which == 0) { PwPlan pl; return wq_codeable(wq) && aq_codeable(aq, 0) && plan_pw(g, 0, aq ? aq->mode : MN_ACTQ_NONE, &pl); }
    if (which == 1) { PwPlan pl; return wq_codeable(wq) && plan_pw(g, 1, MN_ACTQ_NONE, &pl); }
    if (which == 2 && aq && aq->mode == MN_ACTQ_CODE8) { WgPlan pl; return aq->bits >= 2 && aq->bits <= 8 && (pws_wgrad_code8_supported(g) || plan_pw_wgrad(g, &pl)); }
    if (which == 2) { WgPlan pl; return aq_codeable(aq, 1) && plan_pw_wgrad(g, &pl); }
    return 0;
}
int64_t qg_ws_bytes(const mn_conv_geom* g, int which) {
    if (!pw_geom_ok(g)) return kk_ws_bytes(g, which);
    if (which == 0 || which == 1) {   // NT <= 4: the larger Mpad; forward: also the fused sign kernels' workspace
        PwPlan pl;
        PwdPlan pd;
        const int64_t a = plan_pw(g, which, MN_ACTQ_NONE, &pl) ? pl.ws_bytes : 0;
        const int64_t b = which == 0 ? pws_ws_bytes(g) : (plan_pwd(g, &pd) ? pd.ws_bytes : 0);
        return a > b ? a : b;
    }
    if (which == 2) {
        WgPlan pl;
        const int64_t a = plan_pw_wgrad(g, &pl) ? pl.ws_bytes : 0, b = pws_wgrad_ws_bytes(g);
        return a > b ? a : b;
    }
    return 0;
}

static thread_local int g_pw_relu = 0;          // set by qg_fwd_act around its call of qg_fwd
static thread_local float* g_pw_mm = nullptr;
template <int NT>
static void launch_pw(const PwPlan& pl, int xmode, hipStream_t s) {
    if (xmode == MN_ACTQ_DOREFA) hipLaunchKernelGGL((k_pw<NT, MN_ACTQ_DOREFA>), dim3(pl.grid), dim3(256), pl.lds, s, pl.p);
    else if (xmode == MN_ACTQ_IAO) hipLaunchKernelGGL((k_pw<NT, MN_ACTQ_IAO>), dim3(pl.grid), dim3(256), pl.lds, s, pl.p);
    else if (xmode == MN_ACTQ_SIGN8) hipLaunchKernelGGL((k_pw<NT, MN_ACTQ_SIGN8>), dim3(pl.grid), dim3(256), pl.lds, s, pl.p);
    else hipLaunchKernelGGL((k_pw<NT, MN_ACTQ_NONE>), dim3(pl.grid), dim3(256), pl.lds, s, pl.p);
}
static int run_pw(PwPlan& pl, int xmode, hipStream_t s, const char* what) {
    mn_set_last_kernel("k_pw<%d, %d>", pl.NT, xmode);
    mn_prof_begin(s);
    switch (pl.NT) {
        case 1: launch_pw<1>(pl, xmode, s); break;
        case 2: launch_pw<2>(pl, xmode, s); break;
        case 4: launch_pw<4>(pl, xmode, s); break;
        default: MN_FAIL(MN_EINVAL, "%s: bad NT", what);
    }
    mn_prof_end(s);
    MN_CHECK_LAUNCH(what);
    return MN_OK;
}
int qg_fwd(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, const float* x, const float* w, const float* bias, float* y,
           void* ws, int64_t ws_bytes, hipStream_t s) {
    if (aq && aq->mode == MN_ACTQ_CODE8) MN_FAIL(MN_ENOTSUP, "mn_conv2d_fwd: activation codes are read by mn_qconv_bnq_fwd_stash only");
    if (!pw_geom_ok(g)) return kk_fwd(g, aq, wq, x, w, bias, y, ws, ws_bytes, s);
    if (aq && aq->mode == MN_ACTQ_SIGN8 && pws_supported(g, wq) && ws_bytes >= pws_ws_bytes(g))     // sign codes: the prefetching kernel
        return pws_fwd(g, wq, (const int8_t*)x, w, bias, y, ws, ws_bytes, s);
    PwPlan pl;
    if (!wq_codeable(wq) || !aq_codeable(aq, 0) || !plan_pw(g, 0, aq ? aq->mode : MN_ACTQ_NONE, &pl) || !aligned16(x) || !aligned16(y))
        MN_FAIL(MN_ENOTSUP, "mn_conv2d_fwd(qgemm): geometry / quantizer combination not covered");
    if (!ws || ws_bytes < pl.ws_bytes || !aligned16(ws)) MN_FAIL(MN_ENOSPC, "mn_conv2d_fwd(qgemm): workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)pl.ws_bytes);
    Pro pro;
    int rc = make_pro(aq, &pro, 0, "mn_conv2d_fwd(qgemm)");
    if (rc) return rc;
    fill_pack(pl.pk, wq, w, ws, pl.off_codes, pl.off_scale);
    qg_launch_pack(pl.pk, pl.pack_grid, s);
    PwParams& p = pl.p;
    p.x = x; p.y = y; p.wc = pl.pk.codes; p.rowscale = pl.pk.scale_out; p.kscale = nullptr; p.bias = bias; p.aux = nullptr;
    p.pro = pro; p.ste = pro; p.epi = QG_EPI_SCALE_BIAS;
    p.ascale = pro.mode == MN_ACTQ_DOREFA ? pro.s : 1.f;
    p.relu = g_pw_relu; p.mm = g_pw_mm;
    return run_pw(pl, pro.mode, s, "mn_conv2d_fwd(qgemm)");
}
// forward with the ReLU behind the conv and the per-wave (min, max) of the result in the epilogue (pointwise code-domain layers only)
int qg_fwd_act_mm_count(const mn_conv_geom* g) {
    PwPlan pl;
    if (!pw_geom_ok(g) || !plan_pw(g, 0, MN_ACTQ_NONE, &pl)) return 0;
    return 4 * pl.grid;
}
int qg_fwd_act(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, const float* x, const float* w, const float* bias, float* y, int relu, float* mm,
               void* ws, int64_t ws_bytes, hipStream_t s) {
    if (!pw_geom_ok(g) || (aq && (aq->mode == MN_ACTQ_SIGN8 || aq->mode == MN_ACTQ_CODE8)))
        MN_FAIL(MN_ENOTSUP, "mn_conv2d_fwd_act: pointwise code-domain layers with fp32 input only");
    g_pw_relu = relu; g_pw_mm = mm;
    const int rc = qg_fwd(g, aq, wq, x, w, bias, y, ws, ws_bytes, s);
    g_pw_relu = 0; g_pw_mm = nullptr;
    return rc;
}

int qg_bwd_data(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, const float* gy, const float* w, const float* x, float* dx,
                void* ws, int64_t ws_bytes, hipStream_t s) {
    if (!pw_geom_ok(g)) return kk_bwd_data(g, aq, wq, gy, w, x, dx, ws, ws_bytes, s);
    PwPlan pl;
    if (!wq_codeable(wq) || !plan_pw(g, 1, MN_ACTQ_NONE, &pl) || !aligned16(gy) || !aligned16(dx))
        MN_FAIL(MN_ENOTSUP, "mn_conv2d_bwd_data(qgemm): geometry / quantizer combination not covered");
    if (!ws || ws_bytes < pl.ws_bytes || !aligned16(ws)) MN_FAIL(MN_ENOSPC, "mn_conv2d_bwd_data(qgemm): workspace too small");
    Pro ste;
    int rc = make_pro(aq, &ste, 1, "mn_conv2d_bwd_data(qgemm)");
    if (rc) return rc;
    if (ste.mode == MN_ACTQ_SIGN8 || ste.mode == MN_ACTQ_CODE8) ste.mode = MN_ACTQ_NONE;      // the clip-STE lives in mn_bnsign_bwd / mn_qa_bwd_*
    if (ste.mode != MN_ACTQ_NONE && (!x || !aligned16(x))) MN_FAIL(MN_EINVAL, "mn_conv2d_bwd_data(qgemm): x required (16 B aligned) for the clip-STE epilogue");
    if (ste.mode == MN_ACTQ_NONE) {      // no clip-STE epilogue: the prefetching kernel
        PwdPlan pd;
        if (plan_pwd(g, &pd) && ws_bytes >= pd.ws_bytes) {
            pwd_codes(pd, wq, w, ws, s);
            pd.p.gy = gy; pd.p.dx = dx; pd.p.wc = pd.pk.codes; pd.p.kscale = pd.pk.scale_out;
            pd.p.h = nullptr; pd.p.own = nullptr; pd.p.chan = nullptr; pd.p.sums = nullptr; pd.p.training = 0; pd.p.n_f = 1.f;
            mn_set_last_kernel("k_pwd<%d, %d, 0>", pd.NT, pd.p.KS);
            mn_prof_begin(s);
            if (pd.NT == 4) launch_pwd<4>(pd, s); else if (pd.NT == 2) launch_pwd<2>(pd, s); else launch_pwd<1>(pd, s);
            mn_prof_end(s);
            MN_CHECK_LAUNCH("mn_conv2d_bwd_data(qgemm)");
            return MN_OK;
        }
    }
    fill_pack(pl.pk, wq, w, ws, pl.off_codes, pl.off_scale);
    qg_launch_pack(pl.pk, pl.pack_grid, s);
    Pro none; none.mode = MN_ACTQ_NONE; none.s = 1.f; none.qmin = none.qmax = 0.f; none.qp = nullptr;
    PwParams& p = pl.p;
    p.x = gy; p.y = dx; p.wc = pl.pk.codes; p.rowscale = nullptr; p.kscale = pl.pk.scale_out; p.bias = nullptr; p.aux = x;
    p.pro = none; p.ste = ste; p.epi = ste.mode == MN_ACTQ_NONE ? QG_EPI_PLAIN : QG_EPI_STE; p.ascale = 1.f;
    p.relu = 0; p.mm = nullptr;
    return run_pw(pl, MN_ACTQ_NONE, s, "mn_conv2d_bwd_data(qgemm)");
}

template <int MW, int CW, int WGC>
static void launch_wg(const WgPlan& pl, int xmode, hipStream_t s) {
    if (xmode == MN_ACTQ_DOREFA) {
        raise_lds_limit((const void*)k_pw_wgrad<MW, CW, WGC, MN_ACTQ_DOREFA>, pl.lds);
        hipLaunchKernelGGL((k_pw_wgrad<MW, CW, WGC, MN_ACTQ_DOREFA>), dim3(pl.grid), dim3(256), pl.lds, s, pl.p);
    } else if (xmode == MN_ACTQ_IAO) {
        raise_lds_limit((const void*)k_pw_wgrad<MW, CW, WGC, MN_ACTQ_IAO>, pl.lds);
        hipLaunchKernelGGL((k_pw_wgrad<MW, CW, WGC, MN_ACTQ_IAO>), dim3(pl.grid), dim3(256), pl.lds, s, pl.p);
    } else if (xmode == MN_ACTQ_SIGN8) {
        raise_lds_limit((const void*)k_pw_wgrad<MW, CW, WGC, MN_ACTQ_SIGN8>, pl.lds);
        hipLaunchKernelGGL((k_pw_wgrad<MW, CW, WGC, MN_ACTQ_SIGN8>), dim3(pl.grid), dim3(256), pl.lds, s, pl.p);
    } else if (xmode == MN_ACTQ_CODE8) {
        raise_lds_limit((const void*)k_pw_wgrad<MW, CW, WGC, MN_ACTQ_CODE8>, pl.lds);
        hipLaunchKernelGGL((k_pw_wgrad<MW, CW, WGC, MN_ACTQ_CODE8>), dim3(pl.grid), dim3(256), pl.lds, s, pl.p);
    } else {
        raise_lds_limit((const void*)k_pw_wgrad<MW, CW, WGC, MN_ACTQ_NONE>, pl.lds);
        hipLaunchKernelGGL((k_pw_wgrad<MW, CW, WGC, MN_ACTQ_NONE>), dim3(pl.grid), dim3(256), pl.lds, s, pl.p);
    }
}
int qg_bwd_weight(const mn_conv_geom* g, const mn_actq* aq, const float* gy, const float* x, float* dw, float* dbias, void* ws,
                  int64_t ws_bytes, hipStream_t s) {
    if (!pw_geom_ok(g)) return kk_bwd_weight(g, aq, gy, x, dw, dbias, ws, ws_bytes, s);
    if (aq && aq->mode == MN_ACTQ_SIGN8 && pws_wgrad_supported(g) && ws_bytes >= pws_wgrad_ws_bytes(g))
        return pws_bwd_weight(g, gy, (const int8_t*)x, dw, dbias, ws, ws_bytes, s);      // sign codes: fragments straight from global memory
    const int code8 = aq && aq->mode == MN_ACTQ_CODE8;
    if (code8) {        // k-bit activation codes: the LDS-staged kernel, or (small tiles: the classifier conv) the generic kernel reading bytes
        if (aq->bits < 2 || aq->bits > 8) MN_FAIL(MN_ENOTSUP, "mn_conv2d_bwd_weight(code8): 2 ... 8 bit codes");
        if (pws_wgrad_code8_supported(g) && ws_bytes >= pws_wgrad_ws_bytes(g))
            return pws_bwd_weight_code8(g, gy, (const uint8_t*)x, dorefa_scale(aq->bits), dw, dbias, ws, ws_bytes, s);
    }
    WgPlan pl;
    if ((!code8 && !aq_codeable(aq, 1)) || !plan_pw_wgrad(g, &pl) || !aligned16(gy) || (((uintptr_t)x) & (code8 ? 3 : 15)))
        MN_FAIL(MN_ENOTSUP, "mn_conv2d_bwd_weight(qgemm): geometry / quantizer combination not covered");
    if (!ws || ws_bytes < pl.ws_bytes || !aligned16(ws)) MN_FAIL(MN_ENOSPC, "mn_conv2d_bwd_weight(qgemm): workspace too small");
    Pro pro;
    int rc = make_pro(aq, &pro, 0, "mn_conv2d_bwd_weight(qgemm)");
    if (rc) return rc;
    PwWgParams& p = pl.p;
    p.gy = gy; p.x = x; p.part = (float*)ws; p.dbpart = (float*)((char*)ws + pl.off_db); p.pro = pro; p.want_db = dbias != nullptr;
    static const char* cfgname[4] = {"4, 4, 2", "1, 2, 4", "2, 1, 1", "2, 2, 2"};
    mn_set_last_kernel("k_pw_wgrad<%s, %d>", cfgname[pl.cfg], pro.mode);
    mn_prof_begin(s);
    switch (pl.cfg) {
        case 0: launch_wg<4, 4, 2>(pl, pro.mode, s); break;
        case 1: launch_wg<1, 2, 4>(pl, pro.mode, s); break;
        case 2: launch_wg<2, 1, 1>(pl, pro.mode, s); break;
        default: launch_wg<2, 2, 2>(pl, pro.mode, s); break;
    }
    mn_prof_end(s);
    const int64_t total = (int64_t)g->O * (g->C / g->groups) + (dbias ? g->O : 0);
    (void)total;
    qg_launch_wgrad_reduce(p.part, p.dbpart, dw, dbias, p.Z, p.G, p.Mg, p.Cg, p.Mgw, p.Cgw,
                           (pro.mode == MN_ACTQ_DOREFA || pro.mode == MN_ACTQ_CODE8) ? pro.s : 1.f,
                           pro.mode == MN_ACTQ_IAO ? pro.qp : (const float*)nullptr, s);
    MN_CHECK_LAUNCH("mn_conv2d_bwd_weight(qgemm)");
    return MN_OK;
}

// ------------------------------------------------------------------------------------------------ the pointwise packs of a whole net in one launch
int pws_pack_plan(const mn_conv_geom* g, PackParams* pk, int* grid, int64_t* off_scale, int64_t* bytes);          // qgemm_sign.hip
static int qg_packm_plan(const mn_conv_geom* g, int which, PackParams* pk, int* grid, int64_t* off_scale, int64_t* bytes) {
    if (!g || (which != 0 && which != 1)) return 0;
    return which == 0 ? pws_pack_plan(g, pk, grid, off_scale, bytes) : pwd_pack_plan(g, pk, grid, off_scale, bytes);
}
extern "C" int64_t mn_qg_packed_bytes(const mn_conv_geom* g, int which) {
    PackParams pk; int grid; int64_t off, bytes;
    if (!mn_use_packed() || !qg_packm_plan(g, which, &pk, &grid, &off, &bytes)) return 0;
    return (off + (int64_t)pk.G * (pk.transpose ? pk.Mgp : pk.Mpad) * 4 + 255) / 256 * 256;
}
extern "C" int mn_qg_pack_multi(int32_t count, const mn_conv_geom* const* g, const mn_wq* const* wq, const float* const* w, const int32_t* which, void* const* out,
                                mn_stream_t stream) {
    if (count <= 0) return MN_OK;
    if (!g || !wq || !w || !which || !out) MN_FAIL(MN_EINVAL, "mn_qg_pack_multi: null table");
    hipStream_t s = (hipStream_t)stream;
    for (int base = 0; base < count; base += QG_PACKM_MAX) {
        PackTable t;
        const int n = count - base < QG_PACKM_MAX ? count - base : QG_PACKM_MAX;
        int blk = 0;
        for (int i = 0; i < n; ++i) {
            int grid; int64_t off, bytes;
            if (!wq[base + i] || !wq_codeable(wq[base + i]) || !w[base + i] || !out[base + i] || !aligned16(out[base + i]) ||
                !qg_packm_plan(g[base + i], which[base + i], &t.e[i], &grid, &off, &bytes))
                MN_FAIL(MN_EINVAL, "mn_qg_pack_multi: entry %d is not a pointwise layer of the code kernels", base + i);
            fill_pack(t.e[i], wq[base + i], w[base + i], out[base + i], 0, off);
            t.blk0[i] = blk;
            blk += grid;
        }
        for (int i = n; i <= QG_PACKM_MAX; ++i) t.blk0[i] = blk;
        t.count = n;
        hipLaunchKernelGGL(k_qg_pack_multi, dim3(blk), dim3(64), 0, s, t);
    }
    MN_CHECK_LAUNCH("mn_qg_pack_multi");
    return MN_OK;
}
