// Pointwise convolution on packed sign activations with fused BatchNorm + BinaryActivation epilogues, for gfx950.
//
// The reference's W/A-binary block is  a_out = sign(bn(conv(a_in, Wq)))  (models/nin_gc.py:53-59 with the ReLU replaced by
// BinaryActivation, wbwtab/quantize.py:79-94,181-195).  Its input is +-1 and its weights are ternary/binary codes x alpha[o]
// -- so the convolution output y = alpha[o] * acc + bias, with acc an EXACT small integer (|acc| <= Cin/groups), is cheap
// to recompute from one byte per input element (int8 sign codes, MN_ACTQ_SIGN8) on the bf16 matrix cores, while writing y
// (fp32) and reading it back are the HBM passes that dominate the step.  This kernel never needs y in memory: one main
// loop (weight codes as A fragments in LDS, sign codes streamed straight into B fragments, v_mfma_f32_16x16x32_bf16 -- see
// qgemm_kernels.hip for the fragment mapping) ends in one of
//   PWS_Y          y = acc * alpha[o] + bias                                  -> fp32 store   (the plain convolution)
//   PWS_STATS      per-channel sum acc, sum acc^2 (exact integers)            -> partials     (BatchNorm batch statistics)
//   PWS_SIGN8      a = sign(bn(y)) = [acc*flip >= T[o]]                       -> int8 store   (the block's output)
//   PWS_BWD_PART   dz = da * [L[o] <= acc*flip <= U[o]]; sum dz, sum dz*zhat  -> partials     (dgamma, dbeta, BN backward sums)
//   PWS_BWD_APPLY  dy = gamma * invstd * (dz - sum_dz/n - zhat * sum_dzzhat/n) -> fp32 store  (what conv backward consumes)
// Forward of a block = PWS_STATS + PWS_SIGN8: reads |a_in| bytes twice, writes |a_out| bytes once; y is never stored.
// Backward of its BatchNorm+sign = PWS_BWD_PART + PWS_BWD_APPLY: reads da twice, writes dy once; y is recomputed.
//
// Integer-domain epilogues.  z(acc) = ((acc*alpha + bias) - mean) * invstd * gamma + beta, evaluated in fp32 exactly as the
// unfused kernels do, is a monotone step function of the integer acc, so sign(z) and the clip mask |z| < 1 are integer
// interval tests.  k_pws_chan_prep evaluates the fp32 expression for EVERY possible acc in [-K, K] (K <= 128) per channel
// and stores the exact thresholds: the decisions are bit-identical to the unfused path at ~2 instructions per element
// instead of ~9, and the batch statistics are exact integer sums (mean / variance formed in fp64 afterwards).
//
// Throughput: the kernel is bound by instruction issue, not by HBM or MFMA, so a wave contracts ALL output channels of its
// group (NT tiles of 16, up to 128) per 64-pixel chunk -- the sign -> bf16 expansion of a B fragment is shared by NT MFMAs --
// loads are unconditional (indices are clamped instead of masked), channel offsets come from an LDS table, and the codes of the
// wave's next chunk are prefetched into the registers of each K-step as soon as that step's fragments are built.
#include "qgemm_dev.h"

#include <stdio.h>
#include <stdlib.h>

#define PWS_Y 0
#define PWS_STATS 1
#define PWS_SIGN8 2
#define PWS_BWD_PART 3
#define PWS_BWD_APPLY 4
#define PWS_BWD_PART_POOL 5      // as PWS_BWD_PART / PWS_BWD_APPLY, but the incoming gradient is that of the 2x2 / stride-2 max-pool
#define PWS_BWD_APPLY_POOL 6     // behind this block: read pooled (a quarter of the bytes) and routed through the pool here

#define PWS_NCH 8           // per-channel constants (k_pws_chan_prep): T, flip, L, U, A, B, gi, unused
struct PwsParams {
    const char* x;            // int8 sign codes [N][Cin_total][HW]
    const uint16_t* wc;       // weight codes [G][Mpad][Kp] (bf16 bits)
    const float* rowscale;    // [G][Mpad]
    const float* bias;        // [G*Mr] or null
    float* y;                 // PWS_Y: y     PWS_BWD_APPLY: dy        [N][Cout_total][HW]
    char* a8;                 // PWS_SIGN8 output
    int16_t* h16;             // XENC 1 (k-bit activation codes), PWS_STATS: the conv result acc as a 16-bit stash (qact_kernels.hip)
    int32_t* h32;             // XENC 2 (8-bit-wide codes / accumulators beyond int16), PWS_STATS: acc as a 32-bit stash
    unsigned char* h8;        // PWS_SIGN8, optional: h = (acc + nnz[o]) / 2 in [0, 128] -- the conv result in one byte (acc has the parity
                              // of nnz[o], the number of non-zero weight codes of the channel); the streaming BN backward reads it
    float* part;              // PWS_STATS / PWS_BWD_PART: [CB][G*Mpad][2]
    const float* chan;        // [PWS_NCH][Cout_total] per-channel constants
    const float* da;          // gradient w.r.t. the sign output ([N][Cout][H][W]), or -- *_POOL -- w.r.t. the pooled output ([N][Cout][H/2][W/2])
    const char* own;          // *_POOL: this block's own output codes [N][Cout][H][W] (the pool routes a gradient to the first +1 of a window)
    int W;                    // *_POOL: image width
    FastDiv fd_w;
    const float* sums;        // [2][Cout_total] sum dz, sum dz*zhat (PWS_BWD_APPLY, training)
    int training;
    float n_f;                // (float)N * (float)HW, the divisor k_bns_apply uses
    float ascale;             // PWS_Y: scale of the activation codes (XENC 1: the quantizer's s; 1 for sign codes)
    int N, HW, Cin_total, Cout_total, Kc, Mr, G, Kp, Mpad, num_mblk, nchunks, CB;
    uint32_t NP;
    FastDiv fd_hw;
    ChanMap in_map;
};


// XENC 0: x holds int8 sign codes (+-1).  XENC 1: x holds k-bit activation codes j in [0, 127] as bytes (the DoReFa / IAO activation quantizer's
// integer, wqaq/dorefa/quantize.py:43-45): the B fragments are built as bf16 128 + j (high byte 0x43, low byte j: one v_perm + one v_or per
// fragment dword), every product and sum stays an exact integer, and the epilogue subtracts 128 * sum_k code_w[o][k] (c2 = that row constant).
// XENC 2: codes j in [0, 255] (W8A8, the configuration of the reference's CPU run; also any width whose accumulator leaves int16): the B fragments are
// bf16 j itself (v_cvt_f32_ubyte + one v_perm per fragment dword: every integer <= 255 is a bf16), no offset and no row constant; |acc| <= K * 255 * 255
// < 2^24 (planner) stays exact in the fp32 accumulators and leaves as a 32-bit stash.  Statistics: fp32 per lane (a few hundred terms: ~1e-7 relative; an
// int32 sum of acc made the register allocator spill 93 VGPRs at NT = 4), fp64 from the block partial on.
template <int NT, int KS, int EPI, int XENC = 0>
__global__ __launch_bounds__(256, 2) void k_pws(const PwsParams p) {
    HIP_DYNAMIC_SHARED(float, smem)
    constexpr int MB = 16 * NT;
    constexpr bool POOL = EPI == PWS_BWD_PART_POOL || EPI == PWS_BWD_APPLY_POOL;
    constexpr bool PART = EPI == PWS_BWD_PART || EPI == PWS_BWD_PART_POOL;
    constexpr bool APPLY = EPI == PWS_BWD_APPLY || EPI == PWS_BWD_APPLY_POOL;
    constexpr bool RED = EPI == PWS_STATS || PART;
    constexpr bool GRAD = PART || APPLY;
    const int LDW = p.Kp + 8;
    uint16_t* wsm = reinterpret_cast<uint16_t*>(smem);
    float* c0 = smem + (MB * LDW) / 2;     // Y: alpha       SIGN8: T      BWD: L
    float* c1 = c0 + MB;                   // Y: bias        SIGN8: flip   BWD: U
    float* c2 = c1 + MB;                   //                              BWD: flip
    float* c3 = c2 + MB;                   //                              BWD: A  (zhat = acc * A + B)
    float* c4 = c3 + MB;                   //                              BWD: B
    float* c5 = c4 + MB;                   //                              APPLY: gi = gamma * invstd
    float* c6 = c5 + MB;                   //                              APPLY: k1 = sum_dz / n
    float* c7 = c6 + MB;                   //                              APPLY: k2 = sum_dzzhat / n
    uint32_t* coff = reinterpret_cast<uint32_t*>(c7 + MB);      // [Kp] element offset of the (clamped, shuffled) input channel
    double* red = reinterpret_cast<double*>(coff + p.Kp);      // [4][MB][2] cross-wave reduction (RED); 8-byte aligned by layout
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kg = lane >> 4;
    const uint32_t HW = (uint32_t)p.HW;

    uint32_t b = blockIdx.x;
    const uint32_t xcd = b & 7u; b >>= 3;
    const int mblk = b % p.num_mblk; b /= p.num_mblk;
    const uint32_t idx = b * 8u + xcd;
    if (idx >= (uint32_t)(p.G * p.CB)) return;
    const int cb = idx % p.CB, g = idx / p.CB;

    // the input-channel offset table first: the code loads of the wave's first two chunks are issued BEFORE the weights are staged, so the two
    // latencies (HBM for the codes, L2 for the weights) overlap instead of adding up at the head of every block (7000 + 3300 of ~60000 cycles)
    for (int c = tid; c < p.Kp; c += 256) coff[c] = (uint32_t)chan_phys(p.in_map, g * p.Kc + (c < p.Kc ? c : p.Kc - 1)) * HW;
    __syncthreads();
    const int chunk0 = cb * 4 + wave, cstride = p.CB * 4;
    const uint16_t* wl = wsm + j * LDW + kg * 8;
    const uint32_t Pmax = p.NP - 4u;                         // NP is a multiple of 4: the last valid quad

    // codes of K-step s of a chunk: dword (4 pixels) of channel s*32 + kg*8 + jj.  Unconditional: a pixel quad beyond the
    // tensor is clamped to the last one (its results are never stored or summed), channels beyond Kc to channel Kc-1
    // (their weight codes are zero).  All tensors are < 4 GiB (planner): 32-bit offsets from the uniform base pointer.
    auto load_step = [&](uint32_t (&dst)[KS * 8], int s, uint32_t xo) {
        const u32x4 o0 = *reinterpret_cast<const u32x4*>(coff + s * 32 + kg * 8), o1 = *reinterpret_cast<const u32x4*>(coff + s * 32 + kg * 8 + 4);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            dst[s * 8 + jj] = *reinterpret_cast<const uint32_t*>(p.x + (xo + o0[jj]));
            dst[s * 8 + 4 + jj] = *reinterpret_cast<const uint32_t*>(p.x + (xo + o1[jj]));
        }
    };
    auto chunk_xo = [&](int chunk) {
        uint32_t P = (uint32_t)chunk * 64u + 4u * j;
        P = P < Pmax ? P : Pmax;
        const uint32_t n = fd_div(P, p.fd_hw);
        return n * (uint32_t)p.Cin_total * HW + (P - n * HW);
    };

    float s1[NT][4], s2[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s1[t][r] = 0.f; s2[t][r] = 0.f; }

    // Two register sets of codes alternate between consecutive chunks of the wave (forward epilogues; the backward ones keep their
    // gradient rows in registers instead): a set is refilled K-step by K-step while its chunk is contracted, i.e. TWO chunks ahead --
    // one chunk of MFMAs (~1.5k cycles) does not cover the HBM latency under load with two waves per SIMD.  Loads are unconditional
    // (chunk_xo clamps past the end).
    constexpr int PF = GRAD ? 1 : 2;
    uint32_t curA[KS * 8], curB[GRAD ? 1 : KS * 8];
    {
        const uint32_t xo = chunk_xo(chunk0);
#pragma unroll
        for (int s = 0; s < KS; ++s) load_step(curA, s, xo);
        if (!GRAD) {
            const uint32_t xo1 = chunk_xo(chunk0 + cstride);
#pragma unroll
            for (int s = 0; s < KS; ++s) load_step(reinterpret_cast<uint32_t (&)[KS * 8]>(curB), s, xo1);
        }
    }
    {   // stage weight codes and the per-channel constants of this m-block (the first two chunks of codes are already in flight)
        const uint16_t* wg = p.wc + ((int64_t)g * p.Mpad + mblk * MB) * p.Kp;
        const int k8 = p.Kp >> 3;
        // row constant c2 (XENC 1: 128 * sum of the row's codes; statistics pass with byte stash: the row's non-zero count) in the same pass when the k8
        // threads of a row are an aligned lane group (k8 = 4, 8, 16, 32, 64): 8 codes per thread, a log2(k8)-step butterfly.  Else: the loops below.
        const bool rowc_here = (XENC == 1 || (EPI == PWS_STATS && p.h8)) && k8 >= 4 && k8 <= 64 && (k8 & (k8 - 1)) == 0 && (MB * k8) % 256 == 0;
        for (int q = tid; q < MB * k8; q += 256) {
            const int row = q / k8, c8 = q - row * k8;
            // the sign codes are contracted as +-0.5 (one v_perm per B-fragment dword, see the main loop): the weight codes are doubled here --
            // a bf16 integer code times two is its exponent field plus one (0x0080), zero stays zero -- so every product is the exact +-code
            u32x4 wv = *reinterpret_cast<const u32x4*>(wg + (int64_t)row * p.Kp + c8 * 8);
            if (rowc_here) {
                float v = 0.f;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    if (XENC == 1) v += mn_u2f(wv[d] << 16) + mn_u2f(wv[d] & 0xffff0000u);
                    else v += ((wv[d] & 0x00007fffu) ? 1.f : 0.f) + ((wv[d] & 0x7fff0000u) ? 1.f : 0.f);
                }
                for (int o = k8 >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);         // exact: small integers
                if (c8 == 0) c2[row] = XENC == 1 ? 128.f * v : v;
            }
            if (XENC == 0) {
#pragma unroll
                for (int d = 0; d < 4; ++d) wv[d] += ((wv[d] & 0x00007fffu) ? 0x00000080u : 0u) | ((wv[d] & 0x7fff0000u) ? 0x00800000u : 0u);
            }
            *reinterpret_cast<u32x4*>(wsm + row * LDW + c8 * 8) = wv;
        }
        for (int i = tid; i < MB; i += 256) {
            const int m = mblk * MB + i;
            const bool mv = m < p.Mr;
            const int co = g * p.Mr + (mv ? m : 0);
            const int C = p.Cout_total;
            if (EPI == PWS_Y) { c0[i] = p.rowscale[g * p.Mpad + m] * p.ascale; c1[i] = (p.bias && mv) ? p.bias[co] : 0.f; }
            if (EPI == PWS_SIGN8) { c0[i] = p.chan[co]; c1[i] = p.chan[C + co]; c2[i] = p.chan[7 * C + co]; }
            if (GRAD) { c0[i] = p.chan[2 * C + co]; c1[i] = p.chan[3 * C + co]; c2[i] = p.chan[C + co]; c3[i] = p.chan[4 * C + co]; c4[i] = p.chan[5 * C + co]; }
            if (APPLY) {
                c5[i] = p.chan[6 * C + co];
                c6[i] = p.training ? p.sums[co] / p.n_f : 0.f;
                c7[i] = p.training ? p.sums[C + co] / p.n_f : 0.f;
            }
        }
    }
    __syncthreads();
    const int k8_ = p.Kp >> 3;
    const bool rowc_done = (XENC == 1 || (EPI == PWS_STATS && p.h8)) && k8_ >= 4 && k8_ <= 64 && (k8_ & (k8_ - 1)) == 0 && (MB * k8_) % 256 == 0;
    if (rowc_done) {
    } else if (XENC == 1) {                   // c2[row] = 128 * sum of the row's weight codes (exact small integers in bf16)
        for (int rb = 0; rb < MB; rb += 64) {
            const int row = rb + (tid >> 2), part = tid & 3;
            float sm = 0.f;
            if (row < MB) for (int k = part; k < p.Kp; k += 4) sm += mn_u2f((uint32_t)wsm[row * LDW + k] << 16);
            sm += __shfl_xor(sm, 2, 64); sm += __shfl_xor(sm, 1, 64);
            if (row < MB && part == 0) c2[row] = 128.f * sm;
        }
        __syncthreads();
    } else if (EPI == PWS_STATS && p.h8) {          // the statistics pass also writes the byte stash: nnz[row] from the staged codes (4 threads per row)
        const int row = tid >> 2, part = tid & 3;
        int cnt = 0;
        if (row < MB) for (int k = part; k < p.Kp; k += 4) cnt += (wsm[row * LDW + k] & 0x7fffu) != 0;
        cnt += __shfl_xor(cnt, 2, 64); cnt += __shfl_xor(cnt, 1, 64);
        if (row < MB && part == 0) c2[row] = (float)cnt;
        __syncthreads();
    }

    auto body = [&](uint32_t (&cur)[KS * 8], int chunk) {
        const uint32_t P = (uint32_t)chunk * 64u + 4u * j;
        const bool pv = P < p.NP;
        const uint32_t n = fd_div(P, p.fd_hw);
        const uint32_t obase = n * (uint32_t)p.Cout_total * HW + (P - n * HW) + (uint32_t)(g * p.Mr + mblk * MB + kg * 4) * HW;   // + (t*16 + r) * HW
        const uint32_t xo_next = chunk_xo(chunk + PF * cstride);

        float4 gq[NT][4];                        // backward: the gradient rows of every tile, in flight during the MFMAs below
        // *_POOL: the lane's pixel quad (row h, columns w .. w+3) covers half of two pooling windows: gq = {g[win 0], g[win 1],
        // own codes of row h & ~1, own codes of row h | 1} and the quad's gradient is g where the pixel is the window's first maximum
        uint32_t gbase = 0u, cbase = 0u, hb = 0u;
        if (POOL) {
            const uint32_t pp = P - n * HW;
            const uint32_t h = fd_div(pv ? pp : 0u, p.fd_w), w = (pv ? pp : 0u) - h * (uint32_t)p.W;
            const uint32_t ch0 = (uint32_t)(g * p.Mr + mblk * MB + kg * 4);
            hb = h & 1u;
            gbase = (n * (uint32_t)p.Cout_total + ch0) * (HW >> 2) + (h >> 1) * ((uint32_t)p.W >> 1) + (w >> 1);
            cbase = (n * (uint32_t)p.Cout_total + ch0) * HW + (h & ~1u) * (uint32_t)p.W + w;
        }
        auto load_grad = [&](int t, int r) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pv && mblk * MB + t * 16 + kg * 4 + r < p.Mr) {
                if (POOL) {
                    const uint32_t ch = (uint32_t)(t * 16 + r);
                    const float2 g2 = *reinterpret_cast<const float2*>(p.da + (gbase + ch * (HW >> 2)));
                    const uint32_t r0 = *reinterpret_cast<const uint32_t*>(p.own + (cbase + ch * HW));
                    const uint32_t r1 = *reinterpret_cast<const uint32_t*>(p.own + (cbase + ch * HW + (uint32_t)p.W));
                    v = make_float4(g2.x, g2.y, mn_u2f(r0), mn_u2f(r1));
                } else {
                    v = *reinterpret_cast<const float4*>(p.da + (obase + (uint32_t)(t * 16 + r) * HW));
                }
            }
            return v;
        };
        if (GRAD) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) gq[t][r] = load_grad(t, r);
        }
        f32x4 acc[4][NT];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            // B fragments: byte q of the 8 channel dwords -> bf16 +-0.5 (0x3F00 / 0xBF00: the sign bit over 0x3F as the HIGH byte, low byte 0):
            // one v_and_or per input dword, one v_perm per fragment dword
            u32x4 bq[4];
            if (XENC == 0) {
                uint32_t en[8];
#pragma unroll
                for (int d = 0; d < 8; ++d) en[d] = (cur[s * 8 + d] & 0x80808080u) | 0x3F3F3F3Fu;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    bq[0][d] = mn_perm(en[2 * d + 1], en[2 * d], 0x040c000cu);
                    bq[1][d] = mn_perm(en[2 * d + 1], en[2 * d], 0x050c010cu);
                    bq[2][d] = mn_perm(en[2 * d + 1], en[2 * d], 0x060c020cu);
                    bq[3][d] = mn_perm(en[2 * d + 1], en[2 * d], 0x070c030cu);
                }
            } else if (XENC == 2) {          // bf16 j, j <= 255: the float's high half
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const uint32_t lo = cur[s * 8 + 2 * d], hi = cur[s * 8 + 2 * d + 1];
                    bq[0][d] = mn_pack_hi16((float)(lo & 0xffu), (float)(hi & 0xffu));
                    bq[1][d] = mn_pack_hi16((float)((lo >> 8) & 0xffu), (float)((hi >> 8) & 0xffu));
                    bq[2][d] = mn_pack_hi16((float)((lo >> 16) & 0xffu), (float)((hi >> 16) & 0xffu));
                    bq[3][d] = mn_pack_hi16((float)(lo >> 24), (float)(hi >> 24));
                }
            } else {          // bf16 (128 + j): low byte = the code, high byte 0x43
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    bq[0][d] = mn_perm(cur[s * 8 + 2 * d + 1], cur[s * 8 + 2 * d], 0x0c040c00u) | 0x43004300u;
                    bq[1][d] = mn_perm(cur[s * 8 + 2 * d + 1], cur[s * 8 + 2 * d], 0x0c050c01u) | 0x43004300u;
                    bq[2][d] = mn_perm(cur[s * 8 + 2 * d + 1], cur[s * 8 + 2 * d], 0x0c060c02u) | 0x43004300u;
                    bq[3][d] = mn_perm(cur[s * 8 + 2 * d + 1], cur[s * 8 + 2 * d], 0x0c070c03u) | 0x43004300u;
                }
            }
            load_step(cur, s, xo_next);                      // this step's registers are free: prefetch chunk + PF into them
            // A fragments one tile ahead only: the scheduler would otherwise hoist all NT LDS reads (4 VGPRs each) above the MFMAs
            u32x4 a = *reinterpret_cast<const u32x4*>(wl + s * 32);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                u32x4 an = a;
                if (t + 1 < NT) an = *reinterpret_cast<const u32x4*>(wl + s * 32 + (t + 1) * 16 * LDW);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q][t] = mn_mfma_bf16(a, bq[q], acc[q][t]);
                a = an;
                MN_SCHED_FENCE();
            }
        }

        // epilogue: lane (j, kg) holds out-channels t*16 + 4kg + r of pixels P .. P+3; acc is an exact integer
        // Statistics pass, whole chunk and whole m-block valid (wave-uniform; every nin_gc chunk): straight-line code.  The guarded form below is one
        // basic block per (tile, row), each with its own LDS read of the row constant and a full wait for it: 16 exposed LDS round trips per chunk,
        // more cycles than the 64 MFMAs in front of them.
        const bool whole = EPI == PWS_STATS && (uint32_t)chunk * 64u + 64u <= p.NP && mblk * MB + MB <= p.Mr;
        if (whole) {
            float4 rc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) rc[t] = *reinterpret_cast<const float4*>(c2 + t * 16 + kg * 4);      // XENC 1: 128 * sum w;  XENC 0: nnz
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float rcv[4] = {rc[t].x, rc[t].y, rc[t].z, rc[t].w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float xc = XENC == 1 ? rcv[r] : 0.f;
                    const float o[4] = {acc[0][t][r] - xc, acc[1][t][r] - xc, acc[2][t][r] - xc, acc[3][t][r] - xc};
                    const uint32_t off = obase + (uint32_t)(t * 16 + r) * HW;
                    if (XENC == 2) {
                        const int oi[4] = {(int)o[0], (int)o[1], (int)o[2], (int)o[3]};
                        s1[t][r] += (o[0] + o[1]) + (o[2] + o[3]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) s2[t][r] = fmaf(o[e], o[e], s2[t][r]);
                        if (p.h32) *reinterpret_cast<u32x4*>(p.h32 + off) = u32x4{(uint32_t)oi[0], (uint32_t)oi[1], (uint32_t)oi[2], (uint32_t)oi[3]};
                        continue;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) { s1[t][r] += o[e]; s2[t][r] = fmaf(o[e], o[e], s2[t][r]); }
                    if (XENC == 1) {
                        if (p.h16) {
                            const uint32_t f0 = mn_f2u(o[0] + 12582912.f), f1 = mn_f2u(o[1] + 12582912.f), f2 = mn_f2u(o[2] + 12582912.f), f3 = mn_f2u(o[3] + 12582912.f);
                            *reinterpret_cast<u32x2*>(p.h16 + off) = u32x2{mn_perm(f1, f0, 0x05040100u), mn_perm(f3, f2, 0x05040100u)};
                        }
                    } else if (p.h8) {
                        const float nz = rcv[r];
                        const uint32_t f0 = mn_f2u(fmaf(o[0] + nz, 0.5f, 12582912.f)), f1 = mn_f2u(fmaf(o[1] + nz, 0.5f, 12582912.f));
                        const uint32_t f2 = mn_f2u(fmaf(o[2] + nz, 0.5f, 12582912.f)), f3 = mn_f2u(fmaf(o[3] + nz, 0.5f, 12582912.f));
                        *reinterpret_cast<uint32_t*>(p.h8 + off) = mn_perm(f1, f0, 0x0c0c0400u) | mn_perm(f3, f2, 0x04000c0cu);
                    }
                }
            }
        } else
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ml = t * 16 + kg * 4 + r;
                const bool ok = pv && mblk * MB + ml < p.Mr;
                const float xc = XENC == 1 ? c2[ml] : 0.f;
                const float o[4] = {acc[0][t][r] - xc, acc[1][t][r] - xc, acc[2][t][r] - xc, acc[3][t][r] - xc};
                const uint32_t off = obase + (uint32_t)(t * 16 + r) * HW;
                if (EPI == PWS_STATS) {
                    // This epilogue was 60 % of the kernel's cycles (stamps: 2300 per chunk in the K loop, 4100 here): ~40 VALU per (tile, row) in
                    // float -> integer conversions, shifts and unfused multiply-adds.  Integers reach the stash through the float's own mantissa
                    // instead (v + 1.5 * 2^23 holds v in two's complement in its low bits: exact for |v| < 2^22) and one v_perm packs the bytes.
                    if (ok && XENC == 2) {
                        const int oi[4] = {(int)o[0], (int)o[1], (int)o[2], (int)o[3]};
                        s1[t][r] += (o[0] + o[1]) + (o[2] + o[3]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) s2[t][r] = fmaf(o[e], o[e], s2[t][r]);
                        if (p.h32) *reinterpret_cast<u32x4*>(p.h32 + off) = u32x4{(uint32_t)oi[0], (uint32_t)oi[1], (uint32_t)oi[2], (uint32_t)oi[3]};
                    } else if (ok) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { s1[t][r] += o[e]; s2[t][r] = fmaf(o[e], o[e], s2[t][r]); }      // exact: integers below 2^24 (fused or not)
                        if (XENC == 1) {
                            if (p.h16) {
                                const uint32_t f0 = mn_f2u(o[0] + 12582912.f), f1 = mn_f2u(o[1] + 12582912.f), f2 = mn_f2u(o[2] + 12582912.f), f3 = mn_f2u(o[3] + 12582912.f);
                                *reinterpret_cast<u32x2*>(p.h16 + off) = u32x2{mn_perm(f1, f0, 0x05040100u), mn_perm(f3, f2, 0x05040100u)};
                            }
                        } else if (p.h8) {          // byte stash written by the statistics pass: the sign then is a streaming pass over h (k_h_sign)
                            const float nz = c2[ml];           // o + nz is even: half of it plus the magic constant is exact
                            const uint32_t f0 = mn_f2u(fmaf(o[0] + nz, 0.5f, 12582912.f)), f1 = mn_f2u(fmaf(o[1] + nz, 0.5f, 12582912.f));
                            const uint32_t f2 = mn_f2u(fmaf(o[2] + nz, 0.5f, 12582912.f)), f3 = mn_f2u(fmaf(o[3] + nz, 0.5f, 12582912.f));
                            *reinterpret_cast<uint32_t*>(p.h8 + off) = mn_perm(f1, f0, 0x0c0c0400u) | mn_perm(f3, f2, 0x04000c0cu);
                        }
                    }
                } else if (EPI == PWS_Y) {
                    const float a_ = c0[ml], b_ = c1[ml];
                    if (ok) *reinterpret_cast<float4*>(p.y + off) = make_float4(o[0] * a_ + b_, o[1] * a_ + b_, o[2] * a_ + b_, o[3] * a_ + b_);
                } else if (EPI == PWS_SIGN8) {
                    const float T = c0[ml], fl = c1[ml];
                    if (ok) {
                        const uint32_t u = (o[0] * fl >= T ? 0x01u : 0xFFu) | (o[1] * fl >= T ? 0x0100u : 0xFF00u) | (o[2] * fl >= T ? 0x010000u : 0xFF0000u) |
                                           (o[3] * fl >= T ? 0x01000000u : 0xFF000000u);
                        *reinterpret_cast<uint32_t*>(p.a8 + off) = u;
                        if (p.h8) {
                            const float nz = c2[ml];
                            const uint32_t hh = (uint32_t)((o[0] + nz) * 0.5f) | ((uint32_t)((o[1] + nz) * 0.5f) << 8) | ((uint32_t)((o[2] + nz) * 0.5f) << 16) |
                                                ((uint32_t)((o[3] + nz) * 0.5f) << 24);
                            *reinterpret_cast<uint32_t*>(p.h8 + off) = hh;
                        }
                    }
                } else {
                    const float L = c0[ml], U = c1[ml], fl = c2[ml], A = c3[ml], B = c4[ml];
                    const float4 g4 = gq[t][r];
                    float gv[4] = {g4.x, g4.y, g4.z, g4.w};
                    if (POOL) {      // first maximum of each window in row-major order (ATen's max_pool2d): the first +1, else element 0
                        const uint32_t r0 = mn_f2u(g4.z), r1 = mn_f2u(g4.w);
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const bool p00 = !((r0 >> (16 * e)) & 0x80u), p01 = !((r0 >> (16 * e + 8)) & 0x80u);
                            const bool p10 = !((r1 >> (16 * e)) & 0x80u), p11 = !((r1 >> (16 * e + 8)) & 0x80u);
                            const uint32_t win = p00 ? 0u : (p01 ? 1u : (p10 ? 2u : (p11 ? 3u : 0u)));
                            const float ge = e ? g4.y : g4.x;
                            gv[2 * e] = win == hb * 2u ? ge : 0.f;
                            gv[2 * e + 1] = win == hb * 2u + 1u ? ge : 0.f;
                        }
                    }
                    float dz[4], zh[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float u = o[e] * fl;
                        dz[e] = (u >= L && u <= U) ? gv[e] : 0.f;          // BinaryActivation.backward: |z| < 1
                        zh[e] = fmaf(o[e], A, B);                           // (y - mean) * invstd
                    }
                    if (PART) {
                        if (ok) {
                            float t1 = 0.f, t2 = 0.f;
#pragma unroll
                            for (int e = 0; e < 4; ++e) { t1 += dz[e]; t2 += dz[e] * zh[e]; }
                            s1[t][r] += t1; s2[t][r] += t2;
                        }
                    } else {
                        const float gi = c5[ml], k1 = c6[ml], k2 = c7[ml];
                        if (ok) *reinterpret_cast<float4*>(p.y + off) = make_float4(gi * (dz[0] - k1 - zh[0] * k2), gi * (dz[1] - k1 - zh[1] * k2),
                                                                                    gi * (dz[2] - k1 - zh[2] * k2), gi * (dz[3] - k1 - zh[3] * k2));
                    }
                }
            }
        }
    };
    if (GRAD) {
        for (int chunk = chunk0; chunk < p.nchunks; chunk += cstride) body(curA, chunk);
    } else {
        for (int chunk = chunk0; chunk < p.nchunks; chunk += 2 * cstride) {
            body(curA, chunk);
            if (chunk + cstride < p.nchunks) body(reinterpret_cast<uint32_t (&)[KS * 8]>(curB), chunk + cstride);
        }
    }

    if (RED) {
        // block partial in fp64, fixed order: every lane's 2 * 4 NT floats go to LDS ([value][wave][lane]: conflict-free rows), then thread (row, which)
        // adds its 4 waves x 16 pixel lanes.  (The butterfly of 64-bit shuffles this replaces cost ~9000 cycles per block: 256 ds_bpermute.)
        __syncthreads();                                    // every wave is done with the staged weights: the LDS is free
        float* rf = smem;                                   // [NT * 8][4][64]
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                rf[(((t * 4 + r) * 2 + 0) * 4 + wave) * 64 + lane] = s1[t][r];
                rf[(((t * 4 + r) * 2 + 1) * 4 + wave) * 64 + lane] = s2[t][r];
            }
        __syncthreads();
        for (int i = tid; i < 2 * MB; i += 256) {
            const int ml = i >> 1, which = i & 1;
            const int t = ml >> 4, kq = (ml >> 2) & 3, r = ml & 3;
            const float* src = rf + ((t * 4 + r) * 2 + which) * 4 * 64 + kq * 16;
            double v = 0.0;
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int q = 0; q < 16; ++q) v += (double)src[w * 64 + q];
            double* dst = reinterpret_cast<double*>(p.part) + ((int64_t)cb * p.G * p.Mpad + g * p.Mpad + mblk * MB + ml) * 2;
            dst[which] = v;
        }
    }
}

// batch statistics from the PWS_STATS partials (exact integer sums S1 = sum acc, S2 = sum acc^2 over N*HW): y = alpha*acc + bias,
// so mean = alpha * S1/n + bias and the biased variance = alpha^2 * (S2/n - (S1/n)^2), formed in fp64; same outputs as
// k_bns_final_fwd (norm_kernels.hip): save = {mean, invstd}, running stats with the UNBIASED variance.  One wave per channel.
__global__ __launch_bounds__(64) void k_pws_final_fwd(const double* __restrict__ part, int CB, int G, int Mpad, int Mr, const float* __restrict__ rowscale,
                                                     const float* __restrict__ bias, double n, float eps, float momentum, float* __restrict__ running_mean,
                                                     float* __restrict__ running_var, float* __restrict__ save, int Cout) {
    const int co = blockIdx.x, g = co / Mr, m = co - g * Mr, lane = threadIdx.x;
    double a1 = 0.0, a2 = 0.0;
    for (int i = lane; i < CB; i += 64) {
        const double* src = part + ((int64_t)i * G * Mpad + g * Mpad + m) * 2;
        a1 += src[0]; a2 += src[1];
    }
    a1 = wave_reduce(a1, OpAddD()); a2 = wave_reduce(a2, OpAddD());
    if (lane == 0) {
        const double al = (double)rowscale[g * Mpad + m];
        const double ma = a1 / n;
        const double mean = al * ma + (double)(bias ? bias[co] : 0.f);
        double ss = al * al * (a2 - a1 * ma);                 // sum of squared deviations of y
        if (ss < 0.0) ss = 0.0;
        const float var_b = (float)(ss / n);
        save[co] = (float)mean;
        save[Cout + co] = 1.0f / sqrtf(var_b + eps);
        if (running_mean) running_mean[co] = (1.f - momentum) * running_mean[co] + momentum * (float)mean;
        if (running_var) running_var[co] = (1.f - momentum) * running_var[co] + momentum * (float)(ss / (n - 1.0));
    }
}
__global__ __launch_bounds__(64) void k_pws_final_bwd(const double* __restrict__ part, int CB, int G, int Mpad, int Mr, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, float* __restrict__ sums, int Cout) {
    const int co = blockIdx.x, g = co / Mr, m = co - g * Mr, lane = threadIdx.x;
    double a1 = 0.0, a2 = 0.0;
    for (int i = lane; i < CB; i += 64) {
        const double* src = part + ((int64_t)i * G * Mpad + g * Mpad + m) * 2;
        a1 += src[0]; a2 += src[1];
    }
    a1 = wave_reduce(a1, OpAddD()); a2 = wave_reduce(a2, OpAddD());
    if (lane == 0) {
        if (dbeta) dbeta[co] = (float)a1;
        if (dgamma) dgamma[co] = (float)a2;
        sums[co] = (float)a1; sums[Cout + co] = (float)a2;
    }
}
__global__ void k_pws_eval_stats(int C, float eps, const float* __restrict__ running_mean, const float* __restrict__ running_var, float* __restrict__ save) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    save[c] = running_mean[c];
    save[C + c] = 1.0f / sqrtf(running_var[c] + eps);
}
// per-channel integer-domain constants.  z(acc) below is the fp32 expression chain of the unfused kernels (k_pw epilogue:
// y = acc*alpha + bias; k_bns_apply: zh = (y - mean)*invstd, z = zh*gamma + beta; -ffp-contract=off keeps every rounding).
// Each step is monotone in acc, so z is a monotone step function over the integers [-K, K]; with u = acc*flip it is
// non-decreasing and  z < 0  <=>  u < T,   -1 < z < 1  <=>  L <= u <= U.  One wave evaluates all 2K+1 values.
__device__ __forceinline__ float pws_z(float acc, float al, float b, float mean, float invstd, float ga, float be) {
    const float y = acc * al + b;
    const float zh = (y - mean) * invstd;
    return zh * ga + be;
}
__device__ __forceinline__ void pws_prep_core(int K, int Kp, const uint16_t* __restrict__ wc, int g, int m, int Mpad, int co, int lane, float al, float b, float mean,
                                              float invstd, float ga, float be, float* __restrict__ chan, int Cout, const float* __restrict__ nnz9) {
    const float zlo = pws_z(-(float)K, al, b, mean, invstd, ga, be), zhi = pws_z((float)K, al, b, mean, invstd, ga, be);
    const float flip = (zhi < zlo) ? -1.f : 1.f;              // NaN anywhere: flip = +1 and every test below is false -> constant outputs
    // non-decreasing in u: T = min{u : !(z < 0)}, L = min{u : z > -1}, U = max{u : z < 1}
    int T = K + 1, L = K + 1, U = -K - 1;
    for (int u = -K + lane; u <= K; u += 64) {
        const float z = pws_z((float)u * flip, al, b, mean, invstd, ga, be);
        if (!(z < 0.f) && u < T) T = u;
        if (z > -1.f && u < L) L = u;
        if (z < 1.f && u > U) U = u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int t2 = __shfl_xor(T, o, 64), l2 = __shfl_xor(L, o, 64), u2 = __shfl_xor(U, o, 64);
        T = t2 < T ? t2 : T; L = l2 < L ? l2 : L; U = u2 > U ? u2 : U;
    }
    int nnz = 0;                                               // non-zero weight codes of the row: acc has its parity
    for (int k = lane; k < Kp; k += 64) nnz += (wc[((int64_t)g * Mpad + m) * Kp + k] & 0x7fffu) != 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nnz += __shfl_xor(nnz, o, 64);
    if (lane == 0) {
        chan[co] = (float)T; chan[Cout + co] = flip; chan[2 * Cout + co] = (float)L; chan[3 * Cout + co] = (float)U;
        chan[4 * Cout + co] = al * invstd;                    // zhat = acc*A + B  (<= 2 ulp from the unfused chain; dy tolerance 1e-5)
        chan[5 * Cout + co] = (b - mean) * invstd;
        chan[6 * Cout + co] = ga * invstd;
        chan[7 * Cout + co] = nnz9 ? -1.f : (float)nnz;      // -1: per-pixel-class nnz in rows 8..16 (3x3 blocks; stash_nnz_load, common.h)
    }
    if (nnz9 && lane < 9) chan[(8 + lane) * Cout + co] = nnz9[lane * Cout + co];
}
__global__ __launch_bounds__(64) void k_pws_chan_prep(int K, int Kp, const uint16_t* __restrict__ wc, int G, int Mpad, int Mr, const float* __restrict__ rowscale, const float* __restrict__ bias,
                                                     const float* __restrict__ save, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ chan, int Cout) {
    const int co = blockIdx.x, g = co / Mr, m = co - g * Mr, lane = threadIdx.x;
    pws_prep_core(K, Kp, wc, g, m, Mpad, co, lane, rowscale[g * Mpad + m], bias ? bias[co] : 0.f, save[co], save[Cout + co], gamma[co], beta[co], chan, Cout, nullptr);
}
// forward of a stashed block, one launch per block instead of three: batch statistics from the partials (training; as k_pws_final_fwd) or
// from the running statistics (eval), the per-channel constants (as k_pws_chan_prep), the per-pixel-class nnz rows of a 3x3 block and
// BatchNorm's num_batches_tracked counter (nullable).  One wave per channel.
__global__ __launch_bounds__(64) void k_pws_stats_prep(const double* __restrict__ part, int CB, int G, int Mpad, int Mr, const float* __restrict__ rowscale,
                                                      const float* __restrict__ bias, double n, float eps, float momentum, int training,
                                                      float* __restrict__ running_mean, float* __restrict__ running_var, float* __restrict__ save, int Cout,
                                                      int K, int Kp, const uint16_t* __restrict__ wc, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ chan, const float* __restrict__ nnz9, long long* __restrict__ nbt) {
    const int co = blockIdx.x, g = co / Mr, m = co - g * Mr, lane = threadIdx.x;
    float mean_f = 0.f, inv_f = 0.f;
    if (training) {
        double a1 = 0.0, a2 = 0.0;
        for (int i = lane; i < CB; i += 64) {
            const double* src = part + ((int64_t)i * G * Mpad + g * Mpad + m) * 2;
            a1 += src[0]; a2 += src[1];
        }
        a1 = wave_reduce(a1, OpAddD()); a2 = wave_reduce(a2, OpAddD());
        if (lane == 0) {
            const double al = (double)rowscale[g * Mpad + m];
            const double ma = a1 / n;
            const double mean = al * ma + (double)(bias ? bias[co] : 0.f);
            double ss = al * al * (a2 - a1 * ma);                 // sum of squared deviations of y
            if (ss < 0.0) ss = 0.0;
            const float var_b = (float)(ss / n);
            mean_f = (float)mean;
            inv_f = 1.0f / sqrtf(var_b + eps);
            if (running_mean) running_mean[co] = (1.f - momentum) * running_mean[co] + momentum * (float)mean;
            if (running_var) running_var[co] = (1.f - momentum) * running_var[co] + momentum * (float)(ss / (n - 1.0));
        }
    } else if (lane == 0) {
        mean_f = running_mean[co];
        inv_f = 1.0f / sqrtf(running_var[co] + eps);
    }
    if (lane == 0) {
        save[co] = mean_f; save[Cout + co] = inv_f;
        if (nbt && co == 0 && training) *nbt += 1;
    }
    mean_f = __shfl(mean_f, 0, 64); inv_f = __shfl(inv_f, 0, 64);
    pws_prep_core(K, Kp, wc, g, m, Mpad, co, lane, rowscale[g * Mpad + m], bias ? bias[co] : 0.f, mean_f, inv_f, gamma[co], beta[co], chan, Cout, nnz9);
}

// k_pws_stats_prep's arguments, for its fold into k_h_sign (default; MN_HSIGN_FOLD=0 turns it off): every block of the streaming sign pass evaluates its channel's constants itself (one
// wave: the same reduction over the partial rows, the same fp32 chains, the same shuffles -- bit-identical values in every block), the block sp == 0 of a channel is
// the one that writes them (save, running statistics, chan rows, the counter): one latency-bound launch less per block of the net.
struct HsPrep {
    const double* part; int CB, G, Mpad, Mr; const float* rowscale; const float* bias; double n; float eps, momentum; int training;
    float* running_mean; float* running_var; float* save; int Cout, K, Kp; const uint16_t* wc; const float* gamma; const float* beta; float* chan; const float* nnz9;
    long long* nbt;
};
// one wave; returns T, flip and the row's nnz (every lane).  `write`: this wave is the channel's writer.
__device__ __forceinline__ void pws_stats_prep_dev(const HsPrep& q, int co, int lane, bool write, float& T_out, float& flip_out, float& nnz_out) {
    const int g = co / q.Mr, m = co - g * q.Mr;
    float mean_f = 0.f, inv_f = 0.f;
    if (q.training) {
        double a1 = 0.0, a2 = 0.0;
        for (int i = lane; i < q.CB; i += 64) {
            const double* src = q.part + ((int64_t)i * q.G * q.Mpad + g * q.Mpad + m) * 2;
            a1 += src[0]; a2 += src[1];
        }
        a1 = wave_reduce(a1, OpAddD()); a2 = wave_reduce(a2, OpAddD());
        if (lane == 0) {
            const double al = (double)q.rowscale[g * q.Mpad + m];
            const double ma = a1 / q.n;
            const double mean = al * ma + (double)(q.bias ? q.bias[co] : 0.f);
            double ss = al * al * (a2 - a1 * ma);
            if (ss < 0.0) ss = 0.0;
            const float var_b = (float)(ss / q.n);
            mean_f = (float)mean;
            inv_f = 1.0f / sqrtf(var_b + q.eps);
            if (write && q.running_mean) q.running_mean[co] = (1.f - q.momentum) * q.running_mean[co] + q.momentum * (float)mean;
            if (write && q.running_var) q.running_var[co] = (1.f - q.momentum) * q.running_var[co] + q.momentum * (float)(ss / (q.n - 1.0));
        }
    } else if (lane == 0) {
        mean_f = q.running_mean[co];
        inv_f = 1.0f / sqrtf(q.running_var[co] + q.eps);
    }
    if (lane == 0 && write) {
        q.save[co] = mean_f; q.save[q.Cout + co] = inv_f;
        if (q.nbt && co == 0 && q.training) *q.nbt += 1;
    }
    mean_f = __shfl(mean_f, 0, 64); inv_f = __shfl(inv_f, 0, 64);
    const float al = q.rowscale[g * q.Mpad + m], b = q.bias ? q.bias[co] : 0.f, ga = q.gamma[co], be = q.beta[co];
    const int K = q.K;
    const float zlo = pws_z(-(float)K, al, b, mean_f, inv_f, ga, be), zhi = pws_z((float)K, al, b, mean_f, inv_f, ga, be);
    const float flip = (zhi < zlo) ? -1.f : 1.f;
    int T = K + 1, L = K + 1, U = -K - 1;
    for (int u = -K + lane; u <= K; u += 64) {
        const float z = pws_z((float)u * flip, al, b, mean_f, inv_f, ga, be);
        if (!(z < 0.f) && u < T) T = u;
        if (z > -1.f && u < L) L = u;
        if (z < 1.f && u > U) U = u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int t2 = __shfl_xor(T, o, 64), l2 = __shfl_xor(L, o, 64), u2 = __shfl_xor(U, o, 64);
        T = t2 < T ? t2 : T; L = l2 < L ? l2 : L; U = u2 > U ? u2 : U;
    }
    int nnz = 0;
    for (int k = lane; k < q.Kp; k += 64) nnz += (q.wc[((int64_t)g * q.Mpad + m) * q.Kp + k] & 0x7fffu) != 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nnz += __shfl_xor(nnz, o, 64);
    if (write) {
        float* chan = q.chan;
        const int Cout = q.Cout;
        if (lane == 0) {
            chan[co] = (float)T; chan[Cout + co] = flip; chan[2 * Cout + co] = (float)L; chan[3 * Cout + co] = (float)U;
            chan[4 * Cout + co] = al * inv_f;
            chan[5 * Cout + co] = (b - mean_f) * inv_f;
            chan[6 * Cout + co] = ga * inv_f;
            chan[7 * Cout + co] = q.nnz9 ? -1.f : (float)nnz;
        }
        if (q.nnz9 && lane < 9) chan[(8 + lane) * Cout + co] = q.nnz9[lane * Cout + co];
    }
    T_out = (float)T; flip_out = flip; nnz_out = (float)nnz;
}

// ------------------------------------------------------------------------------------------------
// pointwise backward-weight on sign codes, without LDS:  dwq[g][m][c] = sum_{n,p} gy[n][g*Mg+m][p] * a[n][g*Cg+c][p].
// The contraction index is the pixel, and BOTH operands are pixel-contiguous in NCHW: lane (i, kg) loads the 8 pixels of its
// K slots of channel row i straight from global memory -- gy as two float4 (split into three exact bf16 terms in registers),
// the activation as two dwords of sign codes -- so fragments are built without staging, transposition or barriers.  K slot
// (kg, e) of a 32-pixel step is pixel 4kg + e (e < 4) / 16 + 4kg + (e - 4): each load instruction covers 64 contiguous bytes of
// a gy row.  A wave owns a (16 MW) x (16 CW) tile of dw over the block's pixel range, the four waves of a block a 2 x 2
// arrangement of such tiles (their re-reads of gy / codes hit L1 / L2); the next step's loads fly during the MFMAs of the
// current one.  Partial tiles and dbias partials have the layout of k_pw_wgrad and are reduced by k_pw_wgrad_reduce
// (fixed order, fp64): deterministic.
struct Wg2Params {
    const float* gy;
    const char* x;
    float* part;     // [Z][G][Mgw][Cgw]
    float* dbpart;   // [Z][G][Mgw]
    int N, HW, Cin_total, Cout_total, Cg, Mg, G, nmb, ncb, Z, nsteps, Mgw, Cgw, want_db;
    int st_per_z, st_stride;      // block z contracts steps z*st_per_z + i*st_stride, i < its count (contiguous ranges: st_stride = 1)
    FastDiv fd_hw;
    ChanMap in_map;
    // BNH variant: gy is the BatchNorm+sign backward of (da = gy pointer, h), formed in registers (bnh_fold, qgemm_dev.h)
    const unsigned char* h;
    const float* chan;
    const float* sums;
    int training;
    float n_f;
    // BNH 2 (k_pws_wgrad_s only): gy is the POOLED gradient [N][Cout][H/2][W/2] of the 2x2 max-pool behind the block, `own` the block's own sign output: the
    // staging threads route a window's gradient to its first +1 (k_bnh_apply<1>'s rule) before the BatchNorm fold -- the full-size dy is never written
    const char* own;
    int W;
    FastDiv fd_w;
};
template <int MW, int CW, int BNH>
__global__ __launch_bounds__(256, 2) void k_pws_wgrad(const Wg2Params p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kg = lane >> 4;
    // CW == 8: the four waves are four row slices of 16*MW channels each reading ALL 128 columns (gy is loaded once per block, only the
    // one-byte codes are shared by the waves); otherwise a 2 x 2 arrangement
    constexpr int WCN = CW == 8 ? 1 : 2;
    const int wm = WCN == 1 ? wave : (wave >> 1), wc = WCN == 1 ? 0 : (wave & 1);
    uint32_t b = blockIdx.x;
    const int z = b % p.Z; b /= p.Z;
    const int cb = b % p.ncb; b /= p.ncb;
    const int mb = b % p.nmb;
    const int g = b / p.nmb;
    const uint32_t HW = (uint32_t)p.HW;
    constexpr int TM = 16 * MW * (4 / WCN), TC = 16 * CW * WCN;

    uint32_t goff[MW], xoff[CW];          // channel offsets (rows beyond Mg / Cg are clamped: their dw entries are never read)
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
        int m = mb * TM + (wm * MW + mi) * 16 + j;
        m = m < p.Mg ? m : p.Mg - 1;
        goff[mi] = (uint32_t)(g * p.Mg + m) * HW;
    }
#pragma unroll
    for (int ci = 0; ci < CW; ++ci) {
        int c = cb * TC + (wc * CW + ci) * 16 + j;
        c = c < p.Cg ? c : p.Cg - 1;
        xoff[ci] = (uint32_t)chan_phys(p.in_map, g * p.Cg + c) * HW;
    }
    __shared__ float ftab[5 * 16 * MW * (CW == 8 ? 4 : 2)];          // BNH: the per-channel fold of the block's TM channel rows (kept out of the register file)
    if (BNH) {
        for (int i = tid; i < TM; i += 256) {
            int m = mb * TM + i;
            m = m < p.Mg ? m : p.Mg - 1;
            float hlo, hhi, G, E1, E0;
            bnh_fold(p.chan, p.sums, p.Cout_total, g * p.Mg + m, p.training, p.n_f, 1.f, hlo, hhi, G, E1, E0);
            ftab[i] = hlo; ftab[TM + i] = hhi; ftab[2 * TM + i] = G; ftab[3 * TM + i] = E1; ftab[4 * TM + i] = E0;
        }
        __syncthreads();
    }
    f32x4 acc[MW][CW];
    float dbacc[MW];
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
        dbacc[mi] = 0.f;
#pragma unroll
        for (int ci = 0; ci < CW; ++ci) acc[mi][ci] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // two register sets alternate: while one step is contracted, the loads of the next TWO steps are in flight
    constexpr int NSETS = (BNH && MW == 4) ? 1 : 2;          // the BN fold needs 5*MW + 2*MW more registers: one register set then
    struct Raw { float4 ga[MW], gb[MW]; uint32_t ua[CW], ub[CW]; uint32_t ha[MW], hb[MW]; };
    Raw r0, r1;
    const int st0 = p.st_stride == 1 ? z * p.st_per_z : z;
    auto fetch = [&](Raw& R, int st) {
        const uint32_t Pa = (uint32_t)st * 32u + 4u * kg, Pb = Pa + 16u;
        const uint32_t na = fd_div(Pa, p.fd_hw), nb = fd_div(Pb, p.fd_hw);
        const uint32_t pa = Pa - na * HW, pb = Pb - nb * HW;
        const uint32_t oa = na * (uint32_t)p.Cout_total * HW + pa, ob = nb * (uint32_t)p.Cout_total * HW + pb;
        const uint32_t xa = na * (uint32_t)p.Cin_total * HW + pa, xb = nb * (uint32_t)p.Cin_total * HW + pb;
#pragma unroll
        for (int mi = 0; mi < MW; ++mi) {
            R.ga[mi] = *reinterpret_cast<const float4*>(p.gy + (oa + goff[mi]));
            R.gb[mi] = *reinterpret_cast<const float4*>(p.gy + (ob + goff[mi]));
            if (BNH) {
                R.ha[mi] = *reinterpret_cast<const uint32_t*>(p.h + (oa + goff[mi]));
                R.hb[mi] = *reinterpret_cast<const uint32_t*>(p.h + (ob + goff[mi]));
            }
        }
#pragma unroll
        for (int ci = 0; ci < CW; ++ci) {
            R.ua[ci] = *reinterpret_cast<const uint32_t*>(p.x + (xa + xoff[ci]));
            R.ub[ci] = *reinterpret_cast<const uint32_t*>(p.x + (xb + xoff[ci]));
        }
    };
    const int st_end = p.st_stride == 1 ? ((st0 + p.st_per_z) < p.nsteps ? (st0 + p.st_per_z) : p.nsteps) : p.nsteps;
    auto contract = [&](Raw& R, int st) {
        // B fragments: sign codes -> bf16 +-1
        u32x4 bf[CW];
#pragma unroll
        for (int ci = 0; ci < CW; ++ci) {
            const uint32_t u = R.ua[ci], v = R.ub[ci];
            bf[ci] = u32x4{0x3F803F80u | ((u & 0x80u) << 8) | ((u & 0x8000u) << 16), 0x3F803F80u | ((u & 0x800000u) >> 8) | (u & 0x80000000u),
                           0x3F803F80u | ((v & 0x80u) << 8) | ((v & 0x8000u) << 16), 0x3F803F80u | ((v & 0x800000u) >> 8) | (v & 0x80000000u)};
        }
        // A fragments: three exact bf16 terms of the 8 gy values
        u32x4 a0[MW], a1[MW], a2[MW];
#pragma unroll
        for (int mi = 0; mi < MW; ++mi) {
            float v[8] = {R.ga[mi].x, R.ga[mi].y, R.ga[mi].z, R.ga[mi].w, R.gb[mi].x, R.gb[mi].y, R.gb[mi].z, R.gb[mi].w};
            if (BNH) {
                const int row = (wm * MW + mi) * 16 + j;
                const float hlo = ftab[row], hhi = ftab[TM + row], G = ftab[2 * TM + row], E1 = ftab[3 * TM + row], E0 = ftab[4 * TM + row];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float hf = (float)(((e < 4 ? R.ha[mi] : R.hb[mi]) >> (8 * (e & 3))) & 0xffu);
                    const float dz = (hf >= hlo && hf <= hhi) ? v[e] : 0.f;
                    v[e] = fmaf(G, dz, fmaf(E1, hf, E0));
                }
            }
            float t0[8], t1[8], t2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                t0[e] = mn_bf16_head(v[e]);
                const float r1 = v[e] - t0[e];
                t1[e] = mn_bf16_head(r1);
                t2[e] = r1 - t1[e];
            }
            dbacc[mi] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                a0[mi][d] = mn_pack_bf16x2(t0[2 * d], t0[2 * d + 1]);
                a1[mi][d] = mn_pack_bf16x2(t1[2 * d], t1[2 * d + 1]);
                a2[mi][d] = mn_pack_bf16x2(t2[2 * d], t2[2 * d + 1]);
            }
        }
        if (st + NSETS * p.st_stride < st_end) fetch(R, st + NSETS * p.st_stride);     // the registers are free: NSETS steps ahead
        // term-outer: MW*CW independent accumulators between two MFMAs on the same one (a dependent MFMA waits ~2 issue slots)
#pragma unroll
        for (int mi = 0; mi < MW; ++mi)
#pragma unroll
            for (int ci = 0; ci < CW; ++ci) acc[mi][ci] = mn_mfma_bf16(a0[mi], bf[ci], acc[mi][ci]);
#pragma unroll
        for (int mi = 0; mi < MW; ++mi)
#pragma unroll
            for (int ci = 0; ci < CW; ++ci) acc[mi][ci] = mn_mfma_bf16(a1[mi], bf[ci], acc[mi][ci]);
#pragma unroll
        for (int mi = 0; mi < MW; ++mi)
#pragma unroll
            for (int ci = 0; ci < CW; ++ci) acc[mi][ci] = mn_mfma_bf16(a2[mi], bf[ci], acc[mi][ci]);
    };
    if (st0 < st_end) fetch(r0, st0);
    if (NSETS == 2) {
        if (st0 + p.st_stride < st_end) fetch(r1, st0 + p.st_stride);
        for (int st = st0; st < st_end; st += 2 * p.st_stride) {
            contract(r0, st);
            if (st + p.st_stride < st_end) contract(r1, st + p.st_stride);
        }
    } else {
        for (int st = st0; st < st_end; st += p.st_stride) contract(r0, st);
    }
    // partial tile: lane (j, kg) holds rows m = 4kg + r, column c = j
#pragma unroll
    for (int mi = 0; mi < MW; ++mi) {
#pragma unroll
        for (int ci = 0; ci < CW; ++ci) {
            const int mrow = mb * TM + (wm * MW + mi) * 16 + kg * 4;
            const int ccol = cb * TC + (wc * CW + ci) * 16 + j;
            float* dst = p.part + (((int64_t)z * p.G + g) * p.Mgw + mrow) * p.Cgw + ccol;
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(int64_t)r * p.Cgw] = acc[mi][ci][r];
        }
        if (p.want_db && cb == 0 && wc == 0) {
            float v = dbacc[mi];
            v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);        // the four kg groups hold different pixels of row j
            if (kg == 0) p.dbpart[((int64_t)z * p.G + g) * p.Mgw + mb * TM + (wm * MW + mi) * 16 + j] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same contraction with the operands STAGED THROUGH LDS.  In k_pws_wgrad a load instruction's 16 consecutive lanes are the 16
// channel rows of an MFMA fragment: every 16-lane group touches 16 different cache lines (64 tag look-ups per instruction instead
// of 8), and the kernel is bound by the texture-address / L1 tag rate, not by HBM, MFMA or the term split (ablation: neither
// removing the MFMAs, nor the split, nor the HBM stream -- one step re-read from L1 -- moved its time by more than 15 %).  Here
// the 256 threads of a block load the block's 32-pixel step row by row (8 lanes = one 128-byte line of a gy row; 2 lanes = the
// 32 codes of an activation row), two steps ahead into two register sets, and hand it over in a double-buffered LDS image
// (gy rows padded to 160 B, code rows to 48 B and chunk-interleaved: the b128 / b64 fragment reads of a wave are conflict-free);
// the fragments are then built exactly as in k_pws_wgrad.  One barrier per step.  The BatchNorm fold (BNH) and the bias gradient
// move to the staging threads (once per block instead of once per wave).  Same partial-tile layout, same reduction kernel.
#define WG3_RSA 160
#define WG3_RSB 48
#ifndef WG3_PRESPLIT
#define WG3_PRESPLIT 1
#endif
// XENC 1: x holds k-bit activation codes j (bytes): the staging threads expand them ONCE per block to bf16 j (exact; rows of 64 B = 4 chunks of 8
// pixels, chunk c of row r stored at position c ^ ((r >> 2) & 3): the b128 fragment reads of 16 consecutive rows fall on distinct banks) and
// the waves read ready B fragments; the reduction kernel multiplies by the quantizer's scale s.  (No 128-offset trick here: the gradient is
// real-valued, an offset 40x larger than the signal would cost its low bits.)
#define WG3_RSBX 64
#ifndef WG3_NS
#define WG3_NS 4
#endif
// SPEC 1: wave-specialised, 512 threads.  Waves 0-3 are PRODUCERS (global loads, BatchNorm fold, the three-term split, LDS writes: VALU only),
// waves 4-7 CONSUMERS (fragment reads + MFMA only, the 2 x 2 wave grid of the tile).  Every SIMD hosts one of each, so the VALU pipe and the matrix
// pipe run concurrently instead of one after the other in the same wave (PMC of the unspecialised kernel at one wave per SIMD: 46 % of the wave
// cycles issuing VALU, 27 % stalled behind its own MFMAs, 27 % parked; two blocks per CU did not overlap the phases either).  Same barrier sequence.
// SPEC 2: EIGHT consumer waves (768 threads; 4 x 2 wave grid, 32 x 64 per wave).  One wave issues a v_mfma_f32_16x16x32_bf16 only every ~27 cycles
// (measured: 16 independent accumulators, one wave per SIMD); two MFMA-issuing waves per SIMD bring the pipe to ~14-17 cycles per instruction.
template <int MW, int BNH, int XENC = 0, int SPEC = 0>
__global__ __launch_bounds__(SPEC == 2 ? 768 : (SPEC ? 512 : 256), SPEC ? 1 : 2) void k_pws_wgrad_s(const Wg2Params p) {
    // WG3_PRESPLIT: the staging threads split gy into its three bf16 terms ONCE per block and store three bf16 planes (rows of 80 B); the
    // waves then read ready fragments (no VALU between LDS and MFMA; the split is no longer done twice, by both waves of a row half)
    constexpr int CW = MW, TM = 32 * MW, TC = 32 * MW, RPT = TM / 32;
    constexpr int RS2 = 80, PLANE = TM * RS2;
    constexpr int RSB = XENC ? WG3_RSBX : WG3_RSB;
    constexpr int BUF = WG3_PRESPLIT ? 3 * PLANE + TC * RSB : TM * WG3_RSA + TC * RSB;
    HIP_DYNAMIC_SHARED(float, smemw)
    unsigned char* lds = reinterpret_cast<unsigned char*>(smemw);          // [2][BUF], then the fold table
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6, j = lane & 15, kg = lane >> 4;
    const bool prod = !SPEC || threadIdx.x < 256, cons = !SPEC || threadIdx.x >= 256;
    constexpr int MWc = SPEC == 2 ? MW / 2 : MW;                   // row tiles of a consumer wave
    const int cwv = SPEC ? ((int)threadIdx.x - 256) >> 6 : wave;   // consumer wave index
    const int wm = cwv >> 1, wc = cwv & 1;
    uint32_t b = blockIdx.x;
    const int z = b % p.Z; b /= p.Z;
    const int cb = b % p.ncb; b /= p.ncb;
    const int mb = b % p.nmb;
    const int g = b / p.nmb;
    const uint32_t HW = (uint32_t)p.HW;

    // staging roles: gy row sr + 32 i, pixels 4 sq .. 4 sq + 3 of the step; code row cr, pixels 16 chf .. 16 chf + 15
    const int sr = tid >> 3, sq = tid & 7, cr = tid >> 1, chf = tid & 1;
    const bool cdo = TC >= 128 || cr < TC;
    uint32_t goff[RPT], xoff;
    float dbacc[RPT];
    const bool want_db = p.want_db != 0;
    float* ftab = reinterpret_cast<float*>(lds + 2 * BUF);                // BNH: [TM][8] the per-row fold (hlo, hhi, G, E1, E0), kept out of the register file
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        int m = mb * TM + sr + 32 * i;
        m = m < p.Mg ? m : p.Mg - 1;
        goff[i] = (uint32_t)(g * p.Mg + m) * HW;
        dbacc[i] = 0.f;
    }
    if (BNH && prod) {
        for (int r = tid; r < TM; r += 256) {
            int m = mb * TM + r;
            m = m < p.Mg ? m : p.Mg - 1;
            float hlo, hhi, G_, E1, E0;
            bnh_fold(p.chan, p.sums, p.Cout_total, g * p.Mg + m, p.training, p.n_f, 1.f, hlo, hhi, G_, E1, E0);
            ftab[r * 8 + 0] = hlo; ftab[r * 8 + 1] = hhi; ftab[r * 8 + 2] = G_; ftab[r * 8 + 3] = E1; ftab[r * 8 + 4] = E0;
        }
        // visible after the first __syncthreads below (before any commit)
    }
    {
        int c = cb * TC + (cdo ? cr : 0);
        c = c < p.Cg ? c : p.Cg - 1;
        xoff = (uint32_t)chan_phys(p.in_map, g * p.Cg + c) * HW;
    }
    f32x4 acc[MW][CW];
#pragma unroll
    for (int mi = 0; mi < MW; ++mi)
#pragma unroll
        for (int ci = 0; ci < CW; ++ci) acc[mi][ci] = f32x4{0.f, 0.f, 0.f, 0.f};

    struct Stage { float4 gv[RPT]; uint32_t hv[RPT]; u32x4 cv; uint32_t hbit; };
    Stage s0, s1;
    // contiguous ranges (st_stride == 1) or interleaved: block z takes steps z, z + Z, ... -- neighbouring blocks then read neighbouring
    // 128-byte pieces of the same rows at about the same time (DRAM page locality)
    const int st0 = p.st_stride == 1 ? z * p.st_per_z : z;
    const int n = p.st_stride == 1 ? (((st0 + p.st_per_z) < p.nsteps ? (st0 + p.st_per_z) : p.nsteps) - st0)
                                   : (st0 < p.nsteps ? (p.nsteps - st0 + p.st_stride - 1) / p.st_stride : 0);
    // loads are unconditional (a conditional load makes the register set a phi: copies, and a vmcnt(0) right behind the issue)
    auto fetch = [&](Stage& S, int k) {
        // past the block's range: re-read the block's OWN last step (an L2 hit), not the neighbour's first steps (those were real HBM traffic:
        // prefetch depth x 20 KB per block = 20 MB per launch)
        const int kk = k < n ? k : (n > 0 ? n - 1 : 0);
        int st = st0 + kk * p.st_stride;
        st = st < p.nsteps ? st : p.nsteps - 1;
        const uint32_t P = (uint32_t)st * 32u + 4u * sq;
        const uint32_t ni = fd_div(P, p.fd_hw);
        const uint32_t o = ni * (uint32_t)p.Cout_total * HW + (P - ni * HW);
        if (BNH == 2) {          // pooled gradient: {g[win 0], g[win 1], own codes of the window's upper row, of its lower row} per row and pixel quad
            const uint32_t pp = P - ni * HW;
            const uint32_t hr = fd_div(pp, p.fd_w), w = pp - hr * (uint32_t)p.W;
            const uint32_t gbase = ni * (uint32_t)p.Cout_total * (HW >> 2) + (hr >> 1) * ((uint32_t)p.W >> 1) + (w >> 1);
            const uint32_t cbase = ni * (uint32_t)p.Cout_total * HW + (hr & ~1u) * (uint32_t)p.W + w;
            S.hbit = hr & 1u;
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const float2 g2 = *reinterpret_cast<const float2*>(p.gy + (gbase + (goff[i] >> 2)));
                const uint32_t r0 = *reinterpret_cast<const uint32_t*>(p.own + (cbase + goff[i]));
                const uint32_t r1 = *reinterpret_cast<const uint32_t*>(p.own + (cbase + goff[i] + (uint32_t)p.W));
                S.gv[i] = make_float4(g2.x, g2.y, mn_u2f(r0), mn_u2f(r1));
                S.hv[i] = *reinterpret_cast<const uint32_t*>(p.h + (o + goff[i]));
            }
        } else
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            S.gv[i] = *reinterpret_cast<const float4*>(p.gy + (o + goff[i]));
            if (BNH) S.hv[i] = *reinterpret_cast<const uint32_t*>(p.h + (o + goff[i]));
        }
        if (cdo) {
            const uint32_t Pc = (uint32_t)st * 32u + 16u * chf;
            const uint32_t nc = fd_div(Pc, p.fd_hw);
            S.cv = *reinterpret_cast<const u32x4*>(p.x + (nc * (uint32_t)p.Cin_total * HW + (Pc - nc * HW) + xoff));
        }
    };
    auto commit = [&](Stage& S, int buf, bool valid) {
        unsigned char* A = lds + buf * BUF;
        unsigned char* B = A + (WG3_PRESPLIT ? 3 * PLANE : TM * WG3_RSA);
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            float v[4] = {S.gv[i].x, S.gv[i].y, S.gv[i].z, S.gv[i].w};
            if (BNH == 2) {          // the two windows' gradients go to their first +1 in scan order (else element 0) -- if that element lies in this row
                const uint32_t r0 = mn_f2u(S.gv[i].z), r1 = mn_f2u(S.gv[i].w);
#pragma unroll
                for (int e2 = 0; e2 < 2; ++e2) {
                    const bool p00 = !((r0 >> (16 * e2)) & 0x80u), p01 = !((r0 >> (16 * e2 + 8)) & 0x80u);
                    const bool p10 = !((r1 >> (16 * e2)) & 0x80u), p11 = !((r1 >> (16 * e2 + 8)) & 0x80u);
                    const uint32_t win = p00 ? 0u : (p01 ? 1u : (p10 ? 2u : (p11 ? 3u : 0u)));
                    const float ge = e2 ? S.gv[i].y : S.gv[i].x;
                    v[2 * e2] = win == S.hbit * 2u ? ge : 0.f;
                    v[2 * e2 + 1] = win == S.hbit * 2u + 1u ? ge : 0.f;
                }
            }
            if (BNH) {
                const float4 f0 = *reinterpret_cast<const float4*>(ftab + (sr + 32 * i) * 8);        // hlo, hhi, G, E1
                const float fE0 = ftab[(sr + 32 * i) * 8 + 4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float hf = (float)((S.hv[i] >> (8 * e)) & 0xffu);
                    const float dz = (hf >= f0.x && hf <= f0.y) ? v[e] : 0.f;
                    v[e] = fmaf(f0.z, dz, fmaf(f0.w, hf, fE0));
                }
            }
            if (want_db) dbacc[i] += valid ? (v[0] + v[1]) + (v[2] + v[3]) : 0.f;      // wave-uniform: layers without bias skip the 22 instructions
            if (WG3_PRESPLIT) {
                // head = the high half of the word (one v_perm packs two of them); the remainders need the masked value: 4 VALU per element + 3 per pair
                float r1[4], r2[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    r1[e] = v[e] - mn_bf16_head(v[e]);
                    r2[e] = r1[e] - mn_bf16_head(r1[e]);
                }
                unsigned char* d = A + (sr + 32 * i) * RS2 + 8 * sq;
                *reinterpret_cast<u32x2*>(d) = u32x2{mn_pack_hi16(v[0], v[1]), mn_pack_hi16(v[2], v[3])};
                *reinterpret_cast<u32x2*>(d + PLANE) = u32x2{mn_pack_hi16(r1[0], r1[1]), mn_pack_hi16(r1[2], r1[3])};
                *reinterpret_cast<u32x2*>(d + 2 * PLANE) = u32x2{mn_pack_hi16(r2[0], r2[1]), mn_pack_hi16(r2[2], r2[3])};
            } else {
                *reinterpret_cast<float4*>(A + (sr + 32 * i) * WG3_RSA + 16 * sq) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        if (cdo && XENC) {               // 16 code bytes -> 16 bf16 (exact small integers), K slot (kg, e) = pixel 8 kg + e
            u32x4 lo, hi;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t u = S.cv[q];
                const uint32_t p0 = mn_pack_bf16x2((float)(u & 0xffu), (float)((u >> 8) & 0xffu)), p1 = mn_pack_bf16x2((float)((u >> 16) & 0xffu), (float)(u >> 24));
                if (q < 2) { lo[2 * q] = p0; lo[2 * q + 1] = p1; } else { hi[2 * (q - 2)] = p0; hi[2 * (q - 2) + 1] = p1; }
            }
            const int sw = (cr >> 2) & 3;
            *reinterpret_cast<u32x4*>(B + cr * RSB + 16 * ((2 * chf) ^ sw)) = lo;
            *reinterpret_cast<u32x4*>(B + cr * RSB + 16 * ((2 * chf + 1) ^ sw)) = hi;
        } else if (cdo) {
            if (WG3_PRESPLIT) {          // K slot (kg, e) = pixel 8 kg + e: plain row order
                *reinterpret_cast<u32x4*>(B + cr * WG3_RSB + 16 * chf) = S.cv;
            } else {                     // chunk q of this half goes next to chunk q of the other half: lane (j, kg) reads its 8 codes as one b64
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<uint32_t*>(B + cr * WG3_RSB + 8 * q + 4 * chf) = S.cv[q];
            }
        }
    };
    auto contract = [&](int buf) {
        const unsigned char* A = lds + buf * BUF;
        const unsigned char* B = A + (WG3_PRESPLIT ? 3 * PLANE : TM * WG3_RSA);
        u32x4 bf[CW];
#pragma unroll
        for (int ci = 0; ci < CW; ++ci) {
            if (XENC) { bf[ci] = *reinterpret_cast<const u32x4*>(B + ((wc * CW + ci) * 16 + j) * RSB + 16 * (kg ^ ((j >> 2) & 3))); continue; }
            const u32x2 uv = *reinterpret_cast<const u32x2*>(B + ((wc * CW + ci) * 16 + j) * WG3_RSB + 8 * kg);
            // sign byte c -> bf16 +-1.0 = bytes {0x80, (c & 0x80) | 0x3F}: one and-or per 4 codes, one v_perm per 2 (the 0x80 comes from the constant operand)
            const uint32_t u = (uv[0] & 0x80808080u) | 0x3F3F3F3Fu, v = (uv[1] & 0x80808080u) | 0x3F3F3F3Fu;
            bf[ci] = u32x4{mn_perm(u, 0x80u, 0x05000400u), mn_perm(u, 0x80u, 0x07000600u), mn_perm(v, 0x80u, 0x05000400u), mn_perm(v, 0x80u, 0x07000600u)};
        }
        u32x4 a0[MW], a1[MW], a2[MW];
#pragma unroll
        for (int mi = 0; mi < MW; ++mi) {
            if (WG3_PRESPLIT) {
                const unsigned char* rowp = A + ((wm * MW + mi) * 16 + j) * RS2 + 16 * kg;
                a0[mi] = *reinterpret_cast<const u32x4*>(rowp);
                a1[mi] = *reinterpret_cast<const u32x4*>(rowp + PLANE);
                a2[mi] = *reinterpret_cast<const u32x4*>(rowp + 2 * PLANE);
                continue;
            }
            const unsigned char* row = A + ((wm * MW + mi) * 16 + j) * WG3_RSA + 16 * kg;
            const float4 ga = *reinterpret_cast<const float4*>(row), gb = *reinterpret_cast<const float4*>(row + 64);
            const float v[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
            float t0[8], t1[8], t2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                t0[e] = mn_bf16_head(v[e]);
                const float r1 = v[e] - t0[e];
                t1[e] = mn_bf16_head(r1);
                t2[e] = r1 - t1[e];
            }
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                a0[mi][d] = mn_pack_bf16x2(t0[2 * d], t0[2 * d + 1]);
                a1[mi][d] = mn_pack_bf16x2(t1[2 * d], t1[2 * d + 1]);
                a2[mi][d] = mn_pack_bf16x2(t2[2 * d], t2[2 * d + 1]);
            }
        }
        // consumer waves have the registers for all 16 fragments of a step: issue every LDS read before the first MFMA (read just in time, each
        // group of 8 MFMAs waited for its own LDS round trip: 1800 cycles per step against 770 of matrix work)
        if (SPEC) MN_SCHED_FENCE();
#pragma unroll
        for (int mi = 0; mi < MW; ++mi)
#pragma unroll
            for (int ci = 0; ci < CW; ++ci) acc[mi][ci] = mn_mfma_bf16(a0[mi], bf[ci], acc[mi][ci]);
#pragma unroll
        for (int mi = 0; mi < MW; ++mi)
#pragma unroll
            for (int ci = 0; ci < CW; ++ci) acc[mi][ci] = mn_mfma_bf16(a1[mi], bf[ci], acc[mi][ci]);
#pragma unroll
        for (int mi = 0; mi < MW; ++mi)
#pragma unroll
            for (int ci = 0; ci < CW; ++ci) acc[mi][ci] = mn_mfma_bf16(a2[mi], bf[ci], acc[mi][ci]);
    };
    if (SPEC) {
        // NS steps of operands in flight per block (registers of the producer waves): one block per CU and a loaded HBM latency of ~3 us need
        // >= 80 KB in flight per CU to stream at the HBM rate (2 steps = 40 KB gave 3.8 TB/s)
        constexpr int NS = WG3_NS;
        // Barrier k (after the producers committed step k into buffer k & 1) releases the consumers' contraction of step k; the producers overwrite
        // that buffer with step k + 2 only behind barrier k + 1, which the consumers reach after contracting step k.  The loads are issued in the same
        // stage order before the loop and inside it: the compiler's wait-count bookkeeping at the loop header then sees the same pending set on
        // both incoming edges (an asymmetric prologue made every iteration wait for the newest loads).
        if (prod) {
            Stage st[NS];
#pragma unroll
            for (int k = 0; k < NS; ++k) { fetch(st[k], k); MN_SCHED_FENCE(); }
            if (BNH) __syncthreads();
            for (int t = 0; t < n; t += NS) {
#pragma unroll
                for (int u = 0; u < NS; ++u) {
                    commit(st[u], u & 1, t + u < n);
                    fetch(st[u], t + u + NS);
                    __syncthreads();
                }
            }
        } else {
            // Consumers run one step behind: behind barrier k they ISSUE the fragment reads of step k (into register set k & 1) and then contract
            // step k - 1 from the other set while those reads are in flight.  Reading and contracting the same step in one phase made all four waves
            // of the block hit the LDS together right behind the barrier with the matrix pipe idle (1700 cycles per step for 820 of MFMA).
            struct Frag { u32x4 a[3][MWc]; u32x4 b[CW]; };
            Frag fr[2];
            auto ldfrag = [&](Frag& F, int buf) {
                const unsigned char* A = lds + buf * BUF;
                const unsigned char* B = A + 3 * PLANE;
#pragma unroll
                for (int ci = 0; ci < CW; ++ci) {
                    if (XENC) F.b[ci] = *reinterpret_cast<const u32x4*>(B + ((wc * CW + ci) * 16 + j) * RSB + 16 * (kg ^ ((j >> 2) & 3)));
                    else {          // raw sign bytes; expanded to bf16 +-1 right before the contraction
                        const u32x2 uv = *reinterpret_cast<const u32x2*>(B + ((wc * CW + ci) * 16 + j) * WG3_RSB + 8 * kg);
                        F.b[ci] = u32x4{uv[0], uv[1], 0u, 0u};
                    }
                }
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int mi = 0; mi < MWc; ++mi) F.a[pl][mi] = *reinterpret_cast<const u32x4*>(A + pl * PLANE + ((wm * MWc + mi) * 16 + j) * RS2 + 16 * kg);
            };
            auto mma = [&](Frag& F) {
                u32x4 bf[CW];
#pragma unroll
                for (int ci = 0; ci < CW; ++ci) {
                    if (XENC) { bf[ci] = F.b[ci]; continue; }
                    const uint32_t u = (F.b[ci][0] & 0x80808080u) | 0x3F3F3F3Fu, v = (F.b[ci][1] & 0x80808080u) | 0x3F3F3F3Fu;
                    bf[ci] = u32x4{mn_perm(u, 0x80u, 0x05000400u), mn_perm(u, 0x80u, 0x07000600u), mn_perm(v, 0x80u, 0x05000400u), mn_perm(v, 0x80u, 0x07000600u)};
                }
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int mi = 0; mi < MWc; ++mi)
#pragma unroll
                        for (int ci = 0; ci < CW; ++ci) acc[mi][ci] = mn_mfma_bf16(F.a[pl][mi], bf[ci], acc[mi][ci]);
            };
            if (BNH) __syncthreads();
            int kdone = -1;                  // last step whose fragments sit in registers, not yet contracted
            for (int t = 0; t < n; t += NS) {
#pragma unroll
                for (int u = 0; u < NS; ++u) {
                    __syncthreads();
                    if (t + u < n) ldfrag(fr[u & 1], u & 1);
                    MN_SCHED_FENCE();
                    if (t + u >= 1 && t + u - 1 < n) mma(fr[(u & 1) ^ 1]);
                }
                kdone = t + NS - 1;
            }
            if (kdone >= 0 && kdone < n) mma(fr[1]);          // n a multiple of NS: the last step (odd index) is still pending
        }
    } else {
    fetch(s0, 0);
    fetch(s1, 1);
    if (BNH) __syncthreads();                 // the fold table
    commit(s0, 0, n > 0);
    fetch(s0, 2);
    __syncthreads();
    for (int t = 0; t < n; t += 2) {
        commit(s1, 1, t + 1 < n);
        fetch(s1, t + 3);
        contract(0);
        __syncthreads();
        commit(s0, 0, t + 2 < n);
        fetch(s0, t + 4);
        if (t + 1 < n) contract(1);
        __syncthreads();
    }
    }
    if (SPEC && MW == 4) {
        // partial tile through LDS: a store instruction then covers four whole 256-byte rows.  (Straight from the accumulator layout a lane holds
        // four ROWS of one column: 64-byte pieces, and every half-written 128-byte line cost a fetch -- 17 MB per launch, 8 ... 24 % of the reads.)
        __syncthreads();                                   // the staging buffers are free
        if (cons) {
            constexpr int ERS = 16 * CW + 4;               // floats per staged row
            float* T = reinterpret_cast<float*>(lds) + cwv * (16 * MWc) * ERS;
#pragma unroll
            for (int mi = 0; mi < MWc; ++mi)
#pragma unroll
                for (int ci = 0; ci < CW; ++ci)
#pragma unroll
                    for (int r = 0; r < 4; ++r) T[(mi * 16 + kg * 4 + r) * ERS + ci * 16 + j] = acc[mi][ci][r];
            MN_WAVE_SYNC();
            float* dst = p.part + (((int64_t)z * p.G + g) * p.Mgw + mb * TM + wm * (16 * MWc)) * p.Cgw + cb * TC + wc * (16 * CW);
#pragma unroll
            for (int it = 0; it < (16 * MWc) / 4; ++it) {
                const int row = it * 4 + (lane >> 4), col = 4 * (lane & 15);
                *reinterpret_cast<float4*>(dst + (int64_t)row * p.Cgw + col) = *reinterpret_cast<const float4*>(T + row * ERS + col);
            }
        }
    } else if (cons)
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
    if (p.want_db && cb == 0 && prod) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            float v = dbacc[i];
            v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);     // the 8 pixel chunks of the row
            if (sq == 0) p.dbpart[((int64_t)z * p.G + g) * p.Mgw + mb * TM + sr + 32 * i] = v;
        }
    }
}
static int pws_geom_ok(const mn_conv_geom* g);
struct Wg2Plan { Wg2Params p; int MW, CW8, staged, spec; int grid; int64_t off_db, ws_bytes; };
static int plan_pws_wgrad(const mn_conv_geom* g, Wg2Plan* pl) {
    if (!pws_geom_ok(g)) return 0;
    const int Cg = g->C / g->groups, Mg = g->O / g->groups;
    const int64_t NP = (int64_t)g->N * g->H * g->W;
    if (NP % 32 || Mg < 33 || Cg < 33) return 0;           // small tiles stay on the LDS-staged kernel
    Wg2Params& p = pl->p;
    pl->MW = (Mg > 64 || Cg > 64) ? 4 : 2;
    pl->CW8 = 0;          // (the 4 x 1 arrangement of 32 x 128 wave tiles, round 2: slower; its instantiations are gone)
    const int T = 32 * pl->MW;
    p.N = g->N; p.HW = g->H * g->W; p.G = g->groups; p.Cin_total = g->C; p.Cout_total = g->O; p.Cg = Cg; p.Mg = Mg;
    p.in_map = make_chanmap(g->in_shuffle, g->C);
    p.nmb = (Mg + T - 1) / T; p.ncb = (Cg + T - 1) / T;
    p.Mgw = p.nmb * T; p.Cgw = p.ncb * T;
    p.nsteps = (int)(NP / 32);
    const int base = p.G * p.nmb * p.ncb;
    // LDS-staged kernel: 16-byte code loads need HW % 16 == 0 (else the direct-load kernel)
    pl->staged = p.HW % 16 == 0;
    pl->spec = pl->staged && pl->MW == 4 ? 2 : 0;    // wave-specialised variant (4 staging + 8 MFMA waves): 768 threads, one block per CU
    int Z = (pl->spec ? 256 : 512) / base;
    // every block pays a fixed price (pipeline fill, a 64 KB partial tile written and reduced again): keep >= 32 steps per block as long
    // as there is still one block per CU (measured: L5 67 -> 58 us, L8 40 -> 36 us)
    while (Z > 1 && p.nsteps / Z < 32 && base * Z > 256) Z /= 2;
    if (Z > p.nsteps / 2) Z = p.nsteps / 2;
    if (Z < 1) Z = 1;
    p.Z = Z;
    p.st_per_z = (p.nsteps + Z - 1) / Z; p.st_stride = 1;     // contiguous pixel ranges: each block streams its gy rows sequentially
    p.fd_hw = make_fastdiv((uint32_t)p.HW);
    const int64_t nb = (int64_t)base * Z;
    if (nb > 0x7fffffff) return 0;
    pl->grid = (int)nb;
    const int64_t part_bytes = (int64_t)Z * p.G * p.Mgw * p.Cgw * 4;
    pl->off_db = (part_bytes + 255) / 256 * 256;
    pl->ws_bytes = pl->off_db + (int64_t)Z * p.G * p.Mgw * 4;
    return 1;
}
int pws_wgrad_supported(const mn_conv_geom* g) { Wg2Plan pl; return plan_pws_wgrad(g, &pl); }
int pws_wgrad_staged(const mn_conv_geom* g) { Wg2Plan pl; return plan_pws_wgrad(g, &pl) && pl.staged; }
int64_t pws_wgrad_ws_bytes(const mn_conv_geom* g) { Wg2Plan pl; return plan_pws_wgrad(g, &pl) ? pl.ws_bytes : 0; }
int pws_bwd_weight(const mn_conv_geom* g, const float* gy, const int8_t* x, float* dw, float* dbias, void* ws, int64_t ws_bytes, hipStream_t s) {
    return pws_bwd_weight_bnh(g, gy, nullptr, nullptr, nullptr, 0, x, dw, dbias, ws, ws_bytes, s, nullptr);
}
int pws_bwd_weight_bnh(const mn_conv_geom* g, const float* gy, const uint8_t* h, const float* chan, const float* sums, int training, const int8_t* x,
                       float* dw, float* dbias, void* ws, int64_t ws_bytes, hipStream_t s, const int8_t* own) {
    Wg2Plan pl;
    if (!plan_pws_wgrad(g, &pl) || (own ? (((uintptr_t)gy) & 7) != 0 : !aligned16(gy)) || (((uintptr_t)x) & 3)) MN_FAIL(MN_ENOTSUP, "mn_conv2d_bwd_weight(sign): geometry not covered");
    if (own && (!h || !pl.staged || (((uintptr_t)x) & 15) || (((uintptr_t)own) & 3) || (g->H & 1) || (g->W & 3)))
        MN_FAIL(MN_ENOTSUP, "mn_conv2d_bwd_weight_bnh_pool: needs the LDS-staged backward-weight kernel, even H, W %% 4 == 0");
    if (!ws || ws_bytes < pl.ws_bytes || !aligned16(ws)) MN_FAIL(MN_ENOSPC, "mn_conv2d_bwd_weight(sign): workspace too small");
    Wg2Params& p = pl.p;
    p.gy = gy; p.x = (const char*)x; p.part = (float*)ws; p.dbpart = (float*)((char*)ws + pl.off_db); p.want_db = dbias != nullptr;
    p.h = h; p.chan = chan; p.sums = sums; p.training = training; p.n_f = (float)g->N * (float)(g->H * g->W);
    p.own = (const char*)own; p.W = (int)g->W; p.fd_w = make_fastdiv((uint32_t)g->W);
    if (h && (!chan || !sums || (((uintptr_t)h) & 3))) MN_FAIL(MN_EINVAL, "mn_conv2d_bwd_weight_bnh: null / misaligned argument");
    if (pl.staged && ((((uintptr_t)x) & 15) || (h && (((uintptr_t)h) & 3)))) pl.staged = 0;
    if (pl.staged) mn_set_last_kernel(pl.spec == 2 ? "k_pws_wgrad_s<%d, %d, 0, 2>" : "k_pws_wgrad_s<%d, %d>", pl.MW, own ? 2 : (h ? 1 : 0));
    else mn_set_last_kernel("k_pws_wgrad<%d, %d, %d>", pl.MW, pl.MW, h ? 1 : 0);
    { const double nx = (double)g->N * g->C * g->H * g->W, ny = (double)g->N * g->O * g->H * g->W; mn_prof_bytes((own ? 3.0 : (h ? 5.0 : 4.0)) * ny + nx); }
    mn_prof_begin(s);
    if (pl.staged) {
        const int TMs = 32 * pl.MW;
        const size_t buf = WG3_PRESPLIT ? (size_t)3 * TMs * 80 + (size_t)TMs * WG3_RSB : (size_t)TMs * WG3_RSA + (size_t)TMs * WG3_RSB;
        const size_t ldsb = 2 * buf + (p.h ? (size_t)TMs * 32 : 0);
#define WG3_LAUNCH(MWV, BV) { raise_lds_limit((const void*)k_pws_wgrad_s<MWV, BV>, ldsb); hipLaunchKernelGGL((k_pws_wgrad_s<MWV, BV>), dim3(pl.grid), dim3(256), ldsb, s, p); }
#define WG3_LAUNCH_SPEC(BV, SV) { raise_lds_limit((const void*)k_pws_wgrad_s<4, BV, 0, SV>, ldsb); hipLaunchKernelGGL((k_pws_wgrad_s<4, BV, 0, SV>), dim3(pl.grid), dim3(256 + 256 * SV), ldsb, s, p); }
        if (pl.spec == 2) { if (own) WG3_LAUNCH_SPEC(2, 2) else if (p.h) WG3_LAUNCH_SPEC(1, 2) else WG3_LAUNCH_SPEC(0, 2) }
        else if (own) { if (pl.MW == 4) WG3_LAUNCH(4, 2) else WG3_LAUNCH(2, 2) }
        else if (p.h) { if (pl.MW == 4) WG3_LAUNCH(4, 1) else WG3_LAUNCH(2, 1) }
        else { if (pl.MW == 4) WG3_LAUNCH(4, 0) else WG3_LAUNCH(2, 0) }
#undef WG3_LAUNCH
#undef WG3_LAUNCH_SPEC
    } else if (p.h) {
        if (pl.MW == 4) hipLaunchKernelGGL((k_pws_wgrad<4, 4, 1>), dim3(pl.grid), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((k_pws_wgrad<2, 2, 1>), dim3(pl.grid), dim3(256), 0, s, p);
    } else {
        if (pl.MW == 4) hipLaunchKernelGGL((k_pws_wgrad<4, 4, 0>), dim3(pl.grid), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((k_pws_wgrad<2, 2, 0>), dim3(pl.grid), dim3(256), 0, s, p);
    }
    mn_prof_end(s);
    qg_launch_wgrad_reduce(p.part, p.dbpart, dw, dbias, p.Z, p.G, p.Mg, p.Cg, p.Mgw, p.Cgw, 1.f, nullptr, s);
    MN_CHECK_LAUNCH("mn_conv2d_bwd_weight(sign)");
    return MN_OK;
}

// pointwise backward-weight on k-bit activation codes (bytes; dw = s * sum gy * j): the LDS-staged kernel only
int pws_wgrad_code8_supported(const mn_conv_geom* g) { Wg2Plan pl; return plan_pws_wgrad(g, &pl) && pl.staged; }
int pws_bwd_weight_code8(const mn_conv_geom* g, const float* gy, const uint8_t* x, float ascale, float* dw, float* dbias, void* ws, int64_t ws_bytes, hipStream_t s) {
    Wg2Plan pl;
    if (!plan_pws_wgrad(g, &pl) || !pl.staged || !aligned16(gy) || (((uintptr_t)x) & 15)) MN_FAIL(MN_ENOTSUP, "mn_conv2d_bwd_weight(code8): geometry not covered");
    if (!ws || ws_bytes < pl.ws_bytes || !aligned16(ws)) MN_FAIL(MN_ENOSPC, "mn_conv2d_bwd_weight(code8): workspace too small");
    Wg2Params& p = pl.p;
    p.gy = gy; p.x = (const char*)x; p.part = (float*)ws; p.dbpart = (float*)((char*)ws + pl.off_db); p.want_db = dbias != nullptr;
    p.h = nullptr; p.chan = nullptr; p.sums = nullptr; p.training = 0; p.n_f = 1.f;
    mn_set_last_kernel(pl.spec == 2 ? "k_pws_wgrad_s<%d, 0, 1, 2>" : "k_pws_wgrad_s<%d, 0, 1>", pl.MW);
    { const double nx = (double)g->N * g->C * g->H * g->W, ny = (double)g->N * g->O * g->H * g->W; mn_prof_bytes(4.0 * ny + nx); }
    mn_prof_begin(s);
    const int TMs = 32 * pl.MW;
    const size_t ldsb = 2 * ((size_t)3 * TMs * 80 + (size_t)TMs * WG3_RSBX);
    if (pl.spec == 2) { raise_lds_limit((const void*)k_pws_wgrad_s<4, 0, 1, 2>, ldsb); hipLaunchKernelGGL((k_pws_wgrad_s<4, 0, 1, 2>), dim3(pl.grid), dim3(768), ldsb, s, p); }
    else if (pl.MW == 4) { raise_lds_limit((const void*)k_pws_wgrad_s<4, 0, 1>, ldsb); hipLaunchKernelGGL((k_pws_wgrad_s<4, 0, 1>), dim3(pl.grid), dim3(256), ldsb, s, p); }
    else { raise_lds_limit((const void*)k_pws_wgrad_s<2, 0, 1>, ldsb); hipLaunchKernelGGL((k_pws_wgrad_s<2, 0, 1>), dim3(pl.grid), dim3(256), ldsb, s, p); }
    mn_prof_end(s);
    qg_launch_wgrad_reduce(p.part, p.dbpart, dw, dbias, p.Z, p.G, p.Mg, p.Cg, p.Mgw, p.Cgw, ascale, nullptr, s);
    MN_CHECK_LAUNCH("mn_conv2d_bwd_weight(code8)");
    return MN_OK;
}

// ------------------------------------------------------------------------------------------------ host side
struct PwsPlan {
    PwsParams p;
    PackParams pk;
    int NT, KS;
    size_t lds;
    int grid, pack_grid;
    int64_t off_scale, off_part, off_sums, off_chan, ws_bytes;
};
static int pws_geom_ok(const mn_conv_geom* g) {
    if (g->KH != 1 || g->KW != 1 || g->stride_h != 1 || g->stride_w != 1 || g->pad_h != 0 || g->pad_w != 0) return 0;
    const int64_t HW = (int64_t)g->H * g->W, NP = (int64_t)g->N * HW;
    if (HW % 4) return 0;
    if (NP * HW >= ((int64_t)1 << 32) || NP + 256 >= ((int64_t)1 << 31)) return 0;   // FastDiv range
    if (4 * NP * (g->C > g->O ? g->C : g->O) >= ((int64_t)1 << 32)) return 0;        // 32-bit byte offsets
    if (g->in_shuffle > 1 && g->C % g->in_shuffle) return 0;
    return 1;
}
// nt_max: largest tile count per wave (8 = a whole 128-channel group; the backward epilogues also hold gradient rows)
static int plan_pws(const mn_conv_geom* g, int nt_max, PwsPlan* pl) {
    if (!pws_geom_ok(g)) return 0;
    const int Cg = g->C / g->groups, Mg = g->O / g->groups;
    PwsParams& p = pl->p;
    p.N = g->N; p.HW = g->H * g->W; p.G = g->groups; p.NP = (uint32_t)((int64_t)g->N * p.HW);
    p.Cin_total = g->C; p.Cout_total = g->O; p.Kc = Cg; p.Mr = Mg;
    p.in_map = make_chanmap(g->in_shuffle, g->C);
    p.Kp = qg_roundup(Cg, 32);
    const int KS = p.Kp / 32;
    if (KS < 1 || KS > 4) return 0;                       // up to 128 input channels per group
    pl->KS = KS;
    int NT = nt_max;
    while (NT > 1 && 16 * (NT / 2) >= Mg) NT /= 2;
    pl->NT = NT;
    const int MB = 16 * NT;
    pl->lds = (size_t)MB * (p.Kp + 8) * 2 + (size_t)8 * MB * 4 + (size_t)p.Kp * 4 + (size_t)4 * MB * 2 * 8;
    if (pl->lds < (size_t)NT * 8 * 4 * 64 * 4) pl->lds = (size_t)NT * 8 * 4 * 64 * 4;      // the block reduction's [value][wave][lane] image reuses the region
    p.num_mblk = (Mg + MB - 1) / MB;
    p.Mpad = ((Mg + 127) / 128) * 128;                    // one packed-code layout for every tile height
    p.nchunks = (int)((p.NP + 63) / 64);
    int CB = (p.nchunks + 3) / 4;
    const int capb = 512;           // one round of 2 blocks per CU: every block stages its weights and ends in a block reduction -- fewer, longer blocks
                              // (measured against 1024: STATS 41 -> 33 us on L2, 32 -> 23 on L5, 26 -> 18 on L8; SIGN8 44 -> 38, 32 -> 24, 19 -> 16)
    const int cap = capb / (p.G * p.num_mblk) > 0 ? capb / (p.G * p.num_mblk) : 1;
    if (CB > cap) CB = cap;
    p.CB = CB;
    p.fd_hw = make_fastdiv((uint32_t)p.HW);
    p.n_f = (float)g->N * (float)p.HW;
    const int64_t nb = (int64_t)qg_roundup(p.G * CB, 8) * p.num_mblk;
    if (nb > 0x7fffffff) return 0;
    pl->grid = (int)nb;
    const int64_t code_bytes = (int64_t)p.G * p.Mpad * p.Kp * 2;
    pl->off_scale = (code_bytes + 255) / 256 * 256;
    pl->off_part = pl->off_scale + ((int64_t)p.G * p.Mpad * 4 + 255) / 256 * 256;
    pl->off_sums = pl->off_part + ((int64_t)2048 * p.Mpad * 2 * 8 + 255) / 256 * 256;   // CB * G <= 2048 rows of [Mpad][2] doubles
    pl->off_chan = pl->off_sums + ((int64_t)2 * g->O * 4 + 255) / 256 * 256;
    pl->ws_bytes = pl->off_chan + (int64_t)PWS_NCH * g->O * 4 + 256;
    PackParams& k = pl->pk;
    k.G = g->groups; k.Mg = Mg; k.Cg = Cg; k.T = 1; k.KW = 1; k.transpose = 0;
    k.Mpad = p.Mpad; k.Cgp = p.Kp; k.Cpad = 0; k.Mgp = 0;
    pl->pack_grid = k.G * k.Mpad;
    return 1;
}

template <int NT, int KS, int EPI, int XENC>
static void launch_pws3(const PwsPlan& pl, hipStream_t s) {
    raise_lds_limit((const void*)k_pws<NT, KS, EPI, XENC>, pl.lds);
    hipLaunchKernelGGL((k_pws<NT, KS, EPI, XENC>), dim3(pl.grid), dim3(256), pl.lds, s, pl.p);
}
template <int NT, int EPI, int XENC>
static void launch_pws2(const PwsPlan& pl, hipStream_t s) {
    if (pl.KS == 4) launch_pws3<NT, 4, EPI, XENC>(pl, s);
    else if (pl.KS == 3) launch_pws3<NT, 3, EPI, XENC>(pl, s);
    else if (pl.KS == 2) launch_pws3<NT, 2, EPI, XENC>(pl, s);
    else launch_pws3<NT, 1, EPI, XENC>(pl, s);
}
template <int EPI, int XENC = 0>
static int launch_pws(const PwsPlan& pl, hipStream_t s, double nbytes, const char* what) {
    // the name rocprofv3 prints: EPI 0 Y, 1 STATS, 2 SIGN8, 3 BWD_PART, 4 BWD_APPLY, 5 BWD_PART_POOL, 6 BWD_APPLY_POOL; XENC 1 = k-bit activation codes
    if (XENC) mn_set_last_kernel("k_pws<%d, %d, %d, %d>", pl.NT, pl.KS, EPI, XENC);
    else mn_set_last_kernel("k_pws<%d, %d, %d>", pl.NT, pl.KS, EPI);
    mn_prof_bytes(nbytes);
    mn_prof_begin(s);
    switch (pl.NT) {
        case 1: launch_pws2<1, EPI, XENC>(pl, s); break;
        case 2: launch_pws2<2, EPI, XENC>(pl, s); break;
        case 4: launch_pws2<4, EPI, XENC>(pl, s); break;
        case 8: if (EPI >= PWS_BWD_PART) MN_FAIL(MN_EINVAL, "%s: bad NT", what); launch_pws2<(EPI >= PWS_BWD_PART) ? 4 : 8, EPI, XENC>(pl, s); break;
        default: MN_FAIL(MN_EINVAL, "%s: bad NT", what);
    }
    mn_prof_end(s);
    MN_CHECK_LAUNCH(what);
    return MN_OK;
}
#define PWS_NT_FWD 4      // NT = 8 (a whole 128-channel group per wave) spills: 2 x 64 accumulators + statistics do not fit 256 VGPRs
#define PWS_NT_BWD 4

static int pws_prepare(const mn_conv_geom* g, const mn_wq* wq, const int8_t* x, const float* w, void* ws, int64_t ws_bytes, int nt_max, PwsPlan* pl,
                       hipStream_t s, const char* what) {
    if (!wq_codeable(wq) || !plan_pws(g, nt_max, pl)) MN_FAIL(MN_ENOTSUP, "%s: geometry / weight quantizer not covered by the fused sign kernels", what);
    if (!x || !w || (((uintptr_t)x) & 3)) MN_FAIL(MN_EINVAL, "%s: null / misaligned tensor", what);
    if (!ws || ws_bytes < pl->ws_bytes || !aligned16(ws)) MN_FAIL(MN_ENOSPC, "%s: workspace too small (%lld < %lld)", what, (long long)ws_bytes, (long long)pl->ws_bytes);
    if (wq->packed_fwd && mn_use_packed()) {          // the step's pre-packed image (mn_qg_pack_multi: [codes | row scales at off_scale] of this plan)
        pl->pk.codes = (uint16_t*)const_cast<void*>(wq->packed_fwd);
        pl->pk.scale_out = (float*)((char*)const_cast<void*>(wq->packed_fwd) + pl->off_scale);
    } else {
        fill_pack(pl->pk, wq, w, ws, 0, pl->off_scale);
        qg_launch_pack(pl->pk, pl->pack_grid, s);
    }
    PwsParams& p = pl->p;
    p.x = (const char*)x; p.wc = pl->pk.codes; p.rowscale = pl->pk.scale_out;
    p.part = (float*)((char*)ws + pl->off_part);
    p.chan = (const float*)((char*)ws + pl->off_chan);
    p.y = nullptr; p.a8 = nullptr; p.h8 = nullptr; p.h16 = nullptr; p.h32 = nullptr; p.ascale = 1.f; p.da = nullptr; p.sums = nullptr; p.training = 1; p.bias = nullptr; p.own = nullptr;
    p.W = g->W; p.fd_w = make_fastdiv((uint32_t)g->W);
    return MN_OK;
}

int pws_pack_plan(const mn_conv_geom* g, PackParams* pk, int* grid, int64_t* off_scale, int64_t* bytes) {
    PwsPlan pl;
    if (!plan_pws(g, PWS_NT_FWD, &pl)) return 0;
    *pk = pl.pk; *grid = pl.pack_grid; *off_scale = pl.off_scale; *bytes = pl.ws_bytes;
    return 1;
}
int pws_supported(const mn_conv_geom* g, const mn_wq* wq) {
    PwsPlan pl;
    return wq_codeable(wq) && plan_pws(g, PWS_NT_FWD, &pl);
}
// plain forward on sign codes (used by mn_conv2d_fwd for MN_ACTQ_SIGN8 inputs)
int pws_fwd(const mn_conv_geom* g, const mn_wq* wq, const int8_t* x, const float* w, const float* bias, float* y, void* ws, int64_t ws_bytes, hipStream_t s) {
    PwsPlan pl;
    int rc = pws_prepare(g, wq, x, w, ws, ws_bytes, PWS_NT_FWD, &pl, s, "mn_conv2d_fwd(sign)");
    if (rc) return rc;
    if (!y || !aligned16(y)) MN_FAIL(MN_EINVAL, "mn_conv2d_fwd(sign): bad output");
    pl.p.y = y; pl.p.bias = bias;
    const double nx = (double)g->N * g->C * g->H * g->W, ny = (double)g->N * g->O * g->H * g->W;
    return launch_pws<PWS_Y>(pl, s, nx + 4.0 * ny, "mn_conv2d_fwd(sign)");
}
int64_t pws_ws_bytes(const mn_conv_geom* g) {
    PwsPlan pl;
    return plan_pws(g, PWS_NT_FWD, &pl) ? pl.ws_bytes : 0;
}

// the fused BatchNorm epilogues rely on |acc| <= Cin/groups: weight codes in {-1, 0, +1} (ternary / binary weights)
static int pws_bn_ok(const mn_conv_geom* g, const mn_wq* wq) { return g && wq && wq->mode == MN_WQ_TERNARY && pws_supported(g, wq); }
extern "C" int mn_qconv_bnsign_supported(const mn_conv_geom* g, const mn_wq* wq) { return pws_bn_ok(g, wq); }
extern "C" int64_t mn_qconv_bnsign_ws_bytes(const mn_conv_geom* g) { return g ? pws_ws_bytes(g) : -1; }

struct HsPrep;
static int h_sign_launch(int64_t N, int64_t O, int64_t H, int64_t W, const uint8_t* h, const float* chan, int8_t* a, hipStream_t s, const HsPrep* fold = nullptr,
                         int8_t* a_pool = nullptr);
// k_pws_stats_prep's work runs inside the streaming sign pass (k_h_sign_prep; bit-identical; round 5 same-box A/B on c2: 107.2 / 107.7 k -> 108.8 / 108.7 k img/s).
// MN_HSIGN_FOLD=0 restores the two launches.
static bool hsign_fold_enabled() { const char* e = MN_ENV("MN_HSIGN_FOLD"); return !(e && e[0] == '0'); }
static void pws_chan_prep(PwsPlan& pl, const mn_conv_geom* g, const float* bias, const float* save, const float* gamma, const float* beta, hipStream_t s) {
    hipLaunchKernelGGL(k_pws_chan_prep, dim3((unsigned)g->O), dim3(64), 0, s, pl.p.Kc, pl.p.Kp, pl.p.wc, pl.p.G, pl.p.Mpad, pl.p.Mr, pl.p.rowscale, bias, save, gamma, beta,
                       (float*)pl.p.chan, (int)g->O);
}

static int qconv_bnsign_fwd_impl(const mn_conv_geom* g, const mn_wq* wq, const int8_t* x, const float* w, const float* bias, const float* gamma,
                                 const float* beta, float eps, float momentum, int training, float* running_mean, float* running_var, int64_t* nbt,
                                 float* save, int8_t* a, uint8_t* h, float* chan_out, void* ws, int64_t ws_bytes, mn_stream_t stream, int8_t* a_pool = nullptr) {
    if (!g || !gamma || !beta || !save || !a || (((uintptr_t)a) & 3)) MN_FAIL(MN_EINVAL, "mn_qconv_bnsign_fwd: null / misaligned argument");
    if (!pws_bn_ok(g, wq)) MN_FAIL(MN_ENOTSUP, "mn_qconv_bnsign_fwd: needs a pointwise convolution with ternary / binary weights");
    if (!training && (!running_mean || !running_var)) MN_FAIL(MN_EINVAL, "mn_qconv_bnsign_fwd: eval mode needs the running statistics");
    hipStream_t s = (hipStream_t)stream;
    PwsPlan pl;
    int rc = pws_prepare(g, wq, x, w, ws, ws_bytes, PWS_NT_FWD, &pl, s, "mn_qconv_bnsign_fwd");
    if (rc) return rc;
    PwsParams& p = pl.p;
    p.bias = bias;
    const double nx = (double)g->N * g->C * g->H * g->W, ny = (double)g->N * g->O * g->H * g->W;
    const bool stash_in_stats = training && h;      // one MFMA pass instead of two: h from the statistics pass, sign streamed from h
    p.h8 = stash_in_stats ? h : nullptr;
    if (training && (rc = launch_pws<PWS_STATS>(pl, s, nx + (stash_in_stats ? ny : 0.0), "mn_qconv_bnsign_fwd(stats)"))) return rc;
    if (chan_out) p.chan = chan_out;                     // caller-owned [8][O]: kept for the streaming backward (mn_bnh_bwd)
    if (stash_in_stats && hsign_fold_enabled()) {
        const HsPrep q{(const double*)p.part, p.CB, p.G, p.Mpad, p.Mr, p.rowscale, bias, (double)g->N * p.HW, eps, momentum, training, running_mean, running_var, save,
                       (int)g->O, p.Kc, p.Kp, p.wc, gamma, beta, (float*)p.chan, (const float*)nullptr, (long long*)nbt};
        return h_sign_launch(g->N, g->O, g->H, g->W, h, (const float*)p.chan, a, s, &q, a_pool);
    }
    if (a_pool) MN_FAIL(MN_ENOTSUP, "mn_qconv_bnsign_fwd_stash_pool: only the training-mode stash forward with the folded sign pass writes pooled codes");
    hipLaunchKernelGGL(k_pws_stats_prep, dim3((unsigned)g->O), dim3(64), 0, s, (const double*)p.part, p.CB, p.G, p.Mpad, p.Mr, p.rowscale, bias,
                       (double)g->N * p.HW, eps, momentum, training, running_mean, running_var, save, (int)g->O, p.Kc, p.Kp, p.wc, gamma, beta,
                       (float*)p.chan, (const float*)nullptr, (long long*)nbt);
    if (stash_in_stats) return h_sign_launch(g->N, g->O, g->H, g->W, h, (const float*)p.chan, a, s);
    p.a8 = (char*)a; p.h8 = h;
    return launch_pws<PWS_SIGN8>(pl, s, nx + (h ? 2.0 : 1.0) * ny, "mn_qconv_bnsign_fwd(sign)");
}
extern "C" int mn_qconv_bnsign_fwd(const mn_conv_geom* g, const mn_wq* wq, const int8_t* x, const float* w, const float* bias, const float* gamma,
                                   const float* beta, float eps, float momentum, int training, float* running_mean, float* running_var,
                                   float* save, int8_t* a, void* ws, int64_t ws_bytes, mn_stream_t stream) {
    return qconv_bnsign_fwd_impl(g, wq, x, w, bias, gamma, beta, eps, momentum, training, running_mean, running_var, nullptr, save, a, nullptr, nullptr, ws, ws_bytes, stream);
}
static int qconv_kxk_bnsign_fwd_stash(const mn_conv_geom* g, const mn_wq* wq, const int8_t* x, const float* w, const float* bias, const float* gamma,
                                      const float* beta, float eps, float momentum, int training, float* running_mean, float* running_var, int64_t* nbt,
                                      float* save, int8_t* a, uint8_t* h, float* chan, void* ws, int64_t ws_bytes, hipStream_t s);
extern "C" int mn_qconv_bnsign_fwd_stash(const mn_conv_geom* g, const mn_wq* wq, const int8_t* x, const float* w, const float* bias, const float* gamma,
                                         const float* beta, float eps, float momentum, int training, float* running_mean, float* running_var,
                                         int64_t* num_batches_tracked, float* save, int8_t* a, uint8_t* h, float* chan, void* ws, int64_t ws_bytes,
                                         mn_stream_t stream) {
    if (!h || !chan || (((uintptr_t)h) & 3)) MN_FAIL(MN_EINVAL, "mn_qconv_bnsign_fwd_stash: null / misaligned stash");
    if (g && wq && !pws_bn_ok(g, wq) && kk_h8_supported(g, wq))          // a k x k convolution: stash written by the conv kernel, statistics / sign streamed from it
        return qconv_kxk_bnsign_fwd_stash(g, wq, x, w, bias, gamma, beta, eps, momentum, training, running_mean, running_var, num_batches_tracked, save, a, h,
                                          chan, ws, ws_bytes, (hipStream_t)stream);
    return qconv_bnsign_fwd_impl(g, wq, x, w, bias, gamma, beta, eps, momentum, training, running_mean, running_var, num_batches_tracked, save, a, h, chan, ws,
                                 ws_bytes, stream);
}

/* the same forward for a pointwise block whose output goes through a 2x2 / stride-2 max-pool: a_pool [N][O][H/2][W/2] receives the pooled sign codes from the sign pass
 * itself (what mn_maxpool2x2_sign8_fwd would compute from `a`) */
extern "C" int mn_qconv_bnsign_fwd_stash_pool_supported(const mn_conv_geom* g, const mn_wq* wq) {
    return g && wq && pws_bn_ok(g, wq) && hsign_fold_enabled() && !(g->H & 1) && !(g->W & 15) && (g->H * g->W) % 16 == 0;
}
extern "C" int mn_qconv_bnsign_fwd_stash_pool(const mn_conv_geom* g, const mn_wq* wq, const int8_t* x, const float* w, const float* bias, const float* gamma,
                                              const float* beta, float eps, float momentum, int training, float* running_mean, float* running_var,
                                              int64_t* num_batches_tracked, float* save, int8_t* a, int8_t* a_pool, uint8_t* h, float* chan, void* ws, int64_t ws_bytes,
                                              mn_stream_t stream) {
    if (!h || !chan || !a_pool || (((uintptr_t)h) & 15) || (((uintptr_t)a) & 15)) MN_FAIL(MN_EINVAL, "mn_qconv_bnsign_fwd_stash_pool: null / misaligned tensor");
    if (!training || !mn_qconv_bnsign_fwd_stash_pool_supported(g, wq)) MN_FAIL(MN_ENOTSUP, "mn_qconv_bnsign_fwd_stash_pool: geometry / mode not covered");
    return qconv_bnsign_fwd_impl(g, wq, x, w, bias, gamma, beta, eps, momentum, training, running_mean, running_var, num_batches_tracked, save, a, h, chan, ws,
                                 ws_bytes, stream, a_pool);
}

// ------------------------------------------------------------------------------------------------
// The stashed block around a k x k convolution (nin_gc's grouped 3x3 layers): the code-domain k x k kernel writes the byte stash
// h = (acc + nnz[o]) / 2 instead of y (QG_EPI_H8, qgemm_kxk.hip); batch statistics and the sign are then two streaming passes over
// ONE byte per element -- k_h_stats (exact integer sums of acc and acc^2, partials in the layout k_pws_final_fwd reads) and k_h_sign
// (the integer threshold test of k_pws_chan_prep) -- and the backward is the same mn_bnh_bwd_sums / mn_bnh_bwd_apply as for the
// pointwise blocks.  fp32 y is never written or read: 1 + 1 + 2 bytes per element instead of 4 + 4 + 5.
// nnz9[3 rc + cc][o]: non-zero weights of row o among the taps that lie inside the image for a pixel of row class rc (0 top: r >= 1,
// 1 middle, 2 bottom: r <= 1) and column class cc (same with s) -- 3 x 3, padding 1.  One wave per output channel.
__global__ __launch_bounds__(64) void k_row_nnz9(const float* __restrict__ w, int Cg, int O, float* __restrict__ nnz9, float* __restrict__ alpha) {
    const int o = blockIdx.x, lane = threadIdx.x;
    int cnt[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) cnt[k] = 0;
    float amax = 0.f;                          // alpha[o]: the magnitude of the row's non-zero weights (ternary / binary rows)
    for (int k = lane; k < Cg * 9; k += 64) {
        const int tap = k % 9, r = tap / 3, s_ = tap - 3 * r;
        const float wv = w[(int64_t)o * Cg * 9 + k];
        amax = fmaxf(amax, fabsf(wv));
        const int nzv = wv != 0.f;
#pragma unroll
        for (int rc = 0; rc < 3; ++rc)
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                const bool in = (rc != 0 || r >= 1) && (rc != 2 || r <= 1) && (cc != 0 || s_ >= 1) && (cc != 2 || s_ <= 1);
                cnt[rc * 3 + cc] += (in && nzv) ? 1 : 0;
            }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        int c = cnt[k];
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) c += __shfl_xor(c, sft, 64);
        if (lane == 0) nnz9[(int64_t)k * O + o] = (float)c;
    }
    if (alpha) {
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) amax = fmaxf(amax, __shfl_xor(amax, sft, 64));
        if (lane == 0) alpha[o] = amax;
    }
}
// chan row 7 = -1: "the nnz of this block is per pixel class, rows 8..16" (stash_nnz_load, common.h)
__global__ void k_chan_mark_classes(float* __restrict__ chan, const float* __restrict__ nnz9, int O) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= O) return;
    chan[7 * O + o] = -1.f;
    for (int k = 0; k < 9; ++k) chan[(8 + k) * O + o] = nnz9[k * O + o];
}
struct HGeom { int C, H, W4, HW, HW4, Mr, Mpad, G; FastDiv fd_hw4, fd_w4, fd_hwv; int64_t n4; };     // fd_hwv: HW4 / VEC of the launch
// VEC quads (4 VEC bytes) per thread and iteration: one b32 / b128 load
template <int VEC>
__global__ __launch_bounds__(256) void k_h_stats(const HGeom g, const unsigned char* __restrict__ h, const float* __restrict__ nnz9, double* __restrict__ part) {
    __shared__ double scd[16];
    const int c = blockIdx.x, sp = blockIdx.y, S = gridDim.y;
    StashNnz z;
    z.v0 = nnz9[c]; z.v1 = nnz9[g.C + c]; z.v2 = nnz9[2 * g.C + c]; z.v3 = nnz9[3 * g.C + c]; z.v4 = nnz9[4 * g.C + c];
    z.v5 = nnz9[5 * g.C + c]; z.v6 = nnz9[6 * g.C + c]; z.v7 = nnz9[7 * g.C + c]; z.v8 = nnz9[8 * g.C + c];
    long long s1 = 0, s2 = 0;
    const int64_t nv = g.n4 / VEC;
    const uint32_t hwv = (uint32_t)(g.HW4 / VEC);
    for (int64_t i = (int64_t)sp * 256 + threadIdx.x; i < nv; i += (int64_t)S * 256) {
        const uint32_t n = fd_div((uint32_t)i, g.fd_hwv);
        const uint32_t q0 = ((uint32_t)i - n * hwv) * VEC;
        const unsigned char* src = h + ((int64_t)n * g.C + c) * g.HW + (int64_t)q0 * 4;
        uint32_t hb[VEC];
        if (VEC == 4) { const u32x4 v = *reinterpret_cast<const u32x4*>(src); hb[0] = v[0]; hb[VEC > 1 ? 1 : 0] = v[1]; hb[VEC > 2 ? 2 : 0] = v[2]; hb[VEC > 3 ? 3 : 0] = v[3]; }
        else hb[0] = *reinterpret_cast<const uint32_t*>(src);
        int t1 = 0, t2 = 0;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const uint32_t q = q0 + k, row = fd_div(q, g.fd_w4);
            float nz[4];
            stash_nnz_quad(z, (int)row, (int)(q - row * (uint32_t)g.W4), g.H, g.W4, nz);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int a = 2 * (int)((hb[k] >> (8 * e)) & 0xffu) - (int)nz[e];
                t1 += a; t2 += a * a;
            }
        }
        s1 += t1; s2 += t2;
    }
    const double d1 = block_reduce((double)s1, OpAddD(), 0.0, scd);
    const double d2 = block_reduce((double)s2, OpAddD(), 0.0, scd);
    if (threadIdx.x == 0) {
        const int gi = c / g.Mr, m = c - gi * g.Mr;
        double* dst = part + ((int64_t)sp * g.G * g.Mpad + gi * g.Mpad + m) * 2;
        dst[0] = d1; dst[1] = d2;
    }
}
template <int VEC>
__device__ __forceinline__ void h_sign_stream(const HGeom& g, const unsigned char* __restrict__ h, char* __restrict__ a, int c, int sp, int S, float T, float fl, const StashNnz& z) {
    const int64_t nv = g.n4 / VEC;
    const uint32_t hwv = (uint32_t)(g.HW4 / VEC);
    for (int64_t i = (int64_t)sp * 256 + threadIdx.x; i < nv; i += (int64_t)S * 256) {
        const uint32_t n = fd_div((uint32_t)i, g.fd_hwv);
        const uint32_t q0 = ((uint32_t)i - n * hwv) * VEC;
        const int64_t off = ((int64_t)n * g.C + c) * g.HW + (int64_t)q0 * 4;
        uint32_t hb[VEC], out[VEC];
        if (VEC == 4) { const u32x4 v = *reinterpret_cast<const u32x4*>(h + off); hb[0] = v[0]; hb[VEC > 1 ? 1 : 0] = v[1]; hb[VEC > 2 ? 2 : 0] = v[2]; hb[VEC > 3 ? 3 : 0] = v[3]; }
        else hb[0] = *reinterpret_cast<const uint32_t*>(h + off);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const uint32_t q = q0 + k, row = fd_div(q, g.fd_w4);
            float nz[4];
            stash_nnz_quad(z, (int)row, (int)(q - row * (uint32_t)g.W4), g.H, g.W4, nz);
            uint32_t o = 0u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float u = (2.f * (float)((hb[k] >> (8 * e)) & 0xffu) - nz[e]) * fl;
                o |= (u >= T ? 0x01u : 0xffu) << (8 * e);
            }
            out[k] = o;
        }
        if (VEC == 4) *reinterpret_cast<u32x4*>(a + off) = u32x4{out[0], out[VEC > 1 ? 1 : 0], out[VEC > 2 ? 2 : 0], out[VEC > 3 ? 3 : 0]};
        else *reinterpret_cast<uint32_t*>(a + off) = out[0];
    }
}
// The same pass for a POINTWISE block (one nnz per channel): u = (2 h - nnz) flip >= T is a comparison of the stash byte itself with one threshold per channel --
// flip > 0: h >= ceil((T + nnz) / 2), flip < 0: h <= floor((nnz - T) / 2) (T, nnz integers: exact) -- done on the four bytes of a word at once: the even and the odd
// bytes as two 16-bit lanes each, (0x100 | h) - t keeps bit 8 exactly when h >= t (t in [0, 256]: no borrow crosses a lane), the byte 0x01 / 0xff is 0xff - 0xfe m.
// ~4.5 instructions per element instead of ~12 (per-element nnz blend, conversion, multiply, compare, select): the float version was VALU-bound (3.4 TB/s).
template <int VEC>
__device__ __forceinline__ void h_sign_stream_pw(const HGeom& g, const unsigned char* __restrict__ h, char* __restrict__ a, int c, int sp, int S, float T, float fl, float nnz) {
    float tf;
    uint32_t inv;
    if (fl > 0.f) { tf = ceilf((T + nnz) * 0.5f); inv = 0u; }
    else { tf = floorf((nnz - T) * 0.5f) + 1.f; inv = 0x00010001u; }          // h <= t'  <=>  not (h >= t' + 1)
    tf = fminf(fmaxf(tf, 0.f), 256.f);
    const uint32_t tt = (uint32_t)tf * 0x00010001u;
    const int64_t nv = g.n4 / VEC;
    const uint32_t hwv = (uint32_t)(g.HW4 / VEC);
    for (int64_t i = (int64_t)sp * 256 + threadIdx.x; i < nv; i += (int64_t)S * 256) {
        const uint32_t n = fd_div((uint32_t)i, g.fd_hwv);
        const uint32_t q0 = ((uint32_t)i - n * hwv) * VEC;
        const int64_t off = ((int64_t)n * g.C + c) * g.HW + (int64_t)q0 * 4;
        uint32_t hb[VEC], out[VEC];
        if (VEC == 4) { const u32x4 v = *reinterpret_cast<const u32x4*>(h + off); hb[0] = v[0]; hb[VEC > 1 ? 1 : 0] = v[1]; hb[VEC > 2 ? 2 : 0] = v[2]; hb[VEC > 3 ? 3 : 0] = v[3]; }
        else hb[0] = *reinterpret_cast<const uint32_t*>(h + off);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const uint32_t ev = hb[k] & 0x00ff00ffu, od = (hb[k] >> 8) & 0x00ff00ffu;
            const uint32_t me = ((((ev | 0x01000100u) - tt) >> 8) & 0x00010001u) ^ inv, mo = ((((od | 0x01000100u) - tt) >> 8) & 0x00010001u) ^ inv;
            out[k] = (0x00ff00ffu - me * 0xfeu) | ((0x00ff00ffu - mo * 0xfeu) << 8);
        }
        if (VEC == 4) *reinterpret_cast<u32x4*>(a + off) = u32x4{out[0], out[VEC > 1 ? 1 : 0], out[VEC > 2 ? 2 : 0], out[VEC > 3 ? 3 : 0]};
        else *reinterpret_cast<uint32_t*>(a + off) = out[0];
    }
}
// ... and, for a block whose output goes through a 2x2 / stride-2 max-pool (models/nin_gc.py:88,119), the POOLED codes in the same pass: a thread takes 16 pixels of a
// row pair, the window's maximum is +1 iff any of its four pass bits is set (OR of the two rows' masks, then of the even / odd byte lanes); mn_maxpool2x2_sign8_fwd's
// pass over the full-size codes (1.25 B per element, one launch) is not needed.  H even, W a multiple of 16.
__device__ __forceinline__ void h_sign_stream_pw_pool(const HGeom& g, const unsigned char* __restrict__ h, char* __restrict__ a, char* __restrict__ ap, int c, int sp, int S,
                                                      float T, float fl, float nnz) {
    float tf;
    uint32_t inv;
    if (fl > 0.f) { tf = ceilf((T + nnz) * 0.5f); inv = 0u; }
    else { tf = floorf((nnz - T) * 0.5f) + 1.f; inv = 0x00010001u; }
    tf = fminf(fmaxf(tf, 0.f), 256.f);
    const uint32_t tt = (uint32_t)tf * 0x00010001u;
    const int W = 4 * g.W4, W16 = W >> 4, Hh = g.H >> 1;
    const uint32_t upp = (uint32_t)(Hh * W16);          // units per plane
    const int64_t nu = (int64_t)(g.n4 / g.HW4) * upp;   // N planes of this channel
    for (int64_t i = (int64_t)sp * 256 + threadIdx.x; i < nu; i += (int64_t)S * 256) {
        const uint32_t n = (uint32_t)(i / upp), u = (uint32_t)(i - (int64_t)n * upp);
        const uint32_t rp = u / (uint32_t)W16, cx = u - rp * (uint32_t)W16;
        const int64_t plane = (int64_t)n * g.C + c;
        const int64_t off0 = plane * g.HW + (int64_t)(2 * rp) * W + 16 * cx;
        const u32x4 ha = *reinterpret_cast<const u32x4*>(h + off0), hb = *reinterpret_cast<const u32x4*>(h + off0 + W);
        uint32_t oa[4], ob[4], pp[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t ea = ha[k] & 0x00ff00ffu, da = (ha[k] >> 8) & 0x00ff00ffu, eb = hb[k] & 0x00ff00ffu, db = (hb[k] >> 8) & 0x00ff00ffu;
            const uint32_t mea = ((((ea | 0x01000100u) - tt) >> 8) & 0x00010001u) ^ inv, moa = ((((da | 0x01000100u) - tt) >> 8) & 0x00010001u) ^ inv;
            const uint32_t meb = ((((eb | 0x01000100u) - tt) >> 8) & 0x00010001u) ^ inv, mob = ((((db | 0x01000100u) - tt) >> 8) & 0x00010001u) ^ inv;
            oa[k] = (0x00ff00ffu - mea * 0xfeu) | ((0x00ff00ffu - moa * 0xfeu) << 8);
            ob[k] = (0x00ff00ffu - meb * 0xfeu) | ((0x00ff00ffu - mob * 0xfeu) << 8);
            const uint32_t P = mea | moa | meb | mob;          // 16-bit lane j: the window of pixels 4 k + 2 j, 4 k + 2 j + 1
            pp[k] = (0xffu - (P & 1u) * 0xfeu) | ((0xffu - ((P >> 16) & 1u) * 0xfeu) << 8);
        }
        *reinterpret_cast<u32x4*>(a + off0) = u32x4{oa[0], oa[1], oa[2], oa[3]};
        *reinterpret_cast<u32x4*>(a + off0 + W) = u32x4{ob[0], ob[1], ob[2], ob[3]};
        *reinterpret_cast<u32x2*>(ap + (plane * Hh + rp) * (W >> 1) + 8 * cx) = u32x2{pp[0] | (pp[1] << 16), pp[2] | (pp[3] << 16)};
    }
}
template <int VEC>
__global__ __launch_bounds__(256) void k_h_sign(const HGeom g, const unsigned char* __restrict__ h, const float* __restrict__ chan, char* __restrict__ a) {
    const int c = blockIdx.x, sp = blockIdx.y, S = gridDim.y, C = g.C;
    const float T = chan[c], fl = chan[C + c], n7 = chan[7 * C + c];
    if (n7 >= 0.f) { h_sign_stream_pw<VEC>(g, h, a, c, sp, S, T, fl, n7); return; }          // (block-uniform)
    const StashNnz z = stash_nnz_load(chan, C, c);
    h_sign_stream<VEC>(g, h, a, c, sp, S, T, fl, z);
}
// default (MN_HSIGN_FOLD != 0): the constants from the block's own evaluation of k_pws_stats_prep's work (HsPrep above) instead of a launch in front
template <int VEC, int POOLED = 0>
__global__ __launch_bounds__(256) void k_h_sign_prep(const HGeom g, const unsigned char* __restrict__ h, const HsPrep q, char* __restrict__ a, char* __restrict__ ap) {
    __shared__ float hs_[3];
    const int c = blockIdx.x, sp = blockIdx.y, S = gridDim.y, C = g.C;
    if (threadIdx.x < 64) {
        float T_, fl_, nz_;
        pws_stats_prep_dev(q, c, threadIdx.x, sp == 0, T_, fl_, nz_);
        if (threadIdx.x == 0) { hs_[0] = T_; hs_[1] = fl_; hs_[2] = nz_; }
    }
    __syncthreads();
    const float T = hs_[0], fl = hs_[1];
    if (POOLED) { h_sign_stream_pw_pool(g, h, a, ap, c, sp, S, T, fl, hs_[2]); return; }          // (pointwise blocks only: host check)
    if (!q.nnz9) { h_sign_stream_pw<VEC>(g, h, a, c, sp, S, T, fl, hs_[2]); return; }          // a pointwise block: one nnz per channel
    StashNnz z;
    z.v0 = q.nnz9[c]; z.v1 = q.nnz9[C + c]; z.v2 = q.nnz9[2 * C + c]; z.v3 = q.nnz9[3 * C + c]; z.v4 = q.nnz9[4 * C + c];
    z.v5 = q.nnz9[5 * C + c]; z.v6 = q.nnz9[6 * C + c]; z.v7 = q.nnz9[7 * C + c]; z.v8 = q.nnz9[8 * C + c];
    h_sign_stream<VEC>(g, h, a, c, sp, S, T, fl, z);
}
static int h_splits(int C) { int S = 2048 / (C > 0 ? C : 1); return S < 1 ? 1 : (S > 64 ? 64 : S); }
static int h_sign_launch(int64_t N, int64_t O, int64_t H, int64_t W, const uint8_t* h, const float* chan, int8_t* a, hipStream_t s, const HsPrep* fold, int8_t* a_pool) {
    HGeom hg;
    hg.C = (int)O; hg.H = (int)H; hg.W4 = (int)(W / 4); hg.HW = (int)(H * W); hg.HW4 = hg.HW / 4; hg.Mr = 1; hg.Mpad = 1; hg.G = 1;
    hg.fd_hw4 = make_fastdiv((uint32_t)hg.HW4); hg.fd_w4 = make_fastdiv((uint32_t)hg.W4); hg.n4 = N * hg.HW4;
    const bool v4 = hg.HW % 16 == 0 && !(((uintptr_t)h) & 15) && !(((uintptr_t)a) & 15);
    hg.fd_hwv = make_fastdiv((uint32_t)(v4 ? hg.HW4 / 4 : hg.HW4));
    const int S = h_splits((int)O);
    mn_set_last_kernel(fold ? "k_h_sign_prep" : "k_h_sign");
    mn_prof_bytes((a_pool ? 2.25 : 2.0) * (double)N * O * hg.HW);
    mn_prof_begin(s);
    if (a_pool) {
        if (!fold || fold->nnz9 || !v4 || (H & 1) || (W & 15) || (((uintptr_t)a_pool) & 7)) MN_FAIL(MN_ENOTSUP, "mn_qconv_bnsign_fwd_stash_pool: needs a pointwise block in training mode, H even, W a multiple of 16");
        hipLaunchKernelGGL((k_h_sign_prep<4, 1>), dim3((unsigned)O, (unsigned)S), dim3(256), 0, s, hg, (const unsigned char*)h, *fold, (char*)a, (char*)a_pool);
    } else if (fold) {
        if (v4) hipLaunchKernelGGL((k_h_sign_prep<4, 0>), dim3((unsigned)O, (unsigned)S), dim3(256), 0, s, hg, (const unsigned char*)h, *fold, (char*)a, (char*)nullptr);
        else hipLaunchKernelGGL((k_h_sign_prep<1, 0>), dim3((unsigned)O, (unsigned)S), dim3(256), 0, s, hg, (const unsigned char*)h, *fold, (char*)a, (char*)nullptr);
    }
    else if (v4) hipLaunchKernelGGL(k_h_sign<4>, dim3((unsigned)O, (unsigned)S), dim3(256), 0, s, hg, (const unsigned char*)h, chan, (char*)a);
    else hipLaunchKernelGGL(k_h_sign<1>, dim3((unsigned)O, (unsigned)S), dim3(256), 0, s, hg, (const unsigned char*)h, chan, (char*)a);
    mn_prof_end(s);
    MN_CHECK_LAUNCH("mn_qconv_bnsign_fwd_stash(sign from h)");
    return MN_OK;
}
static int kxk_out(int in, int k, int s_, int pd, int d) { return (in + 2 * pd - d * (k - 1) - 1) / s_ + 1; }
static int64_t kxk_stash_ws(const mn_conv_geom* g, int64_t* off_nnz, int64_t* off_part) {
    const int64_t a = (kk_h8_ws_bytes(g) + 255) / 256 * 256;
    const int64_t b = ((int64_t)g->O * 10 * 4 + 255) / 256 * 256;          // nnz9 [9][O] + alpha [O]
    if (off_nnz) *off_nnz = a;
    if (off_part) *off_part = a + b;
    int64_t rows = (int64_t)h_splits(g->O) * g->groups * kk_h8_mpad(g);
    const int64_t rows3 = (int64_t)512 * (g->O / g->groups + 1);                // k_k3s_fwd: at most 512 blocks x Mg partial rows
    if (rows3 > rows) rows = rows3;
    return a + b + rows * 2 * 8;
}
static int qconv_kxk_bnsign_fwd_stash(const mn_conv_geom* g, const mn_wq* wq, const int8_t* x, const float* w, const float* bias, const float* gamma,
                                      const float* beta, float eps, float momentum, int training, float* running_mean, float* running_var, int64_t* nbt,
                                      float* save, int8_t* a, uint8_t* h, float* chan, void* ws, int64_t ws_bytes, hipStream_t s) {
    if (!gamma || !beta || !save || !a || (((uintptr_t)a) & 3) || !x || !w) MN_FAIL(MN_EINVAL, "mn_qconv_bnsign_fwd_stash(k x k): null / misaligned argument");
    if (!training && (!running_mean || !running_var)) MN_FAIL(MN_EINVAL, "mn_qconv_bnsign_fwd_stash(k x k): eval mode needs the running statistics");
    int64_t off_nnz, off_part;
    const int64_t need = kxk_stash_ws(g, &off_nnz, &off_part);
    if (!ws || ws_bytes < need || !aligned16(ws)) MN_FAIL(MN_ENOSPC, "mn_qconv_bnsign_fwd_stash(k x k): workspace too small");
    float* nnzf = (float*)((char*)ws + off_nnz);
    double* part = (double*)((char*)ws + off_part);
    const int Cg = g->C / g->groups, Mg = g->O / g->groups, S = h_splits(g->O);
    const int Ho = kxk_out(g->H, g->KH, g->stride_h, g->pad_h, g->dil_h), Wo = kxk_out(g->W, g->KW, g->stride_w, g->pad_w, g->dil_w);
    float* alphaf = nnzf + 9 * (int64_t)g->O;
    hipLaunchKernelGGL(k_row_nnz9, dim3((unsigned)g->O), dim3(64), 0, s, w, Cg, (int)g->O, nnzf, alphaf);
    const int Hs = kxk_out(g->H, g->KH, g->stride_h, g->pad_h, g->dil_h), Ws = kxk_out(g->W, g->KW, g->stride_w, g->pad_w, g->dil_w);
    if (k3s_fwd_supported(g, wq)) {
        // staged-image kernel: conv -> h and the statistics partials in one launch, no weight pack; then constants + sign
        int rc3 = 0;
        rc3 = k3s_fwd_h8(g, wq, x, w, nnzf, h, part, s);
        if (rc3) return rc3;
        const bool fold3 = hsign_fold_enabled();
        const HsPrep q3{(const double*)part, k3s_fwd_parts(g, wq), (int)g->groups, Mg, Mg, (const float*)alphaf, bias, (double)g->N * Hs * Ws, eps, momentum, training,
                        running_mean, running_var, save, (int)g->O, Cg * 9, 0, (const uint16_t*)nullptr, gamma, beta, chan, (const float*)nnzf, (long long*)nbt};
        if (!fold3)
        hipLaunchKernelGGL(k_pws_stats_prep, dim3((unsigned)g->O), dim3(64), 0, s, (const double*)part, k3s_fwd_parts(g, wq), (int)g->groups, Mg, Mg,
                           (const float*)alphaf, bias, (double)g->N * Hs * Ws, eps, momentum, training, running_mean, running_var, save, (int)g->O,
                           Cg * 9, 0, (const uint16_t*)nullptr, gamma, beta, chan, (const float*)nnzf, (long long*)nbt);
        HGeom hg3;
        hg3.C = g->O; hg3.H = Hs; hg3.W4 = Ws / 4; hg3.HW = Hs * Ws; hg3.HW4 = hg3.HW / 4; hg3.Mr = Mg; hg3.Mpad = Mg; hg3.G = g->groups;
        hg3.fd_hw4 = make_fastdiv((uint32_t)hg3.HW4); hg3.fd_w4 = make_fastdiv((uint32_t)hg3.W4); hg3.n4 = (int64_t)g->N * hg3.HW4;
        const bool v4 = hg3.HW % 16 == 0 && !(((uintptr_t)h) & 15) && !(((uintptr_t)a) & 15);
        hg3.fd_hwv = make_fastdiv((uint32_t)(v4 ? hg3.HW4 / 4 : hg3.HW4));
        mn_set_last_kernel(fold3 ? "k_h_sign_prep" : "k_h_sign");
        mn_prof_bytes(2.0 * (double)g->N * g->O * hg3.HW);
        mn_prof_begin(s);
        if (fold3) {
            if (v4) hipLaunchKernelGGL((k_h_sign_prep<4, 0>), dim3((unsigned)g->O, (unsigned)S), dim3(256), 0, s, hg3, (const unsigned char*)h, q3, (char*)a, (char*)nullptr);
            else hipLaunchKernelGGL((k_h_sign_prep<1, 0>), dim3((unsigned)g->O, (unsigned)S), dim3(256), 0, s, hg3, (const unsigned char*)h, q3, (char*)a, (char*)nullptr);
        }
        else if (v4) hipLaunchKernelGGL(k_h_sign<4>, dim3((unsigned)g->O, (unsigned)S), dim3(256), 0, s, hg3, (const unsigned char*)h, (const float*)chan, (char*)a);
        else hipLaunchKernelGGL(k_h_sign<1>, dim3((unsigned)g->O, (unsigned)S), dim3(256), 0, s, hg3, (const unsigned char*)h, (const float*)chan, (char*)a);
        mn_prof_end(s);
        MN_CHECK_LAUNCH("mn_qconv_bnsign_fwd_stash(3x3)");
        return MN_OK;
    }
    KkH8Info info;
    int rc = kk_fwd_h8(g, wq, x, w, nnzf, h, ws, off_nnz, s, &info);
    if (rc) return rc;
    HGeom hg;
    hg.C = g->O; hg.H = Ho; hg.W4 = Wo / 4; hg.HW = Ho * Wo; hg.HW4 = hg.HW / 4; hg.Mr = Mg; hg.Mpad = info.Mpad; hg.G = g->groups;
    hg.fd_hw4 = make_fastdiv((uint32_t)hg.HW4); hg.fd_w4 = make_fastdiv((uint32_t)hg.W4); hg.n4 = (int64_t)g->N * hg.HW4;
    const double ny = (double)g->N * g->O * hg.HW;
    const bool vec4 = hg.HW % 16 == 0 && !(((uintptr_t)h) & 15) && !(((uintptr_t)a) & 15);      // 16 codes per load
    hg.fd_hwv = make_fastdiv((uint32_t)(vec4 ? hg.HW4 / 4 : hg.HW4));
    if (training) {
        mn_set_last_kernel("k_h_stats");
        mn_prof_bytes(ny);
        mn_prof_begin(s);
        if (vec4) hipLaunchKernelGGL(k_h_stats<4>, dim3((unsigned)g->O, (unsigned)S), dim3(256), 0, s, hg, (const unsigned char*)h, (const float*)nnzf, part);
        else hipLaunchKernelGGL(k_h_stats<1>, dim3((unsigned)g->O, (unsigned)S), dim3(256), 0, s, hg, (const unsigned char*)h, (const float*)nnzf, part);
        mn_prof_end(s);
    }
    hipLaunchKernelGGL(k_pws_stats_prep, dim3((unsigned)g->O), dim3(64), 0, s, (const double*)part, S, (int)g->groups, info.Mpad, Mg, info.rowscale, bias,
                       (double)g->N * hg.HW, eps, momentum, training, running_mean, running_var, save, (int)g->O, info.K, info.Kp, info.codes, gamma, beta,
                       chan, (const float*)nnzf, (long long*)nbt);
    mn_set_last_kernel("k_h_sign");
    mn_prof_bytes(2.0 * ny);
    mn_prof_begin(s);
    if (vec4) hipLaunchKernelGGL(k_h_sign<4>, dim3((unsigned)g->O, (unsigned)S), dim3(256), 0, s, hg, (const unsigned char*)h, (const float*)chan, (char*)a);
    else hipLaunchKernelGGL(k_h_sign<1>, dim3((unsigned)g->O, (unsigned)S), dim3(256), 0, s, hg, (const unsigned char*)h, (const float*)chan, (char*)a);
    mn_prof_end(s);
    MN_CHECK_LAUNCH("mn_qconv_bnsign_fwd_stash(k x k)");
    return MN_OK;
}
extern "C" int mn_qconv_bnsign_stash_supported(const mn_conv_geom* g, const mn_wq* wq) {
    if (!g || !wq) return 0;
    return pws_bn_ok(g, wq) || kk_h8_supported(g, wq);
}
/* rows of the caller-owned per-channel table `chan` the stash forward fills: 8, or 17 for a 3x3 block (per-pixel-class nnz) */
extern "C" int mn_qconv_bnsign_stash_chan_rows(const mn_conv_geom* g) { return (g && g->KH == 1 && g->KW == 1) ? 8 : 17; }
extern "C" int64_t mn_qconv_bnsign_stash_ws_bytes(const mn_conv_geom* g) {
    if (!g) return -1;
    const int64_t a = pws_ws_bytes(g);
    return a > 0 ? a : kxk_stash_ws(g, nullptr, nullptr);
}

static int bnsign_bwd_impl(const mn_conv_geom* g, const mn_wq* wq, const int8_t* x, const float* w, const float* bias, const float* gamma,
                           const float* beta, const float* save, const float* da, const int8_t* own, int training, float* dy, float* dgamma, float* dbeta,
                           void* ws, int64_t ws_bytes, mn_stream_t stream) {
    if (!g || !gamma || !beta || !save || !da || !dy || !aligned16(da) || !aligned16(dy)) MN_FAIL(MN_EINVAL, "mn_qconv_bnsign_bwd: null / misaligned argument");
    if (own && ((g->H & 1) || (g->W & 3) || (((uintptr_t)own) & 3))) MN_FAIL(MN_ENOTSUP, "mn_qconv_bnsign_bwd_pooled: needs even H, W %% 4 == 0");
    if (!pws_bn_ok(g, wq)) MN_FAIL(MN_ENOTSUP, "mn_qconv_bnsign_bwd: needs a pointwise convolution with ternary / binary weights");
    hipStream_t s = (hipStream_t)stream;
    PwsPlan pl;
    int rc = pws_prepare(g, wq, x, w, ws, ws_bytes, PWS_NT_BWD, &pl, s, "mn_qconv_bnsign_bwd");
    if (rc) return rc;
    PwsParams& p = pl.p;
    float* sums = (float*)((char*)ws + pl.off_sums);
    p.bias = bias; p.da = da; p.training = training; p.own = (const char*)own;
    pws_chan_prep(pl, g, bias, save, gamma, beta, s);
    const double nx = (double)g->N * g->C * g->H * g->W, ny = (double)g->N * g->O * g->H * g->W;
    if (own) rc = launch_pws<PWS_BWD_PART_POOL>(pl, s, nx + 2.0 * ny, "mn_qconv_bnsign_bwd(partial, pooled)");
    else rc = launch_pws<PWS_BWD_PART>(pl, s, nx + 4.0 * ny, "mn_qconv_bnsign_bwd(partial)");
    if (rc) return rc;
    hipLaunchKernelGGL(k_pws_final_bwd, dim3((unsigned)g->O), dim3(64), 0, s, (const double*)p.part, p.CB, p.G, p.Mpad, p.Mr, dgamma, dbeta, sums, (int)g->O);
    p.sums = sums; p.y = dy;
    if (own) return launch_pws<PWS_BWD_APPLY_POOL>(pl, s, nx + 6.0 * ny, "mn_qconv_bnsign_bwd(apply, pooled)");
    return launch_pws<PWS_BWD_APPLY>(pl, s, nx + 8.0 * ny, "mn_qconv_bnsign_bwd(apply)");
}
extern "C" int mn_qconv_bnsign_bwd(const mn_conv_geom* g, const mn_wq* wq, const int8_t* x, const float* w, const float* bias, const float* gamma,
                                   const float* beta, const float* save, const float* da, int training, float* dy, float* dgamma, float* dbeta,
                                   void* ws, int64_t ws_bytes, mn_stream_t stream) {
    return bnsign_bwd_impl(g, wq, x, w, bias, gamma, beta, save, da, nullptr, training, dy, dgamma, dbeta, ws, ws_bytes, stream);
}
extern "C" int mn_qconv_bnsign_bwd_pooled(const mn_conv_geom* g, const mn_wq* wq, const int8_t* x, const float* w, const float* bias, const float* gamma,
                                          const float* beta, const float* save, const float* dpool, const int8_t* a_own, int training, float* dy,
                                          float* dgamma, float* dbeta, void* ws, int64_t ws_bytes, mn_stream_t stream) {
    if (!a_own) MN_FAIL(MN_EINVAL, "mn_qconv_bnsign_bwd_pooled: null output codes");
    return bnsign_bwd_impl(g, wq, x, w, bias, gamma, beta, save, dpool, a_own, training, dy, dgamma, dbeta, ws, ws_bytes, stream);
}


// ------------------------------------------------------------------------------------------------
// conv + BatchNorm2d + ReLU + next-layer k-bit quantizer (DoReFa blocks): the forward on activation codes that leaves the 16-bit stash of acc,
// the exact batch statistics and the per-channel constants of qact_kernels.hip.  Pointwise: k_pws<.., PWS_STATS, XENC 1>; 3 x 3: k_k3s_fwd<1>.
static int bnq_amax(int a_bits) { return (1 << a_bits) - 1; }
static int64_t bnq_accmax(const mn_conv_geom* g, const mn_wq* wq, int a_bits) {
    const int64_t K = (int64_t)(g->C / g->groups) * g->KH * g->KW, wmax = (1ll << wq->bits) - 1;
    return K * bnq_amax(a_bits) * wmax;
}
static int bnq_int16_ok(const mn_conv_geom* g, const mn_wq* wq, int a_bits) { return bnq_accmax(g, wq, a_bits) <= 32767; }
// the WIDE variant of the grouped kernels (XENC 2: bf16 j for j <= 255, 32-bit stash): 8-bit activation codes, or an accumulator beyond int16
static int bnq_wide(const mn_conv_geom* g, const mn_wq* wq, int a_bits) { return a_bits == 8 || !bnq_int16_ok(g, wq, a_bits); }
extern "C" int mn_qconv_bnq_supported(const mn_conv_geom* g, const mn_wq* wq, int a_bits_in) {
    if (!g || !wq || wq->mode != MN_WQ_DOREFA || wq->bits < 2 || wq->bits > 8 || a_bits_in < 2 || a_bits_in > 8 || g->groups < 1 || g->C % g->groups || g->O % g->groups) return 0;
    if (qd_fwd_supported(g, wq, a_bits_in) && qd_dgrad_supported(g, wq) && qd_wgrad_supported(g, a_bits_in)) return 1;      // dense layers (ResNets): qgemm_dense.hip
    const int wide = bnq_wide(g, wq, a_bits_in);
    if (wide && bnq_accmax(g, wq, a_bits_in) >= (1ll << 24)) return 0;          // fp32 accumulation of bf16 integer products must stay exact
    if (g->stride_h != 1 || g->stride_w != 1) return 0;
    if (((int64_t)g->H * g->W) % 8) return 0;
    if (g->KH == 1 && g->KW == 1) {
        PwsPlan pl;
        return plan_pws(g, PWS_NT_FWD, &pl) && pws_wgrad_code8_supported(g) && pwd_supported(g, wq);
    }
    return k3s_fwd16_supported(g, wq) && k3s_wgrad_code8_supported(g, a_bits_in) && k3s_dgrad_supported(g, wq);
}
extern "C" int mn_qconv_bnq_stash_bits(const mn_conv_geom* g, const mn_wq* wq, int a_bits_in) {
    if (!mn_qconv_bnq_supported(g, wq, a_bits_in)) return 0;
    if (qd_fwd_supported(g, wq, a_bits_in) && qd_dgrad_supported(g, wq) && qd_wgrad_supported(g, a_bits_in)) return qd_stash32(g, wq, a_bits_in) ? 32 : 16;
    return bnq_wide(g, wq, a_bits_in) ? 32 : 16;
}
extern "C" int64_t mn_qconv_bnq_ws_bytes(const mn_conv_geom* g) {
    if (!g) return -1;
    if (qd_fwd_ws_bytes(g) > 0) return qd_fwd_ws_bytes(g);
    if (g->KH == 1 && g->KW == 1) return pws_ws_bytes(g);
    return ((int64_t)512 * g->O * 2 * 8 + 255) / 256 * 256 + (int64_t)g->O * 4 + 256;      // statistics partials [<= 512][O][2] doubles + the per-channel scale
}
__global__ void k_fill_f32(float* __restrict__ p, int n, float v) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = v; }
extern "C" int mn_qconv_bnq_fwd_stash(const mn_conv_geom* g, const mn_wq* wq, const uint8_t* x_codes, int a_bits_in, const float* w, const float* bias, const float* gamma,
                                      const float* beta, float eps, float momentum, int training, float* running_mean, float* running_var,
                                      int64_t* num_batches_tracked, float* save, int16_t* stash, float* chan, void* ws, int64_t ws_bytes, mn_stream_t stream) {
    if (!mn_qconv_bnq_supported(g, wq, a_bits_in)) MN_FAIL(MN_ENOTSUP, "mn_qconv_bnq_fwd_stash: geometry / quantizer combination not covered");
    if (!x_codes || !w || !gamma || !beta || !save || !stash || !chan || (((uintptr_t)x_codes) & 3) || (((uintptr_t)stash) & 15)) MN_FAIL(MN_EINVAL, "mn_qconv_bnq_fwd_stash: null / misaligned argument");
    if (!training && (!running_mean || !running_var)) MN_FAIL(MN_EINVAL, "mn_qconv_bnq_fwd_stash: eval mode needs the running statistics");
    hipStream_t s = (hipStream_t)stream;
    const float ascale = dorefa_scale(a_bits_in);
    if (qd_fwd_supported(g, wq, a_bits_in) && qd_dgrad_supported(g, wq) && qd_wgrad_supported(g, a_bits_in)) {      // dense layer: stash 16 / 32 bits wide (mn_qconv_bnq_stash_bits)
        const double* part; int nparts; float* rowscale;
        int rc = qd_fwd_stash(g, wq, x_codes, a_bits_in, w, (void*)stash, ws, ws_bytes, s, &part, &nparts, &rowscale);
        if (rc) return rc;
        const int Ho = (int)((g->H + 2 * g->pad_h - g->KH) / g->stride_h + 1), Wo = (int)((g->W + 2 * g->pad_w - g->KW) / g->stride_w + 1);
        (void)rowscale;
        qa_launch_stats_prep_const(part, nparts, (int)g->O, 1.0f / (float)((1ll << wq->bits) - 1), ascale, bias, (double)g->N * Ho * Wo, eps, momentum, training, running_mean,
                                   running_var, save, gamma, beta, chan, (long long*)num_batches_tracked, s);
        MN_CHECK_LAUNCH("mn_qconv_bnq_fwd_stash(dense)");
        return MN_OK;
    }
    const double npix = (double)g->N * g->H * g->W;
    const int wide = bnq_wide(g, wq, a_bits_in);          // XENC 2, 32-bit stash (mn_qconv_bnq_stash_bits)
    if (g->KH == 1 && g->KW == 1) {
        PwsPlan pl;
        int rc = pws_prepare(g, wq, (const int8_t*)x_codes, w, ws, ws_bytes, PWS_NT_FWD, &pl, s, "mn_qconv_bnq_fwd_stash");
        if (rc) return rc;
        PwsParams& p = pl.p;
        if (wide) p.h32 = reinterpret_cast<int32_t*>(stash); else p.h16 = stash;
        const double nx = (double)g->N * g->C * g->H * g->W, ny = (double)g->N * g->O * g->H * g->W;
        // the stash is needed in eval mode too (the streaming forward reads it): the statistics pass always runs, its sums are ignored in eval
        if (wide) rc = launch_pws<PWS_STATS, 2>(pl, s, nx + 4.0 * ny, "mn_qconv_bnq_fwd_stash(stats, wide)");
        else rc = launch_pws<PWS_STATS, 1>(pl, s, nx + 2.0 * ny, "mn_qconv_bnq_fwd_stash(stats)");
        if (rc) return rc;
        qa_launch_stats_prep((const double*)p.part, p.CB, p.G, p.Mpad, p.Mr, p.rowscale, ascale, bias, npix, eps, momentum, training, running_mean, running_var, save,
                             (int)g->O, gamma, beta, chan, (long long*)num_batches_tracked, s);
        MN_CHECK_LAUNCH("mn_qconv_bnq_fwd_stash");
        return MN_OK;
    }
    if (!ws || ws_bytes < mn_qconv_bnq_ws_bytes(g) || !aligned16(ws)) MN_FAIL(MN_ENOSPC, "mn_qconv_bnq_fwd_stash(3x3): workspace too small");
    double* part = (double*)ws;
    float* rowscale = (float*)((char*)ws + ((int64_t)512 * g->O * 2 * 8 + 255) / 256 * 256);
    const int Mg = g->O / g->groups;
    const float wsc = 1.0f / (float)((1ll << wq->bits) - 1);                 // the per-channel weight scale the pack kernel would produce (1 / n)
    hipLaunchKernelGGL(k_fill_f32, dim3((unsigned)((g->O + 255) / 256)), dim3(256), 0, s, rowscale, (int)g->O, wsc);
    int rc = k3s_fwd_h16(g, wq, x_codes, w, stash, wide, part, s);
    if (rc) return rc;
    qa_launch_stats_prep(part, k3s_fwd16_parts(g, wq), g->groups, Mg, Mg, rowscale, ascale, bias, npix, eps, momentum, training, running_mean, running_var, save,
                         (int)g->O, gamma, beta, chan, (long long*)num_batches_tracked, s);
    MN_CHECK_LAUNCH("mn_qconv_bnq_fwd_stash(3x3)");
    return MN_OK;
}
