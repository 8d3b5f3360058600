// BatchNorm2d + ReLU + the NEXT layer's k-bit activation quantizer, fused, for gfx950 -- the block structure of the reference's DoReFa
// nets:  conv -> bn -> relu -> [max-pool 2x2] -> QuantConv2d's ActivationQuantizer  (models/nin_gc.py:53-59,88,119 under
// wqaq/dorefa/quantize.py:36-46,107-122).
//
// A k-bit block's conv output is y = alpha[o] * acc + bias[o] with acc an EXACT integer (input codes j in [0, 2^a - 1] times weight codes
// 2k - n; SURVEY Appendix A "exact-integer GEMM").  So, exactly as for the binary blocks (qgemm_sign.hip), fp32 y is never stored: the conv
// kernels leave acc as a 16-bit STASH (|acc| <= K * amax * wmax <= 32767, checked by the planner) plus exact integer batch-statistics
// partials, and everything behind the conv streams that stash:
//   forward   k_qa_fwd      stash (2 B/elt) -> z = bn(y) -> a = relu(z) -> [2x2 max] -> code j = rha(clamp(0.1 a, 0, 1) / s)   -> 1 B/elt
//             (the next conv reads ONE byte per element instead of 4, and the quantizer is not re-evaluated by its three passes)
//   backward  k_qa_partial  (dq [pooled], stash) -> dz = STE(dq) * [z > 0] (through the pool's first-maximum routing)  -> sum dz, sum dz zhat
//             k_qa_apply    (dq, stash)          -> dy = gamma invstd (dz - sum_dz / n - zhat sum_dzzhat / n)           -> 4 B/elt
// dq = d loss / d (quantised activation) is what the next conv's backward-data produces BEFORE its clip-STE epilogue; the STE
// (wqaq/dorefa/quantize.py:36-46 backward: ((g s) / s) * [0 <= 0.1 a <= 1] * 0.1) is evaluated here from the recomputed a.
// Every value is recomputed with the fp32 expression chain of the unfused kernels (k_pw epilogue: y = acc * alpha + bias; k_bns_apply:
// zh = (y - mean) * invstd, z = zh * gamma + beta, relu; dorefa_act_q), so codes and masks are bit-identical to the unfused path whenever
// the batch statistics are (they differ by fp32 round-off of the mean / variance: exact integer sums here).
// IN = 1: the input is fp32 y (the block behind the un-quantised FIRST conv, whose output is not an integer): same kernels, 4 B/elt in.
#include "qgemm_dev.h"

#define QA_NCH 9          // chan rows: alpha, bias, mean, invstd, gamma, beta, A = alpha*invstd, B = (bias - mean)*invstd, gi = gamma*invstd
struct QaGeom {
    int N, C, H, W, HW, HW8, W8;
    FastDiv fd_hw8, fd_w8;
    int64_t n8;             // 8-element groups per channel (no pool) / 4-window groups per channel (pool)
    float s;                // quantizer scale 1 / (2^a - 1)
    float inv_s;            // RN(1 / s) for the division-free clip-STE (qa_dz_m); 0: the IEEE division
    int interval;           // backward passes: the ReLU / clamp masks as one interval of the streamed value per channel (qa_mask_interval; knob MN_QA_NO_INTERVAL)
    int nthr;               // > 0 (mn_qa_fwd on the integer stash, <= 3 bit codes): levels 2^a - 1 of the integer-threshold forward
    // the "final" step of the backward sums folded into the apply pass (mn_qa_bwd / mn_qr_bwd: one launch less per BatchNorm backward): every block of channel c sums
    // the S partial rows itself, in k_qa_final_bwd / k_qr_final_bwd's order (mn_row_sums: bit-identical statistics in all of them), block sp == 0 writes the outputs
    const double* fin_part; // [C][S][2] (qa) / [C][S][3] (qr); null: `sums` holds the finished values
    int fin_S;
    float* fin_dgamma; float* fin_dbeta; float* fin_sums;               // nullable, nullable, [2][C]
    float* fin_dgamma_s; float* fin_dbeta_s; float* fin_sums_s;         // qr with a shortcut BatchNorm (all nullable)
    uint8_t* mask4;         // mn_qa_fwd_f32_mask (fp32 input, no pool, codes out): also the backward's pass nibbles, one byte per 4 elements (as mn_conv2d_first_bnact_fwd act 2)
};
struct QaCh { float alpha, bias, mean, invstd, ga, be, A, B, gi; };
__device__ __forceinline__ QaCh qa_load_ch(const float* __restrict__ chan, int C, int c) {
    QaCh k;
    k.alpha = chan[c]; k.bias = chan[C + c]; k.mean = chan[2 * C + c]; k.invstd = chan[3 * C + c]; k.ga = chan[4 * C + c]; k.be = chan[5 * C + c];
    k.A = chan[6 * C + c]; k.B = chan[7 * C + c]; k.gi = chan[8 * C + c];
    return k;
}
template <int IN>
__device__ __forceinline__ void qa_eval(float v, const QaCh& k, float& zh, float& z) {
    const float y = IN == 1 ? v : v * k.alpha + k.bias;
    zh = (y - k.mean) * k.invstd;
    z = zh * k.ga + k.be;
}
// 8 consecutive elements (one row segment) of channel c at group index i: element offset
__device__ __forceinline__ int64_t qa_off8(const QaGeom& g, int c, uint32_t i) {
    const uint32_t n = fd_div(i, g.fd_hw8);
    return ((int64_t)n * g.C + c) * g.HW + (int64_t)(i - n * (uint32_t)g.HW8) * 8;
}
template <int IN>
__device__ __forceinline__ void qa_load8(const void* __restrict__ in, int64_t off, float (&v)[8]) {
    if (IN == 1) {
        const float4 a = *reinterpret_cast<const float4*>((const float*)in + off), b = *reinterpret_cast<const float4*>((const float*)in + off + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else if (IN == 2) {          // 32-bit stash (dense layers; the wide grouped blocks: 8-bit codes) -- |acc| < 2^24: exact in fp32
        const u32x4 a = *reinterpret_cast<const u32x4*>((const int32_t*)in + off), b = *reinterpret_cast<const u32x4*>((const int32_t*)in + off + 4);
#pragma unroll
        for (int d = 0; d < 4; ++d) { v[d] = (float)(int)a[d]; v[4 + d] = (float)(int)b[d]; }
    } else {
        const u32x4 u = *reinterpret_cast<const u32x4*>((const int16_t*)in + off);
#pragma unroll
        for (int d = 0; d < 4; ++d) { v[2 * d] = (float)(int16_t)(u[d] & 0xffffu); v[2 * d + 1] = (float)(int16_t)(u[d] >> 16); }
    }
}
// pooled group i of channel c: 4 windows of one pooled row = 8 columns x 2 rows; e0 = element offset of the upper row's first column,
// po = element offset of the first pooled output
__device__ __forceinline__ void qa_pool_off(const QaGeom& g, int c, int64_t i, int64_t& e0, int64_t& po) {
    const int Hh = g.H >> 1;
    const int64_t t = i / g.W8;
    const int q = (int)(i - t * g.W8);
    const int64_t n = t / Hh;
    const int pr = (int)(t - n * Hh);
    const int64_t plane = n * g.C + c;
    e0 = plane * g.HW + (int64_t)(2 * pr) * g.W + 8 * q;
    po = plane * (g.HW >> 2) + (int64_t)pr * (g.W >> 1) + 4 * q;
}
// first maximum of a window in row-major order, ATen's rule (a later element wins if it is greater OR NaN): index 0..3
__device__ __forceinline__ int qa_argmax4(const float (&a)[4]) {
    float m = -INFINITY;
    int k = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (a[e] > m || a[e] != a[e]) { m = a[e]; k = e; }
    return k;
}

// ---------------------------------------------------------------- forward: stash / y  ->  codes (OUT 0) or the fp32 activation (OUT 1)
// Integer-threshold forward (stash input, codes of <= 3 bits): the code is a monotone step function of the integer accumulator -- every step of the
// chain acc -> y -> zhat -> z -> relu -> clamp(0.1 a) -> rha(./s) is monotone in fp32 as well -- so per channel there are n = 2^a - 1 integers T_k with
// code = #{k : u >= T_k}, u = flip * acc (flip = -1 when the chain decreases).  T_k = the smallest u whose EXACT chain value reaches k, found by a
// binary search over the int16 range with that chain: the codes are bit-identical to the element-wise evaluation, at ~8 instead of ~20 VALU per
// element (the pass was VALU-bound at 3.5 TB/s of 3 B/elt).  A channel with a non-finite or absurdly large constant keeps the element-wise path.
#define QA_MAXTHR 7
template <int IN>
__device__ __forceinline__ uint32_t qa_code_of(float v, const QaCh& k, float s) { float zh, z; qa_eval<IN>(v, k, zh, z); return qa_code(qa_relu(z), s); }
// |constant| <= 1e9 for all six: no intermediate of the chain can overflow on an int16 input (|y| <= 3.3e13, |zhat| <= 3.3e22, |z| <= 3.3e31), so no
// inf * 0 = NaN can break the monotonicity the thresholds rely on; anything wilder (or NaN) takes the element-wise path
__device__ __forceinline__ bool qa_finite(float v) { return fabsf(v) <= 1.0e9f; }
template <int IN, int POOL, int OUT>
__global__ __launch_bounds__(256) void k_qa_fwd(const QaGeom g, const void* __restrict__ in, const float* __restrict__ chan, unsigned char* __restrict__ codes,
                                                float* __restrict__ af) {
    const int c = blockIdx.x, sp = blockIdx.y, S = gridDim.y;
    const QaCh k = qa_load_ch(chan, g.C, c);
    if (!IN && !OUT && g.nthr > 0 && qa_finite(k.alpha) && qa_finite(k.bias) && qa_finite(k.mean) && qa_finite(k.invstd) && qa_finite(k.ga) && qa_finite(k.be)) {
        __shared__ float thr[QA_MAXTHR + 1];
        const float flip = (qa_code_of<0>(32767.f, k, g.s) < qa_code_of<0>(-32768.f, k, g.s)) ? -1.f : 1.f;
        if ((int)threadIdx.x < g.nthr) {
            // smallest u in [-32768, 32768] with code(flip * u) >= level, 32769 if none (u = 32768 only occurs as -(-32768))
            const uint32_t level = threadIdx.x + 1u;
            int lo = -32768, hi = 32769;                       // invariant: code(lo - 1) < level (virtually), code(hi) >= level (virtually at 32769)
            while (lo < hi) {
                const int mid = lo + ((hi - lo) >> 1);
                if (qa_code_of<0>(flip * (float)mid, k, g.s) >= level) hi = mid; else lo = mid + 1;
            }
            thr[threadIdx.x] = (float)lo;
        }
        __syncthreads();
        float T[QA_MAXTHR];
#pragma unroll
        for (int q = 0; q < QA_MAXTHR; ++q) T[q] = q < g.nthr ? thr[q] : 40000.f;
        auto code_u = [&](float u) {
            uint32_t j = 0u;
#pragma unroll
            for (int q = 0; q < QA_MAXTHR; ++q) if (q < 3 || g.nthr > 3) j += (u >= T[q]) ? 1u : 0u;
            return j;
        };
        for (int64_t i = (int64_t)sp * 256 + threadIdx.x; i < g.n8; i += (int64_t)S * 256) {
            if (!POOL) {
                const int64_t off = qa_off8(g, c, (uint32_t)i);
                float v[8];
                qa_load8<0>(in, off, v);
                uint32_t lo = 0u, hi = 0u;
#pragma unroll
                for (int e = 0; e < 4; ++e) { lo |= code_u(v[e] * flip) << (8 * e); hi |= code_u(v[4 + e] * flip) << (8 * e); }
                *reinterpret_cast<u32x2*>(codes + off) = u32x2{lo, hi};
            } else {
                int64_t e0, po;
                qa_pool_off(g, c, i, e0, po);
                float r0[8], r1[8];
                qa_load8<0>(in, e0, r0);
                qa_load8<0>(in, e0 + g.W, r1);
                uint32_t w4 = 0u;
#pragma unroll
                for (int w = 0; w < 4; ++w) {        // the window's largest activation is the one with the largest u
                    const float m = fmaxf(fmaxf(r0[2 * w] * flip, r0[2 * w + 1] * flip), fmaxf(r1[2 * w] * flip, r1[2 * w + 1] * flip));
                    w4 |= code_u(m) << (8 * w);
                }
                *reinterpret_cast<uint32_t*>(codes + po) = w4;
            }
        }
        return;
    }
    for (int64_t i = (int64_t)sp * 256 + threadIdx.x; i < g.n8; i += (int64_t)S * 256) {
        if (!POOL) {
            const int64_t off = qa_off8(g, c, (uint32_t)i);
            float v[8], a[8];
            qa_load8<IN>(in, off, v);
            uint32_t mk = 0u;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float zh, z;
                qa_eval<IN>(v[e], k, zh, z);
                a[e] = qa_relu(z);
                if (IN == 1 && !OUT) {          // low nibble: z > 0; high nibble: ... and the quantizer's clamp test (dorefa_act_grad_m)
                    const float tq = a[e] * 0.1f;
                    const bool pos = z > 0.f;
                    mk |= (pos ? (1u << (e & 3)) : 0u) << (8 * (e >> 2));
                    mk |= ((pos && tq >= 0.f && tq <= 1.f) ? (16u << (e & 3)) : 0u) << (8 * (e >> 2));
                }
            }
            if (IN == 1 && !OUT && g.mask4) *reinterpret_cast<uint16_t*>(g.mask4 + (off >> 2)) = (uint16_t)mk;
            if (OUT) {
                *reinterpret_cast<float4*>(af + off) = make_float4(a[0], a[1], a[2], a[3]);
                *reinterpret_cast<float4*>(af + off + 4) = make_float4(a[4], a[5], a[6], a[7]);
            } else {
                uint32_t lo = 0u, hi = 0u;
#pragma unroll
                for (int e = 0; e < 4; ++e) { lo |= qa_code(a[e], g.s) << (8 * e); hi |= qa_code(a[4 + e], g.s) << (8 * e); }
                *reinterpret_cast<u32x2*>(codes + off) = u32x2{lo, hi};
            }
        } else {
            int64_t e0, po;
            qa_pool_off(g, c, i, e0, po);
            float r0[8], r1[8];
            qa_load8<IN>(in, e0, r0);
            qa_load8<IN>(in, e0 + g.W, r1);
            float m[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                float a[4];
                const float vv[4] = {r0[2 * w], r0[2 * w + 1], r1[2 * w], r1[2 * w + 1]};
#pragma unroll
                for (int e = 0; e < 4; ++e) { float zh, z; qa_eval<IN>(vv[e], k, zh, z); a[e] = qa_relu(z); }
                m[w] = a[qa_argmax4(a)];
            }
            if (OUT) *reinterpret_cast<float4*>(af + po) = make_float4(m[0], m[1], m[2], m[3]);
            else *reinterpret_cast<uint32_t*>(codes + po) = qa_code(m[0], g.s) | (qa_code(m[1], g.s) << 8) | (qa_code(m[2], g.s) << 16) | (qa_code(m[3], g.s) << 24);
        }
    }
}

// ---------------------------------------------------------------- backward
// dz of the 8 elements (no pool) / of the 16 elements of 4 windows (pool: only the window's first maximum receives gradient).
// QUANT 1: dq is the gradient w.r.t. the QUANTISED activation (the clip-STE of the quantizer is applied here); 0: w.r.t. the activation itself
// (the block's consumer is not a quantised conv: e.g. the last block of the net).
// (qa_relu, qa_dz: common.h -- shared with the first-layer backward-weight kernel that folds this block, conv_first.hip)
// the block's channel: its backward masks as an interval of the streamed value (stash integer / fp32 y), found by thread 0 and broadcast; use = false: a channel
// with non-finite (or absurd) constants, or gamma == 0, keeps the element-wise masks
struct QaIv { float lo, hi; bool use; };
template <int IN>
__device__ __forceinline__ QaIv qa_block_interval(const QaGeom& g, const QaCh& k, int quant) {
    __shared__ float iv_[3];
    if (threadIdx.x == 0) {
        float use = 0.f;
        QaInterval r; r.lo = 1.f; r.hi = 0.f;
        if (g.interval && qa_finite(k.alpha) && qa_finite(k.bias) && qa_finite(k.mean) && qa_finite(k.invstd) && qa_finite(k.ga) && qa_finite(k.be) && k.ga != 0.f && k.invstd > 0.f &&
            (IN == 1 || k.alpha != 0.f)) {
            auto zf = [&](float v) { float zh, z; qa_eval<IN>(v, k, zh, z); return z; };
            if (IN == 1) r = qa_mask_interval(0x7f7fffff, zf, [](int32_t w) { return mn_keyf(w); }, quant);
            else r = qa_mask_interval(IN == 2 ? (1 << 24) : 32768, zf, [](int32_t w) { return (float)w; }, quant);
            use = 1.f;
        }
        iv_[0] = r.lo; iv_[1] = r.hi; iv_[2] = use;
    }
    __syncthreads();
    QaIv o; o.lo = iv_[0]; o.hi = iv_[1]; o.use = iv_[2] != 0.f;
    return o;
}
template <int IN, int POOL>
__global__ __launch_bounds__(256) void k_qa_partial(const QaGeom g, const void* __restrict__ in, const float* __restrict__ chan, const float* __restrict__ dq,
                                                    int quant, double* __restrict__ part) {
    __shared__ double scd[16];
    const int c = blockIdx.x, sp = blockIdx.y, S = gridDim.y;
    const QaCh k = qa_load_ch(chan, g.C, c);
    QaIv iv; iv.lo = 1.f; iv.hi = 0.f; iv.use = false;
    if (!POOL) iv = qa_block_interval<IN>(g, k, quant);          // (the pooled pass needs the activations themselves: the window's first maximum)
    double s1 = 0.0, s2 = 0.0;
    for (int64_t i = (int64_t)sp * 256 + threadIdx.x; i < g.n8; i += (int64_t)S * 256) {
        float t1 = 0.f, t2 = 0.f;
        if (!POOL) {
            const int64_t off = qa_off8(g, c, (uint32_t)i);
            float v[8];
            qa_load8<IN>(in, off, v);
            const float4 ga = *reinterpret_cast<const float4*>(dq + off), gb = *reinterpret_cast<const float4*>(dq + off + 4);
            const float gv[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
            if (iv.use) {          // block-uniform
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float y = IN == 1 ? v[e] : v[e] * k.alpha + k.bias;
                    const float zh = (y - k.mean) * k.invstd;
                    const float d = quant ? dorefa_ste_core_m(gv[e], g.s, g.inv_s) : gv[e];
                    const float dz = (v[e] >= iv.lo && v[e] <= iv.hi) ? d : 0.f;
                    t1 += dz; t2 += dz * zh;
                }
            } else
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float zh, z;
                qa_eval<IN>(v[e], k, zh, z);
                const float dz = qa_dz_m(gv[e], qa_relu(z), z, g.s, g.inv_s, quant);
                t1 += dz; t2 += dz * zh;
            }
        } else {
            int64_t e0, po;
            qa_pool_off(g, c, i, e0, po);
            float r0[8], r1[8];
            qa_load8<IN>(in, e0, r0);
            qa_load8<IN>(in, e0 + g.W, r1);
            const float4 g4 = *reinterpret_cast<const float4*>(dq + po);
            const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                float a[4], zz[4], zhh[4];
                const float vv[4] = {r0[2 * w], r0[2 * w + 1], r1[2 * w], r1[2 * w + 1]};
#pragma unroll
                for (int e = 0; e < 4; ++e) { qa_eval<IN>(vv[e], k, zhh[e], zz[e]); a[e] = qa_relu(zz[e]); }
                const int kx = qa_argmax4(a);
                float am = a[0], zm = zz[0], zhm = zhh[0];
#pragma unroll
                for (int e = 1; e < 4; ++e) if (kx == e) { am = a[e]; zm = zz[e]; zhm = zhh[e]; }
                const float dz = qa_dz_m(gv[w], am, zm, g.s, g.inv_s, quant);
                t1 += dz; t2 += dz * zhm;
            }
        }
        s1 += (double)t1; s2 += (double)t2;
    }
    s1 = block_reduce(s1, OpAddD(), 0.0, scd);
    s2 = block_reduce(s2, OpAddD(), 0.0, scd);
    if (threadIdx.x == 0) { part[((int64_t)c * S + sp) * 2] = s1; part[((int64_t)c * S + sp) * 2 + 1] = s2; }
}
__global__ void k_qa_final_bwd(int C, const double* __restrict__ part, int S, float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ sums) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double sv[2] = {0.0, 0.0};
    mn_row_sums<2>(part + (int64_t)c * S * 2, S, sv);
    const double s1 = sv[0], s2 = sv[1];
    if (dbeta) dbeta[c] = (float)s1;
    if (dgamma) dgamma[c] = (float)s2;
    sums[c] = (float)s1; sums[C + c] = (float)s2;
}
template <int IN, int POOL>
__global__ __launch_bounds__(256) void k_qa_apply(const QaGeom g, const void* __restrict__ in, const float* __restrict__ chan, const float* __restrict__ dq,
                                                  const float* __restrict__ sums, int training, int quant, float* __restrict__ dy) {
    const int c = blockIdx.x, sp = blockIdx.y, S = gridDim.y;
    const QaCh k = qa_load_ch(chan, g.C, c);
    float k1 = 0.f, k2 = 0.f;
    float s1f, s2f;
    if (g.fin_part) {          // (block-uniform) the final sums of the partial pass in front, here
        __shared__ float fs[2];
        if (threadIdx.x == 0) {
            double sv[2] = {0.0, 0.0};
            mn_row_sums<2>(g.fin_part + (int64_t)c * g.fin_S * 2, g.fin_S, sv);
            fs[0] = (float)sv[0]; fs[1] = (float)sv[1];
            if (sp == 0) {
                if (g.fin_dbeta) g.fin_dbeta[c] = fs[0];
                if (g.fin_dgamma) g.fin_dgamma[c] = fs[1];
                if (g.fin_sums) { g.fin_sums[c] = fs[0]; g.fin_sums[g.C + c] = fs[1]; }
            }
        }
        __syncthreads();
        s1f = fs[0]; s2f = fs[1];
    } else { s1f = sums[c]; s2f = sums[g.C + c]; }
    if (training) { const float n = (float)g.N * (float)g.HW; k1 = s1f / n; k2 = s2f / n; }
    QaIv iv; iv.lo = 1.f; iv.hi = 0.f; iv.use = false;
    if (!POOL) iv = qa_block_interval<IN>(g, k, quant);
    for (int64_t i = (int64_t)sp * 256 + threadIdx.x; i < g.n8; i += (int64_t)S * 256) {
        if (!POOL) {
            const int64_t off = qa_off8(g, c, (uint32_t)i);
            float v[8], r[8];
            qa_load8<IN>(in, off, v);
            const float4 ga = *reinterpret_cast<const float4*>(dq + off), gb = *reinterpret_cast<const float4*>(dq + off + 4);
            const float gv[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
            if (iv.use) {          // block-uniform
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float y = IN == 1 ? v[e] : v[e] * k.alpha + k.bias;
                    const float zh = (y - k.mean) * k.invstd;
                    const float d = quant ? dorefa_ste_core_m(gv[e], g.s, g.inv_s) : gv[e];
                    const float dz = (v[e] >= iv.lo && v[e] <= iv.hi) ? d : 0.f;
                    r[e] = k.gi * (dz - k1 - zh * k2);
                }
            } else
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float zh, z;
                qa_eval<IN>(v[e], k, zh, z);
                const float dz = qa_dz_m(gv[e], qa_relu(z), z, g.s, g.inv_s, quant);
                r[e] = k.gi * (dz - k1 - zh * k2);
            }
            *reinterpret_cast<float4*>(dy + off) = make_float4(r[0], r[1], r[2], r[3]);
            *reinterpret_cast<float4*>(dy + off + 4) = make_float4(r[4], r[5], r[6], r[7]);
        } else {
            int64_t e0, po;
            qa_pool_off(g, c, i, e0, po);
            float r0[8], r1[8], o0[8], o1[8];
            qa_load8<IN>(in, e0, r0);
            qa_load8<IN>(in, e0 + g.W, r1);
            const float4 g4 = *reinterpret_cast<const float4*>(dq + po);
            const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                float a[4], zz[4], zhh[4];
                const float vv[4] = {r0[2 * w], r0[2 * w + 1], r1[2 * w], r1[2 * w + 1]};
#pragma unroll
                for (int e = 0; e < 4; ++e) { qa_eval<IN>(vv[e], k, zhh[e], zz[e]); a[e] = qa_relu(zz[e]); }
                const int kx = qa_argmax4(a);
                float rr[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float dz = (kx == e) ? qa_dz_m(gv[w], a[e], zz[e], g.s, g.inv_s, quant) : 0.f;
                    rr[e] = k.gi * (dz - k1 - zhh[e] * k2);
                }
                o0[2 * w] = rr[0]; o0[2 * w + 1] = rr[1]; o1[2 * w] = rr[2]; o1[2 * w + 1] = rr[3];
            }
            *reinterpret_cast<float4*>(dy + e0) = make_float4(o0[0], o0[1], o0[2], o0[3]);
            *reinterpret_cast<float4*>(dy + e0 + 4) = make_float4(o0[4], o0[5], o0[6], o0[7]);
            *reinterpret_cast<float4*>(dy + e0 + g.W) = make_float4(o1[0], o1[1], o1[2], o1[3]);
            *reinterpret_cast<float4*>(dy + e0 + g.W + 4) = make_float4(o1[4], o1[5], o1[6], o1[7]);
        }
    }
}

// ---------------------------------------------------------------- per-channel constants
// from (mean, invstd) of a BatchNorm over fp32 y (the first block: save = what mn_bnrelu_fwd / k_bns_final_fwd wrote); alpha = 1, bias = 0
__global__ void k_qa_chan_f32(int C, const float* __restrict__ save, const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ chan) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float mean = save[c], invstd = save[C + c];
    chan[c] = 1.f; chan[C + c] = 0.f; chan[2 * C + c] = mean; chan[3 * C + c] = invstd; chan[4 * C + c] = gamma[c]; chan[5 * C + c] = beta[c];
    chan[6 * C + c] = invstd; chan[7 * C + c] = (0.f - mean) * invstd; chan[8 * C + c] = gamma[c] * invstd;
}
// from the exact integer statistics partials of a stashing conv (layout of k_pws / k_k3s_fwd: part[(i * G * Mpad + g * Mpad + m) * 2]):
// mean / biased variance of y = alpha * acc + bias in fp64, running statistics (unbiased variance) and num_batches_tracked like nn.BatchNorm2d;
// eval: the running statistics.  One wave per channel.
__global__ __launch_bounds__(64) void k_qa_stats_prep(const double* __restrict__ part, int CB, int G, int Mpad, int Mr, const float* __restrict__ rowscale, float rowscale_const, float ascale,
                                                     const float* __restrict__ bias, double n, float eps, float momentum, int training,
                                                     float* __restrict__ running_mean, float* __restrict__ running_var, float* __restrict__ save, int Cout,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ chan,
                                                     long long* __restrict__ nbt) {
    const int co = blockIdx.x, g = co / Mr, m = co - g * Mr, lane = threadIdx.x;
    const float alpha = (rowscale ? rowscale[g * Mpad + m] : rowscale_const) * ascale;          // what the unfused k_pw / k_kk epilogue multiplies acc with
    const float b = bias ? bias[co] : 0.f;
    float mean_f = 0.f, inv_f = 0.f;
    if (training) {
        double a1 = 0.0, a2 = 0.0;
        for (int i = lane; i < CB; i += 64) {
            const double* src = part + ((int64_t)i * G * Mpad + g * Mpad + m) * 2;
            a1 += src[0]; a2 += src[1];
        }
        a1 = wave_reduce(a1, OpAddD()); a2 = wave_reduce(a2, OpAddD());
        if (lane == 0) {
            const double al = (double)alpha;
            const double ma = a1 / n;
            const double mean = al * ma + (double)b;
            double ss = al * al * (a2 - a1 * ma);
            if (ss < 0.0) ss = 0.0;
            mean_f = (float)mean;
            inv_f = 1.0f / sqrtf((float)(ss / n) + eps);
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
        const float ga = gamma[co], be = beta[co];
        chan[co] = alpha; chan[Cout + co] = b; chan[2 * Cout + co] = mean_f; chan[3 * Cout + co] = inv_f; chan[4 * Cout + co] = ga; chan[5 * Cout + co] = be;
        chan[6 * Cout + co] = alpha * inv_f; chan[7 * Cout + co] = (b - mean_f) * inv_f; chan[8 * Cout + co] = ga * inv_f;
    }
}
void qa_launch_stats_prep(const double* part, int CB, int G, int Mpad, int Mr, const float* rowscale, float ascale, const float* bias, double n, float eps,
                          float momentum, int training, float* running_mean, float* running_var, float* save, int Cout, const float* gamma, const float* beta,
                          float* chan, long long* nbt, hipStream_t s) {
    hipLaunchKernelGGL(k_qa_stats_prep, dim3((unsigned)Cout), dim3(64), 0, s, part, CB, G, Mpad, Mr, rowscale, 0.f, ascale, bias, n, eps, momentum, training, running_mean,
                       running_var, save, Cout, gamma, beta, chan, nbt);
}
// the same with ONE weight scale for every channel (DoReFa: 1 / (2^w - 1)): no per-channel table to fill first
void qa_launch_stats_prep_const(const double* part, int CB, int Cout, float wscale, float ascale, const float* bias, double n, float eps, float momentum, int training,
                                float* running_mean, float* running_var, float* save, const float* gamma, const float* beta, float* chan, long long* nbt, hipStream_t s) {
    hipLaunchKernelGGL(k_qa_stats_prep, dim3((unsigned)Cout), dim3(64), 0, s, part, CB, 1, Cout, Cout, (const float*)nullptr, wscale, ascale, bias, n, eps, momentum, training,
                       running_mean, running_var, save, Cout, gamma, beta, chan, nbt);
}

// ---------------------------------------------------------------- host side
static int qa_geom(QaGeom* g, int64_t N, int64_t C, int64_t H, int64_t W, int bits, int pool, const char* what) {
    const int64_t HW = H * W;
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || HW % 8 || bits < 2 || bits > 8) MN_FAIL(MN_EINVAL, "%s: bad shape / bits (H*W must be a multiple of 8, 2 <= bits <= 8)", what);
    if (pool && ((H & 1) || W % 8)) MN_FAIL(MN_EINVAL, "%s: pooled variant needs even H and W %% 8 == 0", what);
    if (N * (HW / 8) >= ((int64_t)1 << 31)) MN_FAIL(MN_EINVAL, "%s: tensor too large", what);
    g->N = (int)N; g->C = (int)C; g->H = (int)H; g->W = (int)W; g->HW = (int)HW; g->HW8 = (int)(HW / 8); g->W8 = (int)(W / 8);
    g->fd_hw8 = make_fastdiv((uint32_t)g->HW8); g->fd_w8 = make_fastdiv((uint32_t)(g->W8 > 0 ? g->W8 : 1));
    g->n8 = pool ? N * (H / 2) * (W / 8) : N * (HW / 8);
    g->s = dorefa_scale(bits);
    g->inv_s = mn_qa_inv(g->s);
    g->interval = mn_qa_interval();
    g->nthr = 0;
    g->mask4 = nullptr;
    g->fin_part = nullptr; g->fin_S = 0; g->fin_dgamma = g->fin_dbeta = g->fin_sums = g->fin_dgamma_s = g->fin_dbeta_s = g->fin_sums_s = nullptr;
    return MN_OK;
}
static int qa_split(const QaGeom& g) {
    int64_t S = (2048 + g.C - 1) / g.C;
    const int64_t maxS = (g.n8 + 255) / 256;
    if (S > maxS) S = maxS;
    if (S > 32) S = 32;
    if (S < 1) S = 1;
    return (int)S;
}
extern "C" int mn_qa_supported(int64_t H, int64_t W, int pool) { return (H * W) % 8 == 0 && (!pool || ((H & 1) == 0 && W % 8 == 0)); }
extern "C" int64_t mn_qa_ws_floats(int64_t C) { return C * 32 * 4 + 2 * C + 16; }
extern "C" int mn_qa_chan_from_save(const float* save, const float* gamma, const float* beta, int64_t C, float* chan, mn_stream_t stream) {
    if (!save || !gamma || !beta || !chan || C <= 0) MN_FAIL(MN_EINVAL, "mn_qa_chan_from_save: bad arguments");
    hipLaunchKernelGGL(k_qa_chan_f32, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (int)C, save, gamma, beta, chan);
    MN_CHECK_LAUNCH("mn_qa_chan_from_save");
    return MN_OK;
}
#define QA_DISPATCH(KERNEL, ...)                                                                                         \
    if (in_f32 == 2) { if (pool) hipLaunchKernelGGL((KERNEL<2, 1>), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<2, 0>), __VA_ARGS__); } \
    else if (in_f32) { if (pool) hipLaunchKernelGGL((KERNEL<1, 1>), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<1, 0>), __VA_ARGS__); } \
    else { if (pool) hipLaunchKernelGGL((KERNEL<0, 1>), __VA_ARGS__); else hipLaunchKernelGGL((KERNEL<0, 0>), __VA_ARGS__); }
extern "C" int mn_qa_fwd(int in_f32, const void* in, const float* chan, int64_t N, int64_t C, int64_t H, int64_t W, int a_bits, int pool, uint8_t* codes, float* act_f32,
                         mn_stream_t stream) {
    QaGeom g;
    int rc = qa_geom(&g, N, C, H, W, a_bits, pool, "mn_qa_fwd");
    if (rc) return rc;
    if (!in || !chan || (!codes && !act_f32) || (((uintptr_t)in) & 15) || (codes && (((uintptr_t)codes) & 7)) || (act_f32 && !aligned16(act_f32)))
        MN_FAIL(MN_EINVAL, "mn_qa_fwd: null / misaligned tensor");
    hipStream_t s = (hipStream_t)stream;
    if (!in_f32 && codes && a_bits <= 3) g.nthr = (1 << a_bits) - 1;       // integer-threshold forward
    const dim3 grid((unsigned)C, (unsigned)qa_split(g));
    const double nel = (double)N * C * H * W;
    mn_set_last_kernel("k_qa_fwd<%d, %d>", in_f32, pool ? 1 : 0);
    mn_prof_bytes(nel * (in_f32 ? 4.0 : 2.0) + (codes ? 1.0 : 0.0) * nel / (pool ? 4.0 : 1.0) + (act_f32 ? 4.0 : 0.0) * nel / (pool ? 4.0 : 1.0));
    mn_prof_begin(s);
    if (in_f32 == 2) {
        if (codes) { if (pool) hipLaunchKernelGGL((k_qa_fwd<2, 1, 0>), grid, dim3(256), 0, s, g, in, chan, codes, (float*)nullptr); else hipLaunchKernelGGL((k_qa_fwd<2, 0, 0>), grid, dim3(256), 0, s, g, in, chan, codes, (float*)nullptr); }
        if (act_f32) { if (pool) hipLaunchKernelGGL((k_qa_fwd<2, 1, 1>), grid, dim3(256), 0, s, g, in, chan, (unsigned char*)nullptr, act_f32); else hipLaunchKernelGGL((k_qa_fwd<2, 0, 1>), grid, dim3(256), 0, s, g, in, chan, (unsigned char*)nullptr, act_f32); }
        mn_prof_end(s);
        MN_CHECK_LAUNCH("mn_qa_fwd");
        return MN_OK;
    }
    if (codes) {
        if (in_f32) { if (pool) hipLaunchKernelGGL((k_qa_fwd<1, 1, 0>), grid, dim3(256), 0, s, g, in, chan, codes, (float*)nullptr); else hipLaunchKernelGGL((k_qa_fwd<1, 0, 0>), grid, dim3(256), 0, s, g, in, chan, codes, (float*)nullptr); }
        else { if (pool) hipLaunchKernelGGL((k_qa_fwd<0, 1, 0>), grid, dim3(256), 0, s, g, in, chan, codes, (float*)nullptr); else hipLaunchKernelGGL((k_qa_fwd<0, 0, 0>), grid, dim3(256), 0, s, g, in, chan, codes, (float*)nullptr); }
    }
    if (act_f32) {
        if (in_f32) { if (pool) hipLaunchKernelGGL((k_qa_fwd<1, 1, 1>), grid, dim3(256), 0, s, g, in, chan, (unsigned char*)nullptr, act_f32); else hipLaunchKernelGGL((k_qa_fwd<1, 0, 1>), grid, dim3(256), 0, s, g, in, chan, (unsigned char*)nullptr, act_f32); }
        else { if (pool) hipLaunchKernelGGL((k_qa_fwd<0, 1, 1>), grid, dim3(256), 0, s, g, in, chan, (unsigned char*)nullptr, act_f32); else hipLaunchKernelGGL((k_qa_fwd<0, 0, 1>), grid, dim3(256), 0, s, g, in, chan, (unsigned char*)nullptr, act_f32); }
    }
    mn_prof_end(s);
    MN_CHECK_LAUNCH("mn_qa_fwd");
    return MN_OK;
}
/* mn_qa_fwd(in_f32 = 1, pool = 0, codes) that also writes the pass nibbles mn_conv2d_bwd_first_mask_gram reads (mask4 [N][C][H W / 4]) */
extern "C" int mn_qa_fwd_f32_mask(const float* y, const float* chan, int64_t N, int64_t C, int64_t H, int64_t W, int a_bits, uint8_t* codes, uint8_t* mask4,
                                  mn_stream_t stream) {
    QaGeom g;
    int rc = qa_geom(&g, N, C, H, W, a_bits, 0, "mn_qa_fwd_f32_mask");
    if (rc) return rc;
    if (!y || !chan || !codes || !mask4 || !aligned16(y) || (((uintptr_t)codes) & 7) || (((uintptr_t)mask4) & 1)) MN_FAIL(MN_EINVAL, "mn_qa_fwd_f32_mask: null / misaligned tensor");
    hipStream_t s = (hipStream_t)stream;
    g.mask4 = mask4;
    const dim3 grid((unsigned)C, (unsigned)qa_split(g));
    const double nel = (double)N * C * H * W;
    mn_set_last_kernel("k_qa_fwd<1, 0>");
    mn_prof_bytes(nel * 5.25);
    mn_prof_begin(s);
    hipLaunchKernelGGL((k_qa_fwd<1, 0, 0>), grid, dim3(256), 0, s, g, (const void*)y, chan, codes, (float*)nullptr);
    mn_prof_end(s);
    MN_CHECK_LAUNCH("mn_qa_fwd_f32_mask");
    return MN_OK;
}
extern "C" int mn_qa_bwd_sums(int in_f32, const void* in, const float* chan, const float* dq, int64_t N, int64_t C, int64_t H, int64_t W, int a_bits, int pool, int quant,
                              float* dgamma, float* dbeta, float* sums, float* ws, mn_stream_t stream) {
    QaGeom g;
    int rc = qa_geom(&g, N, C, H, W, a_bits, pool, "mn_qa_bwd_sums");
    if (rc) return rc;
    if (!in || !chan || !dq || !sums || !ws || (((uintptr_t)in) & 15) || !aligned16(dq) || (((uintptr_t)ws) & 7)) MN_FAIL(MN_EINVAL, "mn_qa_bwd_sums: null / misaligned tensor");
    hipStream_t s = (hipStream_t)stream;
    const int S = qa_split(g);
    const dim3 grid((unsigned)C, (unsigned)S);
    const double nel = (double)N * C * H * W;
    mn_set_last_kernel("k_qa_partial<%d, %d>", in_f32, pool ? 1 : 0);
    mn_prof_bytes(nel * (in_f32 ? 4.0 : 2.0) + 4.0 * nel / (pool ? 4.0 : 1.0));
    mn_prof_begin(s);
    QA_DISPATCH(k_qa_partial, grid, dim3(256), 0, s, g, in, chan, dq, quant, (double*)ws)
    mn_prof_end(s);
    hipLaunchKernelGGL(k_qa_final_bwd, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, s, (int)C, (const double*)ws, S, dgamma, dbeta, sums);
    MN_CHECK_LAUNCH("mn_qa_bwd_sums");
    return MN_OK;
}
/* the finish of mn_qa_bwd_sums alone, on partial sums [C][splits][2] (doubles) the producer of dq left (mn_conv2d_bwd_qa_up / mn_conv2d_bwd_codes_up) */
extern "C" int mn_qa_bwd_sums_final(const double* part, int32_t splits, int64_t C, float* dgamma, float* dbeta, float* sums, mn_stream_t stream) {
    if (!part || !sums || splits < 1 || C <= 0 || (((uintptr_t)part) & 7)) MN_FAIL(MN_EINVAL, "mn_qa_bwd_sums_final: bad arguments");
    hipLaunchKernelGGL(k_qa_final_bwd, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, (hipStream_t)stream, (int)C, part, (int)splits, dgamma, dbeta, sums);
    MN_CHECK_LAUNCH("mn_qa_bwd_sums_final");
    return MN_OK;
}
extern "C" int mn_qa_bwd_apply(int in_f32, const void* in, const float* chan, const float* sums, const float* dq, int64_t N, int64_t C, int64_t H, int64_t W, int a_bits,
                               int pool, int quant, int training, float* dy, mn_stream_t stream) {
    QaGeom g;
    int rc = qa_geom(&g, N, C, H, W, a_bits, pool, "mn_qa_bwd_apply");
    if (rc) return rc;
    if (!in || !chan || !dq || !sums || !dy || (((uintptr_t)in) & 15) || !aligned16(dq) || !aligned16(dy)) MN_FAIL(MN_EINVAL, "mn_qa_bwd_apply: null / misaligned tensor");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)C, (unsigned)qa_split(g));
    const double nel = (double)N * C * H * W;
    mn_set_last_kernel("k_qa_apply<%d, %d>", in_f32, pool ? 1 : 0);
    mn_prof_bytes(nel * (in_f32 ? 4.0 : 2.0) + 4.0 * nel / (pool ? 4.0 : 1.0) + 4.0 * nel);
    mn_prof_begin(s);
    QA_DISPATCH(k_qa_apply, grid, dim3(256), 0, s, g, in, chan, dq, sums, training, quant, dy)
    mn_prof_end(s);
    MN_CHECK_LAUNCH("mn_qa_bwd_apply");
    return MN_OK;
}

/* mn_qa_bwd_sums + mn_qa_bwd_apply in TWO launches: the partial pass, then the apply pass whose blocks finish the sums themselves (bit-identical statistics, dgamma /
 * dbeta / sums written by the apply pass); ws as for mn_qa_bwd_sums */
extern "C" int mn_qa_bwd(int in_f32, const void* in, const float* chan, const float* dq, int64_t N, int64_t C, int64_t H, int64_t W, int a_bits, int pool, int quant,
                         int training, float* dgamma, float* dbeta, float* sums, float* dy, float* ws, mn_stream_t stream) {
    QaGeom g;
    int rc = qa_geom(&g, N, C, H, W, a_bits, pool, "mn_qa_bwd");
    if (rc) return rc;
    if (!in || !chan || !dq || !sums || !dy || !ws || (((uintptr_t)in) & 15) || !aligned16(dq) || !aligned16(dy) || (((uintptr_t)ws) & 7)) MN_FAIL(MN_EINVAL, "mn_qa_bwd: null / misaligned tensor");
    hipStream_t s = (hipStream_t)stream;
    const int S = qa_split(g);
    const dim3 grid((unsigned)C, (unsigned)S);
    const double nel = (double)N * C * H * W;
    mn_set_last_kernel("k_qa_partial<%d, %d>", in_f32, pool ? 1 : 0);
    mn_prof_bytes(nel * (in_f32 ? 4.0 : 2.0) + 4.0 * nel / (pool ? 4.0 : 1.0));
    mn_prof_begin(s);
    QA_DISPATCH(k_qa_partial, grid, dim3(256), 0, s, g, in, chan, dq, quant, (double*)ws)
    mn_prof_end(s);
    g.fin_part = (const double*)ws; g.fin_S = S; g.fin_dgamma = dgamma; g.fin_dbeta = dbeta; g.fin_sums = sums;
    mn_set_last_kernel("k_qa_apply<%d, %d>", in_f32, pool ? 1 : 0);
    mn_prof_bytes(nel * (in_f32 ? 4.0 : 2.0) + 4.0 * nel / (pool ? 4.0 : 1.0) + 4.0 * nel);
    mn_prof_begin(s);
    QA_DISPATCH(k_qa_apply, grid, dim3(256), 0, s, g, in, chan, dq, (const float*)sums, training, quant, dy)
    mn_prof_end(s);
    MN_CHECK_LAUNCH("mn_qa_bwd");
    return MN_OK;
}

// ================================================================================================
// The END of a residual block of the reference's ResNets (models/resnet.py:60-65: relu(add(residual_function(x), shortcut(x)))) under the k-bit DoReFa
// scheme, fused like the blocks above:
//     u = bn(y) + res,   a = relu(u)   ->   codes of the NEXT convs' activation quantizer (1 B/elt) and / or fp32 a (the next block's identity shortcut)
// y is the second conv's stash (IN 0 / 2) or fp32 (IN 1); res is nothing (RES 0: the stem block, when it has to emit codes AND fp32 in one pass), the
// fp32 input of the block (RES 1: identity shortcut) or bn_s(y_s) evaluated from the stash of the 1 x 1 / stride 2 shortcut conv (RES 2: int16, 3: int32).
// Backward in two streaming passes with du = d loss / d u stored once (it IS the gradient of an identity shortcut):
//     k_qr_partial   du = (STE(dq) [+ STE(dq2)] [+ g]) * [u > 0];  sum du, sum du zhat (main BatchNorm) [, sum du zhat_s (shortcut BatchNorm)]
//     k_qr_apply     dy = gamma invstd (du - sum_du / n - zhat sum_du_zhat / n)  [and dy_s likewise]
// dq / dq2: gradients w.r.t. the QUANTISED activation from the (up to two) convs that read the codes -- the clip-STE of wqaq/dorefa/quantize.py:36-46 is
// applied here, per consumer as autograd would; g: gradient w.r.t. the activation itself (the next block's identity shortcut, or a foreign consumer).
template <int RES>
__device__ __forceinline__ void qr_load_res8(const void* __restrict__ res, int64_t off, const QaCh& ks, float (&r)[8], float (&zhs)[8]) {
    if (RES == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { r[e] = 0.f; zhs[e] = 0.f; }
    } else if (RES == 1) {
        qa_load8<1>(res, off, r);
#pragma unroll
        for (int e = 0; e < 8; ++e) zhs[e] = 0.f;
    } else {
        float v[8];
        qa_load8<(RES == 2 ? 0 : 2)>(res, off, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) qa_eval<0>(v[e], ks, zhs[e], r[e]);
    }
}
template <int IN, int RES>
__global__ __launch_bounds__(256) void k_qr_fwd(const QaGeom g, const void* __restrict__ in, const float* __restrict__ chan, const void* __restrict__ res,
                                                const float* __restrict__ res_chan, unsigned char* __restrict__ codes, float* __restrict__ af) {
    const int c = blockIdx.x, sp = blockIdx.y, S = gridDim.y;
    const QaCh k = qa_load_ch(chan, g.C, c);
    QaCh ks = k;
    if (RES >= 2) ks = qa_load_ch(res_chan, g.C, c);
    for (int64_t i = (int64_t)sp * 256 + threadIdx.x; i < g.n8; i += (int64_t)S * 256) {
        const int64_t off = qa_off8(g, c, (uint32_t)i);
        float v[8], r[8], zhs[8], a[8];
        qa_load8<IN>(in, off, v);
        qr_load_res8<RES>(res, off, ks, r, zhs);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float zh, z;
            qa_eval<IN>(v[e], k, zh, z);
            a[e] = qa_relu(RES ? z + r[e] : z);
        }
        if (af) {
            *reinterpret_cast<float4*>(af + off) = make_float4(a[0], a[1], a[2], a[3]);
            *reinterpret_cast<float4*>(af + off + 4) = make_float4(a[4], a[5], a[6], a[7]);
        }
        if (codes) {
            uint32_t lo = 0u, hi = 0u;
#pragma unroll
            for (int e = 0; e < 4; ++e) { lo |= qa_code(a[e], g.s) << (8 * e); hi |= qa_code(a[4 + e], g.s) << (8 * e); }
            *reinterpret_cast<u32x2*>(codes + off) = u32x2{lo, hi};
        }
    }
}
template <int IN, int RES>
__global__ __launch_bounds__(256) void k_qr_partial(const QaGeom g, const void* __restrict__ in, const float* __restrict__ chan, const void* __restrict__ res,
                                                    const float* __restrict__ res_chan, const float* __restrict__ dq, const float* __restrict__ dq2,
                                                    const float* __restrict__ gf, float* __restrict__ du, double* __restrict__ part) {
    __shared__ double scd[16];
    const int c = blockIdx.x, sp = blockIdx.y, S = gridDim.y;
    const QaCh k = qa_load_ch(chan, g.C, c);
    QaCh ks = k;
    if (RES >= 2) ks = qa_load_ch(res_chan, g.C, c);
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int64_t i = (int64_t)sp * 256 + threadIdx.x; i < g.n8; i += (int64_t)S * 256) {
        const int64_t off = qa_off8(g, c, (uint32_t)i);
        float v[8], r[8], zhs[8], d1[8], d2[8], d3[8], o[8];
        qa_load8<IN>(in, off, v);
        qr_load_res8<RES>(res, off, ks, r, zhs);
#pragma unroll
        for (int e = 0; e < 8; ++e) { d1[e] = 0.f; d2[e] = 0.f; d3[e] = 0.f; }
        if (dq) qa_load8<1>(dq, off, d1);
        if (dq2) qa_load8<1>(dq2, off, d2);
        if (gf) qa_load8<1>(gf, off, d3);
        float t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float zh, z;
            qa_eval<IN>(v[e], k, zh, z);
            const float u = RES ? z + r[e] : z;
            const float a = qa_relu(u);
            float da = 0.f;
            if (dq) da = dorefa_act_grad_m(d1[e], a, g.s, g.inv_s);
            if (dq2) da = da + dorefa_act_grad_m(d2[e], a, g.s, g.inv_s);
            if (gf) da = (dq || dq2) ? da + d3[e] : d3[e];
            const float dd = (u > 0.f) ? da : 0.f;
            o[e] = dd;
            t1 += dd; t2 += dd * zh; t3 += dd * zhs[e];
        }
        *reinterpret_cast<float4*>(du + off) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(du + off + 4) = make_float4(o[4], o[5], o[6], o[7]);
        s1 += (double)t1; s2 += (double)t2; s3 += (double)t3;
    }
    s1 = block_reduce(s1, OpAddD(), 0.0, scd);
    s2 = block_reduce(s2, OpAddD(), 0.0, scd);
    if (RES >= 2) s3 = block_reduce(s3, OpAddD(), 0.0, scd);
    if (threadIdx.x == 0) { double* d = part + ((int64_t)c * S + sp) * 3; d[0] = s1; d[1] = s2; d[2] = s3; }
}
__global__ void k_qr_final_bwd(int C, const double* __restrict__ part, int S, float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ sums,
                               float* __restrict__ dgamma_s, float* __restrict__ dbeta_s, float* __restrict__ sums_s) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double sv[3] = {0.0, 0.0, 0.0};
    mn_row_sums<3>(part + (int64_t)c * S * 3, S, sv);
    const double s1 = sv[0], s2 = sv[1], s3 = sv[2];
    if (dbeta) dbeta[c] = (float)s1;
    if (dgamma) dgamma[c] = (float)s2;
    sums[c] = (float)s1; sums[C + c] = (float)s2;
    if (sums_s) { sums_s[c] = (float)s1; sums_s[C + c] = (float)s3; }
    if (dbeta_s) dbeta_s[c] = (float)s1;
    if (dgamma_s) dgamma_s[c] = (float)s3;
}
template <int IN, int RES>
__global__ __launch_bounds__(256) void k_qr_apply(const QaGeom g, const void* __restrict__ in, const float* __restrict__ chan, const float* __restrict__ sums,
                                                  const void* __restrict__ res, const float* __restrict__ res_chan, const float* __restrict__ sums_s,
                                                  const float* __restrict__ du, int training, float* __restrict__ dy, float* __restrict__ dy_s) {
    const int c = blockIdx.x, sp = blockIdx.y, S = gridDim.y;
    const QaCh k = qa_load_ch(chan, g.C, c);
    QaCh ks = k;
    if (RES >= 2) ks = qa_load_ch(res_chan, g.C, c);
    float k1 = 0.f, k2 = 0.f, k1s = 0.f, k2s = 0.f;
    float s1f, s2f, s3f = 0.f;
    if (g.fin_part) {          // (block-uniform) k_qr_final_bwd, here
        __shared__ float fs[3];
        if (threadIdx.x == 0) {
            double sv[3] = {0.0, 0.0, 0.0};
            mn_row_sums<3>(g.fin_part + (int64_t)c * g.fin_S * 3, g.fin_S, sv);
            fs[0] = (float)sv[0]; fs[1] = (float)sv[1]; fs[2] = (float)sv[2];
            if (sp == 0) {
                if (g.fin_dbeta) g.fin_dbeta[c] = fs[0];
                if (g.fin_dgamma) g.fin_dgamma[c] = fs[1];
                if (g.fin_sums) { g.fin_sums[c] = fs[0]; g.fin_sums[g.C + c] = fs[1]; }
                if (g.fin_sums_s) { g.fin_sums_s[c] = fs[0]; g.fin_sums_s[g.C + c] = fs[2]; }
                if (g.fin_dbeta_s) g.fin_dbeta_s[c] = fs[0];
                if (g.fin_dgamma_s) g.fin_dgamma_s[c] = fs[2];
            }
        }
        __syncthreads();
        s1f = fs[0]; s2f = fs[1]; s3f = fs[2];
    } else { s1f = sums[c]; s2f = sums[g.C + c]; if (RES >= 2) s3f = sums_s[g.C + c]; }
    if (training) {
        const float n = (float)g.N * (float)g.HW;
        k1 = s1f / n; k2 = s2f / n;
        if (RES >= 2) { k1s = (g.fin_part ? s1f : sums_s[c]) / n; k2s = s3f / n; }
    }
    for (int64_t i = (int64_t)sp * 256 + threadIdx.x; i < g.n8; i += (int64_t)S * 256) {
        const int64_t off = qa_off8(g, c, (uint32_t)i);
        float v[8], d[8], o[8];
        qa_load8<IN>(in, off, v);
        qa_load8<1>(du, off, d);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float zh, z;
            qa_eval<IN>(v[e], k, zh, z);
            o[e] = k.gi * (d[e] - k1 - zh * k2);
        }
        *reinterpret_cast<float4*>(dy + off) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(dy + off + 4) = make_float4(o[4], o[5], o[6], o[7]);
        if (RES >= 2) {
            float vs[8];
            qa_load8<(RES == 2 ? 0 : 2)>(res, off, vs);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float zh, z;
                qa_eval<0>(vs[e], ks, zh, z);
                o[e] = ks.gi * (d[e] - k1s - zh * k2s);
            }
            *reinterpret_cast<float4*>(dy_s + off) = make_float4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<float4*>(dy_s + off + 4) = make_float4(o[4], o[5], o[6], o[7]);
        }
    }
}
#define QR_DISPATCH_RES(KERNEL, INV, ...)                                              \
    switch (res_kind) {                                                                \
        case 0: hipLaunchKernelGGL((KERNEL<INV, 0>), __VA_ARGS__); break;              \
        case 1: hipLaunchKernelGGL((KERNEL<INV, 1>), __VA_ARGS__); break;              \
        case 2: hipLaunchKernelGGL((KERNEL<INV, 2>), __VA_ARGS__); break;              \
        default: hipLaunchKernelGGL((KERNEL<INV, 3>), __VA_ARGS__); break;             \
    }
#define QR_DISPATCH(KERNEL, ...)                                                       \
    if (in_kind == 0) { QR_DISPATCH_RES(KERNEL, 0, __VA_ARGS__) }                      \
    else if (in_kind == 1) { QR_DISPATCH_RES(KERNEL, 1, __VA_ARGS__) }                 \
    else { QR_DISPATCH_RES(KERNEL, 2, __VA_ARGS__) }
static int qr_check(int in_kind, int res_kind, const void* res, const float* res_chan, const char* what) {
    if (in_kind < 0 || in_kind > 2 || res_kind < 0 || res_kind > 3) MN_FAIL(MN_EINVAL, "%s: bad in_kind / res_kind", what);
    if (res_kind && (!res || (((uintptr_t)res) & 15))) MN_FAIL(MN_EINVAL, "%s: null / misaligned residual", what);
    if (res_kind >= 2 && !res_chan) MN_FAIL(MN_EINVAL, "%s: the shortcut's chan table is required", what);
    return MN_OK;
}
extern "C" int64_t mn_qr_ws_floats(int64_t C) { return C * 32 * 6 + 16; }
extern "C" int mn_qr_fwd(int in_kind, const void* in, const float* chan, int res_kind, const void* res, const float* res_chan, int64_t N, int64_t C, int64_t H, int64_t W,
                         int a_bits, uint8_t* codes, float* act_f32, mn_stream_t stream) {
    QaGeom g;
    int rc = qa_geom(&g, N, C, H, W, a_bits, 0, "mn_qr_fwd");
    if (rc) return rc;
    if ((rc = qr_check(in_kind, res_kind, res, res_chan, "mn_qr_fwd"))) return rc;
    if (!in || !chan || (!codes && !act_f32) || (((uintptr_t)in) & 15) || (codes && (((uintptr_t)codes) & 7)) || (act_f32 && !aligned16(act_f32)))
        MN_FAIL(MN_EINVAL, "mn_qr_fwd: null / misaligned tensor");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)C, (unsigned)qa_split(g));
    const double nel = (double)N * C * H * W;
    mn_set_last_kernel("k_qr_fwd<%d, %d>", in_kind, res_kind);
    mn_prof_bytes(nel * ((in_kind == 0 ? 2.0 : 4.0) + (res_kind == 0 ? 0.0 : (res_kind == 2 ? 2.0 : 4.0)) + (codes ? 1.0 : 0.0) + (act_f32 ? 4.0 : 0.0)));
    mn_prof_begin(s);
    QR_DISPATCH(k_qr_fwd, grid, dim3(256), 0, s, g, in, chan, res, res_chan, (unsigned char*)codes, act_f32)
    mn_prof_end(s);
    MN_CHECK_LAUNCH("mn_qr_fwd");
    return MN_OK;
}
extern "C" int mn_qr_bwd_sums(int in_kind, const void* in, const float* chan, int res_kind, const void* res, const float* res_chan, const float* dq, const float* dq2,
                              const float* g_f32, int64_t N, int64_t C, int64_t H, int64_t W, int a_bits, float* du, float* dgamma, float* dbeta, float* sums,
                              float* dgamma_s, float* dbeta_s, float* sums_s, float* ws, mn_stream_t stream) {
    QaGeom g;
    int rc = qa_geom(&g, N, C, H, W, a_bits, 0, "mn_qr_bwd_sums");
    if (rc) return rc;
    if ((rc = qr_check(in_kind, res_kind, res, res_chan, "mn_qr_bwd_sums"))) return rc;
    if (!in || !chan || (!dq && !g_f32) || (dq2 && !dq) || !du || !sums || !ws || (((uintptr_t)in) & 15) || (dq && !aligned16(dq)) || (dq2 && !aligned16(dq2)) ||
        (g_f32 && !aligned16(g_f32)) || !aligned16(du) || (((uintptr_t)ws) & 7) || (res_kind >= 2 && !sums_s))
        MN_FAIL(MN_EINVAL, "mn_qr_bwd_sums: null / misaligned tensor");
    hipStream_t s = (hipStream_t)stream;
    const int S = qa_split(g);
    const dim3 grid((unsigned)C, (unsigned)S);
    const double nel = (double)N * C * H * W;
    mn_set_last_kernel("k_qr_partial<%d, %d>", in_kind, res_kind);
    mn_prof_bytes(nel * ((in_kind == 0 ? 2.0 : 4.0) + (res_kind == 0 ? 0.0 : (res_kind == 2 ? 2.0 : 4.0)) + (dq ? 4.0 : 0.0) + (dq2 ? 4.0 : 0.0) + (g_f32 ? 4.0 : 0.0) + 4.0));
    mn_prof_begin(s);
    QR_DISPATCH(k_qr_partial, grid, dim3(256), 0, s, g, in, chan, res, res_chan, dq, dq2, g_f32, du, (double*)ws)
    mn_prof_end(s);
    hipLaunchKernelGGL(k_qr_final_bwd, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, s, (int)C, (const double*)ws, S, dgamma, dbeta, sums, res_kind >= 2 ? dgamma_s : (float*)nullptr,
                       res_kind >= 2 ? dbeta_s : (float*)nullptr, res_kind >= 2 ? sums_s : (float*)nullptr);
    MN_CHECK_LAUNCH("mn_qr_bwd_sums");
    return MN_OK;
}
extern "C" int mn_qr_bwd_apply(int in_kind, const void* in, const float* chan, const float* sums, int res_kind, const void* res, const float* res_chan, const float* sums_s,
                               const float* du, int64_t N, int64_t C, int64_t H, int64_t W, int training, float* dy, float* dy_s, mn_stream_t stream) {
    QaGeom g;
    int rc = qa_geom(&g, N, C, H, W, 2, 0, "mn_qr_bwd_apply");
    if (rc) return rc;
    if (res_kind == 1) res_kind = 0;          // an identity shortcut's gradient IS du: nothing to form here
    if ((rc = qr_check(in_kind, res_kind, res, res_chan, "mn_qr_bwd_apply"))) return rc;
    if (!in || !chan || !sums || !du || !dy || (((uintptr_t)in) & 15) || !aligned16(du) || !aligned16(dy) || (res_kind >= 2 && (!dy_s || !sums_s || !aligned16(dy_s))))
        MN_FAIL(MN_EINVAL, "mn_qr_bwd_apply: null / misaligned tensor");
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)C, (unsigned)qa_split(g));
    const double nel = (double)N * C * H * W;
    mn_set_last_kernel("k_qr_apply<%d, %d>", in_kind, res_kind);
    mn_prof_bytes(nel * ((in_kind == 0 ? 2.0 : 4.0) + 8.0 + (res_kind >= 2 ? (res_kind == 2 ? 2.0 : 4.0) + 4.0 : 0.0)));
    mn_prof_begin(s);
    QR_DISPATCH(k_qr_apply, grid, dim3(256), 0, s, g, in, chan, sums, res, res_chan, sums_s, du, training, dy, dy_s)
    mn_prof_end(s);
    MN_CHECK_LAUNCH("mn_qr_bwd_apply");
    return MN_OK;
}
/* mn_qr_bwd_sums + mn_qr_bwd_apply in TWO launches (the apply pass finishes the sums itself, as mn_qa_bwd); arguments as for the two calls */
extern "C" int mn_qr_bwd(int in_kind, const void* in, const float* chan, int res_kind, const void* res, const float* res_chan, const float* dq, const float* dq2,
                         const float* g_f32, int64_t N, int64_t C, int64_t H, int64_t W, int a_bits, int training, float* du, float* dgamma, float* dbeta, float* sums,
                         float* dgamma_s, float* dbeta_s, float* sums_s, float* dy, float* dy_s, float* ws, mn_stream_t stream) {
    QaGeom g;
    int rc = qa_geom(&g, N, C, H, W, a_bits, 0, "mn_qr_bwd");
    if (rc) return rc;
    if ((rc = qr_check(in_kind, res_kind, res, res_chan, "mn_qr_bwd"))) return rc;
    if (!in || !chan || (!dq && !g_f32) || (dq2 && !dq) || !du || !sums || !ws || !dy || (((uintptr_t)in) & 15) || (dq && !aligned16(dq)) || (dq2 && !aligned16(dq2)) ||
        (g_f32 && !aligned16(g_f32)) || !aligned16(du) || !aligned16(dy) || (((uintptr_t)ws) & 7) || (res_kind >= 2 && (!sums_s || !dy_s || !aligned16(dy_s))))
        MN_FAIL(MN_EINVAL, "mn_qr_bwd: null / misaligned tensor");
    hipStream_t s = (hipStream_t)stream;
    const int S = qa_split(g);
    const dim3 grid((unsigned)C, (unsigned)S);
    const double nel = (double)N * C * H * W;
    mn_set_last_kernel("k_qr_partial<%d, %d>", in_kind, res_kind);
    mn_prof_bytes(nel * ((in_kind == 0 ? 2.0 : 4.0) + (res_kind == 0 ? 0.0 : (res_kind == 2 ? 2.0 : 4.0)) + (dq ? 4.0 : 0.0) + (dq2 ? 4.0 : 0.0) + (g_f32 ? 4.0 : 0.0) + 4.0));
    mn_prof_begin(s);
    QR_DISPATCH(k_qr_partial, grid, dim3(256), 0, s, g, in, chan, res, res_chan, dq, dq2, g_f32, du, (double*)ws)
    mn_prof_end(s);
    g.fin_part = (const double*)ws; g.fin_S = S; g.fin_dgamma = dgamma; g.fin_dbeta = dbeta; g.fin_sums = sums;
    if (res_kind >= 2) { g.fin_dgamma_s = dgamma_s; g.fin_dbeta_s = dbeta_s; g.fin_sums_s = sums_s; }
    if (res_kind == 1) res_kind = 0;          // an identity shortcut's gradient IS du: nothing to form in the apply pass
    mn_set_last_kernel("k_qr_apply<%d, %d>", in_kind, res_kind);
    mn_prof_bytes(nel * ((in_kind == 0 ? 2.0 : 4.0) + 8.0 + (res_kind >= 2 ? (res_kind == 2 ? 2.0 : 4.0) + 4.0 : 0.0)));
    mn_prof_begin(s);
    QR_DISPATCH(k_qr_apply, grid, dim3(256), 0, s, g, in, chan, (const float*)sums, res, res_chan, (const float*)sums_s, du, training, dy, dy_s)
    mn_prof_end(s);
    MN_CHECK_LAUNCH("mn_qr_bwd");
    return MN_OK;
}
