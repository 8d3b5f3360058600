// The BN-fused IAO convolution block (QuantBNFuseConv2d.forward in training mode, wqaq/iao/quantize.py:837-994) for the GROUPED 3 x 3 layers of nin_gc
// (models/nin_gc.py:74-79: 3 x 3, stride 1, padding 1, 16 input / 32 output channels per group, 16 x 16 or 8 x 8 maps) whose input lies on an 8-bit quantizer
// grid (the output of QuantMaxPool2d: every value = code * scale).
//
// These layers are tiny next to the pointwise ones (67 / 34 MB in, 134 / 67 MB out at batch 256): the reference's dataflow on general kernels spent its time in
// launch prologues, per-tile staging and ~16 launches per layer and direction.  Here every kernel is PERSISTENT over the images of one group -- the group's
// weights are staged (and term-split / turned into codes) once per block, the 256-pixel image tiles stream through LDS with the next tile's loads in flight -- and
// the whole block is 5 + 2 launches:
//   k_g3_fwd<STATS>   raw convolution (843-851) on the matrix cores WITHOUT writing it: W in three exact bf16 terms x the input's grid codes (one term, exact), per
//                     lane a running (mean, M2) pair per output channel (Chan's update, no cancellation), merged per block in fp64 -> k_g3_stats_finish -> batch
//                     mean / unbiased variance (853-855)
//   k_g3_fwd<QUANT>   quantised convolution (947-955): activation codes (Markstein division, bit-identical to the reference's x / s) x folded-weight codes, the
//                     block's ReLU and the (min, max) partials of the stored activation in the epilogue
//   k_g3_wgrad        backward-weight, used twice: d out x activation codes (the quantised path, + d bias), and d y_raw x grid codes (the statistics path); per
//                     wave an own 64-pixel K slice of every tile, 18 accumulator tiles (32 out x 9 taps x 16 in) live for the block's lifetime -> k_g3_wsum
//   k_g3_fwd<DY>      d y_raw = dmean / n + 2 dvar (y_raw - mean) / (n - 1) (autograd of 853-855) from a recomputed y_raw
//   k_g3_dgrad        backward-data of BOTH paths on one accumulator set: W_q^T (x) d out with the activation quantizer's clip-STE, then W^T (x) d y_raw (six term
//                     products), one store of dx through the channel shuffle
// Real-valued operands are split in three exact bf16 terms (qgemm_kxk.hip), products of two real operands use the six largest term products (iao_bnfuse.hip).
//
// LDS layouts: forward -- channel-innermost patch records [position incl. halo][16 ch] (48-byte records: the 16 pixels of a B fragment, consecutive positions, fall
// on distinct 16-byte bank groups), K = (tap, channel), a K step of 32 = two taps; backward-weight -- K = pixels: A rows [term][out channel][256 px], B rows
// [column shift][in channel][rows incl. halo][W] so that the 8 pixels of a fragment shifted by a tap are one aligned 16-byte read; backward-data -- patch records
// [position][term][32 out channels] (208 bytes), K step = one tap x 32 channels, transposed / flipped weights as A.
#include "qgemm_dev.h"

#include <stdlib.h>

#define G3_PSB 48          // bytes per forward patch record: 16 channels bf16 + 16 pad
#define G3_LDW 168         // u16 per forward weight row: 9 taps x 16 channels (+ one zero half step) + 8 pad
#define G3_LDA 264         // u16 per backward-weight A row: 256 pixels + 8 pad
#define G3_GSB 208         // bytes per backward-data patch record: 3 terms x 32 channels bf16 + 16 pad
#define G3_LDT 296         // u16 per transposed weight row: 9 taps x 32 channels + 8 pad

struct G3Geom {
    int N, W, wsh, hwsh, HW, IMG, PH, PW, npos, C_total, O_total, G, tiles, NB;
    ChanMap in_map;
};
__device__ __forceinline__ void g3_pix(const G3Geom& m, int pix, int& img, int& row, int& col) {
    img = pix >> m.hwsh;
    const int rem = pix & (m.HW - 1);
    row = rem >> m.wsh;
    col = rem & (m.W - 1);
}
__device__ __forceinline__ uint16_t g3_bf(float v) { return (uint16_t)(mn_f2u(v) >> 16); }
__device__ __forceinline__ void g3_split(float v, float& t0, float& t1, float& t2) {
    t0 = mn_bf16_head(v);
    const float r1 = v - t0;
    t1 = mn_bf16_head(r1);
    t2 = r1 - t1;
}

// ------------------------------------------------------------------------------------------------ forward family
#define G3_STATS 0
#define G3_DY 1
#define G3_QUANT 2
struct G3FParams {
    const float* x;
    const float* w;          // STATS / DY: the raw weights [O][16][3][3]; QUANT: the fake-quantised folded weights
    const float* bias;       // STATS / DY: the conv bias (nullable); QUANT: the folded bias
    const float* xqp;        // {scale, zero point, ...} of the quantizer whose codes are contracted: the input's grid (STATS / DY), the activation quantizer (QUANT)
    const float* wqp;        // QUANT: [O][4] per-channel weight qparams
    const float* stats;      // DY: [2][O] batch mean / var
    const float* coef;       // DY: [4][O] {dmean / n, 2 dvar / (n - 1), ...}
    double* part;            // STATS: [G * NB][32][3] (n, mean, M2)
    float* out;              // QUANT: the block's output; DY: d y_raw
    float* mm;               // QUANT: (min, max) partials [2 * grid], nullable
    float qmin, qmax;
    int relu;
    G3Geom m;
};
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_g3_fwd(const G3FParams p) {
    HIP_DYNAMIC_SHARED(float, smem)
    constexpr int NTT = MODE == G3_QUANT ? 2 : 6;
    const G3Geom& m = p.m;
    char* patch = reinterpret_cast<char*>(smem);                                   // [2][npos][G3_PSB]
    const int PB = m.npos * G3_PSB;
    uint16_t* wsm = reinterpret_cast<uint16_t*>(patch + 2 * PB);                   // [NTT * 16][G3_LDW]: row = term * 32 + out channel
    float* cst = reinterpret_cast<float*>(wsm + NTT * 16 * G3_LDW);                // [5][32] per-channel constants, [16] reduction scratch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kg = lane >> 4;
    const int g = blockIdx.x / m.NB, b = blockIdx.x - g * m.NB;
    const float sc = p.xqp[0], zp = p.xqp[1], inv_sc = 1.0f / sc;

    for (int i = tid; i < 2 * PB / 16; i += 256) reinterpret_cast<u32x4*>(patch)[i] = u32x4{0u, 0u, 0u, 0u};          // the halo stays zero
    for (int i = tid; i < NTT * 16 * G3_LDW / 8; i += 256) reinterpret_cast<u32x4*>(wsm)[i] = u32x4{0u, 0u, 0u, 0u};
    __syncthreads();
    for (int i = tid; i < 32 * 144; i += 256) {
        const int o = i / 144, k = i - o * 144, c = k / 9, tap = k - c * 9;
        const float v = p.w[(int64_t)(g * 32 + o) * 144 + k];
        uint16_t* d = wsm + o * G3_LDW + tap * 16 + c;
        if (MODE == G3_QUANT) {
            d[0] = g3_bf(rintf(v / p.wqp[4 * (g * 32 + o)]));          // (clamp(r) + zp): an integer, |.| <= 255
        } else {
            float t0, t1, t2;
            g3_split(v, t0, t1, t2);
            d[0] = g3_bf(t0); d[32 * G3_LDW] = g3_bf(t1); d[64 * G3_LDW] = g3_bf(t2);
        }
    }
    if (tid < 32) {
        const int o = g * 32 + tid;
        cst[tid] = MODE == G3_QUANT ? sc * p.wqp[4 * o] : sc;
        cst[32 + tid] = p.bias ? p.bias[o] : 0.f;
        if (MODE == G3_DY) { cst[64 + tid] = p.coef[o]; cst[96 + tid] = p.coef[m.O_total + o]; cst[128 + tid] = p.stats[o]; }
    }
    // staging item of this thread: 4 channels x 4 consecutive pixels of one row
    const int cq_sh = m.wsh - 2;
    const int s_cqd = tid & ((1 << cq_sh) - 1);
    const int s_t1 = tid >> cq_sh, s_cq = s_t1 & 3, s_t2 = s_t1 >> 2, s_row = s_t2 & (m.W - 1), s_img = s_t2 >> m.wsh;
    int64_t s_goff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) s_goff[i] = (int64_t)chan_phys(m.in_map, g * 16 + s_cq * 4 + i) * m.HW + s_row * m.W + s_cqd * 4;
    const int64_t img_stride = (int64_t)m.C_total * m.HW;
    const int s_pos = ((s_img * m.PH + s_row + 1) * m.PW + s_cqd * 4 + 1) * G3_PSB + s_cq * 8;
    float4 rx[4];
    auto fetch = [&](int tile) {
        const int64_t base = (int64_t)(tile * m.IMG + s_img) * img_stride;
#pragma unroll
        for (int i = 0; i < 4; ++i) rx[i] = *reinterpret_cast<const float4*>(p.x + base + s_goff[i]);
    };
    auto commit = [&](int buf) {
        char* d = patch + buf * PB + s_pos;
        const float v[4][4] = {{rx[0].x, rx[0].y, rx[0].z, rx[0].w}, {rx[1].x, rx[1].y, rx[1].z, rx[1].w},
                               {rx[2].x, rx[2].y, rx[2].z, rx[2].w}, {rx[3].x, rx[3].y, rx[3].z, rx[3].w}};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float c[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = iao_code_m(v[i][e], sc, inv_sc, zp, p.qmin, p.qmax);
            *reinterpret_cast<u32x2*>(d + e * G3_PSB) = u32x2{mn_pack_bf16x2(c[0], c[1]), mn_pack_bf16x2(c[2], c[3])};
        }
    };
    // MFMA operands of this lane: pixel 64 wave + 16 q + j of the tile; K step ks = taps 2 ks, 2 ks + 1 (tap 9 does not exist: its weight columns are zero, the
    // lane reads tap 0's record -- finite codes)
    int pos0[4], orem[4], oimg[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int img, row, col;
        g3_pix(m, wave * 64 + 16 * q + j, img, row, col);
        pos0[q] = ((img * m.PH + row) * m.PW + col) * G3_PSB;
        orem[q] = row * m.W + col;
        oimg[q] = img;
    }
    int toff[5];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        int tap = 2 * ks + (kg >> 1);
        if (tap > 8) tap = 0;
        toff[ks] = ((tap / 3) * m.PW + tap % 3) * G3_PSB + (kg & 1) * 16;
    }
    float mean_[8], m2_[8], lo = INFINITY, hi = -INFINITY;
    int mnan = 0;          // (plain min / max ignore a NaN: it is flagged and propagated at the end, as torch.min / max would)
#pragma unroll
    for (int s = 0; s < 8; ++s) { mean_[s] = 0.f; m2_[s] = 0.f; }

    int it = 0;
    if (b < m.tiles) fetch(b);
    for (int tile = b; tile < m.tiles; tile += m.NB, ++it) {
        const int buf = it & 1;
        commit(buf);
        __syncthreads();          // (one barrier per tile: the buffer written now was last read two iterations ago)
        if (tile + m.NB < m.tiles) fetch(tile + m.NB);
        f32x4 acc[4][NTT];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int t = 0; t < NTT; ++t) acc[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const char* pb = patch + buf * PB;
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
            u32x4 bf[4], af[NTT];
#pragma unroll
            for (int q = 0; q < 4; ++q) bf[q] = *reinterpret_cast<const u32x4*>(pb + pos0[q] + toff[ks]);
#pragma unroll
            for (int t = 0; t < NTT; ++t) af[t] = *reinterpret_cast<const u32x4*>(wsm + (t * 16 + j) * G3_LDW + ks * 32 + kg * 8);
#pragma unroll
            for (int t = 0; t < NTT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q][t] = mn_mfma_bf16(af[t], bf[q], acc[q][t]);
        }
        // lane (j, kg) holds out channels oh * 16 + 4 kg + r of its four pixels
#pragma unroll
        for (int oh = 0; oh < 2; ++oh)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ol = oh * 16 + 4 * kg + r;
                float y[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (MODE == G3_QUANT) y[q] = acc[q][oh][r] * cst[ol] + cst[32 + ol];
                    else y[q] = ((acc[q][4 + oh][r] + acc[q][2 + oh][r]) + acc[q][oh][r]) * cst[ol] + cst[32 + ol];
                }
                if (MODE == G3_STATS) {
                    const int s = oh * 4 + r;
                    const float m4 = ((y[0] + y[1]) + (y[2] + y[3])) * 0.25f;
                    const float d0 = y[0] - m4, d1 = y[1] - m4, d2 = y[2] - m4, d3 = y[3] - m4;
                    const float q4 = (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                    if (it == 0) { mean_[s] = m4; m2_[s] = q4; }
                    else {          // Chan's merge of (4 it, mean, M2) with (4, m4, q4)
                        const float dl = m4 - mean_[s], f1 = 1.0f / (float)(it + 1), f2 = 4.0f * (float)it * f1;
                        mean_[s] += dl * f1;
                        m2_[s] += q4 + dl * dl * f2;
                    }
                } else {
                    if (MODE == G3_DY) {
                        const float mu = cst[128 + ol];
#pragma unroll
                        for (int q = 0; q < 4; ++q) y[q] = cst[64 + ol] + cst[96 + ol] * (y[q] - mu);
                    } else if (p.relu) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) y[q] = qa_relu(y[q]);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        p.out[((int64_t)(tile * m.IMG + oimg[q]) * m.O_total + g * 32 + ol) * m.HW + orem[q]] = y[q];
                        if (MODE == G3_QUANT) { lo = fminf(lo, y[q]); hi = fmaxf(hi, y[q]); mnan |= (int)(y[q] != y[q]); }
                    }
                }
            }
    }
    if (MODE == G3_STATS) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(patch);          // [256][8][2]
#pragma unroll
        for (int s = 0; s < 8; ++s) { red[(tid * 8 + s) * 2] = mean_[s]; red[(tid * 8 + s) * 2 + 1] = m2_[s]; }
        __syncthreads();
        if (tid < 32) {
            const int oh = tid >> 4, kgo = (tid & 15) >> 2, r = tid & 3, s = oh * 4 + r;
            double n = 0.0, mu = 0.0, M2 = 0.0;
            const double ni = 4.0 * (double)it;
            if (it > 0)
                for (int w_ = 0; w_ < 4; ++w_)
                    for (int jj = 0; jj < 16; ++jj) {
                        const int t = w_ * 64 + kgo * 16 + jj;
                        const double a = (double)red[(t * 8 + s) * 2], q = (double)red[(t * 8 + s) * 2 + 1];
                        if (n == 0.0) { n = ni; mu = a; M2 = q; }
                        else {
                            const double d = a - mu, nn = n + ni;
                            mu += d * ni / nn;
                            M2 += q + d * d * n * ni / nn;
                            n = nn;
                        }
                    }
            double* o_ = p.part + ((int64_t)blockIdx.x * 32 + tid) * 3;
            o_[0] = n; o_[1] = mu; o_[2] = M2;
        }
    }
    if (MODE == G3_QUANT && p.mm) {
        if (mnan) lo = hi = NAN;
        lo = block_reduce(lo, OpMinF(), INFINITY, cst + 160);
        hi = block_reduce(hi, OpMaxF(), -INFINITY, cst + 160);
        if (tid == 0) { p.mm[blockIdx.x] = lo; p.mm[gridDim.x + blockIdx.x] = hi; }
    }
}

// merge of the per-block (n, mean, M2) triples: batch mean and unbiased variance of the raw convolution's output
__global__ __launch_bounds__(64) void k_g3_stats_finish(const double* __restrict__ part, int NB, int O, float* __restrict__ stats) {
    const int o = blockIdx.x * 64 + threadIdx.x;
    if (o >= O) return;
    const int g = o >> 5, ol = o & 31;
    double n = 0.0, mu = 0.0, M2 = 0.0;
    for (int b = 0; b < NB; ++b) {
        const double* q = part + ((int64_t)(g * NB + b) * 32 + ol) * 3;
        const double ni = q[0];
        if (ni == 0.0) continue;
        if (n == 0.0) { n = ni; mu = q[1]; M2 = q[2]; }
        else {
            const double d = q[1] - mu, nn = n + ni;
            mu += d * ni / nn;
            M2 += q[2] + d * d * n * ni / nn;
            n = nn;
        }
    }
    stats[o] = (float)mu;
    stats[O + o] = (float)(M2 / (n - 1.0));
}

// ------------------------------------------------------------------------------------------------ backward-weight
struct G3WParams {
    const float* a;          // d out (quantised path) or d y_raw (statistics path)  [N][O_total][HW]
    const float* mask;       // nullable: the block's rectified output -- a *= [mask > 0]
    const float* x;
    const float* xqp;        // quantizer whose codes of x are contracted
    float qmin, qmax;
    float* part;             // [G * NB * 4][32][9][16]
    float* dbpart;           // [G * NB][32], nullable
    G3Geom m;
};
__global__ __launch_bounds__(256, 1) void k_g3_wgrad(const G3WParams p) {
    HIP_DYNAMIC_SHARED(float, smem)
    const G3Geom& m = p.m;
    uint16_t* as = reinterpret_cast<uint16_t*>(smem);          // [3][32][G3_LDA]
    uint16_t* xs = as + 3 * 32 * G3_LDA;                       // [3 column shifts][16][CS]
    const int CS = m.IMG * m.PH * m.W + 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kg = lane >> 4;
    const int g = blockIdx.x / m.NB, b = blockIdx.x - g * m.NB;
    const float sc = p.xqp[0], zp = p.xqp[1], inv_sc = 1.0f / sc;
    for (int i = tid; i < 3 * 16 * CS / 8; i += 256) reinterpret_cast<u32x4*>(xs)[i] = u32x4{0u, 0u, 0u, 0u};          // halo rows stay zero

    const int cq_sh = m.wsh - 2, cqn = 1 << cq_sh;
    const int s_cqd = tid & (cqn - 1);
    const int s_t1 = tid >> cq_sh, s_cq = s_t1 & 3, s_t2 = s_t1 >> 2, s_row = s_t2 & (m.W - 1), s_img = s_t2 >> m.wsh;
    int64_t s_goff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) s_goff[i] = (int64_t)chan_phys(m.in_map, g * 16 + s_cq * 4 + i) * m.HW + s_row * m.W + s_cqd * 4;
    const int64_t img_stride = (int64_t)m.C_total * m.HW;
    const int s_xo = (s_cq * 4) * CS + (s_img * m.PH + s_row + 1) * m.W + s_cqd * 4;
    // A item: out channel wave + 4 u, pixels 4 lane .. 4 lane + 3
    int a_img, a_row, a_col;
    g3_pix(m, 4 * lane, a_img, a_row, a_col);
    const int64_t a_goff = (int64_t)a_img * m.O_total * m.HW + (int64_t)(g * 32 + wave) * m.HW + a_row * m.W + a_col;
    float4 rx[4], ra[8], rk[8];
    auto fetch = [&](int tile) {
        const int64_t base = (int64_t)(tile * m.IMG + s_img) * img_stride;
#pragma unroll
        for (int i = 0; i < 4; ++i) rx[i] = *reinterpret_cast<const float4*>(p.x + base + s_goff[i]);
        const int64_t ab = (int64_t)tile * m.IMG * m.O_total * m.HW + a_goff;
#pragma unroll
        for (int u = 0; u < 8; ++u) ra[u] = *reinterpret_cast<const float4*>(p.a + ab + (int64_t)4 * u * m.HW);
        if (p.mask) {
#pragma unroll
            for (int u = 0; u < 8; ++u) rk[u] = *reinterpret_cast<const float4*>(p.mask + ab + (int64_t)4 * u * m.HW);
        }
    };
    float db[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) db[u] = 0.f;
    auto commit = [&]() {
        const float v[4][4] = {{rx[0].x, rx[0].y, rx[0].z, rx[0].w}, {rx[1].x, rx[1].y, rx[1].z, rx[1].w},
                               {rx[2].x, rx[2].y, rx[2].z, rx[2].w}, {rx[3].x, rx[3].y, rx[3].z, rx[3].w}};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float c[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) c[e] = iao_code_m(v[i][e], sc, inv_sc, zp, p.qmin, p.qmax);
            // the neighbouring quads of the same row sit in the neighbouring lanes (the column quad is the fastest index of the item)
            float lf = __shfl(c[3], (lane + 63) & 63, 64), rt = __shfl(c[0], (lane + 1) & 63, 64);
            if (s_cqd == 0) lf = 0.f;
            if (s_cqd == cqn - 1) rt = 0.f;
            uint16_t* d = xs + s_xo + i * CS;
            *reinterpret_cast<u32x2*>(d) = u32x2{mn_pack_bf16x2(lf, c[0]), mn_pack_bf16x2(c[1], c[2])};                       // tap column 0: x[col - 1]
            *reinterpret_cast<u32x2*>(d + 16 * CS) = u32x2{mn_pack_bf16x2(c[0], c[1]), mn_pack_bf16x2(c[2], c[3])};          // tap column 1
            *reinterpret_cast<u32x2*>(d + 32 * CS) = u32x2{mn_pack_bf16x2(c[1], c[2]), mn_pack_bf16x2(c[3], rt)};            // tap column 2: x[col + 1]
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float a4[4] = {ra[u].x, ra[u].y, ra[u].z, ra[u].w};
            if (p.mask) {
                const float k4[4] = {rk[u].x, rk[u].y, rk[u].z, rk[u].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) a4[e] = k4[e] > 0.f ? a4[e] : 0.f;
            }
            db[u] += (a4[0] + a4[1]) + (a4[2] + a4[3]);
            float t0[4], t1[4], t2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) g3_split(a4[e], t0[e], t1[e], t2[e]);
            uint16_t* d = as + (wave + 4 * u) * G3_LDA + 4 * lane;
            *reinterpret_cast<u32x2*>(d) = u32x2{mn_pack_bf16x2(t0[0], t0[1]), mn_pack_bf16x2(t0[2], t0[3])};
            *reinterpret_cast<u32x2*>(d + 32 * G3_LDA) = u32x2{mn_pack_bf16x2(t1[0], t1[1]), mn_pack_bf16x2(t1[2], t1[3])};
            *reinterpret_cast<u32x2*>(d + 64 * G3_LDA) = u32x2{mn_pack_bf16x2(t2[0], t2[1]), mn_pack_bf16x2(t2[2], t2[3])};
        }
    };
    // this wave's K slice: pixel octets 8 wave .. 8 wave + 7 (K steps 2 wave, 2 wave + 1)
    int bb[2], ab_[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int ks = 2 * wave + s;
        int img, row, col;
        g3_pix(m, 8 * (4 * ks + kg), img, row, col);
        bb[s] = j * CS + (img * m.PH + row) * m.W + col;
        ab_[s] = j * G3_LDA + ks * 32 + kg * 8;
    }
    f32x4 acc[2][9];
#pragma unroll
    for (int oh = 0; oh < 2; ++oh)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[oh][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (b < m.tiles) fetch(b);
    for (int tile = b; tile < m.tiles; tile += m.NB) {
        __syncthreads();          // the previous tile's fragments are consumed
        commit();
        __syncthreads();
        if (tile + m.NB < m.tiles) fetch(tile + m.NB);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 bf[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) bf[t] = *reinterpret_cast<const u32x4*>(xs + bb[s] + (t % 3) * 16 * CS + (t / 3) * m.W);
#pragma unroll
            for (int term = 2; term >= 0; --term) {
                const u32x4 a0 = *reinterpret_cast<const u32x4*>(as + term * 32 * G3_LDA + ab_[s]);
                const u32x4 a1 = *reinterpret_cast<const u32x4*>(as + (term * 32 + 16) * G3_LDA + ab_[s]);
#pragma unroll
                for (int t = 0; t < 9; ++t) { acc[0][t] = mn_mfma_bf16(a0, bf[t], acc[0][t]); acc[1][t] = mn_mfma_bf16(a1, bf[t], acc[1][t]); }
            }
        }
    }
    // per-wave partial [32 out][9 taps][16 in]: lane (j, kg) holds rows 4 kg + r (out channel within the 16-row tile), column j (in channel)
    float* po = p.part + ((int64_t)blockIdx.x * 4 + wave) * (32 * 144);
#pragma unroll
    for (int oh = 0; oh < 2; ++oh)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) po[((oh * 16 + 4 * kg + r) * 9 + t) * 16 + j] = acc[oh][t][r];
    if (p.dbpart) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float s = wave_reduce(db[u], OpAddF());
            if (lane == 0) p.dbpart[(int64_t)blockIdx.x * 32 + wave + 4 * u] = s;
        }
    }
}
// dw[o][c][tap] (+)= scale * sum of the 4 NB per-wave partials (fixed order, fp64); dbias[o] = sum of the NB block partials
__global__ __launch_bounds__(256) void k_g3_wsum(const float* __restrict__ part, const float* __restrict__ dbpart, const float* __restrict__ xqp, int NB, int O,
                                                 int accumulate, float* __restrict__ dw, float* __restrict__ dbias) {
    const int i = blockIdx.x * 256 + threadIdx.x;          // (o, tap, c) in partial order
    if (i < O * 144) {
        const int o = i / 144, k = i - o * 144, tap = k >> 4, c = k & 15;
        const int g = o >> 5, ol = o & 31;
        const float* q = part + (int64_t)g * NB * 4 * (32 * 144) + ol * 144 + k;
        double s = 0.0;
        for (int z = 0; z < 4 * NB; ++z) s += (double)q[(int64_t)z * (32 * 144)];
        const float v = (float)s * xqp[0];
        float* d = dw + (int64_t)o * 144 + c * 9 + tap;
        *d = accumulate ? *d + v : v;
    }
    if (dbias && i < O) {
        const int g = i >> 5, ol = i & 31;
        double s = 0.0;
        for (int z = 0; z < NB; ++z) s += (double)dbpart[((int64_t)g * NB + z) * 32 + ol];
        dbias[i] = (float)s;
    }
}

// ------------------------------------------------------------------------------------------------ backward-data
struct G3DParams {
    const float* gy;         // d out
    const float* mask;       // nullable: the block's rectified output
    const float* dy;         // d y_raw
    const float* x;
    const float* aqp;        // activation quantizer {scale, zp, lo, hi}
    float qmin, qmax;
    const float* qw;         // fake-quantised folded weights
    const float* wqp;        // [O][4]
    const float* w;          // raw weights
    float* dx;
    int relu_in;             // x is the output of a ReLU whose mask is applied here
    G3Geom m;
};
__global__ __launch_bounds__(256, 1) void k_g3_dgrad(const G3DParams p) {
    HIP_DYNAMIC_SHARED(float, smem)
    const G3Geom& m = p.m;
    char* gp = reinterpret_cast<char*>(smem);                                       // [npos][G3_GSB]
    uint16_t* wtq = reinterpret_cast<uint16_t*>(gp + m.npos * G3_GSB);              // [16][G3_LDT]: row = in channel, k = flipped tap * 32 + out channel
    uint16_t* wtr = wtq + 16 * G3_LDT;                                              // [3][16][G3_LDT]
    float* sw = reinterpret_cast<float*>(wtr + 3 * 16 * G3_LDT);                    // [32] weight scales
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kg = lane >> 4;
    const int g = blockIdx.x / m.NB, b = blockIdx.x - g * m.NB;
    const float sc = p.aqp[0], zp = p.aqp[1], slo = p.aqp[2], shi = p.aqp[3], inv_sc = 1.0f / sc;

    for (int i = tid; i < m.npos * G3_GSB / 16; i += 256) reinterpret_cast<u32x4*>(gp)[i] = u32x4{0u, 0u, 0u, 0u};
    for (int i = tid; i < 32 * 144; i += 256) {
        const int o = i / 144, k = i - o * 144, c = k / 9, tap = k - c * 9;
        const int col = (8 - tap) * 32 + o;
        const int64_t wi = (int64_t)(g * 32 + o) * 144 + k;
        wtq[c * G3_LDT + col] = g3_bf(rintf(p.qw[wi] / p.wqp[4 * (g * 32 + o)]));
        float t0, t1, t2;
        g3_split(p.w[wi], t0, t1, t2);
        wtr[c * G3_LDT + col] = g3_bf(t0); wtr[(16 + c) * G3_LDT + col] = g3_bf(t1); wtr[(32 + c) * G3_LDT + col] = g3_bf(t2);
    }
    if (tid < 32) sw[tid] = p.wqp[4 * (g * 32 + tid)];

    // staging item: pixel tid of the tile, all 32 out channels (4 x 8 loads of one float, coalesced along the pixels)
    int s_img, s_row, s_col;
    g3_pix(m, tid, s_img, s_row, s_col);
    const int s_pos = ((s_img * m.PH + s_row + 1) * m.PW + s_col + 1) * G3_GSB;
    const int64_t s_goff = (int64_t)s_img * m.O_total * m.HW + (int64_t)g * 32 * m.HW + s_row * m.W + s_col;
    float rg[32];
    auto fetch_g = [&](int tile) {
        const int64_t base = (int64_t)tile * m.IMG * m.O_total * m.HW + s_goff;
#pragma unroll
        for (int o = 0; o < 32; ++o) rg[o] = p.gy[base + (int64_t)o * m.HW];
        if (p.mask) {
#pragma unroll
            for (int o = 0; o < 32; ++o) rg[o] = p.mask[base + (int64_t)o * m.HW] > 0.f ? rg[o] : 0.f;
        }
    };
    auto fetch_d = [&](int tile) {
        const int64_t base = (int64_t)tile * m.IMG * m.O_total * m.HW + s_goff;
#pragma unroll
        for (int o = 0; o < 32; ++o) rg[o] = p.dy[base + (int64_t)o * m.HW];
    };
    auto commit = [&](bool scaled) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float t0[8], t1[8], t2[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) g3_split(scaled ? rg[8 * u + i] * sw[8 * u + i] : rg[8 * u + i], t0[i], t1[i], t2[i]);
            char* d = gp + s_pos + u * 16;
            *reinterpret_cast<u32x4*>(d) = u32x4{mn_pack_bf16x2(t0[0], t0[1]), mn_pack_bf16x2(t0[2], t0[3]), mn_pack_bf16x2(t0[4], t0[5]), mn_pack_bf16x2(t0[6], t0[7])};
            *reinterpret_cast<u32x4*>(d + 64) = u32x4{mn_pack_bf16x2(t1[0], t1[1]), mn_pack_bf16x2(t1[2], t1[3]), mn_pack_bf16x2(t1[4], t1[5]), mn_pack_bf16x2(t1[6], t1[7])};
            *reinterpret_cast<u32x4*>(d + 128) = u32x4{mn_pack_bf16x2(t2[0], t2[1]), mn_pack_bf16x2(t2[2], t2[3]), mn_pack_bf16x2(t2[4], t2[5]), mn_pack_bf16x2(t2[6], t2[7])};
        }
    };
    int pos0[4], orem[4], oimg[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int img, row, col;
        g3_pix(m, wave * 64 + 16 * q + j, img, row, col);
        pos0[q] = ((img * m.PH + row) * m.PW + col) * G3_GSB + kg * 16;
        orem[q] = row * m.W + col;
        oimg[q] = img;
    }
    int64_t xoff[4];          // this lane's output rows: in channels 4 kg + r
#pragma unroll
    for (int r = 0; r < 4; ++r) xoff[r] = (int64_t)chan_phys(m.in_map, g * 16 + 4 * kg + r) * m.HW;
    const int64_t img_stride = (int64_t)m.C_total * m.HW;

    if (b < m.tiles) fetch_g(b);
    for (int tile = b; tile < m.tiles; tile += m.NB) {
        __syncthreads();          // (first pass: zeroed halo and weights visible; later: the previous tile's fragments are consumed)
        commit(true);
        __syncthreads();
        fetch_d(tile);
        float xv[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) xv[q][r] = p.x[(int64_t)(tile * m.IMG + oimg[q]) * img_stride + xoff[r] + orem[q]];
        f32x4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        // ---- quantised path: d out (three terms, pre-scaled by the weight scale of its channel) x transposed weight codes
#pragma unroll
        for (int ks = 0; ks < 9; ++ks) {
            const int to = ((ks / 3) * m.PW + ks % 3) * G3_GSB;
            const u32x4 a = *reinterpret_cast<const u32x4*>(wtq + j * G3_LDT + ks * 32 + kg * 8);
#pragma unroll
            for (int term = 2; term >= 0; --term)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = mn_mfma_bf16(a, *reinterpret_cast<const u32x4*>(gp + pos0[q] + to + term * 64), acc[q]);
        }
        // the activation quantizer's clip-STE (ref 163-168, 232) on the quantised path only
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[q][r] = iao_fq_grad_m(acc[q][r], xv[q][r], sc, inv_sc, zp, slo, shi, p.qmin, p.qmax);
        __syncthreads();
        commit(false);
        __syncthreads();
        if (tile + m.NB < m.tiles) fetch_g(tile + m.NB);
        // ---- statistics path: d y_raw x the raw weights, six term products, smallest first
#pragma unroll
        for (int ks = 0; ks < 9; ++ks) {
            const int to = ((ks / 3) * m.PW + ks % 3) * G3_GSB;
            u32x4 a[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) a[t] = *reinterpret_cast<const u32x4*>(wtr + (t * 16 + j) * G3_LDT + ks * 32 + kg * 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const char* rec = gp + pos0[q] + to;
                const u32x4 b0 = *reinterpret_cast<const u32x4*>(rec), b1 = *reinterpret_cast<const u32x4*>(rec + 64), b2 = *reinterpret_cast<const u32x4*>(rec + 128);
                acc[q] = mn_mfma_bf16(a[0], b2, acc[q]);
                acc[q] = mn_mfma_bf16(a[2], b0, acc[q]);
                acc[q] = mn_mfma_bf16(a[1], b1, acc[q]);
                acc[q] = mn_mfma_bf16(a[0], b1, acc[q]);
                acc[q] = mn_mfma_bf16(a[1], b0, acc[q]);
                acc[q] = mn_mfma_bf16(a[0], b0, acc[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[q][r];
                if (p.relu_in) v = xv[q][r] > 0.f ? v : 0.f;
                p.dx[(int64_t)(tile * m.IMG + oimg[q]) * img_stride + xoff[r] + orem[q]] = v;
            }
    }
}

// ------------------------------------------------------------------------------------------------ host side
static int g3_nb_cap() {          // A/B knob (and the tests' way to several tiles per block at small batch): the grid every g3 kernel aims at
    const char* e = getenv("MN_G3_BLOCKS");
    const int n = e ? atoi(e) : 0;
    return n > 0 ? n : 0;
}
// geometry check + tiling; target_blocks: the grid the kernel wants (2 blocks per CU for the forward family, 1 for the backward kernels)
static int plan_g3(const mn_conv_geom* g, int target_blocks, G3Geom* m) {
    if (!g || g->N <= 0 || g->KH != 3 || g->KW != 3 || g->stride_h != 1 || g->stride_w != 1 || g->pad_h != 1 || g->pad_w != 1 || g->dil_h != 1 || g->dil_w != 1) return 0;
    if (g->groups < 1 || g->C != 16 * g->groups || g->O != 32 * g->groups) return 0;
    if (g->H != g->W || (g->W != 8 && g->W != 16)) return 0;
    const int HW = g->H * g->W, IMG = 256 / HW;
    if (g->N % IMG) return 0;
    if (g->in_shuffle > 1 && g->C % g->in_shuffle) return 0;
    if ((int64_t)g->N * g->O * HW >= (1ll << 31)) return 0;
    if (!m) return 1;
    m->N = g->N; m->W = g->W; m->wsh = g->W == 16 ? 4 : 3; m->hwsh = 2 * m->wsh; m->HW = HW; m->IMG = IMG; m->PH = g->H + 2; m->PW = g->W + 2;
    m->npos = IMG * m->PH * m->PW; m->C_total = g->C; m->O_total = g->O; m->G = g->groups; m->tiles = g->N / IMG;
    m->in_map = make_chanmap(g->in_shuffle, g->C);
    int want = g3_nb_cap() > 0 ? g3_nb_cap() : target_blocks;
    int nb = want / g->groups;
    if (nb < 1) nb = 1;
    if (nb > m->tiles) nb = m->tiles;
    while (m->tiles % nb) --nb;          // equal tile counts per block (the statistics merge assumes nothing, the tails are simply avoided)
    m->NB = nb;
    return 1;
}
static size_t g3_fwd_lds(const G3Geom& m, int ntt) { return (size_t)2 * m.npos * G3_PSB + (size_t)ntt * 16 * G3_LDW * 2 + (160 + 16) * 4; }
static size_t g3_wg_lds(const G3Geom& m) { return (size_t)3 * 32 * G3_LDA * 2 + (size_t)3 * 16 * (m.IMG * m.PH * m.W + 8) * 2; }
static size_t g3_dg_lds(const G3Geom& m) { return (size_t)m.npos * G3_GSB + (size_t)4 * 16 * G3_LDT * 2 + 32 * 4; }
#define G3_FWD_BLOCKS 512
#define G3_BWD_BLOCKS 256

extern "C" int mn_iaobf_g3_supported(const mn_conv_geom* g) { return plan_g3(g, G3_FWD_BLOCKS, nullptr); }
// workspace: statistics partials / backward-weight partials (the larger of the two)
extern "C" int64_t mn_iaobf_g3_ws_bytes(const mn_conv_geom* g) {
    G3Geom mf, mb;
    if (!plan_g3(g, G3_FWD_BLOCKS, &mf) || !plan_g3(g, G3_BWD_BLOCKS, &mb)) return 0;
    const int64_t st = (int64_t)mf.G * mf.NB * 32 * 3 * 8;
    const int64_t wg = (int64_t)mb.G * mb.NB * 4 * (32 * 144) * 4 + (int64_t)mb.G * mb.NB * 32 * 4;
    return (st > wg ? st : wg) + 256;
}
extern "C" int64_t mn_iaobf_g3_mm_count(const mn_conv_geom* g) {
    G3Geom m;
    return plan_g3(g, G3_FWD_BLOCKS, &m) ? (int64_t)m.G * m.NB : 0;
}
static int g3_bits_ok(int bits) { return bits >= 2 && bits <= 8; }

// batch mean / unbiased variance of conv2d(x, w, bias) without writing it; xgrid = {scale, 0, ..} of the symmetric `grid_bits`-bit quantizer x lies on
extern "C" int mn_iaobf_g3_stats(const mn_conv_geom* g, const float* x, const float* xgrid, int grid_bits, const float* w, const float* bias, float* stats, void* ws,
                                 int64_t ws_bytes, mn_stream_t stream) {
    G3FParams p;
    if (!plan_g3(g, G3_FWD_BLOCKS, &p.m)) MN_FAIL(MN_ENOTSUP, "mn_iaobf_g3_stats: geometry not covered (3x3 / stride 1 / padding 1, 16 -> 32 channels per group, 8x8 or 16x16)");
    if (!x || !xgrid || !w || !stats || !ws || !g3_bits_ok(grid_bits) || ws_bytes < mn_iaobf_g3_ws_bytes(g)) MN_FAIL(MN_EINVAL, "mn_iaobf_g3_stats: bad arguments");
    p.x = x; p.w = w; p.bias = bias; p.xqp = xgrid; p.wqp = nullptr; p.stats = nullptr; p.coef = nullptr; p.part = (double*)ws; p.out = nullptr; p.mm = nullptr; p.relu = 0;
    const IaoRange r = iao_range(grid_bits, 0, 1);
    p.qmin = r.qmin; p.qmax = r.qmax;
    const size_t lds = g3_fwd_lds(p.m, 6);
    raise_lds_limit((const void*)k_g3_fwd<G3_STATS>, lds);
    mn_set_last_kernel("k_g3_fwd<STATS>");
    hipLaunchKernelGGL(k_g3_fwd<G3_STATS>, dim3((unsigned)(p.m.G * p.m.NB)), dim3(256), lds, (hipStream_t)stream, p);
    MN_CHECK_LAUNCH("mn_iaobf_g3_stats");
    mn_set_last_kernel("k_g3_stats_finish");
    hipLaunchKernelGGL(k_g3_stats_finish, dim3((unsigned)((g->O + 63) / 64)), dim3(64), 0, (hipStream_t)stream, (const double*)ws, p.m.NB, (int)g->O, stats);
    MN_CHECK_LAUNCH("mn_iaobf_g3_stats");
    return MN_OK;
}
// out = [relu](conv2d(Q_a(x), qw, bias_f)); mm (nullable): 2 * mn_iaobf_g3_mm_count(g) floats
extern "C" int mn_iaobf_g3_fwd(const mn_conv_geom* g, const float* x, const float* aqp, int a_bits, const float* qw, const float* wqp, const float* bias_f, int relu,
                               float* out, float* mm, mn_stream_t stream) {
    G3FParams p;
    if (!plan_g3(g, G3_FWD_BLOCKS, &p.m)) MN_FAIL(MN_ENOTSUP, "mn_iaobf_g3_fwd: geometry not covered");
    if (!x || !aqp || !qw || !wqp || !out || !g3_bits_ok(a_bits)) MN_FAIL(MN_EINVAL, "mn_iaobf_g3_fwd: bad arguments");
    p.x = x; p.w = qw; p.bias = bias_f; p.xqp = aqp; p.wqp = wqp; p.stats = nullptr; p.coef = nullptr; p.part = nullptr; p.out = out; p.mm = mm; p.relu = relu;
    const IaoRange r = iao_range(a_bits, 0, 1);
    p.qmin = r.qmin; p.qmax = r.qmax;
    const size_t lds = g3_fwd_lds(p.m, 2);
    raise_lds_limit((const void*)k_g3_fwd<G3_QUANT>, lds);
    mn_set_last_kernel("k_g3_fwd<QUANT>");
    hipLaunchKernelGGL(k_g3_fwd<G3_QUANT>, dim3((unsigned)(p.m.G * p.m.NB)), dim3(256), lds, (hipStream_t)stream, p);
    MN_CHECK_LAUNCH("mn_iaobf_g3_fwd");
    return MN_OK;
}
// dy = coef[0][o] + coef[1][o] * (conv2d(x, w, bias) - stats[0][o]): the gradient of the batch statistics w.r.t. the raw convolution's output
extern "C" int mn_iaobf_g3_dyraw(const mn_conv_geom* g, const float* x, const float* xgrid, int grid_bits, const float* w, const float* bias, const float* stats,
                                 const float* coef, float* dy, mn_stream_t stream) {
    G3FParams p;
    if (!plan_g3(g, G3_FWD_BLOCKS, &p.m)) MN_FAIL(MN_ENOTSUP, "mn_iaobf_g3_dyraw: geometry not covered");
    if (!x || !xgrid || !w || !stats || !coef || !dy || !g3_bits_ok(grid_bits)) MN_FAIL(MN_EINVAL, "mn_iaobf_g3_dyraw: bad arguments");
    p.x = x; p.w = w; p.bias = bias; p.xqp = xgrid; p.wqp = nullptr; p.stats = stats; p.coef = coef; p.part = nullptr; p.out = dy; p.mm = nullptr; p.relu = 0;
    const IaoRange r = iao_range(grid_bits, 0, 1);
    p.qmin = r.qmin; p.qmax = r.qmax;
    const size_t lds = g3_fwd_lds(p.m, 6);
    raise_lds_limit((const void*)k_g3_fwd<G3_DY>, lds);
    mn_set_last_kernel("k_g3_fwd<DY>");
    hipLaunchKernelGGL(k_g3_fwd<G3_DY>, dim3((unsigned)(p.m.G * p.m.NB)), dim3(256), lds, (hipStream_t)stream, p);
    MN_CHECK_LAUNCH("mn_iaobf_g3_dyraw");
    return MN_OK;
}
// dw (+)= xqp[0] * conv2d_backward_weight(a * [mask > 0], codes(x)); dbias (nullable) = sum of the masked a
extern "C" int mn_iaobf_g3_bwd_weight(const mn_conv_geom* g, const float* a, const float* mask, const float* x, const float* xqp, int x_bits, int accumulate, float* dw,
                                      float* dbias, void* ws, int64_t ws_bytes, mn_stream_t stream) {
    G3WParams p;
    if (!plan_g3(g, G3_BWD_BLOCKS, &p.m)) MN_FAIL(MN_ENOTSUP, "mn_iaobf_g3_bwd_weight: geometry not covered");
    if (!a || !x || !xqp || !dw || !ws || !g3_bits_ok(x_bits) || ws_bytes < mn_iaobf_g3_ws_bytes(g)) MN_FAIL(MN_EINVAL, "mn_iaobf_g3_bwd_weight: bad arguments");
    p.a = a; p.mask = mask; p.x = x; p.xqp = xqp;
    const IaoRange r = iao_range(x_bits, 0, 1);
    p.qmin = r.qmin; p.qmax = r.qmax;
    p.part = (float*)ws;
    p.dbpart = dbias ? p.part + (int64_t)p.m.G * p.m.NB * 4 * (32 * 144) : nullptr;
    const size_t lds = g3_wg_lds(p.m);
    raise_lds_limit((const void*)k_g3_wgrad, lds);
    mn_set_last_kernel("k_g3_wgrad");
    hipLaunchKernelGGL(k_g3_wgrad, dim3((unsigned)(p.m.G * p.m.NB)), dim3(256), lds, (hipStream_t)stream, p);
    MN_CHECK_LAUNCH("mn_iaobf_g3_bwd_weight");
    mn_set_last_kernel("k_g3_wsum");
    hipLaunchKernelGGL(k_g3_wsum, dim3((unsigned)((g->O * 144 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)p.part, (const float*)p.dbpart, xqp, p.m.NB,
                       (int)g->O, accumulate, dw, dbias);
    MN_CHECK_LAUNCH("mn_iaobf_g3_bwd_weight");
    return MN_OK;
}
// dx = clip-STE_a(conv2d_backward_data(gy * [mask > 0], qw)) + conv2d_backward_data(dy, w)  [* [x > 0] when relu_in]
extern "C" int mn_iaobf_g3_bwd_data(const mn_conv_geom* g, const float* gy, const float* mask, const float* dy, const float* x, const float* aqp, int a_bits,
                                    const float* qw, const float* wqp, const float* w, int relu_in, float* dx, mn_stream_t stream) {
    G3DParams p;
    if (!plan_g3(g, G3_BWD_BLOCKS, &p.m)) MN_FAIL(MN_ENOTSUP, "mn_iaobf_g3_bwd_data: geometry not covered");
    if (!gy || !dy || !x || !aqp || !qw || !wqp || !w || !dx || !g3_bits_ok(a_bits)) MN_FAIL(MN_EINVAL, "mn_iaobf_g3_bwd_data: bad arguments");
    p.gy = gy; p.mask = mask; p.dy = dy; p.x = x; p.aqp = aqp; p.qw = qw; p.wqp = wqp; p.w = w; p.dx = dx; p.relu_in = relu_in;
    const IaoRange r = iao_range(a_bits, 0, 1);
    p.qmin = r.qmin; p.qmax = r.qmax;
    const size_t lds = g3_dg_lds(p.m);
    raise_lds_limit((const void*)k_g3_dgrad, lds);
    mn_set_last_kernel("k_g3_dgrad");
    hipLaunchKernelGGL(k_g3_dgrad, dim3((unsigned)(p.m.G * p.m.NB)), dim3(256), lds, (hipStream_t)stream, p);
    MN_CHECK_LAUNCH("mn_iaobf_g3_bwd_data");
    return MN_OK;
}
