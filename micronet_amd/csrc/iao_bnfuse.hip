// The BN-fused IAO convolution block of the reference -- QuantBNFuseConv2d.forward in training / QAT mode, wqaq/iao/quantize.py:837-994 (raw conv 843-851,
// batch statistics 853-855, running statistics 856-879, fold 881-901, quantizers 944-945, quantised conv 947-955) -- for POINTWISE (1 x 1, stride 1) grouped
// layers, without ever running the "raw" statistics convolution or its backward.
//
// The raw convolution y = W x + b is linear, and the block needs from it ONLY the per-channel mean / unbiased variance of y (forward) and the gradient of
// those two statistics (backward).  With x_bar[c] = mean of input channel c over (N, H, W) and the centred second moment S[c][c'] = sum_p (x[c,p] - x_bar[c])
// (x[c',p] - x_bar[c']) of the group's input channels:
//     mean[o] = W[o,:] . x_bar + b[o]                     var[o] = W[o,:] S W[o,:]^T / (n - 1)
//     d y_raw[o,p] = dmean[o] / n + B[o] (y[o,p] - mean[o]),   B[o] = 2 dvar[o] / (n - 1)                               (autograd of mean / var, ref 853-855)
//     d W_raw[o,:] = sum_p d y_raw[o,p] x[:,p] = dmean[o] x_bar + B[o] W[o,:] S                     d b_raw[o] = dmean[o]
//     d x_raw[:,p] = W^T d y_raw[:,p] = M (x[:,p] - x_bar) + v,     M = W^T diag(B) W  (Cg x Cg per group),   v = W^T dmean / n
// so ONE pass over x (k_bf_gram: the Gram matrix of the group's channels on the matrix cores, fp32 partial tiles over <= 2048 pixels, combined in fp64)
// replaces the statistics convolution + its two-pass statistics; the raw backward-weight is a (Cg x Cg) product per output channel in fp64 (k_bf_prep_bwd); and the
// raw backward-data is a second pointwise contraction INSIDE the quantised conv's backward-data kernel (k_bf_dgrad: phase A = W_q^T d out with the activation
// quantizer's clip-STE, phase B = M (x - x_bar) on the tile of x the STE reads anyway, then the ReLU mask of the block in front).  x is read twice in the forward
// (Gram, quantised conv) and twice in the backward (backward-weight, backward-data) instead of 2 + 5 times, the raw output y_raw (as large as the block's output)
// is never formed.
//
// Real-valued operands on bf16 matrix cores: v = t0 + t1 + t2 (bf16 head of the running remainder, exact), and a product of two reals uses the six term products
// down to 2^-24 relative (t0 t0, t0 t1, t1 t0, t1 t1, t0 t2, t2 t0): the accuracy of an fp32 multiply at 6/16 of the fp32-MFMA cost.
//
// The small kernels: k_bf_prep_fwd = statistics from the Gram data (or given), running statistics, fold, per-channel weight observer + qparams + fake-quant
// (ref 856-901, 945 with 15-36 / 62-74 / 293-321 / 227-239) in ONE launch per layer; k_bf_prep_bwd = the weight quantizer's clip-STE, the fold's backward, dmean /
// dvar and the raw-path weight gradient in one launch; k_bf_M = M, v, x_bar for the backward-data kernel.
#include "qgemm_dev.h"

#include <stdlib.h>

#define BF_LDP 72          // u16 per staged row: 64 pixels + 8 pad (144-byte rows: the 16 rows of a b128 fragment read cover all 64 banks)

// ------------------------------------------------------------------------------------------------ Gram matrix of the input channels of a pointwise group
struct GramParams {
    const float* x;
    float* part;        // [G * CP * CP / 64][Z][64]: segment-major, so that the reduction streams
    float* sxpart;      // [Z][G][CP]
    int N, HW, C_total, Cg, G, Z, nchunks, CP;
    uint32_t NP;
    FastDiv fd_hw;
    ChanMap in_map;
    // PATCH mode (the first layer: k x k, stride 1, "same" padding, groups 1, Cin * KH * KW <= 128): the "channels" are the Cin * KH * KW elements of the patch
    // around each pixel, gathered from the (tiny, cache-resident) image -- the Gram matrix of the im2col matrix without materialising it
    int H, W, KH, KW, pad_h, pad_w;
    FastDiv fd_w;
};
// CW = CP / 32: a block of 4 waves (2 x 2) owns the CP x CP tile of one group and a strided set of 64-pixel slabs
template <int CW, int PATCH>
__global__ __launch_bounds__(256, 2) void k_bf_gram(const GramParams p) {
    HIP_DYNAMIC_SHARED(float, smem)
    constexpr int CP = 32 * CW, RC = CP / 16;
    uint16_t* gt = reinterpret_cast<uint16_t*>(smem);        // [3][CP][BF_LDP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kg = lane >> 4;
    const int r0 = tid >> 4, qd = tid & 15;
    const int z = blockIdx.x % p.Z, g = blockIdx.x / p.Z;
    // The Gram matrix is symmetric.  CW == 4 (the 128-channel groups of nin_gc: an 8 x 8 grid of 16 x 16 tiles): wave w owns tile rows w and 7 - w of the UPPER
    // triangle -- (8 - w) + (w + 1) = 9 tiles instead of 16 --, the reduction mirrors the strictly upper tiles.  Other widths keep the 2 x 2 wave grid (the pairing
    // does not balance on 4 x 4 / 6 x 6 grids); their lower tiles are computed and ignored.
    constexpr bool SYM = CW == 4;
    const int wm = wave >> 1, wc = wave & 1;
    const int nA = 8 - wave;                                  // SYM: tiles 0 .. nA - 1 lie in row `wave` (columns wave ..), the rest in row 7 - wave (columns 7 - wave ..)
    const int64_t HW = p.HW;

    f32x4 acc[CW][CW];          // SYM: acc[t / CW][t % CW], t < 9
#pragma unroll
    for (int mi = 0; mi < CW; ++mi)
#pragma unroll
        for (int ci = 0; ci < CW; ++ci) acc[mi][ci] = f32x4{0.f, 0.f, 0.f, 0.f};
    float sx[RC];
#pragma unroll
    for (int i = 0; i < RC; ++i) sx[i] = 0.f;

    float4 rx[RC];
    int pci[RC], pdy[RC], pdx[RC];          // PATCH: (input channel, row shift, column shift) of this thread's patch elements
    if (PATCH) {
#pragma unroll
        for (int i = 0; i < RC; ++i) {
            const int k = r0 + 16 * i, t = p.KH * p.KW;
            pci[i] = k / t;
            pdy[i] = (k - pci[i] * t) / p.KW - p.pad_h;
            pdx[i] = (k - pci[i] * t) % p.KW - p.pad_w;
        }
    }
    auto fetch = [&](int chunk) {
        const uint32_t P = (uint32_t)chunk * 64u + 4u * qd;
        const bool pv = P < p.NP;
        const uint32_t n = fd_div(P, p.fd_hw);
        const int pp = (int)(P - n * (uint32_t)p.HW);
        if (PATCH) {
            const int y = (int)fd_div((uint32_t)pp, p.fd_w), x0 = pp - y * p.W;
#pragma unroll
            for (int i = 0; i < RC; ++i) {
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                const int iy = y + pdy[i];
                if (pv && r0 + 16 * i < p.Cg && iy >= 0 && iy < p.H) {
                    const float* row = p.x + (((int64_t)n * p.C_total + pci[i]) * p.H + iy) * p.W;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int ix = x0 + e + pdx[i];
                        if (ix >= 0 && ix < p.W) v[e] = row[ix];
                    }
                }
                rx[i] = make_float4(v[0], v[1], v[2], v[3]);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < RC; ++i) {
            const int c = r0 + 16 * i;
            rx[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pv && c < p.Cg) rx[i] = *reinterpret_cast<const float4*>(p.x + ((int64_t)n * p.C_total + chan_phys(p.in_map, g * p.Cg + c)) * HW + pp);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < RC; ++i) {
            const float v[4] = {rx[i].x, rx[i].y, rx[i].z, rx[i].w};
            float t0[4], t1[4], t2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                t0[e] = mn_bf16_head(v[e]);
                const float r1 = v[e] - t0[e];
                t1[e] = mn_bf16_head(r1);
                t2[e] = r1 - t1[e];
            }
            sx[i] += (v[0] + v[1]) + (v[2] + v[3]);
            uint16_t* d = gt + (r0 + 16 * i) * BF_LDP + qd * 4;
            *reinterpret_cast<u32x2*>(d) = u32x2{mn_pack_bf16x2(t0[0], t0[1]), mn_pack_bf16x2(t0[2], t0[3])};
            *reinterpret_cast<u32x2*>(d + CP * BF_LDP) = u32x2{mn_pack_bf16x2(t1[0], t1[1]), mn_pack_bf16x2(t1[2], t1[3])};
            *reinterpret_cast<u32x2*>(d + 2 * CP * BF_LDP) = u32x2{mn_pack_bf16x2(t2[0], t2[1]), mn_pack_bf16x2(t2[2], t2[3])};
        }
    };
    auto contract = [&]() {
#pragma unroll
        for (int ksx = 0; ksx < 2; ++ksx) {
            const int ko = ksx * 32 + kg * 8;
            if (SYM) {
                // A fragments of the wave's two tile rows; the B fragment of every tile is read where its (run-time) column lives
                u32x4 a0[3], a1[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    a0[t] = *reinterpret_cast<const u32x4*>(gt + (t * CP + wave * 16 + j) * BF_LDP + ko);
                    a1[t] = *reinterpret_cast<const u32x4*>(gt + (t * CP + (7 - wave) * 16 + j) * BF_LDP + ko);
                }
                // three tiles at a time: three independent accumulators between two MFMAs on the same one (a dependent MFMA stalls the issue), 9 B fragments live
#pragma unroll
                for (int tg = 0; tg < 3; ++tg) {
                    u32x4 bb[3][3], aa[3][3];
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        const int tl = tg * 3 + u;
                        const bool first = tl < nA;                                    // wave-uniform
                        const int col = first ? wave + tl : 7 - wave + (tl - nA);
#pragma unroll
                        for (int t = 0; t < 3; ++t) {
                            bb[u][t] = *reinterpret_cast<const u32x4*>(gt + (t * CP + col * 16 + j) * BF_LDP + ko);
                            aa[u][t] = first ? a0[t] : a1[t];
                        }
                    }
#define BF_GRAM_SYM(TA, TB) _Pragma("unroll") for (int u = 0; u < 3; ++u) { f32x4& c = acc[(tg * 3 + u) / CW][(tg * 3 + u) % CW]; c = mn_mfma_bf16(aa[u][TA], bb[u][TB], c); }
                    BF_GRAM_SYM(0, 2) BF_GRAM_SYM(2, 0) BF_GRAM_SYM(1, 1) BF_GRAM_SYM(0, 1) BF_GRAM_SYM(1, 0) BF_GRAM_SYM(0, 0)
#undef BF_GRAM_SYM
                }
                continue;
            }
            u32x4 a[3][CW], b[3][CW];
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < CW; ++i) {
                    a[t][i] = *reinterpret_cast<const u32x4*>(gt + (t * CP + (wm * CW + i) * 16 + j) * BF_LDP + ko);
                    b[t][i] = *reinterpret_cast<const u32x4*>(gt + (t * CP + (wc * CW + i) * 16 + j) * BF_LDP + ko);
                }
            // the six term products of two exact three-term splits, smallest first; CW * CW independent accumulators between two MFMAs on the same one
#define BF_GRAM_PAIR(TA, TB)                                                                             \
    _Pragma("unroll") for (int mi = 0; mi < CW; ++mi)                                                    \
        _Pragma("unroll") for (int ci = 0; ci < CW; ++ci) acc[mi][ci] = mn_mfma_bf16(a[TA][mi], b[TB][ci], acc[mi][ci]);
            BF_GRAM_PAIR(0, 2) BF_GRAM_PAIR(2, 0) BF_GRAM_PAIR(1, 1) BF_GRAM_PAIR(0, 1) BF_GRAM_PAIR(1, 0) BF_GRAM_PAIR(0, 0)
#undef BF_GRAM_PAIR
        }
    };

    int chunk = z;
    if (chunk < p.nchunks) fetch(chunk);
    for (; chunk < p.nchunks; chunk += p.Z) {
        __syncthreads();          // previous slab fully consumed
        commit();
        __syncthreads();
        if (chunk + p.Z < p.nchunks) fetch(chunk + p.Z);     // in flight during the MFMA phase
        contract();
    }
    // partial tile: lane (j, kg) holds rows m = 4 kg + r, column c = j of tile (mi, ci)
#pragma unroll
    for (int mi = 0; mi < CW; ++mi)
#pragma unroll
        for (int ci = 0; ci < CW; ++ci) {
            const int tl = mi * CW + ci;
            if (SYM && tl >= 9) continue;
            const int trow = SYM ? (tl < nA ? wave : 7 - wave) : wm * CW + mi;
            const int tcol = SYM ? (tl < nA ? wave + tl : 7 - wave + (tl - nA)) : wc * CW + ci;
            const int mrow = trow * 16 + kg * 4, ccol = tcol * 16 + j;
            // partial layout [segment of 64 tile floats][Z][64]: the reduction reads Z x 256 contiguous bytes per segment (a [Z][tile] layout made it gather
            // 256-byte pieces 128 KB apart: 0.8 TB/s)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t e = ((int64_t)g * CP + mrow + r) * CP + ccol;
                p.part[((e >> 6) * p.Z + z) * 64 + (e & 63)] = acc[mi][ci][r];
            }
        }
#pragma unroll
    for (int i = 0; i < RC; ++i) {
        float v = sx[i];
        v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
        if (qd == 0) p.sxpart[((int64_t)z * p.G + g) * CP + r0 + 16 * i] = v;
    }
}
// fixed-order fp64 combination of the Z partial tiles: a block owns 64 consecutive floats of the padded [G][CP][CP] tile; its 16 groups of 16 lanes each sum a z
// subset with 16-byte loads, the 16 sums are combined through LDS in a fixed order.  Blocks >= nblk_w: the channel sums.
__global__ __launch_bounds__(256) void k_bf_gram_reduce(const float* __restrict__ part, const float* __restrict__ sxpart, double* __restrict__ gram,
                                                        double* __restrict__ sx, int Z, int G, int Cg, int CP, int nblk_w) {
    __shared__ double sm[16][16][4];
    const int q = threadIdx.x >> 4, l = threadIdx.x & 15;
    if ((int)blockIdx.x < nblk_w) {
        // the block's 64 floats: row m of one group, columns cb .. cb + 63 = tile columns cb / 16 .. + 3; only tiles with tile row <= tile column were computed
        const int64_t e0 = (int64_t)blockIdx.x * 64;
        const int cb0 = (int)(e0 % CP), mrow = (int)((e0 / CP) % CP);
        if ((mrow >> 4) > ((cb0 + 63) >> 4)) return;          // (block-uniform) every tile of this segment lies below the diagonal: its mirror image is written instead
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll 4
        for (int z = q; z < Z; z += 16) {
            const float4 v = *reinterpret_cast<const float4*>(part + ((int64_t)blockIdx.x * Z + z) * 64 + 4 * l);
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
            const int c = (int)(ei % CP);
            const int64_t o = ei / CP;
            const int m = (int)(o % CP), gg = (int)(o / CP);
            if (m < Cg && c < Cg && (m >> 4) <= (c >> 4)) {
                gram[((int64_t)gg * Cg + m) * Cg + c] = t;
                if ((m >> 4) < (c >> 4)) gram[((int64_t)gg * Cg + c) * Cg + m] = t;          // the mirror image of a strictly upper tile (diagonal tiles hold both halves)
            }
        }
    } else {
        // channel sums: 16 lanes per channel, each a strided z subset with four loads in flight, combined by a fixed-order butterfly (one thread per channel walking
        // all Z partials made this tail -- Z serialised L2 round trips -- the longest part of the launch: 40 us)
        const int64_t total = (int64_t)G * Cg, nslots = ((total + 15) / 16) * 16;
        const int per = 256 / 16;
        for (int64_t i0 = (int64_t)(blockIdx.x - nblk_w) * per; i0 < nslots; i0 += (int64_t)(gridDim.x - nblk_w) * per) {
            const int64_t i = i0 + q;
            double t = 0.0;
            if (i < total) {
                const int gg = (int)(i / Cg), c = (int)(i % Cg);
#pragma unroll 4
                for (int z = l; z < Z; z += 16) t += (double)sxpart[((int64_t)z * G + gg) * CP + c];
            }
            t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 1, 64);
            if (l == 0 && i < total) sx[i] = t;
        }
    }
}

struct GramPlan { GramParams p; int CW, patch; size_t lds; int grid; int64_t off_sx, ws_bytes; };
static int plan_gram(const mn_conv_geom* g, GramPlan* pl) {
    if (g->stride_h != 1 || g->stride_w != 1 || g->dil_h != 1 || g->dil_w != 1 || g->groups < 1) return 0;
    const int pointwise = g->KH == 1 && g->KW == 1 && g->pad_h == 0 && g->pad_w == 0;
    // the first layer of a net: k x k with "same" padding on a few input channels (nin_gc: 5 x 5 on RGB = 75 patch elements)
    const int patch = !pointwise && g->groups == 1 && (g->KH & 1) && (g->KW & 1) && g->pad_h == g->KH / 2 && g->pad_w == g->KW / 2 && g->C * g->KH * g->KW <= 128 &&
                      g->in_shuffle <= 1 && g->W % 4 == 0;
    if (!pointwise && !patch) return 0;
    const int64_t HW = (int64_t)g->H * g->W, NP = (int64_t)g->N * HW;
    if (HW % 4 || NP * HW >= ((int64_t)1 << 32) || NP + 256 >= ((int64_t)1 << 31) || NP < 2) return 0;
    if (g->C % g->groups || g->O % g->groups) return 0;
    const int Cg = patch ? g->C * g->KH * g->KW : g->C / g->groups;
    if (Cg < 1 || Cg > 128) return 0;
    if (g->in_shuffle > 1 && g->C % g->in_shuffle) return 0;
    GramParams& p = pl->p;
    pl->patch = patch;
    pl->CW = Cg > 96 ? 4 : (Cg > 64 ? 3 : 2);
    p.CP = 32 * pl->CW;
    p.H = g->H; p.W = g->W; p.KH = g->KH; p.KW = g->KW; p.pad_h = g->pad_h; p.pad_w = g->pad_w; p.fd_w = make_fastdiv((uint32_t)g->W);
    p.N = g->N; p.HW = (int)HW; p.C_total = g->C; p.Cg = Cg; p.G = patch ? 1 : g->groups; p.NP = (uint32_t)NP;
    p.nchunks = (int)((NP + 63) / 64);
    p.in_map = make_chanmap(g->in_shuffle, g->C);
    p.fd_hw = make_fastdiv((uint32_t)HW);
    // every fp32 accumulator sums <= 32 slabs (2048 pixels); about two blocks per CU; a partial tile costs CP * CP * 4 bytes written + read back
    int Z = 512 / p.G;
    if (Z > (p.nchunks + 7) / 8) Z = (p.nchunks + 7) / 8;
    if (Z < (p.nchunks + 31) / 32) Z = (p.nchunks + 31) / 32;
    if (Z < 1) Z = 1;
    p.Z = Z;
    pl->lds = (size_t)3 * p.CP * BF_LDP * 2;
    const int64_t nb = (int64_t)p.G * Z;
    if (nb > 0x7fffffff) return 0;
    pl->grid = (int)nb;
    const int64_t part_bytes = (int64_t)Z * p.G * p.CP * p.CP * 4;
    pl->off_sx = (part_bytes + 255) / 256 * 256;
    pl->ws_bytes = pl->off_sx + (int64_t)Z * p.G * p.CP * 4;
    return 1;
}
extern "C" int mn_iaobf_gram_supported(const mn_conv_geom* g) { GramPlan pl; return g && plan_gram(g, &pl); }
extern "C" int64_t mn_iaobf_gram_ws_bytes(const mn_conv_geom* g) { GramPlan pl; return (g && plan_gram(g, &pl)) ? pl.ws_bytes : 0; }
extern "C" int mn_iaobf_gram(const mn_conv_geom* g, const float* x, double* gram, double* sx, void* ws, int64_t ws_bytes, mn_stream_t stream) {
    GramPlan pl;
    if (!g || !x || !gram || !sx) MN_FAIL(MN_EINVAL, "mn_iaobf_gram: null argument");
    if (!plan_gram(g, &pl) || !aligned16(x)) MN_FAIL(MN_ENOTSUP, "mn_iaobf_gram: pointwise layers with <= 128 channels per group, or a first layer with <= 128 patch elements");
    if (!ws || ws_bytes < pl.ws_bytes || !aligned16(ws) || (((uintptr_t)gram) & 7) || (((uintptr_t)sx) & 7)) MN_FAIL(MN_ENOSPC, "mn_iaobf_gram: workspace too small / misaligned");
    hipStream_t s = (hipStream_t)stream;
    GramParams& p = pl.p;
    p.x = x; p.part = (float*)ws; p.sxpart = (float*)((char*)ws + pl.off_sx);
    mn_set_last_kernel("k_bf_gram<%d, %d>", pl.CW, pl.patch);
    mn_prof_bytes(4.0 * (double)g->N * g->C * g->H * g->W + 2.0 * (double)pl.off_sx);
    mn_prof_begin(s);
#define BF_GRAM_LAUNCH(CWV, PV)                                                                   \
    do {                                                                                          \
        raise_lds_limit((const void*)k_bf_gram<CWV, PV>, pl.lds);                                 \
        hipLaunchKernelGGL((k_bf_gram<CWV, PV>), dim3(pl.grid), dim3(256), pl.lds, s, p);         \
    } while (0)
    if (pl.patch) { if (pl.CW == 4) BF_GRAM_LAUNCH(4, 1); else if (pl.CW == 3) BF_GRAM_LAUNCH(3, 1); else BF_GRAM_LAUNCH(2, 1); }
    else { if (pl.CW == 4) BF_GRAM_LAUNCH(4, 0); else if (pl.CW == 3) BF_GRAM_LAUNCH(3, 0); else BF_GRAM_LAUNCH(2, 0); }
#undef BF_GRAM_LAUNCH
    mn_prof_end(s);
    const int nblk_w = (int)((int64_t)p.G * p.CP * p.CP / 64);
    hipLaunchKernelGGL(k_bf_gram_reduce, dim3(nblk_w + 8), dim3(256), 0, s, (const float*)p.part, (const float*)p.sxpart, gram, sx, p.Z, p.G, p.Cg, p.CP, nblk_w);
    MN_CHECK_LAUNCH("mn_iaobf_gram");
    return MN_OK;
}

// ------------------------------------------------------------------------------------------------ statistics of the raw convolution's output from the Gram data
// A block owns BF_GS_CH output channels of one group: mean[o] = W[o,:] . x_bar + b[o],  Vc[o,:] = W[o,:] S (S = G - n x_bar x_bar^T, the centred second moment),
// var[o] = Vc[o,:] . W[o,:] / (n - 1), everything in fp64; Vc (rounded to fp32: it is the centred quantity, no cancellation left) is kept for the backward, where
// the raw convolution's weight gradient is dmean x_bar + B Vc.  G is read once per BF_GS_CH channels (a block per channel re-read all of it: 128 KB x O).
// global -> LDS copy of n4 16-byte units by `nthr` threads with eight loads in flight per thread (a plain `dst[e] = src[e]` loop waits for every load)
template <typename V>
__device__ __forceinline__ void bf_stage16(V* __restrict__ dst, const V* __restrict__ src, int n4, int tid, int nthr) {
    int e = tid;
    for (; e + 7 * nthr < n4; e += 8 * nthr) {
        V v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[e + u * nthr];
#pragma unroll
        for (int u = 0; u < 8; ++u) dst[e + u * nthr] = v[u];
    }
    for (; e < n4; e += nthr) dst[e] = src[e];
}
struct __attribute__((aligned(16))) bf_d2 { double a, b; };

#define BF_GS_CH 8
// One block per (group, 8 output channels).  256 threads = two halves of 128: thread i of a half owns column i, the halves split the rows of the Gram matrix (the
// contraction index) and meet in LDS.  The group's whole Gram matrix is staged in LDS first (bulk coalesced loads, all in flight) -- read from global inside the loop
// it cost one serialised L2 round trip per row (31 us for 128 rows).
__global__ __launch_bounds__(256) void k_bf_gram_stats(const float* __restrict__ w, const float* __restrict__ bias, const double* __restrict__ gram,
                                                       const double* __restrict__ sx, int O, int Mg, int Cg, double n, float* __restrict__ stats,
                                                       float* __restrict__ vc) {
    HIP_DYNAMIC_SHARED(double, gsm)          // [Cg * Cg] G, then comb [128][BF_GS_CH] doubles, red [BF_GS_CH][2][2] doubles, ws [BF_GS_CH][128] floats
    double* comb = gsm + Cg * Cg;
    double* red = comb + 128 * BF_GS_CH;
    float* ws = reinterpret_cast<float*>(red + BF_GS_CH * 4);
    const int nb = (Mg + BF_GS_CH - 1) / BF_GS_CH;
    const int g = blockIdx.x / nb, ob = (blockIdx.x % nb) * BF_GS_CH;
    const int tid = threadIdx.x, half = tid >> 7, i = tid & 127;
    const int o0 = g * Mg + ob;
    const double* __restrict__ G = gram + (int64_t)g * Cg * Cg;
    const double* __restrict__ sxg = sx + (int64_t)g * Cg;
    int nch = Mg - ob;
    nch = nch < BF_GS_CH ? nch : BF_GS_CH;
    if ((Cg & 1) == 0 && !(((uintptr_t)G) & 15)) bf_stage16(reinterpret_cast<bf_d2*>(gsm), reinterpret_cast<const bf_d2*>(G), Cg * Cg / 2, tid, 256);
    else for (int e = tid; e < Cg * Cg; e += 256) gsm[e] = G[e];
    for (int e = tid; e < BF_GS_CH * 128; e += 256) {
        const int k = e >> 7, c = e & 127;
        ws[e] = (k < nch && c < Cg) ? w[(int64_t)(o0 + k) * Cg + c] : 0.f;
    }
    __syncthreads();
    double acc[BF_GS_CH], m1p[BF_GS_CH];
#pragma unroll
    for (int k = 0; k < BF_GS_CH; ++k) { acc[k] = 0.0; m1p[k] = 0.0; }
    const double xbi = i < Cg ? sxg[i] / n : 0.0;
    if (i < Cg) {
        const int ch = (Cg + 1) >> 1, cb = half * ch, ce = (cb + ch < Cg) ? cb + ch : Cg;
#pragma unroll 4
        for (int c = cb; c < ce; ++c) {
            const double gv = gsm[c * Cg + i];
#pragma unroll
            for (int k = 0; k < BF_GS_CH; ++k) acc[k] += (double)ws[k * 128 + c] * gv;
        }
    }
    if (half == 1) {
#pragma unroll
        for (int k = 0; k < BF_GS_CH; ++k) comb[i * BF_GS_CH + k] = acc[k];
    }
    __syncthreads();
    if (half == 0) {
#pragma unroll
        for (int k = 0; k < BF_GS_CH; ++k) { acc[k] += comb[i * BF_GS_CH + k]; m1p[k] = i < Cg ? (double)ws[k * 128 + i] * xbi : 0.0; }
        // m1[k] = sum_i w[k][i] x_bar[i]: wave shuffles, the two waves of the half combined through LDS
#pragma unroll
        for (int k = 0; k < BF_GS_CH; ++k) {
            double v = m1p[k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if ((tid & 63) == 0) red[k * 4 + ((tid >> 6) & 1) * 2] = v;
        }
    }
    __syncthreads();
    double qp_[BF_GS_CH];
    if (half == 0) {
#pragma unroll
        for (int k = 0; k < BF_GS_CH; ++k) {
            const double m1 = red[k * 4] + red[k * 4 + 2];
            const double vcv = acc[k] - n * m1 * xbi;          // centred: (W S)[o, i]
            if (k < nch && i < Cg) vc[(int64_t)(o0 + k) * Cg + i] = (float)vcv;
            qp_[k] = i < Cg ? vcv * (double)ws[k * 128 + i] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < BF_GS_CH; ++k) {
            double v = qp_[k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if ((tid & 63) == 0) red[k * 4 + ((tid >> 6) & 1) * 2 + 1] = v;
        }
    }
    __syncthreads();
    if (tid < nch) {
        const int k = tid;
        const double m1 = red[k * 4] + red[k * 4 + 2], q = red[k * 4 + 1] + red[k * 4 + 3];
        stats[o0 + k] = (float)(m1 + (bias ? (double)bias[o0 + k] : 0.0));
        stats[O + o0 + k] = (float)((q > 0.0 ? q : 0.0) / (n - 1.0));          // (a near-constant channel with a large mean can cancel to a slightly negative sum: sqrt(var + eps) must stay real)
    }
}
extern "C" int mn_iaobf_gram_stats(const float* w, const float* bias, const double* gram, const double* sx, int64_t O, int64_t Cg, int64_t groups, double n, float* stats,
                                   float* vc, mn_stream_t stream) {
    if (!w || !gram || !sx || !stats || !vc || O <= 0 || Cg <= 0 || Cg > 128 || groups < 1 || O % groups || !(n > 1.0)) MN_FAIL(MN_EINVAL, "mn_iaobf_gram_stats: bad arguments");
    const int Mg = (int)(O / groups), nb = (Mg + BF_GS_CH - 1) / BF_GS_CH;
    const size_t lds = (size_t)Cg * Cg * 8 + (size_t)128 * BF_GS_CH * 8 + (size_t)BF_GS_CH * 4 * 8 + (size_t)BF_GS_CH * 128 * 4;
    mn_set_last_kernel("k_bf_gram_stats");
    raise_lds_limit((const void*)k_bf_gram_stats, lds);
    hipLaunchKernelGGL(k_bf_gram_stats, dim3((unsigned)(groups * nb)), dim3(256), lds, (hipStream_t)stream, w, bias, gram, sx, (int)O, Mg, (int)Cg, n, stats, vc);
    MN_CHECK_LAUNCH("mn_iaobf_gram_stats");
    return MN_OK;
}

// ------------------------------------------------------------------------------------------------ forward preparation of a layer: one launch, one block per out-channel
struct PrepFwd {
    const float* w; const float* bias; const float* gamma; const float* beta;
    const float* stats_in;                    // [2][O] = mean, unbiased var of the raw conv output: mn_bn_stats_fwd on y_raw, or mn_iaobf_gram_stats (pointwise, no y_raw)
    float* running_mean; float* running_var;
    float* wmin; float* wmax; float* wscale; float* wzp;       // the per-channel ('C') or per-layer weight quantizer's buffers
    float* stats; float* kfold; float* bias_f; float* qw; float* qp;
    int O, K, per_channel, first_bn, first_w, obs_kind, q_type;
    float eps, momentum, quant_range, qmin, qmax;
    double momentum_w;
};
__global__ __launch_bounds__(256) void k_bf_prep_fwd(const PrepFwd p) {
    __shared__ float scf[16];
    __shared__ float sh[4];
    const int o = blockIdx.x, tid = threadIdx.x;
    const float* __restrict__ wr = p.w + (int64_t)o * p.K;
    const float mean = p.stats_in[o], var = p.stats_in[p.O + o];
    // running statistics (ref 856-879): the first training forward of a net that is not pretrained copies the batch statistics
    if (tid == 0) {
        p.stats[o] = mean; p.stats[p.O + o] = var;
        if (p.first_bn) { p.running_mean[o] = mean; p.running_var[o] = var; }
        else {
            const float a = (float)(1.0 - (double)p.momentum), b = p.momentum;      // python doubles (1 - m), m become fp32 scalars
            p.running_mean[o] = a * p.running_mean[o] + b * mean;
            p.running_var[o] = a * p.running_var[o] + b * var;
        }
    }
    // fold (ref 881-901), every step rounded as the reference's separate fp32 ops
    const float kf = p.gamma[o] / sqrtf(var + p.eps);
    float lo = INFINITY, hi = -INFINITY;
    for (int i = tid; i < p.K; i += 256) { const float v = wr[i] * kf; lo = OpMinF()(lo, v); hi = OpMaxF()(hi, v); }
    if (p.per_channel) {
        lo = block_reduce(lo, OpMinF(), INFINITY, scf);
        hi = block_reduce(hi, OpMaxF(), -INFINITY, scf);
        if (tid == 0) {
            observer_update(p.obs_kind, p.first_w, p.momentum_w, lo, hi, p.wmin + o, p.wmax + o);
            float* qp = p.qp + 4 * o;
            iao_qparams_row(p.wmin[o], p.wmax[o], p.q_type, p.quant_range, 1, p.wscale + o, p.wzp + o, qp);
            sh[0] = qp[0]; sh[1] = qp[1];
            p.kfold[o] = kf;
            p.bias_f[o] = p.bias ? p.beta[o] + (p.bias[o] - mean) * kf : p.beta[o] - mean * kf;
        }
        __syncthreads();
        const float s_ = sh[0], zp = sh[1];
        float* __restrict__ dst = p.qw + (int64_t)o * p.K;
        for (int i = tid; i < p.K; i += 256) dst[i] = iao_fq(wr[i] * kf, s_, zp, p.qmin, p.qmax);
    } else if (tid == 0) {
        p.kfold[o] = kf;
        p.bias_f[o] = p.bias ? p.beta[o] + (p.bias[o] - mean) * kf : p.beta[o] - mean * kf;
    }
}
extern "C" int mn_iaobf_prep_fwd(const float* w, const float* bias, const float* gamma, const float* beta, int64_t O, int64_t K, const float* stats_in, float eps,
                                 float momentum, int first_bn, float* running_mean, float* running_var,
                                 int w_bits, int w_qtype, int w_obs_kind, int first_w, double momentum_w, float* wmin, float* wmax, float* wscale, float* wzp,
                                 float* stats, float* kfold, float* bias_f, float* qw, float* qp, mn_stream_t stream) {
    if (!w || !gamma || !beta || !stats_in || !running_mean || !running_var || !wmin || !wmax || !wscale || !wzp || !stats || !kfold || !bias_f || !qw || !qp || O <= 0 ||
        K <= 0 || K > (1 << 20) || w_bits < 2 || w_bits > 24 || (w_qtype != 0 && w_qtype != 1) || (w_obs_kind != 0 && w_obs_kind != 1))
        MN_FAIL(MN_EINVAL, "mn_iaobf_prep_fwd: bad arguments");
    PrepFwd p;
    p.w = w; p.bias = bias; p.gamma = gamma; p.beta = beta; p.stats_in = stats_in; p.running_mean = running_mean; p.running_var = running_var;
    p.wmin = wmin; p.wmax = wmax; p.wscale = wscale; p.wzp = wzp; p.stats = stats; p.kfold = kfold; p.bias_f = bias_f; p.qw = qw; p.qp = qp;
    p.O = (int)O; p.K = (int)K; p.per_channel = 1; p.first_bn = first_bn; p.first_w = first_w; p.obs_kind = w_obs_kind;
    p.q_type = w_qtype; p.eps = eps; p.momentum = momentum; p.momentum_w = momentum_w;
    const IaoRange r = iao_range(w_bits, w_qtype, 0);
    p.qmin = r.qmin; p.qmax = r.qmax;
    p.quant_range = (w_qtype == 0) ? (float)((double)(r.qmax - r.qmin) / 2.0) : (float)(r.qmax - r.qmin);
    mn_set_last_kernel("k_bf_prep_fwd");
    hipLaunchKernelGGL(k_bf_prep_fwd, dim3((unsigned)O), dim3(256), 0, (hipStream_t)stream, p);
    MN_CHECK_LAUNCH("mn_iaobf_prep_fwd");
    return MN_OK;
}

// ------------------------------------------------------------------------------------------------ backward preparation: one launch, one block per out-channel
struct PrepBwd {
    const float* dwq; const float* dbf; const float* w; const float* bias; const float* gamma; const float* stats; const float* qp;
    const float* vc; const double* sx;     // gram path: Vc = W S of mn_iaobf_gram_stats [O][Cg], channel sums
    float* dw; float* dbias; float* dgamma; float* dbeta; float* coef;       // coef [4][O] = {dmean / n, B = 2 dvar / (n - 1), dmean, dvar}
    int O, K, Mg, Cg;
    float eps, qmin, qmax;
    double n;
};
__global__ __launch_bounds__(256) void k_bf_prep_bwd(const PrepBwd p) {
    __shared__ double scd[16];
    __shared__ double shd[4];
    const int o = blockIdx.x, tid = threadIdx.x;
    const float mean = p.stats[o], var = p.stats[p.O + o];
    const float sd = sqrtf(var + p.eps), rw = 1.0f / sd, kf = p.gamma[o] / sd;
    const float* __restrict__ wr = p.w + (int64_t)o * p.K;
    const float* __restrict__ gr = p.dwq + (int64_t)o * p.K;
    const float s_ = p.qp[4 * o], zp = p.qp[4 * o + 1], lo = p.qp[4 * o + 2], hi = p.qp[4 * o + 3];
    // the weight quantizer's clip-STE (ref 163-168 + the clamp of 232) on w_f = w * kf, then w_f = w * kf backwards
    double S = 0.0;
    for (int i = tid; i < p.K; i += 256) {
        const float gq = iao_fq_grad(gr[i], wr[i] * kf, s_, zp, lo, hi, p.qmin, p.qmax);
        if (!p.vc) p.dw[(int64_t)o * p.K + i] = gq * kf;
        S += (double)gq * (double)wr[i];
    }
    S = block_reduce(S, OpAddD(), 0.0, scd);
    if (tid == 0) {
        const float g = p.dbf[o];
        const double D = (double)g * (p.bias ? (double)p.bias[o] - (double)mean : -(double)mean);
        const double v = (double)var + (double)p.eps;
        const float dvar = (float)(S * (double)p.gamma[o] * -0.5 / (v * sqrt(v))) + (float)(D * (double)p.gamma[o] * -0.5 / (v * sqrt(v)));
        const float dmean = -(g * kf);
        if (p.dgamma) p.dgamma[o] = (float)(S * (double)rw + D * (double)rw);
        if (p.dbeta) p.dbeta[o] = g;
        if (p.dbias) p.dbias[o] = g * kf + dmean;        // d b through bias_f plus through the batch mean of the raw conv (sum_p d y_raw = dmean): cancels
        const float nf = (float)p.n;
        p.coef[o] = dmean / nf;
        p.coef[p.O + o] = dvar * 2.f / (nf - 1.f);
        p.coef[2 * p.O + o] = dmean;
        p.coef[3 * p.O + o] = dvar;
        shd[0] = (double)dmean; shd[1] = (double)(dvar * 2.f / (nf - 1.f));
    }
    if (!p.vc) return;
    __syncthreads();
    // raw-path weight gradient from the Gram data: dmean x_bar + B (W[o,:] S), the second factor kept from the forward (mn_iaobf_gram_stats)
    const int g = o / p.Mg;
    const double* __restrict__ sxg = p.sx + (int64_t)g * p.Cg;
    const double dmean = shd[0], B = shd[1];
    for (int i = tid; i < p.Cg; i += 256) {
        const double xb = sxg[i] / p.n;
        const float gq = iao_fq_grad(gr[i], wr[i] * kf, s_, zp, lo, hi, p.qmin, p.qmax);
        p.dw[(int64_t)o * p.K + i] = (float)((double)(gq * kf) + dmean * xb + B * (double)p.vc[(int64_t)o * p.Cg + i]);
    }
}
extern "C" int mn_iaobf_prep_bwd(const float* dwq, const float* dbf, const float* w, const float* bias, const float* gamma, const float* stats, const float* qp,
                                 int64_t O, int64_t K, int64_t groups, const float* vc, const double* sx, double n, float eps, int w_bits, int w_qtype, float* dw,
                                 float* dbias, float* dgamma, float* dbeta, float* coef, mn_stream_t stream) {
    if (!dwq || !dbf || !w || !gamma || !stats || !qp || !dw || !coef || O <= 0 || K <= 0 || K > (1 << 20) || groups < 1 || O % groups || w_bits < 2 || w_bits > 24 ||
        (w_qtype != 0 && w_qtype != 1) || !(n > 1.0) || (vc && !sx))
        MN_FAIL(MN_EINVAL, "mn_iaobf_prep_bwd: bad arguments");
    PrepBwd p;
    p.dwq = dwq; p.dbf = dbf; p.w = w; p.bias = bias; p.gamma = gamma; p.stats = stats; p.qp = qp; p.vc = vc; p.sx = sx;
    p.dw = dw; p.dbias = dbias; p.dgamma = dgamma; p.dbeta = dbeta; p.coef = coef;
    p.O = (int)O; p.K = (int)K; p.Mg = (int)(O / groups); p.Cg = (int)K; p.eps = eps; p.n = n;
    const IaoRange r = iao_range(w_bits, w_qtype, 0);
    p.qmin = r.qmin; p.qmax = r.qmax;
    mn_set_last_kernel("k_bf_prep_bwd");
    hipLaunchKernelGGL(k_bf_prep_bwd, dim3((unsigned)O), dim3(256), 0, (hipStream_t)stream, p);
    MN_CHECK_LAUNCH("mn_iaobf_prep_bwd");
    return MN_OK;
}

// ------------------------------------------------------------------------------------------------ pointwise backward-data, quantised path + raw path in one kernel
struct BfDgParams {
    const float* gy;          // d out [N][G*Mg][HW] (already masked by the block's own ReLU)
    const float* x;           // the block's input [N][G*Cg][HW] (physical layout; the channel shuffle is in `map`)
    float* dx;
    const uint16_t* wc;       // transposed quantised-weight codes [G][Mpad][KpA]   (row = input channel, k = output channel)
    const float* kscale;      // [G][KpA] weight scale of output channel k
    const uint16_t* mt;       // [3][G][Mpad][KpB] the three bf16 terms of M (row = input channel c, k = input channel c')
    const float* xbar;        // [G*Cg] logical
    const float* vadd;        // [G*Cg] logical
    const float* qp;          // activation quantizer {scale, zp, lo, hi}
    float qmin, qmax;
    int relu_mask;            // the input is the output of a ReLU: dx *= [x > 0]
    int N, HW, C_total, O_total, Cg, Mg, G;
    int KpA, KSA, KpB, KSB, Mpad, num_mblk, nchunks, CB;
    uint32_t NP;
    FastDiv fd_hw;
    ChanMap map;
};
// The clip-STE of the activation quantizer as an interval [XL, XH] of x (out[0], out[1]): both of its conditions -- qmin <= rha(x / s - zp) <= qmax (the clamp of 232)
// and lo <= x / s - zp <= hi (Round.backward 163-168) -- are monotone in x (every fp32 step of the chain is), so the pass set is ONE interval of floats.  Its two ends
// are found by bisection over the ordered float keys with the exact expressions (2 x 33 evaluations, one lane per block); the epilogue then needs two compares per
// element instead of two IEEE divisions and a floor.  (NaN x: every compare fails, the gradient is dropped, as in the reference.)
__device__ __forceinline__ void bf_ste_interval(const float* __restrict__ qp, float qmin, float qmax, float* out) {
    const float sc = qp[0], zp = qp[1], slo = qp[2], shi = qp[3];
    auto key2f = [](uint32_t k) { return mn_u2f((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k); };          // ordered key -> float
    auto lower_ok = [&](float x) { const float v = x / sc - zp; const float r = mn_rha(v); return r >= qmin && !(v < slo); };
    auto upper_ok = [&](float x) { const float v = x / sc - zp; const float r = mn_rha(v); return r <= qmax && !(v > shi); };
    const uint32_t kmin = 0x007fffffu, kmax = 0xff800000u;          // keys of -inf and +inf
    uint32_t lo_k = kmin, hi_k = kmax;
    if (lower_ok(key2f(kmin))) hi_k = kmin;
    else if (!lower_ok(key2f(kmax))) lo_k = hi_k = kmax;          // nothing passes
    else while (hi_k - lo_k > 1u) { const uint32_t mid = lo_k + ((hi_k - lo_k) >> 1); if (lower_ok(key2f(mid))) hi_k = mid; else lo_k = mid; }
    out[0] = lower_ok(key2f(hi_k)) ? key2f(hi_k) : INFINITY;
    lo_k = kmin; hi_k = kmax;
    if (upper_ok(key2f(kmax))) lo_k = kmax;
    else if (!upper_ok(key2f(kmin))) lo_k = hi_k = kmin;
    else while (hi_k - lo_k > 1u) { const uint32_t mid = lo_k + ((hi_k - lo_k) >> 1); if (upper_ok(key2f(mid))) lo_k = mid; else hi_k = mid; }
    out[1] = upper_ok(key2f(lo_k)) ? key2f(lo_k) : -INFINITY;
}

template <int NT>
__global__ __launch_bounds__(256, 2) void k_bf_dgrad(const BfDgParams p) {
    HIP_DYNAMIC_SHARED(float, smem)
    constexpr int MB = 16 * NT;
    const int LDA = p.KpA + 8, LDB = p.KpB + 8;
    uint16_t* wa = reinterpret_cast<uint16_t*>(smem);            // [MB][LDA]
    uint16_t* wb = wa + MB * LDA;                                 // [3][MB][LDB]
    float* ks = reinterpret_cast<float*>(wb + 3 * MB * LDB);      // [KpA]
    float* xb = ks + p.KpA;                                       // [KpB]
    float* va = xb + p.KpB;                                       // [MB]
    uint32_t* koffA = reinterpret_cast<uint32_t*>(va + MB);      // [KpA] element offset of d out channel k
    uint32_t* koffB = koffA + p.KpA;                              // [KpB] element offset of x channel k (physical)
    uint32_t* ooff = koffB + p.KpB;                               // [MB]  element offset of the (physical) dx channel
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kg = lane >> 4;
    const uint32_t HW = (uint32_t)p.HW;

    uint32_t b = blockIdx.x;
    const uint32_t xcd = b & 7u; b >>= 3;
    const int mblk = b % p.num_mblk; b /= p.num_mblk;
    const uint32_t idx = b * 8u + xcd;
    if (idx >= (uint32_t)(p.G * p.CB)) return;
    const int cb = idx % p.CB, g = idx / p.CB;
    {
        const uint16_t* wg = p.wc + ((int64_t)g * p.Mpad + mblk * MB) * p.KpA;
        const int k8 = p.KpA >> 3;
        for (int q = tid; q < MB * k8; q += 256) {
            const int row = q / k8, c8 = q - row * k8;
            *reinterpret_cast<u32x4*>(wa + row * LDA + c8 * 8) = *reinterpret_cast<const u32x4*>(wg + (int64_t)row * p.KpA + c8 * 8);
        }
        const int k8b = p.KpB >> 3;
        for (int t = 0; t < 3; ++t) {
            const uint16_t* mg = p.mt + (((int64_t)t * p.G + g) * p.Mpad + mblk * MB) * p.KpB;
            for (int q = tid; q < MB * k8b; q += 256) {
                const int row = q / k8b, c8 = q - row * k8b;
                *reinterpret_cast<u32x4*>(wb + (t * MB + row) * LDB + c8 * 8) = *reinterpret_cast<const u32x4*>(mg + (int64_t)row * p.KpB + c8 * 8);
            }
        }
        for (int k = tid; k < p.KpA; k += 256) {
            ks[k] = k < p.Mg ? p.kscale[g * p.KpA + k] : 0.f;
            koffA[k] = (uint32_t)(g * p.Mg + (k < p.Mg ? k : p.Mg - 1)) * HW;
        }
        for (int k = tid; k < p.KpB; k += 256) {
            const int kc = k < p.Cg ? k : p.Cg - 1;
            xb[k] = p.xbar[g * p.Cg + kc];
            koffB[k] = (uint32_t)chan_phys(p.map, g * p.Cg + kc) * HW;
        }
        for (int i = tid; i < MB; i += 256) {
            const int m = mblk * MB + i, mc = m < p.Cg ? m : p.Cg - 1;
            va[i] = p.vadd[g * p.Cg + mc];
            ooff[i] = (uint32_t)chan_phys(p.map, g * p.Cg + mc) * HW;
        }
    }
    float* xlh = reinterpret_cast<float*>(ooff + MB);
    if (tid == 0) bf_ste_interval(p.qp, p.qmin, p.qmax, xlh);
    __syncthreads();
    const float XL = xlh[0], XH = xlh[1], ssc = p.qp[0];

    const int chunk0 = cb * 4 + wave, cstride = p.CB * 4;
    const int my_chunks = chunk0 < p.nchunks ? (p.nchunks - chunk0 + cstride - 1) / cstride : 0;
    const int KST = p.KSA + p.KSB;
    const int total = my_chunks * KST;
    const uint32_t Pmax = p.NP - 4u;

    f32x4 acc[4][NT];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint32_t rbits[(4 * NT + 7) / 8] = {};          // [x > 0] of the chunk's own (channel, pixel) elements, written by the phase-A epilogue
    // step `it` = (chunk it / KST, step s): s < KSA streams d out (phase A), else x (phase B); unconditional loads from clamped offsets
    auto issue = [&](float4 (&raw)[8], int ci, int s) {
        uint32_t P = (uint32_t)(chunk0 + ci * cstride) * 64u + 4u * j;
        P = P < Pmax ? P : Pmax;
        const uint32_t n = fd_div(P, p.fd_hw);
        const uint32_t pp = P - n * HW;
        if (s < p.KSA) {
            const uint32_t go = n * (uint32_t)p.O_total * HW + pp;
#pragma unroll
            for (int e = 0; e < 8; ++e) raw[e] = *reinterpret_cast<const float4*>(p.gy + (go + koffA[s * 32 + kg * 8 + e]));
        } else {
            const uint32_t go = n * (uint32_t)p.C_total * HW + pp;
            const int sb = s - p.KSA;
#pragma unroll
            for (int e = 0; e < 8; ++e) raw[e] = *reinterpret_cast<const float4*>(p.x + (go + koffB[sb * 32 + kg * 8 + e]));
        }
    };
    int ci_i = 0, s_i = 0;           // the next step to be issued (no division by the runtime step count in the loop)
    auto issue_next = [&](float4 (&raw)[8]) {
        issue(raw, ci_i, s_i);
        if (++s_i == KST) { s_i = 0; ++ci_i; }
    };
    auto step = [&](float4 (&raw)[8], int it, int ci, int s) {
        const bool phaseA = s < p.KSA;
        const int sb = phaseA ? s : s - p.KSA;
        // the three exact bf16 terms of the streamed operand (phase A: d out times the weight scale; phase B: x - x_bar), one B fragment per pixel column q
        u32x4 bq[4][3];
        {
            float cf[8];
            bool kv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = sb * 32 + kg * 8 + e;
                cf[e] = phaseA ? ks[k] : xb[k];
                kv[e] = phaseA || k < p.Cg;          // padded contraction channels of phase B contribute nothing (M's padding is zero too)
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const float4 ra0 = raw[2 * d], ra1 = raw[2 * d + 1];
                    const float u0 = q == 0 ? ra0.x : (q == 1 ? ra0.y : (q == 2 ? ra0.z : ra0.w));
                    const float u1 = q == 0 ? ra1.x : (q == 1 ? ra1.y : (q == 2 ? ra1.z : ra1.w));
                    float x0 = phaseA ? u0 * cf[2 * d] : u0 - cf[2 * d], x1 = phaseA ? u1 * cf[2 * d + 1] : u1 - cf[2 * d + 1];
                    x0 = kv[2 * d] ? x0 : 0.f; x1 = kv[2 * d + 1] ? x1 : 0.f;
                    const float h0 = mn_bf16_head(x0), h1 = mn_bf16_head(x1);
                    const float r0 = x0 - h0, r1 = x1 - h1;
                    const float m0 = mn_bf16_head(r0), m1 = mn_bf16_head(r1);
                    bq[q][0][d] = mn_pack_bf16x2(h0, h1);
                    bq[q][1][d] = mn_pack_bf16x2(m0, m1);
                    bq[q][2][d] = mn_pack_bf16x2(r0 - m0, r1 - m1);
                }
        }
        if (it + 2 < total) issue_next(raw);              // the registers are free: two steps (16 KB per wave) are always in flight
        if (phaseA) {
            u32x4 av[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) av[t] = *reinterpret_cast<const u32x4*>(wa + (t * 16 + j) * LDA + sb * 32 + kg * 8);
#pragma unroll
            for (int tb = 2; tb >= 0; --tb)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[q][t] = mn_mfma_bf16(av[t], bq[q][tb], acc[q][t]);
        } else {
            // the six term products down to 2^-24, grouped by the term of M so that only one set of A fragments is live: a2 b0 | a1 b1, a1 b0 | a0 b2, a0 b1, a0 b0
#pragma unroll
            for (int ta = 2; ta >= 0; --ta) {
                MN_SCHED_FENCE();          // keep the three terms' A fragments from being hoisted together (-32 VGPRs)
                u32x4 av[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) av[t] = *reinterpret_cast<const u32x4*>(wb + (ta * MB + t * 16 + j) * LDB + sb * 32 + kg * 8);
#pragma unroll
                for (int tb = 2 - ta; tb >= 0; --tb)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[q][t] = mn_mfma_bf16(av[t], bq[q][tb], acc[q][t]);
            }
        }
        if (s == p.KSA - 1 || s == KST - 1) {
            // end of phase A: the clip-STE of the activation quantizer on W_q^T d out (Round.backward 163-168 + the clamp of 232), in place;
            // end of phase B: + v, the ReLU mask of the block in front, store
            const bool last = s == KST - 1;
            const uint32_t P = (uint32_t)(chunk0 + ci * cstride) * 64u + 4u * j;
            if (P < p.NP) {
                const uint32_t n = fd_div(P, p.fd_hw);
                const uint32_t ob = n * (uint32_t)p.C_total * HW + (P - n * HW);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    MN_SCHED_FENCE();          // one tile's four x rows in flight at a time (all 4 NT hoisted: +48 VGPRs, spills at NT = 4)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ml = t * 16 + kg * 4 + r;
                        if (mblk * MB + ml < p.Cg) {
                            const bool ste = !last || p.KSB == 0;          // the clip-STE epilogue (end of phase A) is the ONE place that reads x of the own channels:
                            const int bsh = ((t * 4 + r) & 7) * 4;         // the ReLU mask of the final epilogue is kept as 4 bits per (channel, pixel quad) meanwhile
                            uint32_t rb = (rbits[(t * 4 + r) >> 3] >> bsh) & 15u;          // (round 4 read x a second time there: 4 of the 20 bytes per element the kernel moved;
                            if (ste) {                                                      //  PMC 830 -> 722 MB per launch, 247 -> 200 us)
                                const float4 xv = *reinterpret_cast<const float4*>(p.x + (ob + ooff[ml]));
                                // ((g s) / s) as the chain rule of 227-239 writes it -- one IEEE division -- masked by the interval test
                                acc[0][t][r] = (xv.x >= XL && xv.x <= XH) ? (acc[0][t][r] * ssc) / ssc : 0.f;
                                acc[1][t][r] = (xv.y >= XL && xv.y <= XH) ? (acc[1][t][r] * ssc) / ssc : 0.f;
                                acc[2][t][r] = (xv.z >= XL && xv.z <= XH) ? (acc[2][t][r] * ssc) / ssc : 0.f;
                                acc[3][t][r] = (xv.w >= XL && xv.w <= XH) ? (acc[3][t][r] * ssc) / ssc : 0.f;
                                rb = (xv.x > 0.f ? 1u : 0u) | (xv.y > 0.f ? 2u : 0u) | (xv.z > 0.f ? 4u : 0u) | (xv.w > 0.f ? 8u : 0u);
                                rbits[(t * 4 + r) >> 3] = (rbits[(t * 4 + r) >> 3] & ~(15u << bsh)) | (rb << bsh);
                            }
                            if (last) {
                                const float c_ = va[ml];
                                float o0 = acc[0][t][r] + c_, o1 = acc[1][t][r] + c_, o2 = acc[2][t][r] + c_, o3 = acc[3][t][r] + c_;
                                if (p.relu_mask) {
                                    o0 = (rb & 1u) ? o0 : 0.f; o1 = (rb & 2u) ? o1 : 0.f; o2 = (rb & 4u) ? o2 : 0.f; o3 = (rb & 8u) ? o3 : 0.f;
                                }
                                *reinterpret_cast<float4*>(p.dx + (ob + ooff[ml])) = make_float4(o0, o1, o2, o3);
                            }
                        }
                    }
                }
            }
            if (last) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };
    float4 ra[8], rb[8];
    if (total > 0) issue_next(ra);
    if (total > 1) issue_next(rb);
    int ci_c = 0, s_c = 0;
    for (int it = 0; it < total; it += 2) {
        step(ra, it, ci_c, s_c);
        if (++s_c == KST) { s_c = 0; ++ci_c; }
        if (it + 1 < total) {
            step(rb, it + 1, ci_c, s_c);
            if (++s_c == KST) { s_c = 0; ++ci_c; }
        }
    }
}

// M = W^T diag(B) W, v = W^T (dmean / n), x_bar -- one block per (group, input channel) row; M leaves as three bf16 term planes in the [G][Mpad][KpB] layout of
// k_bf_dgrad's A operand.  Also the transposed quantised-weight codes [G][Mpad][KpA] + their scales (what k_qg_pack(transpose = 1) writes for one tap).
struct BfMParams {
    const float* w; const float* qw; const float* coef; const double* sx; const float* wscale;
    uint16_t* mt; uint16_t* wc; float* kscale; float* xbar; float* vadd;
    int O, Cg, Mg, G, Mpad, KpA, KpB, wscale_stride;
    double n;
};
#define BF_M_ROWS 8
// One block per (group, 8 rows of M).  256 threads = two halves of 128: thread i of a half owns column i, the halves split the sum over the output channels and meet
// in LDS.  The group's weights (Mg x Cg fp32) are staged in LDS first (the loop over o then reads LDS only: from global it was one serialised L2 round trip per o);
// the transposed weight codes come from ONE 16-byte load per thread (4 input channels of one output channel), v = W^T dmean / n from all 256 threads.  Round 4
// (first version: 16 rows per block, halves over the rows, strided 4-byte loads for the codes, 8 lanes for v): 54 us -> see DESIGN 4d.
__global__ __launch_bounds__(256) void k_bf_M(const BfMParams p) {
    HIP_DYNAMIC_SHARED(float, wsm)          // [Mg][Cg] W of the group, [Mg (+1)] B, then doubles [128][BF_M_ROWS]
    float* bsm = wsm + p.Mg * p.Cg;
    double* dsm = reinterpret_cast<double*>(wsm + ((p.Mg * p.Cg + p.Mg + 1) & ~1));
    const int nrb = p.Mpad / BF_M_ROWS;
    const int g = blockIdx.x / nrb, tid = threadIdx.x, half = tid >> 7, i = tid & 127;
    const int c0 = (blockIdx.x % nrb) * BF_M_ROWS;
    const float* __restrict__ wg = p.w + (int64_t)g * p.Mg * p.Cg;
    const float* __restrict__ A = p.coef + g * p.Mg;
    if (((p.Mg * p.Cg) & 3) == 0 && !(((uintptr_t)wg) & 15)) bf_stage16(reinterpret_cast<float4*>(wsm), reinterpret_cast<const float4*>(wg), p.Mg * p.Cg / 4, tid, 256);
    else for (int e = tid; e < p.Mg * p.Cg; e += 256) wsm[e] = wg[e];
    for (int o = tid; o < p.Mg; o += 256) bsm[o] = p.coef[p.O + g * p.Mg + o];
    // transposed codes of the quantised weights: code = rha(qw / scale[o]) (exact small integers); every entry of the block's rows is written (zero padding)
    {
        const bool vec = (p.Cg & 3) == 0 && !(((uintptr_t)p.qw) & 15);
        for (int e = tid; e < 2 * p.KpA; e += 256) {
            const int o = e >> 1, q4 = (e & 1) * 4;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (o < p.Mg) {
                const int oo = g * p.Mg + o;
                const float sc = p.wscale[(int64_t)oo * p.wscale_stride];
                const float* src = p.qw + (int64_t)oo * p.Cg + c0 + q4;
                if (vec && c0 + q4 + 3 < p.Cg) {
                    const float4 f = *reinterpret_cast<const float4*>(src);
                    v[0] = mn_rha(f.x / sc); v[1] = mn_rha(f.y / sc); v[2] = mn_rha(f.z / sc); v[3] = mn_rha(f.w / sc);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (c0 + q4 + k < p.Cg) v[k] = mn_rha(src[k] / sc);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) p.wc[((int64_t)g * p.Mpad + c0 + q4 + k) * p.KpA + o] = (uint16_t)(mn_f2u(v[k]) >> 16);
        }
        if (blockIdx.x % nrb == 0)
            for (int o = tid; o < p.KpA; o += 256) p.kscale[g * p.KpA + o] = o < p.Mg ? p.wscale[(int64_t)(g * p.Mg + o) * p.wscale_stride] : 0.f;
    }
    __syncthreads();
    // M[c][c2] = sum_o B[o] W[o][c] W[o][c2] (fp64): this half's share of the output channels
    const int oh = (p.Mg + 1) >> 1, ob = half * oh, oe = (ob + oh < p.Mg) ? ob + oh : p.Mg;
    int rc[BF_M_ROWS];
#pragma unroll
    for (int r = 0; r < BF_M_ROWS; ++r) rc[r] = c0 + r < p.Cg ? c0 + r : p.Cg - 1;          // (rows beyond Cg are padding: computed on a valid column, stored as zero)
    double m[BF_M_ROWS];
#pragma unroll
    for (int r = 0; r < BF_M_ROWS; ++r) m[r] = 0.0;
    if (i < p.Cg) {
#pragma unroll 4
        for (int o = ob; o < oe; ++o) {
            const float* wo = wsm + o * p.Cg;
            const double wv = (double)bsm[o] * (double)wo[i];
#pragma unroll
            for (int r = 0; r < BF_M_ROWS; ++r) m[r] += wv * (double)wo[rc[r]];
        }
    }
    if (half == 1) {
#pragma unroll
        for (int r = 0; r < BF_M_ROWS; ++r) dsm[i * BF_M_ROWS + r] = m[r];
    }
    __syncthreads();
    const int64_t plane = (int64_t)p.G * p.Mpad * p.KpB;
    if (half == 0 && i < p.KpB) {
#pragma unroll
        for (int r = 0; r < BF_M_ROWS; ++r) {
            const float v = (i < p.Cg && c0 + r < p.Cg) ? (float)(m[r] + dsm[i * BF_M_ROWS + r]) : 0.f;
            const float t0 = mn_bf16_head(v), r1 = v - t0, t1 = mn_bf16_head(r1), t2 = r1 - t1;
            const int64_t at = ((int64_t)g * p.Mpad + c0 + r) * p.KpB + i;
            p.mt[at] = (uint16_t)(mn_f2u(t0) >> 16);
            p.mt[plane + at] = (uint16_t)(mn_f2u(t1) >> 16);
            p.mt[2 * plane + at] = (uint16_t)(mn_f2u(t2) >> 16);
        }
    }
    __syncthreads();          // dsm is reused
    {   // v[c] = sum_o (dmean[o] / n) W[o][c] for the block's rows: 32 partial sums per row (thread = (row, o mod 32)), then one thread per row; x_bar
        const int r = tid & 7, seg = tid >> 3;
        double vs = 0.0;
        if (c0 + r < p.Cg)
            for (int o = seg; o < p.Mg; o += 32) vs += (double)A[o] * (double)wsm[o * p.Cg + c0 + r];
        dsm[seg * BF_M_ROWS + r] = vs;
    }
    __syncthreads();
    if (tid < BF_M_ROWS && c0 + tid < p.Cg) {
        double vsum = 0.0;
        for (int seg = 0; seg < 32; ++seg) vsum += dsm[seg * BF_M_ROWS + tid];
        p.vadd[g * p.Cg + c0 + tid] = (float)vsum;
        p.xbar[g * p.Cg + c0 + tid] = (float)(p.sx[g * p.Cg + c0 + tid] / p.n);
    }
}

struct BfDgPlan { BfDgParams p; int NT; size_t lds; int grid; int64_t off_wc, off_ks, off_xbar, off_v, ws_bytes; };
static int plan_bf_dgrad(const mn_conv_geom* g, BfDgPlan* pl) {
    GramPlan gp;
    if (!plan_gram(g, &gp) || gp.patch) return 0;          // (the first layer has no backward-data)
    const int Cg = g->C / g->groups, Mg = g->O / g->groups;
    const int64_t NP = (int64_t)g->N * g->H * g->W;
    if (4 * NP * (g->C > g->O ? g->C : g->O) >= ((int64_t)1 << 32) || NP < 4) return 0;        // 32-bit element offsets
    BfDgParams& p = pl->p;
    p.N = g->N; p.HW = g->H * g->W; p.G = g->groups; p.NP = (uint32_t)NP;
    p.C_total = g->C; p.O_total = g->O; p.Cg = Cg; p.Mg = Mg;
    p.map = make_chanmap(g->in_shuffle, g->C);
    p.KpA = qg_roundup(Mg, 32); p.KSA = p.KpA / 32;
    p.KpB = qg_roundup(Cg, 32); p.KSB = p.KpB / 32;
    if (p.KSA > 8) return 0;
    int NT = Cg > 32 ? 4 : (Cg > 16 ? 2 : 1);
    const int MB = 16 * NT;
    p.nchunks = (int)((NP + 63) / 64);
    p.fd_hw = make_fastdiv((uint32_t)p.HW);
    pl->NT = NT;
    p.num_mblk = (Cg + MB - 1) / MB; p.Mpad = p.num_mblk * MB;
    pl->lds = (size_t)MB * (p.KpA + 8) * 2 + (size_t)3 * MB * (p.KpB + 8) * 2 + (size_t)(2 * p.KpA + 2 * p.KpB + 2 * MB + 4) * 4;
    if (pl->lds > 72 * 1024) return 0;
    int CB = (p.nchunks + 3) / 4;
    const int cap = 512 / (p.G * p.num_mblk) > 0 ? 512 / (p.G * p.num_mblk) : 1;
    if (CB > cap) CB = cap;
    p.CB = CB;
    const int64_t nb = (int64_t)qg_roundup(p.G * CB, 8) * p.num_mblk;
    if (nb > 0x7fffffff) return 0;
    pl->grid = (int)nb;
    const int64_t mt_bytes = (int64_t)3 * p.G * p.Mpad * p.KpB * 2;
    pl->off_wc = (mt_bytes + 255) / 256 * 256;
    pl->off_ks = (pl->off_wc + (int64_t)p.G * p.Mpad * p.KpA * 2 + 255) / 256 * 256;
    pl->off_xbar = pl->off_ks + (int64_t)p.G * p.KpA * 4;
    pl->off_v = pl->off_xbar + (int64_t)g->C * 4;
    pl->ws_bytes = pl->off_v + (int64_t)g->C * 4;
    return 1;
}
extern "C" int mn_iaobf_bwd_data_supported(const mn_conv_geom* g) { BfDgPlan pl; return g && plan_bf_dgrad(g, &pl); }
extern "C" int64_t mn_iaobf_bwd_data_ws_bytes(const mn_conv_geom* g) { BfDgPlan pl; return (g && plan_bf_dgrad(g, &pl)) ? pl.ws_bytes : 0; }
extern "C" int mn_iaobf_bwd_data(const mn_conv_geom* g, const mn_actq* aq, const float* gy, const float* x, const float* w, const float* qw, const float* wqp,
                                 const float* coef, const double* sx, int relu_mask, float* dx, void* ws, int64_t ws_bytes, mn_stream_t stream) {
    BfDgPlan pl;
    if (!g || !aq || !gy || !x || !w || !qw || !wqp || !coef || !sx || !dx) MN_FAIL(MN_EINVAL, "mn_iaobf_bwd_data: null argument");
    if (aq->mode != MN_ACTQ_IAO || !aq->qp || aq->bits < 2 || aq->bits > 24) MN_FAIL(MN_EINVAL, "mn_iaobf_bwd_data: an IAO activation quantizer snapshot is required");
    if (!plan_bf_dgrad(g, &pl) || !aligned16(gy) || !aligned16(x) || !aligned16(dx)) MN_FAIL(MN_ENOTSUP, "mn_iaobf_bwd_data: geometry not covered");
    if (!ws || ws_bytes < pl.ws_bytes || !aligned16(ws)) MN_FAIL(MN_ENOSPC, "mn_iaobf_bwd_data: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    BfDgParams& p = pl.p;
    BfMParams m;
    m.w = w; m.qw = qw; m.coef = coef; m.sx = sx; m.wscale = wqp; m.wscale_stride = 4;
    m.mt = (uint16_t*)ws; m.wc = (uint16_t*)((char*)ws + pl.off_wc); m.kscale = (float*)((char*)ws + pl.off_ks);
    m.xbar = (float*)((char*)ws + pl.off_xbar); m.vadd = (float*)((char*)ws + pl.off_v);
    m.O = g->O; m.Cg = p.Cg; m.Mg = p.Mg; m.G = p.G; m.Mpad = p.Mpad; m.KpA = p.KpA; m.KpB = p.KpB; m.n = (double)g->N * g->H * g->W;
    {
        const size_t ldsm = (size_t)((p.Mg * p.Cg + p.Mg + 1) & ~1) * 4 + (size_t)128 * BF_M_ROWS * 8;
        raise_lds_limit((const void*)k_bf_M, ldsm);
        hipLaunchKernelGGL(k_bf_M, dim3((unsigned)(p.G * p.Mpad / BF_M_ROWS)), dim3(256), ldsm, s, m);
    }
    const IaoRange r = iao_range(aq->bits, aq->q_type, 1);
    p.gy = gy; p.x = x; p.dx = dx; p.wc = m.wc; p.kscale = m.kscale; p.mt = m.mt; p.xbar = m.xbar; p.vadd = m.vadd; p.qp = aq->qp; p.qmin = r.qmin; p.qmax = r.qmax;
    p.relu_mask = relu_mask;
    mn_set_last_kernel("k_bf_dgrad<%d>", pl.NT);
    { const double nx = (double)g->N * g->C * g->H * g->W, ny = (double)g->N * g->O * g->H * g->W; mn_prof_bytes(4.0 * ny + 12.0 * nx); }          // d out once, x twice (operand of the raw path + the clip-STE mask of the own channels), dx once
    mn_prof_begin(s);
    raise_lds_limit(pl.NT == 4 ? (const void*)k_bf_dgrad<4> : (pl.NT == 2 ? (const void*)k_bf_dgrad<2> : (const void*)k_bf_dgrad<1>), pl.lds);
    if (pl.NT == 4) hipLaunchKernelGGL(k_bf_dgrad<4>, dim3(pl.grid), dim3(256), pl.lds, s, p);
    else if (pl.NT == 2) hipLaunchKernelGGL(k_bf_dgrad<2>, dim3(pl.grid), dim3(256), pl.lds, s, p);
    else hipLaunchKernelGGL(k_bf_dgrad<1>, dim3(pl.grid), dim3(256), pl.lds, s, p);
    mn_prof_end(s);
    MN_CHECK_LAUNCH("mn_iaobf_bwd_data");
    return MN_OK;
}
