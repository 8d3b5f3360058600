// Pointwise (1 x 1, stride 1, groups 1) convolutions with a THIN output (<= 16 channels): the classifier layer of the reference's nets (models/nin_gc.py:83,
// 1024 -> 10 on 8 x 8 maps) inside the BN-fused IAO block (wqaq/iao/quantize.py:837-994).  Per pixel the layer is a 10 x 1024 matrix-vector product: 20 flop per
// input byte -- HBM-bound streaming work, not matrix-core work.  The general kernels (16-row MFMA tiles of which 10 rows are live, one block per 64-pixel tile
// over the whole K) spent 120 + 67 us on its two forward convolutions and ~150 us on the backward; these VALU kernels stream x once per pass with enough loads in
// flight: lanes = pixels (coalesced 256-byte rows), the waves of a block split the input channels, the <= 16 weights of a channel come as ONE scalar load from a
// transposed copy [C][16] (k_thin_pack), partial sums meet in LDS in a fixed order.
//   k_thin_fwd      y[o] = sum_c wt[c][o] * v(x[c]) + bias[o], v = identity (the raw convolution, 843-851) or the activation quantizer's fake-quantised value
//                   (947-955; Markstein division: the reference's x / s bit for bit), [ReLU], (min, max) partials of what is stored
//   k_thin_wgrad    dw[o][c] (+)= sum_p a[o][p] * v(x[c][p]): one block per 4 input channels over all pixels (no cross-block reduction), d bias from block 0
//   k_thin_dgrad    dx[c] = clip-STE(sum_o qw[o][c] g[o]) + sum_o w[o][c] dy[o]  [* [x > 0]]: the two paths of the BN-fused block's backward-data in one pass
#include "qgemm_dev.h"

#define TH_O 16

struct ThinGeom { int N, C, HW, O; uint32_t NP; FastDiv fd_hw; ChanMap in_map; };

// wt[c][o] = w[o][c] (o < O), 0 (O <= o < 16)
__global__ __launch_bounds__(256) void k_thin_pack(const float* __restrict__ w, int O, int Cc, float* __restrict__ wt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Cc * TH_O) return;
    const int c = i >> 4, o = i & 15;
    wt[i] = o < O ? w[(int64_t)o * Cc + c] : 0.f;
}

struct ThinFParams {
    const float* x; const float* wt; const float* bias; const float* aqp;
    float* out; float* mm;
    float qmin, qmax;
    int relu;
    ThinGeom m;
};
#define TH_FW 8          // waves per forward block: K slices
#define TH_FU 8          // channels per unrolled step
__global__ __launch_bounds__(64 * TH_FW) void k_thin_fwd(const ThinFParams p) {
    __shared__ float red[TH_FW][TH_O][64];
    __shared__ float scf[16];
    const ThinGeom& m = p.m;
    const int tid = threadIdx.x, lane = tid & 63, wave = mn_uniform(tid >> 6);
    const uint32_t P = blockIdx.x * 64u + lane;
    const bool pv = P < m.NP;
    const uint32_t n = fd_div(pv ? P : 0u, m.fd_hw);
    const int pp = (int)((pv ? P : 0u) - n * (uint32_t)m.HW);
    const float* __restrict__ xb = p.x + (int64_t)n * m.C * m.HW + pp;
    const bool quant = p.aqp != nullptr;
    float sc = 1.f, zp = 0.f, inv_sc = 1.f;
    if (quant) { sc = p.aqp[0]; zp = p.aqp[1]; inv_sc = 1.0f / sc; }
    const int cs = m.C / TH_FW, c0 = wave * cs;
    float acc[TH_O];
#pragma unroll
    for (int o = 0; o < TH_O; ++o) acc[o] = 0.f;
    float xn[TH_FU];          // the next step's TH_FU loads of 256 B are in flight while this step's are consumed
#pragma unroll
    for (int u = 0; u < TH_FU; ++u) xn[u] = pv ? xb[(int64_t)chan_phys(m.in_map, c0 + u) * m.HW] : 0.f;
    for (int cb = 0; cb < cs; cb += TH_FU) {
        float xv[TH_FU];
#pragma unroll
        for (int u = 0; u < TH_FU; ++u) xv[u] = xn[u];
        if (cb + TH_FU < cs) {
#pragma unroll
            for (int u = 0; u < TH_FU; ++u) xn[u] = pv ? xb[(int64_t)chan_phys(m.in_map, c0 + cb + TH_FU + u) * m.HW] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < TH_FU; ++u) {
            float v = xv[u];
            if (quant) v = iao_code_m(v, sc, inv_sc, zp, p.qmin, p.qmax) * sc;
            const float* __restrict__ wr = p.wt + (int64_t)(c0 + cb + u) * TH_O;
#pragma unroll
            for (int o = 0; o < TH_O; ++o) acc[o] = fmaf(wr[o], v, acc[o]);
        }
    }
#pragma unroll
    for (int o = 0; o < TH_O; ++o) red[wave][o][lane] = acc[o];
    __syncthreads();
    float lo = INFINITY, hi = -INFINITY;
    int mnan = 0;
    for (int i = tid; i < m.O * 64; i += 64 * TH_FW) {
        const int o = i >> 6, l = i & 63;
        float s = 0.f;
#pragma unroll
        for (int w_ = 0; w_ < TH_FW; ++w_) s += red[w_][o][l];
        s += p.bias ? p.bias[o] : 0.f;
        if (p.relu) s = qa_relu(s);
        const uint32_t Pl = blockIdx.x * 64u + l;
        if (Pl < m.NP) {
            const uint32_t nl = fd_div(Pl, m.fd_hw);
            p.out[((int64_t)nl * m.O + o) * m.HW + (Pl - nl * (uint32_t)m.HW)] = s;
            lo = fminf(lo, s); hi = fmaxf(hi, s); mnan |= (int)(s != s);
        }
    }
    if (p.mm) {
        if (mnan) lo = hi = NAN;
        lo = block_reduce(lo, OpMinF(), INFINITY, scf);
        hi = block_reduce(hi, OpMaxF(), -INFINITY, scf);
        if (tid == 0) { p.mm[blockIdx.x] = lo; p.mm[gridDim.x + blockIdx.x] = hi; }
    }
}

struct ThinWParams {
    const float* a; const float* x; const float* aqp;
    float* dw; float* dbias;
    float qmin, qmax;
    int accumulate;
    ThinGeom m;
};
#define TH_CL 4          // input channels per backward-weight block
#define TH_WW 8          // waves per backward-weight block (pixel chunks c, c + 8, ...)
#define TH_WU 4          // chunks per unrolled step
__global__ __launch_bounds__(64 * TH_WW) void k_thin_wgrad(const ThinWParams p) {
    __shared__ float red[TH_WW][TH_O * TH_CL + TH_O];
    const ThinGeom& m = p.m;
    const int tid = threadIdx.x, lane = tid & 63, wave = mn_uniform(tid >> 6);
    const int c0 = blockIdx.x * TH_CL;
    const bool quant = p.aqp != nullptr;
    float sc = 1.f, zp = 0.f, inv_sc = 1.f;
    if (quant) { sc = p.aqp[0]; zp = p.aqp[1]; inv_sc = 1.0f / sc; }
    int64_t coff[TH_CL];
#pragma unroll
    for (int k = 0; k < TH_CL; ++k) coff[k] = (int64_t)chan_phys(m.in_map, c0 + k) * m.HW;
    const bool want_db = p.dbias != nullptr && blockIdx.x == 0;
    float acc[TH_O][TH_CL], db[TH_O];
#pragma unroll
    for (int o = 0; o < TH_O; ++o) {
        db[o] = 0.f;
#pragma unroll
        for (int k = 0; k < TH_CL; ++k) acc[o][k] = 0.f;
    }
    const uint32_t nchunks = (m.NP + 63u) / 64u;
    for (uint32_t ch = wave; ch < nchunks; ch += TH_WU * TH_WW) {          // TH_WU chunks per step: their loads are in flight together
        float xv[TH_WU][TH_CL], av[TH_WU][TH_O];
#pragma unroll
        for (int h = 0; h < TH_WU; ++h) {
            const uint32_t P = (ch + h * TH_WW) * 64u + lane;
            const bool pv = P < m.NP;
            const uint32_t n = fd_div(pv ? P : 0u, m.fd_hw);
            const int pp = (int)((pv ? P : 0u) - n * (uint32_t)m.HW);
#pragma unroll
            for (int k = 0; k < TH_CL; ++k) xv[h][k] = pv ? p.x[(int64_t)n * m.C * m.HW + coff[k] + pp] : 0.f;
#pragma unroll
            for (int o = 0; o < TH_O; ++o) av[h][o] = (pv && o < m.O) ? p.a[((int64_t)n * m.O + o) * m.HW + pp] : 0.f;
        }
#pragma unroll
        for (int h = 0; h < TH_WU; ++h) {
            if (quant) {
#pragma unroll
                for (int k = 0; k < TH_CL; ++k) xv[h][k] = iao_code_m(xv[h][k], sc, inv_sc, zp, p.qmin, p.qmax) * sc;          // (an invalid pixel has a = 0)
            }
#pragma unroll
            for (int o = 0; o < TH_O; ++o) {
#pragma unroll
                for (int k = 0; k < TH_CL; ++k) acc[o][k] = fmaf(av[h][o], xv[h][k], acc[o][k]);
                if (want_db) db[o] += av[h][o];
            }
        }
    }
#pragma unroll
    for (int o = 0; o < TH_O; ++o) {
#pragma unroll
        for (int k = 0; k < TH_CL; ++k) {
            const float s = wave_reduce(acc[o][k], OpAddF());
            if (lane == 0) red[wave][o * TH_CL + k] = s;
        }
        if (want_db) {
            const float s = wave_reduce(db[o], OpAddF());
            if (lane == 0) red[wave][TH_O * TH_CL + o] = s;
        }
    }
    __syncthreads();
    if (tid < TH_O * TH_CL + TH_O) {
        float s = 0.f;
#pragma unroll
        for (int w_ = 0; w_ < TH_WW; ++w_) s += red[w_][tid];
        if (tid < TH_O * TH_CL) {
            const int o = tid / TH_CL, k = tid - o * TH_CL;
            if (o < m.O && c0 + k < m.C) {
                float* d = p.dw + (int64_t)o * m.C + c0 + k;
                *d = p.accumulate ? *d + s : s;
            }
        } else if (want_db && tid - TH_O * TH_CL < m.O) {
            p.dbias[tid - TH_O * TH_CL] = s;
        }
    }
}

struct ThinDParams {
    const float* gy; const float* dy; const float* x; const float* aqp;
    const float* qwt; const float* wt;          // transposed [C][16]: quantised folded weights, raw weights
    float* dx;
    float qmin, qmax;
    int relu_in;
    ThinGeom m;
};
#define TH_DW 8          // waves per backward-data block: channel slices
__global__ __launch_bounds__(64 * TH_DW) void k_thin_dgrad(const ThinDParams p) {
    const ThinGeom& m = p.m;
    const int tid = threadIdx.x, lane = tid & 63, wave = mn_uniform(tid >> 6);
    const uint32_t P = blockIdx.x * 64u + lane;
    if (P >= m.NP) return;
    const uint32_t n = fd_div(P, m.fd_hw);
    const int pp = (int)(P - n * (uint32_t)m.HW);
    const float sc = p.aqp[0], zp = p.aqp[1], slo = p.aqp[2], shi = p.aqp[3], inv_sc = 1.0f / sc;
    float g[TH_O], d[TH_O];
#pragma unroll
    for (int o = 0; o < TH_O; ++o) {
        const bool ov = o < m.O;
        g[o] = ov ? p.gy[((int64_t)n * m.O + o) * m.HW + pp] : 0.f;
        d[o] = ov ? p.dy[((int64_t)n * m.O + o) * m.HW + pp] : 0.f;
    }
    const int cs = m.C / TH_DW, c0 = wave * cs;
    const int64_t xb = (int64_t)n * m.C * m.HW + pp;
    float xn[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) xn[u] = p.x[xb + (int64_t)chan_phys(m.in_map, c0 + u) * m.HW];
    for (int cb = 0; cb < cs; cb += 8) {
        float xv[8];
        int64_t off[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { off[u] = xb + (int64_t)chan_phys(m.in_map, c0 + cb + u) * m.HW; xv[u] = xn[u]; }
        if (cb + 8 < cs) {          // the next step's loads fly during this step's arithmetic and stores
#pragma unroll
            for (int u = 0; u < 8; ++u) xn[u] = p.x[xb + (int64_t)chan_phys(m.in_map, c0 + cb + 8 + u) * m.HW];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float* __restrict__ q = p.qwt + (int64_t)(c0 + cb + u) * TH_O;
            const float* __restrict__ r = p.wt + (int64_t)(c0 + cb + u) * TH_O;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int o = 0; o < TH_O; ++o) { s1 = fmaf(q[o], g[o], s1); s2 = fmaf(r[o], d[o], s2); }
            float v = iao_fq_grad_m(s1, xv[u], sc, inv_sc, zp, slo, shi, p.qmin, p.qmax) + s2;
            if (p.relu_in) v = xv[u] > 0.f ? v : 0.f;
            p.dx[off[u]] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
static int plan_thin(const mn_conv_geom* g, ThinGeom* m) {
    if (!g || g->N <= 0 || g->KH != 1 || g->KW != 1 || g->stride_h != 1 || g->stride_w != 1 || g->pad_h || g->pad_w || g->groups != 1) return 0;
    if (g->O < 1 || g->O > TH_O || g->C < 64 || g->C % 64) return 0;          // (8 forward K slices of whole 8-channel steps; 4-channel backward-weight blocks)
    if (g->in_shuffle > 1 && g->C % g->in_shuffle) return 0;
    const int64_t NP = (int64_t)g->N * g->H * g->W;
    if (NP * g->C >= (1ll << 31) || NP < 64) return 0;
    if (!m) return 1;
    m->N = g->N; m->C = g->C; m->HW = g->H * g->W; m->O = g->O; m->NP = (uint32_t)NP; m->fd_hw = make_fastdiv((uint32_t)m->HW);
    m->in_map = make_chanmap(g->in_shuffle, g->C);
    return 1;
}
extern "C" int mn_iaobf_thin_supported(const mn_conv_geom* g) { return plan_thin(g, nullptr); }
extern "C" int64_t mn_iaobf_thin_mm_count(const mn_conv_geom* g) {
    ThinGeom m;
    return plan_thin(g, &m) ? (int64_t)((m.NP + 63u) / 64u) : 0;
}
// wt[C][16] = w[O][C] transposed, zero-padded
extern "C" int mn_iaobf_thin_pack(const float* w, int64_t O, int64_t Cc, float* wt, mn_stream_t stream) {
    if (!w || !wt || O < 1 || O > TH_O || Cc < 1 || Cc > (1 << 24)) MN_FAIL(MN_EINVAL, "mn_iaobf_thin_pack: bad arguments");
    mn_set_last_kernel("k_thin_pack");
    hipLaunchKernelGGL(k_thin_pack, dim3((unsigned)((Cc * TH_O + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (int)O, (int)Cc, wt);
    MN_CHECK_LAUNCH("mn_iaobf_thin_pack");
    return MN_OK;
}
// out = [relu](conv2d(v(x), w, bias)), v = the activation quantizer's fake-quantised value (aqp != NULL) or x itself; wt = mn_iaobf_thin_pack(w); mm nullable
extern "C" int mn_iaobf_thin_fwd(const mn_conv_geom* g, const float* x, const float* aqp, int a_bits, const float* wt, const float* bias, int relu, float* out, float* mm,
                                 mn_stream_t stream) {
    ThinFParams p;
    if (!plan_thin(g, &p.m)) MN_FAIL(MN_ENOTSUP, "mn_iaobf_thin_fwd: pointwise layers with <= 16 output channels and C %% 64 == 0 only");
    if (!x || !wt || !out || (aqp && (a_bits < 2 || a_bits > 24))) MN_FAIL(MN_EINVAL, "mn_iaobf_thin_fwd: bad arguments");
    p.x = x; p.wt = wt; p.bias = bias; p.aqp = aqp; p.out = out; p.mm = mm; p.relu = relu;
    const IaoRange r = iao_range(aqp ? a_bits : 8, 0, 1);
    p.qmin = r.qmin; p.qmax = r.qmax;
    mn_set_last_kernel("k_thin_fwd");
    hipLaunchKernelGGL(k_thin_fwd, dim3((p.m.NP + 63u) / 64u), dim3(64 * TH_FW), 0, (hipStream_t)stream, p);
    MN_CHECK_LAUNCH("mn_iaobf_thin_fwd");
    return MN_OK;
}
// dw (+)= conv2d_backward_weight(a, v(x)); dbias (nullable) = sum a
extern "C" int mn_iaobf_thin_bwd_weight(const mn_conv_geom* g, const float* a, const float* x, const float* aqp, int a_bits, int accumulate, float* dw, float* dbias,
                                        mn_stream_t stream) {
    ThinWParams p;
    if (!plan_thin(g, &p.m)) MN_FAIL(MN_ENOTSUP, "mn_iaobf_thin_bwd_weight: geometry not covered");
    if (!a || !x || !dw || (aqp && (a_bits < 2 || a_bits > 24))) MN_FAIL(MN_EINVAL, "mn_iaobf_thin_bwd_weight: bad arguments");
    p.a = a; p.x = x; p.aqp = aqp; p.dw = dw; p.dbias = dbias; p.accumulate = accumulate;
    const IaoRange r = iao_range(aqp ? a_bits : 8, 0, 1);
    p.qmin = r.qmin; p.qmax = r.qmax;
    mn_set_last_kernel("k_thin_wgrad");
    hipLaunchKernelGGL(k_thin_wgrad, dim3((unsigned)(p.m.C / TH_CL)), dim3(64 * TH_WW), 0, (hipStream_t)stream, p);
    MN_CHECK_LAUNCH("mn_iaobf_thin_bwd_weight");
    return MN_OK;
}
// dx = clip-STE_a(conv2d_backward_data(gy, qw)) + conv2d_backward_data(dy, w)  [* [x > 0]]; qwt / wt = mn_iaobf_thin_pack of qw / w
extern "C" int mn_iaobf_thin_bwd_data(const mn_conv_geom* g, const float* gy, const float* dy, const float* x, const float* aqp, int a_bits, const float* qwt,
                                      const float* wt, int relu_in, float* dx, mn_stream_t stream) {
    ThinDParams p;
    if (!plan_thin(g, &p.m)) MN_FAIL(MN_ENOTSUP, "mn_iaobf_thin_bwd_data: geometry not covered");
    if (!gy || !dy || !x || !aqp || !qwt || !wt || !dx || a_bits < 2 || a_bits > 24) MN_FAIL(MN_EINVAL, "mn_iaobf_thin_bwd_data: bad arguments");
    p.gy = gy; p.dy = dy; p.x = x; p.aqp = aqp; p.qwt = qwt; p.wt = wt; p.dx = dx; p.relu_in = relu_in;
    const IaoRange r = iao_range(a_bits, 0, 1);
    p.qmin = r.qmin; p.qmax = r.qmax;
    mn_set_last_kernel("k_thin_dgrad");
    hipLaunchKernelGGL(k_thin_dgrad, dim3((p.m.NP + 63u) / 64u), dim3(64 * TH_DW), 0, (hipStream_t)stream, p);
    MN_CHECK_LAUNCH("mn_iaobf_thin_bwd_data");
    return MN_OK;
}
