// The rest of the IAO module surface on gfx950 (reference: wqaq/iao/quantize.py):
//   * fake-quant fused with the activation behind it -- QuantReLU / QuantLeakyReLU / QuantSigmoid (ref 1160-1330): y = act(Q(x)) in ONE streaming
//     pass (the reference runs ~12 ATen kernels for Q plus one for the activation), backward dx = Q'(x) * act'(Q(x)) * g in one pass;
//   * fake-quant fused with average pooling -- QuantAvgPool2d (k x k, stride k, no padding) and QuantAdaptiveAvgPool2d((1, 1)) (ref 1375-1438):
//     the quantised tensor is never written;
//   * the PTQ percentile calibrator -- HistogramObserver (ref 116-139): the k-th smallest |x| by an exact 3-pass radix select on the fp32 bit
//     patterns (11 + 11 + 10 bits; integer atomics only, so the result is deterministic and bit-identical to torch.kthvalue), with the
//     first-call / moving-average update of max_val in the last launch: no host synchronisation.
// All HBM-bound streaming work: float4 per lane, grid-stride loops.
#include "common.h"

#include <math.h>

#define ACT_RELU 1
#define ACT_LEAKY 2
#define ACT_SIGMOID 3

struct FqC { float sc, zp, lo, hi, qmin, qmax, slope; int act; };

__device__ __forceinline__ float act_apply(float q, const FqC& c) {
    if (c.act == ACT_RELU) return q > 0.f ? q : 0.f;                       // at::relu = clamp_min(0): NaN propagates below
    if (c.act == ACT_LEAKY) return q > 0.f ? q : q * c.slope;
    return 1.f / (1.f + expf(-q));
}
__device__ __forceinline__ float fq_act_fwd1(float x, const FqC& c) {
    const float q = iao_fq(x, c.sc, c.zp, c.qmin, c.qmax);
    if (q != q) return q;
    return act_apply(q, c);
}
__device__ __forceinline__ float fq_act_bwd1(float g, float x, const FqC& c) {
    const float q = iao_fq(x, c.sc, c.zp, c.qmin, c.qmax);
    float d;
    if (c.act == ACT_RELU) d = q > 0.f ? g : 0.f;                           // threshold_backward: grad where result > 0
    else if (c.act == ACT_LEAKY) d = q > 0.f ? g : g * c.slope;
    else { const float y = 1.f / (1.f + expf(-q)); d = g * (1.f - y) * y; }  // sigmoid_backward: grad * (1 - y) * y
    return iao_fq_grad(d, x, c.sc, c.zp, c.lo, c.hi, c.qmin, c.qmax);
}

__global__ __launch_bounds__(256) void k_fq_act_fwd(const float* __restrict__ x, float* __restrict__ y, int64_t n, const float* __restrict__ qp, FqC c, int vec) {
    c.sc = qp[0]; c.zp = qp[1]; c.lo = qp[2]; c.hi = qp[3];
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    if (vec) {
        const int64_t n4 = n >> 2;
        for (int64_t j = t0; j < n4; j += stride) {
            const float4 v = reinterpret_cast<const float4*>(x)[j];
            reinterpret_cast<float4*>(y)[j] = make_float4(fq_act_fwd1(v.x, c), fq_act_fwd1(v.y, c), fq_act_fwd1(v.z, c), fq_act_fwd1(v.w, c));
        }
        for (int64_t j = (n4 << 2) + t0; j < n; j += stride) y[j] = fq_act_fwd1(x[j], c);
    } else {
        for (int64_t j = t0; j < n; j += stride) y[j] = fq_act_fwd1(x[j], c);
    }
}
__global__ __launch_bounds__(256) void k_fq_act_bwd(const float* __restrict__ g, const float* __restrict__ x, float* __restrict__ dx, int64_t n,
                                                    const float* __restrict__ qp, FqC c, int vec) {
    c.sc = qp[0]; c.zp = qp[1]; c.lo = qp[2]; c.hi = qp[3];
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
    if (vec) {
        const int64_t n4 = n >> 2;
        for (int64_t j = t0; j < n4; j += stride) {
            const float4 v = reinterpret_cast<const float4*>(x)[j], gg = reinterpret_cast<const float4*>(g)[j];
            reinterpret_cast<float4*>(dx)[j] = make_float4(fq_act_bwd1(gg.x, v.x, c), fq_act_bwd1(gg.y, v.y, c), fq_act_bwd1(gg.z, v.z, c), fq_act_bwd1(gg.w, v.w, c));
        }
        for (int64_t j = (n4 << 2) + t0; j < n; j += stride) dx[j] = fq_act_bwd1(g[j], x[j], c);
    } else {
        for (int64_t j = t0; j < n; j += stride) dx[j] = fq_act_bwd1(g[j], x[j], c);
    }
}
static int fq_c(FqC* c, int bits, int q_type, int act, float slope, const char* what) {
    if (bits < 2 || bits > 24 || act < ACT_RELU || act > ACT_SIGMOID) MN_FAIL(MN_EINVAL, "%s: bits=%d act=%d", what, bits, act);
    const IaoRange r = iao_range(bits, q_type, 1);
    c->qmin = r.qmin; c->qmax = r.qmax; c->slope = slope; c->act = act; c->sc = 1.f; c->zp = c->lo = c->hi = 0.f;
    return MN_OK;
}
extern "C" int mn_iao_fq_act_fwd(const float* x, float* y, int64_t n, const float* qp, int bits, int q_type, int act, float slope, mn_stream_t stream) {
    FqC c;
    if (n <= 0 || !x || !y || !qp) MN_FAIL(MN_EINVAL, "mn_iao_fq_act_fwd: bad arguments");
    int rc = fq_c(&c, bits, q_type, act, slope, "mn_iao_fq_act_fwd");
    if (rc) return rc;
    const int vec = aligned16(x) && aligned16(y);
    hipLaunchKernelGGL(k_fq_act_fwd, dim3(mn_grid_for(vec ? (n + 3) / 4 : n, 256, 2048)), dim3(256), 0, (hipStream_t)stream, x, y, n, qp, c, vec);
    MN_CHECK_LAUNCH("mn_iao_fq_act_fwd");
    return MN_OK;
}
extern "C" int mn_iao_fq_act_bwd(const float* g, const float* x, float* dx, int64_t n, const float* qp, int bits, int q_type, int act, float slope,
                                 mn_stream_t stream) {
    FqC c;
    if (n <= 0 || !g || !x || !dx || !qp) MN_FAIL(MN_EINVAL, "mn_iao_fq_act_bwd: bad arguments");
    int rc = fq_c(&c, bits, q_type, act, slope, "mn_iao_fq_act_bwd");
    if (rc) return rc;
    const int vec = aligned16(x) && aligned16(g) && aligned16(dx);
    hipLaunchKernelGGL(k_fq_act_bwd, dim3(mn_grid_for(vec ? (n + 3) / 4 : n, 256, 2048)), dim3(256), 0, (hipStream_t)stream, g, x, dx, n, qp, c, vec);
    MN_CHECK_LAUNCH("mn_iao_fq_act_bwd");
    return MN_OK;
}

// ------------------------------------------------------------------------------------------------ fake-quant + average pooling
// y[p][oh][ow] = (sum over the k x k window, rows then columns, fp32 -- the order of ATen's avg_pool2d) / (k*k); one thread per output.
__global__ __launch_bounds__(256) void k_fq_avgpool_fwd(const float* __restrict__ x, float* __restrict__ y, int64_t planes, int H, int W, int k,
                                                        const float* __restrict__ qp, FqC c) {
    c.sc = qp[0]; c.zp = qp[1];
    const int Ho = H / k, Wo = W / k;
    const int64_t total = planes * Ho * Wo;
    const float div = (float)(k * k);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ow = (int)(i % Wo);
        const int64_t t = i / Wo;
        const int oh = (int)(t % Ho);
        const int64_t p = t / Ho;
        const float* src = x + (p * H + (int64_t)oh * k) * W + (int64_t)ow * k;
        float s = 0.f;
        for (int r = 0; r < k; ++r)
            for (int q = 0; q < k; ++q) s += iao_fq(src[(int64_t)r * W + q], c.sc, c.zp, c.qmin, c.qmax);
        y[i] = s / div;
    }
}
// dx = Q'(x) * g[window] / (k*k); one thread per input element, float4 when W % 4 == 0 and k % 4 == 0 or k divides 4 ... kept scalar-simple: coalesced anyway
__global__ __launch_bounds__(256) void k_fq_avgpool_bwd(const float* __restrict__ g, const float* __restrict__ x, float* __restrict__ dx, int64_t planes, int H, int W,
                                                        int k, const float* __restrict__ qp, FqC c) {
    c.sc = qp[0]; c.zp = qp[1]; c.lo = qp[2]; c.hi = qp[3];
    const int Ho = H / k, Wo = W / k;
    const int64_t total = planes * H * W;
    const float div = (float)(k * k);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % W);
        const int64_t t = i / W;
        const int h = (int)(t % H);
        const int64_t p = t / H;
        const float gv = g[(p * Ho + h / k) * Wo + w / k] / div;
        dx[i] = iao_fq_grad(gv, x[i], c.sc, c.zp, c.lo, c.hi, c.qmin, c.qmax);
    }
}
// global average (AdaptiveAvgPool2d((1, 1))): one wave per plane, fp64 accumulation (order-independent to the last fp32 bit)
__global__ __launch_bounds__(64) void k_fq_gap_fwd(const float* __restrict__ x, float* __restrict__ y, int HW, const float* __restrict__ qp, FqC c) {
    c.sc = qp[0]; c.zp = qp[1];
    const float* src = x + (int64_t)blockIdx.x * HW;
    double s = 0.0;
    for (int i = threadIdx.x; i < HW; i += 64) s += (double)iao_fq(src[i], c.sc, c.zp, c.qmin, c.qmax);
    s = wave_reduce(s, OpAddD());
    if (threadIdx.x == 0) y[blockIdx.x] = (float)(s / (double)HW);
}
extern "C" int mn_iao_fq_avgpool_supported(int64_t H, int64_t W, int64_t k) { return k >= 1 && k <= 64 && H >= k && W >= k && H % k == 0 && W % k == 0; }
extern "C" int mn_iao_fq_avgpool_fwd(const float* x, float* y, int64_t planes, int64_t H, int64_t W, int64_t k, const float* qp, int bits, int q_type,
                                     mn_stream_t stream) {
    FqC c;
    if (planes <= 0 || !x || !y || !qp || !mn_iao_fq_avgpool_supported(H, W, k)) MN_FAIL(MN_EINVAL, "mn_iao_fq_avgpool_fwd: bad arguments");
    int rc = fq_c(&c, bits, q_type, ACT_RELU, 0.f, "mn_iao_fq_avgpool_fwd");
    if (rc) return rc;
    if (k == H && k == W) {
        hipLaunchKernelGGL(k_fq_gap_fwd, dim3((unsigned)planes), dim3(64), 0, (hipStream_t)stream, x, y, (int)(H * W), qp, c);
    } else {
        const int64_t total = planes * (H / k) * (W / k);
        hipLaunchKernelGGL(k_fq_avgpool_fwd, dim3(mn_grid_for(total, 256, 4096)), dim3(256), 0, (hipStream_t)stream, x, y, planes, (int)H, (int)W, (int)k, qp, c);
    }
    MN_CHECK_LAUNCH("mn_iao_fq_avgpool_fwd");
    return MN_OK;
}
extern "C" int mn_iao_fq_avgpool_bwd(const float* g, const float* x, float* dx, int64_t planes, int64_t H, int64_t W, int64_t k, const float* qp, int bits,
                                     int q_type, mn_stream_t stream) {
    FqC c;
    if (planes <= 0 || !g || !x || !dx || !qp || !mn_iao_fq_avgpool_supported(H, W, k)) MN_FAIL(MN_EINVAL, "mn_iao_fq_avgpool_bwd: bad arguments");
    int rc = fq_c(&c, bits, q_type, ACT_RELU, 0.f, "mn_iao_fq_avgpool_bwd");
    if (rc) return rc;
    hipLaunchKernelGGL(k_fq_avgpool_bwd, dim3(mn_grid_for(planes * H * W, 256, 4096)), dim3(256), 0, (hipStream_t)stream, g, x, dx, planes, (int)H, (int)W, (int)k, qp, c);
    MN_CHECK_LAUNCH("mn_iao_fq_avgpool_bwd");
    return MN_OK;
}

// ------------------------------------------------------------------------------------------------ HistogramObserver: exact k-th smallest |x|
// ws (uint32): [0 .. 2048) histogram of the current digit, [2048] prefix (the bits decided so far), [2049] remaining rank (1-based) inside the prefix
// bucket.  Pass d (0, 1, 2) histograms digit d of the elements whose higher digits equal the prefix; k_kth_pick then finds the digit that holds the
// wanted rank.  |x| of a finite or infinite float orders like its bit pattern; NaN patterns sort above +inf, where torch.kthvalue also puts them.
#define KTH_BINS 2048
__device__ __forceinline__ uint32_t kth_digit(uint32_t u, int pass) { return pass == 0 ? (u >> 21) : (pass == 1 ? ((u >> 10) & 2047u) : (u & 1023u)); }
__device__ __forceinline__ uint32_t kth_hi(uint32_t u, int pass) { return pass == 0 ? 0u : (pass == 1 ? (u >> 21) : (u >> 10)); }

__global__ __launch_bounds__(256) void k_kth_hist(const float* __restrict__ x, int64_t n, int pass, uint32_t* __restrict__ ws) {
    __shared__ uint32_t h[KTH_BINS];
    for (int i = threadIdx.x; i < KTH_BINS; i += 256) h[i] = 0u;
    __syncthreads();
    const uint32_t prefix = pass == 0 ? 0u : ws[KTH_BINS];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t u = mn_f2u(x[i]) & 0x7fffffffu;
        if (kth_hi(u, pass) == prefix) atomicAdd(&h[kth_digit(u, pass)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < KTH_BINS; i += 256)
        if (h[i]) atomicAdd(&ws[i], h[i]);
}
// one block: scan the histogram for the bucket that holds the rank; last pass: write the value and update max_val (first call copies, else EMA)
__global__ __launch_bounds__(256) void k_kth_pick(uint32_t* __restrict__ ws, int pass, uint32_t k_rank, int first, float keep, float momentum, float* __restrict__ max_val,
                                                  float* __restrict__ out) {
    __shared__ uint32_t cnt[256];
    __shared__ uint32_t sel[2];
    const int nb = pass == 2 ? 1024 : 2048, per = nb / 256;
    uint32_t local = 0u;
    for (int i = 0; i < per; ++i) local += ws[threadIdx.x * per + i];
    cnt[threadIdx.x] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t rank = pass == 0 ? k_rank : ws[KTH_BINS + 1];
        int t = 0;
        while (t < 255 && rank > cnt[t]) { rank -= cnt[t]; ++t; }
        int b = t * per;
        while (b < t * per + per - 1 && rank > ws[b]) { rank -= ws[b]; ++b; }
        sel[0] = (uint32_t)b; sel[1] = rank;
    }
    __syncthreads();
    const uint32_t b = sel[0], rank = sel[1];
    __syncthreads();
    for (int i = threadIdx.x; i < KTH_BINS; i += 256) ws[i] = 0u;          // ready for the next pass
    if (threadIdx.x == 0) {
        const uint32_t prefix = pass == 0 ? b : (pass == 1 ? ((ws[KTH_BINS] << 11) | b) : ((ws[KTH_BINS] << 10) | b));
        ws[KTH_BINS] = prefix; ws[KTH_BINS + 1] = rank;
        if (pass == 2) {
            const float cur = mn_u2f(prefix);
            if (out) *out = cur;
            if (max_val) *max_val = first ? cur : keep * (*max_val) + momentum * cur;
        }
    }
}
extern "C" int64_t mn_kth_abs_ws_bytes(void) { return (KTH_BINS + 16) * 4; }
/* HistogramObserver.forward (ref 126-139): cur = k-th smallest |x| (k 1-based, = int(percentile * n)); max_val = cur on the first call, else
 * (1 - momentum) * max_val + momentum * cur.  `out` (nullable) receives cur.  ws: mn_kth_abs_ws_bytes() bytes, 16-byte aligned. */
extern "C" int mn_hist_observe(const float* x, int64_t n, int64_t k, int first, double momentum, float* max_val, float* out, void* ws, mn_stream_t stream) {
    if (!x || n <= 0 || k < 1 || k > n || n > 0xffffffffll || !ws || (((uintptr_t)ws) & 3)) MN_FAIL(MN_EINVAL, "mn_hist_observe: bad arguments (n=%lld k=%lld)", (long long)n, (long long)k);
    hipStream_t s = (hipStream_t)stream;
    uint32_t* w = (uint32_t*)ws;
    if (hipMemsetAsync(w, 0, (KTH_BINS + 16) * 4, s) != hipSuccess) MN_FAIL(MN_EHIP, "mn_hist_observe: memset failed");
    const int grid = mn_grid_for(n, 256 * 8, 1024);
    for (int pass = 0; pass < 3; ++pass) {
        hipLaunchKernelGGL(k_kth_hist, dim3(grid), dim3(256), 0, s, x, n, pass, w);
        hipLaunchKernelGGL(k_kth_pick, dim3(1), dim3(256), 0, s, w, pass, (uint32_t)k, first, (float)(1.0 - momentum), (float)momentum, max_val, out);
    }
    MN_CHECK_LAUNCH("mn_hist_observe");
    return MN_OK;
}


// ---------------------------------------------------------------- BatchNorm folding of QuantBNFuseConv2d (wqaq/iao/quantize.py:900-956)
//   k_b = gamma / sqrt(var_b + eps);  bias_f = beta + (bias - mean) * k_b   (beta - mean * k_b without a conv bias)
//   w_f[o][:] = w[o][:] * (gamma / sqrt(var_w + eps))
// as ONE launch (one block per out-channel) instead of ~8 element-wise launches forward and ~25 backward per layer; every step rounds as the
// reference's separate fp32 ops do (IEEE divide / sqrt, no contraction).  Backward: the analytic gradients of the same expressions, row sums in fp64.
__global__ __launch_bounds__(256) void k_iao_bnfold_fwd(const float* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ var_b,
                                                        const float* __restrict__ var_w, float eps, int K, float* __restrict__ wf, float* __restrict__ bf) {
    const int o = blockIdx.x;
    const float kw = gamma[o] / sqrtf(var_w[o] + eps);
    const float* __restrict__ wr = w + (int64_t)o * K;
    float* __restrict__ dst = wf + (int64_t)o * K;
    for (int i = threadIdx.x; i < K; i += blockDim.x) dst[i] = wr[i] * kw;
    if (threadIdx.x == 0) {
        const float kb = gamma[o] / sqrtf(var_b[o] + eps);
        bf[o] = bias ? beta[o] + (bias[o] - mean[o]) * kb : beta[o] - mean[o] * kb;
    }
}
__global__ __launch_bounds__(256) void k_iao_bnfold_bwd(const float* __restrict__ dwf, const float* __restrict__ dbf, const float* __restrict__ w,
                                                        const float* __restrict__ bias, const float* __restrict__ gamma, const float* __restrict__ mean,
                                                        const float* __restrict__ var_b, const float* __restrict__ var_w, float eps, int K,
                                                        float* __restrict__ dw, float* __restrict__ dbias, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta, float* __restrict__ dmean, float* __restrict__ dvar_b,
                                                        float* __restrict__ dvar_w) {
    __shared__ double scd[16];
    const int o = blockIdx.x;
    const float rw = 1.0f / sqrtf(var_w[o] + eps), kw = gamma[o] / sqrtf(var_w[o] + eps);
    const float* __restrict__ wr = w + (int64_t)o * K;
    const float* __restrict__ gr = dwf + (int64_t)o * K;
    double S = 0.0;
    for (int i = threadIdx.x; i < K; i += blockDim.x) {
        const float g = gr[i];
        if (dw) dw[(int64_t)o * K + i] = g * kw;
        S += (double)g * (double)wr[i];
    }
    S = block_reduce(S, OpAddD(), 0.0, scd);
    if (threadIdx.x == 0) {
        const float rb = 1.0f / sqrtf(var_b[o] + eps), kb = gamma[o] / sqrtf(var_b[o] + eps);
        const float g = dbf[o];
        const double D = (double)g * (bias ? (double)bias[o] - (double)mean[o] : -(double)mean[o]);
        if (dgamma) dgamma[o] = (float)(S * (double)rw + D * (double)rb);
        if (dbeta) dbeta[o] = g;
        if (dbias) dbias[o] = g * kb;
        if (dmean) dmean[o] = -(g * kb);
        const double vw = (double)var_w[o] + (double)eps, vb = (double)var_b[o] + (double)eps;
        if (dvar_w) dvar_w[o] = (float)(S * (double)gamma[o] * -0.5 / (vw * sqrt(vw)));
        if (dvar_b) dvar_b[o] = (float)(D * (double)gamma[o] * -0.5 / (vb * sqrt(vb)));
    }
}
extern "C" int mn_iao_bnfold_fwd(const float* w, const float* bias, const float* gamma, const float* beta, const float* mean, const float* var_b, const float* var_w,
                                 float eps, int64_t O, int64_t K, float* wf, float* bf, mn_stream_t stream) {
    if (!w || !gamma || !beta || !mean || !var_b || !var_w || !wf || !bf || O <= 0 || K <= 0 || K > 0x7fffffff) MN_FAIL(MN_EINVAL, "mn_iao_bnfold_fwd: bad arguments");
    hipLaunchKernelGGL(k_iao_bnfold_fwd, dim3((unsigned)O), dim3(256), 0, (hipStream_t)stream, w, bias, gamma, beta, mean, var_b, var_w, eps, (int)K, wf, bf);
    MN_CHECK_LAUNCH("mn_iao_bnfold_fwd");
    return MN_OK;
}
extern "C" int mn_iao_bnfold_bwd(const float* dwf, const float* dbf, const float* w, const float* bias, const float* gamma, const float* mean, const float* var_b,
                                 const float* var_w, float eps, int64_t O, int64_t K, float* dw, float* dbias, float* dgamma, float* dbeta, float* dmean,
                                 float* dvar_b, float* dvar_w, mn_stream_t stream) {
    if (!dwf || !dbf || !w || !gamma || !mean || !var_b || !var_w || O <= 0 || K <= 0 || K > 0x7fffffff) MN_FAIL(MN_EINVAL, "mn_iao_bnfold_bwd: bad arguments");
    hipLaunchKernelGGL(k_iao_bnfold_bwd, dim3((unsigned)O), dim3(256), 0, (hipStream_t)stream, dwf, dbf, w, bias, gamma, mean, var_b, var_w, eps, (int)K, dw, dbias,
                       dgamma, dbeta, dmean, dvar_b, dvar_w);
    MN_CHECK_LAUNCH("mn_iao_bnfold_bwd");
    return MN_OK;
}

// ---------------------------------------------------------------- QuantMaxPool2d(2, 2) in one pass per direction (wqaq/iao/quantize.py:1347-1359: max_pool2d(Q(x)))
// forward: y = the 2 x 2 / stride-2 maximum of the fake-quantised input with ATen's window rule (row-major, a later element replaces the maximum if it is greater
// or NaN), the argmax of every window as one byte, and -- mm != NULL -- per-block (min, max) of y for the observer of the layer that reads it.  Q(x) is never
// written.  backward: the pooled gradient goes to the window's argmax, through the quantizer's clip-STE (Round.backward 163-168 + the clamp of 232), and
// -- relu_mask != 0: x is the output of a ReLU -- through that ReLU's mask [x > 0]: one read of (gy, idx, x), one write of dx.
// Thread = 4 consecutive windows of one output row.
__global__ __launch_bounds__(256) void k_iao_fq_pool2_fwd(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ idx, int64_t nq, int H, int W,
                                                          const float* __restrict__ qp, float qmin, float qmax, float* __restrict__ mm) {
    __shared__ float scf[16];
    const float sc = qp[0], zp = qp[1];
    const int Wo = W >> 1, Ho = H >> 1, q4 = Wo >> 2;
    float mlo = INFINITY, mhi = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / q4;
        const int qc = (int)(i - row * q4);
        const int64_t pl = row / Ho;
        const int orow = (int)(row - pl * Ho);
        const float* src = x + (pl * H + 2 * orow) * W + qc * 8;
        const float4 a0 = *reinterpret_cast<const float4*>(src), a1 = *reinterpret_cast<const float4*>(src + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(src + W), b1 = *reinterpret_cast<const float4*>(src + W + 4);
        const float r0[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, r1[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        float o[4];
        uint32_t ib = 0u;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float v[4] = {iao_fq(r0[2 * w], sc, zp, qmin, qmax), iao_fq(r0[2 * w + 1], sc, zp, qmin, qmax), iao_fq(r1[2 * w], sc, zp, qmin, qmax),
                                iao_fq(r1[2 * w + 1], sc, zp, qmin, qmax)};
            float m = -INFINITY;
            uint32_t k = 0u;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (v[e] > m || v[e] != v[e]) { m = v[e]; k = (uint32_t)e; }
            o[w] = m; ib |= k << (8 * w);
            mlo = OpMinF()(mlo, m); mhi = OpMaxF()(mhi, m);
        }
        const int64_t oo = (pl * Ho + orow) * Wo + qc * 4;
        *reinterpret_cast<float4*>(y + oo) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<uint32_t*>(idx + oo) = ib;
    }
    if (mm) {
        mlo = block_reduce(mlo, OpMinF(), INFINITY, scf);
        mhi = block_reduce(mhi, OpMaxF(), -INFINITY, scf);
        if (threadIdx.x == 0) { mm[blockIdx.x] = mlo; mm[gridDim.x + blockIdx.x] = mhi; }
    }
}
__global__ __launch_bounds__(256) void k_iao_fq_pool2_bwd(const float* __restrict__ gy, const unsigned char* __restrict__ idx, const float* __restrict__ x,
                                                          float* __restrict__ dx, int64_t nq, int H, int W, const float* __restrict__ qp, float qmin, float qmax,
                                                          int relu_mask) {
    const float sc = qp[0], zp = qp[1], lo = qp[2], hi = qp[3];
    const int Wo = W >> 1, Ho = H >> 1, q4 = Wo >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / q4;
        const int qc = (int)(i - row * q4);
        const int64_t pl = row / Ho;
        const int orow = (int)(row - pl * Ho);
        const int64_t oo = (pl * Ho + orow) * Wo + qc * 4;
        const float4 g4 = *reinterpret_cast<const float4*>(gy + oo);
        const uint32_t ib = *reinterpret_cast<const uint32_t*>(idx + oo);
        const float g[4] = {g4.x, g4.y, g4.z, g4.w};
        const int64_t so = (pl * H + 2 * orow) * W + qc * 8;
        const float4 a0 = *reinterpret_cast<const float4*>(x + so), a1 = *reinterpret_cast<const float4*>(x + so + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(x + so + W), b1 = *reinterpret_cast<const float4*>(x + so + W + 4);
        const float x0[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, x1[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        float r0[8], r1[8];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t k = (ib >> (8 * w)) & 3u;
            r0[2 * w] = k == 0u ? g[w] : 0.f; r0[2 * w + 1] = k == 1u ? g[w] : 0.f;
            r1[2 * w] = k == 2u ? g[w] : 0.f; r1[2 * w + 1] = k == 3u ? g[w] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            r0[e] = iao_fq_grad(r0[e], x0[e], sc, zp, lo, hi, qmin, qmax);
            r1[e] = iao_fq_grad(r1[e], x1[e], sc, zp, lo, hi, qmin, qmax);
            if (relu_mask) { r0[e] = x0[e] > 0.f ? r0[e] : 0.f; r1[e] = x1[e] > 0.f ? r1[e] : 0.f; }
        }
        float* dst = dx + so;
        *reinterpret_cast<float4*>(dst) = make_float4(r0[0], r0[1], r0[2], r0[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(r0[4], r0[5], r0[6], r0[7]);
        *reinterpret_cast<float4*>(dst + W) = make_float4(r1[0], r1[1], r1[2], r1[3]);
        *reinterpret_cast<float4*>(dst + W + 4) = make_float4(r1[4], r1[5], r1[6], r1[7]);
    }
}
static int fq_pool_grid(int64_t planes, int64_t H, int64_t W) { return mn_grid_for(planes * (H / 2) * (W / 8), 256, 4096); }
extern "C" int mn_iao_fq_maxpool2x2_supported(int64_t H, int64_t W) { return H >= 2 && H % 2 == 0 && W >= 8 && W % 8 == 0; }
extern "C" int64_t mn_iao_fq_maxpool2x2_mm_count(int64_t planes, int64_t H, int64_t W) {
    return (planes > 0 && mn_iao_fq_maxpool2x2_supported(H, W)) ? fq_pool_grid(planes, H, W) : 0;
}
extern "C" int mn_iao_fq_maxpool2x2_fwd(const float* x, int64_t planes, int64_t H, int64_t W, const float* qp, int bits, int q_type, float* y, uint8_t* idx, float* mm,
                                        mn_stream_t stream) {
    if (!x || !y || !idx || !qp || planes <= 0 || bits < 2 || bits > 24 || (q_type != 0 && q_type != 1) || !mn_iao_fq_maxpool2x2_supported(H, W) || !aligned16(x) ||
        !aligned16(y) || (((uintptr_t)idx) & 3))
        MN_FAIL(MN_EINVAL, "mn_iao_fq_maxpool2x2_fwd: needs even H, W %% 8 == 0, aligned tensors, 2..24 bits");
    const IaoRange r = iao_range(bits, q_type, 1);
    const int64_t nq = planes * (H / 2) * (W / 8);
    mn_set_last_kernel("k_iao_fq_pool2_fwd"); mn_prof_bytes(5.25 * (double)planes * H * W); mn_prof_begin((hipStream_t)stream);
    hipLaunchKernelGGL(k_iao_fq_pool2_fwd, dim3(fq_pool_grid(planes, H, W)), dim3(256), 0, (hipStream_t)stream, x, y, (unsigned char*)idx, nq, (int)H, (int)W, qp, r.qmin,
                       r.qmax, mm);
    mn_prof_end((hipStream_t)stream);
    MN_CHECK_LAUNCH("mn_iao_fq_maxpool2x2_fwd");
    return MN_OK;
}
extern "C" int mn_iao_fq_maxpool2x2_bwd(const float* gy, const uint8_t* idx, const float* x, int64_t planes, int64_t H, int64_t W, const float* qp, int bits, int q_type,
                                        int relu_mask, float* dx, mn_stream_t stream) {
    if (!gy || !dx || !idx || !x || !qp || planes <= 0 || bits < 2 || bits > 24 || (q_type != 0 && q_type != 1) || !mn_iao_fq_maxpool2x2_supported(H, W) ||
        !aligned16(gy) || !aligned16(dx) || !aligned16(x) || (((uintptr_t)idx) & 3))
        MN_FAIL(MN_EINVAL, "mn_iao_fq_maxpool2x2_bwd: needs even H, W %% 8 == 0, aligned tensors, 2..24 bits");
    const IaoRange r = iao_range(bits, q_type, 1);
    const int64_t nq = planes * (H / 2) * (W / 8);
    mn_set_last_kernel("k_iao_fq_pool2_bwd"); mn_prof_bytes(9.25 * (double)planes * H * W); mn_prof_begin((hipStream_t)stream);
    hipLaunchKernelGGL(k_iao_fq_pool2_bwd, dim3(fq_pool_grid(planes, H, W)), dim3(256), 0, (hipStream_t)stream, gy, (const unsigned char*)idx, x, dx, nq, (int)H, (int)W, qp,
                       r.qmin, r.qmax, relu_mask);
    mn_prof_end((hipStream_t)stream);
    MN_CHECK_LAUNCH("mn_iao_fq_maxpool2x2_bwd");
    return MN_OK;
}

// ---------------------------------------------------------------- small streaming helpers of the BN-fused blocks
// out = (a [+ b]) * [x > 0] (x == NULL: no mask): the sum of the two input gradients of QuantBNFuseConv2d (quantised path + raw statistics path) with the ReLU mask of
// the block in front, or the ReLU backward alone -- one pass instead of ATen's add / gt / where kernels.  n % 4 == 0, 16-byte aligned.
__global__ __launch_bounds__(256) void k_add_mask(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ x, float* __restrict__ out, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<const float4*>(a)[i];
        if (b) { const float4 w = reinterpret_cast<const float4*>(b)[i]; v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
        if (x) {
            const float4 m = reinterpret_cast<const float4*>(x)[i];
            v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
        }
        reinterpret_cast<float4*>(out)[i] = v;
    }
}
extern "C" int mn_add_relu_mask(const float* a, const float* b, const float* x, float* out, int64_t n, mn_stream_t stream) {
    if (!a || !out || n <= 0 || n % 4 || !aligned16(a) || !aligned16(out) || (b && !aligned16(b)) || (x && !aligned16(x)))
        MN_FAIL(MN_EINVAL, "mn_add_relu_mask: needs n %% 4 == 0 and 16-byte aligned tensors");
    mn_set_last_kernel("k_add_mask");
    hipLaunchKernelGGL(k_add_mask, dim3(mn_grid_for(n / 4, 256, 4096)), dim3(256), 0, (hipStream_t)stream, a, b, x, out, n / 4);
    MN_CHECK_LAUNCH("mn_add_relu_mask");
    return MN_OK;
}
// y = relu(x) in place or out of place with per-block (min, max) of the result (mm nullable: 2 * mn_relu_mm_count(n) floats): the ReLU behind a BN-fused conv whose
// kernel has no ReLU epilogue, feeding the next layer's observer
static int relu_mm_grid(int64_t n) { return mn_grid_for(n / 4, 256, 2048); }
__global__ __launch_bounds__(256) void k_relu_mm(const float* __restrict__ x, float* __restrict__ y, int64_t n4, float* __restrict__ mm) {
    __shared__ float scf[16];
    float lo = INFINITY, hi = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<const float4*>(x)[i];
        v.x = qa_relu(v.x); v.y = qa_relu(v.y); v.z = qa_relu(v.z); v.w = qa_relu(v.w);
        lo = OpMinF()(OpMinF()(lo, v.x), OpMinF()(OpMinF()(v.y, v.z), v.w));
        hi = OpMaxF()(OpMaxF()(hi, v.x), OpMaxF()(OpMaxF()(v.y, v.z), v.w));
        reinterpret_cast<float4*>(y)[i] = v;
    }
    if (mm) {
        lo = block_reduce(lo, OpMinF(), INFINITY, scf);
        hi = block_reduce(hi, OpMaxF(), -INFINITY, scf);
        if (threadIdx.x == 0) { mm[blockIdx.x] = lo; mm[gridDim.x + blockIdx.x] = hi; }
    }
}
extern "C" int64_t mn_relu_mm_count(int64_t n) { return (n > 0 && n % 4 == 0) ? relu_mm_grid(n) : 0; }
extern "C" int mn_relu_mm(const float* x, float* y, int64_t n, float* mm, mn_stream_t stream) {
    if (!x || !y || n <= 0 || n % 4 || !aligned16(x) || !aligned16(y)) MN_FAIL(MN_EINVAL, "mn_relu_mm: needs n %% 4 == 0 and 16-byte aligned tensors");
    mn_set_last_kernel("k_relu_mm");
    hipLaunchKernelGGL(k_relu_mm, dim3(relu_mm_grid(n)), dim3(256), 0, (hipStream_t)stream, x, y, n / 4, mm);
    MN_CHECK_LAUNCH("mn_relu_mm");
    return MN_OK;
}
