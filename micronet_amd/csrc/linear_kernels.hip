// QuantLinear with few outputs (the classifier of the reference's ResNets: 512 -> 10, models/resnet.py:141,176 under wqaq/dorefa/quantize.py:192-199 /
// wqaq/iao/quantize.py:1150-1157) for gfx950:
//     y[n][o] = bias[o] + sum over c of Q_a(x[n][c]) * wq[o][c]
// with the activation quantizer Q_a (DoReFa k-bit, IAO per-tensor, or none) evaluated in registers and its clip-STE applied to dx in the same launch.
// N x C x O is tiny (256 x 512 x 10): three latency-bound launches of a few microseconds -- the generic kernels treat a linear layer as a 1 x 1 conv over
// 1 x 1 images and ran this one on the direct VALU path (210 us forward).  One wave per sample row (forward, backward-data); backward-weight: see k_qlin_bwd_weight.
#include "common.h"

#define QL_OMAX 16            // outputs per pass (larger O: several passes)

// forward: a block (4 waves) per sample row -- every lane's loads are issued at once (one wave per row paid a full memory latency per 64 channels: 38 us
// for 256 x 512 x 10), the four waves' partial sums are added in wave order through LDS (fixed order: deterministic)
__global__ __launch_bounds__(256) void k_qlin_fwd(const Pro pro, const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y,
                                                  int N, int C, int O) {
    __shared__ float part[4][QL_OMAX];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float sc = 1.f, zp = 0.f;
    if (pro.mode == MN_ACTQ_IAO) { sc = pro.qp[0]; zp = pro.qp[1]; }
    for (int o0 = 0; o0 < O; o0 += QL_OMAX) {
        float acc[QL_OMAX];
#pragma unroll
        for (int k = 0; k < QL_OMAX; ++k) acc[k] = 0.f;
        for (int c = tid; c < C; c += 256) {
            const float q = pro_apply(pro, x[(int64_t)n * C + c], sc, zp);
#pragma unroll
            for (int k = 0; k < QL_OMAX; ++k)
                if (o0 + k < O) acc[k] = fmaf(q, w[(int64_t)(o0 + k) * C + c], acc[k]);
        }
        __syncthreads();                      // (the previous pass's reads of `part` are done)
#pragma unroll
        for (int k = 0; k < QL_OMAX; ++k) {
            if (o0 + k >= O) break;
            const float v = wave_reduce(acc[k], OpAddF());
            if (lane == 0) part[wave][k] = v;
        }
        __syncthreads();
        if (tid < QL_OMAX && o0 + tid < O)
            y[(int64_t)n * O + o0 + tid] = ((part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid])) + (bias ? bias[o0 + tid] : 0.f);
    }
}
// dx[n][c] = STE(sum over o of gy[n][o] * wq[o][c])
__global__ __launch_bounds__(256) void k_qlin_bwd_data(const Pro ste, const float* __restrict__ gy, const float* __restrict__ w, const float* __restrict__ x,
                                                       float* __restrict__ dx, int N, int C, int O) {
    const int64_t total = (int64_t)N * C;
    float sc = 1.f, zp = 0.f, lo = 0.f, hi = 0.f;
    if (ste.mode == MN_ACTQ_IAO) { sc = ste.qp[0]; zp = ste.qp[1]; lo = ste.qp[2]; hi = ste.qp[3]; }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int n = (int)(i / C), c = (int)(i - (int64_t)n * C);
        float acc = 0.f;
        for (int o = 0; o < O; ++o) acc = fmaf(gy[(int64_t)n * O + o], w[(int64_t)o * C + c], acc);
        if (ste.mode == MN_ACTQ_DOREFA) acc = dorefa_act_grad(acc, x[i], ste.s);
        else if (ste.mode == MN_ACTQ_IAO) acc = iao_fq_grad(acc, x[i], sc, zp, lo, hi, ste.qmin, ste.qmax);
        dx[i] = acc;
    }
}
// dw[o][c] = sum over n of gy[n][o] * Q_a(x[n][c]);  db[o] = sum over n of gy[n][o].  A block owns 16 input channels; its 256 threads are 16 groups that each sum
// the samples n = group (mod 16) for their channel, the sixteen partial sums are then added in group order through LDS (fixed order: deterministic).  (64 channels
// per block left 8 blocks on 256 CUs with 32 dependent iterations each: 44 us.)
#define QL_NG 16
#define QL_CB 16
__global__ __launch_bounds__(256) void k_qlin_bwd_weight(const Pro pro, const float* __restrict__ gy, const float* __restrict__ x, float* __restrict__ dw, float* __restrict__ db,
                                                         int N, int C, int O) {
    __shared__ float red[QL_NG][QL_OMAX + 1][QL_CB];
    const int cl = threadIdx.x & (QL_CB - 1), grp = threadIdx.x / QL_CB, c = blockIdx.x * QL_CB + cl;
    float sc = 1.f, zp = 0.f;
    if (pro.mode == MN_ACTQ_IAO) { sc = pro.qp[0]; zp = pro.qp[1]; }
    for (int o0 = 0; o0 < O; o0 += QL_OMAX) {
        float acc[QL_OMAX], bacc[QL_OMAX];
#pragma unroll
        for (int k = 0; k < QL_OMAX; ++k) { acc[k] = 0.f; bacc[k] = 0.f; }
#pragma unroll 4
        for (int n = grp; n < N; n += QL_NG) {
            const float q = c < C ? pro_apply(pro, x[(int64_t)n * C + c], sc, zp) : 0.f;
#pragma unroll
            for (int k = 0; k < QL_OMAX; ++k)
                if (o0 + k < O) {
                    const float g = gy[(int64_t)n * O + o0 + k];
                    acc[k] = fmaf(g, q, acc[k]);
                    bacc[k] += g;
                }
        }
        __syncthreads();                      // (the previous pass's reads of `red` are done)
#pragma unroll
        for (int k = 0; k < QL_OMAX; ++k) red[grp][k][cl] = acc[k];
        if (cl == 0) {
#pragma unroll
            for (int k = 0; k < QL_OMAX; ++k) red[grp][QL_OMAX][k] = bacc[k];
        }
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int k = 0; k < QL_OMAX; ++k) {
                if (o0 + k >= O) break;
                float a = 0.f;
#pragma unroll
                for (int g2 = 0; g2 < QL_NG; ++g2) a += red[g2][k][cl];
                if (c < C) dw[(int64_t)(o0 + k) * C + c] = a;
            }
            if (db && blockIdx.x == 0 && cl < QL_OMAX && o0 + cl < O) {
                float a = 0.f;
#pragma unroll
                for (int g2 = 0; g2 < QL_NG; ++g2) a += red[g2][QL_OMAX][cl];
                db[o0 + cl] = a;
            }
        }
    }
}

extern "C" int mn_qlinear_supported(int64_t N, int64_t C, int64_t O) { return N >= 1 && C >= 1 && O >= 1 && O <= 64 && N * C < ((int64_t)1 << 31); }
extern "C" int mn_qlinear_fwd(const mn_actq* aq, const float* x, const float* w, const float* bias, float* y, int64_t N, int64_t C, int64_t O, mn_stream_t stream) {
    if (!mn_qlinear_supported(N, C, O) || !x || !w || !y) MN_FAIL(MN_EINVAL, "mn_qlinear_fwd: bad arguments");
    Pro pro;
    int rc = make_pro(aq, &pro, 0, "mn_qlinear_fwd");
    if (rc) return rc;
    if (pro.mode == MN_ACTQ_SIGN8 || pro.mode == MN_ACTQ_CODE8) MN_FAIL(MN_ENOTSUP, "mn_qlinear_fwd: fp32 activations only");
    mn_set_last_kernel("k_qlin_fwd");
    hipLaunchKernelGGL(k_qlin_fwd, dim3((unsigned)N), dim3(256), 0, (hipStream_t)stream, pro, x, w, bias, y, (int)N, (int)C, (int)O);
    MN_CHECK_LAUNCH("mn_qlinear_fwd");
    return MN_OK;
}
extern "C" int mn_qlinear_bwd_data(const mn_actq* aq, const float* gy, const float* w, const float* x, float* dx, int64_t N, int64_t C, int64_t O, mn_stream_t stream) {
    if (!mn_qlinear_supported(N, C, O) || !gy || !w || !dx) MN_FAIL(MN_EINVAL, "mn_qlinear_bwd_data: bad arguments");
    Pro ste;
    int rc = make_pro(aq, &ste, 1, "mn_qlinear_bwd_data");
    if (rc) return rc;
    if (ste.mode == MN_ACTQ_SIGN8 || ste.mode == MN_ACTQ_CODE8) MN_FAIL(MN_ENOTSUP, "mn_qlinear_bwd_data: fp32 activations only");
    if (ste.mode != MN_ACTQ_NONE && !x) MN_FAIL(MN_EINVAL, "mn_qlinear_bwd_data: x is required for the clip-STE");
    mn_set_last_kernel("k_qlin_bwd_data");
    hipLaunchKernelGGL(k_qlin_bwd_data, dim3((unsigned)mn_grid_for(N * C, 256, 2048)), dim3(256), 0, (hipStream_t)stream, ste, gy, w, x, dx, (int)N, (int)C, (int)O);
    MN_CHECK_LAUNCH("mn_qlinear_bwd_data");
    return MN_OK;
}
extern "C" int mn_qlinear_bwd_weight(const mn_actq* aq, const float* gy, const float* x, float* dw, float* dbias, int64_t N, int64_t C, int64_t O, mn_stream_t stream) {
    if (!mn_qlinear_supported(N, C, O) || !gy || !x || !dw) MN_FAIL(MN_EINVAL, "mn_qlinear_bwd_weight: bad arguments");
    Pro pro;
    int rc = make_pro(aq, &pro, 0, "mn_qlinear_bwd_weight");
    if (rc) return rc;
    if (pro.mode == MN_ACTQ_SIGN8 || pro.mode == MN_ACTQ_CODE8) MN_FAIL(MN_ENOTSUP, "mn_qlinear_bwd_weight: fp32 activations only");
    mn_set_last_kernel("k_qlin_bwd_weight");
    hipLaunchKernelGGL(k_qlin_bwd_weight, dim3((unsigned)((C + QL_CB - 1) / QL_CB)), dim3(256), 0, (hipStream_t)stream, pro, gy, x, dw, dbias, (int)N, (int)C, (int)O);
    MN_CHECK_LAUNCH("mn_qlinear_bwd_weight");
    return MN_OK;
}
