// BatchNorm2d followed by the binary activation, fused, for gfx950.
//
// In the reference's W/A-binary nets every quantised conv is followed by `relu(bn(.))` with the ReLU replaced by
// BinaryActivation (models/nin_gc.py:53-59, wbwtab/quantize.py:79-94, 319-322).  Run as separate modules that is
// BN (3 tensor passes) + sign (2 passes) forward and sign-backward (3) + BN-backward (5) backward, all HBM-bound on
// the largest tensors of the net.  Fused, the normalised value z never exists in memory:
//   forward : pass 1 per-channel sum / sum of squares of y (fp64 partials, pivoted) -> mean, invstd, running stats
//             pass 2 a = sign((y - mean) * invstd * gamma + beta), 0 -> +1                 (read y, write a)
//   backward: pass 1 z recomputed from y; dz = da * [|z| < 1]; per-channel sum dz, sum dz*zhat  (read da, y)
//             pass 2 dy = gamma * invstd * (dz - sum_dz/n - zhat * sum_dzzhat/n)                (read da, y, write dy)
// i.e. 3 + 5 passes instead of 5 + 8.  All kernels address the tensor as C channels x (N planes of HW contiguous floats),
// float4 per lane, every lane busy whatever HW is.
#include "qgemm_dev.h"

// streaming loops: iterations are independent; MN_STREAM_2 handles two quads per trip -- both sets of loads are issued before the
// first is consumed (twice the bytes in flight per thread; same per-thread accumulation order)
#define MN_STREAM_2(i_, start_, stride_, n_, LOAD, FIN)                                  \
    {                                                                                    \
        int64_t i_ = (start_);                                                           \
        for (; i_ + (stride_) < (n_); i_ += 2 * (stride_)) {                             \
            LOAD(0, i_) LOAD(1, i_ + (stride_)) FIN(0) FIN(1)                            \
        }                                                                                \
        if (i_ < (n_)) { LOAD(0, i_) FIN(0) }                                            \
    }

#define BNS_SPLIT 32

struct BnsGeom {
    int N, C, HW, HW4;      // HW4 = HW / 4
    FastDiv fd_hw4;
    int64_t n4;             // float4 per channel = N * HW4
    int act;                // 0: BinaryActivation (sign; backward mask |z| < 1)   1: ReLU (max(z, 0); backward mask z > 0)   2: none (plain BatchNorm2d)
};
static BnsGeom bns_geom(int64_t N, int64_t C, int64_t HW) {
    BnsGeom g;
    g.N = (int)N; g.C = (int)C; g.HW = (int)HW; g.HW4 = (int)(HW / 4); g.fd_hw4 = make_fastdiv((uint32_t)g.HW4); g.n4 = N * (HW / 4); g.act = 0;
    return g;
}
// float4 index i of channel c -> element offset
__device__ __forceinline__ int64_t bns_off(const BnsGeom& g, int c, uint32_t i) {
    const uint32_t n = fd_div(i, g.fd_hw4);
    const uint32_t q = i - n * (uint32_t)g.HW4;
    return ((int64_t)n * g.C + c) * g.HW + (int64_t)q * 4;
}
__device__ __forceinline__ float bns_sign(float z) { return (z < 0.f) ? -1.f : ((z != z) ? z : 1.f); }   // 0, -0 -> +1; NaN stays
__device__ __forceinline__ float bns_relu(float z) { return (z > 0.f) ? z : ((z != z) ? z : 0.f); }       // torch.relu: NaN stays

// MODE 0: s1 = sum(y - pivot), s2 = sum((y - pivot)^2).   MODE 1: s1 = sum dz, s2 = sum dz * zhat.
// QB (MODE 1, plain BatchNorm): `da` is the gradient of the OUTPUT of the QuantAdd [+ ReLU] this BatchNorm feeds (wqaq/iao/quantize.py:1484-1498); the gradient of the
// BatchNorm's own output is formed on the way -- qb.bits: one bit per element = (ReLU passes) and (the shared quantizer's clip-STE passes this input), written by the
// fused forward (k_qadd_bn_fwd); a passing element carries (g * sc) / sc, iao_fq_grad's value.  Same thread -> element map and summation order as without QB.
struct BnsQB { const unsigned char* bits; const unsigned char* bits_sc; float* dsc; const float* qp; };
__device__ __forceinline__ uint32_t bns_qb_nibble(const unsigned char* __restrict__ bits, int64_t off) { return ((uint32_t)bits[off >> 3] >> (uint32_t)(off & 4)) & 15u; }
template <int MODE, int QB = 0>
__global__ __launch_bounds__(256) void k_bns_partial(const BnsGeom g, const float* __restrict__ y, const float* __restrict__ da,
                                                     const float* __restrict__ save, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, double* __restrict__ part, const BnsQB qb) {
    __shared__ double scd[16];
    const int c = blockIdx.x, sp = blockIdx.y, S = gridDim.y;
    float pivot = 0.f, mean = 0.f, invstd = 0.f, ga = 0.f, be = 0.f;
    if (MODE == 0) pivot = y[(int64_t)c * g.HW];
    else { mean = save[c]; invstd = save[g.C + c]; ga = gamma[c]; be = beta[c]; }
    float q_sc = 1.f, q_inv = 1.f;
    if (QB) { q_sc = qb.qp[0]; q_inv = 1.0f / q_sc; }
    double s1 = 0.0, s2 = 0.0;
    float4 v_[2], gg_[2];
    uint32_t nb_[2];
#define BNSP_LOAD(k, idx) { const int64_t off = bns_off(g, c, (uint32_t)(idx)); v_[k] = *reinterpret_cast<const float4*>(y + off); \
                            if (MODE == 1) gg_[k] = *reinterpret_cast<const float4*>(da + off);                                    \
                            if (QB) nb_[k] = bns_qb_nibble(qb.bits, off); }
#define BNSP_FIN(k) { const float4 v = v_[k];                                                                                         \
        if (MODE == 0) {                                                                                                              \
            const float a = v.x - pivot, b = v.y - pivot, cc = v.z - pivot, d = v.w - pivot;                                          \
            s1 += (double)((a + b) + (cc + d));                                                                                       \
            s2 += (double)((a * a + b * b) + (cc * cc + d * d));                                                                      \
        } else {                                                                                                                      \
            const float4 gg = gg_[k];                                                                                                 \
            const float zh[4] = {(v.x - mean) * invstd, (v.y - mean) * invstd, (v.z - mean) * invstd, (v.w - mean) * invstd};         \
            float gv[4] = {gg.x, gg.y, gg.z, gg.w};                                                                                   \
            if (QB) { _Pragma("unroll") for (int e = 0; e < 4; ++e) gv[e] = ((nb_[k] >> e) & 1u) ? mn_div_m(gv[e] * q_sc, q_sc, q_inv) : 0.f; } \
            float t1 = 0.f, t2 = 0.f;                                                                                                 \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                                           \
                const float z = zh[e] * ga + be;                                                                                      \
                const float dz = (g.act == 2 || (g.act ? (z > 0.f) : (z > -1.f && z < 1.f))) ? gv[e] : 0.f;   /* clip-STE of the sign / ReLU mask */   \
                t1 += dz;                                                                                                             \
                t2 += dz * zh[e];                                                                                                     \
            }                                                                                                                         \
            s1 += (double)t1;                                                                                                         \
            s2 += (double)t2;                                                                                                         \
        } }
    MN_STREAM_2(i, (int64_t)sp * 256 + threadIdx.x, (int64_t)S * 256, g.n4, BNSP_LOAD, BNSP_FIN)
#undef BNSP_LOAD
#undef BNSP_FIN
    s1 = block_reduce(s1, OpAddD(), 0.0, scd);
    s2 = block_reduce(s2, OpAddD(), 0.0, scd);
    if (threadIdx.x == 0) { part[((int64_t)c * S + sp) * 2] = s1; part[((int64_t)c * S + sp) * 2 + 1] = s2; }
}
// forward: mean / invstd (+ running statistics, momentum update with the UNBIASED variance like nn.BatchNorm2d)
__global__ void k_bns_final_fwd(const BnsGeom g, const float* __restrict__ y, const double* __restrict__ part, int S, float eps, float momentum,
                                float* __restrict__ running_mean, float* __restrict__ running_var, float* __restrict__ save) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= g.C) return;
    double sv[2] = {0.0, 0.0};
    mn_row_sums<2>(part + (int64_t)c * S * 2, S, sv);
    const double s1 = sv[0], s2 = sv[1];
    const double n = (double)g.N * (double)g.HW;
    const double m = s1 / n;
    const double mean = (double)y[(int64_t)c * g.HW] + m;
    const double ss = s2 - s1 * m;                 // sum of squared deviations
    const float var_b = (float)(ss / n);
    save[c] = (float)mean;
    save[g.C + c] = 1.0f / sqrtf(var_b + eps);
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(ss / (n - 1.0));
}
__global__ void k_bns_final_bwd(const BnsGeom g, const double* __restrict__ part, int S, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                float* __restrict__ sums) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= g.C) return;
    double sv[2] = {0.0, 0.0};
    mn_row_sums<2>(part + (int64_t)c * S * 2, S, sv);
    const double s1 = sv[0], s2 = sv[1];
    if (dbeta) dbeta[c] = (float)s1;
    if (dgamma) dgamma[c] = (float)s2;
    sums[c] = (float)s1; sums[g.C + c] = (float)s2;
}
// eval mode: mean / invstd from the running statistics
__global__ void k_bns_eval_stats(int C, float eps, const float* __restrict__ running_mean, const float* __restrict__ running_var, float* __restrict__ save) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    save[c] = running_mean[c];
    save[C + c] = 1.0f / sqrtf(running_var[c] + eps);
}
// MODE 0: a = sign(bn(y)).   MODE 1: dy (training: full BN backward; eval: statistics are constants)
// OUT8 (MODE 0 only): a is written as int8 sign codes (0x01 / 0xFF), 4 per lane and store
// The "final" step of the partial sums folded into the apply pass (fin.part != null): the S partials of channel c (written by the k_bns_partial launch in front)
// are summed by EVERY block of that channel in the fixed order of k_bns_final_fwd / _bwd -- so all of them work with bit-identical statistics -- and block
// sp == 0 writes what the final kernel used to write (save + running statistics; dgamma, dbeta, sums).  One launch less per BatchNorm and direction: 40 of a
// resnet18 IAO step's ~300.
struct BnsFin {
    const double* part;            // [C][S][2] or null (then `save` / `sums` hold the finished values)
    int S;
    float eps, momentum;           // MODE 0
    float* running_mean;           // MODE 0, nullable
    float* running_var;
    float* save_out;               // MODE 0: [2][C]
    float* dgamma;                 // MODE 1, nullable
    float* dbeta;
    float* sums_out;               // MODE 1: [2][C]
    // kind 1 (MODE 0): part = [S][C][2] EXACT sums of the integer accumulator of the dense IAO conv that produced y = al[c] * acc + cb[c] (mn_actq.stats): mean and
    // variance of y in fp64 from them -- the arithmetic of k_qa_stats_prep -- and no statistics pass over y at all (mn_bn_fwd_acc)
    int kind;
    const float* sa;               // al[c] = sa[0] * sw[c * sw_stride], the fp32 product of the conv's epilogue
    const float* sw;
    int sw_stride;
    const float* cbias;            // nullable
};
template <int MODE, int OUT8 = 0, int QB = 0>
__global__ __launch_bounds__(256) void k_bns_apply(const BnsGeom g, const float* __restrict__ y, const float* __restrict__ da,
                                                   const float* __restrict__ save, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   const float* __restrict__ sums, int training, float* __restrict__ out, float* __restrict__ mm, const BnsFin fin,
                                                   const BnsQB qb) {
    // QB (MODE 1): as k_bns_partial<1, 1>; qb.dsc (nullable): the gradient of the QuantAdd's OTHER input (an identity shortcut) from the same read of the output
    // gradient -- (g * sc) / sc where qb.bits_sc passes
    // mm (forward, fp32 output only; may be null): per-block min / max of the values written -> mm[block], mm[nblocks + block]: the NEXT layer's IAO observer
    // (wqaq/iao/quantize.py:23-36) reduces these instead of reading the activation again (mn_iao_observe_partials)
    const int c = blockIdx.x, sp = blockIdx.y, S = gridDim.y;
    float mean, invstd;
    const float ga = gamma[c], be = beta[c];
    float mlo = INFINITY, mhi = -INFINITY;
    float k1 = 0.f, k2 = 0.f;
    if (fin.part) {
        __shared__ double fsum[2];
        if (MODE == 0 && fin.kind == 1) {          // partial rows [S][C][2]: up to 512 of them -- one wave sums them (fixed lane order), then a fixed-order tree
            if (threadIdx.x < 64) {
                double a1 = 0.0, a2 = 0.0;
                for (int i = threadIdx.x; i < fin.S; i += 64) { a1 += fin.part[((int64_t)i * g.C + c) * 2]; a2 += fin.part[((int64_t)i * g.C + c) * 2 + 1]; }
                a1 = wave_reduce(a1, OpAddD()); a2 = wave_reduce(a2, OpAddD());
                if (threadIdx.x == 0) { fsum[0] = a1; fsum[1] = a2; }
            }
        } else if (threadIdx.x == 0) {          // (block-uniform branch; S <= BNS_SPLIT sequential adds: the order of the former final kernels)
            double s1 = 0.0, s2 = 0.0;
            for (int i = 0; i < fin.S; ++i) { s1 += fin.part[((int64_t)c * fin.S + i) * 2]; s2 += fin.part[((int64_t)c * fin.S + i) * 2 + 1]; }
            fsum[0] = s1; fsum[1] = s2;
        }
        __syncthreads();
        const double s1 = fsum[0], s2 = fsum[1];
        if (MODE == 0) {
            const double n = (double)g.N * (double)g.HW;
            double m, mean_d, ss;
            if (fin.kind == 1) {
                const double al = (double)(fin.sa[0] * fin.sw[(int64_t)c * fin.sw_stride]);
                m = s1 / n;
                mean_d = al * m + (double)(fin.cbias ? fin.cbias[c] : 0.f);
                ss = al * al * (s2 - s1 * m);
                if (ss < 0.0) ss = 0.0;
            } else {
                m = s1 / n;
                mean_d = (double)y[(int64_t)c * g.HW] + m;
                ss = s2 - s1 * m;                 // sum of squared deviations
            }
            const float var_b = (float)(ss / n);
            mean = (float)mean_d;
            invstd = 1.0f / sqrtf(var_b + fin.eps);
            if (sp == 0 && threadIdx.x == 0) {
                fin.save_out[c] = mean;
                fin.save_out[g.C + c] = invstd;
                if (fin.running_mean) fin.running_mean[c] = (1.f - fin.momentum) * fin.running_mean[c] + fin.momentum * (float)mean_d;
                if (fin.running_var) fin.running_var[c] = (1.f - fin.momentum) * fin.running_var[c] + fin.momentum * (float)(ss / (n - 1.0));
            }
        } else {
            mean = save[c]; invstd = save[g.C + c];
            if (training) {
                const float n = (float)g.N * (float)g.HW;
                k1 = (float)s1 / n; k2 = (float)s2 / n;
            }
            if (sp == 0 && threadIdx.x == 0) {
                if (fin.dbeta) fin.dbeta[c] = (float)s1;
                if (fin.dgamma) fin.dgamma[c] = (float)s2;
                fin.sums_out[c] = (float)s1; fin.sums_out[g.C + c] = (float)s2;
            }
        }
    } else {
        mean = save[c]; invstd = save[g.C + c];
        if (MODE == 1 && training) {
            const float n = (float)g.N * (float)g.HW;
            k1 = sums[c] / n; k2 = sums[g.C + c] / n;
        }
    }
    const float gi = ga * invstd;
    float q_sc = 1.f, q_inv = 1.f;
    if (QB) { q_sc = qb.qp[0]; q_inv = 1.0f / q_sc; }
    float4 v_[2], gg_[2];
    int64_t off_[2];
    uint32_t nb_[2], ns_[2];
#define BNSA_LOAD(k, idx) { off_[k] = bns_off(g, c, (uint32_t)(idx)); v_[k] = *reinterpret_cast<const float4*>(y + off_[k]); \
                            if (MODE == 1) gg_[k] = *reinterpret_cast<const float4*>(da + off_[k]);                          \
                            if (QB) { nb_[k] = bns_qb_nibble(qb.bits, off_[k]); if (qb.dsc) ns_[k] = bns_qb_nibble(qb.bits_sc, off_[k]); } }
#define BNSA_FIN(k) { const float4 v = v_[k]; const int64_t off = off_[k];                                                            \
        const float zh[4] = {(v.x - mean) * invstd, (v.y - mean) * invstd, (v.z - mean) * invstd, (v.w - mean) * invstd};             \
        float r[4];                                                                                                                   \
        if (MODE == 0) {                                                                                                              \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) { const float z = zh[e] * ga + be; r[e] = g.act == 2 ? z : (g.act ? bns_relu(z) : bns_sign(z)); } \
        } else {                                                                                                                      \
            const float4 gg = gg_[k];                                                                                                 \
            float gv[4] = {gg.x, gg.y, gg.z, gg.w};                                                                                   \
            if (QB) {                                                                                                                 \
                float t[4];                                                                                                           \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) { t[e] = mn_div_m(gv[e] * q_sc, q_sc, q_inv); gv[e] = ((nb_[k] >> e) & 1u) ? t[e] : 0.f; } \
                if (qb.dsc) *reinterpret_cast<float4*>(qb.dsc + off) = make_float4((ns_[k] & 1u) ? t[0] : 0.f, (ns_[k] & 2u) ? t[1] : 0.f, (ns_[k] & 4u) ? t[2] : 0.f, \
                                                                                  (ns_[k] & 8u) ? t[3] : 0.f);                      \
            }                                                                                                                         \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                                           \
                const float z = zh[e] * ga + be;                                                                                      \
                const float dz = (g.act == 2 || (g.act ? (z > 0.f) : (z > -1.f && z < 1.f))) ? gv[e] : 0.f;                                           \
                r[e] = gi * (dz - k1 - zh[e] * k2);                                                                                   \
            }                                                                                                                         \
        }                                                                                                                             \
        if (OUT8) {                                                                                                                   \
            const uint32_t u = (r[0] < 0.f ? 0xFFu : 0x01u) | (r[1] < 0.f ? 0xFF00u : 0x0100u) | (r[2] < 0.f ? 0xFF0000u : 0x010000u) | \
                               (r[3] < 0.f ? 0xFF000000u : 0x01000000u);                                                              \
            *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(out) + off) = u;                                                     \
        } else {                                                                                                                      \
            *reinterpret_cast<float4*>(out + off) = make_float4(r[0], r[1], r[2], r[3]);                                              \
            if (MODE == 0 && mm) {                                                                                                    \
                mlo = OpMinF()(OpMinF()(mlo, r[0]), OpMinF()(OpMinF()(r[1], r[2]), r[3]));                                            \
                mhi = OpMaxF()(OpMaxF()(mhi, r[0]), OpMaxF()(OpMaxF()(r[1], r[2]), r[3]));                                            \
            }                                                                                                                         \
        } }
    MN_STREAM_2(i, (int64_t)sp * 256 + threadIdx.x, (int64_t)S * 256, g.n4, BNSA_LOAD, BNSA_FIN)
#undef BNSA_LOAD
#undef BNSA_FIN
    if (MODE == 0 && !OUT8 && mm) {
        __shared__ float scm[16];
        mlo = block_reduce(mlo, OpMinF(), INFINITY, scm);
        mhi = block_reduce(mhi, OpMaxF(), -INFINITY, scm);
        if (threadIdx.x == 0) { const int b = sp * (int)gridDim.x + c, nb = (int)(gridDim.x * gridDim.y); mm[b] = mlo; mm[nb + b] = mhi; }
    }
}

// ---------------------------------------------------------------- 2x2 / stride-2 max-pool on int8 sign codes
// max over +-1 = +1 unless all four are -1.  Thread = 4 output pixels of one output row (8 input bytes of two rows).
__global__ __launch_bounds__(256) void k_pool2_sign8_fwd(const char* __restrict__ a, char* __restrict__ out, int64_t nq, int H, int W) {
    const int Wo = W >> 1, Ho = H >> 1, q4 = Wo >> 2;      // q4 quads per output row
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / q4;                         // (plane, out row)
        const int qc = (int)(i - row * q4);
        const int64_t pl = row / Ho;
        const int orow = (int)(row - pl * Ho);
        const char* src = a + (pl * H + 2 * orow) * W + qc * 8;
        const uint64_t r0 = *reinterpret_cast<const uint64_t*>(src), r1 = *reinterpret_cast<const uint64_t*>(src + W);
        const uint64_t vv = r0 & r1 & 0x8080808080808080ull;                                // vertical AND of the sign bits
        const uint32_t v0 = (uint32_t)vv, v1 = (uint32_t)(vv >> 32);
        const uint32_t n0 = v0 & (v0 >> 8), n1 = v1 & (v1 >> 8);                            // horizontal: bytes (0,1) -> bit 7, (2,3) -> bit 23
        const uint32_t o = ((n0 & 0x80u) ? 0xFFu : 0x01u) | ((n0 & 0x800000u) ? 0xFF00u : 0x0100u) |
                           ((n1 & 0x80u) ? 0xFF0000u : 0x010000u) | ((n1 & 0x800000u) ? 0xFF000000u : 0x01000000u);
        *reinterpret_cast<uint32_t*>(out + (pl * Ho + orow) * Wo + qc * 4) = o;
    }
}
// backward: the gradient of an output pixel goes to the FIRST maximum of its window in row-major order (ATen's
// max_pool2d picks `val > max`): the first +1, or the first element when all four are -1.  Thread = 2 output pixels.
__global__ __launch_bounds__(256) void k_pool2_sign8_bwd(const float* __restrict__ dout, const char* __restrict__ a, float* __restrict__ din,
                                                         int64_t np, int H, int W) {
    const int Wo = W >> 1, Ho = H >> 1, p2 = Wo >> 1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < np; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / p2;
        const int pc = (int)(i - row * p2);
        const int64_t pl = row / Ho;
        const int orow = (int)(row - pl * Ho);
        const float2 g = *reinterpret_cast<const float2*>(dout + (pl * Ho + orow) * Wo + pc * 2);
        const int64_t ioff = (pl * H + 2 * orow) * W + pc * 4;
        const uint32_t r0 = *reinterpret_cast<const uint32_t*>(a + ioff), r1 = *reinterpret_cast<const uint32_t*>(a + ioff + W);
        float t[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const bool p00 = !((r0 >> (16 * e)) & 0x80u), p01 = !((r0 >> (16 * e + 8)) & 0x80u);
            const bool p10 = !((r1 >> (16 * e)) & 0x80u), p11 = !((r1 >> (16 * e + 8)) & 0x80u);
            const float gv = e ? g.y : g.x;
            const int win = p00 ? 0 : (p01 ? 1 : (p10 ? 2 : (p11 ? 3 : 0)));
            t[2 * e] = win == 0 ? gv : 0.f; t[2 * e + 1] = win == 1 ? gv : 0.f;
            b[2 * e] = win == 2 ? gv : 0.f; b[2 * e + 1] = win == 3 ? gv : 0.f;
        }
        *reinterpret_cast<float4*>(din + ioff) = make_float4(t[0], t[1], t[2], t[3]);
        *reinterpret_cast<float4*>(din + ioff + W) = make_float4(b[0], b[1], b[2], b[3]);
    }
}
extern "C" int mn_maxpool2x2_sign8_fwd(const int8_t* a, int64_t planes, int64_t H, int64_t W, int8_t* out, mn_stream_t stream) {
    if (!a || !out || planes <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 7) || (((uintptr_t)a) & 7) || (((uintptr_t)out) & 3))
        MN_FAIL(MN_EINVAL, "mn_maxpool2x2_sign8_fwd: needs even H, W %% 8 == 0, 8-byte aligned input");
    const int64_t nq = planes * (H / 2) * (W / 8);
    mn_set_last_kernel("k_pool2_sign8_fwd"); mn_prof_bytes(1.25 * (double)planes * H * W); mn_prof_begin((hipStream_t)stream);
    hipLaunchKernelGGL(k_pool2_sign8_fwd, dim3(mn_grid_for(nq, 256, 4096)), dim3(256), 0, (hipStream_t)stream, (const char*)a, (char*)out, nq, (int)H, (int)W);
    mn_prof_end((hipStream_t)stream);
    MN_CHECK_LAUNCH("mn_maxpool2x2_sign8_fwd");
    return MN_OK;
}
extern "C" int mn_maxpool2x2_sign8_bwd(const float* dout, const int8_t* a, int64_t planes, int64_t H, int64_t W, float* din, mn_stream_t stream) {
    if (!dout || !a || !din || planes <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 3) || (((uintptr_t)a) & 3) || !aligned16(din) || (((uintptr_t)dout) & 7))
        MN_FAIL(MN_EINVAL, "mn_maxpool2x2_sign8_bwd: needs even H, W %% 4 == 0, aligned tensors");
    const int64_t np = planes * (H / 2) * (W / 4);
    mn_set_last_kernel("k_pool2_sign8_bwd"); mn_prof_bytes(6.0 * (double)planes * H * W); mn_prof_begin((hipStream_t)stream);
    hipLaunchKernelGGL(k_pool2_sign8_bwd, dim3(mn_grid_for(np, 256, 8192)), dim3(256), 0, (hipStream_t)stream, dout, (const char*)a, din, np, (int)H, (int)W);
    mn_prof_end((hipStream_t)stream);
    MN_CHECK_LAUNCH("mn_maxpool2x2_sign8_bwd");
    return MN_OK;
}

// ---------------------------------------------------------------- 2x2 / stride-2 max-pool on fp32 activations (DoReFa / IAO nets)
// Forward keeps the argmax of every window as one byte (0..3, row-major in the window) -- ATen's rule: a later element replaces the
// maximum if it is greater OR NaN -- so the backward is a pure scatter of 4 B + 1 B per window into 16 B (ATen's backward kernel reads
// int64 indices and is several times slower).  Thread = 4 consecutive windows of one output row.
__global__ __launch_bounds__(256) void k_pool2_f32_fwd(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ idx, int64_t nq, int H, int W) {
    const int Wo = W >> 1, Ho = H >> 1, q4 = Wo >> 2;
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
            const float v[4] = {r0[2 * w], r0[2 * w + 1], r1[2 * w], r1[2 * w + 1]};
            float m = -INFINITY;
            uint32_t k = 0u;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (v[e] > m || v[e] != v[e]) { m = v[e]; k = (uint32_t)e; }
            o[w] = m; ib |= k << (8 * w);
        }
        const int64_t oo = (pl * Ho + orow) * Wo + qc * 4;
        *reinterpret_cast<float4*>(y + oo) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<uint32_t*>(idx + oo) = ib;
    }
}
__global__ __launch_bounds__(256) void k_pool2_f32_bwd(const float* __restrict__ gy, const unsigned char* __restrict__ idx, float* __restrict__ dx, int64_t nq, int H, int W) {
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
        float r0[8], r1[8];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t k = (ib >> (8 * w)) & 3u;
            r0[2 * w] = k == 0u ? g[w] : 0.f; r0[2 * w + 1] = k == 1u ? g[w] : 0.f;
            r1[2 * w] = k == 2u ? g[w] : 0.f; r1[2 * w + 1] = k == 3u ? g[w] : 0.f;
        }
        float* dst = dx + (pl * H + 2 * orow) * W + qc * 8;
        *reinterpret_cast<float4*>(dst) = make_float4(r0[0], r0[1], r0[2], r0[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(r0[4], r0[5], r0[6], r0[7]);
        *reinterpret_cast<float4*>(dst + W) = make_float4(r1[0], r1[1], r1[2], r1[3]);
        *reinterpret_cast<float4*>(dst + W + 4) = make_float4(r1[4], r1[5], r1[6], r1[7]);
    }
}
extern "C" int mn_maxpool2x2_f32_supported(int64_t H, int64_t W) { return H >= 2 && H % 2 == 0 && W >= 8 && W % 8 == 0; }
extern "C" int mn_maxpool2x2_f32_fwd(const float* x, int64_t planes, int64_t H, int64_t W, float* y, uint8_t* idx, mn_stream_t stream) {
    if (!x || !y || !idx || planes <= 0 || !mn_maxpool2x2_f32_supported(H, W) || !aligned16(x) || !aligned16(y) || (((uintptr_t)idx) & 3))
        MN_FAIL(MN_EINVAL, "mn_maxpool2x2_f32_fwd: needs even H, W %% 8 == 0, aligned tensors");
    const int64_t nq = planes * (H / 2) * (W / 8);
    mn_set_last_kernel("k_pool2_f32_fwd"); mn_prof_bytes(5.25 * (double)planes * H * W); mn_prof_begin((hipStream_t)stream);
    hipLaunchKernelGGL(k_pool2_f32_fwd, dim3(mn_grid_for(nq, 256, 8192)), dim3(256), 0, (hipStream_t)stream, x, y, (unsigned char*)idx, nq, (int)H, (int)W);
    mn_prof_end((hipStream_t)stream);
    MN_CHECK_LAUNCH("mn_maxpool2x2_f32_fwd");
    return MN_OK;
}
extern "C" int mn_maxpool2x2_f32_bwd(const float* gy, const uint8_t* idx, int64_t planes, int64_t H, int64_t W, float* dx, mn_stream_t stream) {
    if (!gy || !dx || !idx || planes <= 0 || !mn_maxpool2x2_f32_supported(H, W) || !aligned16(gy) || !aligned16(dx) || (((uintptr_t)idx) & 3))
        MN_FAIL(MN_EINVAL, "mn_maxpool2x2_f32_bwd: needs even H, W %% 8 == 0, aligned tensors");
    const int64_t nq = planes * (H / 2) * (W / 8);
    mn_set_last_kernel("k_pool2_f32_bwd"); mn_prof_bytes(5.25 * (double)planes * H * W); mn_prof_begin((hipStream_t)stream);
    hipLaunchKernelGGL(k_pool2_f32_bwd, dim3(mn_grid_for(nq, 256, 8192)), dim3(256), 0, (hipStream_t)stream, gy, (const unsigned char*)idx, dx, nq, (int)H, (int)W);
    mn_prof_end((hipStream_t)stream);
    MN_CHECK_LAUNCH("mn_maxpool2x2_f32_bwd");
    return MN_OK;
}

// nn.AvgPool2d whose window is the whole image (the tail of the reference's nin / nin_gc: AvgPool2d(8) on 8 x 8 maps, models/nin_gc.py:139): one wave per
// plane, lanes stride over the pixels, fixed-order wave reduction -- ATen's generic avg_pool2d kernel takes 32 us for this 160 KB tensor.
__global__ __launch_bounds__(256) void k_gap_fwd(const float* __restrict__ x, float* __restrict__ y, int64_t planes, int HW, float inv) {
    const int lane = threadIdx.x & 63;
    const int64_t pl = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    float s = 0.f;
    if (pl < planes) for (int i = lane; i < HW; i += 64) s += x[pl * HW + i];
    s = wave_reduce(s, OpAddF());
    if (pl < planes && lane == 0) y[pl] = s / inv;          // (inv holds the window size: ATen divides)
}
__global__ __launch_bounds__(256) void k_gap_bwd(const float* __restrict__ gy, float* __restrict__ dx, int64_t n, int HW, float inv) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dx[i] = gy[i / HW] / inv;
}
extern "C" int mn_avgpool_global_fwd(const float* x, int64_t planes, int64_t HW, float* y, mn_stream_t stream) {
    if (!x || !y || planes <= 0 || HW <= 0 || HW > (1 << 20)) MN_FAIL(MN_EINVAL, "mn_avgpool_global_fwd: bad arguments");
    mn_set_last_kernel("k_gap_fwd");
    hipLaunchKernelGGL(k_gap_fwd, dim3((unsigned)((planes + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, y, planes, (int)HW, (float)HW);
    MN_CHECK_LAUNCH("mn_avgpool_global_fwd");
    return MN_OK;
}
extern "C" int mn_avgpool_global_bwd(const float* gy, int64_t planes, int64_t HW, float* dx, mn_stream_t stream) {
    if (!gy || !dx || planes <= 0 || HW <= 0 || HW > (1 << 20)) MN_FAIL(MN_EINVAL, "mn_avgpool_global_bwd: bad arguments");
    const int64_t n = planes * HW;
    mn_set_last_kernel("k_gap_bwd");
    hipLaunchKernelGGL(k_gap_bwd, dim3(mn_grid_for(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, gy, dx, n, (int)HW, (float)HW);
    MN_CHECK_LAUNCH("mn_avgpool_global_bwd");
    return MN_OK;
}

extern "C" int64_t mn_bnsign_ws_floats(int64_t C) { return C * BNS_SPLIT * 4 + 2 * C + 16; }   // fp64 partials + {sum dz, sum dz*zhat}

static int bns_check(int64_t N, int64_t C, int64_t HW, const void* a, const void* b, const char* what) {
    if (N <= 0 || C <= 0 || HW <= 0 || HW % 4 || N * (HW / 4) >= ((int64_t)1 << 31)) MN_FAIL(MN_EINVAL, "%s: bad shape (HW must be a multiple of 4)", what);
    if (!aligned16(a) || !aligned16(b)) MN_FAIL(MN_EINVAL, "%s: tensors must be 16-byte aligned", what);
    return MN_OK;
}
static int bns_split(const BnsGeom& g) {
    // enough workgroups to fill 256 CUs several times, at least one 256-thread sweep per workgroup
    int64_t S = (2048 + g.C - 1) / g.C;
    const int64_t maxS = (g.n4 + 255) / 256;
    if (S > maxS) S = maxS;
    if (S > BNS_SPLIT) S = BNS_SPLIT;
    if (S < 1) S = 1;
    return (int)S;
}

static int bnsign_fwd_impl(const float* y, int64_t N, int64_t C, int64_t HW, const float* gamma, const float* beta, float eps, float momentum,
                           int training, float* running_mean, float* running_var, float* save, float* a, int out8, float* ws, mn_stream_t stream,
                           int act = 0, float* mm = nullptr, bool stats_given = false) {
    int rc = bns_check(N, C, HW, y, out8 ? (const void*)y : (const void*)a, "mn_bnsign_fwd");
    if (!rc && out8 && (((uintptr_t)a) & 3)) MN_FAIL(MN_EINVAL, "mn_bnsign_fwd_i8: output must be 4-byte aligned");
    if (rc) return rc;
    if (!y || !gamma || !beta || !save || !a || (!stats_given && (!ws || (((uintptr_t)ws) & 7)))) MN_FAIL(MN_EINVAL, "mn_bnsign_fwd: null / misaligned argument");
    if (!stats_given && !training && (!running_mean || !running_var)) MN_FAIL(MN_EINVAL, "mn_bnsign_fwd: eval mode needs the running statistics");
    hipStream_t s = (hipStream_t)stream;
    BnsGeom g = bns_geom(N, C, HW);
    g.act = act;
    const int S = bns_split(g);
    const double nel = (double)N * C * HW;
    BnsFin fin = {};
    if (stats_given) {
        // save = {mean, invstd} is the caller's (mn_conv2d_first_gram_bnstats): the apply pass alone
    } else if (training) {
        mn_set_last_kernel("k_bns_partial<0>"); mn_prof_bytes(4.0 * nel); mn_prof_begin(s);
        hipLaunchKernelGGL(k_bns_partial<0>, dim3((unsigned)C, (unsigned)S), dim3(256), 0, s, g, y, (const float*)nullptr, (const float*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, (double*)ws, BnsQB());
        mn_prof_end(s);
        fin.part = (const double*)ws; fin.S = S; fin.eps = eps; fin.momentum = momentum; fin.running_mean = running_mean; fin.running_var = running_var; fin.save_out = save;
    } else {
        hipLaunchKernelGGL(k_bns_eval_stats, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, s, (int)C, eps, (const float*)running_mean, (const float*)running_var, save);
    }
    mn_set_last_kernel(out8 ? "k_bns_apply<0, 1>" : "k_bns_apply<0, 0>"); mn_prof_bytes((out8 ? 5.0 : 8.0) * nel); mn_prof_begin(s);
    if (out8) hipLaunchKernelGGL((k_bns_apply<0, 1>), dim3((unsigned)C, (unsigned)S), dim3(256), 0, s, g, y, (const float*)nullptr, (const float*)save, gamma, beta,
                                 (const float*)nullptr, training, a, (float*)nullptr, fin, BnsQB());
    else hipLaunchKernelGGL((k_bns_apply<0, 0>), dim3((unsigned)C, (unsigned)S), dim3(256), 0, s, g, y, (const float*)nullptr, (const float*)save, gamma, beta,
                            (const float*)nullptr, training, a, mm, fin, BnsQB());
    mn_prof_end(s);
    MN_CHECK_LAUNCH("mn_bnsign_fwd");
    return MN_OK;
}
// statistics only: save = {mean, invstd} of a BatchNorm over fp32 y (training: batch statistics + running update; eval: the running ones) -- the
// first half of mn_bnrelu_fwd for a consumer that normalises itself (mn_qa_fwd with in_f32 = 1)
extern "C" int mn_bn_save_stats(const float* y, int64_t N, int64_t C, int64_t HW, float eps, float momentum, int training, float* running_mean,
                                float* running_var, float* save, float* ws, mn_stream_t stream) {
    int rc = bns_check(N, C, HW, y, y, "mn_bn_save_stats");
    if (rc) return rc;
    if (!y || !save || !ws || (((uintptr_t)ws) & 7)) MN_FAIL(MN_EINVAL, "mn_bn_save_stats: null / misaligned argument");
    if (!training && (!running_mean || !running_var)) MN_FAIL(MN_EINVAL, "mn_bn_save_stats: eval mode needs the running statistics");
    hipStream_t s = (hipStream_t)stream;
    BnsGeom g = bns_geom(N, C, HW);
    const int S = bns_split(g);
    if (training) {
        mn_set_last_kernel("k_bns_partial<0>"); mn_prof_bytes(4.0 * (double)N * C * HW); mn_prof_begin(s);
        hipLaunchKernelGGL(k_bns_partial<0>, dim3((unsigned)C, (unsigned)S), dim3(256), 0, s, g, y, (const float*)nullptr, (const float*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, (double*)ws, BnsQB());
        mn_prof_end(s);
        hipLaunchKernelGGL(k_bns_final_fwd, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, s, g, y, (const double*)ws, S, eps, momentum, running_mean, running_var, save);
    } else {
        hipLaunchKernelGGL(k_bns_eval_stats, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, s, (int)C, eps, (const float*)running_mean, (const float*)running_var, save);
    }
    MN_CHECK_LAUNCH("mn_bn_save_stats");
    return MN_OK;
}
extern "C" int mn_bnsign_fwd(const float* y, int64_t N, int64_t C, int64_t HW, const float* gamma, const float* beta, float eps, float momentum,
                             int training, float* running_mean, float* running_var, float* save, float* a, float* ws, mn_stream_t stream) {
    return bnsign_fwd_impl(y, N, C, HW, gamma, beta, eps, momentum, training, running_mean, running_var, save, a, 0, ws, stream);
}
/* a = sign(bn(y)) with GIVEN statistics save = {mean, invstd} [2][C] (mn_conv2d_first_gram_bnstats): the apply pass of mn_bnsign_fwd / _i8 alone */
extern "C" int mn_bnsign_apply(const float* y, int64_t N, int64_t C, int64_t HW, const float* gamma, const float* beta, const float* save, void* a, int out8,
                               mn_stream_t stream) {
    return bnsign_fwd_impl(y, N, C, HW, gamma, beta, 0.f, 0.f, 1, nullptr, nullptr, const_cast<float*>(save), (float*)a, out8 ? 1 : 0, nullptr, stream, 0, nullptr, true);
}
extern "C" int mn_bnsign_fwd_i8(const float* y, int64_t N, int64_t C, int64_t HW, const float* gamma, const float* beta, float eps, float momentum,
                                int training, float* running_mean, float* running_var, float* save, int8_t* a, float* ws, mn_stream_t stream) {
    return bnsign_fwd_impl(y, N, C, HW, gamma, beta, eps, momentum, training, running_mean, running_var, save, (float*)a, 1, ws, stream);
}

// first half of mn_bnsign_bwd only: dgamma, dbeta and sums [2][C] = {sum dz, sum dz*zhat}; a consumer that forms dy itself
// (mn_conv2d_bwd_weight_first_bn) takes it from there
extern "C" int mn_bnsign_bwd_sums(const float* da, const float* y, const float* save, const float* gamma, const float* beta, int64_t N, int64_t C,
                                  int64_t HW, float* dgamma, float* dbeta, float* sums, float* ws, mn_stream_t stream) {
    int rc = bns_check(N, C, HW, y, da, "mn_bnsign_bwd_sums");
    if (rc) return rc;
    if (!da || !y || !save || !gamma || !beta || !sums || !ws || (((uintptr_t)ws) & 7)) MN_FAIL(MN_EINVAL, "mn_bnsign_bwd_sums: null / misaligned argument");
    hipStream_t s = (hipStream_t)stream;
    const BnsGeom g = bns_geom(N, C, HW);
    const int S = bns_split(g);
    mn_set_last_kernel("k_bns_partial<1>"); mn_prof_bytes(8.0 * (double)N * C * HW); mn_prof_begin(s);
    hipLaunchKernelGGL(k_bns_partial<1>, dim3((unsigned)C, (unsigned)S), dim3(256), 0, s, g, y, da, save, gamma, beta, (double*)ws, BnsQB());
    mn_prof_end(s);
    hipLaunchKernelGGL(k_bns_final_bwd, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, s, g, (const double*)ws, S, dgamma, dbeta, sums);
    MN_CHECK_LAUNCH("mn_bnsign_bwd_sums");
    return MN_OK;
}
static int bnsign_bwd_impl(const float* da, const float* y, const float* save, const float* gamma, const float* beta, int64_t N, int64_t C,
                           int64_t HW, int training, float* dy, float* dgamma, float* dbeta, float* ws, mn_stream_t stream, int act);
extern "C" int mn_bnsign_bwd(const float* da, const float* y, const float* save, const float* gamma, const float* beta, int64_t N, int64_t C,
                             int64_t HW, int training, float* dy, float* dgamma, float* dbeta, float* ws, mn_stream_t stream) {
    return bnsign_bwd_impl(da, y, save, gamma, beta, N, C, HW, training, dy, dgamma, dbeta, ws, stream, 0);
}
// BatchNorm2d + ReLU (the ConvBNReLU blocks of the DoReFa / IAO nets, models/nin_gc.py:53-59): the same three / five streaming passes with
// max(z, 0) instead of the sign and the ReLU mask z > 0 instead of the clip-STE mask -- torch's relu(batch_norm(y)) and its backward
extern "C" int mn_bnrelu_fwd(const float* y, int64_t N, int64_t C, int64_t HW, const float* gamma, const float* beta, float eps, float momentum,
                             int training, float* running_mean, float* running_var, float* save, float* a, float* ws, mn_stream_t stream) {
    return bnsign_fwd_impl(y, N, C, HW, gamma, beta, eps, momentum, training, running_mean, running_var, save, a, 0, ws, stream, 1);
}
extern "C" int mn_bnrelu_bwd(const float* da, const float* y, const float* save, const float* gamma, const float* beta, int64_t N, int64_t C,
                             int64_t HW, int training, float* dy, float* dgamma, float* dbeta, float* ws, mn_stream_t stream) {
    return bnsign_bwd_impl(da, y, save, gamma, beta, N, C, HW, training, dy, dgamma, dbeta, ws, stream, 1);
}
// the same + per-block min / max of the output for the next layer's IAO observer: mm holds 2 * mn_bnrelu_mm_count(N, C, HW) floats
extern "C" int64_t mn_bnrelu_mm_count(int64_t N, int64_t C, int64_t HW) {
    if (N <= 0 || C <= 0 || HW <= 0 || HW % 4) return 0;
    BnsGeom g = bns_geom(N, C, HW);
    return C * bns_split(g);
}
extern "C" int mn_bnrelu_fwd_mm(const float* y, int64_t N, int64_t C, int64_t HW, const float* gamma, const float* beta, float eps, float momentum,
                                int training, float* running_mean, float* running_var, float* save, float* a, float* ws, float* mm, mn_stream_t stream) {
    return bnsign_fwd_impl(y, N, C, HW, gamma, beta, eps, momentum, training, running_mean, running_var, save, a, 0, ws, stream, 1, mm);
}
// plain nn.BatchNorm2d on the same streaming kernels (no activation: the BatchNorms in front of a residual add, models/resnet.py:21-29)
extern "C" int mn_bn2d_fwd(const float* y, int64_t N, int64_t C, int64_t HW, const float* gamma, const float* beta, float eps, float momentum,
                           int training, float* running_mean, float* running_var, float* save, float* a, float* ws, mn_stream_t stream) {
    return bnsign_fwd_impl(y, N, C, HW, gamma, beta, eps, momentum, training, running_mean, running_var, save, a, 0, ws, stream, 2);
}
// ... + per-block min / max of the output (mm: 2 * mn_bnrelu_mm_count(N, C, HW) floats): the BatchNorms in front of an IAO QuantAdd, whose input observers then need no
// pass of their own (mn_iao_qadd_observe_partials)
extern "C" int mn_bn2d_fwd_mm(const float* y, int64_t N, int64_t C, int64_t HW, const float* gamma, const float* beta, float eps, float momentum,
                              int training, float* running_mean, float* running_var, float* save, float* a, float* ws, float* mm, mn_stream_t stream) {
    return bnsign_fwd_impl(y, N, C, HW, gamma, beta, eps, momentum, training, running_mean, running_var, save, a, 0, ws, stream, 2, mm);
}
// training-mode BatchNorm2d [+ ReLU] from the exact accumulator sums of the dense IAO conv in front (mn_actq.stats): ONE pass over y
extern "C" int mn_bn_fwd_acc(const float* y, int64_t N, int64_t C, int64_t HW, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                             float* running_var, float* save, float* a, float* mm, int act, const double* stats, int64_t rows, const float* sa, const float* sw,
                             int64_t sw_stride, const float* conv_bias, mn_stream_t stream) {
    int rc = bns_check(N, C, HW, y, a, "mn_bn_fwd_acc");
    if (rc) return rc;
    if (!y || !gamma || !beta || !save || !a || !stats || !sa || !sw || rows <= 0 || rows > 65536 || (act != 1 && act != 2) || (((uintptr_t)stats) & 7))
        MN_FAIL(MN_EINVAL, "mn_bn_fwd_acc: null / misaligned argument");
    hipStream_t s = (hipStream_t)stream;
    BnsGeom g = bns_geom(N, C, HW);
    g.act = act;
    const int S = bns_split(g);
    BnsFin fin = {};
    fin.part = stats; fin.S = (int)rows; fin.eps = eps; fin.momentum = momentum; fin.running_mean = running_mean; fin.running_var = running_var; fin.save_out = save;
    fin.kind = 1; fin.sa = sa; fin.sw = sw; fin.sw_stride = (int)sw_stride; fin.cbias = conv_bias;
    mn_set_last_kernel("k_bns_apply<0, 0>"); mn_prof_bytes(8.0 * (double)N * C * HW); mn_prof_begin(s);
    hipLaunchKernelGGL((k_bns_apply<0, 0>), dim3((unsigned)C, (unsigned)S), dim3(256), 0, s, g, y, (const float*)nullptr, (const float*)save, gamma, beta,
                       (const float*)nullptr, 1, a, mm, fin, BnsQB());
    mn_prof_end(s);
    MN_CHECK_LAUNCH("mn_bn_fwd_acc");
    return MN_OK;
}
// ---------------------------------------------------------------- BatchNorm [+ ReLU] behind a dense IAO conv, fused with the NEXT conv's activation quantizer
// (models/resnet.py:17-29 conv -> bn -> relu -> conv under wqaq/iao/quantize.py:492-507; the observer of that quantizer, :23-36 / :214-240, must see the whole activation
// before one element is quantised).  y = al[c] * acc + cb[c] with acc the conv's integer accumulator, so per channel every step of
//     acc -> y -> zh = (y - mean) * invstd -> z = zh * gamma + beta -> a = relu(z)
// is a monotone fp32 function of acc (rounding is monotone; a negative gamma only swaps the ends): the extrema of `a` over the channel are `a` at the accumulator's
// extrema, which the conv's epilogue leaves beside its exact sums (mn_actq.acc_mm).  k_bn_acc_prep: one wave per channel, no pass over y.
struct BnAccPrep {
    int C, rows, act;
    double n;                      // N * HW
    float eps, momentum;
    const double* stats;           // [rows][C][2]
    const int32_t* accmm;          // [rows][C][2]
    const float *sa, *sw, *cbias, *gamma, *beta;
    int sw_stride;
    float *running_mean, *running_var, *save, *mm;
};
__global__ __launch_bounds__(256) void k_bn_acc_prep(const BnAccPrep p) {
    const int lane = threadIdx.x & 63, c = blockIdx.x * 4 + ((int)threadIdx.x >> 6);
    if (c < p.C) {                 // (wave-uniform)
        double s1 = 0.0, s2 = 0.0;     // exact integers < 2^53: any order of summation gives the same double
        int lo = INT_MAX, hi = INT_MIN;
        for (int i = lane; i < p.rows; i += 64) {
            s1 += p.stats[((int64_t)i * p.C + c) * 2]; s2 += p.stats[((int64_t)i * p.C + c) * 2 + 1];
            const int a = p.accmm[((int64_t)i * p.C + c) * 2], b = p.accmm[((int64_t)i * p.C + c) * 2 + 1];
            lo = a < lo ? a : lo; hi = b > hi ? b : hi;
        }
        s1 = wave_reduce(s1, OpAddD()); s2 = wave_reduce(s2, OpAddD());
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const int a = __shfl_down(lo, o, 64), b = __shfl_down(hi, o, 64); lo = a < lo ? a : lo; hi = b > hi ? b : hi; }
        if (lane == 0) {
            // the statistics: k_bns_apply's fin.kind == 1 arithmetic, expression for expression
            const float alf = p.sa[0] * p.sw[(int64_t)c * p.sw_stride], cb = p.cbias ? p.cbias[c] : 0.f;
            const double al = (double)alf;
            const double m = s1 / p.n;
            const double mean_d = al * m + (double)cb;
            double ss = al * al * (s2 - s1 * m);
            if (ss < 0.0) ss = 0.0;
            const float var_b = (float)(ss / p.n);
            const float mean = (float)mean_d, invstd = 1.0f / sqrtf(var_b + p.eps);
            p.save[c] = mean; p.save[p.C + c] = invstd;
            if (p.running_mean) p.running_mean[c] = (1.f - p.momentum) * p.running_mean[c] + p.momentum * (float)mean_d;
            if (p.running_var) p.running_var[c] = (1.f - p.momentum) * p.running_var[c] + p.momentum * (float)(ss / (p.n - 1.0));
            // the activation at the two ends: the conv epilogue's y (k_qd_fwd8), k_bns_apply's z
            const float ga = p.gamma[c], be = p.beta[c];
            float e[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float y = (float)(k ? hi : lo) * alf + cb;
                const float zh = (y - mean) * invstd;
                const float z = zh * ga + be;
                e[k] = p.act == 2 ? z : bns_relu(z);
            }
            p.mm[c] = OpMinF()(e[0], e[1]); p.mm[p.C + c] = OpMaxF()(e[0], e[1]);
        }
    }
}
// (Tried and dropped: the consumer's observer / qparams launch as a tail of this one -- the block that draws the last ticket reduces the C partials.  The partials of
// the other blocks have to cross XCDs, whose L2s are not coherent with each other: the release fence every block needs writes its XCD's L2 back, and the launch went from
// 5.7 to 13.8 us -- as much as the two launches it replaced.  profiles/README.md.)
extern "C" int mn_bn_acc_prep(int64_t N, int64_t C, int64_t HW, const float* gamma, const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                              float* save, int act, const double* stats, const int32_t* acc_mm, int64_t rows, const float* sa, const float* sw, int64_t sw_stride,
                              const float* conv_bias, float* mm, mn_stream_t stream) {
    if (N <= 0 || C <= 0 || HW <= 0 || !gamma || !beta || !save || !stats || !acc_mm || !sa || !sw || !mm || rows <= 0 || rows > 65536 || (act != 1 && act != 2) ||
        (((uintptr_t)stats) & 7) || (((uintptr_t)acc_mm) & 3))
        MN_FAIL(MN_EINVAL, "mn_bn_acc_prep: null / misaligned argument");
    BnAccPrep p;
    p.C = (int)C; p.rows = (int)rows; p.act = act; p.n = (double)N * (double)HW; p.eps = eps; p.momentum = momentum; p.stats = stats; p.accmm = acc_mm;
    p.sa = sa; p.sw = sw; p.cbias = conv_bias; p.gamma = gamma; p.beta = beta; p.sw_stride = (int)sw_stride;
    p.running_mean = running_mean; p.running_var = running_var; p.save = save; p.mm = mm;
    mn_set_last_kernel("k_bn_acc_prep");
    hipLaunchKernelGGL(k_bn_acc_prep, dim3((unsigned)((C + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
    MN_CHECK_LAUNCH("mn_bn_acc_prep");
    return MN_OK;
}
// a = act(bn(y)) (k_bns_apply<0, 0>'s expressions) -> the symmetric quantizer's signed codes + clip-STE bits (k_qd_iao_codes' decisions on a): thread = 8 consecutive
// elements of one plane (two float4 in, one 8-byte code store, one mask byte).  The pass is VALU-bound before it is HBM-bound (first version: 55 instructions per
// element, 2.4 TB/s), so: the activation is a template parameter, a / sc is Markstein's three-instruction correctly rounded quotient (mn_div_m: the same float as
// the IEEE sequence for in-range operands), and with zero_point == 0 (the symmetric quantizer: always) the rounded value serves the code and the clip test.
template <int ACT, int ZP0>
__device__ __forceinline__ void bn_apply_codes_body(const BnsGeom& g, const float* __restrict__ y, float mean, float invstd, float ga, float be, float sc, float zp, float rlo,
                                                    float rhi, float qmin, float qmax, signed char* __restrict__ codes, unsigned char* __restrict__ mask) {
    const int c = blockIdx.x, sp = blockIdx.y, S = gridDim.y;
    const float inv = 1.0f / sc;
    const uint32_t HW8 = (uint32_t)g.HW4 >> 1;
    const int64_t n8 = g.n4 >> 1;
    // rha(v) as copysign(floor(|v| + 0.5), v): the float of mn_rha except for the sign of a zero, which neither the code byte nor the comparisons see
    auto rha = [](float v) { return copysignf(floorf(fabsf(v) + 0.5f), v); };
    float4 va_[2], vb_[2];
    int64_t off_[2];
#define BNC_LOAD(k, idx) { const uint32_t n_ = fd_div(2u * (uint32_t)(idx), g.fd_hw4); off_[k] = ((int64_t)n_ * g.C + c) * g.HW + (int64_t)((uint32_t)(idx) - n_ * HW8) * 8; \
                           va_[k] = *reinterpret_cast<const float4*>(y + off_[k]); vb_[k] = *reinterpret_cast<const float4*>(y + off_[k] + 4); }
#define BNC_FIN(k) { const float v[8] = {va_[k].x, va_[k].y, va_[k].z, va_[k].w, vb_[k].x, vb_[k].y, vb_[k].z, vb_[k].w};                                  \
        uint32_t m = 0u, lo = 0u, hi = 0u;                                                                                                                  \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                                                                     \
            const float zh = (v[e] - mean) * invstd;                                                                                                        \
            const float z = zh * ga + be;                                                                                                                   \
            /* relu: a NaN stays a NaN in bns_relu -- here it becomes 0 (code 0 either way) and is failed by the z == z term of the clip test */             \
            /* (+inf and whatever the quotient's fma chain would overflow on: clamped far outside the quantizer's range -- same code, same failed clip test) */ \
            const float a = ACT == 2 ? fminf(fmaxf(z, -1.0e30f), 1.0e30f) : fminf(fmaxf(z, 0.f), 1.0e30f);                                                  \
            const float q = mn_div_m(a, sc, inv);                                                                                                           \
            const float vv = ZP0 ? q : q - zp, r = rha(vv);                                                                                                 \
            m |= ((r >= qmin && r <= qmax && !(vv > rhi || vv < rlo) && (ACT == 2 || z == z)) ? 1u : 0u) << e;                                              \
            const float rq = ZP0 ? r : rha(q);                                                                                                              \
            const float cl = fminf(fmaxf(rq, qmin), qmax);                             /* clamp(rha(a / sc)); NaN -> 0 (a byte cannot hold it) */            \
            const float cc = ACT == 2 ? ((z == z) ? cl : 0.f) : cl;                                                                                         \
            const uint32_t byte = (uint32_t)(int)cc & 0xffu;                                                                                                \
            if (e < 4) lo |= byte << (8 * e); else hi |= byte << (8 * (e - 4));                                                                             \
        }                                                                                                                                                   \
        *reinterpret_cast<u32x2*>(codes + off_[k]) = u32x2{lo, hi};                                                                                         \
        mask[off_[k] >> 3] = (unsigned char)m; }
    MN_STREAM_2(i, (int64_t)sp * 256 + threadIdx.x, (int64_t)S * 256, n8, BNC_LOAD, BNC_FIN)
#undef BNC_LOAD
#undef BNC_FIN
}
template <int ACT>
__global__ __launch_bounds__(256) void k_bn_apply_codes(const BnsGeom g, const float* __restrict__ y, const float* __restrict__ save, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ qp, float qmin, float qmax,
                                                        signed char* __restrict__ codes, unsigned char* __restrict__ mask) {
    const int c = blockIdx.x;
    const float mean = save[c], invstd = save[g.C + c], ga = gamma[c], be = beta[c];
    const float sc = qp[0], zp = qp[1], rlo = qp[2], rhi = qp[3];
    if (zp == 0.f) bn_apply_codes_body<ACT, 1>(g, y, mean, invstd, ga, be, sc, zp, rlo, rhi, qmin, qmax, codes, mask);          // (the symmetric quantizer: always)
    else bn_apply_codes_body<ACT, 0>(g, y, mean, invstd, ga, be, sc, zp, rlo, rhi, qmin, qmax, codes, mask);
}
extern "C" int mn_bn_apply_codes(const float* y, int64_t N, int64_t C, int64_t HW, const float* gamma, const float* beta, const float* save, int act, const float* qp, int bits,
                                 int8_t* codes, uint8_t* ste_mask, mn_stream_t stream) {
    int rc = bns_check(N, C, HW, y, y, "mn_bn_apply_codes");
    if (rc) return rc;
    if (!y || !gamma || !beta || !save || !qp || !codes || !ste_mask || HW % 8 || bits < 2 || bits > 8 || (act != 1 && act != 2) || (((uintptr_t)codes) & 7))
        MN_FAIL(MN_EINVAL, "mn_bn_apply_codes: null / misaligned argument, HW not a multiple of 8 or bits outside 2..8");
    BnsGeom g = bns_geom(N, C, HW);
    g.act = act;
    const int S = bns_split(g);
    const IaoRange r = iao_range(bits, 0, 1);
    hipStream_t s = (hipStream_t)stream;
    mn_set_last_kernel("k_bn_apply_codes"); mn_prof_bytes(5.125 * (double)N * C * HW); mn_prof_begin(s);
    if (act == 1) hipLaunchKernelGGL(k_bn_apply_codes<1>, dim3((unsigned)C, (unsigned)S), dim3(256), 0, s, g, y, save, gamma, beta, qp, r.qmin, r.qmax, (signed char*)codes, (unsigned char*)ste_mask);
    else hipLaunchKernelGGL(k_bn_apply_codes<2>, dim3((unsigned)C, (unsigned)S), dim3(256), 0, s, g, y, save, gamma, beta, qp, r.qmin, r.qmax, (signed char*)codes, (unsigned char*)ste_mask);
    mn_prof_end(s);
    MN_CHECK_LAUNCH("mn_bn_apply_codes");
    return MN_OK;
}
extern "C" int mn_bn_apply(const float* y, int64_t N, int64_t C, int64_t HW, const float* gamma, const float* beta, const float* save, int act, float* a, mn_stream_t stream) {
    if (act != 1 && act != 2) MN_FAIL(MN_EINVAL, "mn_bn_apply: act must be 1 (ReLU) or 2 (none)");
    return bnsign_fwd_impl(y, N, C, HW, gamma, beta, 0.f, 0.f, 1, nullptr, nullptr, const_cast<float*>(save), a, 0, nullptr, stream, act, nullptr, true);
}
// ---------------------------------------------------------------- the END of an IAO residual block: BatchNorm(s) + QuantAdd [+ ReLU] in one pass
// out = [relu] (Q(res) + Q(shortcut)) with res = bn(y_res) and shortcut = a plain tensor (identity) or bn(y_sc) (the down-sampling blocks), one shared per-tensor
// quantizer (models/resnet.py:21-29, 60-65 under wqaq/iao/quantize.py:1484-1498).  Unfused that is a BatchNorm apply pass per side (8 B per element each) + k_qadd_fwd
// (12 B); here 12.25 B: the BatchNorm outputs never exist.  Their ranges -- the QuantAdd's two input observers see the whole tensors first -- come from the convs'
// accumulator extrema (mn_bn_acc_prep).  Expression for expression k_bns_apply<0, 0> (act none) -> iao_fq x 2 -> add -> qa_relu, with a / sc as Markstein's
// correctly rounded quotient.  Also leaves, for the backward, one bit per element and side: (the ReLU passes) and (the quantizer's clip-STE passes that input) --
// k_bns_partial<1, 1> / k_bns_apply<1, 0, 1> read them next to the output gradient, so neither d res nor d shortcut is written for a BatchNorm to read back.
struct QabParams {
    BnsGeom g;
    const float *res_y, *res_save, *res_gamma, *res_beta;
    const float *sc_x, *sc_save, *sc_gamma, *sc_beta;          // sc_save == null: the shortcut is sc_x itself
    const float* qp;
    float qmin, qmax;
    float *out, *mm;
    unsigned char *bits_res, *bits_sc;
};
template <int RELU, int SCBN>
__global__ __launch_bounds__(256) void k_qadd_bn_fwd(const QabParams p) {
    const BnsGeom& g = p.g;
    const int c = blockIdx.x, sp = blockIdx.y, S = gridDim.y;
    const float mean = p.res_save[c], invstd = p.res_save[g.C + c], ga = p.res_gamma[c], be = p.res_beta[c];
    float mean2 = 0.f, invstd2 = 1.f, ga2 = 1.f, be2 = 0.f;
    if (SCBN) { mean2 = p.sc_save[c]; invstd2 = p.sc_save[g.C + c]; ga2 = p.sc_gamma[c]; be2 = p.sc_beta[c]; }
    const float sc = p.qp[0], zp = p.qp[1], rlo = p.qp[2], rhi = p.qp[3], qmin = p.qmin, qmax = p.qmax;
    const float inv = 1.0f / sc;
    const uint32_t HW8 = (uint32_t)g.HW4 >> 1;
    const int64_t n8 = g.n4 >> 1;
    float mlo = INFINITY, mhi = -INFINITY;
    float4 ra_[2], rb_[2], sa_[2], sb_[2];
    int64_t off_[2];
#define QAB_LOAD(k, idx) { const uint32_t n_ = fd_div(2u * (uint32_t)(idx), g.fd_hw4); off_[k] = ((int64_t)n_ * g.C + c) * g.HW + (int64_t)((uint32_t)(idx) - n_ * HW8) * 8; \
                           ra_[k] = *reinterpret_cast<const float4*>(p.res_y + off_[k]); rb_[k] = *reinterpret_cast<const float4*>(p.res_y + off_[k] + 4);                   \
                           sa_[k] = *reinterpret_cast<const float4*>(p.sc_x + off_[k]); sb_[k] = *reinterpret_cast<const float4*>(p.sc_x + off_[k] + 4); }
#define QAB_FIN(k) { const float rv[8] = {ra_[k].x, ra_[k].y, ra_[k].z, ra_[k].w, rb_[k].x, rb_[k].y, rb_[k].z, rb_[k].w};                                                   \
        const float sv[8] = {sa_[k].x, sa_[k].y, sa_[k].z, sa_[k].w, sb_[k].x, sb_[k].y, sb_[k].z, sb_[k].w};                                                                \
        float o[8];                                                                                                                                                          \
        uint32_t mr = 0u, ms = 0u;                                                                                                                                           \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                                                                                      \
            const float zh = (rv[e] - mean) * invstd;                                                                                                                        \
            const float a = zh * ga + be;                                                                                                                                    \
            float b = sv[e];                                                                                                                                                 \
            if (SCBN) { const float zh2 = (b - mean2) * invstd2; b = zh2 * ga2 + be2; }                                                                                     \
            const float va = mn_div_m(a, sc, inv) - zp, vb = mn_div_m(b, sc, inv) - zp;                                                                                      \
            const float ra = mn_rha(va), rb = mn_rha(vb);                                                                                                                    \
            const float sum = (mn_clamp(ra, qmin, qmax) + zp) * sc + (mn_clamp(rb, qmin, qmax) + zp) * sc;                                                                   \
            o[e] = RELU ? qa_relu(sum) : sum;                                                                                                                                \
            const bool live = !RELU || sum > 0.f;                                                                                                                            \
            mr |= ((live && ra >= qmin && ra <= qmax && !(va > rhi || va < rlo)) ? 1u : 0u) << e;                                                                            \
            ms |= ((live && rb >= qmin && rb <= qmax && !(vb > rhi || vb < rlo)) ? 1u : 0u) << e;                                                                            \
        }                                                                                                                                                                    \
        *reinterpret_cast<float4*>(p.out + off_[k]) = make_float4(o[0], o[1], o[2], o[3]);                                                                                   \
        *reinterpret_cast<float4*>(p.out + off_[k] + 4) = make_float4(o[4], o[5], o[6], o[7]);                                                                               \
        p.bits_res[off_[k] >> 3] = (unsigned char)mr; p.bits_sc[off_[k] >> 3] = (unsigned char)ms;                                                                           \
        if (p.mm) { _Pragma("unroll") for (int e = 0; e < 8; ++e) { mlo = OpMinF()(mlo, o[e]); mhi = OpMaxF()(mhi, o[e]); } } }
    MN_STREAM_2(i, (int64_t)sp * 256 + threadIdx.x, (int64_t)S * 256, n8, QAB_LOAD, QAB_FIN)
#undef QAB_LOAD
#undef QAB_FIN
    if (p.mm) {
        __shared__ float scm[16];
        mlo = block_reduce(mlo, OpMinF(), INFINITY, scm);
        mhi = block_reduce(mhi, OpMaxF(), -INFINITY, scm);
        if (threadIdx.x == 0) { const int b = sp * (int)gridDim.x + c, nb = (int)(gridDim.x * gridDim.y); p.mm[b] = mlo; p.mm[nb + b] = mhi; }
    }
}
extern "C" int mn_iao_qadd_bn_fwd(const float* res_y, const float* res_save, const float* res_gamma, const float* res_beta, const float* sc_x, const float* sc_save,
                                  const float* sc_gamma, const float* sc_beta, int64_t N, int64_t C, int64_t HW, const float* qp, int bits, int q_type, int relu, float* out,
                                  float* mm, uint8_t* bits_res, uint8_t* bits_sc, mn_stream_t stream) {
    int rc = bns_check(N, C, HW, res_y, out, "mn_iao_qadd_bn_fwd");
    if (rc) return rc;
    if (!res_y || !res_save || !res_gamma || !res_beta || !sc_x || !aligned16(sc_x) || (sc_save && (!sc_gamma || !sc_beta)) || !qp || !out || !bits_res || !bits_sc || HW % 8 ||
        bits < 2 || bits > 24 || (q_type != 0 && q_type != 1))
        MN_FAIL(MN_EINVAL, "mn_iao_qadd_bn_fwd: null / misaligned argument, HW not a multiple of 8 or a bad bit width");
    QabParams p;
    p.g = bns_geom(N, C, HW);
    p.g.act = 2;
    const int S = bns_split(p.g);
    const IaoRange r = iao_range(bits, q_type, 1);
    p.res_y = res_y; p.res_save = res_save; p.res_gamma = res_gamma; p.res_beta = res_beta; p.sc_x = sc_x; p.sc_save = sc_save; p.sc_gamma = sc_gamma; p.sc_beta = sc_beta;
    p.qp = qp; p.qmin = r.qmin; p.qmax = r.qmax; p.out = out; p.mm = mm; p.bits_res = bits_res; p.bits_sc = bits_sc;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)C, (unsigned)S);
    mn_set_last_kernel("k_qadd_bn_fwd<%d, %d>", relu ? 1 : 0, sc_save ? 1 : 0); mn_prof_bytes(12.25 * (double)N * C * HW); mn_prof_begin(s);
    if (relu && sc_save) hipLaunchKernelGGL((k_qadd_bn_fwd<1, 1>), grid, dim3(256), 0, s, p);
    else if (relu) hipLaunchKernelGGL((k_qadd_bn_fwd<1, 0>), grid, dim3(256), 0, s, p);
    else if (sc_save) hipLaunchKernelGGL((k_qadd_bn_fwd<0, 1>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((k_qadd_bn_fwd<0, 0>), grid, dim3(256), 0, s, p);
    mn_prof_end(s);
    MN_CHECK_LAUNCH("mn_iao_qadd_bn_fwd");
    return MN_OK;
}
// backward of ONE BatchNorm side of that block: g = the gradient of the block's output; the gradient of the BatchNorm's own output is formed from (g, bits) on the way
// (two passes over (g, y) like mn_bn2d_bwd, same order of summation: bit-identical to mn_iao_qadd_bwd -> mn_bn2d_bwd).  d_other (nullable; with bits_other): the
// gradient of the QuantAdd's other input from the same read of g -- the identity shortcut's.
extern "C" int mn_iao_qadd_bn_bwd(const float* g, const float* y, const float* save, const float* gamma, const float* beta, int64_t N, int64_t C, int64_t HW, const float* qp,
                                  const uint8_t* bits, const uint8_t* bits_other, float* dy, float* d_other, float* dgamma, float* dbeta, float* ws, mn_stream_t stream) {
    int rc = bns_check(N, C, HW, y, dy, "mn_iao_qadd_bn_bwd");
    if (rc) return rc;
    if (!g || !y || !save || !gamma || !beta || !dy || !ws || !qp || !bits || (d_other && (!bits_other || !aligned16(d_other))) || !aligned16(g) || (((uintptr_t)ws) & 7) || HW % 8)
        MN_FAIL(MN_EINVAL, "mn_iao_qadd_bn_bwd: null / misaligned argument");
    hipStream_t s = (hipStream_t)stream;
    BnsGeom gg = bns_geom(N, C, HW);
    gg.act = 2;
    const int S = bns_split(gg);
    float* sums = ws + C * BNS_SPLIT * 4;
    const double nel = (double)N * C * HW;
    BnsQB qb;
    qb.bits = bits; qb.bits_sc = bits_other; qb.dsc = d_other; qb.qp = qp;
    mn_set_last_kernel("k_bns_partial<1, 1>"); mn_prof_bytes(8.125 * nel); mn_prof_begin(s);
    hipLaunchKernelGGL((k_bns_partial<1, 1>), dim3((unsigned)C, (unsigned)S), dim3(256), 0, s, gg, y, g, save, gamma, beta, (double*)ws, qb);
    mn_prof_end(s);
    BnsFin fin = {};
    fin.part = (const double*)ws; fin.S = S; fin.dgamma = dgamma; fin.dbeta = dbeta; fin.sums_out = sums;
    mn_set_last_kernel("k_bns_apply<1, 0, 1>"); mn_prof_bytes((d_other ? 16.25 : 12.125) * nel); mn_prof_begin(s);
    hipLaunchKernelGGL((k_bns_apply<1, 0, 1>), dim3((unsigned)C, (unsigned)S), dim3(256), 0, s, gg, y, g, save, gamma, beta, (const float*)sums, 1, dy, (float*)nullptr, fin, qb);
    mn_prof_end(s);
    MN_CHECK_LAUNCH("mn_iao_qadd_bn_bwd");
    return MN_OK;
}
extern "C" int mn_bn2d_bwd(const float* da, const float* y, const float* save, const float* gamma, const float* beta, int64_t N, int64_t C,
                           int64_t HW, int training, float* dy, float* dgamma, float* dbeta, float* ws, mn_stream_t stream) {
    return bnsign_bwd_impl(da, y, save, gamma, beta, N, C, HW, training, dy, dgamma, dbeta, ws, stream, 2);
}
static int bnsign_bwd_impl(const float* da, const float* y, const float* save, const float* gamma, const float* beta, int64_t N, int64_t C,
                           int64_t HW, int training, float* dy, float* dgamma, float* dbeta, float* ws, mn_stream_t stream, int act) {
    int rc = bns_check(N, C, HW, y, dy, "mn_bnsign_bwd");
    if (rc) return rc;
    if (!da || !y || !save || !gamma || !beta || !dy || !ws || !aligned16(da) || (((uintptr_t)ws) & 7)) MN_FAIL(MN_EINVAL, "mn_bnsign_bwd: null / misaligned argument");
    hipStream_t s = (hipStream_t)stream;
    BnsGeom g = bns_geom(N, C, HW);
    g.act = act;
    const int S = bns_split(g);
    float* sums = ws + C * BNS_SPLIT * 4;
    const double nel = (double)N * C * HW;
    mn_set_last_kernel("k_bns_partial<1>"); mn_prof_bytes(8.0 * nel); mn_prof_begin(s);
    hipLaunchKernelGGL(k_bns_partial<1>, dim3((unsigned)C, (unsigned)S), dim3(256), 0, s, g, y, da, save, gamma, beta, (double*)ws, BnsQB());
    mn_prof_end(s);
    BnsFin fin = {};
    fin.part = (const double*)ws; fin.S = S; fin.dgamma = dgamma; fin.dbeta = dbeta; fin.sums_out = sums;
    mn_set_last_kernel("k_bns_apply<1, 0>"); mn_prof_bytes(12.0 * nel); mn_prof_begin(s);
    hipLaunchKernelGGL((k_bns_apply<1, 0>), dim3((unsigned)C, (unsigned)S), dim3(256), 0, s, g, y, da, save, gamma, beta, (const float*)sums, training, dy, (float*)nullptr, fin, BnsQB());
    mn_prof_end(s);
    MN_CHECK_LAUNCH("mn_bnsign_bwd");
    return MN_OK;
}

// ---------------------------------------------------------------- the TAIL of the reference's nets: BatchNorm2d -> ReLU -> global average pool, and the loss
// models/nin_gc.py:136-147 ends conv(-> 10) -> bn -> relu -> AvgPool2d(8) -> view; main.py's criterion is nn.CrossEntropyLoss() (wqaq/dorefa/main.py:87-92).  The
// tensors are tiny (N x 10 x 8 x 8), so the step pays launches, not bytes: MIOpen's BatchNorm, ATen's ReLU / softmax / nll_loss and their backward + fills were 18
// launches of ~5 us per nin_gc step (4 % of the c2 step).  Here: one kernel per direction for bn + relu + pool (one block per channel; training mode), one for the
// loss WITH its gradient (the backward only scales it by the incoming scalar).
// One block per channel; the channel's N * HW / 4 float4 live in REGISTERS (<= TAIL_Q per thread, all loads in flight at once -- the first version walked each
// image with one thread and paid a memory round trip per float4: 50 us), the three passes (mean, variance, normalise) run on them; float4 i belongs to image
// i / (HW / 4): with HW / 4 a power of two <= 64 the lanes of one image are neighbours and the per-image pool is a shuffle reduction (deterministic).
#define TAIL_Q 16
struct TailGeom { int N, C, HW, HW4, n4, sh; };          // n4 = N * HW4 <= 256 * TAIL_Q, HW4 = 1 << sh
__device__ __forceinline__ void tail_load(const TailGeom& g, int c, const float* __restrict__ y, float4 (&v)[TAIL_Q], int64_t (&off)[TAIL_Q]) {
#pragma unroll
    for (int k = 0; k < TAIL_Q; ++k) {
        const int i = (int)threadIdx.x + 256 * k;
        const int ii = i < g.n4 ? i : 0;
        const int im = ii >> g.sh, q = ii & (g.HW4 - 1);
        off[k] = ((int64_t)im * g.C + c) * g.HW + 4 * q;
        v[k] = *reinterpret_cast<const float4*>(y + off[k]);
    }
}
// sum over the HW4 lanes that hold one image (aligned groups of HW4 lanes)
__device__ __forceinline__ float tail_group_sum(float t, int HW4) {
    for (int o = HW4 >> 1; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    return t;
}
__global__ __launch_bounds__(256) void k_bnrelu_gap_fwd(const TailGeom g, const float* __restrict__ y, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
                                                        float* __restrict__ save, float* __restrict__ pooled) {
    __shared__ double scd[16];
    __shared__ double bc[2];
    const int c = blockIdx.x;
    const double n = (double)g.N * (double)g.HW;
    float4 v[TAIL_Q];
    int64_t off[TAIL_Q];
    tail_load(g, c, y, v, off);
    double s1 = 0.0;
#pragma unroll
    for (int k = 0; k < TAIL_Q; ++k) if ((int)threadIdx.x + 256 * k < g.n4) s1 += (double)((v[k].x + v[k].y) + (v[k].z + v[k].w));
    s1 = block_reduce(s1, OpAddD(), 0.0, scd);
    if (threadIdx.x == 0) bc[0] = s1 / n;
    __syncthreads();
    const float mean = (float)bc[0];
    double s2 = 0.0;
#pragma unroll
    for (int k = 0; k < TAIL_Q; ++k) {
        if ((int)threadIdx.x + 256 * k < g.n4) { const float a = v[k].x - mean, b = v[k].y - mean, cc = v[k].z - mean, d = v[k].w - mean; s2 += (double)((a * a + b * b) + (cc * cc + d * d)); }
    }
    s2 = block_reduce(s2, OpAddD(), 0.0, scd);
    if (threadIdx.x == 0) bc[1] = s2;
    __syncthreads();
    const double ss = bc[1];
    const float invstd = 1.0f / sqrtf((float)(ss / n) + eps);
    if (threadIdx.x == 0) {
        save[c] = mean; save[g.C + c] = invstd;
        if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        if (running_var) running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(ss / (n - 1.0));
    }
    const float ga = gamma[c], be = beta[c];
#pragma unroll
    for (int k = 0; k < TAIL_Q; ++k) {
        const int i = (int)threadIdx.x + 256 * k;
        const float e[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
        float t = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) { const float zh = (e[u] - mean) * invstd; t += bns_relu(zh * ga + be); }
        t = tail_group_sum(i < g.n4 ? t : 0.f, g.HW4);          // (every lane takes part in the shuffles)
        if (i < g.n4 && (i & (g.HW4 - 1)) == 0) pooled[(int64_t)(i >> g.sh) * g.C + c] = t / (float)g.HW;
    }
}
__global__ __launch_bounds__(256) void k_bnrelu_gap_bwd(const TailGeom g, const float* __restrict__ dpool, const float* __restrict__ y, const float* __restrict__ save,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ dy,
                                                        float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ double scd[16];
    __shared__ double bc[2];
    const int c = blockIdx.x;
    const float mean = save[c], invstd = save[g.C + c], ga = gamma[c], be = beta[c];
    float4 v[TAIL_Q];
    int64_t off[TAIL_Q];
    float gq[TAIL_Q];
    tail_load(g, c, y, v, off);
#pragma unroll
    for (int k = 0; k < TAIL_Q; ++k) { const int i = (int)threadIdx.x + 256 * k; gq[k] = dpool[(int64_t)((i < g.n4 ? i : 0) >> g.sh) * g.C + c] / (float)g.HW; }
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < TAIL_Q; ++k) {
        if ((int)threadIdx.x + 256 * k >= g.n4) continue;
        const float e[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) { const float zh = (e[u] - mean) * invstd; const float dz = (zh * ga + be > 0.f) ? gq[k] : 0.f; t1 += dz; t2 += dz * zh; }
        s1 += (double)t1; s2 += (double)t2;
    }
    s1 = block_reduce(s1, OpAddD(), 0.0, scd);
    s2 = block_reduce(s2, OpAddD(), 0.0, scd);
    if (threadIdx.x == 0) { bc[0] = s1; bc[1] = s2; dbeta[c] = (float)s1; dgamma[c] = (float)s2; }
    __syncthreads();
    const float nf = (float)g.N * (float)g.HW;
    const float k1 = (float)bc[0] / nf, k2 = (float)bc[1] / nf, gi = ga * invstd;
#pragma unroll
    for (int k = 0; k < TAIL_Q; ++k) {
        if ((int)threadIdx.x + 256 * k >= g.n4) continue;
        const float e[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
        float o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const float zh = (e[u] - mean) * invstd; const float dz = (zh * ga + be > 0.f) ? gq[k] : 0.f; o[u] = gi * (dz - k1 - zh * k2); }
        *reinterpret_cast<float4*>(dy + off[k]) = make_float4(o[0], o[1], o[2], o[3]);
    }
}
static int tail_geom(int64_t N, int64_t C, int64_t HW, TailGeom* g) {
    if (N <= 0 || C <= 0 || C > 65535 || HW <= 0 || HW % 4) return 0;
    const int64_t HW4 = HW / 4;
    int sh = 0;
    while ((1 << sh) < HW4) ++sh;
    if ((1 << sh) != HW4 || HW4 > 64 || N * HW4 > 256 * TAIL_Q || N * HW < 2) return 0;
    g->N = (int)N; g->C = (int)C; g->HW = (int)HW; g->HW4 = (int)HW4; g->n4 = (int)(N * HW4); g->sh = sh;
    return 1;
}
extern "C" int mn_bnrelu_gap_supported(int64_t N, int64_t C, int64_t HW) { TailGeom g; return tail_geom(N, C, HW, &g); }
extern "C" int mn_bnrelu_gap_fwd(const float* y, int64_t N, int64_t C, int64_t HW, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                 float* running_var, float* save, float* pooled, mn_stream_t stream) {
    TailGeom g;
    if (!y || !gamma || !beta || !save || !pooled || !aligned16(y) || !tail_geom(N, C, HW, &g))
        MN_FAIL(MN_EINVAL, "mn_bnrelu_gap_fwd: bad arguments (mn_bnrelu_gap_supported: HW / 4 a power of two <= 64, N * HW <= 16384; 16-byte aligned input)");
    mn_set_last_kernel("k_bnrelu_gap_fwd");
    hipLaunchKernelGGL(k_bnrelu_gap_fwd, dim3((unsigned)C), dim3(256), 0, (hipStream_t)stream, g, y, gamma, beta, eps, momentum, running_mean, running_var, save, pooled);
    MN_CHECK_LAUNCH("mn_bnrelu_gap_fwd");
    return MN_OK;
}
extern "C" int mn_bnrelu_gap_bwd(const float* dpool, const float* y, const float* save, const float* gamma, const float* beta, int64_t N, int64_t C, int64_t HW, float* dy,
                                 float* dgamma, float* dbeta, mn_stream_t stream) {
    TailGeom g;
    if (!dpool || !y || !save || !gamma || !beta || !dy || !dgamma || !dbeta || !aligned16(y) || !aligned16(dy) || !tail_geom(N, C, HW, &g))
        MN_FAIL(MN_EINVAL, "mn_bnrelu_gap_bwd: bad arguments");
    mn_set_last_kernel("k_bnrelu_gap_bwd");
    hipLaunchKernelGGL(k_bnrelu_gap_bwd, dim3((unsigned)C), dim3(256), 0, (hipStream_t)stream, g, dpool, y, save, gamma, beta, dy, dgamma, dbeta);
    MN_CHECK_LAUNCH("mn_bnrelu_gap_bwd");
    return MN_OK;
}
// loss = mean over the samples with target != ignore_index of (logsumexp(x_i) - x_i[target_i]), and dlogits = d loss / d x for an incoming gradient of 1
__global__ __launch_bounds__(256) void k_ce_fwd(const float* __restrict__ x, const int64_t* __restrict__ target, int N, int K, int64_t ignore_index, float* __restrict__ loss,
                                                float* __restrict__ dlogits) {
    __shared__ double scd[16];
    __shared__ float s_cnt;
    double acc = 0.0, cnt = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) {
        const int64_t t = target[i];
        if (t == ignore_index || t < 0 || t >= K) continue;
        const float* r = x + (int64_t)i * K;
        float m = r[0];
        for (int j = 1; j < K; ++j) m = fmaxf(m, r[j]);
        float se = 0.f;
        for (int j = 0; j < K; ++j) se += expf(r[j] - m);
        acc += (double)((m + logf(se)) - r[t]);
        cnt += 1.0;
    }
    acc = block_reduce(acc, OpAddD(), 0.0, scd);
    cnt = block_reduce(cnt, OpAddD(), 0.0, scd);
    if (threadIdx.x == 0) { loss[0] = (float)(acc / cnt); s_cnt = (float)cnt; }
    __syncthreads();
    const float inv_n = 1.0f / s_cnt;
    for (int i = threadIdx.x; i < N; i += 256) {
        const int64_t t = target[i];
        const float* r = x + (int64_t)i * K;
        float* d = dlogits + (int64_t)i * K;
        if (t == ignore_index || t < 0 || t >= K) { for (int j = 0; j < K; ++j) d[j] = 0.f; continue; }
        float m = r[0];
        for (int j = 1; j < K; ++j) m = fmaxf(m, r[j]);
        float se = 0.f;
        for (int j = 0; j < K; ++j) se += expf(r[j] - m);
        for (int j = 0; j < K; ++j) d[j] = (expf(r[j] - m) / se - (j == (int)t ? 1.f : 0.f)) * inv_n;
    }
}
__global__ __launch_bounds__(256) void k_scale_by(const float* __restrict__ a, const float* __restrict__ s, float* __restrict__ out, int64_t n) {
    const float f = s[0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = a[i] * f;
}
extern "C" int mn_cross_entropy_fwd(const float* logits, const int64_t* target, int64_t N, int64_t K, int64_t ignore_index, float* loss, float* dlogits, mn_stream_t stream) {
    if (!logits || !target || !loss || !dlogits || N <= 0 || K <= 0 || K > 4096 || N * K >= ((int64_t)1 << 31)) MN_FAIL(MN_EINVAL, "mn_cross_entropy_fwd: bad arguments");
    mn_set_last_kernel("k_ce_fwd");
    hipLaunchKernelGGL(k_ce_fwd, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, target, (int)N, (int)K, ignore_index, loss, dlogits);
    MN_CHECK_LAUNCH("mn_cross_entropy_fwd");
    return MN_OK;
}
extern "C" int mn_scale_by(const float* a, const float* scalar, float* out, int64_t n, mn_stream_t stream) {
    if (!a || !scalar || !out || n <= 0) MN_FAIL(MN_EINVAL, "mn_scale_by: bad arguments");
    mn_set_last_kernel("k_scale_by");
    hipLaunchKernelGGL(k_scale_by, dim3(mn_grid_for(n, 256, 1024)), dim3(256), 0, (hipStream_t)stream, a, scalar, out, n);
    MN_CHECK_LAUNCH("mn_scale_by");
    return MN_OK;
}

// ---------------------------------------------------------------- last layer of a binary net: 1x1 conv, few outputs, sign-code input
// The classifier conv of the WbWtAb nets (models/nin_gc.py: 1024 -> 10, 1x1) keeps full-precision weights (the rewrite skips the
// last conv, wbwtab/quantize.py:251) but its INPUT is the +-1 output of the previous block.  O is tiny, so this is a per-pixel dot
// product over C sign codes: y[n][o][p] = bias[o] + sum_c w[o][c] * a[n][c][p].  Block = (image, 64 pixels), 16 waves split the
// channels (wave-uniform ranges: the weights come through the scalar cache), keep O <= 16 running sums per lane and combine them in
// wave order through LDS.  The layer is bound by the latency of its byte loads: each lane keeps 16 in flight, 4 waves per SIMD.
#define SC_MAXO 16
#define SC_WAVES 16
// Block = 64 consecutive pixels of the [N][HW] pixel axis (HW % 4 == 0: a lane's 4 pixels are one dword of one image plane); 16 waves
// split the channels, and inside a wave the four 16-lane groups take every fourth channel: lane (pq, cs) accumulates 4 pixels x OP outputs
// over channels c0 + cs + 4u.  The weights come from an LDS image [C][OP] (OP = O rounded up to 4; broadcast b128 reads, no per-output
// branches, no scalar-load latency chain), the 16 code dwords of a lane are all in flight at once (the layer is latency-, not byte-bound).
// The four channel groups of a wave are combined by two shuffles, the 16 waves through LDS in a fixed order: deterministic.
// ENC 1: the bytes are k-bit activation codes j (DoReFa quantizer output, q = j * ascale): y = bias + ascale * sum_c w[o][c] * j.
template <int OP, int ENC = 0>
__global__ __launch_bounds__(1024) void k_sconv_fwd(const char* __restrict__ a, const float* __restrict__ w, const float* __restrict__ bias,
                                                    float* __restrict__ y, int C, int HW, int O, int64_t NP, float ascale) {
    HIP_DYNAMIC_SHARED(float, smem)
    float* wl = smem;                          // [C][OP]
    float* red = smem + (size_t)C * OP;        // [8][OP][64]
    const int tid = threadIdx.x, lane = tid & 63, wv = mn_uniform(tid >> 6), pq = lane & 15, cs = lane >> 4;
    for (int i = tid; i < C * OP; i += 1024) wl[i] = 0.f;
    __syncthreads();
    for (int i = tid; i < C * O; i += 1024) {       // coalesced read of w[o][c], transposed LDS write
        const int o = i / C, c = i - o * C;
        wl[c * OP + o] = w[i];
    }
    const int64_t P = (int64_t)blockIdx.x * 64 + 4 * pq;
    const int64_t Pc = P < NP ? P : 0;
    const int64_t n = Pc / HW;
    const int p = (int)(Pc - n * HW);
    float acc[4][OP];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int o = 0; o < OP; ++o) acc[e][o] = 0.f;
    const int per = (C + SC_WAVES - 1) / SC_WAVES;
    const int c0 = wv * per, c1 = (c0 + per) < C ? (c0 + per) : C;
    const char* src = a + n * C * HW + p;
    __syncthreads();
    auto add = [&](uint32_t v, int c) {
        float s0, s1, s2, s3;
        if (ENC) { s0 = (float)(v & 0xffu); s1 = (float)((v >> 8) & 0xffu); s2 = (float)((v >> 16) & 0xffu); s3 = (float)(v >> 24); }
        else { s0 = (v & 0x80u) ? -1.f : 1.f; s1 = (v & 0x8000u) ? -1.f : 1.f; s2 = (v & 0x800000u) ? -1.f : 1.f; s3 = (v & 0x80000000u) ? -1.f : 1.f; }
#pragma unroll
        for (int o4 = 0; o4 < OP; o4 += 4) {
            const float4 w4 = *reinterpret_cast<const float4*>(wl + c * OP + o4);
            const float wv_[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc[0][o4 + k] += wv_[k] * s0; acc[1][o4 + k] += wv_[k] * s1; acc[2][o4 + k] += wv_[k] * s2; acc[3][o4 + k] += wv_[k] * s3;
            }
        }
    };
    for (int cb = c0 + cs; cb < c1; cb += 64) {               // 16 channels of this lane per trip (c = cb + 4u)
        uint32_t v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int c = cb + 4 * u;
            v[u] = *reinterpret_cast<const uint32_t*>(src + (int64_t)(c < c1 ? c : c1 - 1) * HW);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (cb + 4 * u < c1) add(v[u], cb + 4 * u);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int o = 0; o < OP; ++o) {
            float t = acc[e][o];
            t += __shfl_xor(t, 16, 64); t += __shfl_xor(t, 32, 64);
            acc[e][o] = t;
        }
    if (wv >= 8 && cs == 0) {
#pragma unroll
        for (int o = 0; o < OP; ++o)
            *reinterpret_cast<float4*>(red + ((wv - 8) * OP + o) * 64 + 4 * pq) = make_float4(acc[0][o], acc[1][o], acc[2][o], acc[3][o]);
    }
    __syncthreads();
    if (wv < 8 && cs == 0) {
#pragma unroll
        for (int o = 0; o < OP; ++o) {
            float4* r = reinterpret_cast<float4*>(red + (wv * OP + o) * 64 + 4 * pq);
            const float4 t = *r;
            *r = make_float4(acc[0][o] + t.x, acc[1][o] + t.y, acc[2][o] + t.z, acc[3][o] + t.w);
        }
    }
    __syncthreads();
    for (int i = tid; i < O * 16; i += 1024) {
        const int o = i >> 4, q = i & 15;                       // quad q of the block
        const int64_t Pq = (int64_t)blockIdx.x * 64 + 4 * q;
        if (Pq < NP) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k = 0; k < 8; ++k) {                        // fixed order
                const float4 t = *reinterpret_cast<const float4*>(red + (k * OP + o) * 64 + 4 * q);
                v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
            }
            const float bb = bias ? bias[o] : 0.f;
            const int64_t nq = Pq / HW;
            if (ENC) { v.x *= ascale; v.y *= ascale; v.z *= ascale; v.w *= ascale; }
            *reinterpret_cast<float4*>(y + (nq * O + o) * HW + (Pq - nq * HW)) = make_float4(v.x + bb, v.y + bb, v.z + bb, v.w + bb);
        }
    }
}
// its backward-data: dx[n][c][p] = sum_o w[o][c] * gy[n][o][p]  (O terms); thread = one float4 of 4 pixels of one channel
__global__ __launch_bounds__(256) void k_sconv_dgrad(const float* __restrict__ gy, const float* __restrict__ w, float* __restrict__ dx, int C, int HW, int O) {
    __shared__ float gs[SC_MAXO * 64];
    const int tid = threadIdx.x, q4 = tid & 15, cl = tid >> 4;
    const int chunks = (HW + 63) >> 6, cblks = (C + 63) >> 6;
    uint32_t b = blockIdx.x;
    const int cb = b % cblks; b /= cblks;
    const int ch = b % chunks;
    const int n = b / chunks;
    const int p0 = ch * 64;
    for (int i = tid; i < O * 64; i += 256) {
        const int o = i >> 6, q = i & 63;
        gs[i] = (p0 + q < HW) ? gy[((int64_t)n * O + o) * HW + p0 + q] : 0.f;
    }
    __syncthreads();
    if (p0 + q4 * 4 >= HW) return;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int c = cb * 64 + pass * 16 + cl;
        if (c < C) {
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int o = 0; o < O; ++o) {
                const float wv = w[(int64_t)o * C + c];
                const float4 g4 = *reinterpret_cast<const float4*>(gs + o * 64 + q4 * 4);
                r.x += wv * g4.x; r.y += wv * g4.y; r.z += wv * g4.z; r.w += wv * g4.w;
            }
            *reinterpret_cast<float4*>(dx + ((int64_t)n * C + c) * HW + p0 + q4 * 4) = r;
        }
    }
}
extern "C" int mn_signconv1x1_small_supported(int64_t C, int64_t HW, int64_t O) {
    if (!(O >= 1 && O <= SC_MAXO && C >= 4 && HW % 4 == 0)) return 0;
    const int64_t OP = (O + 3) / 4 * 4;
    return (C * OP + 8 * OP * 64) * 4 <= 128 * 1024;          // the weight image and the partial sums live in LDS
}
static int sconv_fwd(const void* a, int enc, float ascale, const float* w, const float* bias, float* y, int64_t N, int64_t C, int64_t HW, int64_t O,
                     mn_stream_t stream, const char* what) {
    if (!a || !w || !y || N <= 0 || !mn_signconv1x1_small_supported(C, HW, O)) MN_FAIL(MN_EINVAL, "%s: needs O <= 16, HW %% 4 == 0", what);
    hipStream_t s = (hipStream_t)stream;
    const int64_t NP = N * HW, nb = (NP + 63) / 64;
    if (nb > 0x7fffffff || (((uintptr_t)a) & 3) || !aligned16(y)) MN_FAIL(MN_EINVAL, "%s: too many blocks / misaligned tensor", what);
    mn_set_last_kernel(enc ? "k_sconv_fwd<code8>" : "k_sconv_fwd"); mn_prof_bytes((double)N * C * HW + 4.0 * N * O * HW); mn_prof_begin(s);
    const int OP = (int)((O + 3) / 4 * 4);
    const size_t lds = ((size_t)C * OP + (size_t)8 * OP * 64) * 4;
    if (lds > 128 * 1024) MN_FAIL(MN_ENOTSUP, "%s: too many input channels for the LDS weight image", what);
#define SC_LAUNCH(OPV, EV) { raise_lds_limit((const void*)k_sconv_fwd<OPV, EV>, lds); \
        hipLaunchKernelGGL((k_sconv_fwd<OPV, EV>), dim3((unsigned)nb), dim3(1024), lds, s, (const char*)a, w, bias, y, (int)C, (int)HW, (int)O, NP, ascale); }
    if (enc) { if (OP == 4) SC_LAUNCH(4, 1) else if (OP == 8) SC_LAUNCH(8, 1) else if (OP == 12) SC_LAUNCH(12, 1) else SC_LAUNCH(16, 1) }
    else { if (OP == 4) SC_LAUNCH(4, 0) else if (OP == 8) SC_LAUNCH(8, 0) else if (OP == 12) SC_LAUNCH(12, 0) else SC_LAUNCH(16, 0) }
#undef SC_LAUNCH
    mn_prof_end(s);
    MN_CHECK_LAUNCH(what);
    return MN_OK;
}
extern "C" int mn_signconv1x1_small_fwd(const int8_t* a, const float* w, const float* bias, float* y, int64_t N, int64_t C, int64_t HW, int64_t O, mn_stream_t stream) {
    return sconv_fwd(a, 0, 1.f, w, bias, y, N, C, HW, O, stream, "mn_signconv1x1_small_fwd");
}
extern "C" int mn_codeconv1x1_small_fwd(const uint8_t* codes, int a_bits, const float* w, const float* bias, float* y, int64_t N, int64_t C, int64_t HW, int64_t O,
                                        mn_stream_t stream) {
    if (a_bits < 2 || a_bits > 8) MN_FAIL(MN_EINVAL, "mn_codeconv1x1_small_fwd: 2 ... 8 bit codes");
    return sconv_fwd(codes, 1, dorefa_scale(a_bits), w, bias, y, N, C, HW, O, stream, "mn_codeconv1x1_small_fwd");
}
extern "C" int mn_conv1x1_small_bwd_data(const float* gy, const float* w, float* dx, int64_t N, int64_t C, int64_t HW, int64_t O, mn_stream_t stream) {
    if (!gy || !w || !dx || N <= 0 || !mn_signconv1x1_small_supported(C, HW, O) || !aligned16(dx)) MN_FAIL(MN_EINVAL, "mn_conv1x1_small_bwd_data: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int64_t nb = N * ((HW + 63) / 64) * ((C + 63) / 64);
    if (nb > 0x7fffffff) MN_FAIL(MN_EINVAL, "mn_conv1x1_small_bwd_data: too many blocks");
    mn_set_last_kernel("k_sconv_dgrad"); mn_prof_bytes(4.0 * N * C * HW + 4.0 * N * O * HW); mn_prof_begin(s);
    hipLaunchKernelGGL(k_sconv_dgrad, dim3((unsigned)nb), dim3(256), 0, s, gy, w, dx, (int)C, (int)HW, (int)O);
    mn_prof_end(s);
    MN_CHECK_LAUNCH("mn_conv1x1_small_bwd_data");
    return MN_OK;
}

// ---------------------------------------------------------------- BatchNorm + sign backward on the one-byte conv stash
// For a binary block the conv output is y = alpha[o] * acc + bias with acc an exact integer of known parity (qgemm_sign.hip); the fused
// forward stashes h = (acc + nnz[o]) / 2 as ONE BYTE per element.  The backward of the BatchNorm + sign then is a pure streaming pass
// over (da, h) -- no convolution recompute, a quarter of the bytes of reading y as fp32.  chan = the [8][C] per-channel constants of
// k_pws_chan_prep: T, flip, L, U (integer thresholds in the acc*flip domain), A, B (zhat = acc*A + B), gi = gamma*invstd, nnz.
// POOL: da is the gradient of the 2x2 / stride-2 max-pool behind the block ([N][C][H/2][W/2]) and `own` the block's output codes.
struct BnhGeom { int N, C, H, W, HW, HW4; FastDiv fd_hw4, fd_w4; int64_t n4; };
template <int POOL>
__device__ __forceinline__ void bnh_load(const BnhGeom& g, int c, uint32_t i, const float* __restrict__ da, const unsigned char* __restrict__ h,
                                         const char* __restrict__ own, float (&gv)[4], float (&acc)[4], const StashNnz& nz9, int64_t& off) {
    const uint32_t n = fd_div(i, g.fd_hw4);
    const uint32_t q = i - n * (uint32_t)g.HW4;              // quad index inside the plane
    off = ((int64_t)n * g.C + c) * g.HW + (int64_t)q * 4;
    const uint32_t hb = *reinterpret_cast<const uint32_t*>(h + off);
    const int W4 = g.W >> 2;
    const uint32_t row = fd_div(q, g.fd_w4);
    float nz[4];
    stash_nnz_quad(nz9, (int)row, (int)(q - row * (uint32_t)W4), g.H, W4, nz);
    acc[0] = 2.f * (float)(hb & 0xffu) - nz[0]; acc[1] = 2.f * (float)((hb >> 8) & 0xffu) - nz[1];
    acc[2] = 2.f * (float)((hb >> 16) & 0xffu) - nz[2]; acc[3] = 2.f * (float)(hb >> 24) - nz[3];
    if (!POOL) {
        const float4 g4 = *reinterpret_cast<const float4*>(da + off);
        gv[0] = g4.x; gv[1] = g4.y; gv[2] = g4.z; gv[3] = g4.w;
    } else {
        const uint32_t w = (q - row * (uint32_t)W4) * 4u;
        const uint32_t hbit = row & 1u;
        const int64_t pb = ((int64_t)n * g.C + c) * (g.HW >> 2) + (int64_t)(row >> 1) * (g.W >> 1) + (w >> 1);
        const float2 g2 = *reinterpret_cast<const float2*>(da + pb);
        const int64_t cb = ((int64_t)n * g.C + c) * g.HW + (int64_t)(row & ~1u) * g.W + w;
        const uint32_t r0 = *reinterpret_cast<const uint32_t*>(own + cb), r1 = *reinterpret_cast<const uint32_t*>(own + cb + g.W);
#pragma unroll
        for (int e = 0; e < 2; ++e) {      // first maximum of the window in row-major order (ATen): the first +1, else element 0
            const bool p00 = !((r0 >> (16 * e)) & 0x80u), p01 = !((r0 >> (16 * e + 8)) & 0x80u);
            const bool p10 = !((r1 >> (16 * e)) & 0x80u), p11 = !((r1 >> (16 * e + 8)) & 0x80u);
            const uint32_t win = p00 ? 0u : (p01 ? 1u : (p10 ? 2u : (p11 ? 3u : 0u)));
            const float ge = e ? g2.y : g2.x;
            gv[2 * e] = win == hbit * 2u ? ge : 0.f;
            gv[2 * e + 1] = win == hbit * 2u + 1u ? ge : 0.f;
        }
    }
}
template <int POOL>
__global__ __launch_bounds__(256) void k_bnh_partial(const BnhGeom g, const float* __restrict__ da, const unsigned char* __restrict__ h,
                                                     const char* __restrict__ own, const float* __restrict__ chan, double* __restrict__ part) {
    __shared__ double scd[16];
    const int c = blockIdx.x, sp = blockIdx.y, S = gridDim.y, C = g.C;
    const float fl = chan[C + c], L = chan[2 * C + c], U = chan[3 * C + c], A = chan[4 * C + c], B = chan[5 * C + c];
    const StashNnz nnz = stash_nnz_load(chan, C, c);
    double s1 = 0.0, s2 = 0.0;
    float gv_[2][4], acc_[2][4];
    int64_t off_[2];
#define BNHP_LOAD(k, idx) bnh_load<POOL>(g, c, (uint32_t)(idx), da, h, own, gv_[k], acc_[k], nnz, off_[k]);
#define BNHP_FIN(k) { float t1 = 0.f, t2 = 0.f;                                              \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                      \
            const float u = acc_[k][e] * fl;                                                 \
            const float dz = (u >= L && u <= U) ? gv_[k][e] : 0.f;                           \
            t1 += dz;                                                                        \
            t2 += dz * fmaf(acc_[k][e], A, B);                                               \
        }                                                                                    \
        s1 += (double)t1; s2 += (double)t2; }
    MN_STREAM_2(i, (int64_t)sp * 256 + threadIdx.x, (int64_t)S * 256, g.n4, BNHP_LOAD, BNHP_FIN)
#undef BNHP_LOAD
#undef BNHP_FIN
    s1 = block_reduce(s1, OpAddD(), 0.0, scd);
    s2 = block_reduce(s2, OpAddD(), 0.0, scd);
    if (threadIdx.x == 0) { part[((int64_t)c * S + sp) * 2] = s1; part[((int64_t)c * S + sp) * 2 + 1] = s2; }
}
// The pooled partial sums visit every 2x2 WINDOW once instead of every element: only the window's first maximum receives gradient, so
// a thread takes four windows of one pooled row (a float4 of the pooled gradient, 8 codes and 8 stash bytes of each of the two image rows:
// five loads for 16 elements instead of sixteen) and evaluates the clip mask / zhat for the four receiving pixels only.  W % 8 == 0.
__global__ __launch_bounds__(256) void k_bnh_partial_pool(const BnhGeom g, const float* __restrict__ da, const unsigned char* __restrict__ h,
                                                          const char* __restrict__ own, const float* __restrict__ chan, double* __restrict__ part) {
    __shared__ double scd[16];
    const int c = blockIdx.x, sp = blockIdx.y, S = gridDim.y, C = g.C;
    const float fl = chan[C + c], L = chan[2 * C + c], U = chan[3 * C + c], A = chan[4 * C + c], B = chan[5 * C + c];
    const StashNnz zn = stash_nnz_load(chan, C, c);
    const int W8 = g.W >> 3, Hh = g.H >> 1;
    const int64_t npq = (int64_t)g.N * Hh * W8;                 // pooled quads of this channel
    double s1 = 0.0, s2 = 0.0;
    for (int64_t i = (int64_t)sp * 256 + threadIdx.x; i < npq; i += (int64_t)S * 256) {
        const int64_t t = i / W8;
        const int q = (int)(i - t * W8);
        const int64_t n = t / Hh;
        const int pr = (int)(t - n * Hh);
        const int64_t plane = n * C + c;
        const float4 g4 = *reinterpret_cast<const float4*>(da + plane * (g.HW >> 2) + (int64_t)pr * (g.W >> 1) + 4 * q);
        const int64_t e0 = plane * g.HW + (int64_t)(2 * pr) * g.W + 8 * q;
        const uint64_t o0 = *reinterpret_cast<const uint64_t*>(own + e0), o1 = *reinterpret_cast<const uint64_t*>(own + e0 + g.W);
        const uint64_t h0 = *reinterpret_cast<const uint64_t*>(h + e0), h1 = *reinterpret_cast<const uint64_t*>(h + e0 + g.W);
        const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {       // first maximum of the window in row-major order (ATen): the first +1, else element 0
            const bool p00 = !((o0 >> (16 * w)) & 0x80u), p01 = !((o0 >> (16 * w + 8)) & 0x80u);
            const bool p10 = !((o1 >> (16 * w)) & 0x80u), p11 = !((o1 >> (16 * w + 8)) & 0x80u);
            const bool lower = !p00 && !p01 && (p10 || p11);
            const bool right = p00 ? false : (p01 ? true : (p10 ? false : p11));
            const uint64_t hr = lower ? h1 : h0;
            const float hv = (float)((uint32_t)(hr >> (16 * w + (right ? 8 : 0))) & 0xffu);
            const float acc = 2.f * hv - stash_nnz_px(zn, 2 * pr + (lower ? 1 : 0), 8 * q + 2 * w + (right ? 1 : 0), g.H, g.W);
            const float u = acc * fl;
            const float dz = (u >= L && u <= U) ? gv[w] : 0.f;
            t1 += dz;
            t2 += dz * fmaf(acc, A, B);
        }
        s1 += (double)t1; s2 += (double)t2;
    }
    s1 = block_reduce(s1, OpAddD(), 0.0, scd);
    s2 = block_reduce(s2, OpAddD(), 0.0, scd);
    if (threadIdx.x == 0) { part[((int64_t)c * S + sp) * 2] = s1; part[((int64_t)c * S + sp) * 2 + 1] = s2; }
}
template <int POOL>
__global__ __launch_bounds__(256) void k_bnh_apply(const BnhGeom g, const float* __restrict__ da, const unsigned char* __restrict__ h,
                                                   const char* __restrict__ own, const float* __restrict__ chan, const float* __restrict__ sums,
                                                   int training, float* __restrict__ dy) {
    const int c = blockIdx.x, sp = blockIdx.y, S = gridDim.y, C = g.C;
    const float fl = chan[C + c], L = chan[2 * C + c], U = chan[3 * C + c], A = chan[4 * C + c], B = chan[5 * C + c], gi = chan[6 * C + c];
    const StashNnz nnz = stash_nnz_load(chan, C, c);
    float k1 = 0.f, k2 = 0.f;
    if (training) { const float n = (float)g.N * (float)g.HW; k1 = sums[c] / n; k2 = sums[C + c] / n; }
    float gv_[2][4], acc_[2][4];
    int64_t off_[2];
#define BNHA_LOAD(k, idx) bnh_load<POOL>(g, c, (uint32_t)(idx), da, h, own, gv_[k], acc_[k], nnz, off_[k]);
#define BNHA_FIN(k) { float r[4];                                                            \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                      \
            const float u = acc_[k][e] * fl;                                                 \
            const float dz = (u >= L && u <= U) ? gv_[k][e] : 0.f;                           \
            r[e] = gi * (dz - k1 - fmaf(acc_[k][e], A, B) * k2);                             \
        }                                                                                    \
        *reinterpret_cast<float4*>(dy + off_[k]) = make_float4(r[0], r[1], r[2], r[3]); }
    MN_STREAM_2(i, (int64_t)sp * 256 + threadIdx.x, (int64_t)S * 256, g.n4, BNHA_LOAD, BNHA_FIN)
#undef BNHA_LOAD
#undef BNHA_FIN
}
static int bnh_common(int64_t N, int64_t C, int64_t H, int64_t W, int pooled, const void* da, const void* h, const void* own, BnhGeom* g, const char* what) {
    const int64_t HW = H * W;
    if (N <= 0 || C <= 0 || HW <= 0 || W % 4 || N * (HW / 4) >= ((int64_t)1 << 31)) MN_FAIL(MN_EINVAL, "%s: bad shape (W must be a multiple of 4)", what);
    if (pooled && ((H & 1) || !own || (((uintptr_t)own) & 3))) MN_FAIL(MN_EINVAL, "%s: pooled gradient needs even H and the block's output codes", what);
    if (!da || !h || (((uintptr_t)h) & 3) || (((uintptr_t)da) & (pooled ? 7 : 15))) MN_FAIL(MN_EINVAL, "%s: null / misaligned tensor", what);
    g->N = (int)N; g->C = (int)C; g->H = (int)H; g->W = (int)W; g->HW = (int)HW; g->HW4 = (int)(HW / 4);
    g->fd_hw4 = make_fastdiv((uint32_t)g->HW4); g->fd_w4 = make_fastdiv((uint32_t)(W / 4)); g->n4 = N * (HW / 4);
    return MN_OK;
}
static int bnh_split(const BnhGeom& g) {
    int64_t S = (2048 + g.C - 1) / g.C;
    const int64_t maxS = (g.n4 + 255) / 256;
    if (S > maxS) S = maxS;
    if (S > BNS_SPLIT) S = BNS_SPLIT;
    if (S < 1) S = 1;
    return (int)S;
}
// sums [2][C] = {sum dz, sum dz*zhat}, dgamma, dbeta (nullable).  own == NULL: da is full size; else da is the pooled gradient.
extern "C" int mn_bnh_bwd_sums(const float* da, const uint8_t* h, const int8_t* own, const float* chan, int64_t N, int64_t C, int64_t H, int64_t W,
                               float* dgamma, float* dbeta, float* sums, float* ws, mn_stream_t stream) {
    BnhGeom g;
    int rc = bnh_common(N, C, H, W, own != nullptr, da, h, own, &g, "mn_bnh_bwd_sums");
    if (rc) return rc;
    if (!chan || !sums || !ws || (((uintptr_t)ws) & 7)) MN_FAIL(MN_EINVAL, "mn_bnh_bwd_sums: null / misaligned argument");
    hipStream_t s = (hipStream_t)stream;
    const int S = bnh_split(g);
    const double nel = (double)N * C * H * W;
    // pooled fast path: W % 8 == 0, 8-byte aligned rows
    const bool pool_fast = own && W % 8 == 0 && !(((uintptr_t)h) & 7) && !(((uintptr_t)own) & 7) && !(((uintptr_t)da) & 15);
    mn_set_last_kernel(pool_fast ? "k_bnh_partial_pool" : (own ? "k_bnh_partial<1>" : "k_bnh_partial<0>")); mn_prof_bytes((own ? 3.0 : 5.0) * nel); mn_prof_begin(s);
    if (pool_fast) hipLaunchKernelGGL(k_bnh_partial_pool, dim3((unsigned)C, (unsigned)S), dim3(256), 0, s, g, da, (const unsigned char*)h, (const char*)own, chan, (double*)ws);
    else if (own) hipLaunchKernelGGL(k_bnh_partial<1>, dim3((unsigned)C, (unsigned)S), dim3(256), 0, s, g, da, (const unsigned char*)h, (const char*)own, chan, (double*)ws);
    else hipLaunchKernelGGL(k_bnh_partial<0>, dim3((unsigned)C, (unsigned)S), dim3(256), 0, s, g, da, (const unsigned char*)h, (const char*)nullptr, chan, (double*)ws);
    mn_prof_end(s);
    const BnsGeom bg = bns_geom(N, C, H * W);
    hipLaunchKernelGGL(k_bns_final_bwd, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, s, bg, (const double*)ws, S, dgamma, dbeta, sums);
    MN_CHECK_LAUNCH("mn_bnh_bwd_sums");
    return MN_OK;
}
// ... when the producer of d a already left the partial sums (mn_conv2d_bwd_bnh_up: part [C][splits][2] doubles): only the fixed-order finish
extern "C" int mn_bnh_bwd_sums_final(const double* part, int32_t splits, int64_t N, int64_t C, int64_t H, int64_t W, float* dgamma, float* dbeta, float* sums,
                                     mn_stream_t stream) {
    if (!part || !sums || splits < 1 || N <= 0 || C <= 0 || H <= 0 || W <= 0 || (((uintptr_t)part) & 7)) MN_FAIL(MN_EINVAL, "mn_bnh_bwd_sums_final: bad arguments");
    const BnsGeom bg = bns_geom(N, C, H * W);
    hipLaunchKernelGGL(k_bns_final_bwd, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, (hipStream_t)stream, bg, part, (int)splits, dgamma, dbeta, sums);
    MN_CHECK_LAUNCH("mn_bnh_bwd_sums_final");
    return MN_OK;
}
// dy = d loss / d y from (da, h) and the sums
extern "C" int mn_bnh_bwd_apply(const float* da, const uint8_t* h, const int8_t* own, const float* chan, const float* sums, int64_t N, int64_t C, int64_t H,
                                int64_t W, int training, float* dy, mn_stream_t stream) {
    BnhGeom g;
    int rc = bnh_common(N, C, H, W, own != nullptr, da, h, own, &g, "mn_bnh_bwd_apply");
    if (rc) return rc;
    if (!chan || !sums || !dy || !aligned16(dy)) MN_FAIL(MN_EINVAL, "mn_bnh_bwd_apply: null / misaligned argument");
    hipStream_t s = (hipStream_t)stream;
    const int S = bnh_split(g);
    const double nel = (double)N * C * H * W;
    mn_set_last_kernel(own ? "k_bnh_apply<1>" : "k_bnh_apply<0>"); mn_prof_bytes((own ? 7.0 : 9.0) * nel); mn_prof_begin(s);
    if (own) hipLaunchKernelGGL(k_bnh_apply<1>, dim3((unsigned)C, (unsigned)S), dim3(256), 0, s, g, da, (const unsigned char*)h, (const char*)own, chan, sums, training, dy);
    else hipLaunchKernelGGL(k_bnh_apply<0>, dim3((unsigned)C, (unsigned)S), dim3(256), 0, s, g, da, (const unsigned char*)h, (const char*)nullptr, chan, sums, training, dy);
    mn_prof_end(s);
    MN_CHECK_LAUNCH("mn_bnh_bwd_apply");
    return MN_OK;
}
