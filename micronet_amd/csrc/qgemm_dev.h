// Device/host helpers shared by the code-domain kernel files (qgemm_kernels.hip: pointwise; qgemm_kxk.hip: k x k).
#pragma once
#include "qgemm.h"

#include <map>
#include <mutex>

typedef unsigned int u32x2 __attribute__((vector_size(8)));

#define QG_EPI_SCALE_BIAS 0
#define QG_EPI_PLAIN 1
#define QG_EPI_STE 2
#define QG_EPI_H8 3          // forward of a ternary-weight conv on sign codes: out8 = (acc + bias[o]) / 2 as one byte (bias = nnz[o]: the byte stash h)

static inline int qg_roundup(int a, int b) { return (a + b - 1) / b * b; }

// ------------------------------------------------------------------------------------------------
// activation codes in the conv prologue
template <int XMODE>
__device__ __forceinline__ float act_code(float x, const Pro& p, float sc, float zp) {
    if (XMODE == MN_ACTQ_DOREFA) return mn_rha(mn_clamp(x * 0.1f, 0.f, 1.f) / p.s);           // j in [0, 2^a - 1]
    if (XMODE == MN_ACTQ_IAO) return mn_clamp(mn_rha(x / sc - zp), p.qmin, p.qmax) + zp;       // clamp(r) + zp
    return x;
}

// fl(x / s) without the hardware division sequence: with y = RN(1 / s), q0 = RN(x y), r = x - s q0 (exact: one fma), RN(q0 + r y) IS the correctly rounded quotient
// (Markstein's division theorem; no overflow / underflow here: |x / s| <= a few hundred for in-range activations, and out-of-range ones are clamped far from any
// rounding boundary).  3 instructions instead of ~10, bit-identical codes (tests: check_iao_codes_at_boundaries).
__device__ __forceinline__ float mn_div_m(float x, float s, float inv) {
    const float q0 = x * inv;
    const float r = fmaf(-q0, s, x);
    return fmaf(r, inv, q0);
}
__device__ __forceinline__ float iao_code_m(float x, float sc, float inv, float zp, float qmin, float qmax) {
    return mn_clamp(mn_rha(mn_div_m(x, sc, inv) - zp), qmin, qmax) + zp;
}

// iao_fq_grad (common.h) with both IEEE divisions replaced by the three-instruction correctly rounded quotient: bit-identical results for in-range operands at a
// third of the instructions (the clip-STE inside an MFMA kernel's epilogue is VALU-bound)
__device__ __forceinline__ float iao_fq_grad_m(float g, float x, float sc, float inv, float zp, float lo, float hi, float qmin, float qmax) {
    const float v = mn_div_m(x, sc, inv) - zp;
    const float r = mn_rha(v);
    float d = g * sc;
    d = (r >= qmin && r <= qmax) ? d : 0.f;
    d = (v > hi || v < lo) ? 0.f : d;
    return mn_div_m(d, sc, inv);
}

// per-channel fold of the BatchNorm+sign backward for the BNH variants (same algebra as k_bnh_apply, reassociated; `scale` = the weight
// scale the consumer multiplies the operand with)
__device__ __forceinline__ void bnh_fold(const float* __restrict__ chan, const float* __restrict__ sums, int C, int co, int training, float n_f, float scale,
                                         float& hlo, float& hhi, float& G, float& E1, float& E0) {
    const float fl = chan[C + co], L = chan[2 * C + co], U = chan[3 * C + co], A = chan[4 * C + co], B = chan[5 * C + co], gi = chan[6 * C + co], nnz = chan[7 * C + co];
    const float k1 = training ? sums[co] / n_f : 0.f, k2 = training ? sums[C + co] / n_f : 0.f;
    // L <= (2h - nnz)*flip <= U   (L, U, nnz integers; 2h - nnz has the parity of every admissible value)
    if (fl > 0.f) { hlo = ceilf((L + nnz) * 0.5f); hhi = floorf((U + nnz) * 0.5f); }
    else { hlo = ceilf((nnz - U) * 0.5f); hhi = floorf((nnz - L) * 0.5f); }
    G = gi * scale;
    E1 = -gi * k2 * 2.f * A * scale;
    E0 = -gi * (k1 + k2 * (B - nnz * A)) * scale;
}

// ------------------------------------------------------------------------------------------------
// weight codes: one workgroup (one wave) per padded row; recovers code and scale from the fake-quantised fp32 weights
struct PackParams {
    const float* w;        // [G*Mg][Cg*T]
    uint16_t* codes;
    float* scale_out;      // per out-channel scale (fwd: rowscale [G][Mpad]; bwd: kscale [G][Mgp])
    const float* scale_in; // IAO
    int G, Mg, Cg, T, KW;
    int mode, bits, per_channel;
    int transpose;         // 0: codes[(g*Mpad + m)*T*Cgp + tap*Cgp + c]   1: codes[((g*Cpad + c)*T + tapflip)*Mgp + m]
    int Mpad, Cgp, Cpad, Mgp;
};

void qg_launch_pack(const PackParams& p, int grid, hipStream_t s);

// fixed-order fp64 reduction of Z partial dw tiles [Z][G][Mgw][Cgw] (+ dbias [Z][G][Mgw]); dw = sum * ascale (or * qp[0])
void qg_launch_wgrad_reduce(const float* part, const float* dbpart, float* dw, float* db, int Z, int G, int Mg, int Cg, int Mgw, int Cgw,
                            float ascale, const float* qp, hipStream_t s);

// k x k kernels (qgemm_kxk.hip)
int kk_supported(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, int which);
int64_t kk_ws_bytes(const mn_conv_geom* g, int which);
int kk_fwd(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, const float* x, const float* w, const float* bias, float* y,
           void* ws, int64_t ws_bytes, hipStream_t s);
int kk_bwd_data(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, const float* gy, const float* w, const float* x, float* dx,
                void* ws, int64_t ws_bytes, hipStream_t s);
int kk_bwd_weight(const mn_conv_geom* g, const mn_actq* aq, const float* gy, const float* x, float* dw, float* dbias, void* ws,
                  int64_t ws_bytes, hipStream_t s);

// k x k forward on sign codes that writes the byte stash h instead of y (qgemm_kxk.hip); info: what k_pws_final_fwd / k_pws_chan_prep need
struct KkH8Info { const uint16_t* codes; const float* rowscale; int Mpad, Kp, K; };
int kk_h8_supported(const mn_conv_geom* g, const mn_wq* wq);
int64_t kk_h8_ws_bytes(const mn_conv_geom* g);
int kk_h8_mpad(const mn_conv_geom* g);
int kk_fwd_h8(const mn_conv_geom* g, const mn_wq* wq, const int8_t* x, const float* w, const float* nnzf, uint8_t* h, void* ws, int64_t ws_bytes,
              hipStream_t s, KkH8Info* info);

// 3 x 3 backward-weight on sign codes (qgemm_k3s.hip)
int k3s_wgrad_supported(const mn_conv_geom* g);
int64_t k3s_wgrad_ws_bytes(const mn_conv_geom* g);
// 3 x 3 forward of a ternary / binary-weight layer on sign codes writing the byte stash + statistics partials [parts][O][2] (qgemm_k3s.hip)
int k3s_fwd_supported(const mn_conv_geom* g, const mn_wq* wq);
int k3s_fwd_parts(const mn_conv_geom* g, const mn_wq* wq);
int k3s_fwd_h8(const mn_conv_geom* g, const mn_wq* wq, const int8_t* x, const float* w, const float* nnz9, uint8_t* h, double* part, hipStream_t s);
// 3 x 3 backward-data of a ternary / binary-weight layer (qgemm_k3s.hip); no clip-STE epilogue
int k3s_dgrad_supported(const mn_conv_geom* g, const mn_wq* wq);
int k3s_bwd_data(const mn_conv_geom* g, const mn_wq* wq, const float* gy, const float* w, float* dx, hipStream_t s);
int k3s_bwd_weight(const mn_conv_geom* g, const float* gy, const int8_t* x, float* dw, float* dbias, void* ws, int64_t ws_bytes, hipStream_t s);
// the k-bit activation-code (MN_ACTQ_CODE8) variants of the staged-image kernels (qgemm_k3s.hip, qgemm_sign.hip) and the stats / constants launch of qact_kernels.hip
int k3s_wgrad_code8_supported(const mn_conv_geom* g, int a_bits);
int k3s_bwd_weight_code8(const mn_conv_geom* g, const float* gy, const uint8_t* x, int a_bits, float ascale, float* dw, float* dbias, void* ws, int64_t ws_bytes, hipStream_t s);
int k3s_fwd16_supported(const mn_conv_geom* g, const mn_wq* wq);
int k3s_fwd16_parts(const mn_conv_geom* g, const mn_wq* wq);
int k3s_fwd_h16(const mn_conv_geom* g, const mn_wq* wq, const uint8_t* x, const float* w, int16_t* h16, int wide, double* part, hipStream_t s);
int pws_wgrad_code8_supported(const mn_conv_geom* g);
int pws_bwd_weight_code8(const mn_conv_geom* g, const float* gy, const uint8_t* x, float ascale, float* dw, float* dbias, void* ws, int64_t ws_bytes, hipStream_t s);
void qa_launch_stats_prep(const double* part, int CB, int G, int Mpad, int Mr, const float* rowscale, float ascale, const float* bias, double n, float eps,
                          float momentum, int training, float* running_mean, float* running_var, float* save, int Cout, const float* gamma, const float* beta,
                          float* chan, long long* nbt, hipStream_t s);

// dense (groups = 1, C and O multiples of 64) layers on k-bit activation codes (qgemm_dense.hip): the ResNet family
int qd_fwd_supported(const mn_conv_geom* g, const mn_wq* wq, int a_bits);
int qd_stash32(const mn_conv_geom* g, const mn_wq* wq, int a_bits);
int64_t qd_fwd_ws_bytes(const mn_conv_geom* g);
int qd_fwd_stash(const mn_conv_geom* g, const mn_wq* wq, const uint8_t* x, int a_bits, const float* w, void* stash, void* ws, int64_t ws_bytes, hipStream_t s,
                 const double** parts, int* nparts, float** rowscale);
int qd_dgrad_supported(const mn_conv_geom* g, const mn_wq* wq);      // covered by qd_bwd_data OR by the generic backward-data
int qd_dgrad_native(const mn_conv_geom* g, const mn_wq* wq);         // covered by qd_bwd_data itself
int64_t qd_dgrad_ws_bytes(const mn_conv_geom* g);
int qd_bwd_data(const mn_conv_geom* g, const mn_wq* wq, const float* gy, const float* w, float* dx, void* ws, int64_t ws_bytes, hipStream_t s);
int qd_wgrad_supported(const mn_conv_geom* g, int a_bits);
int64_t qd_wgrad_ws_bytes(const mn_conv_geom* g);
int qd_bwd_weight(const mn_conv_geom* g, const float* gy, const uint8_t* x, float ascale, float* dw, void* ws, int64_t ws_bytes, hipStream_t s);
int qd_bwd_weight_ex(const mn_conv_geom* g, const float* gy, const uint8_t* x, int xsgn, float ascale, const float* ascale_dev, float* dw, void* ws, int64_t ws_bytes,
                     hipStream_t s);
// IAO layers (symmetric 2..8-bit activation / weight quantizers) on the dense kernels: which = 0 forward, 1 backward-data, 2 backward-weight (wq unused)
int qd_iao_supported(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, int which);
int64_t qd_iao_ws_bytes(const mn_conv_geom* g, int which);
int qd_iao_fwd(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, const float* x, const float* w, const float* bias, float* y, void* ws, int64_t ws_bytes,
               hipStream_t s);
int qd_iao_dx_add_supported(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq);
int qd_iao_bwd_data(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, const float* gy, const float* w, const float* x, float* dx, void* ws, int64_t ws_bytes,
                    hipStream_t s);
int qd_iao_bwd_weight(const mn_conv_geom* g, const mn_actq* aq, const float* gy, const float* x, float* dw, float* dbias, void* ws, int64_t ws_bytes, hipStream_t s);

void qa_launch_stats_prep_const(const double* part, int CB, int Cout, float wscale, float ascale, const float* bias, double n, float eps, float momentum, int training,
                                float* running_mean, float* running_var, float* save, const float* gamma, const float* beta, float* chan, long long* nbt, hipStream_t s);

static inline int aq_codeable(const mn_actq* aq, int need_exact_x) {
    (void)need_exact_x;   // real-valued x is handled exactly by term splitting (zero terms are skipped)
    if (!aq || aq->mode == MN_ACTQ_NONE) return 1;
    if (aq->mode == MN_ACTQ_DOREFA) return aq->bits >= 2 && aq->bits <= 8;
    if (aq->mode == MN_ACTQ_IAO) return aq->bits >= 2 && aq->bits <= 8 && aq->q_type == 0 && aq->qp;
    if (aq->mode == MN_ACTQ_SIGN8) return 1;
    // MN_ACTQ_CODE8 is NOT codeable for the generic kernels (they would read the bytes as fp32): the entry points that read codes test for it
    return 0;
}
static inline int wq_codeable(const mn_wq* wq) {
    if (!wq) return 0;
    if (wq->mode == MN_WQ_TERNARY) return 1;
    if (wq->mode == MN_WQ_DOREFA) return wq->bits >= 2 && wq->bits <= 8;
    if (wq->mode == MN_WQ_IAO) return wq->bits >= 2 && wq->bits <= 8 && wq->q_type == 0 && wq->scale;
    return 0;
}

// raise a kernel's dynamic-LDS limit once per (kernel, size high-water mark): not per launch, so that nothing but kernel
// launches is issued while a HIP graph is being captured (the training step is captured after a few eager warm-up steps)
static inline void raise_lds_limit(const void* fn, size_t bytes) {
#ifndef MN_EMULATION
    if (bytes <= 48 * 1024) return;
    static std::mutex mu;
    static std::map<const void*, size_t> done;
    std::lock_guard<std::mutex> lk(mu);
    size_t& have = done[fn];
    if (have >= bytes) return;
    (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    have = bytes;
#else
    (void)fn; (void)bytes;
#endif
}
static inline void fill_pack(PackParams& k, const mn_wq* wq, const float* w, void* ws, int64_t off_codes, int64_t off_scale) {
    k.w = w; k.codes = (uint16_t*)((char*)ws + off_codes); k.scale_out = (float*)((char*)ws + off_scale);
    k.mode = wq->mode; k.bits = wq->bits > 0 ? wq->bits : 8; k.per_channel = wq->per_channel; k.scale_in = wq->scale;
}


static const size_t QG_LDS_CAP = 64 * 1024;
