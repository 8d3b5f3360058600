// Shared device/host helpers for libmicronet_hip (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/micronet_hip.h"

typedef float f32x4 __attribute__((vector_size(16)));

// ---------------------------------------------------------------- errors
void mn_set_error(const char* fmt, ...);
#define MN_FAIL(code, ...)        \
    do {                          \
        mn_set_error(__VA_ARGS__); \
        return (code);            \
    } while (0)
#define MN_CHECK_LAUNCH(what)                                             \
    do {                                                                  \
        hipError_t e_ = hipGetLastError();                                \
        if (e_ != hipSuccess) MN_FAIL(MN_EHIP, "%s: %s", what, hipGetErrorString(e_)); \
    } while (0)

// ---------------------------------------------------------------- arithmetic shared by all schemes
// round-half-away-from-zero evaluated in fp32 exactly like the reference's
// sign(v) * floor(|v| + 0.5)  (dorefa/quantize.py:13-16, iao/quantize.py:158-160).
// torch.sign: NaN and +-0 map to 0
__device__ __forceinline__ float mn_sign(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }
__device__ __forceinline__ float mn_rha(float v) { return mn_sign(v) * floorf(fabsf(v) + 0.5f); }
// torch.clamp(v, lo, hi) (NaN propagates)
__device__ __forceinline__ float mn_clamp(float v, float lo, float hi) {
    return (v != v) ? v : fminf(fmaxf(v, lo), hi);
}

// ---------------------------------------------------------------- activation quantizers (shared by the streaming
// kernels and by the conv prologue / clip-STE epilogue)
// DoReFa activation, wqaq/dorefa/quantize.py:43-45
__device__ __forceinline__ float dorefa_act_q(float x, float s) {
    float c = mn_clamp(x * 0.1f, 0.f, 1.f);
    return mn_rha(c / s) * s;
}
__device__ __forceinline__ float dorefa_act_grad(float g, float x, float s) {
    float t = x * 0.1f;
    float d = (g * s) / s;
    d = (t >= 0.f && t <= 1.f) ? d : 0.f;   // clamp backward: inclusive at both ends
    return d * 0.1f;
}
// IAO fake-quant, wqaq/iao/quantize.py:227-239 and Round.backward 163-168
__device__ __forceinline__ float iao_fq(float x, float sc, float zp, float qmin, float qmax) {
    float r = mn_rha(x / sc - zp);
    return (mn_clamp(r, qmin, qmax) + zp) * sc;
}
__device__ __forceinline__ float iao_fq_grad(float g, float x, float sc, float zp, float lo, float hi, float qmin, float qmax) {
    float v = x / sc - zp;
    float r = mn_rha(v);
    float d = g * sc;
    d = (r >= qmin && r <= qmax) ? d : 0.f;   // clamp backward
    d = (v > hi || v < lo) ? 0.f : d;          // Round.backward 166-167
    return d / sc;
}
struct IaoRange { float qmin, qmax; };
static inline IaoRange iao_range(int bits, int q_type, int is_act) {
    IaoRange r;
    if (q_type == 0) {
        r.qmin = is_act ? -(float)(1ll << (bits - 1)) : -(float)((1ll << (bits - 1)) - 1);
        r.qmax = (float)((1ll << (bits - 1)) - 1);
    } else {
        r.qmin = 0.f;
        r.qmax = is_act ? (float)((1ll << bits) - 1) : (float)((1ll << bits) - 2);
    }
    return r;
}
static inline float dorefa_scale(int bits) { return (float)(1.0 / (double)((1ll << bits) - 1)); }
__host__ __device__ static inline int aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// exact unsigned division by a runtime-uniform divisor (host-precomputed); valid while n*d < 2^32
struct FastDiv {
    uint32_t d, m;
};
static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    f.m = (d <= 1) ? 0u : (uint32_t)((((uint64_t)1 << 32) + d - 1) / d);
    return f;
}
__device__ __forceinline__ uint32_t fd_div(uint32_t n, FastDiv f) { return f.d <= 1 ? n : __umulhi(n, f.m); }

// ---------------------------------------------------------------- wave / block reductions (wave = 64 lanes)
template <typename T, typename Op>
__device__ __forceinline__ T wave_reduce(T v, Op op) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = op(v, __shfl_down(v, o, 64));
    return v;  // valid in lane 0
}
// all threads of a <=1024-thread block get the result; scratch: >= 16 T in LDS
template <typename T, typename Op>
__device__ __forceinline__ T block_reduce(T v, Op op, T identity, T* scratch) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_reduce(v, op);
    __syncthreads();  // scratch may still be read from a previous reduction
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    T r = identity;
    for (int i = 0; i < nw; ++i) r = op(r, scratch[i]);
    return r;
}
struct OpAddD { __device__ __forceinline__ double operator()(double a, double b) const { return a + b; } };
struct OpAddF { __device__ __forceinline__ float operator()(float a, float b) const { return a + b; } };
struct OpMaxF { __device__ __forceinline__ float operator()(float a, float b) const { return (a != a || b != b) ? (a != a ? a : b) : fmaxf(a, b); } };
struct OpMinF { __device__ __forceinline__ float operator()(float a, float b) const { return (a != a || b != b) ? (a != a ? a : b) : fminf(a, b); } };

static inline int mn_grid_for(int64_t n_items, int per_block, int cap) {
    int64_t b = (n_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}
