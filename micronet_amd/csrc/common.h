// Shared device/host helpers for libmicronet_hip (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include <stdio.h>
#include <string.h>

#include "../../include/micronet_hip.h"

typedef float f32x4 __attribute__((vector_size(16)));

// ---------------------------------------------------------------- errors
void mn_set_error(const char* fmt, ...);
// name of the dominant kernel the last conv entry point launched on this thread (read back by mn_last_kernel())
void mn_set_last_kernel(const char* fmt, ...);
// optional HIP-event bracket around the MAIN kernel of the next conv entry point (armed by mn_profile_next)
void mn_prof_bytes(double nbytes);     // designed HBM bytes of the main kernel about to be launched (read + written once); resets the flop count
void mn_prof_flops(double nflops);     // algorithmic FLOPs (2 x MACs) of that kernel -- the matrix-bound kernels (qgemm_dense.hip); call AFTER mn_prof_bytes
// tuning / A-B knobs from the environment, read ONCE per process and call site (the planners run on every launch)
#define MN_ENV(name) ([]() -> const char* { static const char* v_ = getenv(name); return v_; }())
void mn_prof_begin(hipStream_t s);
void mn_prof_end(hipStream_t s);
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


// ---------------------------------------------------------------- bf16 MFMA (v_mfma_f32_16x16x32_bf16) plumbing
// A/B fragments are 8 bf16 per lane packed into 4 dwords (element e in dword e/2, low half first):
//   A[i = lane&15][k = 8*(lane>>4) + e]     B[k = 8*(lane>>4) + e][j = lane&15]     D[row = 4*(lane>>4) + reg][col = lane&15]
typedef unsigned int u32x4 __attribute__((vector_size(16)));
#ifdef MN_EMULATION
__device__ __forceinline__ f32x4 mn_mfma_bf16(u32x4 a, u32x4 b, f32x4 c) { return emu_mfma_f32_16x16x32_bf16(a, b, c); }
typedef float f32x16 __attribute__((vector_size(64)));
__device__ __forceinline__ f32x16 mn_mfma32_bf16(u32x4 a, u32x4 b, f32x16 c) { return emu_mfma_f32_32x32x16_bf16(a, b, c); }
typedef int i32x4 __attribute__((vector_size(16)));
__device__ __forceinline__ i32x4 mn_mfma_i8(u32x4 a, u32x4 b, i32x4 c) { return emu_mfma_i32_16x16x64_i8(a, b, c); }
__device__ __forceinline__ int mn_wave_any(int pred) { return emu_wave_any(pred); }
__device__ __forceinline__ unsigned mn_f2u(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
__device__ __forceinline__ float mn_u2f(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
#else
typedef __bf16 mn_bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mn_mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mn_bf16x8, a), __builtin_bit_cast(mn_bf16x8, b),
                                                    __builtin_bit_cast(v4f, c), 0, 0, 0);
    return __builtin_bit_cast(f32x4, r);
}
// v_mfma_f32_32x32x16_bf16: A[i = lane&31][k = 8*(lane>>5) + e], B[k = 8*(lane>>5) + e][j = lane&31], D[row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)][col = lane&31], reg = 0..15:
// 8 MACs per operand byte (the 16x16x32 form: 4) at the same matrix-core rate
typedef float f32x16 __attribute__((vector_size(64)));
__device__ __forceinline__ f32x16 mn_mfma32_bf16(u32x4 a, u32x4 b, f32x16 c) {
    typedef float v16f __attribute__((ext_vector_type(16)));
    v16f r = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mn_bf16x8, a), __builtin_bit_cast(mn_bf16x8, b), __builtin_bit_cast(v16f, c), 0, 0, 0);
    return __builtin_bit_cast(f32x16, r);
}
// v_mfma_i32_16x16x64_i8: signed bytes, A[i = lane&15][k = 16*(lane>>4) + e], B[k = 16*(lane>>4) + e][j = lane&15], D like the bf16 form; exact i32 accumulation
typedef int i32x4 __attribute__((vector_size(16)));
__device__ __forceinline__ i32x4 mn_mfma_i8(u32x4 a, u32x4 b, i32x4 c) {
    typedef int v4i __attribute__((ext_vector_type(4)));
    v4i r = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(v4i, a), __builtin_bit_cast(v4i, b), __builtin_bit_cast(v4i, c), 0, 0, 0);
    return __builtin_bit_cast(i32x4, r);
}
__device__ __forceinline__ int mn_wave_any(int pred) { return __builtin_amdgcn_ballot_w64(pred != 0) != 0ull; }
__device__ __forceinline__ unsigned mn_f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float mn_u2f(unsigned u) { return __uint_as_float(u); }
#endif
// a value that is the same in every lane of the wave, told to the compiler (scalar registers / scalar loads downstream)
#ifdef MN_EMULATION
__device__ __forceinline__ int mn_uniform(int v) { return v; }
#else
__device__ __forceinline__ int mn_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
#endif
// scheduling fence: no instruction may be moved across it (keeps software-pipelined loads from being hoisted en bloc)
#ifdef MN_EMULATION
#define MN_SCHED_FENCE() do { } while (0)
#define MN_SETPRIO(n) do { } while (0)
#else
#define MN_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define MN_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
#endif
// wave-level hand-over of LDS data between the lanes of ONE wave (no block barrier): the hardware executes a wave's DS operations in
// order, so only the compiler (and the emulator's lane scheduler) has to be told
#ifdef MN_EMULATION
#define MN_WAVE_SYNC() emu::yield(emu::WAIT_WAVE)
__device__ __forceinline__ uint32_t mn_alignbyte(uint32_t hi, uint32_t lo, int sh) { return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (8 * sh)); }
__device__ __forceinline__ uint32_t mn_perm(uint32_t a, uint32_t b, uint32_t sel) {      // v_perm_b32: bytes 0-3 = b, 4-7 = a, 0x0c = 0x00
    uint32_t r = 0;
    for (int i = 0; i < 4; ++i) {
        const uint32_t s_ = (sel >> (8 * i)) & 0xffu;
        const uint32_t byte = s_ < 4 ? (b >> (8 * s_)) & 0xffu : s_ < 8 ? (a >> (8 * (s_ - 4))) & 0xffu : 0u;
        r |= byte << (8 * i);
    }
    return r;
}
#else
#define MN_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
__device__ __forceinline__ uint32_t mn_alignbyte(uint32_t hi, uint32_t lo, int sh) { return __builtin_amdgcn_alignbyte(hi, lo, sh); }
__device__ __forceinline__ uint32_t mn_perm(uint32_t a, uint32_t b, uint32_t sel) { return __builtin_amdgcn_perm(a, b, sel); }
#endif
// ds_read_b64_tr_b16: the LDS transpose read of gfx950.  Every lane gives the LDS address of 4 consecutive bf16 (8-byte aligned); within a 16-lane group the
// 16 x 4 elements are a [4][16] block -- row e = lanes 4e .. 4e+3 of the group, four columns each -- and lane c receives COLUMN c: result element e = element
// (c & 3) of lane 4e + (c >> 2).  Feeds an MFMA operand whose K index is the row of a row-major LDS image (measured: scripts/probe/tr16_probe.hip).
typedef unsigned int mn_u32x2 __attribute__((vector_size(8)));
#ifdef MN_EMULATION
__device__ __forceinline__ mn_u32x2 mn_lds_tr16_b64(const unsigned char* p) { return emu_ds_read_tr16_b64(p); }
#else
__device__ __forceinline__ mn_u32x2 mn_lds_tr16_b64(const unsigned char* p) {
    typedef __bf16 mn_bf4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) mn_bf4 mn_lds_bf4;
    const mn_bf4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((mn_lds_bf4*)(__attribute__((address_space(3))) unsigned char*)p);
    return __builtin_bit_cast(mn_u32x2, v);
}
#endif
// a store with the non-temporal hint (a stream that is not read again by this kernel)
#ifdef MN_EMULATION
__device__ __forceinline__ void mn_store_nt(float* p, float v) { *p = v; }
#else
__device__ __forceinline__ void mn_store_nt(float* p, float v) { __builtin_nontemporal_store(v, p); }
#endif
// a value the optimiser must treat as unknown (stops hoisting / rematerialisation decisions that cost registers)
#ifdef MN_EMULATION
__device__ __forceinline__ uint32_t mn_opaque(uint32_t v) { return v; }
#else
__device__ __forceinline__ uint32_t mn_opaque(uint32_t v) { asm volatile("" : "+v"(v)); return v; }
#endif
// bf16 "head" of an fp32 (truncation): exact for integers |v| <= 256; v - head is exact in fp32, so
// v = t0 + t1 + t2 with t_i = head(remainder) reproduces all 24 significant bits (three-term split).
__device__ __forceinline__ float mn_bf16_head(float v) { return mn_u2f(mn_f2u(v) & 0xffff0000u); }
// pack the bf16 heads of two floats: low half = lo_elem, high half = hi_elem
__device__ __forceinline__ unsigned mn_pack_bf16x2(float lo_elem, float hi_elem) {
    return (mn_f2u(lo_elem) >> 16) | (mn_f2u(hi_elem) & 0xffff0000u);
}

// Two-term split (round 5): v ~ t0 + t1 with t0 = bf16 round-to-nearest-even of v and t1 = bf16 rne of the exact remainder v - t0: |v - t0 - t1| <= 2^-18 |v|
// (the three-term truncation split above is exact; see DESIGN "two bf16 terms").  mn_rne_bf16x2 = v_cvt_pk_bf16_f32: low half = lo_elem.
#ifdef MN_EMULATION
__device__ __forceinline__ unsigned mn_rne_bf16_bits(float v) { const unsigned u = mn_f2u(v); return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16; }        // (finite inputs)
__device__ __forceinline__ unsigned mn_rne_bf16x2(float lo_elem, float hi_elem) { return mn_rne_bf16_bits(lo_elem) | (mn_rne_bf16_bits(hi_elem) << 16); }
#else
__device__ __forceinline__ unsigned mn_rne_bf16x2(float lo_elem, float hi_elem) {
    typedef __bf16 mn_bf2 __attribute__((ext_vector_type(2)));
    typedef float mn_f2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector((mn_f2){lo_elem, hi_elem}, mn_bf2));
}
#endif
// How many bf16 terms carry an fp32 gradient operand through the matrix cores: 3 = exact (truncation split), 2 (default since round 5) = the split below.
// MN_GRAD_TERMS=3 (or the older MN_QD_TERMS=3) restores the exact split everywhere.
static inline int mn_grad_terms() { const char* e = MN_ENV("MN_GRAD_TERMS"); return (e && e[0] == '3') ? 3 : 2; }
// The few A/B knobs that stay (each is exercised by a test or by bench.py; the tuning knobs of rounds 1-5 are gone, their findings are in profiles/README.md):
//   MN_GRAD_TERMS=3    the exact three-term split in the dense backward kernels (tests/test_gpu_kernels.py, bench.py values_exact_terms)
//   MN_HSIGN_FOLD=0    the per-channel constants of the sign pass from a launch of their own (tests/test_kernels_emulated.py)
//   MN_QA_IEEE_DIV=1   the IEEE division in the DoReFa clip-STE instead of Markstein's correctly rounded quotient (bit-identical: tests/kernel_cases.py)
//   MN_QA_NO_INTERVAL=1  element-wise ReLU / clamp masks instead of the per-channel interval (bit-identical)
//   MN_NO_PACKED_PW=1  every conv call packs its own weight codes instead of reading the step's pre-packed image (bit-identical: check_qg_pack_multi)
static inline float mn_qa_inv(float s) { return (s > 0.f && !MN_ENV("MN_QA_IEEE_DIV")) ? 1.0f / s : 0.f; }          // RN(1 / s) of the division-free clip-STE; 0 selects the IEEE form
static inline int mn_qa_interval() { return MN_ENV("MN_QA_NO_INTERVAL") ? 0 : 1; }
static inline bool mn_use_packed() { return !MN_ENV("MN_NO_PACKED_PW"); }
// the two term words of a pair of floats: hi = rne pair, lo = rne pair of the remainders
__device__ __forceinline__ void mn_split2_bf16x2(float a, float b, unsigned& hi, unsigned& lo) {
    hi = mn_rne_bf16x2(a, b);
    lo = mn_rne_bf16x2(a - mn_u2f(hi << 16), b - mn_u2f(hi & 0xffff0000u));
}
// the same from the high halves as they are (no masking of the low halves needed): one v_perm
__device__ __forceinline__ unsigned mn_pack_hi16(float lo_elem, float hi_elem) { return mn_perm(mn_f2u(hi_elem), mn_f2u(lo_elem), 0x07060302u); }

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
// the DoReFa block BatchNorm + ReLU + next-layer quantizer (qact_kernels.hip): ReLU with ATen's NaN rule, and dz from the incoming gradient.
// quant 1: gq is the gradient w.r.t. the QUANTISED activation (the quantizer's clip-STE is applied here); 0: w.r.t. the activation itself
__device__ __forceinline__ float qa_relu(float z) { return (z > 0.f) ? z : ((z != z) ? z : 0.f); }
// j = rha(c / s), c = clamp(0.1 a, 0, 1) >= 0, i.e. floor(fl(c / s) + 0.5).  The IEEE division is ~10 instructions per element and made k_qa_fwd
// VALU-bound (2.9 TB/s of 3 B/elt).  q = c * n differs from fl(c / s) by a few ulp only (s = fl(1/n)), so floor(q + 0.5) is the same integer unless
// q + 0.5 lies within 2e-4 of one (q <= 255: 3 ulp < 5e-5); only then the division is evaluated -- bit-identical codes, the branch is rarely taken.
__device__ __forceinline__ uint32_t qa_code(float a, float s) {
    const float c = mn_clamp(a * 0.1f, 0.f, 1.f);
    const float nf = (float)(int)(1.0f / s + 0.5f);      // 2^bits - 1
    const float t = c * nf + 0.5f;
    float j = floorf(t);
    const float r = t - j;
    if (r < 2e-4f || r > 1.f - 2e-4f) j = floorf(c / s + 0.5f);
    return (j > 0.f) ? (uint32_t)j : 0u;                 // NaN -> 0 (a byte cannot hold it)
}
__device__ __forceinline__ float qa_dz(float gq, float a, float z, float s, int quant) {
    const float d = quant ? dorefa_act_grad(gq, a, s) : gq;
    return (z > 0.f) ? d : 0.f;
}
// The same with the IEEE division of (g s) / s replaced by Markstein's three-instruction correctly rounded quotient (inv = RN(1 / s): q0 = RN(d inv),
// r = d - s q0 exactly (one fma), RN(q0 + r inv)): bit-identical for every finite, normal d -- checked exhaustively-at-random on the host for the seven DoReFa
// scales 1 / (2^a - 1), a = 2 .. 8, 3e8 gradients each over 2^-60 .. 2^60 (0 mismatches) -- at a third of the instructions; the streaming backward passes of the
// k-bit blocks and the first-layer backward-weight that folds them are VALU-limited by that division.  inv == 0 selects the IEEE form (A/B knob MN_QA_IEEE_DIV).
__device__ __forceinline__ float dorefa_act_grad_m(float g, float x, float s, float inv) {
    const float t = x * 0.1f;
    const float d0 = g * s;
    float d;
    if (inv != 0.f) {
        const float q0 = d0 * inv;
        const float r = fmaf(-q0, s, d0);
        d = fmaf(r, inv, q0);
    } else {
        d = d0 / s;
    }
    d = (t >= 0.f && t <= 1.f) ? d : 0.f;
    return d * 0.1f;
}
// ---- the masks of that backward as an INTERVAL of the block's input value.  dz = STE(gq) * [z > 0] * [0.1 relu(z) <= 1] and z is a monotone function of the
// value v the pass streams (the fp32 conv output y, or the integer stash: every step of v -> y -> zhat -> z -> relu -> 0.1 a is monotone in fp32 as well), so the
// two conditions select ONE interval [lo, hi] of v per channel.  Its ends are found by bisection with the EXACT expressions (a lane per channel, ~60 evaluations
// once per block), after which an element costs two compares instead of the ~10 instructions of z, relu, the clamp test and their selects -- same decisions bit
// for bit.  Float domain: bisection over the ordered integer image of the floats.
__device__ __forceinline__ int32_t mn_fkey(float f) { const int32_t b = (int32_t)mn_f2u(f); return b >= 0 ? b : (int32_t)(0x80000000u - (uint32_t)b); }
__device__ __forceinline__ float mn_keyf(int32_t k) { return mn_u2f(k >= 0 ? (uint32_t)k : (0x80000000u - (uint32_t)k)); }
// s[k] += the rows i = 0 .. S-1 of a [S][NV] fp64 table, eight (then four) rows in flight, added in index order: the one-thread-per-channel "final" kernels
// issue one dependent row of loads at a time otherwise -- pure latency (same sums bit for bit: the order of the additions is the plain loop's)
template <int NV>
__device__ __forceinline__ void mn_row_sums(const double* __restrict__ src, int S, double (&s)[NV]) {
    int i = 0;
    for (; i + 8 <= S; i += 8) {
        double v[8][NV];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int k = 0; k < NV; ++k) v[u][k] = src[(i + u) * NV + k];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int k = 0; k < NV; ++k) s[k] += v[u][k];
    }
    for (; i + 4 <= S; i += 4) {
        double v[4][NV];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < NV; ++k) v[u][k] = src[(i + u) * NV + k];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < NV; ++k) s[k] += v[u][k];
    }
    for (; i < S; ++i)
#pragma unroll
        for (int k = 0; k < NV; ++k) s[k] += src[i * NV + k];
}
struct QaInterval { float lo, hi; };          // pass iff lo <= v && v <= hi (an empty set is lo = 1, hi = 0)
// zfun(v) -> z; the domain is the integers / float keys w in [-R, R], v = vof(w).  quant: the clamp condition applies.
// sgn: the BinaryActivation's clip-STE instead (wbwtab/quantize.py:26-36): pass iff -1 < z < 1.
template <class ZF, class VF>
__device__ __forceinline__ QaInterval qa_mask_interval(int32_t R, ZF zfun, VF vof, int quant, bool sgn = false) {
    const bool flip = zfun(vof(R)) < zfun(vof(-R));          // z decreases with v: search in w = -v
    auto zw = [&](int64_t w) { return zfun(vof((int32_t)(flip ? -w : w))); };
    auto first = [&](bool second) {          // smallest w in [-R, R] with the (monotone) predicate true, R + 1 if none
        int64_t lo = -(int64_t)R, hi = (int64_t)R + 1;
        while (lo < hi) {
            const int64_t mid = lo + ((hi - lo) >> 1);
            const float z = zw(mid);
            const bool pr = sgn ? (second ? !(z < 1.f) && z > -1.f : z > -1.f) : (second ? !((z > 0.f ? z : 0.f) * 0.1f <= 1.f) && z > 0.f : z > 0.f);
            if (pr) hi = mid; else lo = mid + 1;
        }
        return lo;
    };
    const int64_t w1 = first(false);
    const int64_t w2 = quant ? first(true) : (int64_t)R + 1;
    QaInterval r;
    if (w1 > w2 - 1) { r.lo = 1.f; r.hi = 0.f; return r; }
    const int64_t a = flip ? -(w2 - 1) : w1, b = flip ? -w1 : w2 - 1;
    r.lo = vof((int32_t)a); r.hi = vof((int32_t)b);
    return r;
}
__device__ __forceinline__ float dorefa_ste_core_m(float g, float s, float inv) {          // ((g s) / s) * 0.1: the clip-STE without its clamp test (the interval holds it)
    const float d0 = g * s;
    float d;
    if (inv != 0.f) {
        const float q0 = d0 * inv;
        const float r = fmaf(-q0, s, d0);
        d = fmaf(r, inv, q0);
    } else {
        d = d0 / s;
    }
    return d * 0.1f;
}
__device__ __forceinline__ float qa_dz_m(float gq, float a, float z, float s, float inv, int quant) {
    const float d = quant ? dorefa_act_grad_m(gq, a, s, inv) : gq;
    return (z > 0.f) ? d : 0.f;
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
// observer update (MinMax 62-74: running extremes; MovingAverage 101-113) and update_qparams -- shared by quant_kernels.hip and iao_bnfuse.hip
struct OpMaxF_ { __device__ __forceinline__ float operator()(float a, float b) const { return (a != a || b != b) ? (a != a ? a : b) : fmaxf(a, b); } };
struct OpMinF_ { __device__ __forceinline__ float operator()(float a, float b) const { return (a != a || b != b) ? (a != a ? a : b) : fminf(a, b); } };
__device__ __forceinline__ void observer_update(int obs_kind, int first, double momentum, float cmin, float cmax,
                                                float* min_val, float* max_val) {
    float lo, hi;
    if (first) { lo = cmin; hi = cmax; }
    else if (obs_kind == 0) { lo = OpMinF_()(cmin, *min_val); hi = OpMaxF_()(cmax, *max_val); }
    else {
        const float a = (float)(1.0 - momentum), b = (float)momentum;   // python doubles (1 - m), m become fp32 scalars
        lo = a * (*min_val) + b * cmin;
        hi = a * (*max_val) + b * cmax;
    }
    *min_val = lo;
    *max_val = hi;
}
// qparams (293-321) + clip-STE bounds (148-157)
__device__ __forceinline__ void iao_qparams_row(float mn, float mx, int q_type, float quant_range, int update, float* scale, float* zero_point, float* qp) {
    const float EPS = 1.1920928955078125e-07f;   // torch.finfo(float32).eps
    float sc, zp;
    if (update) {
        if (q_type == 0) {
            float fr = OpMaxF_()(fabsf(mn), fabsf(mx));
            sc = OpMaxF_()(fr / quant_range, EPS);
            zp = 0.f;
        } else {
            sc = OpMaxF_()((mx - mn) / quant_range, EPS);
            zp = mn_sign(mn) * floorf(fabsf(mn / sc) + 0.5f);
        }
        *scale = sc;
        *zero_point = zp;
    } else {
        sc = *scale;
        zp = *zero_point;
    }
    float lo = mn / sc - zp, hi = mx / sc - zp;
    if (q_type == 0) { hi = OpMaxF_()(fabsf(lo), fabsf(hi)); lo = -hi; }
    qp[0] = sc; qp[1] = zp; qp[2] = lo; qp[3] = hi;
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

// activation-quantizer descriptor (device side) shared by the conv kernels
struct Pro {           // prologue applied to loaded input elements
    int mode;          // MN_ACTQ_*
    float s;           // dorefa scale
    float qmin, qmax;  // iao
    const float* qp;   // iao {scale, zp, lo, hi}
};
__device__ __forceinline__ float pro_apply(const Pro& p, float v, float sc, float zp) {
    if (p.mode == MN_ACTQ_DOREFA) return dorefa_act_q(v, p.s);
    if (p.mode == MN_ACTQ_IAO) return iao_fq(v, sc, zp, p.qmin, p.qmax);
    return v;
}

static int make_pro(const mn_actq* aq, Pro* p, int need_bounds, const char* what) {
    p->mode = MN_ACTQ_NONE; p->s = 1.f; p->qmin = p->qmax = 0.f; p->qp = nullptr;
    if (!aq || aq->mode == MN_ACTQ_NONE) return MN_OK;
    if (aq->mode == MN_ACTQ_DOREFA) {
        if (aq->bits < 2 || aq->bits > 31) MN_FAIL(MN_EINVAL, "%s: dorefa bits=%d", what, aq->bits);
        p->mode = MN_ACTQ_DOREFA; p->s = dorefa_scale(aq->bits);
        return MN_OK;
    }
    if (aq->mode == MN_ACTQ_IAO) {
        if (aq->bits < 2 || aq->bits > 24 || !aq->qp) MN_FAIL(MN_EINVAL, "%s: iao bits=%d / null qp", what, aq->bits);
        IaoRange r = iao_range(aq->bits, aq->q_type, 1);
        p->mode = MN_ACTQ_IAO; p->qmin = r.qmin; p->qmax = r.qmax; p->qp = aq->qp;
        return MN_OK;
    }
    if (aq->mode == MN_ACTQ_SIGN8) { p->mode = MN_ACTQ_SIGN8; return MN_OK; }
    if (aq->mode == MN_ACTQ_CODE8) {
        if (aq->bits < 2 || aq->bits > 8) MN_FAIL(MN_EINVAL, "%s: code8 bits=%d", what, aq->bits);
        p->mode = MN_ACTQ_CODE8; p->s = dorefa_scale(aq->bits);
        return MN_OK;
    }
    (void)need_bounds;
    MN_FAIL(MN_EINVAL, "%s: unknown activation quantizer mode %d", what, aq->mode);
}


// four consecutive elements of the streamed conv input at element offset `off` (a multiple of 4): fp32, or -- MN_ACTQ_SIGN8 --
// int8 sign codes widened to +-1.0f (any negative byte is -1, anything else +1)
template <int XMODE>
__device__ __forceinline__ float4 mn_ld_x4(const float* base, int64_t off) {
    if (XMODE == MN_ACTQ_SIGN8) {
        const uint32_t u = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(base) + off);
        return make_float4((u & 0x80u) ? -1.f : 1.f, (u & 0x8000u) ? -1.f : 1.f, (u & 0x800000u) ? -1.f : 1.f, (u & 0x80000000u) ? -1.f : 1.f);
    }
    return *reinterpret_cast<const float4*>(base + off);
}
// bf16 pair {sign of byte q of u_lo, sign of byte q of u_hi} as +-1 (0x3F80 / 0xBF80): low half = u_lo's element
__device__ __forceinline__ unsigned mn_sign8_pair(unsigned u_lo, unsigned u_hi, int q) {
    return 0x3F803F80u | (((u_lo >> (8 * q)) & 0x80u) << 8) | (((u_hi >> (8 * q)) & 0x80u) << 24);
}

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
__device__ __forceinline__ uint32_t fd_div(uint32_t n, FastDiv f) { const uint32_t q = __umulhi(n, f.m); return f.d <= 1 ? n : q; }     // select, not a branch

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

// ---- tails shared by the stand-alone observer kernels (quant_kernels.hip) and by the last block of k_bn_acc_prep (norm_kernels.hip); `sc`: 16 floats of shared memory,
// called by ALL 256 threads of one block.
// the observer update from per-block (min, max) partials (mm[0 .. count) minima, mm[count .. 2 count) maxima) [+ the per-tensor quantizer's update_qparams, qp != null]
__device__ __forceinline__ void mn_obs_partials_tail(const float* __restrict__ mm, int count, int obs_kind, int first, double momentum, float* __restrict__ min_val,
                                                     float* __restrict__ max_val, int q_type, float quant_range, float* __restrict__ scale, float* __restrict__ zero_point,
                                                     float* __restrict__ qp, float* sc) {
    float lo = INFINITY, hi = -INFINITY;
    {   // eight loads in flight per thread and pass (a plain loop waited one L2 round trip per 256 partials: 8-10 us for the 2-4 k partials of a ResNet layer);
        // plain min / max ignore a NaN: it is flagged and propagated below, as torch.min / max would
        int nan = 0, i = threadIdx.x;
        for (; i + 3 * 256 < count; i += 4 * 256) {
            const float a0 = mm[i], a1 = mm[i + 256], a2 = mm[i + 512], a3 = mm[i + 768];
            const float b0 = mm[count + i], b1 = mm[count + i + 256], b2 = mm[count + i + 512], b3 = mm[count + i + 768];
            lo = fminf(lo, fminf(fminf(a0, a1), fminf(a2, a3)));
            hi = fmaxf(hi, fmaxf(fmaxf(b0, b1), fmaxf(b2, b3)));
            nan |= (int)((a0 != a0) | (a1 != a1) | (a2 != a2) | (a3 != a3) | (b0 != b0) | (b1 != b1) | (b2 != b2) | (b3 != b3));
        }
        for (; i < count; i += 256) {
            const float a = mm[i], b = mm[count + i];
            lo = fminf(lo, a); hi = fmaxf(hi, b);
            nan |= (int)((a != a) | (b != b));
        }
        if (nan) lo = hi = NAN;
    }
    lo = block_reduce(lo, OpMinF(), INFINITY, sc);
    hi = block_reduce(hi, OpMaxF(), -INFINITY, sc);
    if (threadIdx.x == 0) {
        observer_update(obs_kind, first, momentum, lo, hi, min_val, max_val);
        if (qp) iao_qparams_row(*min_val, *max_val, q_type, quant_range, 1, scale, zero_point, qp);          // + the quantizer's update_qparams in the same launch
    }
}
// QuantAdd's bookkeeping (wqaq/iao/quantize.py:1484-1498) from the two producers' partials: both input observers, their union range, the shared quantizer's qparams
struct QaddFinal {
    int nb, obs_kind, first_a, first_b, q_type, update; double momentum; float quant_range;
    float *min_a, *max_a, *min_b, *max_b, *min_o, *max_o, *scale, *zero_point, *qp;
};
__device__ __forceinline__ void mn_qadd_final_tail(const float* __restrict__ ma, int ca, const float* __restrict__ mb, int cb, const QaddFinal& f, float* sc) {
    float la = INFINITY, ha = -INFINITY, lb = INFINITY, hb = -INFINITY;
    // eight loads in flight per thread and pass, like mn_obs_partials_tail (the plain loops waited one L2 round trip per 256 partials: 18 us per residual add of
    // a ResNet step, 2-4 k partials per side); min / max are order-free, so the result is the same
    auto side = [&](const float* __restrict__ m, int cnt, float& lo, float& hi) {
        int i = threadIdx.x;
        for (; i + 3 * 256 < cnt; i += 4 * 256) {
            const float a0 = m[i], a1 = m[i + 256], a2 = m[i + 512], a3 = m[i + 768];
            const float b0 = m[cnt + i], b1 = m[cnt + i + 256], b2 = m[cnt + i + 512], b3 = m[cnt + i + 768];
            lo = OpMinF()(OpMinF()(lo, a0), OpMinF()(OpMinF()(a1, a2), a3));
            hi = OpMaxF()(OpMaxF()(hi, b0), OpMaxF()(OpMaxF()(b1, b2), b3));
        }
        for (; i < cnt; i += 256) { lo = OpMinF()(lo, m[i]); hi = OpMaxF()(hi, m[cnt + i]); }
    };
    side(ma, ca, la, ha);
    side(mb, cb, lb, hb);
    la = block_reduce(la, OpMinF(), INFINITY, sc); ha = block_reduce(ha, OpMaxF(), -INFINITY, sc);
    lb = block_reduce(lb, OpMinF(), INFINITY, sc); hb = block_reduce(hb, OpMaxF(), -INFINITY, sc);
    if (threadIdx.x == 0) {
        observer_update(f.obs_kind, f.first_a, f.momentum, la, ha, f.min_a, f.max_a);
        observer_update(f.obs_kind, f.first_b, f.momentum, lb, hb, f.min_b, f.max_b);
        const float mn = OpMinF()(*f.min_a, *f.min_b), mx = OpMaxF()(*f.max_a, *f.max_b);
        *f.min_o = mn; *f.max_o = mx;
        iao_qparams_row(mn, mx, f.q_type, f.quant_range, f.update, f.scale, f.zero_point, f.qp);
    }
}

// Byte stash h = (acc + nnz) / 2 of a ternary-weight convolution on +-1 codes: acc has the parity of the number of non-zero weight codes
// that meet a non-zero input.  For a pointwise block that is nnz[o] everywhere (chan row 7).  For a 3x3 / padding 1 block the taps
// outside the image meet zeros, so the count depends on the pixel CLASS (top / middle / bottom row x left / middle / right column):
// chan row 7 is then -1 and rows 8..16 hold nnz[class = 3 rc + cc][o].
struct StashNnz { float v0, v1, v2, v3, v4, v5, v6, v7, v8; };
__device__ __forceinline__ StashNnz stash_nnz_load(const float* __restrict__ chan, int C, int c) {
    StashNnz z;
    const float n7 = chan[7 * C + c];
    const bool t = n7 < 0.f;
    z.v0 = t ? chan[8 * C + c] : n7;  z.v1 = t ? chan[9 * C + c] : n7;  z.v2 = t ? chan[10 * C + c] : n7;
    z.v3 = t ? chan[11 * C + c] : n7; z.v4 = t ? chan[12 * C + c] : n7; z.v5 = t ? chan[13 * C + c] : n7;
    z.v6 = t ? chan[14 * C + c] : n7; z.v7 = t ? chan[15 * C + c] : n7; z.v8 = t ? chan[16 * C + c] : n7;
    return z;
}
// nnz of the four pixels of quad col4 in row `row` of an H x (4 W4) plane.  Blended arithmetically (the values are small integers: exact)
// -- a chain of selects between the struct's fields is turned into an indexed load from a scratch copy of the struct by the compiler.
__device__ __forceinline__ void stash_nnz_quad(const StashNnz& z, int row, int col4, int H, int W4, float (&nz)[4]) {
    const float top = row == 0 ? 1.f : 0.f, bot = row == H - 1 ? 1.f : 0.f;
    const float m0 = z.v3 + top * (z.v0 - z.v3) + bot * (z.v6 - z.v3);
    const float m1 = z.v4 + top * (z.v1 - z.v4) + bot * (z.v7 - z.v4);
    const float m2 = z.v5 + top * (z.v2 - z.v5) + bot * (z.v8 - z.v5);
    const float lf = col4 == 0 ? 1.f : 0.f, rt = col4 == W4 - 1 ? 1.f : 0.f;
    nz[0] = m1 + lf * (m0 - m1); nz[1] = m1; nz[2] = m1; nz[3] = m1 + rt * (m2 - m1);
}

// nnz of one pixel (row, col) of an H x W plane
__device__ __forceinline__ float stash_nnz_px(const StashNnz& z, int row, int col, int H, int W) {
    const float top = row == 0 ? 1.f : 0.f, bot = row == H - 1 ? 1.f : 0.f;
    const float m0 = z.v3 + top * (z.v0 - z.v3) + bot * (z.v6 - z.v3);
    const float m1 = z.v4 + top * (z.v1 - z.v4) + bot * (z.v7 - z.v4);
    const float m2 = z.v5 + top * (z.v2 - z.v5) + bot * (z.v8 - z.v5);
    const float lf = col == 0 ? 1.f : 0.f, rt = col == W - 1 ? 1.f : 0.f;
    return m1 + lf * (m0 - m1) + rt * (m2 - m1);
}

// logical -> physical channel through a channel shuffle with `sg` groups over C channels (identity when sg <= 1)
struct ChanMap { int sg, cps; FastDiv fd_sg; };
static inline ChanMap make_chanmap(int sg, int C) {
    ChanMap m;
    m.sg = sg > 1 ? sg : 1; m.cps = sg > 1 ? C / sg : 0; m.fd_sg = make_fastdiv((uint32_t)m.sg);
    return m;
}
__device__ __forceinline__ int chan_phys(const ChanMap& m, int cl) {
    if (m.sg <= 1) return cl;
    const uint32_t jq = fd_div((uint32_t)cl, m.fd_sg);
    return (cl - (int)jq * m.sg) * m.cps + (int)jq;
}

static inline int mn_grid_for(int64_t n_items, int per_block, int cap) {
    int64_t b = (n_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}
