// Quantizer kernels of the three micronet schemes (DoReFa, WbWtAb, IAO) for gfx950.
//
// Everything here is HBM-bound streaming work (activations) or tiny per-channel work (weights):
//   * activation-sized tensors: one fused pass, 16 B per lane coalesced (float4), grid capped at
//     256 CUs x 8 blocks and grid-strided -- replaces the 5..15 separate ATen pointwise kernels the
//     reference launches per quantizer;
//   * weight tensors: one workgroup per output channel, statistics accumulated in fp64 so the
//     result does not depend on the reduction order (deterministic, within 1 ulp of any fp32 order);
//   * the integer step (divide by scale, round half away, clamp) uses IEEE fp32 division and the
//     exact expression order of the reference, so codes are bit-identical to the CPU path.
// Compile with -ffp-contract=off: no fma contraction may change a rounding.
#include "common.h"

#include <math.h>
#include <stdarg.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

// ------------------------------------------------------------------------------------------------
// error plumbing
static thread_local char g_err[512] = "";
void mn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* mn_last_error(void) { return g_err; }
static thread_local char g_kernel[96] = "";
void mn_set_last_kernel(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
    va_end(ap);
}
extern "C" const char* mn_last_kernel(void) { return g_kernel; }
// ---- measurement: HIP events around the main kernel of each entry point, on the stream it is launched on.
// (a) mn_profile_next: caller-owned events for the next conv call;  (b) mn_profile_enable / mn_profile_collect: the library
// keeps its own spans (name, designed bytes, two pooled events per launch) for EVERY main kernel and aggregates them by name.
static thread_local void* g_prof_ev[2] = {nullptr, nullptr};
static thread_local double g_prof_bytes = 0.0;
static thread_local double g_prof_flops = 0.0;
extern "C" void mn_profile_next(void* start_event, void* stop_event) { g_prof_ev[0] = start_event; g_prof_ev[1] = stop_event; }
void mn_prof_bytes(double nbytes) { g_prof_bytes = nbytes; g_prof_flops = 0.0; }
void mn_prof_flops(double nflops) { g_prof_flops = nflops; }
#ifndef MN_EMULATION
struct ProfSpan { std::string name; double bytes, flops; hipEvent_t a, b; };
// process-wide (autograd runs the backward on its own thread); guarded by g_prof_mu
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfSpan>* g_spans = nullptr;
static std::vector<hipEvent_t>* g_evpool = nullptr;
static thread_local hipEvent_t g_open_stop = nullptr;     // stop event of the span this thread has open
static hipEvent_t prof_event() {
    if (g_evpool && !g_evpool->empty()) { hipEvent_t e = g_evpool->back(); g_evpool->pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
#endif
extern "C" void mn_profile_enable(int on) {
#ifndef MN_EMULATION
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_spans) { g_spans = new std::vector<ProfSpan>(); g_evpool = new std::vector<hipEvent_t>(); }
    for (auto& sp : *g_spans) { g_evpool->push_back(sp.a); g_evpool->push_back(sp.b); }
    g_spans->clear();
    g_prof_on = on != 0;
#else
    (void)on;
#endif
}
extern "C" int mn_profile_collect(mn_prof_entry* out, int cap) {
#ifndef MN_EMULATION
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_spans) return 0;
    std::map<std::string, mn_prof_entry> agg;
    for (auto& sp : *g_spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
            mn_prof_entry& e = agg[sp.name];
            if (e.launches == 0) { memset(e.name, 0, sizeof(e.name)); strncpy(e.name, sp.name.c_str(), sizeof(e.name) - 1); }
            e.launches += 1; e.total_ms += ms; e.bytes += sp.bytes; e.flops += sp.flops;
        }
        g_evpool->push_back(sp.a); g_evpool->push_back(sp.b);
    }
    g_spans->clear();
    int n = 0;
    for (auto& kv : agg) { if (n < cap && out) out[n] = kv.second; ++n; }
    return n < cap ? n : cap;
#else
    (void)out; (void)cap;
    return 0;
#endif
}
void mn_prof_begin(hipStream_t s) {
#ifndef MN_EMULATION
    if (g_prof_ev[0]) (void)hipEventRecord((hipEvent_t)g_prof_ev[0], s);
    if (g_prof_on) {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (g_prof_on && g_spans) {
            ProfSpan sp; sp.name = g_kernel; sp.bytes = g_prof_bytes; sp.flops = g_prof_flops; sp.a = prof_event(); sp.b = prof_event();
            (void)hipEventRecord(sp.a, s);
            g_open_stop = sp.b;
            g_spans->push_back(sp);
        }
    }
#else
    (void)s;
#endif
}
void mn_prof_end(hipStream_t s) {
#ifndef MN_EMULATION
    if (g_prof_ev[1]) (void)hipEventRecord((hipEvent_t)g_prof_ev[1], s);
    if (g_open_stop) { (void)hipEventRecord(g_open_stop, s); g_open_stop = nullptr; }
#else
    (void)s;
#endif
    g_prof_ev[0] = g_prof_ev[1] = nullptr;
}
extern "C" int mn_version(void) { return 100; }
extern "C" int mn_dense_grad_terms(void) { return mn_grad_terms(); }
extern "C" int mn_is_emulation(void) {
#ifdef MN_EMULATION
    return 1;
#else
    return 0;
#endif
}

// ------------------------------------------------------------------------------------------------
// generic streaming maps: out = f(in0[, in1]) ; float4 body + scalar tail
static const int EW_BLOCK = 256;
static const int EW_GRID_CAP = 256 * 8;

template <typename F>
__global__ __launch_bounds__(256) void k_map1(const float* __restrict__ a, float* __restrict__ y, int64_t n, int vec, F f) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        const int64_t n4 = n >> 2;
        const float4* a4 = reinterpret_cast<const float4*>(a);
        float4* y4 = reinterpret_cast<float4*>(y);
        for (int64_t j = i; j < n4; j += stride) {
            float4 v = a4[j], r;
            r.x = f(v.x); r.y = f(v.y); r.z = f(v.z); r.w = f(v.w);
            y4[j] = r;
        }
        for (int64_t j = (n4 << 2) + i; j < n; j += stride) y[j] = f(a[j]);
    } else {
        for (int64_t j = i; j < n; j += stride) y[j] = f(a[j]);
    }
}
template <typename F>
__global__ __launch_bounds__(256) void k_map2(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y,
                                              int64_t n, int vec, F f) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        const int64_t n4 = n >> 2;
        const float4* a4 = reinterpret_cast<const float4*>(a);
        const float4* b4 = reinterpret_cast<const float4*>(b);
        float4* y4 = reinterpret_cast<float4*>(y);
        for (int64_t j = i; j < n4; j += stride) {
            float4 v = a4[j], w = b4[j], r;
            r.x = f(v.x, w.x); r.y = f(v.y, w.y); r.z = f(v.z, w.z); r.w = f(v.w, w.w);
            y4[j] = r;
        }
        for (int64_t j = (n4 << 2) + i; j < n; j += stride) y[j] = f(a[j], b[j]);
    } else {
        for (int64_t j = i; j < n; j += stride) y[j] = f(a[j], b[j]);
    }
}
template <typename F>
static int launch_map1(const float* a, float* y, int64_t n, F f, hipStream_t s, const char* what) {
    if (n < 0 || (n > 0 && (!a || !y))) MN_FAIL(MN_EINVAL, "%s: bad arguments", what);
    if (n == 0) return MN_OK;
    int vec = aligned16(a) && aligned16(y);
    int grid = mn_grid_for(vec ? (n + 3) / 4 : n, EW_BLOCK, EW_GRID_CAP);
    hipLaunchKernelGGL(k_map1<F>, dim3(grid), dim3(EW_BLOCK), 0, s, a, y, n, vec, f);
    MN_CHECK_LAUNCH(what);
    return MN_OK;
}
template <typename F>
static int launch_map2(const float* a, const float* b, float* y, int64_t n, F f, hipStream_t s, const char* what) {
    if (n < 0 || (n > 0 && (!a || !b || !y))) MN_FAIL(MN_EINVAL, "%s: bad arguments", what);
    if (n == 0) return MN_OK;
    int vec = aligned16(a) && aligned16(b) && aligned16(y);
    int grid = mn_grid_for(vec ? (n + 3) / 4 : n, EW_BLOCK, EW_GRID_CAP);
    hipLaunchKernelGGL(k_map2<F>, dim3(grid), dim3(EW_BLOCK), 0, s, a, b, y, n, vec, f);
    MN_CHECK_LAUNCH(what);
    return MN_OK;
}

// ------------------------------------------------------------------------------------------------
// DoReFa activation (wqaq/dorefa/quantize.py:36-46) and Round (11-21)
struct FRha { __device__ float operator()(float v) const { return mn_rha(v); } };
struct FDorefaActFwd {
    float s;
    __device__ float operator()(float x) const { return dorefa_act_q(x, s); }
};
struct FDorefaActBwd {
    float s;
    __device__ float operator()(float g, float x) const { return dorefa_act_grad(g, x, s); }
};

extern "C" int mn_round_half_away(const float* v, float* out, int64_t n, mn_stream_t stream) {
    return launch_map1(v, out, n, FRha(), (hipStream_t)stream, "mn_round_half_away");
}
extern "C" int mn_dorefa_act_fwd(const float* x, float* y, int64_t n, int a_bits, mn_stream_t stream) {
    if (a_bits < 2 || a_bits > 31) MN_FAIL(MN_EINVAL, "mn_dorefa_act_fwd: a_bits=%d (1 unsupported, 32 is a pass-through handled by the caller)", a_bits);
    FDorefaActFwd f{dorefa_scale(a_bits)};
    return launch_map1(x, y, n, f, (hipStream_t)stream, "mn_dorefa_act_fwd");
}
extern "C" int mn_dorefa_act_bwd(const float* g, const float* x, float* dx, int64_t n, int a_bits, mn_stream_t stream) {
    if (a_bits < 2 || a_bits > 31) MN_FAIL(MN_EINVAL, "mn_dorefa_act_bwd: a_bits=%d", a_bits);
    FDorefaActBwd f{dorefa_scale(a_bits)};
    return launch_map2(g, x, dx, n, f, (hipStream_t)stream, "mn_dorefa_act_bwd");
}

// ------------------------------------------------------------------------------------------------
// tanh as the DoReFa weight quantizer needs it.  The reference evaluates torch.tanh on the CPU, which in the reference's environment (torch 2.10, MKL build) is
// Intel MKL VML vsTanh in HA mode -- verified bit for bit; it is NOT Sleef's tanhf_u10 (1.5 % of inputs differ) nor libm -- a closed-source kernel with no published
// algorithm, so it cannot be restated.  What can be done is to take the rounding error out of OUR side: the value is evaluated in fp64 and rounded once, i.e. the
// correctly rounded fp32 tanh (MKL HA differs from it in the last ulp for 1.5 % of inputs; ocml's tanhf, used until round 2, for 5.4 %).  The tensors are small
// (<= 11 M weights per net): the fp64 evaluation costs microseconds.  Pinned by tests/golden/tanh_device_vs_cpu.json.
// Round 6: |x| <= 0.25 (every weight of a net in training, in practice) takes the Maclaurin series in fp64 -- x + x z (c3 + z (c5 + ...)), z = x^2, 13 terms: truncation
// below 2^-66 relative, evaluation error below one fp64 ulp (measured 0.99 ulp against a 60-digit reference), i.e. the same "fp64 value rounded once" at a tenth of the
// instructions of the library tanh (exp + division), which made the absmax pass ALU-bound (66 us for resnet18's 11 M weights); larger arguments take the library call.
__device__ __forceinline__ float mn_tanh_cr(float x) {
    const double xd = (double)x;
    if (fabsf(x) > 0.25f) return (float)tanh(xd);
    const double z = xd * xd;
    double q = 0x1.0b132d39a6050p-16;
    q = fma(q, z, -0x1.497d8eea25259p-15);
    q = fma(q, z, 0x1.967e18afcafadp-14);
    q = fma(q, z, -0x1.f57d7734d1664p-13);
    q = fma(q, z, 0x1.3558248036744p-11);
    q = fma(q, z, -0x1.7da36452b75e3p-10);
    q = fma(q, z, 0x1.d6d3d0e157de0p-9);
    q = fma(q, z, -0x1.226e355e6c23dp-7);
    q = fma(q, z, 0x1.664f4882c10fap-6);
    q = fma(q, z, -0x1.ba1ba1ba1ba1cp-5);
    q = fma(q, z, 0x1.1111111111111p-3);
    q = fma(q, z, -0x1.5555555555555p-2);
    return (float)fma(xd, z * q, xd);
}
// DoReFa weight (61-73): global max of |tanh w| -> normalise -> round -> 2q-1.
// ws layout (floats): [0] M, [1] dM (bwd), [2] tie count (bwd), [16 .. 16+3*NB) per-block partials.
static const int DW_NB = 1024;  // partial blocks per tensor (a 2.4 M-element resnet layer: 9 elements per thread; 128 blocks left the absmax pass latency-bound at 97 us)
extern "C" int64_t mn_dorefa_w_ws_floats(int64_t) { return 16 + 3 * DW_NB; }

__global__ __launch_bounds__(256) void k_dorefa_w_absmax(const float* __restrict__ w, int64_t n, float* __restrict__ ws) {
    __shared__ float sc[16];
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        m = OpMaxF()(m, fabsf(mn_tanh_cr(w[i])));
    m = block_reduce(m, OpMaxF(), 0.f, sc);
    if (threadIdx.x == 0) ws[16 + blockIdx.x] = m;
}
__device__ __forceinline__ float dorefa_w_global_max(const float* ws, int nb, float* sc) {
    float m = 0.f;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) m = OpMaxF()(m, ws[16 + i]);
    return block_reduce(m, OpMaxF(), 0.f, sc);
}
__global__ __launch_bounds__(256) void k_dorefa_w_fwd(const float* __restrict__ w, float* __restrict__ qw, int64_t n, float s,
                                                      float* __restrict__ ws, int nb) {
    __shared__ float sc[16];
    const float M = dorefa_w_global_max(ws, nb, sc);
    if (blockIdx.x == 0 && threadIdx.x == 0) ws[0] = M;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float t = mn_tanh_cr(w[i]);
        float u = (t / 2.f) / M + 0.5f;
        float q = mn_rha(u / s) * s;
        qw[i] = 2.f * q - 1.f;
    }
}
// backward pass 1: per-block partial of dM = -sum(du * (t/2) / M^2) (fp64) and of the tie count
__global__ __launch_bounds__(256) void k_dorefa_w_bwd_partial(const float* __restrict__ g, const float* __restrict__ w, int64_t n,
                                                              float s, float* __restrict__ ws, int nb) {
    __shared__ float sc[16];
    __shared__ double scd[16];
    const float M = dorefa_w_global_max(ws, nb, sc);
    double acc = 0.0;
    float ties = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float t = mn_tanh_cr(w[i]);
        float du = ((g[i] * 2.f) * s) / s;
        float v = t / 2.f;
        acc += (double)(-du * v / (M * M));
        ties += (fabsf(t) == M) ? 1.f : 0.f;
    }
    acc = block_reduce(acc, OpAddD(), 0.0, scd);
    ties = block_reduce(ties, OpAddF(), 0.f, sc);
    if (threadIdx.x == 0) {
        ws[16 + nb + blockIdx.x] = (float)acc;       // partial sums are small in count; final sum again in fp64
        ws[16 + 2 * nb + blockIdx.x] = ties;
    }
}
__global__ __launch_bounds__(256) void k_dorefa_w_bwd(const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ dw,
                                                      int64_t n, float s, float* __restrict__ ws, int nb) {
    __shared__ float sc[16];
    __shared__ double scd[16];
    const float M = dorefa_w_global_max(ws, nb, sc);
    double a = 0.0;
    float c = 0.f;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) {
        a += (double)ws[16 + nb + i];
        c += ws[16 + 2 * nb + i];
    }
    const float dM = (float)block_reduce(a, OpAddD(), 0.0, scd);
    const float cnt = block_reduce(c, OpAddF(), 0.f, sc);
    if (blockIdx.x == 0 && threadIdx.x == 0) { ws[1] = dM; ws[2] = cnt; }
    const float share = dM / cnt;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float t = mn_tanh_cr(w[i]);
        float du = ((g[i] * 2.f) * s) / s;
        float dt = (du / M) / 2.f;
        if (fabsf(t) == M) dt += share * mn_sign(t);
        dw[i] = dt * (1.f - t * t);
    }
}
extern "C" int mn_dorefa_w_fwd(const float* w, float* qw, int64_t n, int w_bits, float* ws, mn_stream_t stream) {
    if (w_bits < 2 || w_bits > 31 || n <= 0 || !w || !qw || !ws) MN_FAIL(MN_EINVAL, "mn_dorefa_w_fwd: bad arguments (w_bits=%d n=%lld)", w_bits, (long long)n);
    hipStream_t s = (hipStream_t)stream;
    int nb = mn_grid_for(n, 256, DW_NB);
    hipLaunchKernelGGL(k_dorefa_w_absmax, dim3(nb), dim3(256), 0, s, w, n, ws);
    hipLaunchKernelGGL(k_dorefa_w_fwd, dim3(mn_grid_for(n, 256, 1024)), dim3(256), 0, s, w, qw, n, dorefa_scale(w_bits), ws, nb);
    MN_CHECK_LAUNCH("mn_dorefa_w_fwd");
    return MN_OK;
}
extern "C" int mn_dorefa_w_bwd(const float* g, const float* w, float* dw, int64_t n, int w_bits, float* ws, mn_stream_t stream) {
    if (w_bits < 2 || w_bits > 31 || n <= 0 || !g || !w || !dw || !ws) MN_FAIL(MN_EINVAL, "mn_dorefa_w_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    int nb = mn_grid_for(n, 256, DW_NB);
    float sc = dorefa_scale(w_bits);
    hipLaunchKernelGGL(k_dorefa_w_absmax, dim3(nb), dim3(256), 0, s, w, n, ws);
    hipLaunchKernelGGL(k_dorefa_w_bwd_partial, dim3(nb), dim3(256), 0, s, g, w, n, sc, ws, nb);
    hipLaunchKernelGGL(k_dorefa_w_bwd, dim3(mn_grid_for(n, 256, 1024)), dim3(256), 0, s, g, w, dw, n, sc, ws, nb);
    MN_CHECK_LAUNCH("mn_dorefa_w_bwd");
    return MN_OK;
}

// the tanh the kernels above evaluate, element-wise: lets the tests pin it against torch-CPU's (tests/golden/tanh_device_vs_cpu.json)
__global__ void k_tanh_f32(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) y[i] = mn_tanh_cr(x[i]);
}
extern "C" int mn_tanh_f32(const float* x, float* y, int64_t n, mn_stream_t stream) {
    if (!x || !y || n <= 0) MN_FAIL(MN_EINVAL, "mn_tanh_f32: bad arguments");
    hipLaunchKernelGGL(k_tanh_f32, dim3(mn_grid_for(n, 256, 4096)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    MN_CHECK_LAUNCH("mn_tanh_f32");
    return MN_OK;
}

// The same quantizer over up to MN_DW_MAX_TENSORS weight tensors in ONE launch per phase (a training step quantizes the weights of every conv: as separate
// calls that is 2 launches forward and 3 backward per layer, each ~5 us): a by-value table maps a block to (tensor, block-in-tensor).  Same
// arithmetic per tensor (same partial-block count, same reduction order) as the single-tensor entry points: bit-identical results.
#define MN_DW_MAX_TENSORS 32
struct DwTable {
    const float* w[MN_DW_MAX_TENSORS];
    const float* g[MN_DW_MAX_TENSORS];
    float* out[MN_DW_MAX_TENSORS];       // qw (forward) / dw (backward)
    float* ws[MN_DW_MAX_TENSORS];
    float* th[MN_DW_MAX_TENSORS];        // optional cache of tanh(w) (n floats per tensor): written by the absmax pass, read by every later pass of the step
    long long n[MN_DW_MAX_TENSORS];
    int b0[MN_DW_MAX_TENSORS + 1];       // first block of each tensor
    int nb[MN_DW_MAX_TENSORS];           // partial blocks of each tensor (<= DW_NB)
    int count;
    float s;
};
__device__ __forceinline__ int dw_find(const DwTable& t, int b) {
    int ti = 0;
    while (ti + 1 < t.count && t.b0[ti + 1] <= b) ++ti;
    return ti;
}
__device__ __forceinline__ bool dw_vec_ok(const void* a, const void* b, const void* c, const void* d, long long n) {
    return (n & 3) == 0 && ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)d)) & 15) == 0;
}
__global__ __launch_bounds__(256) void k_dorefa_w_absmax_multi(const DwTable t) {
    __shared__ float sc[16];
    const int ti = dw_find(t, blockIdx.x), lb = blockIdx.x - t.b0[ti], nb = t.nb[ti];
    const float* __restrict__ w = t.w[ti];
    const long long n = t.n[ti];
    float m = 0.f;
    float* __restrict__ th = t.th[ti];
    if (dw_vec_ok(w, th, nullptr, nullptr, n)) {          // (max is order-independent and the cache element-wise: any assignment of elements to blocks gives the same bits)
        const long long n4 = n >> 2;
        for (long long i = (long long)lb * 256 + threadIdx.x; i < n4; i += (long long)nb * 256) {
            const float4 v = *reinterpret_cast<const float4*>(w + 4 * i);
            const float4 tt = make_float4(mn_tanh_cr(v.x), mn_tanh_cr(v.y), mn_tanh_cr(v.z), mn_tanh_cr(v.w));
            if (th) *reinterpret_cast<float4*>(th + 4 * i) = tt;
            m = OpMaxF()(OpMaxF()(OpMaxF()(OpMaxF()(m, fabsf(tt.x)), fabsf(tt.y)), fabsf(tt.z)), fabsf(tt.w));
        }
    } else {
        for (long long i = (long long)lb * 256 + threadIdx.x; i < n; i += (long long)nb * 256) {
            const float tt = mn_tanh_cr(w[i]);
            if (th) th[i] = tt;
            m = OpMaxF()(m, fabsf(tt));
        }
    }
    m = block_reduce(m, OpMaxF(), 0.f, sc);
    if (threadIdx.x == 0) t.ws[ti][16 + lb] = m;
}
// One block per tensor finishes what the partial passes left (round 6): which 0: M = max of the absmax partials -> ws[0]; which 1: dM (fp64 sum of the partials, the
// single-tensor kernel's order) and the tie count -> ws[1], ws[2].  Until round 6 EVERY block of the element-wise phases redid these reductions over up to 1024
// partials as its prologue (two to four block-wide reductions in front of ~2 float4 of work per thread: 44 us per phase for resnet18's 11 M weights, 2 TB/s).
__global__ __launch_bounds__(256) void k_dorefa_w_final_multi(const DwTable t, int which) {
    __shared__ float sc[16];
    __shared__ double scd[16];
    const int ti = blockIdx.x, nb = t.nb[ti];
    float* __restrict__ ws = t.ws[ti];
    if (which == 0) {
        const float M = dorefa_w_global_max(ws, nb, sc);
        if (threadIdx.x == 0) ws[0] = M;
    } else {
        double a = 0.0;
        float c = 0.f;
        for (int i = threadIdx.x; i < nb; i += blockDim.x) { a += (double)ws[16 + nb + i]; c += ws[16 + 2 * nb + i]; }
        const float dM = (float)block_reduce(a, OpAddD(), 0.0, scd);
        const float cnt = block_reduce(c, OpAddF(), 0.f, sc);
        if (threadIdx.x == 0) { ws[1] = dM; ws[2] = cnt; }
    }
}
// phase: 0 forward (qw), 1 backward partial sums, 2 backward (dw).  Blocks of a tensor: the table's partition for phase 1 (one partial per block), b1 for 0 / 2.
// M (and, phase 2, dM and the tie count) come finished from k_dorefa_w_final_multi.
__global__ __launch_bounds__(256) void k_dorefa_w_multi(const DwTable t, int phase) {
    __shared__ float sc[16];
    __shared__ double scd[16];
    const int ti = dw_find(t, blockIdx.x), lb = blockIdx.x - t.b0[ti], nbk = t.b0[ti + 1] - t.b0[ti], nb = t.nb[ti];
    const float* __restrict__ w = t.w[ti];
    const float* __restrict__ g = t.g[ti];
    float* __restrict__ ws = t.ws[ti];
    const float* __restrict__ th = t.th[ti];
    const long long n = t.n[ti];
    const float s = t.s;
    const float M = ws[0];
    if (phase == 0) {
        float* __restrict__ qw = t.out[ti];
        auto q1 = [&](float tt) { const float u = (tt / 2.f) / M + 0.5f; const float q = mn_rha(u / s) * s; return 2.f * q - 1.f; };
        if (th && dw_vec_ok(th, qw, nullptr, nullptr, n)) {
            const long long n4 = n >> 2;
            for (long long i = (long long)lb * 256 + threadIdx.x; i < n4; i += (long long)nbk * 256) {
                const float4 tt = *reinterpret_cast<const float4*>(th + 4 * i);
                *reinterpret_cast<float4*>(qw + 4 * i) = make_float4(q1(tt.x), q1(tt.y), q1(tt.z), q1(tt.w));
            }
        } else {
            for (long long i = (long long)lb * 256 + threadIdx.x; i < n; i += (long long)nbk * 256) qw[i] = q1(th ? th[i] : mn_tanh_cr(w[i]));
        }
    } else if (phase == 1) {
        double acc = 0.0;
        float ties = 0.f;
        // (the element -> block assignment and each thread's order of additions are the single-tensor kernel's: the partials round identically; four strides are
        //  FETCHED at a time so that eight loads are in flight per thread)
        const long long st = (long long)nbk * 256;
        long long i = (long long)lb * 256 + threadIdx.x;
        if (th) {
            for (; i + 3 * st < n; i += 4 * st) {
                float tv[4], gv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { tv[u] = th[i + u * st]; gv[u] = g[i + u * st]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float du = ((gv[u] * 2.f) * s) / s;
                    acc += (double)(-du * (tv[u] / 2.f) / (M * M));
                    ties += (fabsf(tv[u]) == M) ? 1.f : 0.f;
                }
            }
        }
        for (; i < n; i += st) {
            const float tt = th ? th[i] : mn_tanh_cr(w[i]);
            const float du = ((g[i] * 2.f) * s) / s;
            acc += (double)(-du * (tt / 2.f) / (M * M));
            ties += (fabsf(tt) == M) ? 1.f : 0.f;
        }
        acc = block_reduce(acc, OpAddD(), 0.0, scd);
        ties = block_reduce(ties, OpAddF(), 0.f, sc);
        if (threadIdx.x == 0) { ws[16 + nb + lb] = (float)acc; ws[16 + 2 * nb + lb] = ties; }
    } else {
        const float dM = ws[1], cnt = ws[2];
        const float share = dM / cnt;
        float* __restrict__ dw = t.out[ti];
        auto d1 = [&](float tt, float gi) {
            const float du = ((gi * 2.f) * s) / s;
            float dt = (du / M) / 2.f;
            if (fabsf(tt) == M) dt += share * mn_sign(tt);
            return dt * (1.f - tt * tt);
        };
        if (th && dw_vec_ok(th, g, dw, nullptr, n)) {
            const long long n4 = n >> 2;
            for (long long i = (long long)lb * 256 + threadIdx.x; i < n4; i += (long long)nbk * 256) {
                const float4 tt = *reinterpret_cast<const float4*>(th + 4 * i), gv = *reinterpret_cast<const float4*>(g + 4 * i);
                *reinterpret_cast<float4*>(dw + 4 * i) = make_float4(d1(tt.x, gv.x), d1(tt.y, gv.y), d1(tt.z, gv.z), d1(tt.w, gv.w));
            }
        } else {
            for (long long i = (long long)lb * 256 + threadIdx.x; i < n; i += (long long)nbk * 256) dw[i] = d1(th ? th[i] : mn_tanh_cr(w[i]), g[i]);
        }
    }
}
// table for the phases: `wide` != 0 gives every tensor up to 1024 blocks (elementwise phases 0 / 2), else exactly its partial-block count
static int dw_table(DwTable* t, const float* const* w, const float* const* g, float* const* out, float* const* ws, const int64_t* n, int count, int w_bits,
                    int wide, int* grid, const char* what, float* const* th = nullptr) {
    if (count < 1 || count > MN_DW_MAX_TENSORS || w_bits < 2 || w_bits > 31 || !w || !out || !ws || !n) MN_FAIL(MN_EINVAL, "%s: bad arguments (count=%d w_bits=%d)", what, count, w_bits);
    int b = 0;
    for (int i = 0; i < count; ++i) {
        if (!w[i] || !out[i] || !ws[i] || n[i] <= 0 || (g && !g[i])) MN_FAIL(MN_EINVAL, "%s: tensor %d invalid", what, i);
        t->w[i] = w[i]; t->g[i] = g ? g[i] : nullptr; t->out[i] = out[i]; t->ws[i] = ws[i]; t->n[i] = n[i]; t->th[i] = th ? th[i] : nullptr;
        t->nb[i] = mn_grid_for(n[i], 256, DW_NB);
        t->b0[i] = b;
        b += wide ? mn_grid_for(n[i], 256, 1024) : t->nb[i];
    }
    t->b0[count] = b; t->count = count; t->s = dorefa_scale(w_bits);
    *grid = b;
    return MN_OK;
}
extern "C" int mn_dorefa_w_fwd_multi(const float* const* w, float* const* qw, float* const* ws, const int64_t* n, int32_t count, int w_bits, mn_stream_t stream) {
    DwTable t;
    int grid;
    int rc = dw_table(&t, w, nullptr, qw, ws, n, count, w_bits, 0, &grid, "mn_dorefa_w_fwd_multi");
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_dorefa_w_absmax_multi, dim3(grid), dim3(256), 0, s, t);
    hipLaunchKernelGGL(k_dorefa_w_final_multi, dim3(count), dim3(256), 0, s, t, 0);
    if ((rc = dw_table(&t, w, nullptr, qw, ws, n, count, w_bits, 1, &grid, "mn_dorefa_w_fwd_multi"))) return rc;
    hipLaunchKernelGGL(k_dorefa_w_multi, dim3(grid), dim3(256), 0, s, t, 0);
    MN_CHECK_LAUNCH("mn_dorefa_w_fwd_multi");
    return MN_OK;
}
extern "C" int mn_dorefa_w_bwd_multi(const float* const* g, const float* const* w, float* const* dw, float* const* ws, const int64_t* n, int32_t count, int w_bits,
                                     mn_stream_t stream) {
    DwTable t;
    int grid;
    if (!g) MN_FAIL(MN_EINVAL, "mn_dorefa_w_bwd_multi: null gradient table");
    int rc = dw_table(&t, w, g, dw, ws, n, count, w_bits, 0, &grid, "mn_dorefa_w_bwd_multi");
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_dorefa_w_absmax_multi, dim3(grid), dim3(256), 0, s, t);
    hipLaunchKernelGGL(k_dorefa_w_final_multi, dim3(count), dim3(256), 0, s, t, 0);
    hipLaunchKernelGGL(k_dorefa_w_multi, dim3(grid), dim3(256), 0, s, t, 1);
    hipLaunchKernelGGL(k_dorefa_w_final_multi, dim3(count), dim3(256), 0, s, t, 1);
    if ((rc = dw_table(&t, w, g, dw, ws, n, count, w_bits, 1, &grid, "mn_dorefa_w_bwd_multi"))) return rc;
    hipLaunchKernelGGL(k_dorefa_w_multi, dim3(grid), dim3(256), 0, s, t, 2);
    MN_CHECK_LAUNCH("mn_dorefa_w_bwd_multi");
    return MN_OK;
}
// The same with tanh(w) cached for the whole step (th[i]: n[i] floats; the correctly rounded tanh is an fp64 evaluation -- five of them per weight and step
// were 0.45 ms of a resnet18 step): the forward writes the cache in its absmax pass, the backward -- given the forward's `ws` and `th` -- needs neither a tanh
// nor the absmax pass again (2 launches instead of 3).  Same arithmetic on the same values: bit-identical to the uncached entry points.
extern "C" int mn_dorefa_w_fwd_multi_cached(const float* const* w, float* const* qw, float* const* ws, float* const* th, const int64_t* n, int32_t count, int w_bits,
                                            mn_stream_t stream) {
    DwTable t;
    int grid;
    if (!th) MN_FAIL(MN_EINVAL, "mn_dorefa_w_fwd_multi_cached: null tanh cache table");
    int rc = dw_table(&t, w, nullptr, qw, ws, n, count, w_bits, 0, &grid, "mn_dorefa_w_fwd_multi_cached", th);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_dorefa_w_absmax_multi, dim3(grid), dim3(256), 0, s, t);
    hipLaunchKernelGGL(k_dorefa_w_final_multi, dim3(count), dim3(256), 0, s, t, 0);
    if ((rc = dw_table(&t, w, nullptr, qw, ws, n, count, w_bits, 1, &grid, "mn_dorefa_w_fwd_multi_cached", th))) return rc;
    hipLaunchKernelGGL(k_dorefa_w_multi, dim3(grid), dim3(256), 0, s, t, 0);
    MN_CHECK_LAUNCH("mn_dorefa_w_fwd_multi_cached");
    return MN_OK;
}
extern "C" int mn_dorefa_w_bwd_multi_cached(const float* const* g, const float* const* w, float* const* dw, float* const* ws, float* const* th, const int64_t* n,
                                            int32_t count, int w_bits, mn_stream_t stream) {
    DwTable t;
    int grid;
    if (!g || !th) MN_FAIL(MN_EINVAL, "mn_dorefa_w_bwd_multi_cached: null gradient / tanh cache table");
    int rc = dw_table(&t, w, g, dw, ws, n, count, w_bits, 0, &grid, "mn_dorefa_w_bwd_multi_cached", th);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_dorefa_w_multi, dim3(grid), dim3(256), 0, s, t, 1);          // (ws[0] still holds the forward's M)
    hipLaunchKernelGGL(k_dorefa_w_final_multi, dim3(count), dim3(256), 0, s, t, 1);
    if ((rc = dw_table(&t, w, g, dw, ws, n, count, w_bits, 1, &grid, "mn_dorefa_w_bwd_multi_cached", th))) return rc;
    hipLaunchKernelGGL(k_dorefa_w_multi, dim3(grid), dim3(256), 0, s, t, 2);
    MN_CHECK_LAUNCH("mn_dorefa_w_bwd_multi_cached");
    return MN_OK;
}

// ------------------------------------------------------------------------------------------------
// WbWtAb binary activation (wbwtab/quantize.py:11-36)
struct FBinActFwd { __device__ float operator()(float x) const { return (x < 0.f) ? -1.f : 1.f; } };  // 0, -0, NaN -> +1
struct FBinActBwd { __device__ float operator()(float g, float x) const { return (x >= 1.f || x <= -1.f) ? 0.f : g; } };
extern "C" int mn_binact_fwd(const float* x, float* y, int64_t n, mn_stream_t stream) {
    return launch_map1(x, y, n, FBinActFwd(), (hipStream_t)stream, "mn_binact_fwd");
}
extern "C" int mn_binact_bwd(const float* g, const float* x, float* dx, int64_t n, mn_stream_t stream) {
    return launch_map2(g, x, dx, n, FBinActBwd(), (hipStream_t)stream, "mn_binact_bwd");
}

// ------------------------------------------------------------------------------------------------
// Ternary weights (55-75, 132-146): one workgroup per output channel.
__global__ __launch_bounds__(256) void k_ternary_w_fwd(const float* __restrict__ w, float* __restrict__ qw, float* __restrict__ stats, int64_t K) {
    __shared__ double scd[16];
    const float* wr = w + (int64_t)blockIdx.x * K;
    float* qr = qw + (int64_t)blockIdx.x * K;
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < K; i += blockDim.x) s += (double)fabsf(wr[i]);
    s = block_reduce(s, OpAddD(), 0.0, scd);
    const float E = (float)s / (float)K;          // torch.mean = fp32 sum / n
    const float thr = E * 0.7f;
    double sa = 0.0, sc = 0.0;
    for (int64_t i = threadIdx.x; i < K; i += blockDim.x) {
        float a = fabsf(wr[i]);
        if (a > thr) { sa += (double)a; sc += 1.0; }
    }
    sa = block_reduce(sa, OpAddD(), 0.0, scd);
    sc = block_reduce(sc, OpAddD(), 0.0, scd);
    const float ssum = (float)sa, cnt = (float)sc;
    const float alpha = ssum / cnt;               // 0/0 = NaN for an all-zero channel, as in the reference
    for (int64_t i = threadIdx.x; i < K; i += blockDim.x) {
        float v = wr[i];
        float t = mn_sign(mn_sign(v + thr) + mn_sign(v - thr));
        qr[i] = t * alpha;
    }
    if (threadIdx.x == 0) {
        float* st = stats + (int64_t)blockIdx.x * 4;
        st[0] = alpha; st[1] = thr; st[2] = cnt; st[3] = ssum;
    }
}
__global__ __launch_bounds__(256) void k_ternary_w_bwd(const float* __restrict__ g, const float* __restrict__ w, const float* __restrict__ stats,
                                                       float* __restrict__ dw, int64_t K) {
    __shared__ double scd[16];
    const int64_t off = (int64_t)blockIdx.x * K;
    const float alpha = stats[blockIdx.x * 4 + 0], thr = stats[blockIdx.x * 4 + 1], cnt = stats[blockIdx.x * 4 + 2];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < K; i += blockDim.x) {
        float v = w[off + i];
        float t = mn_sign(mn_sign(v + thr) + mn_sign(v - thr));
        s += (double)(g[off + i] * t);
    }
    s = block_reduce(s, OpAddD(), 0.0, scd);
    const float share = (float)s / cnt;           // d(alpha)/d|w_i| = [|w_i|>thr]/cnt, upstream = sum(g*t)
    for (int64_t i = threadIdx.x; i < K; i += blockDim.x) {
        float v = w[off + i];
        float d = g[off + i] * alpha;
        if (fabsf(v) > thr) d += mn_sign(v) * share;
        dw[off + i] = d;
    }
}
// The same two kernels over SEVERAL weight tensors in one launch (a step of nin_gc quantizes 7 of them: 7 + 7 launches of ~5 us otherwise).
// The tensor table travels by value in the kernel arguments: nothing is allocated, the launch is graph-capturable.
#define MN_TERN_MAXT 32
struct TernTable {
    const float* w[MN_TERN_MAXT];
    const float* g[MN_TERN_MAXT];      // backward: upstream gradient
    float* out[MN_TERN_MAXT];          // forward: qw; backward: dw
    float* stats[MN_TERN_MAXT];
    int K[MN_TERN_MAXT];
    int row_end[MN_TERN_MAXT];         // cumulative row count
    int n;
};
template <int BWD>
__global__ __launch_bounds__(256) void k_ternary_w_multi(const TernTable t) {
    __shared__ double scd[16];
    int ti = 0;
    while (ti + 1 < t.n && (int)blockIdx.x >= t.row_end[ti]) ++ti;
    const int row = (int)blockIdx.x - (ti ? t.row_end[ti - 1] : 0);
    const int64_t K = t.K[ti];
    const float* wr = t.w[ti] + (int64_t)row * K;
    float* orow = t.out[ti] + (int64_t)row * K;
    float* st = t.stats[ti] + (int64_t)row * 4;
    if (!BWD) {
        double s = 0.0;
        for (int64_t i = threadIdx.x; i < K; i += blockDim.x) s += (double)fabsf(wr[i]);
        s = block_reduce(s, OpAddD(), 0.0, scd);
        const float E = (float)s / (float)K;
        const float thr = E * 0.7f;
        double sa = 0.0, sc = 0.0;
        for (int64_t i = threadIdx.x; i < K; i += blockDim.x) {
            float a = fabsf(wr[i]);
            if (a > thr) { sa += (double)a; sc += 1.0; }
        }
        sa = block_reduce(sa, OpAddD(), 0.0, scd);
        sc = block_reduce(sc, OpAddD(), 0.0, scd);
        const float ssum = (float)sa, cnt = (float)sc;
        const float alpha = ssum / cnt;
        for (int64_t i = threadIdx.x; i < K; i += blockDim.x) {
            float v = wr[i];
            float tt = mn_sign(mn_sign(v + thr) + mn_sign(v - thr));
            orow[i] = tt * alpha;
        }
        if (threadIdx.x == 0) { st[0] = alpha; st[1] = thr; st[2] = cnt; st[3] = ssum; }
    } else {
        const float* gr = t.g[ti] + (int64_t)row * K;
        const float alpha = st[0], thr = st[1], cnt = st[2];
        double s = 0.0;
        for (int64_t i = threadIdx.x; i < K; i += blockDim.x) {
            float v = wr[i];
            float tt = mn_sign(mn_sign(v + thr) + mn_sign(v - thr));
            s += (double)(gr[i] * tt);
        }
        s = block_reduce(s, OpAddD(), 0.0, scd);
        const float share = (float)s / cnt;
        for (int64_t i = threadIdx.x; i < K; i += blockDim.x) {
            float v = wr[i];
            float d = gr[i] * alpha;
            if (fabsf(v) > thr) d += mn_sign(v) * share;
            orow[i] = d;
        }
    }
}
static int ternary_multi(int bwd, const float* const* w, const float* const* g, float* const* out, float* const* stats, const int64_t* O, const int64_t* K,
                         int n, mn_stream_t stream, const char* what) {
    if (n <= 0 || n > MN_TERN_MAXT || !w || !out || !stats || !O || !K || (bwd && !g)) MN_FAIL(MN_EINVAL, "%s: bad arguments (1 .. %d tensors)", what, MN_TERN_MAXT);
    TernTable t;
    int64_t rows = 0;
    for (int i = 0; i < n; ++i) {
        if (!w[i] || !out[i] || !stats[i] || (bwd && !g[i]) || O[i] <= 0 || K[i] <= 0 || K[i] > 0x7fffffff) MN_FAIL(MN_EINVAL, "%s: bad tensor %d", what, i);
        rows += O[i];
        if (rows > 0x7fffffff) MN_FAIL(MN_EINVAL, "%s: too many rows", what);
        t.w[i] = w[i]; t.g[i] = bwd ? g[i] : nullptr; t.out[i] = out[i]; t.stats[i] = stats[i]; t.K[i] = (int)K[i]; t.row_end[i] = (int)rows;
    }
    t.n = n;
    if (bwd) hipLaunchKernelGGL(k_ternary_w_multi<1>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, t);
    else hipLaunchKernelGGL(k_ternary_w_multi<0>, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, t);
    MN_CHECK_LAUNCH(what);
    return MN_OK;
}
extern "C" int mn_ternary_w_fwd_multi(const float* const* w, float* const* qw, float* const* stats, const int64_t* O, const int64_t* K, int32_t n,
                                      mn_stream_t stream) {
    return ternary_multi(0, w, nullptr, qw, stats, O, K, n, stream, "mn_ternary_w_fwd_multi");
}
extern "C" int mn_ternary_w_bwd_multi(const float* const* g, const float* const* w, float* const* stats, float* const* dw, const int64_t* O, const int64_t* K,
                                      int32_t n, mn_stream_t stream) {
    return ternary_multi(1, w, g, dw, stats, O, K, n, stream, "mn_ternary_w_bwd_multi");
}
extern "C" int mn_ternary_w_fwd(const float* w, float* qw, float* stats, int64_t O, int64_t K, mn_stream_t stream) {
    if (O <= 0 || K <= 0 || !w || !qw || !stats) MN_FAIL(MN_EINVAL, "mn_ternary_w_fwd: bad arguments");
    hipLaunchKernelGGL(k_ternary_w_fwd, dim3((unsigned)O), dim3(256), 0, (hipStream_t)stream, w, qw, stats, K);
    MN_CHECK_LAUNCH("mn_ternary_w_fwd");
    return MN_OK;
}
extern "C" int mn_ternary_w_bwd(const float* g, const float* w, const float* stats, float* dw, int64_t O, int64_t K, mn_stream_t stream) {
    if (O <= 0 || K <= 0 || !g || !w || !stats || !dw) MN_FAIL(MN_EINVAL, "mn_ternary_w_bwd: bad arguments");
    hipLaunchKernelGGL(k_ternary_w_bwd, dim3((unsigned)O), dim3(256), 0, (hipStream_t)stream, g, w, stats, dw, K);
    MN_CHECK_LAUNCH("mn_ternary_w_bwd");
    return MN_OK;
}

// ------------------------------------------------------------------------------------------------
// Binary weights (98-102, 121-130): in-place mean-centre over Cin + clamp, then sign * mean|w|.
// One workgroup per output channel; w row = [C][R].
__device__ __forceinline__ void binary_w_fwd_row(float* __restrict__ w, float* __restrict__ qw, float* __restrict__ alpha_out, int C, int R, int row, double* colsum) {
    __shared__ double scd[16];
    const int64_t K = (int64_t)C * R;
    float* wr = w + (int64_t)row * K;
    float* qr = qw + (int64_t)row * K;
    // column (kh,kw) sums over the Cin axis: thread t owns column t % R, rows t / R, t / R + stride ...
    for (int r = threadIdx.x; r < R; r += blockDim.x) colsum[r] = 0.0;
    __syncthreads();
    // deterministic two-level: partial per thread -> LDS slot -> ordered sum by one thread per column
    double* part = colsum + R;           // [blockDim.x]
    {
        const int lanes_per_col = blockDim.x / R > 0 ? blockDim.x / R : 1;
        const int col = threadIdx.x % R, sub = threadIdx.x / R;
        double p = 0.0;
        if (sub < lanes_per_col)
            for (int c = sub; c < C; c += lanes_per_col) p += (double)wr[(int64_t)c * R + col];
        part[threadIdx.x] = p;
        __syncthreads();
        if (threadIdx.x < R) {
            double s = 0.0;
            for (int j = 0; j < lanes_per_col; ++j) {
                int t = j * R + threadIdx.x;
                if (t < (int)blockDim.x) s += part[t];
            }
            colsum[threadIdx.x] = s;
        }
        __syncthreads();
    }
    double sabs = 0.0;
    for (int64_t i = threadIdx.x; i < K; i += blockDim.x) {
        const int col = (int)(i % R);
        const float mean = (float)colsum[col] / (float)C;
        float v = wr[i] - mean;
        v = fminf(fmaxf(v, -1.f), 1.f);
        wr[i] = v;                                  // the reference mutates weight.data here
        sabs += (double)fabsf(v);
    }
    sabs = block_reduce(sabs, OpAddD(), 0.0, scd);
    const float alpha = (float)sabs / (float)K;
    for (int64_t i = threadIdx.x; i < K; i += blockDim.x) qr[i] = ((wr[i] < 0.f) ? -1.f : 1.f) * alpha;
    if (threadIdx.x == 0) alpha_out[row] = alpha;
}
__global__ __launch_bounds__(256) void k_binary_w_fwd(float* __restrict__ w, float* __restrict__ qw, float* __restrict__ alpha_out,
                                                      int C, int R) {
    HIP_DYNAMIC_SHARED(double, colsum)   // [R] column sums, then [blockDim] partials
    binary_w_fwd_row(w, qw, alpha_out, C, R, (int)blockIdx.x, colsum);
}
__device__ __forceinline__ void binary_w_bwd_row(const float* __restrict__ g, const float* __restrict__ w, const float* __restrict__ alpha_in,
                                                 float* __restrict__ dw, int64_t K, int row) {
    __shared__ double scd[16];
    const int64_t off = (int64_t)row * K;
    const float alpha = alpha_in[row];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < K; i += blockDim.x) s += (double)(g[off + i] * ((w[off + i] < 0.f) ? -1.f : 1.f));
    s = block_reduce(s, OpAddD(), 0.0, scd);
    const float share = (float)s / (float)K;      // d mean|w| / dw_i = sign(w_i)/K
    for (int64_t i = threadIdx.x; i < K; i += blockDim.x) dw[off + i] = g[off + i] * alpha + mn_sign(w[off + i]) * share;
}
__global__ __launch_bounds__(256) void k_binary_w_bwd(const float* __restrict__ g, const float* __restrict__ w, const float* __restrict__ alpha_in,
                                                      float* __restrict__ dw, int64_t K) {
    binary_w_bwd_row(g, w, alpha_in, dw, K, (int)blockIdx.x);
}
// The same two kernels over SEVERAL weight tensors in one launch (round 6: the W = 2 nin_gc step spent 7 + 7 launches here, and -- its code images being packed per
// conv call because the quantizer had not run yet at the start of the step -- 10 more in k_qg_pack: 186 us of a 1.9 ms step).  Same arithmetic per tensor.
struct BinTable {
    float* w[MN_TERN_MAXT];            // forward: mutated in place (the reference's weight.data), backward: read
    const float* g[MN_TERN_MAXT];
    float* out[MN_TERN_MAXT];          // forward: qw; backward: dw
    float* alpha[MN_TERN_MAXT];
    int C[MN_TERN_MAXT], R[MN_TERN_MAXT];
    int row_end[MN_TERN_MAXT];
    int n;
};
template <int BWD>
__global__ __launch_bounds__(256) void k_binary_w_multi(const BinTable t) {
    HIP_DYNAMIC_SHARED(double, colsum)
    int ti = 0;
    while (ti + 1 < t.n && (int)blockIdx.x >= t.row_end[ti]) ++ti;
    const int row = (int)blockIdx.x - (ti ? t.row_end[ti - 1] : 0);
    if (!BWD) binary_w_fwd_row(t.w[ti], t.out[ti], t.alpha[ti], t.C[ti], t.R[ti], row, colsum);
    else binary_w_bwd_row(t.g[ti], t.w[ti], t.alpha[ti], t.out[ti], (int64_t)t.C[ti] * t.R[ti], row);
}
static int binary_multi(int bwd, float* const* w, const float* const* g, float* const* out, float* const* alpha, const int64_t* O, const int64_t* Cn, const int64_t* R,
                        int n, mn_stream_t stream, const char* what) {
    if (n <= 0 || n > MN_TERN_MAXT || !w || !out || !alpha || !O || !Cn || !R || (bwd && !g)) MN_FAIL(MN_EINVAL, "%s: bad arguments (1 .. %d tensors)", what, MN_TERN_MAXT);
    BinTable t;
    int64_t rows = 0, rmax = 1;
    for (int i = 0; i < n; ++i) {
        if (!w[i] || !out[i] || !alpha[i] || (bwd && !g[i]) || O[i] <= 0 || Cn[i] <= 0 || R[i] <= 0 || R[i] > 256 || Cn[i] * R[i] > 0x7fffffff) MN_FAIL(MN_EINVAL, "%s: bad tensor %d", what, i);
        rows += O[i];
        if (rows > 0x7fffffff) MN_FAIL(MN_EINVAL, "%s: too many rows", what);
        t.w[i] = w[i]; t.g[i] = bwd ? g[i] : nullptr; t.out[i] = out[i]; t.alpha[i] = alpha[i]; t.C[i] = (int)Cn[i]; t.R[i] = (int)R[i]; t.row_end[i] = (int)rows;
        if (R[i] > rmax) rmax = R[i];
    }
    t.n = n;
    const size_t sh = (size_t)(rmax + 256) * sizeof(double);
    if (bwd) hipLaunchKernelGGL(k_binary_w_multi<1>, dim3((unsigned)rows), dim3(256), sh, (hipStream_t)stream, t);
    else hipLaunchKernelGGL(k_binary_w_multi<0>, dim3((unsigned)rows), dim3(256), sh, (hipStream_t)stream, t);
    MN_CHECK_LAUNCH(what);
    return MN_OK;
}
extern "C" int mn_binary_w_fwd_multi(float* const* w, float* const* qw, float* const* alpha, const int64_t* O, const int64_t* C, const int64_t* R, int32_t n,
                                     mn_stream_t stream) {
    return binary_multi(0, w, nullptr, qw, alpha, O, C, R, n, stream, "mn_binary_w_fwd_multi");
}
extern "C" int mn_binary_w_bwd_multi(const float* const* g, const float* const* w, float* const* alpha, float* const* dw, const int64_t* O, const int64_t* C,
                                     const int64_t* R, int32_t n, mn_stream_t stream) {
    return binary_multi(1, const_cast<float* const*>(reinterpret_cast<const float* const*>(w)), g, dw, alpha, O, C, R, n, stream, "mn_binary_w_bwd_multi");
}
extern "C" int mn_binary_w_fwd(float* w, float* qw, float* alpha, int64_t O, int64_t C, int64_t R, mn_stream_t stream) {
    if (O <= 0 || C <= 0 || R <= 0 || R > 256 || !w || !qw || !alpha) MN_FAIL(MN_EINVAL, "mn_binary_w_fwd: bad arguments (R=%lld must be <= 256)", (long long)R);
    size_t sh = (size_t)(R + 256) * sizeof(double);
    hipLaunchKernelGGL(k_binary_w_fwd, dim3((unsigned)O), dim3(256), sh, (hipStream_t)stream, w, qw, alpha, (int)C, (int)R);
    MN_CHECK_LAUNCH("mn_binary_w_fwd");
    return MN_OK;
}
extern "C" int mn_binary_w_bwd(const float* g, const float* w, const float* alpha, float* dw, int64_t O, int64_t K, mn_stream_t stream) {
    if (O <= 0 || K <= 0 || !g || !w || !alpha || !dw) MN_FAIL(MN_EINVAL, "mn_binary_w_bwd: bad arguments");
    hipLaunchKernelGGL(k_binary_w_bwd, dim3((unsigned)O), dim3(256), 0, (hipStream_t)stream, g, w, alpha, dw, K);
    MN_CHECK_LAUNCH("mn_binary_w_bwd");
    return MN_OK;
}

// ------------------------------------------------------------------------------------------------
// IAO observers (15-113).  rows == 1: two-stage grid reduction over the whole tensor (wave shuffles +
// LDS, no atomics -> deterministic); rows > 1: one workgroup per row.
static const int OBS_NB = 1024;
extern "C" int64_t mn_iao_observe_ws_floats(int64_t rows, int64_t) { return rows == 1 ? 2 * OBS_NB : 0; }

__global__ __launch_bounds__(256) void k_minmax_partial(const float* __restrict__ x, int64_t n, int vec, float* __restrict__ ws) {
    __shared__ float sc[16];
    float lo = INFINITY, hi = -INFINITY;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (vec) {
        const int64_t n4 = n >> 2;
        const float4* x4 = reinterpret_cast<const float4*>(x);
        for (int64_t j = i0; j < n4; j += stride) {
            float4 v = x4[j];
            lo = OpMinF()(OpMinF()(lo, v.x), OpMinF()(OpMinF()(v.y, v.z), v.w));
            hi = OpMaxF()(OpMaxF()(hi, v.x), OpMaxF()(OpMaxF()(v.y, v.z), v.w));
        }
        for (int64_t j = (n4 << 2) + i0; j < n; j += stride) { lo = OpMinF()(lo, x[j]); hi = OpMaxF()(hi, x[j]); }
    } else {
        for (int64_t j = i0; j < n; j += stride) { lo = OpMinF()(lo, x[j]); hi = OpMaxF()(hi, x[j]); }
    }
    lo = block_reduce(lo, OpMinF(), INFINITY, sc);
    hi = block_reduce(hi, OpMaxF(), -INFINITY, sc);
    if (threadIdx.x == 0) { ws[blockIdx.x] = lo; ws[OBS_NB + blockIdx.x] = hi; }
}
__global__ __launch_bounds__(256) void k_minmax_final(const float* __restrict__ ws, int nb, int obs_kind, int first, double momentum,
                                                      float* __restrict__ min_val, float* __restrict__ max_val) {
    __shared__ float sc[16];
    float lo = INFINITY, hi = -INFINITY;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) { lo = OpMinF()(lo, ws[i]); hi = OpMaxF()(hi, ws[OBS_NB + i]); }
    lo = block_reduce(lo, OpMinF(), INFINITY, sc);
    hi = block_reduce(hi, OpMaxF(), -INFINITY, sc);
    if (threadIdx.x == 0) observer_update(obs_kind, first, momentum, lo, hi, min_val, max_val);
}
__global__ __launch_bounds__(256) void k_minmax_rows(const float* __restrict__ x, int64_t cols, int obs_kind, int first, double momentum,
                                                     float* __restrict__ min_val, float* __restrict__ max_val) {
    __shared__ float sc[16];
    const float* xr = x + (int64_t)blockIdx.x * cols;
    float lo = INFINITY, hi = -INFINITY;
    for (int64_t j = threadIdx.x; j < cols; j += blockDim.x) { lo = OpMinF()(lo, xr[j]); hi = OpMaxF()(hi, xr[j]); }
    lo = block_reduce(lo, OpMinF(), INFINITY, sc);
    hi = block_reduce(hi, OpMaxF(), -INFINITY, sc);
    if (threadIdx.x == 0) observer_update(obs_kind, first, momentum, lo, hi, min_val + blockIdx.x, max_val + blockIdx.x);
}
extern "C" int mn_iao_observe(const float* x, int64_t rows, int64_t cols, int obs_kind, int first, double momentum,
                              float* min_val, float* max_val, float* ws, mn_stream_t stream) {
    if (rows <= 0 || cols <= 0 || !x || !min_val || !max_val || (obs_kind != 0 && obs_kind != 1)) MN_FAIL(MN_EINVAL, "mn_iao_observe: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (rows == 1) {
        if (!ws) MN_FAIL(MN_EINVAL, "mn_iao_observe: workspace required for rows == 1");
        int vec = aligned16(x);
        int nb = mn_grid_for(vec ? (cols + 3) / 4 : cols, 256 * 4, OBS_NB);
        hipLaunchKernelGGL(k_minmax_partial, dim3(nb), dim3(256), 0, s, x, cols, vec, ws);
        hipLaunchKernelGGL(k_minmax_final, dim3(1), dim3(256), 0, s, ws, nb, obs_kind, first, momentum, min_val, max_val);
    } else {
        hipLaunchKernelGGL(k_minmax_rows, dim3((unsigned)rows), dim3(256), 0, s, x, cols, obs_kind, first, momentum, min_val, max_val);
    }
    MN_CHECK_LAUNCH("mn_iao_observe");
    return MN_OK;
}

// the observer update from per-block (min, max) partials written by the kernel that PRODUCED the tensor (mn_bnrelu_fwd_mm, mn_iao_qadd_fwd_mm): mm[0 .. count)
// minima, mm[count .. 2 count) maxima.  min / max are exact and order-free, so this equals mn_iao_observe on the tensor itself bit for bit -- without reading it.
__global__ __launch_bounds__(256) void k_minmax_from_partials(const float* __restrict__ mm, int count, int obs_kind, int first, double momentum,
                                                              float* __restrict__ min_val, float* __restrict__ max_val, int q_type, float quant_range,
                                                              float* __restrict__ scale, float* __restrict__ zero_point, float* __restrict__ qp) {
    __shared__ float sc[16];
    mn_obs_partials_tail(mm, count, obs_kind, first, momentum, min_val, max_val, q_type, quant_range, scale, zero_point, qp, sc);
}
extern "C" int mn_iao_observe_partials(const float* mm, int64_t count, int obs_kind, int first, double momentum, float* min_val, float* max_val, mn_stream_t stream) {
    if (!mm || count <= 0 || count > (1 << 24) || !min_val || !max_val || (obs_kind != 0 && obs_kind != 1)) MN_FAIL(MN_EINVAL, "mn_iao_observe_partials: bad arguments");
    hipLaunchKernelGGL(k_minmax_from_partials, dim3(1), dim3(256), 0, (hipStream_t)stream, mm, (int)count, obs_kind, first, momentum, min_val, max_val, 0, 1.f,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr);
    MN_CHECK_LAUNCH("mn_iao_observe_partials");
    return MN_OK;
}
// the same + the per-tensor quantizer's update_qparams (scale / zero_point recomputed from the updated range; qp = {scale, zero_point, lo, hi}) in ONE launch
extern "C" int mn_iao_observe_partials_qparams(const float* mm, int64_t count, int obs_kind, int first, double momentum, float* min_val, float* max_val, int bits,
                                               int q_type, int is_act, float* scale, float* zero_point, float* qp, mn_stream_t stream) {
    if (!mm || count <= 0 || count > (1 << 24) || !min_val || !max_val || !scale || !zero_point || !qp || bits < 2 || bits > 24 || (obs_kind != 0 && obs_kind != 1) ||
        (q_type != 0 && q_type != 1))
        MN_FAIL(MN_EINVAL, "mn_iao_observe_partials_qparams: bad arguments");
    const IaoRange r = iao_range(bits, q_type, is_act);
    const float qr = (q_type == 0) ? (float)((double)(r.qmax - r.qmin) / 2.0) : (float)(r.qmax - r.qmin);
    hipLaunchKernelGGL(k_minmax_from_partials, dim3(1), dim3(256), 0, (hipStream_t)stream, mm, (int)count, obs_kind, first, momentum, min_val, max_val, q_type, qr, scale,
                       zero_point, qp);
    MN_CHECK_LAUNCH("mn_iao_observe_partials_qparams");
    return MN_OK;
}

__global__ __launch_bounds__(256) void k_iao_qparams(const float* __restrict__ min_val, const float* __restrict__ max_val, int64_t rows,
                                                     int q_type, float quant_range, int update, float* __restrict__ scale,
                                                     float* __restrict__ zero_point, float* __restrict__ qp) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (int64_t)gridDim.x * blockDim.x)
        iao_qparams_row(min_val[i], max_val[i], q_type, quant_range, update, scale + i, zero_point + i, qp + i * 4);
}
extern "C" int mn_iao_qparams(const float* min_val, const float* max_val, int64_t rows, int bits, int q_type, int is_act,
                              int update, float* scale, float* zero_point, float* qp, mn_stream_t stream) {
    if (rows <= 0 || bits < 2 || bits > 24 || !min_val || !max_val || !scale || !zero_point || !qp || (q_type != 0 && q_type != 1))
        MN_FAIL(MN_EINVAL, "mn_iao_qparams: bad arguments (bits=%d)", bits);
    IaoRange r = iao_range(bits, q_type, is_act);
    // python: float(qmax - qmin) / 2 for symmetric, float(qmax - qmin) for asymmetric; the fp32 tensor is divided by fp32(that)
    float qr = (q_type == 0) ? (float)((double)(r.qmax - r.qmin) / 2.0) : (float)(r.qmax - r.qmin);
    hipLaunchKernelGGL(k_iao_qparams, dim3(mn_grid_for(rows, 256, 64)), dim3(256), 0, (hipStream_t)stream, min_val, max_val, rows,
                       q_type, qr, update, scale, zero_point, qp);
    MN_CHECK_LAUNCH("mn_iao_qparams");
    return MN_OK;
}

// fake-quant (227-239) and its clip-STE backward; x viewed as [rows][cols] with per-row qp
__global__ __launch_bounds__(256) void k_iao_fq_fwd(const float* __restrict__ x, float* __restrict__ y, int64_t rows, int64_t cols,
                                                    const float* __restrict__ qp, float qmin, float qmax, int vec) {
    for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
        const float sc = qp[row * 4], zp = qp[row * 4 + 1];
        const float* xr = x + row * cols;
        float* yr = y + row * cols;
        const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
        if (vec) {
            const int64_t n4 = cols >> 2;
            for (int64_t j = t0; j < n4; j += stride) {
                float4 v = reinterpret_cast<const float4*>(xr)[j], o;
                o.x = iao_fq(v.x, sc, zp, qmin, qmax); o.y = iao_fq(v.y, sc, zp, qmin, qmax);
                o.z = iao_fq(v.z, sc, zp, qmin, qmax); o.w = iao_fq(v.w, sc, zp, qmin, qmax);
                reinterpret_cast<float4*>(yr)[j] = o;
            }
            for (int64_t j = (n4 << 2) + t0; j < cols; j += stride) yr[j] = iao_fq(xr[j], sc, zp, qmin, qmax);
        } else {
            for (int64_t j = t0; j < cols; j += stride) yr[j] = iao_fq(xr[j], sc, zp, qmin, qmax);
        }
    }
}
__global__ __launch_bounds__(256) void k_iao_fq_bwd(const float* __restrict__ g, const float* __restrict__ x, float* __restrict__ dx,
                                                    int64_t rows, int64_t cols, const float* __restrict__ qp, float qmin, float qmax, int vec) {
    for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
        const float sc = qp[row * 4], zp = qp[row * 4 + 1], lo = qp[row * 4 + 2], hi = qp[row * 4 + 3];
        const float* xr = x + row * cols;
        const float* gr = g + row * cols;
        float* dr = dx + row * cols;
        const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
        if (vec) {
            const int64_t n4 = cols >> 2;
            for (int64_t j = t0; j < n4; j += stride) {
                float4 v = reinterpret_cast<const float4*>(xr)[j], gg = reinterpret_cast<const float4*>(gr)[j], o;
                o.x = iao_fq_grad(gg.x, v.x, sc, zp, lo, hi, qmin, qmax); o.y = iao_fq_grad(gg.y, v.y, sc, zp, lo, hi, qmin, qmax);
                o.z = iao_fq_grad(gg.z, v.z, sc, zp, lo, hi, qmin, qmax); o.w = iao_fq_grad(gg.w, v.w, sc, zp, lo, hi, qmin, qmax);
                reinterpret_cast<float4*>(dr)[j] = o;
            }
            for (int64_t j = (n4 << 2) + t0; j < cols; j += stride) dr[j] = iao_fq_grad(gr[j], xr[j], sc, zp, lo, hi, qmin, qmax);
        } else {
            for (int64_t j = t0; j < cols; j += stride) dr[j] = iao_fq_grad(gr[j], xr[j], sc, zp, lo, hi, qmin, qmax);
        }
    }
}
static void fq_grid(int64_t rows, int64_t cols, dim3* grid, int* vec, const void* a, const void* b, const void* c) {
    *vec = aligned16(a) && aligned16(b) && (c ? aligned16(c) : 1) && (rows == 1 || cols % 4 == 0);
    int per_row = mn_grid_for(*vec ? (cols + 3) / 4 : cols, 256, EW_GRID_CAP);
    int64_t gy = EW_GRID_CAP / per_row;
    if (gy < 1) gy = 1;
    if (gy > rows) gy = rows;
    *grid = dim3((unsigned)per_row, (unsigned)gy);
}
extern "C" int mn_iao_fq_fwd(const float* x, float* y, int64_t rows, int64_t cols, const float* qp, int bits, int q_type,
                             int is_act, mn_stream_t stream) {
    if (rows <= 0 || cols <= 0 || !x || !y || !qp || bits < 2 || bits > 24) MN_FAIL(MN_EINVAL, "mn_iao_fq_fwd: bad arguments");
    IaoRange r = iao_range(bits, q_type, is_act);
    dim3 grid;
    int vec;
    fq_grid(rows, cols, &grid, &vec, x, y, nullptr);
    hipLaunchKernelGGL(k_iao_fq_fwd, grid, dim3(256), 0, (hipStream_t)stream, x, y, rows, cols, qp, r.qmin, r.qmax, vec);
    MN_CHECK_LAUNCH("mn_iao_fq_fwd");
    return MN_OK;
}
extern "C" int mn_iao_fq_bwd(const float* g, const float* x, float* dx, int64_t rows, int64_t cols, const float* qp, int bits,
                             int q_type, int is_act, mn_stream_t stream) {
    if (rows <= 0 || cols <= 0 || !g || !x || !dx || !qp || bits < 2 || bits > 24) MN_FAIL(MN_EINVAL, "mn_iao_fq_bwd: bad arguments");
    IaoRange r = iao_range(bits, q_type, is_act);
    dim3 grid;
    int vec;
    fq_grid(rows, cols, &grid, &vec, g, x, dx);
    hipLaunchKernelGGL(k_iao_fq_bwd, grid, dim3(256), 0, (hipStream_t)stream, g, x, dx, rows, cols, qp, r.qmin, r.qmax, vec);
    MN_CHECK_LAUNCH("mn_iao_fq_bwd");
    return MN_OK;
}
__global__ void k_iao_union(const float* a0, const float* a1, const float* b0, const float* b1, float* o0, float* o1) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { *o0 = OpMinF()(*a0, *b0); *o1 = OpMaxF()(*a1, *b1); }
}
extern "C" int mn_iao_union_range(const float* min_a, const float* max_a, const float* min_b, const float* max_b,
                                  float* min_out, float* max_out, mn_stream_t stream) {
    if (!min_a || !max_a || !min_b || !max_b || !min_out || !max_out) MN_FAIL(MN_EINVAL, "mn_iao_union_range: null pointer");
    hipLaunchKernelGGL(k_iao_union, dim3(1), dim3(64), 0, (hipStream_t)stream, min_a, max_a, min_b, max_b, min_out, max_out);
    MN_CHECK_LAUNCH("mn_iao_union_range");
    return MN_OK;
}

// ------------------------------------------------------------------------------------------------
// The IAO weight path of a whole net in one launch per direction (per-channel weight quantizers: wqaq/iao/quantize.py:15-36 observer, 293-321 qparams, 227-239
// fake-quant): a block owns one output channel (row) of one tensor -- min / max of the row, the observer update (running min / max, or moving average), scale /
// zero_point / the {scale, zp, lo, hi} snapshot, and the fake-quantised row -- instead of mn_iao_observe + mn_iao_qparams + mn_iao_fq_fwd per layer.  Same
// arithmetic per row: bit-identical to those entry points.
#define MN_IW_MAX 32
struct IwTable {
    const float* w[MN_IW_MAX]; const float* g[MN_IW_MAX]; float* out[MN_IW_MAX];
    float *min_val[MN_IW_MAX], *max_val[MN_IW_MAX], *scale[MN_IW_MAX], *zero_point[MN_IW_MAX], *qp[MN_IW_MAX];
    int rows[MN_IW_MAX], first[MN_IW_MAX], row0[MN_IW_MAX + 1];
    long long cols[MN_IW_MAX];
    int count, obs_kind, q_type; float quant_range, qmin, qmax; double momentum;
};
__device__ __forceinline__ int iw_find(const IwTable& t, int b) { int i = 0; while (i + 1 < t.count && t.row0[i + 1] <= b) ++i; return i; }
__global__ __launch_bounds__(256) void k_iao_w_fwd_multi(const IwTable t) {
    __shared__ float sc[16];
    __shared__ float sq[2];
    const int ti = iw_find(t, blockIdx.x), row = blockIdx.x - t.row0[ti];
    const long long cols = t.cols[ti];
    const float* xr = t.w[ti] + (long long)row * cols;
    float lo = INFINITY, hi = -INFINITY;
    for (long long j = threadIdx.x; j < cols; j += 256) { lo = OpMinF()(lo, xr[j]); hi = OpMaxF()(hi, xr[j]); }
    lo = block_reduce(lo, OpMinF(), INFINITY, sc);
    hi = block_reduce(hi, OpMaxF(), -INFINITY, sc);
    if (threadIdx.x == 0) {
        observer_update(t.obs_kind, t.first[ti], t.momentum, lo, hi, t.min_val[ti] + row, t.max_val[ti] + row);
        float* qp = t.qp[ti] + 4 * row;
        iao_qparams_row(t.min_val[ti][row], t.max_val[ti][row], t.q_type, t.quant_range, 1, t.scale[ti] + row, t.zero_point[ti] + row, qp);
        sq[0] = qp[0]; sq[1] = qp[1];
    }
    __syncthreads();
    const float s_ = sq[0], zp = sq[1];
    float* yr = t.out[ti] + (long long)row * cols;
    for (long long j = threadIdx.x; j < cols; j += 256) yr[j] = iao_fq(xr[j], s_, zp, t.qmin, t.qmax);
}
__global__ __launch_bounds__(256) void k_iao_w_bwd_multi(const IwTable t) {
    const int ti = iw_find(t, blockIdx.x), row = blockIdx.x - t.row0[ti];
    const long long cols = t.cols[ti];
    const float* qp = t.qp[ti] + 4 * row;
    const float s_ = qp[0], zp = qp[1], lo = qp[2], hi = qp[3];
    const float* xr = t.w[ti] + (long long)row * cols;
    const float* gr = t.g[ti] + (long long)row * cols;
    float* dr = t.out[ti] + (long long)row * cols;
    for (long long j = threadIdx.x; j < cols; j += 256) dr[j] = iao_fq_grad(gr[j], xr[j], s_, zp, lo, hi, t.qmin, t.qmax);
}
static int iw_table(IwTable* t, const float* const* w, const float* const* g, float* const* out, float* const* min_val, float* const* max_val, float* const* scale,
                    float* const* zero_point, float* const* qp, const int64_t* rows, const int64_t* cols, const int32_t* first, int count, int obs_kind, double momentum,
                    int bits, int q_type, const char* what) {
    if (count < 1 || count > MN_IW_MAX || !w || !out || !qp || !rows || !cols || bits < 2 || bits > 24 || (q_type != 0 && q_type != 1) || (obs_kind != 0 && obs_kind != 1))
        MN_FAIL(MN_EINVAL, "%s: bad arguments (count=%d bits=%d)", what, count, bits);
    int b = 0;
    for (int i = 0; i < count; ++i) {
        if (!w[i] || !out[i] || !qp[i] || rows[i] <= 0 || cols[i] <= 0 || (g && !g[i]) || (min_val && (!min_val[i] || !max_val[i] || !scale[i] || !zero_point[i])))
            MN_FAIL(MN_EINVAL, "%s: tensor %d invalid", what, i);
        t->w[i] = w[i]; t->g[i] = g ? g[i] : nullptr; t->out[i] = out[i]; t->qp[i] = qp[i];
        t->min_val[i] = min_val ? min_val[i] : nullptr; t->max_val[i] = max_val ? max_val[i] : nullptr;
        t->scale[i] = scale ? scale[i] : nullptr; t->zero_point[i] = zero_point ? zero_point[i] : nullptr;
        t->rows[i] = (int)rows[i]; t->cols[i] = cols[i]; t->first[i] = first ? first[i] : 0; t->row0[i] = b;
        b += (int)rows[i];
    }
    t->row0[count] = b; t->count = count; t->obs_kind = obs_kind; t->q_type = q_type; t->momentum = momentum;
    const IaoRange r = iao_range(bits, q_type, 0);
    t->qmin = r.qmin; t->qmax = r.qmax;
    t->quant_range = (q_type == 0) ? (float)((double)(r.qmax - r.qmin) / 2.0) : (float)(r.qmax - r.qmin);
    return MN_OK;
}
extern "C" int mn_iao_w_fwd_multi(const float* const* w, float* const* qw, float* const* min_val, float* const* max_val, float* const* scale, float* const* zero_point,
                                  float* const* qp, const int64_t* rows, const int64_t* cols, const int32_t* first, int32_t count, int obs_kind, double momentum,
                                  int bits, int q_type, mn_stream_t stream) {
    IwTable t;
    if (!min_val || !max_val || !scale || !zero_point) MN_FAIL(MN_EINVAL, "mn_iao_w_fwd_multi: null buffer table");
    int rc = iw_table(&t, w, nullptr, qw, min_val, max_val, scale, zero_point, qp, rows, cols, first, count, obs_kind, momentum, bits, q_type, "mn_iao_w_fwd_multi");
    if (rc) return rc;
    mn_set_last_kernel("k_iao_w_fwd_multi");
    hipLaunchKernelGGL(k_iao_w_fwd_multi, dim3((unsigned)t.row0[count]), dim3(256), 0, (hipStream_t)stream, t);
    MN_CHECK_LAUNCH("mn_iao_w_fwd_multi");
    return MN_OK;
}
extern "C" int mn_iao_w_bwd_multi(const float* const* g, const float* const* w, float* const* dw, float* const* qp, const int64_t* rows, const int64_t* cols,
                                  int32_t count, int bits, int q_type, mn_stream_t stream) {
    IwTable t;
    if (!g) MN_FAIL(MN_EINVAL, "mn_iao_w_bwd_multi: null gradient table");
    int rc = iw_table(&t, w, g, dw, nullptr, nullptr, nullptr, nullptr, qp, rows, cols, nullptr, count, 0, 0.0, bits, q_type, "mn_iao_w_bwd_multi");
    if (rc) return rc;
    mn_set_last_kernel("k_iao_w_bwd_multi");
    hipLaunchKernelGGL(k_iao_w_bwd_multi, dim3((unsigned)t.row0[count]), dim3(256), 0, (hipStream_t)stream, t);
    MN_CHECK_LAUNCH("mn_iao_w_bwd_multi");
    return MN_OK;
}

// ------------------------------------------------------------------------------------------------
// QuantAdd (wqaq/iao/quantize.py:1484-1498) in three launches instead of nine: both input observers (per-tensor min / max, running or moving-average update),
// their union range, the shared quantizer's qparams -- k_qadd_partial + k_qadd_final -- and out = fq(res) + fq(shortcut) in one pass (k_qadd_fwd); backward: both
// clip-STE gradients from one read of g (k_qadd_bwd).  Same arithmetic as mn_iao_observe x 2 + mn_iao_union_range + mn_iao_qparams + mn_iao_fq_fwd x 2 + add.
__global__ __launch_bounds__(256) void k_qadd_partial(const float* __restrict__ a, const float* __restrict__ b, int64_t n4, float* __restrict__ ws) {
    __shared__ float sc[16];
    float la = INFINITY, ha = -INFINITY, lb = INFINITY, hb = -INFINITY;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n4; j += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(a)[j], w = reinterpret_cast<const float4*>(b)[j];
        la = OpMinF()(OpMinF()(la, v.x), OpMinF()(OpMinF()(v.y, v.z), v.w)); ha = OpMaxF()(OpMaxF()(ha, v.x), OpMaxF()(OpMaxF()(v.y, v.z), v.w));
        lb = OpMinF()(OpMinF()(lb, w.x), OpMinF()(OpMinF()(w.y, w.z), w.w)); hb = OpMaxF()(OpMaxF()(hb, w.x), OpMaxF()(OpMaxF()(w.y, w.z), w.w));
    }
    la = block_reduce(la, OpMinF(), INFINITY, sc); ha = block_reduce(ha, OpMaxF(), -INFINITY, sc);
    lb = block_reduce(lb, OpMinF(), INFINITY, sc); hb = block_reduce(hb, OpMaxF(), -INFINITY, sc);
    if (threadIdx.x == 0) { ws[blockIdx.x] = la; ws[OBS_NB + blockIdx.x] = ha; ws[2 * OBS_NB + blockIdx.x] = lb; ws[3 * OBS_NB + blockIdx.x] = hb; }
}
__global__ __launch_bounds__(256) void k_qadd_final(const float* __restrict__ ws, const QaddFinal f) {
    __shared__ float sc[16];
    float la = INFINITY, ha = -INFINITY, lb = INFINITY, hb = -INFINITY;
    for (int i = threadIdx.x; i < f.nb; i += 256) {
        la = OpMinF()(la, ws[i]); ha = OpMaxF()(ha, ws[OBS_NB + i]); lb = OpMinF()(lb, ws[2 * OBS_NB + i]); hb = OpMaxF()(hb, ws[3 * OBS_NB + i]);
    }
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
// RELU: the ReLU the ResNet block applies to the sum (models/resnet.py:63) in the same pass -- relu(s) with ATen's NaN rule; backward mask s > 0 from the recomputed sum
__device__ __forceinline__ float qadd_sum(float a, float b, float sc, float zp, float qmin, float qmax) { return iao_fq(a, sc, zp, qmin, qmax) + iao_fq(b, sc, zp, qmin, qmax); }
template <int RELU>
__global__ __launch_bounds__(256) void k_qadd_fwd(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int64_t n4,
                                                  const float* __restrict__ qp, float qmin, float qmax, float* __restrict__ mm) {
    // mm (may be null): per-block min / max of the output -> mm[block], mm[nblocks + block] for the next layers' observers (mn_iao_observe_partials)
    __shared__ float scm[16];
    float mlo = INFINITY, mhi = -INFINITY;
    const float sc = qp[0], zp = qp[1];
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n4; j += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(a)[j], w = reinterpret_cast<const float4*>(b)[j];
        float4 o = make_float4(qadd_sum(v.x, w.x, sc, zp, qmin, qmax), qadd_sum(v.y, w.y, sc, zp, qmin, qmax), qadd_sum(v.z, w.z, sc, zp, qmin, qmax), qadd_sum(v.w, w.w, sc, zp, qmin, qmax));
        if (RELU) o = make_float4(qa_relu(o.x), qa_relu(o.y), qa_relu(o.z), qa_relu(o.w));
        reinterpret_cast<float4*>(y)[j] = o;
        if (mm) {
            mlo = OpMinF()(OpMinF()(mlo, o.x), OpMinF()(OpMinF()(o.y, o.z), o.w));
            mhi = OpMaxF()(OpMaxF()(mhi, o.x), OpMaxF()(OpMaxF()(o.y, o.z), o.w));
        }
    }
    if (mm) {
        mlo = block_reduce(mlo, OpMinF(), INFINITY, scm);
        mhi = block_reduce(mhi, OpMaxF(), -INFINITY, scm);
        if (threadIdx.x == 0) { mm[blockIdx.x] = mlo; mm[gridDim.x + blockIdx.x] = mhi; }
    }
}
template <int RELU>
__global__ __launch_bounds__(256) void k_qadd_bwd(const float* __restrict__ g, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ da,
                                                  float* __restrict__ db, int64_t n4, const float* __restrict__ qp, float qmin, float qmax) {
    const float sc = qp[0], zp = qp[1], lo = qp[2], hi = qp[3];
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n4; j += (int64_t)gridDim.x * 256) {
        float4 gv = reinterpret_cast<const float4*>(g)[j];
        const float4 v = reinterpret_cast<const float4*>(a)[j], w = reinterpret_cast<const float4*>(b)[j];
        if (RELU) {          // threshold_backward: the gradient passes where the (recomputed) sum is > 0
            gv.x = qadd_sum(v.x, w.x, sc, zp, qmin, qmax) > 0.f ? gv.x : 0.f; gv.y = qadd_sum(v.y, w.y, sc, zp, qmin, qmax) > 0.f ? gv.y : 0.f;
            gv.z = qadd_sum(v.z, w.z, sc, zp, qmin, qmax) > 0.f ? gv.z : 0.f; gv.w = qadd_sum(v.w, w.w, sc, zp, qmin, qmax) > 0.f ? gv.w : 0.f;
        }
        reinterpret_cast<float4*>(da)[j] = make_float4(iao_fq_grad(gv.x, v.x, sc, zp, lo, hi, qmin, qmax), iao_fq_grad(gv.y, v.y, sc, zp, lo, hi, qmin, qmax),
                                                       iao_fq_grad(gv.z, v.z, sc, zp, lo, hi, qmin, qmax), iao_fq_grad(gv.w, v.w, sc, zp, lo, hi, qmin, qmax));
        reinterpret_cast<float4*>(db)[j] = make_float4(iao_fq_grad(gv.x, w.x, sc, zp, lo, hi, qmin, qmax), iao_fq_grad(gv.y, w.y, sc, zp, lo, hi, qmin, qmax),
                                                       iao_fq_grad(gv.z, w.z, sc, zp, lo, hi, qmin, qmax), iao_fq_grad(gv.w, w.w, sc, zp, lo, hi, qmin, qmax));
    }
}
extern "C" int64_t mn_iao_qadd_ws_floats(void) { return 4 * OBS_NB; }
extern "C" int mn_iao_qadd_observe(const float* res, const float* shortcut, int64_t n, int obs_kind, int first_res, int first_shortcut, double momentum,
                                   float* min_res, float* max_res, float* min_shortcut, float* max_shortcut, float* min_out, float* max_out, int bits, int q_type,
                                   int update, float* scale, float* zero_point, float* qp, float* ws, mn_stream_t stream) {
    if (n <= 0 || n % 4 || !res || !shortcut || !aligned16(res) || !aligned16(shortcut) || !min_res || !max_res || !min_shortcut || !max_shortcut || !min_out || !max_out ||
        !scale || !zero_point || !qp || !ws || bits < 2 || bits > 24 || (obs_kind != 0 && obs_kind != 1) || (q_type != 0 && q_type != 1))
        MN_FAIL(MN_EINVAL, "mn_iao_qadd_observe: bad arguments (n must be a multiple of 4, tensors 16-byte aligned)");
    hipStream_t s = (hipStream_t)stream;
    const int nb = mn_grid_for(n / 4, 256 * 4, OBS_NB);
    hipLaunchKernelGGL(k_qadd_partial, dim3(nb), dim3(256), 0, s, res, shortcut, n / 4, ws);
    const IaoRange r = iao_range(bits, q_type, 1);
    QaddFinal f;
    f.nb = nb; f.obs_kind = obs_kind; f.first_a = first_res; f.first_b = first_shortcut; f.q_type = q_type; f.update = update; f.momentum = momentum;
    f.quant_range = (q_type == 0) ? (float)((double)(r.qmax - r.qmin) / 2.0) : (float)(r.qmax - r.qmin);
    f.min_a = min_res; f.max_a = max_res; f.min_b = min_shortcut; f.max_b = max_shortcut; f.min_o = min_out; f.max_o = max_out;
    f.scale = scale; f.zero_point = zero_point; f.qp = qp;
    hipLaunchKernelGGL(k_qadd_final, dim3(1), dim3(256), 0, s, (const float*)ws, f);
    MN_CHECK_LAUNCH("mn_iao_qadd_observe");
    return MN_OK;
}
// the same bookkeeping from the (min, max) partials the PRODUCERS of res / shortcut left (mm_x[0 .. count_x) minima, mm_x[count_x .. 2 count_x) maxima: mn_bn2d_fwd_mm,
// mn_bnrelu_fwd_mm, mn_iao_qadd_fwd_mm): one launch, neither tensor is read.  min / max are exact and order-free: bit-identical to mn_iao_qadd_observe.
__global__ __launch_bounds__(256) void k_qadd_final_p(const float* __restrict__ ma, int ca, const float* __restrict__ mb, int cb, const QaddFinal f) {
    __shared__ float sc[16];
    mn_qadd_final_tail(ma, ca, mb, cb, f, sc);
}
extern "C" int mn_iao_qadd_observe_partials(const float* mm_res, int64_t count_res, const float* mm_shortcut, int64_t count_shortcut, int obs_kind, int first_res,
                                            int first_shortcut, double momentum, float* min_res, float* max_res, float* min_shortcut, float* max_shortcut,
                                            float* min_out, float* max_out, int bits, int q_type, int update, float* scale, float* zero_point, float* qp,
                                            mn_stream_t stream) {
    if (!mm_res || !mm_shortcut || count_res <= 0 || count_shortcut <= 0 || count_res > (1 << 24) || count_shortcut > (1 << 24) || !min_res || !max_res ||
        !min_shortcut || !max_shortcut || !min_out || !max_out || !scale || !zero_point || !qp || bits < 2 || bits > 24 || (obs_kind != 0 && obs_kind != 1) ||
        (q_type != 0 && q_type != 1))
        MN_FAIL(MN_EINVAL, "mn_iao_qadd_observe_partials: bad arguments");
    const IaoRange r = iao_range(bits, q_type, 1);
    QaddFinal f;
    f.nb = 0; f.obs_kind = obs_kind; f.first_a = first_res; f.first_b = first_shortcut; f.q_type = q_type; f.update = update; f.momentum = momentum;
    f.quant_range = (q_type == 0) ? (float)((double)(r.qmax - r.qmin) / 2.0) : (float)(r.qmax - r.qmin);
    f.min_a = min_res; f.max_a = max_res; f.min_b = min_shortcut; f.max_b = max_shortcut; f.min_o = min_out; f.max_o = max_out;
    f.scale = scale; f.zero_point = zero_point; f.qp = qp;
    hipLaunchKernelGGL(k_qadd_final_p, dim3(1), dim3(256), 0, (hipStream_t)stream, mm_res, (int)count_res, mm_shortcut, (int)count_shortcut, f);
    MN_CHECK_LAUNCH("mn_iao_qadd_observe_partials");
    return MN_OK;
}
static int qadd_fwd_grid(int64_t n) { return mn_grid_for(n / 4, 256, 4096); }
extern "C" int64_t mn_iao_qadd_mm_count(int64_t n) { return (n > 0 && n % 4 == 0) ? qadd_fwd_grid(n) : 0; }
extern "C" int mn_iao_qadd_fwd_mm(const float* res, const float* shortcut, float* out, int64_t n, const float* qp, int bits, int q_type, int relu, float* mm, mn_stream_t stream);
extern "C" int mn_iao_qadd_fwd(const float* res, const float* shortcut, float* out, int64_t n, const float* qp, int bits, int q_type, int relu, mn_stream_t stream) {
    return mn_iao_qadd_fwd_mm(res, shortcut, out, n, qp, bits, q_type, relu, nullptr, stream);
}
// the same + per-block min / max of `out` (mm: 2 * mn_iao_qadd_mm_count(n) floats) for the observers of the layers that read it
extern "C" int mn_iao_qadd_fwd_mm(const float* res, const float* shortcut, float* out, int64_t n, const float* qp, int bits, int q_type, int relu, float* mm, mn_stream_t stream) {
    if (n <= 0 || n % 4 || !res || !shortcut || !out || !qp || !aligned16(res) || !aligned16(shortcut) || !aligned16(out) || bits < 2 || bits > 24)
        MN_FAIL(MN_EINVAL, "mn_iao_qadd_fwd: bad arguments");
    const IaoRange r = iao_range(bits, q_type, 1);
    mn_prof_bytes(12.0 * (double)n);
    mn_set_last_kernel("k_qadd_fwd<%d>", relu ? 1 : 0);
    mn_prof_begin((hipStream_t)stream);
    if (relu) hipLaunchKernelGGL(k_qadd_fwd<1>, dim3(qadd_fwd_grid(n)), dim3(256), 0, (hipStream_t)stream, res, shortcut, out, n / 4, qp, r.qmin, r.qmax, mm);
    else hipLaunchKernelGGL(k_qadd_fwd<0>, dim3(qadd_fwd_grid(n)), dim3(256), 0, (hipStream_t)stream, res, shortcut, out, n / 4, qp, r.qmin, r.qmax, mm);
    mn_prof_end((hipStream_t)stream);
    MN_CHECK_LAUNCH("mn_iao_qadd_fwd");
    return MN_OK;
}
extern "C" int mn_iao_qadd_bwd(const float* g, const float* res, const float* shortcut, float* dres, float* dshortcut, int64_t n, const float* qp, int bits, int q_type,
                               int relu, mn_stream_t stream) {
    if (n <= 0 || n % 4 || !g || !res || !shortcut || !dres || !dshortcut || !qp || !aligned16(g) || !aligned16(res) || !aligned16(shortcut) || !aligned16(dres) ||
        !aligned16(dshortcut) || bits < 2 || bits > 24)
        MN_FAIL(MN_EINVAL, "mn_iao_qadd_bwd: bad arguments");
    const IaoRange r = iao_range(bits, q_type, 1);
    mn_prof_bytes(20.0 * (double)n);
    mn_set_last_kernel("k_qadd_bwd<%d>", relu ? 1 : 0);
    mn_prof_begin((hipStream_t)stream);
    if (relu) hipLaunchKernelGGL(k_qadd_bwd<1>, dim3(mn_grid_for(n / 4, 256, 4096)), dim3(256), 0, (hipStream_t)stream, g, res, shortcut, dres, dshortcut, n / 4, qp, r.qmin, r.qmax);
    else hipLaunchKernelGGL(k_qadd_bwd<0>, dim3(mn_grid_for(n / 4, 256, 4096)), dim3(256), 0, (hipStream_t)stream, g, res, shortcut, dres, dshortcut, n / 4, qp, r.qmin, r.qmax);
    mn_prof_end((hipStream_t)stream);
    MN_CHECK_LAUNCH("mn_iao_qadd_bwd");
    return MN_OK;
}

// ------------------------------------------------------------------------------------------------
// BN-fuse statistics (853-855): per-channel mean / unbiased variance over (N, HW) of o[N][C][HW].
// Pass 1: grid (C, S) blocks, each sums a slice of images of one channel in fp64 (sum, sum of squares about a
// pivot = first element of the channel, which keeps the one-pass variance well conditioned).
// Pass 2: one block per channel combines the S partials.
static const int BN_SPLIT = 32;
extern "C" int64_t mn_bn_stats_ws_floats(int64_t, int64_t C, int64_t) { return C * BN_SPLIT * 4; }  // 2 doubles per partial

__global__ __launch_bounds__(256) void k_bn_stats_partial(const float* __restrict__ o, int N, int C, int HW, double* __restrict__ part) {
    __shared__ double scd[16];
    const int c = blockIdx.x, sp = blockIdx.y, S = gridDim.y;
    const float pivot = o[(int64_t)c * HW];
    double s1 = 0.0, s2 = 0.0;
    const int vec = (HW % 4 == 0) && aligned16(o);
    for (int n = sp; n < N; n += S) {
        const float* p = o + ((int64_t)n * C + c) * HW;
        if (vec) {
            for (int j = threadIdx.x; j < HW / 4; j += blockDim.x) {
                float4 v = reinterpret_cast<const float4*>(p)[j];
                double a = (double)v.x - pivot, b = (double)v.y - pivot, cc = (double)v.z - pivot, d = (double)v.w - pivot;
                s1 += (a + b) + (cc + d);
                s2 += (a * a + b * b) + (cc * cc + d * d);
            }
        } else {
            for (int j = threadIdx.x; j < HW; j += blockDim.x) {
                double a = (double)p[j] - pivot;
                s1 += a;
                s2 += a * a;
            }
        }
    }
    s1 = block_reduce(s1, OpAddD(), 0.0, scd);
    s2 = block_reduce(s2, OpAddD(), 0.0, scd);
    if (threadIdx.x == 0) { part[((int64_t)c * S + sp) * 2] = s1; part[((int64_t)c * S + sp) * 2 + 1] = s2; }
}
__global__ void k_bn_stats_final(const float* __restrict__ o, const double* __restrict__ part, int N, int C, int HW, int S,
                                 float* __restrict__ stats) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double sv[2] = {0.0, 0.0};
    mn_row_sums<2>(part + (int64_t)c * S * 2, S, sv);
    const double s1 = sv[0], s2 = sv[1];
    const double n = (double)N * (double)HW;
    const double pivot = (double)o[(int64_t)c * HW];
    const double m = s1 / n;
    stats[c] = (float)(pivot + m);
    stats[C + c] = (float)((s2 - s1 * m) / (n - 1.0));    // unbiased; n == 1 -> NaN like torch.var
}
extern "C" int mn_bn_stats_fwd(const float* o, int64_t N, int64_t C, int64_t HW, float* stats, float* ws, mn_stream_t stream) {
    if (N <= 0 || C <= 0 || HW <= 0 || !o || !stats || !ws || (((uintptr_t)ws) & 7)) MN_FAIL(MN_EINVAL, "mn_bn_stats_fwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    int S = (int)(N < BN_SPLIT ? N : BN_SPLIT);
    hipLaunchKernelGGL(k_bn_stats_partial, dim3((unsigned)C, (unsigned)S), dim3(256), 0, s, o, (int)N, (int)C, (int)HW, (double*)ws);
    hipLaunchKernelGGL(k_bn_stats_final, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, s, o, (const double*)ws, (int)N, (int)C, (int)HW, S, stats);
    MN_CHECK_LAUNCH("mn_bn_stats_fwd");
    return MN_OK;
}
__global__ __launch_bounds__(256) void k_bn_stats_bwd(const float* __restrict__ o, const float* __restrict__ stats, const float* __restrict__ dmean,
                                                      const float* __restrict__ dvar, float* __restrict__ d_o, int N, int C, int HW) {
    // grid: (C, N-slices). d_o = dmean/n + dvar * 2 (o - mean) / (n - 1)   (autograd of mean / var(unbiased))
    const int c = blockIdx.x;
    const float n = (float)N * (float)HW;
    const float a = dmean[c] / n, mean = stats[c];
    const float k = dvar[c] * 2.f / (n - 1.f);
    const int vec = (HW % 4 == 0) && aligned16(o) && aligned16(d_o);
    for (int img = blockIdx.y; img < N; img += gridDim.y) {
        const int64_t off = ((int64_t)img * C + c) * HW;
        if (vec) {
            for (int j = threadIdx.x; j < HW / 4; j += blockDim.x) {
                float4 v = reinterpret_cast<const float4*>(o + off)[j], r;
                r.x = a + k * (v.x - mean); r.y = a + k * (v.y - mean); r.z = a + k * (v.z - mean); r.w = a + k * (v.w - mean);
                reinterpret_cast<float4*>(d_o + off)[j] = r;
            }
        } else {
            for (int j = threadIdx.x; j < HW; j += blockDim.x) d_o[off + j] = a + k * (o[off + j] - mean);
        }
    }
}
extern "C" int mn_bn_stats_bwd(const float* o, const float* stats, const float* dmean, const float* dvar, float* d_o,
                               int64_t N, int64_t C, int64_t HW, mn_stream_t stream) {
    if (N <= 0 || C <= 0 || HW <= 0 || !o || !stats || !dmean || !dvar || !d_o) MN_FAIL(MN_EINVAL, "mn_bn_stats_bwd: bad arguments");
    int S = (int)(N < 64 ? N : 64);
    hipLaunchKernelGGL(k_bn_stats_bwd, dim3((unsigned)C, (unsigned)S), dim3(256), 0, (hipStream_t)stream, o, stats, dmean, dvar, d_o, (int)N, (int)C, (int)HW);
    MN_CHECK_LAUNCH("mn_bn_stats_bwd");
    return MN_OK;
}
