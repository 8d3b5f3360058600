// TEST TOOL ONLY -- a minimal single-threaded SIMT emulator so the *unmodified* HIP kernel
// sources under micronet_amd/csrc/ can be compiled with g++ and executed on a CPU:
//     g++ -I tests/emu/include -x c++ micronet_amd/csrc/*.hip -> tests/emu/libmicronet_emu.so
// It exists because the build container has no GPU: the unit tests drive the same extern "C"
// ABI against this build to check index math, tiling, fragment layouts and reductions before
// any GPU time is spent.  It is never loaded by micronet_amd (the product refuses to run
// without the gfx950 library).
//
// Model: one OS thread; every GPU thread of the running block is a ucontext fiber; a fiber runs
// until it reaches a block barrier or a wave collective, where it yields.  Blocks run one after
// another.  Wave size 64.  MFMA v_mfma_f32_16x16x4_f32 is emulated with the lane->element maps
// of the CDNA4 ISA: A[i=lane&15][k=lane>>4], B[k=lane>>4][j=lane&15],
// D[row=(lane>>4)*4+reg][col=lane&15], as an in-order fp32 fma chain over k.
#pragma once
#define MN_EMULATION 1
#include <ucontext.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __restrict__ __restrict

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu { unsigned x, y, z; };
struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
struct int4 { int x, y, z, w; };
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }

typedef int hipError_t;
typedef void* hipStream_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
#define hipMemcpyDeviceToDevice 3

namespace emu {
enum State { RUNNABLE, WAIT_BLOCK, WAIT_WAVE, DONE };
struct Fiber {
    ucontext_t ctx;
    State st;
    uint3_emu tid;
    int wave;
    char* stack;
};
struct Ctx {
    std::vector<Fiber> fibers;
    ucontext_t sched;
    Fiber* cur = nullptr;
    std::function<void()> body;
    char* dyn_smem = nullptr;
    // per-wave exchange buffers
    double xd[64 * 8];  // up to 8 waves x 64 lanes (reallocated as needed)
    std::vector<double> wbuf;
    std::vector<float> wa, wb;
    std::vector<float> wa8, wb8;   // bf16 MFMA operands, 8 per lane
    std::vector<int> wi16a, wi16b; // int8 MFMA operands, 16 per lane
    std::vector<int> wflag;
    std::vector<const void*> wptr;  // ds_read_b64_tr_b16: the lanes' addresses
};
inline Ctx& C() { static Ctx c; return c; }
}  // namespace emu

extern thread_local uint3_emu threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;
#ifdef MN_EMU_MAIN
thread_local uint3_emu threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;
#endif

namespace emu {
inline void yield(State s) {
    Ctx& c = C();
    Fiber* f = c.cur;
    f->st = s;
    swapcontext(&f->ctx, &c.sched);
}
inline void trampoline() {
    Ctx& c = C();
    c.body();
    c.cur->st = DONE;
    swapcontext(&c.cur->ctx, &c.sched);
}
inline void run_block(unsigned nthreads, dim3 bdim) {
    Ctx& c = C();
    const size_t STK = 512 * 1024;
    if (c.fibers.size() < nthreads) {
        size_t old = c.fibers.size();
        c.fibers.resize(nthreads);
        for (size_t i = old; i < nthreads; ++i) c.fibers[i].stack = (char*)malloc(STK);
    }
    unsigned nw = (nthreads + 63) / 64;
    c.wbuf.assign(nw * 64, 0.0);
    c.wa.assign(nw * 64, 0.f);
    c.wb.assign(nw * 64, 0.f);
    c.wa8.assign(nw * 64 * 8, 0.f);
    c.wb8.assign(nw * 64 * 8, 0.f);
    c.wi16a.assign(nw * 64 * 16, 0);
    c.wi16b.assign(nw * 64 * 16, 0);
    c.wflag.assign(nw * 64, 0);
    for (unsigned i = 0; i < nthreads; ++i) {
        Fiber& f = c.fibers[i];
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack;
        f.ctx.uc_stack.ss_size = STK;
        f.ctx.uc_link = &c.sched;
        makecontext(&f.ctx, (void (*)())trampoline, 0);
        f.st = RUNNABLE;
        f.tid.x = i % bdim.x;
        f.tid.y = (i / bdim.x) % bdim.y;
        f.tid.z = i / (bdim.x * bdim.y);
        f.wave = i / 64;
    }
    unsigned alive = nthreads;
    while (alive) {
        bool progressed = false;
        for (unsigned i = 0; i < nthreads; ++i) {
            Fiber& f = c.fibers[i];
            if (f.st != RUNNABLE) continue;
            c.cur = &f;
            threadIdx = f.tid;
            swapcontext(&c.sched, &f.ctx);
            progressed = true;
            if (f.st == DONE) --alive;
        }
        // release barriers
        bool released = false;
        unsigned nblock = 0, nlive = 0;
        for (unsigned i = 0; i < nthreads; ++i) {
            if (c.fibers[i].st != DONE) ++nlive;
            if (c.fibers[i].st == WAIT_BLOCK) ++nblock;
        }
        if (nlive && nblock == nlive) {
            for (unsigned i = 0; i < nthreads; ++i)
                if (c.fibers[i].st == WAIT_BLOCK) c.fibers[i].st = RUNNABLE;
            released = true;
        }
        for (unsigned w = 0; w < nw; ++w) {
            unsigned lo = w * 64, hi = std::min(nthreads, lo + 64), nl = 0, nwv = 0;
            for (unsigned i = lo; i < hi; ++i) {
                if (c.fibers[i].st != DONE) ++nl;
                if (c.fibers[i].st == WAIT_WAVE) ++nwv;
            }
            if (nl && nwv == nl) {
                for (unsigned i = lo; i < hi; ++i)
                    if (c.fibers[i].st == WAIT_WAVE) c.fibers[i].st = RUNNABLE;
                released = true;
            }
        }
        if (!progressed && !released && alive) {
            fprintf(stderr, "emu: deadlock (divergent barrier?)\n");
            abort();
        }
    }
}
inline int lane() { return (int)((threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z)) & 63); }
inline int flat_tid() { return (int)(threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z)); }
template <typename T>
inline T wave_exchange(T v, int src_lane) {
    Ctx& c = C();
    int base = (flat_tid() / 64) * 64;
    double d;
    static_assert(sizeof(T) <= 8, "");
    d = 0;
    memcpy(&d, &v, sizeof(T));
    c.wbuf[base + lane()] = d;
    yield(WAIT_WAVE);
    T out = v;
    if (src_lane >= 0 && src_lane < 64) memcpy(&out, &c.wbuf[base + src_lane], sizeof(T));
    yield(WAIT_WAVE);
    return out;
}
}  // namespace emu

static inline void __syncthreads() { emu::yield(emu::WAIT_BLOCK); }
static inline void __threadfence() {}          // one OS thread: every store is visible at once
template <typename T>
static inline T __shfl_down(T v, unsigned d, int width = 64) {
    int l = emu::lane();
    int src = l + (int)d;
    if ((src / width) != (l / width)) src = l;  // out of segment: own value
    return emu::wave_exchange(v, src);
}
template <typename T>
static inline T __shfl_xor(T v, int m, int width = 64) {
    return emu::wave_exchange(v, emu::lane() ^ m);
}
template <typename T>
static inline T __shfl(T v, int src, int width = 64) {
    int l = emu::lane();
    return emu::wave_exchange(v, (l / width) * width + (src % width));
}
// one OS thread, fibers switch only at barriers / wave collectives: a plain read-modify-write is atomic here
template <typename T>
static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((uint64_t)a * b) >> 32); }

typedef float __emu_f32x4 __attribute__((vector_size(16)));
static inline __emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, __emu_f32x4 c, int, int, int) {
    emu::Ctx& cx = emu::C();
    int base = (emu::flat_tid() / 64) * 64, l = emu::lane();
    cx.wa[base + l] = a;
    cx.wb[base + l] = b;
    emu::yield(emu::WAIT_WAVE);
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(cx.wa[base + k * 16 + row], cx.wb[base + k * 16 + col], acc);
        c[r] = acc;
    }
    emu::yield(emu::WAIT_WAVE);
    return c;
}


// v_mfma_f32_16x16x32_bf16: A[i = lane&15][k = 8*(lane>>4) + e], B[k = 8*(lane>>4) + e][j = lane&15], e = 0..7 packed two per
// dword (low half first); D[row = 4*(lane>>4) + reg][col = lane&15].  Products of two bf16 are exact in fp32; the sum is
// emulated as an in-order fp32 chain over k (the hardware's internal order is unspecified; tests use tolerances).
static inline float emu_bf16_to_f32(uint32_t h) { uint32_t u = h << 16; float f; memcpy(&f, &u, 4); return f; }
typedef unsigned __emu_u32x4 __attribute__((vector_size(16)));
static inline __emu_f32x4 emu_mfma_f32_16x16x32_bf16(__emu_u32x4 a, __emu_u32x4 b, __emu_f32x4 c) {
    emu::Ctx& cx = emu::C();
    int base = (emu::flat_tid() / 64) * 64, l = emu::lane();
    for (int e = 0; e < 8; ++e) {
        cx.wa8[(base + l) * 8 + e] = emu_bf16_to_f32((a[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
        cx.wb8[(base + l) * 8 + e] = emu_bf16_to_f32((b[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
    }
    emu::yield(emu::WAIT_WAVE);
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int kg = 0; kg < 4; ++kg)
            for (int e = 0; e < 8; ++e) acc = fmaf(cx.wa8[(base + kg * 16 + row) * 8 + e], cx.wb8[(base + kg * 16 + col) * 8 + e], acc);
        c[r] = acc;
    }
    emu::yield(emu::WAIT_WAVE);
    return c;
}
// v_mfma_f32_32x32x16_bf16: A[i = lane&31][k = 8*(lane>>5) + e], B[k = 8*(lane>>5) + e][j = lane&31]; D[row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)][col = lane&31],
// reg = 0..15 (cdna_hip_programming.md "Fragment layout"; the same map as ck_tile's WarpGemmAttributeMfmaImplBf16Bf16F32M32N32K16 in /opt/rocm/include: kAMLane 32,
// kABKLane 2, kABKPerLane 8; kCMLane 2, kCNLane 32, kCM0PerLane 4, kCM1PerLane 4).  Same in-order fp32 chain over k as the 16x16x32 form.
typedef float __emu_f32x16 __attribute__((vector_size(64)));
static inline __emu_f32x16 emu_mfma_f32_32x32x16_bf16(__emu_u32x4 a, __emu_u32x4 b, __emu_f32x16 c) {
    emu::Ctx& cx = emu::C();
    int base = (emu::flat_tid() / 64) * 64, l = emu::lane();
    for (int e = 0; e < 8; ++e) {
        cx.wa8[(base + l) * 8 + e] = emu_bf16_to_f32((a[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
        cx.wb8[(base + l) * 8 + e] = emu_bf16_to_f32((b[e >> 1] >> ((e & 1) * 16)) & 0xffffu);
    }
    emu::yield(emu::WAIT_WAVE);
    int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int kb = 0; kb < 2; ++kb)
            for (int e = 0; e < 8; ++e) acc = fmaf(cx.wa8[(base + kb * 32 + row) * 8 + e], cx.wb8[(base + kb * 32 + col) * 8 + e], acc);
        c[r] = acc;
    }
    emu::yield(emu::WAIT_WAVE);
    return c;
}
// v_mfma_i32_16x16x64_i8: A[i = lane&15][k = 16*(lane>>4) + e], B[k = 16*(lane>>4) + e][j = lane&15] (signed bytes, little endian in the four dwords), D as above
typedef int __emu_i32x4 __attribute__((vector_size(16)));
static inline __emu_i32x4 emu_mfma_i32_16x16x64_i8(__emu_u32x4 a, __emu_u32x4 b, __emu_i32x4 c) {
    emu::Ctx& cx = emu::C();
    int base = (emu::flat_tid() / 64) * 64, l = emu::lane();
    for (int e = 0; e < 16; ++e) {
        cx.wi16a[(base + l) * 16 + e] = (int)(signed char)((a[e >> 2] >> ((e & 3) * 8)) & 0xffu);
        cx.wi16b[(base + l) * 16 + e] = (int)(signed char)((b[e >> 2] >> ((e & 3) * 8)) & 0xffu);
    }
    emu::yield(emu::WAIT_WAVE);
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r;
        int acc = c[r];
        for (int kg = 0; kg < 4; ++kg)
            for (int e = 0; e < 16; ++e) acc += cx.wi16a[(base + kg * 16 + row) * 16 + e] * cx.wi16b[(base + kg * 16 + col) * 16 + e];
        c[r] = acc;
    }
    emu::yield(emu::WAIT_WAVE);
    return c;
}
// ds_read_b64_tr_b16 (gfx950 LDS transpose read): every lane supplies the address of 4 consecutive 16-bit elements; within each 16-lane group the 16 x 4
// elements form a [4 rows][16 columns] block (row e = the four lanes 4e .. 4e+3, four columns each) and lane c of the group receives column c: element e
// of its result = element (c & 3) of lane 4e + (c >> 2).  (Measured on MI355X: scripts/probe/tr16_probe.hip.)
typedef unsigned __emu_u32x2 __attribute__((vector_size(8)));
static inline __emu_u32x2 emu_ds_read_tr16_b64(const void* addr) {
    emu::Ctx& cx = emu::C();
    int base = (emu::flat_tid() / 64) * 64, l = emu::lane();
    if (cx.wptr.size() < cx.wflag.size()) cx.wptr.assign(cx.wflag.size(), nullptr);
    cx.wptr[base + l] = addr;
    emu::yield(emu::WAIT_WAVE);
    const int grp = l & ~15, c = l & 15;
    uint16_t h[4];
    for (int e = 0; e < 4; ++e) {
        const uint16_t* src = (const uint16_t*)cx.wptr[base + grp + 4 * e + (c >> 2)];
        h[e] = src[c & 3];
    }
    emu::yield(emu::WAIT_WAVE);
    __emu_u32x2 r;
    r[0] = (unsigned)h[0] | ((unsigned)h[1] << 16);
    r[1] = (unsigned)h[2] | ((unsigned)h[3] << 16);
    return r;
}
// wave-wide "any lane has pred != 0"
static inline int emu_wave_any(int pred) {
    emu::Ctx& cx = emu::C();
    int base = (emu::flat_tid() / 64) * 64, l = emu::lane();
    cx.wflag[base + l] = pred ? 1 : 0;
    emu::yield(emu::WAIT_WAVE);
    int any = 0;
    unsigned nthreads = blockDim.x * blockDim.y * blockDim.z;
    for (int i = 0; i < 64 && (unsigned)(base + i) < nthreads; ++i) any |= cx.wflag[base + i];
    emu::yield(emu::WAIT_WAVE);
    return any;
}

#define HIP_DYNAMIC_SHARED(type, var) type* var = (type*)emu::C().dyn_smem;

template <typename K, typename... Args>
static inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t shmem, hipStream_t, Args... args) {
    emu::Ctx& c = emu::C();
    std::vector<char> sm(shmem + 64);
    c.dyn_smem = (char*)(((uintptr_t)sm.data() + 63) & ~(uintptr_t)63);
    gridDim = grid;
    blockDim = block;
    unsigned nthreads = block.x * block.y * block.z;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                blockIdx.x = bx;
                blockIdx.y = by;
                blockIdx.z = bz;
                c.body = [&]() { kernel(args...); };
                emu::run_block(nthreads, block);
            }
}
