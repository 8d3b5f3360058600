// Multi-tensor Adam step for gfx950: ONE launch updates every parameter tensor of the model.
//
// The reference training scripts build torch.optim.Adam with one parameter group per tensor
// (wqaq/dorefa/main.py:308-315); torch then runs ~7 small kernels per group -- 252 launches per nin_gc step, more time
// than the convolutions.  This kernel applies exactly torch's update
//     g' = g + wd*p ; m += (g' - m)*(1 - b1) ; v = v*b2 + (1 - b2)*g'*g' ; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
// to a table of tensors passed by value in the kernel arguments (pointers change every step because autograd
// re-allocates the gradients), float4 per lane, one workgroup per 2048-element chunk.
#include "common.h"

#define ADAM_CHUNK 2048
struct AdamTable {
    float* p[MN_ADAM_MAX_TENSORS];
    const float* g[MN_ADAM_MAX_TENSORS];
    float* m[MN_ADAM_MAX_TENSORS];
    float* v[MN_ADAM_MAX_TENSORS];
    int n[MN_ADAM_MAX_TENSORS];
    float lr[MN_ADAM_MAX_TENSORS], wd[MN_ADAM_MAX_TENSORS];
    int chunk0[MN_ADAM_MAX_TENSORS + 1];     // first chunk of each tensor
    int count;
    float beta1, beta2, eps, bc1, bc2_sqrt;
    const int* step_dev;     // non-null: the step count lives in device memory (HIP-graph replay), bias corrections computed here
    const float* hyper_dev;  // non-null: {lr, weight_decay} of tensor i at hyper_dev[2 * (hyper_base + slot_src[i])] -- a replayed graph then
    int hyper_base;          // follows the host's param_group['lr'] edits (the reference's adjust_learning_rate, wbwtab/main.py:62-66)
    int slot_src[MN_ADAM_MAX_TENSORS];
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float lr_bc1, float wd, const AdamTable& t, float bc2_sqrt) {
    if (wd != 0.f) g = g + wd * p;
    m = m + (g - m) * (1.f - t.beta1);
    v = v * t.beta2 + (1.f - t.beta2) * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + t.eps;
    p = p - lr_bc1 * (m / denom);
}

__global__ __launch_bounds__(256) void k_adam(const AdamTable t) {
    float bc1 = t.bc1, bc2_sqrt = t.bc2_sqrt;
    if (t.step_dev) {        // uniform: same expressions as the host path of mn_adam_step (the table itself stays read-only:
        const double st = (double)*t.step_dev;          // writing to a by-value kernel argument would spill it to scratch)
        bc1 = (float)(1.0 - pow((double)t.beta1, st));
        bc2_sqrt = (float)sqrt(1.0 - pow((double)t.beta2, st));
    }
    int ti = 0;
    const int b = blockIdx.x;
    while (ti + 1 < t.count && t.chunk0[ti + 1] <= b) ++ti;      // <= 32 steps, uniform
    const int off = (b - t.chunk0[ti]) * ADAM_CHUNK;
    const int n = t.n[ti];
    float* __restrict__ P = t.p[ti];
    const float* __restrict__ G = t.g[ti];
    float* __restrict__ M = t.m[ti];
    float* __restrict__ V = t.v[ti];
    float lr = t.lr[ti], wd = t.wd[ti];
    if (t.hyper_dev) {
        const float* hp = t.hyper_dev + 2 * (t.hyper_base + t.slot_src[ti]);
        lr = hp[0]; wd = hp[1];
    }
    const float lr_bc1 = lr / bc1;
    const bool vec = aligned16(P) && aligned16(G) && aligned16(M) && aligned16(V);
    for (int i = off + threadIdx.x * 4; i < off + ADAM_CHUNK && i < n; i += 256 * 4) {
        if (vec && i + 3 < n) {
            float4 p4 = *reinterpret_cast<float4*>(P + i), m4 = *reinterpret_cast<float4*>(M + i), v4 = *reinterpret_cast<float4*>(V + i);
            const float4 g4 = *reinterpret_cast<const float4*>(G + i);
            adam_one(p4.x, g4.x, m4.x, v4.x, lr_bc1, wd, t, bc2_sqrt); adam_one(p4.y, g4.y, m4.y, v4.y, lr_bc1, wd, t, bc2_sqrt);
            adam_one(p4.z, g4.z, m4.z, v4.z, lr_bc1, wd, t, bc2_sqrt); adam_one(p4.w, g4.w, m4.w, v4.w, lr_bc1, wd, t, bc2_sqrt);
            *reinterpret_cast<float4*>(P + i) = p4; *reinterpret_cast<float4*>(M + i) = m4; *reinterpret_cast<float4*>(V + i) = v4;
        } else {
            for (int k = i; k < i + 4 && k < n; ++k) {
                float p = P[k], m = M[k], v = V[k];
                adam_one(p, G[k], m, v, lr_bc1, wd, t, bc2_sqrt);
                P[k] = p; M[k] = m; V[k] = v;
            }
        }
    }
}

static int adam_impl(const mn_adam_tensor* tensors, int count, int step, const int32_t* step_dev, const float* hyper_dev, float beta1, float beta2, float eps,
                     mn_stream_t stream) {
    if (count < 0 || (count > 0 && !tensors) || (!step_dev && step < 1)) MN_FAIL(MN_EINVAL, "mn_adam_step: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const double bc1 = step_dev ? 1.0 : 1.0 - pow((double)beta1, (double)step), bc2 = step_dev ? 1.0 : 1.0 - pow((double)beta2, (double)step);
    for (int base = 0; base < count; base += MN_ADAM_MAX_TENSORS) {
        AdamTable t;
        const int cnt = count - base < MN_ADAM_MAX_TENSORS ? count - base : MN_ADAM_MAX_TENSORS;
        int chunks = 0, used = 0;
        for (int i = 0; i < cnt; ++i) {
            const mn_adam_tensor& a = tensors[base + i];
            if (a.n == 0) continue;
            if (!a.p || !a.g || !a.m || !a.v || a.n < 0 || a.n > 0x7fffffff - ADAM_CHUNK) MN_FAIL(MN_EINVAL, "mn_adam_step: tensor %d invalid", base + i);
            t.p[used] = a.p; t.g[used] = a.g; t.m[used] = a.m; t.v[used] = a.v; t.n[used] = (int)a.n; t.lr[used] = a.lr; t.wd[used] = a.weight_decay;
            t.chunk0[used] = chunks;
            t.slot_src[used] = i;
            chunks += (int)((a.n + ADAM_CHUNK - 1) / ADAM_CHUNK);
            ++used;
        }
        if (!used) continue;
        t.chunk0[used] = chunks;
        t.count = used; t.beta1 = beta1; t.beta2 = beta2; t.eps = eps; t.bc1 = (float)bc1; t.bc2_sqrt = (float)sqrt(bc2);
        t.step_dev = (const int*)step_dev;
        t.hyper_dev = hyper_dev; t.hyper_base = base;
        hipLaunchKernelGGL(k_adam, dim3(chunks), dim3(256), 0, s, t);
    }
    MN_CHECK_LAUNCH("mn_adam_step");
    return MN_OK;
}
extern "C" int mn_adam_step(const mn_adam_tensor* tensors, int count, int step, float beta1, float beta2, float eps, mn_stream_t stream) {
    return adam_impl(tensors, count, step, nullptr, nullptr, beta1, beta2, eps, stream);
}
extern "C" int mn_adam_step_dev(const mn_adam_tensor* tensors, int count, const int32_t* step_dev, const float* hyper_dev, float beta1, float beta2, float eps,
                                mn_stream_t stream) {
    if (!step_dev) MN_FAIL(MN_EINVAL, "mn_adam_step_dev: null step counter");
    return adam_impl(tensors, count, 0, step_dev, hyper_dev, beta1, beta2, eps, stream);
}
