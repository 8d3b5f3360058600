// ONE backward kernel per pointwise quantised block (round 6): backward-data AND backward-weight of a 1 x 1 grouped convolution on activation codes, with the
// block's own BatchNorm backward formed while the operands stream in:
//   BNH 1 / 2   wbwtab: the gradient is the BatchNorm+sign backward (2: behind a 2 x 2 max-pool) of (da, h) -- wbwtab/quantize.py:11-36 (BinaryActivation),
//               :181-195 (QuantConv2d) + autograd's conv backward.  k_pwd<4, 4, BNH> (qgemm_kernels.hip) and k_pws_wgrad_s<4, BNH, 0, 2> (qgemm_sign.hip) each rebuilt
//               the same dy tile from the same (da, h): 15 B per element between them.  Here (da, h) cross HBM once: 10 B per element.
//   BNH 3       DoReFa: the gradient is the BatchNorm + ReLU + next-quantizer backward of (dq, stash) -- wqaq/dorefa/quantize.py:36-46, 107-122 -- what k_qa_apply
//               (qact_kernels.hip) wrote as fp32 dy for k_pwd<4, 4, 0> and k_pws_wgrad_s<4, 0, 1, 2> to read back: 23 B per element between the three, 11 here.
//   BNH 0       a plain fp32 gradient dy (the pooled DoReFa blocks, whose dy k_qa_apply<.., 1> still writes): 9 B per element instead of 13.
//   XENC 0 / 1  the input codes are sign bytes (+-1) / k-bit activation codes j (dW = s sum dy j: the reduction multiplies by the quantizer's scale s).
//
// Geometry: groups of 128 -> 128 channels (every pointwise layer of nin_gc), HW a multiple of 32.  A block owns one group and a contiguous range of 32-pixel steps;
// 768 threads = 12 waves, three per SIMD, one of each role:
//   waves 0-3   PRODUCERS   global loads (NS steps in flight in registers), the BatchNorm fold (already times the weight scale of the row), exact three-term bf16
//                           split, LDS image: three bf16 planes [128 rows o][32 pixels] (64-byte rows, 16-byte slots XOR-swizzled) + the input codes;
//   waves 4-7   dW          2 x 2 grid of 64 x 64 tiles of dW[o][c] += dy'[o][p] x[c][p]: v_mfma_f32_16x16x32_bf16, A = b128 rows of the planes (K = pixel);
//   waves 8-11  dx          wave w owns input channels 32 w .. 32 w + 31: dx[c][p] = sum_o W[o][c] dy'[o][p] on v_mfma_f32_32x32x16_bf16 with the weight codes held in
//                           registers (A) and the SAME planes read through the LDS transpose read ds_read_b64_tr_b16 (B: K = plane row o, N = pixel) -- no second,
//                           transposed image; every accumulator register is one 128-byte row segment of dx.
// dy' = alpha[o] dy is what backward-data contracts with the integer codes (k_pwd does the same); backward-weight wants dy, so the fixed-order fp64 reduction of the
// partial tiles divides row o by alpha[o] (a row with alpha = 0 has all-zero codes: it is staged unscaled and divided by 1).
//   UP 1        (round 6) the block IN FRONT of this one is a BatchNorm+sign block on a pointwise conv as well: this kernel's dx IS that block's d a, so the dx waves
//               also form the per-channel sums of its BatchNorm backward -- sum dz and sum dz h with dz = dx [hlo <= h <= hhi] from the upstream block's byte stash,
//               staged through LDS by the producers like the input codes (1 B per element) -- and leave them as the partials k_bns_final_bwd reads: the upstream
//               block's own pass over (d a, h) (k_bnh_partial: 5 B per element) is not launched (mn_conv2d_bwd_bnh_up / mn_bnh_bwd_sums_final).
//   UP 3        UP 1 behind a 3x3 / padding-1 block (models/nin_gc.py: layers 4 | 5 and 7 | 8): the stash's offset nnz depends on the pixel's border class (chan rows
//               8..16, common.h: StashNnz), so the pass test runs on acc = 2 h - nnz(pixel) itself -- k_bnh_partial<0>'s arithmetic (mn_conv2d_bwd_bnh_up9).
//   UP 2        the same for a k-bit (DoReFa) block in front: this dx is its d q; dz = clip-STE(dx) [ReLU / clamp masks of the upstream 16-bit stash, k_qa_partial's
//               arithmetic] -- sum dz, sum dz zhat; k_qa_partial (6 B per element) is not launched (mn_conv2d_bwd_qa_up / mn_conv2d_bwd_codes_up / mn_qa_bwd_sums_final).
// One barrier per step: barrier k publishes step k (buffer k & 1); the producers refill that buffer with step k + 2 only behind barrier k + 1, which both consumer
// groups reach after their reads of step k.  What the timeline (PWB_TRACE) and the ablations of round 6 found on the way is in profiles/README.md: the SLP
// vectoriser drains the producers' software pipeline (this file is built with -fno-slp-vectorize), a memory instruction whose address / data register is rewritten
// for the next one serialises on the memory pipeline's operand read, LDS reads issued just in time expose one round trip per MFMA group.
#include "qgemm.h"

#include "qgemm_dev.h"

#include <stdlib.h>

#define PWB_PLANE (128 * 64)
#define PWB_CODES (128 * 48)
#define PWB_BUF (3 * PWB_PLANE + PWB_CODES)
#define PWB_BUF_UP (PWB_BUF + PWB_CODES)                             // + the upstream block's stash bytes, laid out like the input codes
#define PWB_UP2_ROW 80                                               // UP 2: bytes per row of the staged 16-bit upstream stash [128 c][32 px]
#define PWB_BUF_UP2 (PWB_BUF + 128 * PWB_UP2_ROW)
#define PWB_FROW 12                                                  // floats per row of the fold table
#define PWB_ERS 68                                                   // floats per staged row of a wave's 64 x 64 partial tile
#define PWB_LDS_STAGE (2 * PWB_BUF + 128 * PWB_FROW * 4)
#define PWB_LDS_EPI (4 * 64 * PWB_ERS * 4)
#define PWB_LDS (PWB_LDS_STAGE > PWB_LDS_EPI ? PWB_LDS_STAGE : PWB_LDS_EPI)
#define PWB_UTS 36                                                   // floats per row of a dx wave's transposition tile [32 c][32 px] (16-byte aligned rows)
#define PWB_LDS_UP (2 * PWB_BUF_UP + 128 * PWB_FROW * 4 + 4 * 32 * PWB_UTS * 4)
#define PWB_LDS_UP2 (2 * PWB_BUF_UP2 + 128 * PWB_FROW * 4 + 4 * 32 * PWB_UTS * 4)

struct PwbParams {
    const float* gy;            // BNH 1: da [N][O][HW]; 2: the pooled gradient [N][O][H/2][W/2]; 3: dq [N][O][HW]; 0: dy itself
    const void* h;              // BNH 1 / 2: the one-byte conv stash [N][O][HW]; 3: the 16-bit (WIDE: 32-bit) stash of acc
    const float* chan;          // BNH 1 / 2: [8][O] (qgemm_sign.hip); 3: [9][O] (qact_kernels.hip)
    const float* sums;          // [2][O]
    const char* own;            // BNH 2: the block's own sign output [N][O][H][W]
    const char* x;              // [N][C][HW] input codes (physical channel order: in_map)
    const uint16_t* wc;         // [G][128 c][128 o] transposed weight codes (bf16)
    const float* kscale;        // [G][128] weight scale alpha of output channel o
    float* dx;                  // [N][C][HW]
    float* part;                // [Z][G][128][128]
    float* dbpart;              // [Z][G][128]
    int N, HW, C, O, G, Z, nsteps, st_per_z, want_db, training, W, quant;
    float n_f, qs, qinv;        // BNH 3: scale of the quantizer behind the block, RN(1 / qs) (0: IEEE division)
    FastDiv fd_hw, fd_w;
    ChanMap in_map;
    const unsigned char* up_h;  // UP 1: the upstream block's byte stash [N][C][HW] (the layout of x); UP 2: its 16-bit stash
    const float* up_chan;       // UP 1: its [8][C] channel constants (qgemm_sign.hip: T, flip, L, U, A, B, gi, nnz); UP 2: [9][C] (qact_kernels.hip)
    double* up_part;            // UP: [C][Z][2] partial sums {sum dz, sum dz zhat} of this launch's blocks (k_bns_final_bwd's layout, S = Z)
    float up_qs, up_qinv;       // UP 2: scale of the quantizer behind the upstream block, RN(1 / qs) (0: IEEE division)
    int up_quant;               // UP 2: this dx is w.r.t. the upstream block's QUANTISED output (the clip-STE applies)
    int up_wsh, up_H, up_W;     // UP 3: log2 of the plane's width (a power of two >= 8), its height and width
};

// 16-byte slot swizzle of a plane row: conflict-free b128 row reads (16 rows of one fragment) AND transpose reads (4 consecutive rows = 256 contiguous bytes)
__device__ __forceinline__ uint32_t pwb_sw(uint32_t row) { return (0x1320u >> (4u * ((row >> 2) & 3u))) & 3u; }

#ifdef PWB_TRACE          // s_memtime timeline of one block (scripts/variant_lib.sh trace qgemm_pwb.hip -DPWB_TRACE; MN_PWB_TRACE=1 prints steps 8..19 of the fifth call)
__device__ unsigned long long g_pwb_trace[3 * 64 * 8];
#define PWB_T(role_, t_, k_) do { if (blockIdx.x == 5 && wave == 1 && lane == 0 && (t_) < 64) g_pwb_trace[((role_) * 64 + (t_)) * 8 + (k_)] = clock64(); } while (0)
#else
#define PWB_T(role_, t_, k_) do { } while (0)
#endif

template <int BNH, int XENC, int WIDE, int UP = 0>
__global__ __launch_bounds__(768, 1) void k_pwb(const PwbParams p) {
    constexpr int NS = (WIDE || UP == 2) ? 3 : 4, PLANE = PWB_PLANE, BUF = UP == 2 ? PWB_BUF_UP2 : (UP ? PWB_BUF_UP : PWB_BUF);
    HIP_DYNAMIC_SHARED(float, smemw)
    unsigned char* lds = reinterpret_cast<unsigned char*>(smemw);          // [2][BUF], then the fold table [128][PWB_FROW]
    float* ftab = reinterpret_cast<float*>(lds + 2 * BUF);
    const int role = (int)threadIdx.x >> 8;                                // 0 producer, 1 dW, 2 dx
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const uint32_t bz = blockIdx.x;
    const int z = (int)(bz % (uint32_t)p.Z), g = (int)(bz / (uint32_t)p.Z);
    const uint32_t HW = (uint32_t)p.HW;
    // The block takes the contiguous range [z st_per_z, ...) of steps, started at a block-dependent ROTATION: the rows of a step are 4 KB apart, and every block of
    // a launch starting at offset 0 of its range walks the HBM channels in lock-step with all the others (L2: 224 -> 203 us; interleaved steps z, z + Z, ..: 226)
    const int st0 = z * p.st_per_z;
    int n = (st0 + p.st_per_z < p.nsteps ? st0 + p.st_per_z : p.nsteps) - st0;
    n = n > 0 ? n : 0;
    const int rot = n > 0 ? z % n : 0;
    auto pwb_step = [&](int k) -> int {          // global step index of the block's k-th step, k < n
        int kk = k + rot;
        kk = kk >= n ? kk - n : kk;
        return st0 + kk;
    };
    const int nit = (n + NS - 1) / NS * NS;                                // every role passes the same number of barriers

    if (role == 0) {
        // ---------------------------------------------------------------------------------------------------------------- producers
        const int sr = tid >> 3, sq = tid & 7, cr = tid >> 1, chf = tid & 1;          // dy rows sr + 32 i, pixels 4 sq ..; code row cr, pixels 16 chf ..
        uint32_t goff[4];
        float dbacc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { goff[i] = (uint32_t)(g * 128 + sr + 32 * i) * HW; dbacc[i] = 0.f; }
        const uint32_t xoff = (uint32_t)chan_phys(p.in_map, g * 128 + cr) * HW;
        const bool want_db = p.want_db != 0;
        if (tid < 128) {
            const float ks = p.kscale[g * 128 + tid];
            const float sc = ks == 0.f ? 1.f : ks;
            float* fr = ftab + tid * PWB_FROW;
            if (BNH == 3) {
                // the DoReFa block's channel (k_qa_apply's arithmetic): masks as ONE interval of the stash integer where the constants allow it (qa_mask_interval)
                const int co = g * 128 + tid, Cc = p.O;
                const float alpha = p.chan[co], bias = p.chan[Cc + co], mean = p.chan[2 * Cc + co], invstd = p.chan[3 * Cc + co], ga = p.chan[4 * Cc + co],
                            be = p.chan[5 * Cc + co], gi = p.chan[8 * Cc + co];
                auto fin = [](float v) { return fabsf(v) <= 1.0e9f; };
                float use = 0.f;
                QaInterval r; r.lo = 1.f; r.hi = 0.f;
                if (fin(alpha) && fin(bias) && fin(mean) && fin(invstd) && fin(ga) && fin(be) && ga != 0.f && invstd > 0.f && alpha != 0.f) {
                    auto zf = [&](float v) { const float y = v * alpha + bias; const float zh = (y - mean) * invstd; return zh * ga + be; };
                    r = qa_mask_interval(WIDE ? (1 << 24) : 32768, zf, [](int32_t w) { return (float)w; }, p.quant);
                    use = 1.f;
                }
                const float k1 = p.training ? p.sums[co] / p.n_f : 0.f, k2 = p.training ? p.sums[Cc + co] / p.n_f : 0.f;
                fr[0] = r.lo; fr[1] = r.hi; fr[2] = gi * sc; fr[3] = use;
                fr[4] = alpha; fr[5] = bias; fr[6] = mean; fr[7] = invstd;
                fr[8] = k1; fr[9] = k2; fr[10] = ga; fr[11] = be;
            } else {
                float hlo = 0.f, hhi = 0.f, G_ = sc, E1 = 0.f, E0 = 0.f;
                if (BNH) bnh_fold(p.chan, p.sums, p.O, g * 128 + tid, p.training, p.n_f, sc, hlo, hhi, G_, E1, E0);
                fr[0] = hlo; fr[1] = hhi; fr[2] = G_; fr[3] = E1; fr[4] = E0;
            }
        }
        constexpr int HV = WIDE ? 4 : (BNH == 3 ? 2 : 1), H1 = HV > 1 ? 1 : 0, H2 = HV > 2 ? 2 : 0, H3 = HV > 3 ? 3 : 0;          // stash words per row (H1..: in-range indices)
        struct Stage { float4 gv[4]; uint32_t hv[4][HV]; u32x4 cv; u32x4 uv; u32x4 uw; uint32_t hbit; };
        auto fetch = [&](Stage& S, int k) {
            // past the block's range: re-read the block's own last step (an L2 hit); loads stay unconditional
            const int kk = k < n ? k : (n > 0 ? n - 1 : 0);
            int st = pwb_step(kk);
            st = st < p.nsteps ? st : p.nsteps - 1;
            const uint32_t P = (uint32_t)st * 32u + 4u * sq;
            const uint32_t ni = fd_div(P, p.fd_hw);
            const uint32_t o = ni * (uint32_t)p.O * HW + (P - ni * HW);
            if (BNH == 2) {          // pooled gradient: {g[win 0], g[win 1], own codes of the windows' upper row, of their lower row} per row and pixel quad
                const uint32_t pp = P - ni * HW;
                const uint32_t hr = fd_div(pp, p.fd_w), w = pp - hr * (uint32_t)p.W;
                const uint32_t gbase = ni * (uint32_t)p.O * (HW >> 2) + (hr >> 1) * ((uint32_t)p.W >> 1) + (w >> 1);
                const uint32_t cbase = ni * (uint32_t)p.O * HW + (hr & ~1u) * (uint32_t)p.W + w;
                S.hbit = hr & 1u;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float2 g2 = *reinterpret_cast<const float2*>(p.gy + (gbase + (goff[i] >> 2)));
                    const uint32_t r0 = *reinterpret_cast<const uint32_t*>(p.own + (cbase + goff[i]));
                    const uint32_t r1 = *reinterpret_cast<const uint32_t*>(p.own + (cbase + goff[i] + (uint32_t)p.W));
                    S.gv[i] = make_float4(g2.x, g2.y, mn_u2f(r0), mn_u2f(r1));
                    S.hv[i][0] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const unsigned char*>(p.h) + (o + goff[i]));
                }
            } else {
                uint32_t bo[4];          // byte offsets (plan: 4 N O HW < 2^32): uniform base + 32-bit lane offset, each load from its own offset register
#pragma unroll
                for (int i = 0; i < 4; ++i) bo[i] = (o + goff[i]) * 4u;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    S.gv[i] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(p.gy) + bo[i]);
                    if (BNH == 1) S.hv[i][0] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const unsigned char*>(p.h) + (o + goff[i]));
                    if (BNH == 3 && !WIDE) {
                        const u32x2 u = *reinterpret_cast<const u32x2*>(reinterpret_cast<const char*>(p.h) + (bo[i] >> 1));
                        S.hv[i][0] = u[0]; S.hv[i][H1] = u[1];
                    }
                    if (BNH == 3 && WIDE) {
                        const u32x4 u = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.h) + bo[i]);
                        S.hv[i][0] = u[0]; S.hv[i][H1] = u[1]; S.hv[i][H2] = u[2]; S.hv[i][H3] = u[3];
                    }
                }
            }
            const uint32_t Pc = (uint32_t)st * 32u + 16u * chf;
            const uint32_t nc = fd_div(Pc, p.fd_hw);
            S.cv = *reinterpret_cast<const u32x4*>(p.x + (nc * (uint32_t)p.C * HW + (Pc - nc * HW) + xoff));
            if (UP == 1 || UP == 3) S.uv = *reinterpret_cast<const u32x4*>(p.up_h + (nc * (uint32_t)p.C * HW + (Pc - nc * HW) + xoff));
            if (UP == 2) {
                const unsigned char* us = p.up_h + 2u * (nc * (uint32_t)p.C * HW + (Pc - nc * HW) + xoff);          // (plan: 4 N C HW < 2^32)
                S.uv = *reinterpret_cast<const u32x4*>(us); S.uw = *reinterpret_cast<const u32x4*>(us + 16);
            }
        };
        const uint32_t doff = (uint32_t)sr * 64u + ((((uint32_t)sq >> 1) ^ pwb_sw((uint32_t)sr)) << 4) + (((uint32_t)sq & 1u) << 3);      // rows sr + 32 i share the swizzle
        auto commit = [&](Stage& S, int buf, bool valid) {
            unsigned char* A = lds + buf * BUF;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v[4] = {S.gv[i].x, S.gv[i].y, S.gv[i].z, S.gv[i].w};
                if (BNH == 2) {          // the two windows' gradients go to their first +1 in scan order (else element 0) -- if that element lies in this row
                    const uint32_t r0 = mn_f2u(S.gv[i].z), r1 = mn_f2u(S.gv[i].w);
#pragma unroll
                    for (int e2 = 0; e2 < 2; ++e2) {
                        const bool p00 = !((r0 >> (16 * e2)) & 0x80u), p01 = !((r0 >> (16 * e2 + 8)) & 0x80u);
                        const bool p10 = !((r1 >> (16 * e2)) & 0x80u), p11 = !((r1 >> (16 * e2 + 8)) & 0x80u);
                        const uint32_t win = p00 ? 0u : (p01 ? 1u : (p10 ? 2u : (p11 ? 3u : 0u)));
                        const float ge = e2 ? S.gv[i].y : S.gv[i].x;
                        v[2 * e2] = win == S.hbit * 2u ? ge : 0.f;
                        v[2 * e2 + 1] = win == S.hbit * 2u + 1u ? ge : 0.f;
                    }
                }
                const float* fr = ftab + (sr + 32 * i) * PWB_FROW;
                const float4 f0 = *reinterpret_cast<const float4*>(fr);
                if (BNH == 3) {
                    // f0 = lo, hi, gi * scale, use; f1 = alpha, bias, mean, invstd; f2 = k1, k2, gamma, beta
                    const float4 f1 = *reinterpret_cast<const float4*>(fr + 4), f2 = *reinterpret_cast<const float4*>(fr + 8);
                    float sv[4];
                    if (WIDE) {
                        sv[0] = (float)(int)S.hv[i][0]; sv[1] = (float)(int)S.hv[i][H1]; sv[2] = (float)(int)S.hv[i][H2]; sv[3] = (float)(int)S.hv[i][H3];
                    } else {
                        sv[0] = (float)(int16_t)(S.hv[i][0] & 0xffffu); sv[1] = (float)(int16_t)(S.hv[i][0] >> 16);
                        sv[2] = (float)(int16_t)(S.hv[i][H1] & 0xffffu); sv[3] = (float)(int16_t)(S.hv[i][H1] >> 16);
                    }
                    if (f0.w != 0.f) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float y = sv[e] * f1.x + f1.y;
                            const float zh = (y - f1.z) * f1.w;
                            const float d = p.quant ? dorefa_ste_core_m(v[e], p.qs, p.qinv) : v[e];
                            const float dz = (sv[e] >= f0.x && sv[e] <= f0.y) ? d : 0.f;
                            v[e] = f0.z * (dz - f2.x - zh * f2.y);
                        }
                    } else {          // a channel with non-finite (or absurd) constants, or gamma == 0: the element-wise masks
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float y = sv[e] * f1.x + f1.y;
                            const float zh = (y - f1.z) * f1.w;
                            const float zz = zh * f2.z + f2.w;
                            const float dz = qa_dz_m(v[e], qa_relu(zz), zz, p.qs, p.qinv, p.quant);
                            v[e] = f0.z * (dz - f2.x - zh * f2.y);
                        }
                    }
                } else if (BNH) {
                    const float fE0 = fr[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float hf = (float)((S.hv[i][0] >> (8 * e)) & 0xffu);
                        const float dz = (hf >= f0.x && hf <= f0.y) ? v[e] : 0.f;
                        v[e] = fmaf(f0.z, dz, fmaf(f0.w, hf, fE0));
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= f0.z;
                }
                if (want_db) dbacc[i] += valid ? (v[0] + v[1]) + (v[2] + v[3]) : 0.f;
                float r1[4], r2[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    r1[e] = v[e] - mn_bf16_head(v[e]);
                    r2[e] = r1[e] - mn_bf16_head(r1[e]);
                }
                unsigned char* d = A + doff + 32 * 64 * i;
                *reinterpret_cast<u32x2*>(d) = u32x2{mn_pack_hi16(v[0], v[1]), mn_pack_hi16(v[2], v[3])};
                *reinterpret_cast<u32x2*>(d + PLANE) = u32x2{mn_pack_hi16(r1[0], r1[1]), mn_pack_hi16(r1[2], r1[3])};
                *reinterpret_cast<u32x2*>(d + 2 * PLANE) = u32x2{mn_pack_hi16(r2[0], r2[1]), mn_pack_hi16(r2[2], r2[3])};
            }
            *reinterpret_cast<u32x4*>(A + 3 * PLANE + cr * 48 + 16 * chf) = S.cv;
            if (UP == 1 || UP == 3) *reinterpret_cast<u32x4*>(A + 3 * PLANE + PWB_CODES + cr * 48 + 16 * chf) = S.uv;
            if (UP == 2) {
                *reinterpret_cast<u32x4*>(A + 3 * PLANE + PWB_CODES + cr * PWB_UP2_ROW + 32 * chf) = S.uv;
                *reinterpret_cast<u32x4*>(A + 3 * PLANE + PWB_CODES + cr * PWB_UP2_ROW + 32 * chf + 16) = S.uw;
            }
        };
        Stage st[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) { fetch(st[k], k); MN_SCHED_FENCE(); }
        __syncthreads();                                                   // the fold table
        for (int t = 0; t < nit; t += NS) {
#pragma unroll
            for (int u = 0; u < NS; ++u) {
                PWB_T(0, t + u, 0);
                commit(st[u], (t + u) & 1, t + u < n);
                PWB_T(0, t + u, 1);
                fetch(st[u], t + u + NS);
                PWB_T(0, t + u, 2);
                __syncthreads();
                PWB_T(0, t + u, 3);
            }
        }
        __syncthreads();                                                   // the staging buffers are free (the dW waves stage their tiles through them)
        if (want_db) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = dbacc[i];
                v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);     // the 8 pixel quads of the row
                if (sq == 0) p.dbpart[((int64_t)z * p.G + g) * 128 + sr + 32 * i] = v;
            }
        }
    } else if (role == 1) {
        // ---------------------------------------------------------------------------------------------------------------- dW: 2 x 2 waves of 64 x 64
        const int j = lane & 15, kg = lane >> 4, wm = wave >> 1, wcn = wave & 1;
        const uint32_t aoff = (uint32_t)(wm * 64 + j) * 64u + ((((uint32_t)kg) ^ pwb_sw((uint32_t)j)) << 4);        // + mi * 1024 + plane * PLANE
        const uint32_t boff = 3u * PLANE + (uint32_t)(wcn * 64 + j) * 48u + 8u * (uint32_t)kg;                        // + ci * 768
        f32x4 acc[4][4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) acc[mi][ci] = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        for (int t = 0; t < nit; ++t) {
            __syncthreads();
            PWB_T(1, t, 0);
            if (t < n) {
                const unsigned char* A = lds + (t & 1) * BUF;
                // every LDS read of the step is issued before the first MFMA (read just in time, each group of MFMAs waits for its own LDS round trip)
                u32x2 braw[4];
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) braw[ci] = *reinterpret_cast<const u32x2*>(A + boff + ci * 768);
                u32x4 af[3][4];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) af[pl][mi] = *reinterpret_cast<const u32x4*>(A + pl * PLANE + aoff + mi * 1024);
                PWB_T(1, t, 1);
                MN_SCHED_FENCE();
                u32x4 bf[4];
#pragma unroll
                for (int ci = 0; ci < 4; ++ci) {
                    if (XENC) {          // k-bit activation codes j (bytes) -> bf16 j (exact)
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const uint32_t u = braw[ci][d >> 1] >> (16 * (d & 1));
                            bf[ci][d] = mn_pack_bf16x2((float)(u & 0xffu), (float)((u >> 8) & 0xffu));
                        }
                    } else {             // sign byte c -> bf16 +-1.0 = bytes {0x80, (c & 0x80) | 0x3F}
                        const uint32_t u = (braw[ci][0] & 0x80808080u) | 0x3F3F3F3Fu, v = (braw[ci][1] & 0x80808080u) | 0x3F3F3F3Fu;
                        bf[ci] = u32x4{mn_perm(u, 0x80u, 0x05000400u), mn_perm(u, 0x80u, 0x07000600u), mn_perm(v, 0x80u, 0x05000400u), mn_perm(v, 0x80u, 0x07000600u)};
                    }
                }
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                        for (int ci = 0; ci < 4; ++ci) acc[mi][ci] = mn_mfma_bf16(af[pl][mi], bf[ci], acc[mi][ci]);
                PWB_T(1, t, 2);
            }
        }
        __syncthreads();                                                   // every wave is done with the staging buffers
        // partial tile through LDS: a store instruction covers four whole 256-byte rows
        float* T = reinterpret_cast<float*>(lds) + wave * 64 * PWB_ERS;
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                for (int r = 0; r < 4; ++r) T[(mi * 16 + kg * 4 + r) * PWB_ERS + ci * 16 + j] = acc[mi][ci][r];
        MN_WAVE_SYNC();
        float* dst = p.part + (((int64_t)z * p.G + g) * 128 + wm * 64) * 128 + wcn * 64;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int row = it * 4 + (lane >> 4), col = 4 * (lane & 15);
            *reinterpret_cast<float4*>(dst + (int64_t)row * 128 + col) = *reinterpret_cast<const float4*>(T + row * PWB_ERS + col);
        }
    } else {
        // ---------------------------------------------------------------------------------------------------------------- dx: wave w = input channels 32 w ..
        const int m = lane & 31, kgrp = lane >> 5, i16 = lane & 15, nhalf = (lane >> 4) & 1;
        u32x4 wf[8];                       // A fragments: W[o = 16 ks + 8 kgrp + e][c = 32 wave + m]
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            wf[ks] = *reinterpret_cast<const u32x4*>(p.wc + ((int64_t)(g * 128 + 32 * wave + m) * 128 + 16 * ks + 8 * kgrp));
        // BYTE offset of accumulator register r = row (r & 3) + 8 (r >> 2) + 4 kgrp of the wave's 32, column m (plan: 4 N C HW < 2^32), split into a UNIFORM part per
        // register (scalar registers: the row of kgrp 0) and ONE lane part (kgrp, m): the channel shuffle is separable over these rows because the shuffle groups
        // divide 32 (plan_pwb) -- phys(c0 + d + 4 k) = phys(c0 + d) + phys(c0 + 4 k) - phys(c0).  (Round 6: sixteen per-lane offsets cost 16-32 VGPRs.)
        const int wvu = mn_uniform(wave);
        uint32_t urow[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) urow[r] = (uint32_t)chan_phys(p.in_map, g * 128 + 32 * wvu + (r & 3) + 8 * (r >> 2)) * HW * 4u;
        const uint32_t vlane = (kgrp ? (uint32_t)(chan_phys(p.in_map, g * 128 + 32 * wvu + 4) - chan_phys(p.in_map, g * 128 + 32 * wvu)) * HW * 4u : 0u) + (uint32_t)m * 4u;
        uint32_t toff[2];                  // transpose reads r2 = 0, 1 of a K-step: plane rows 16 ks + 8 kgrp + 4 r2 + (i16 >> 2), pixels 16 nhalf + 4 (i16 & 3) ..
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
            const uint32_t row = (uint32_t)(8 * kgrp + 4 * r2 + (i16 >> 2));
            const uint32_t slot = (uint32_t)(2 * nhalf + ((i16 & 3) >> 1));
            toff[r2] = row * 64u + ((slot ^ pwb_sw(row)) << 4) + (((uint32_t)i16 & 1u) << 3);
        }
        // UP: the dx tile of a step goes through a private LDS tile [32 c][36] per wave and comes back TRANSPOSED -- lane = one channel (lane & 31) x one half of the
        // step's 32 pixels -- so that the sums of the upstream BatchNorm backward are TWO registers per lane (per-register accumulators for the 16 channels of the
        // MFMA layout: 32 + 16 VGPRs, the weight fragments spilled).  ulo / usp: the channel's mask interval [hlo, hhi] as (lo, hi - lo); an empty interval is (255, 0):
        // the stash of a conv with K <= 254 never reaches 255 (host check).
        uint32_t ulo = 255u, usp = 0u;
        float us1 = 0.f, ush = 0.f, unnz = 0.f;          // ush: sum dz (2 h - nnz) -- the integer the stash stands for, not h itself (2 h and nnz cancel)
        float* UT = reinterpret_cast<float*>(lds + 2 * BUF + 128 * PWB_FROW * 4) + wave * (32 * PWB_UTS);
        float qa_alpha = 0.f, qa_bias = 0.f, qa_mean = 0.f, qa_invstd = 0.f, qa_ga = 0.f, qa_be = 0.f, qa_lo = 1.f, qa_hi = 0.f;          // UP 2: the lane's upstream channel
        bool qa_use = false;
        if (UP == 2) {
            const int ch = chan_phys(p.in_map, g * 128 + 32 * wave + (lane & 31)), Cu = p.C;
            qa_alpha = p.up_chan[ch]; qa_bias = p.up_chan[Cu + ch]; qa_mean = p.up_chan[2 * Cu + ch]; qa_invstd = p.up_chan[3 * Cu + ch];
            qa_ga = p.up_chan[4 * Cu + ch]; qa_be = p.up_chan[5 * Cu + ch];
            auto fin = [](float v) { return fabsf(v) <= 1.0e9f; };
            if (fin(qa_alpha) && fin(qa_bias) && fin(qa_mean) && fin(qa_invstd) && fin(qa_ga) && fin(qa_be) && qa_ga != 0.f && qa_invstd > 0.f && qa_alpha != 0.f) {
                auto zf = [&](float v) { const float y = v * qa_alpha + qa_bias; const float zh = (y - qa_mean) * qa_invstd; return zh * qa_ga + qa_be; };
                const QaInterval r = qa_mask_interval(32768, zf, [](int32_t w) { return (float)w; }, p.up_quant);
                qa_lo = r.lo; qa_hi = r.hi; qa_use = true;
            }
        }
        // UP 3: the lane's upstream channel -- flip, the pass interval [L, U] of acc * flip as (mid, half): L <= u <= U <=> |u - mid| <= half, exact for the small
        // integers u takes once the bounds are clamped to +-512 (|acc| <= 254); NaN bounds / an empty interval: half < 0, nothing passes -- and the nnz of the nine
        // pixel classes.  Two elements per instruction (packed fp32): the dx waves set the pace of the un-pooled variants, a VALU instruction costs them 4 cycles.
        typedef float pf2 __attribute__((vector_size(8)));
        pf2 u9_fl2 = {0.f, 0.f}, u9_mid2 = {0.f, 0.f}, us1v = {0.f, 0.f}, ushv = {0.f, 0.f};
        float u9_half = -1.f;
        StashNnz u9;
        u9.v0 = u9.v1 = u9.v2 = u9.v3 = u9.v4 = u9.v5 = u9.v6 = u9.v7 = u9.v8 = 0.f;
        if (UP == 3) {
            const int ch = chan_phys(p.in_map, g * 128 + 32 * wave + (lane & 31)), Cu = p.C;
            const float fl = p.up_chan[Cu + ch], L = p.up_chan[2 * Cu + ch], U = p.up_chan[3 * Cu + ch];
            const bool ok = L == L && U == U && fl == fl;
            const float Lc = fmaxf(L, -512.f), Uc = fminf(U, 512.f);
            const float mid = 0.5f * (Lc + Uc);
            u9_half = ok ? 0.5f * (Uc - Lc) : -1.f;
            u9_fl2 = pf2{fl, fl}; u9_mid2 = pf2{ok ? mid : 0.f, ok ? mid : 0.f};
            u9 = stash_nnz_load(p.up_chan, Cu, ch);
        }
        if (UP == 1) {
            const int ch = chan_phys(p.in_map, g * 128 + 32 * wave + (lane & 31)), Cu = p.C;
            const float fl = p.up_chan[Cu + ch], L = p.up_chan[2 * Cu + ch], U = p.up_chan[3 * Cu + ch], nnz = p.up_chan[7 * Cu + ch];
            float hlo, hhi;
            if (fl > 0.f) { hlo = ceilf((L + nnz) * 0.5f); hhi = floorf((U + nnz) * 0.5f); }
            else { hlo = ceilf((nnz - U) * 0.5f); hhi = floorf((nnz - L) * 0.5f); }
            const bool ok = hlo <= hhi && hhi >= 0.f && hlo <= 254.f;          // (NaN constants: nothing passes, as in the streaming kernels)
            ulo = ok ? (uint32_t)fmaxf(hlo, 0.f) : 255u;
            usp = ok ? (uint32_t)fminf(hhi, 254.f) - ulo : 0u;
            unnz = nnz;
        }
        __syncthreads();
        for (int t = 0; t < nit; ++t) {
            __syncthreads();
            PWB_T(2, t, 0);
            if (t < n) {
                const unsigned char* A = lds + (t & 1) * BUF;
                f32x16 a0, a1;
#pragma unroll
                for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
                // B fragments of K-step ks (three planes).  K-steps go in pairs -- the even one into a0, the odd one into a1, alternating, so that two MFMAs on one
                // accumulator are never adjacent -- and a ring of two pairs sits in registers: an LDS round trip is hidden behind the six MFMAs of a pair
                u32x4 bq[4][3];
                auto ldk = [&](u32x4 (&dstq)[3], int ks) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        const mn_u32x2 lo = mn_lds_tr16_b64(A + pl * PLANE + ks * 1024 + toff[0]);
                        const mn_u32x2 hi = mn_lds_tr16_b64(A + pl * PLANE + ks * 1024 + toff[1]);
                        dstq[pl] = u32x4{lo[0], lo[1], hi[0], hi[1]};
                    }
                };
                ldk(bq[0], 0); ldk(bq[1], 1); ldk(bq[2], 2); ldk(bq[3], 3);
                MN_SCHED_FENCE();
#pragma unroll
                for (int kp = 0; kp < 4; ++kp) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
                        a0 = mn_mfma32_bf16(wf[2 * kp], bq[(2 * kp) & 3][pl], a0);
                        a1 = mn_mfma32_bf16(wf[2 * kp + 1], bq[(2 * kp + 1) & 3][pl], a1);
                    }
                    if (kp + 2 < 4) { ldk(bq[(2 * kp) & 3], 2 * kp + 4); ldk(bq[(2 * kp + 1) & 3], 2 * kp + 5); }
                    MN_SCHED_FENCE();
                }
                PWB_T(2, t, 1);
                const uint32_t P = (uint32_t)pwb_step(t) * 32u;
                const uint32_t ni = fd_div(P, p.fd_hw);
                // the 16 row segments leave from 16 DIFFERENT registers, addressed as uniform base + 32-bit lane offset: a store whose address / data register is
                // rewritten for the next one waits until the memory pipeline has read it (measured: ~150 cycles per store, half of the step)
                char* dst = reinterpret_cast<char*>(p.dx + (ni * (uint32_t)p.C * HW + (P - ni * HW)));
                float outv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) outv[r] = a0[r] + a1[r];
                const uint32_t vv = mn_opaque(vlane);          // (opaque: the zero-extension stays in this block -> saddr form, base = dst + urow[r] in scalar registers)
                MN_SCHED_FENCE();
#pragma unroll
                for (int r = 0; r < 16; ++r) mn_store_nt(reinterpret_cast<float*>((dst + urow[r]) + vv), outv[r]);          // streaming: L5 95 -> 81 us, L6 90 -> 84, L2 / L3 / L8 unchanged
                if (UP) {          // (behind the stores: the MFMA operands' registers are free; this step's buffer stays valid until the next barrier)
                    MN_WAVE_SYNC();          // (emulation: the previous step's transposed reads are done)
#pragma unroll
                    for (int r = 0; r < 16; ++r) UT[((r & 3) + 8 * (r >> 2) + 4 * kgrp) * PWB_UTS + m] = outv[r];
                    MN_WAVE_SYNC();
                    const int cl = lane & 31, phx = lane >> 5;
                    float4 dq[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) dq[q] = *reinterpret_cast<const float4*>(UT + cl * PWB_UTS + 16 * phx + 4 * q);
                    if (UP == 1) {
                        const u32x4 hq = *reinterpret_cast<const u32x4*>(A + 3 * PLANE + PWB_CODES + (32 * wave + cl) * 48 + 16 * phx);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float dv[4] = {dq[q].x, dq[q].y, dq[q].z, dq[q].w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const uint32_t hb = (hq[q] >> (8 * e)) & 0xffu;
                                const float dz = (hb - ulo) <= usp ? dv[e] : 0.f;
                                us1 += dz;
                                ush += dz * (2.f * (float)hb - unnz);
                            }
                        }
                    } else if (UP == 3) {          // k_bnh_partial<0>'s arithmetic with the nnz of each pixel's border class
                        const u32x4 hq = *reinterpret_cast<const u32x4*>(A + 3 * PLANE + PWB_CODES + (32 * wave + cl) * 48 + 16 * phx);
                        const uint32_t pix0 = (P - ni * HW) + 16u * (uint32_t)phx;          // first of the lane's 16 pixels inside its plane
#pragma unroll
                        for (int hc = 0; hc < 2; ++hc) {          // two runs of 8 pixels: each inside one row (W >= 8, a power of two)
                            const uint32_t px = pix0 + 8u * hc;
                            const int row = (int)(px >> p.up_wsh), col = (int)(px & (uint32_t)(p.up_W - 1));
                            const float top = row == 0 ? 1.f : 0.f, bot = row == p.up_H - 1 ? 1.f : 0.f;
                            const float m0 = u9.v3 + top * (u9.v0 - u9.v3) + bot * (u9.v6 - u9.v3);
                            const float m1 = u9.v4 + top * (u9.v1 - u9.v4) + bot * (u9.v7 - u9.v4);
                            const float m2 = u9.v5 + top * (u9.v2 - u9.v5) + bot * (u9.v8 - u9.v5);
                            const float nfirst = col == 0 ? m0 : m1, nlast = col + 7 == p.up_W - 1 ? m2 : m1;
#pragma unroll
                            for (int q2 = 0; q2 < 2; ++q2) {
                                const int q = 2 * hc + q2;
                                const float dv[4] = {dq[q].x, dq[q].y, dq[q].z, dq[q].w};
#pragma unroll
                                for (int pr = 0; pr < 2; ++pr) {
                                    const pf2 h2 = {(float)((hq[q] >> (16 * pr)) & 0xffu), (float)((hq[q] >> (16 * pr + 8)) & 0xffu)};
                                    const pf2 nz2 = {(q2 == 0 && pr == 0) ? nfirst : m1, (q2 == 1 && pr == 1) ? nlast : m1};
                                    const pf2 acc2 = (h2 + h2) - nz2;
                                    const pf2 w2 = acc2 * u9_fl2 - u9_mid2;
                                    const pf2 dz2 = {fabsf(w2[0]) <= u9_half ? dv[2 * pr] : 0.f, fabsf(w2[1]) <= u9_half ? dv[2 * pr + 1] : 0.f};
                                    us1v += dz2;
                                    ushv += dz2 * acc2;
                                }
                            }
                        }
                    } else {          // UP 2: k_qa_partial's arithmetic on (this dx, the upstream 16-bit stash)
                        const unsigned char* hs = A + 3 * PLANE + PWB_CODES + (32 * wave + cl) * PWB_UP2_ROW + 32 * phx;
                        const u32x4 s0 = *reinterpret_cast<const u32x4*>(hs), s1 = *reinterpret_cast<const u32x4*>(hs + 16);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float dv[4] = {dq[q].x, dq[q].y, dq[q].z, dq[q].w};
                            const uint32_t w0 = q < 2 ? s0[2 * q] : s1[2 * q - 4], w1 = q < 2 ? s0[2 * q + 1] : s1[2 * q - 3];
                            const float sv[4] = {(float)(int16_t)(w0 & 0xffffu), (float)(int16_t)(w0 >> 16), (float)(int16_t)(w1 & 0xffffu), (float)(int16_t)(w1 >> 16)};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float y = sv[e] * qa_alpha + qa_bias;
                                const float zh = (y - qa_mean) * qa_invstd;
                                float dz;
                                if (qa_use) {
                                    const float d = p.up_quant ? dorefa_ste_core_m(dv[e], p.up_qs, p.up_qinv) : dv[e];
                                    dz = (sv[e] >= qa_lo && sv[e] <= qa_hi) ? d : 0.f;
                                } else {
                                    const float zz = zh * qa_ga + qa_be;
                                    dz = qa_dz_m(dv[e], qa_relu(zz), zz, p.up_qs, p.up_qinv, p.up_quant);
                                }
                                us1 += dz;
                                ush += dz * zh;
                            }
                        }
                    }
                }
                PWB_T(2, t, 2);
            }
        }
        __syncthreads();
        if (UP) {
            if (UP == 3) { us1 = us1v[0] + us1v[1]; ush = ushv[0] + ushv[1]; }
            us1 += __shfl_xor(us1, 32, 64); ush += __shfl_xor(ush, 32, 64);          // the two pixel halves of the channel
            if (lane < 32) {
                const int ch = chan_phys(p.in_map, g * 128 + 32 * wave + lane), Cu = p.C;
                (void)Cu;
                const double d1 = (double)us1, da_ = (double)ush;
                double* dstp = p.up_part + ((int64_t)ch * p.Z + z) * 2;
                dstp[0] = d1;
                if (UP == 1 || UP == 3) {
                    const double A_ = (double)p.up_chan[4 * Cu + ch], B_ = (double)p.up_chan[5 * Cu + ch];
                    dstp[1] = A_ * da_ + B_ * d1;          // zhat = A (2 h - nnz) + B
                } else {
                    dstp[1] = da_;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
int pwd_pack_plan(const mn_conv_geom* g, PackParams* pk, int* grid, int64_t* off_scale, int64_t* bytes);          // qgemm_kernels.hip
void qg_launch_wgrad_reduce_div(const float* part, const float* dbpart, float* dw, float* db, int Z, int G, int Mg, int Cg, int Mgw, int Cgw, float ascale,
                                const float* rowdiv, hipStream_t s);

struct PwbPlan { PwbParams p; PackParams pk; int pack_grid, grid; int64_t off_scale, pack_bytes, off_part, off_db, ws_bytes; };
static int plan_pwb(const mn_conv_geom* g, PwbPlan* pl) {
    if (g->KH != 1 || g->KW != 1 || g->stride_h != 1 || g->stride_w != 1 || g->pad_h != 0 || g->pad_w != 0) return 0;
    if (g->groups < 1 || g->C != 128 * g->groups || g->O != 128 * g->groups) return 0;
    const int64_t HW = g->H * g->W, NP = (int64_t)g->N * HW;
    if (HW % 32 || NP <= 0) return 0;
    if (4 * NP * g->C >= ((int64_t)1 << 32)) return 0;                     // 32-bit byte offsets
    if (g->in_shuffle > 1 && (g->C % g->in_shuffle || 32 % g->in_shuffle)) return 0;          // (the dx rows' shuffle must be separable over a wave's 32 channels)
    if (!pwd_pack_plan(g, &pl->pk, &pl->pack_grid, &pl->off_scale, &pl->pack_bytes)) return 0;
    if (pl->pk.Cpad != 128 || pl->pk.Mgp != 128) return 0;
    PwbParams& p = pl->p;
    p.N = (int)g->N; p.HW = (int)HW; p.C = (int)g->C; p.O = (int)g->O; p.G = (int)g->groups; p.W = (int)g->W;
    p.nsteps = (int)(NP / 32);
    int Z = 256 / p.G;                                                     // one block per CU (Z = 64 / 256 / 512 per group of L2: 297 / 231 / 236 us against 212)
    if (Z > p.nsteps) Z = p.nsteps;
    if (Z < 1) Z = 1;
    p.Z = Z;
    p.st_per_z = (p.nsteps + Z - 1) / Z;
    p.fd_hw = make_fastdiv((uint32_t)HW); p.fd_w = make_fastdiv((uint32_t)g->W);
    p.in_map = make_chanmap(g->in_shuffle, g->C);
    p.n_f = (float)g->N * (float)HW;
    p.quant = 0; p.qs = 1.f; p.qinv = 0.f;
    pl->grid = p.G * Z;
    pl->off_part = (pl->pack_bytes + 255) / 256 * 256;
    const int64_t part_bytes = (int64_t)Z * p.G * 128 * 128 * 4;
    pl->off_db = pl->off_part + (part_bytes + 255) / 256 * 256;
    pl->ws_bytes = pl->off_db + (int64_t)Z * p.G * 128 * 4;
    return 1;
}
int pwb_supported(const mn_conv_geom* g, const mn_wq* wq, int pooled) {
    PwbPlan pl;
    if (!wq_codeable(wq) || !plan_pwb(g, &pl)) return 0;
    if (pooled && ((g->H & 1) || (g->W & 3))) return 0;
    return 1;
}
int64_t pwb_ws_bytes(const mn_conv_geom* g) { PwbPlan pl; return plan_pwb(g, &pl) ? pl.ws_bytes : 0; }

template <int BNH, int XENC, int WIDE, int UP = 0>
static void pwb_launch(const PwbPlan& pl, hipStream_t s) {
    constexpr size_t lds_up = UP == 2 ? PWB_LDS_UP2 : PWB_LDS_UP;
    constexpr size_t lds = UP ? (lds_up > PWB_LDS ? lds_up : PWB_LDS) : PWB_LDS;
    raise_lds_limit((const void*)k_pwb<BNH, XENC, WIDE, UP>, lds);
    hipLaunchKernelGGL((k_pwb<BNH, XENC, WIDE, UP>), dim3(pl.grid), dim3(768), lds, s, pl.p);
}
#ifdef PWB_TRACE
static void pwb_trace_dump(hipStream_t s) {
    static int once = 0;
    if (!MN_ENV("MN_PWB_TRACE") || once++ != 4) return;
    (void)hipStreamSynchronize(s);
    static unsigned long long hbuf[3 * 64 * 8];
    (void)hipMemcpyFromSymbol(hbuf, HIP_SYMBOL(g_pwb_trace), sizeof hbuf);
    const unsigned long long b0 = hbuf[(0 * 64 + 8) * 8 + 0];
    for (int t = 8; t < 20; ++t) {
        auto v = [&](int role, int k) { return (long long)(hbuf[(role * 64 + t) * 8 + k] - b0); };
        fprintf(stderr, "t %2d  prod: top %6lld commit %6lld fetch %6lld barrier %6lld | dW: bar %6lld reads %6lld mfma %6lld | dx: bar %6lld mfma %6lld stores %6lld\n", t,
                v(0, 0), v(0, 1), v(0, 2), v(0, 3), v(1, 0), v(1, 1), v(1, 2), v(2, 0), v(2, 1), v(2, 2));
    }
}
#endif
// mode: 1 wbwtab (h = byte stash; own != NULL: pooled), 3 DoReFa fold (h = 16 / 32-bit stash), 0 plain dy; xenc: 0 sign codes, 1 k-bit codes (dW times ascale)
// the upstream block (UP variants): kind 1: byte stash + [8][C] constants (wbwtab); kind 2: 16-bit stash + [9][C] constants, the width of the quantizer behind it and
// whether this dx is w.r.t. its quantised output (DoReFa); partials [C][Z][2]
struct PwbUp { int kind; const void* h; const float* chan; double* part; int bits, quant; };
static int pwb_run(const char* what, int mode, int xenc, int wide, const mn_conv_geom* g, const mn_wq* wq, const float* gy, const void* h, const int8_t* own,
                   const float* chan, const float* sums, int training, int quant, float qs, float ascale, const float* w, const void* x, float* dx, float* dw,
                   float* dbias, void* ws, int64_t ws_bytes, hipStream_t s, const PwbUp* up = nullptr) {
    PwbPlan pl;
    if (!wq_codeable(wq) || !plan_pwb(g, &pl) || (own && ((g->H & 1) || (g->W & 3)))) MN_FAIL(MN_ENOTSUP, "%s: geometry / quantizer combination not covered", what);
    if (!aligned16(dx) || !aligned16(dw) || (((uintptr_t)x) & 15) || (h && (((uintptr_t)h) & 15)) || (own ? ((((uintptr_t)gy) & 7) || (((uintptr_t)own) & 3)) : !aligned16(gy)))
        MN_FAIL(MN_ENOTSUP, "%s: misaligned tensor", what);
    if (!ws || ws_bytes < pl.ws_bytes || !aligned16(ws)) MN_FAIL(MN_ENOSPC, "%s: workspace too small", what);
    PwbParams& p = pl.p;
    if (wq->packed_bwd && mn_use_packed()) {
        p.wc = (const uint16_t*)wq->packed_bwd;
        p.kscale = (const float*)((const char*)wq->packed_bwd + pl.off_scale);
    } else {
        fill_pack(pl.pk, wq, w, ws, 0, pl.off_scale);
        qg_launch_pack(pl.pk, pl.pack_grid, s);
        p.wc = pl.pk.codes; p.kscale = pl.pk.scale_out;
    }
    p.gy = gy; p.h = h; p.chan = chan; p.sums = sums; p.own = (const char*)own; p.x = (const char*)x; p.dx = dx;
    p.part = (float*)((char*)ws + pl.off_part); p.dbpart = (float*)((char*)ws + pl.off_db);
    p.want_db = dbias != nullptr; p.training = training;
    p.quant = quant; p.qs = qs; p.qinv = mn_qa_inv(qs);
    p.up_h = up ? (const unsigned char*)up->h : nullptr; p.up_chan = up ? up->chan : nullptr; p.up_part = up ? up->part : nullptr;
    p.up_quant = up ? up->quant : 0; p.up_qs = (up && up->kind == 2) ? dorefa_scale(up->bits) : 1.f; p.up_qinv = mn_qa_inv(p.up_qs);
    p.up_H = (int)g->H; p.up_W = (int)g->W; p.up_wsh = 0;
    while ((1 << p.up_wsh) < p.up_W) ++p.up_wsh;
    const int bnh = mode == 1 ? (own ? 2 : 1) : mode;
    if (up) {
        if (up->kind == 1 ? (mode != 1 || xenc || wide) :
            up->kind == 3 ? (mode != 1 || own || xenc || wide || g->W < 8 || (g->W & (g->W - 1))) :
            (up->kind != 2 || (mode != 3 && mode != 0) || !xenc || wide || up->bits < 1 || up->bits > 8))
            MN_FAIL(MN_ENOTSUP, "%s: the upstream sums do not ride on this variant", what);
        if (!up->h || !up->chan || !up->part || (((uintptr_t)up->h) & 15) || (((uintptr_t)up->part) & 7)) MN_FAIL(MN_EINVAL, "%s: null / misaligned upstream operand", what);
        mn_set_last_kernel("k_pwb<%d, %d, %d, %d>", bnh, xenc, wide, up->kind);
    } else
    mn_set_last_kernel("k_pwb<%d, %d, %d>", bnh, xenc, wide);
    {
        const double nx = (double)g->N * g->C * g->H * g->W, ny = (double)g->N * g->O * g->H * g->W;
        mn_prof_bytes((mode == 1 ? (own ? 3.0 : 5.0) : (mode == 3 ? (wide ? 8.0 : 6.0) : 4.0)) * ny + (up ? (up->kind == 2 ? 7.0 : 6.0) : 5.0) * nx);
    }
    mn_prof_begin(s);
    if (up && up->kind == 3) pwb_launch<1, 0, 0, 3>(pl, s);
    else if (up && bnh == 1) pwb_launch<1, 0, 0, 1>(pl, s);
    else if (up && bnh == 2) pwb_launch<2, 0, 0, 1>(pl, s);
    else if (up && bnh == 3) pwb_launch<3, 1, 0, 2>(pl, s);
    else if (up && bnh == 0) pwb_launch<0, 1, 0, 2>(pl, s);
    else if (bnh == 1 && !xenc) pwb_launch<1, 0, 0>(pl, s);
    else if (bnh == 2 && !xenc) pwb_launch<2, 0, 0>(pl, s);
    else if (bnh == 0 && !xenc) pwb_launch<0, 0, 0>(pl, s);
    else if (bnh == 0) pwb_launch<0, 1, 0>(pl, s);
    else if (bnh == 3 && xenc && !wide) pwb_launch<3, 1, 0>(pl, s);
    else if (bnh == 3 && xenc) pwb_launch<3, 1, 1>(pl, s);
    else MN_FAIL(MN_ENOTSUP, "%s: variant not built", what);
    mn_prof_end(s);
#ifdef PWB_TRACE
    pwb_trace_dump(s);
#endif
    qg_launch_wgrad_reduce_div(p.part, p.dbpart, dw, dbias, p.Z, p.G, 128, 128, 128, 128, ascale, p.kscale, s);
    MN_CHECK_LAUNCH(what);
    return MN_OK;
}
int pwb_bwd_bnh(const mn_conv_geom* g, const mn_wq* wq, const float* da, const uint8_t* h, const int8_t* own, const float* chan, const float* sums, int training,
                const float* w, const int8_t* x, float* dx, float* dw, float* dbias, void* ws, int64_t ws_bytes, hipStream_t s) {
    return pwb_run("mn_conv2d_bwd_bnh", 1, 0, 0, g, wq, da, h, own, chan, sums, training, 0, 1.f, 1.f, w, x, dx, dw, dbias, ws, ws_bytes, s);
}
// ... with the BatchNorm-backward sums of the block in front (UP): *splits = the number of partials per channel the launch leaves in up_part ([C][splits][2] doubles)
int pwb_up_splits(const mn_conv_geom* g) { PwbPlan pl; return plan_pwb(g, &pl) ? pl.p.Z : 0; }
int pwb_bwd_bnh_up(const mn_conv_geom* g, const mn_wq* wq, const float* da, const uint8_t* h, const int8_t* own, const float* chan, const float* sums, int training,
                   const float* w, const int8_t* x, float* dx, float* dw, float* dbias, void* ws, int64_t ws_bytes, const uint8_t* up_h, const float* up_chan,
                   double* up_part, hipStream_t s) {
    const PwbUp up{1, up_h, up_chan, up_part, 0, 0};
    return pwb_run("mn_conv2d_bwd_bnh_up", 1, 0, 0, g, wq, da, h, own, chan, sums, training, 0, 1.f, 1.f, w, x, dx, dw, dbias, ws, ws_bytes, s, &up);
}
// the block in front is a 3x3 / padding-1 BatchNorm+sign block: up_chan = its [17][C] constants (rows 8..16: nnz of the nine pixel classes); un-pooled consumers only
int pwb_up9_splits(const mn_conv_geom* g) { PwbPlan pl; return (plan_pwb(g, &pl) && g->W >= 8 && !(g->W & (g->W - 1))) ? pl.p.Z : 0; }
int pwb_bwd_bnh_up9(const mn_conv_geom* g, const mn_wq* wq, const float* da, const uint8_t* h, const float* chan, const float* sums, int training, const float* w,
                    const int8_t* x, float* dx, float* dw, float* dbias, void* ws, int64_t ws_bytes, const uint8_t* up_h, const float* up_chan, double* up_part,
                    hipStream_t s) {
    const PwbUp up{3, up_h, up_chan, up_part, 0, 0};
    return pwb_run("mn_conv2d_bwd_bnh_up9", 1, 0, 0, g, wq, da, h, nullptr, chan, sums, training, 0, 1.f, 1.f, w, x, dx, dw, dbias, ws, ws_bytes, s, &up);
}
// the k-bit (DoReFa) counterparts: the block in front left a 16-bit stash; up_bits = the width of ITS output quantizer (= x_bits), up_quant: this dx is w.r.t. its codes
int pwb_bwd_plain_up(const mn_conv_geom* g, const mn_wq* wq, const float* gy, const float* w, const void* x, int x_bits, float* dx, float* dw, float* dbias, void* ws,
                     int64_t ws_bytes, const void* up_stash, const float* up_chan, int up_quant, double* up_part, hipStream_t s) {
    if (x_bits < 1 || x_bits > 8) MN_FAIL(MN_EINVAL, "mn_conv2d_bwd_codes_up: bad bit width");
    const PwbUp up{2, up_stash, up_chan, up_part, x_bits, up_quant};
    return pwb_run("mn_conv2d_bwd_codes_up", 0, 1, 0, g, wq, gy, nullptr, nullptr, nullptr, nullptr, 0, 0, 1.f, dorefa_scale(x_bits), w, x, dx, dw, dbias, ws, ws_bytes, s, &up);
}
int pwb_bwd_qa_up(const mn_conv_geom* g, const mn_wq* wq, const float* dq, const void* stash, const float* chan, const float* sums, int out_bits, int quant, int training,
                  const float* w, const uint8_t* x, int x_bits, float* dx, float* dw, float* dbias, void* ws, int64_t ws_bytes, const void* up_stash, const float* up_chan,
                  int up_quant, double* up_part, hipStream_t s) {
    if (x_bits < 1 || x_bits > 8 || out_bits < 1 || out_bits > 8) MN_FAIL(MN_EINVAL, "mn_conv2d_bwd_qa_up: bad bit width");
    const PwbUp up{2, up_stash, up_chan, up_part, x_bits, up_quant};
    return pwb_run("mn_conv2d_bwd_qa_up", 3, 1, 0, g, wq, dq, stash, nullptr, chan, sums, training, quant, dorefa_scale(out_bits), dorefa_scale(x_bits), w, x, dx, dw, dbias, ws,
                   ws_bytes, s, &up);
}
// plain gradient: x_bits == 0: sign codes, else k-bit activation codes of that width
int pwb_bwd_plain(const mn_conv_geom* g, const mn_wq* wq, const float* gy, const float* w, const void* x, int x_bits, float* dx, float* dw, float* dbias, void* ws,
                  int64_t ws_bytes, hipStream_t s) {
    const float ascale = x_bits ? dorefa_scale(x_bits) : 1.f;
    return pwb_run("mn_conv2d_bwd_codes", 0, x_bits ? 1 : 0, 0, g, wq, gy, nullptr, nullptr, nullptr, nullptr, 0, 0, 1.f, ascale, w, x, dx, dw, dbias, ws, ws_bytes, s);
}
// DoReFa block: dq = gradient w.r.t. the block's output (quant: w.r.t. its out_bits-quantised output), stash of stash_bits = 16 / 32, chan [9][O], sums [2][O]
int pwb_bwd_qa(const mn_conv_geom* g, const mn_wq* wq, const float* dq, const void* stash, int stash_bits, const float* chan, const float* sums, int out_bits, int quant,
               int training, const float* w, const uint8_t* x, int x_bits, float* dx, float* dw, float* dbias, void* ws, int64_t ws_bytes, hipStream_t s) {
    if ((stash_bits != 16 && stash_bits != 32) || x_bits < 1 || x_bits > 8 || out_bits < 1 || out_bits > 8) MN_FAIL(MN_EINVAL, "mn_conv2d_bwd_qa: bad bit width");
    const float ascale = dorefa_scale(x_bits), qs = dorefa_scale(out_bits);
    return pwb_run("mn_conv2d_bwd_qa", 3, 1, stash_bits == 32, g, wq, dq, stash, nullptr, chan, sums, training, quant, qs, ascale, w, x, dx, dw, dbias, ws, ws_bytes, s);
}
