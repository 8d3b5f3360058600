// First-layer convolution (few input channels, real-valued fp32 operands) for gfx950: forward and backward-weight.
//
// The first convolution of the reference's nets reads the image (3 channels) with full-precision weights (it is skipped by
// the DoReFa and WbWtAb rewrites: wqaq/dorefa/quantize.py:206, wbwtab/quantize.py:251).  K = Cin*KH*KW is tiny (75 for
// nin_gc's 5x5, 27 for resnet's 3x3) while the output is the LARGEST tensor of the net, so the layer is bound by writing y
// (forward) / reading gy (backward-weight) -- unless the contraction is done badly: generic implicit-GEMM tilings pad the
// 3 channels of every tap to a K-step of 4 and stage operands they barely reuse (this library's fp32 kernel: 196 us forward,
// 1078 us backward-weight on nin_gc L1 at batch 256; MIOpen: 341 / 393 us incl. its bias kernels and layout transposes).
// Here the whole im2col row k = (c, r, s) is the contraction index: ceil(K/4) steps of v_mfma_f32_16x16x4_f32 (exact fp32
// products, fp32 accumulate -- the arithmetic of the reference's F.conv2d), the image strip sits in LDS with its zero halo,
// and the OTHER operand streams through registers exactly once:
//   forward : weights are A fragments held in registers for the whole kernel (19 x MT floats per lane); the patch value
//             B[k][pixel] is one LDS read shared by the MT out-channel tiles; D[channel][pixel] leaves each lane with
//             float4 = 4 consecutive pixels per out-channel (256-B runs per 16 lanes) + bias.
//   wgrad   : gy streams as A[channel][pixel] (float4 per lane = 4 K-steps), the patch as B[pixel][k] from LDS;
//             D[channel][k] accumulates over the block's pixels; dbias falls out of the streamed gy; a second kernel reduces
//             the per-block partial tiles in a fixed order (fp64), deterministic.
// Geometry: groups 1, stride 1, dilation 1, "same" padding (Ho = H, Wo = W), W % 4 == 0, K <= 76 -- anything else stays on
// the generic kernels.
#include "qgemm_dev.h"

#include <stdlib.h>

#define C1_KS 19          // K-steps of 4: K <= 76
#define MN_MFMA_F32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

struct C1Params {
    const float* x;       // [N][C][H][W]
    const float* wp;      // packed weights [KS*4][Opad]  (fwd)
    const float* bias;
    float* y;             // fwd out [N][O][H][W]
    int relu;             // fwd: y = relu(conv + bias) (the block's ReLU behind a BN-fused IAO first layer)
    float* mm;            // fwd, nullable: per-block (min, max) of what is stored: mm[block], mm[grid + block]
    const float* gy;      // wgrad in
    float* part;          // wgrad partials [Z][Opad][80]
    float* dbpart;        // [Z][Opad]
    int N, C, H, W, O, KH, KW, ph, pw, K, KS, Opad;
    int R, strips, PR, PW, CS;      // strip of R output rows; LDS patch rows / row pitch / channel stride
    int Z, want_db, xs_bytes;       // xs_bytes: LDS bytes of the image patch (the backward-weight's row images follow)
    FastDiv fd_w;
    // BN variant of the backward-weight: gy is not given, it is the BatchNorm+sign backward of (da, yb) formed in registers
    const float* da;      // d loss / d sign output      [N][O][H][W]
    const float* yb;      // the conv output the BatchNorm saw
    const float* save;    // [2][O] mean, invstd
    const float* gamma;
    const float* beta;
    const float* sums;    // [2][O] sum dz, sum dz*zhat
    int training;
    float n_f;
    // BN = 2: the DoReFa block (BatchNorm2d + ReLU + the next conv's activation quantizer, qact_kernels.hip): da = dq, chan = its [9][O] constants
    const float* chan;
    int quant;            // dq is the gradient w.r.t. the QUANTISED activation (the clip-STE is applied here)
    float qs;             // quantizer scale 1 / (2^a - 1)
    float qs_inv;         // RN(1 / qs) (qa_dz_m); 0: IEEE division
    int interval;         // BN 2: the block's masks as one interval of y per channel (A/B knob MN_QA_NO_INTERVAL)
    // fused first block (k_c1b_fwd<MT, EPI 1 / 2>, k_c1_wgrad<MT, 3, 1>): the forward applies the BatchNorm (statistics known from the Gram data of x) and the
    // activation in its epilogue and writes CODES (1 byte) + the backward's masks (1 byte per 4 pixels) instead of y (4 bytes)
    const float* bn_save; // [2][O] mean, invstd
    const float* bn_gamma;
    const float* bn_beta;
    void* codes;          // EPI 1: int8 sign codes; EPI 2: uint8 codes of the next conv's a-bit quantizer
    uint8_t* mask4;       // [N][O][H W / 4]: low nibble = pass bits of 4 consecutive pixels (EPI 1: |z| < 1; EPI 2: z > 0), high nibble (EPI 2): ... and the clamp test
    int mask_shift;       // backward: which nibble (4: the gradient is w.r.t. the QUANTISED activation)
};

// stage the image strip (with zero halo) of image n, rows [row0 - ph, row0 + R + KH - 1 - ph) into xs[c][prow][pcol]
__device__ __forceinline__ void c1_stage(const C1Params& p, float* xs, int n, int row0) {
    const int per_c = p.PR * p.PW, total = p.C * per_c;
    for (int i = threadIdx.x; i < total; i += 256) {
        const int c = i / per_c, rem = i - c * per_c;
        const int pr = rem / p.PW, pc = rem - pr * p.PW;
        const int ir = row0 - p.ph + pr, ic = pc - p.pw;
        float v = 0.f;
        if (ir >= 0 && ir < p.H && ic >= 0 && ic < p.W) v = p.x[(((int64_t)n * p.C + c) * p.H + ir) * p.W + ic];
        xs[c * p.CS + pr * p.PW + pc] = v;
    }
}
// LDS offset of im2col row k = (c, r, s) relative to the top-left tap of a pixel
__device__ __forceinline__ int c1_koff(const C1Params& p, int k) {
    if (k >= p.K) return 0;          // padded rows: their weights / accumulators are never used
    const int T = p.KH * p.KW;
    const int c = k / T, t = k - c * T, r = t / p.KW, s = t - r * p.KW;
    return c * p.CS + r * p.PW + s;
}

// forward: block = (image, strip of R rows, 64*MT out-channels per wave x 4 waves)
template <int MT>
__global__ __launch_bounds__(256, 2) void k_c1_fwd(const C1Params p) {
    HIP_DYNAMIC_SHARED(float, smem)
    float* xs = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kq = lane >> 4;
    uint32_t b = blockIdx.x;
    const int strip = b % p.strips; b /= p.strips;
    const int n = b % p.N;
    const int cblk = b / p.N;
    const int row0 = strip * p.R;
    const int m0 = (cblk * 4 + wave) * 16 * MT;              // first out-channel of this wave

    c1_stage(p, xs, n, row0);
    // A fragments: wa[s][t] = w[m0 + t*16 + j][k = 4s + kq]
    float wa[C1_KS][MT];
    int koff[C1_KS];
#pragma unroll
    for (int s = 0; s < C1_KS; ++s) {
        koff[s] = c1_koff(p, 4 * s + kq);
#pragma unroll
        for (int t = 0; t < MT; ++t) wa[s][t] = (s < p.KS) ? p.wp[(int64_t)(4 * s + kq) * p.Opad + m0 + t * 16 + j] : 0.f;
    }
    __syncthreads();

    float lo = INFINITY, hi = -INFINITY;
    int mnan = 0;
    const int npix = p.R * p.W, nchunks = (npix + 63) >> 6;
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int pix = chunk * 64 + 4 * j;
        const bool pv = pix < npix;
        const uint32_t prow = fd_div(pv ? pix : 0, p.fd_w);
        const int pcol = (pv ? pix : 0) - prow * p.W;
        const int pb = (int)prow * p.PW + pcol;               // top-left tap of pixel (4j + 0)
        f32x4 acc[4][MT];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the patch values of step s + 1 are read from LDS before the MFMAs of step s are issued (padded steps read offset 0: their
        // weights are zero), so the matrix pipe never waits for an LDS round trip
        float bn[4];
        {
            const float* src = xs + pb + koff[0];
            bn[0] = src[0]; bn[1] = src[1]; bn[2] = src[2]; bn[3] = src[3];
        }
#pragma unroll
        for (int s = 0; s < C1_KS; ++s) {
            const float b0 = bn[0], b1 = bn[1], b2 = bn[2], b3 = bn[3];
            if (s + 1 < C1_KS) {
                const float* src = xs + pb + koff[s + 1];
                bn[0] = src[0]; bn[1] = src[1]; bn[2] = src[2]; bn[3] = src[3];
            }
            if (s < p.KS) {
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    acc[0][t] = MN_MFMA_F32(wa[s][t], b0, acc[0][t]);
                    acc[1][t] = MN_MFMA_F32(wa[s][t], b1, acc[1][t]);
                    acc[2][t] = MN_MFMA_F32(wa[s][t], b2, acc[2][t]);
                    acc[3][t] = MN_MFMA_F32(wa[s][t], b3, acc[3][t]);
                }
            }
            MN_SCHED_FENCE();
        }
        // D[row = channel 4kq + r][col = pixel j]: across q a float4 of 4 consecutive pixels per channel
        if (pv) {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + t * 16 + kq * 4 + r;
                    if (m < p.O) {
                        const float bb = p.bias ? p.bias[m] : 0.f;
                        float* dst = p.y + (((int64_t)n * p.O + m) * p.H + row0 + (int)prow) * p.W + pcol;
                        float4 v = make_float4(acc[0][t][r] + bb, acc[1][t][r] + bb, acc[2][t][r] + bb, acc[3][t][r] + bb);
                        if (p.relu) { v.x = qa_relu(v.x); v.y = qa_relu(v.y); v.z = qa_relu(v.z); v.w = qa_relu(v.w); }
                        if (p.mm) {          // plain min / max (NaN-ignoring) + a NaN flag: torch.min / max propagate a NaN
                            lo = fminf(lo, fminf(fminf(v.x, v.y), fminf(v.z, v.w)));
                            hi = fmaxf(hi, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
                            mnan |= (int)((v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w));
                        }
                        *reinterpret_cast<float4*>(dst) = v;
                    }
                }
        }
    }
    if (p.mm) {          // (block-uniform) the patch image is dead: its first words serve as reduction scratch
        if (mnan) lo = hi = NAN;
        lo = block_reduce(lo, OpMinF(), INFINITY, xs);
        hi = block_reduce(hi, OpMaxF(), -INFINITY, xs);
        if (tid == 0) { p.mm[blockIdx.x] = lo; p.mm[gridDim.x + blockIdx.x] = hi; }
    }
}

// Forward on the bf16 matrix cores (round 4).  v_mfma_f32_16x16x4_f32 made the kernel above MFMA-bound (5 GMAC of fp32 MFMA: 64 us at the fp32 peak, 123 us measured,
// against 55 us for writing y).  Both operands are real, so each is written as three exact bf16 terms and a product uses the six largest term products (2^-24
// relative: the accuracy of an fp32 multiply) on v_mfma_f32_16x16x32_bf16 -- 6 / 16 of the fp32-MFMA time.  The B operand needs 8 consecutive K per lane, which
// the channel / tap-strided patch cannot give, so every 64-pixel chunk is expanded ONCE per block into an im2col tile [term][pixel][96 k] in LDS (24 consecutive k
// of one pixel per thread: patch reads conflict-free along the pixels, three 16-byte writes per term) that the four waves -- each 64 out-channels -- share; the
// weights are A fragments in registers for the block's lifetime (3 terms x 3 K steps x MT).  Pixel p sits in tile row (p & 3) * 16 + (p >> 2): MFMA column j of
// n-tile q is pixel 4 j + q, so a lane ends with float4 = 4 consecutive pixels per out-channel (the write side takes the bank conflicts of that permutation: 9
// writes against 36 fragment reads per chunk).
#define C1B_KP 96            // padded K (3 K steps of 32)
#define C1B_LD 104           // u16 per im2col row: 96 + 8 pad (208-byte rows: the 16 rows of a fragment read cover all banks)
template <int MT, int EPI = 0>
__global__ __launch_bounds__(256, 2) void k_c1b_fwd(const C1Params p) {
    HIP_DYNAMIC_SHARED(float, smem)
    float* xs = smem;
    uint16_t* im = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(smem) + p.xs_bytes);          // [3 terms][64][C1B_LD]
    int* ktab = reinterpret_cast<int*>(im + 3 * 64 * C1B_LD);                                      // [96] patch offset of im2col row k (-1: padding)
    const int tid = threadIdx.x, lane = tid & 63, wave = mn_uniform(tid >> 6), j = lane & 15, kg = lane >> 4;
    uint32_t b = blockIdx.x;
    const int strip = b % p.strips; b /= p.strips;
    const int n = b % p.N;
    const int cblk = b / p.N;
    const int row0 = strip * p.R;
    const int m0 = (cblk * 4 + wave) * 16 * MT;

    float* bs = reinterpret_cast<float*>(ktab + C1B_KP);                                           // [64 MT] the block's bias
    c1_stage(p, xs, n, row0);
    for (int k = tid; k < C1B_KP; k += 256) ktab[k] = k < p.K ? c1_koff(p, k) : -1;
    for (int i = tid; i < 64 * MT; i += 256) { const int m = cblk * 64 * MT + i; bs[i] = (p.bias && m < p.O) ? p.bias[m] : 0.f; }
    float4* bc = reinterpret_cast<float4*>(bs + 64 * MT);          // EPI: [64 MT] mean, invstd, gamma, beta
    if (EPI)
        for (int i = tid; i < 64 * MT; i += 256) {
            const int m = cblk * 64 * MT + i, mc = m < p.O ? m : p.O - 1;
            bc[i] = make_float4(p.bn_save[mc], p.bn_save[p.O + mc], p.bn_gamma[mc], p.bn_beta[mc]);
        }
    // A fragments: three exact bf16 terms of w[m0 + t * 16 + j][ks * 32 + kg * 8 .. + 7].  The block's weight rows are one contiguous range: they come through LDS
    // (the tile buffer, not yet in use: 16 MT rows x K <= 64 x 96 floats fit) one wave's rows per round, read from memory coalesced -- a lane fetching its 24 MT
    // elements itself touches a different 300-byte row per lane and element (measured on nin_gc's first layer: ~20 us of a 113 us kernel, all 512 blocks at once).
    u32x4 wa[3][3][MT];
    float* wl = reinterpret_cast<float*>(im);
    for (int r = 0; r < 4; ++r) {
        if (r) __syncthreads();          // the previous round's rows are consumed
        const int row_r = (cblk * 4 + r) * 16 * MT;
        const int nrows = p.O - row_r < 16 * MT ? p.O - row_r : 16 * MT;
        const int cnt = nrows > 0 ? nrows * p.K : 0;
        const float* src = p.wp + (int64_t)row_r * p.K;
        for (int i = tid; i < cnt; i += 256) wl[i] = src[i];
        __syncthreads();
        if (wave == r) {
#pragma unroll
            for (int ks = 0; ks < 3; ++ks)
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const int lrow = t * 16 + j;
                    const bool mv = row_r + lrow < p.O;
                    float t0[8], t1[8], t2[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int k = ks * 32 + kg * 8 + e;
                        const float v = (mv && k < p.K) ? wl[lrow * p.K + k] : 0.f;
                        t0[e] = mn_bf16_head(v);
                        const float r1 = v - t0[e];
                        t1[e] = mn_bf16_head(r1);
                        t2[e] = r1 - t1[e];
                    }
                    wa[0][ks][t] = u32x4{mn_pack_bf16x2(t0[0], t0[1]), mn_pack_bf16x2(t0[2], t0[3]), mn_pack_bf16x2(t0[4], t0[5]), mn_pack_bf16x2(t0[6], t0[7])};
                    wa[1][ks][t] = u32x4{mn_pack_bf16x2(t1[0], t1[1]), mn_pack_bf16x2(t1[2], t1[3]), mn_pack_bf16x2(t1[4], t1[5]), mn_pack_bf16x2(t1[6], t1[7])};
                    wa[2][ks][t] = u32x4{mn_pack_bf16x2(t2[0], t2[1]), mn_pack_bf16x2(t2[2], t2[3]), mn_pack_bf16x2(t2[4], t2[5]), mn_pack_bf16x2(t2[6], t2[7])};
                }
        }
    }
    __syncthreads();
    // (this wave expands k = 24 wave .. 24 wave + 23 of every pixel; keeping the 24 wave-uniform patch offsets in scalar registers instead of re-reading the LDS table
    //  spilled 59 SGPRs into an already full vector register file: the table read stays)

    float lo = INFINITY, hi = -INFINITY;
    int mnan = 0;
    const int npix = p.R * p.W, nchunks = (npix + 63) >> 6;
    const int irow = (lane & 3) * 16 + (lane >> 2);          // im2col tile row of the pixel this thread expands (pixel `lane` of the chunk)
    constexpr int TB = 3 * 64 * C1B_LD;                      // u16 per tile buffer
    auto expand = [&](int chunk, int buf) {
        const int pix = chunk * 64 + lane;
        const bool pv = pix < npix;
        const uint32_t prow = fd_div(pv ? pix : 0, p.fd_w);
        const int pcol = (pv ? pix : 0) - prow * p.W;
        const float* src = xs + (int)prow * p.PW + pcol;
        uint16_t* dst = im + buf * TB + irow * C1B_LD + wave * 24;
#pragma unroll
        for (int o8 = 0; o8 < 3; ++o8) {
            float t0[8], t1[8], t2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int kk = ktab[wave * 24 + o8 * 8 + e];
                const float v = (pv && kk >= 0) ? src[kk < 0 ? 0 : kk] : 0.f;
                t0[e] = mn_bf16_head(v);
                const float r1 = v - t0[e];
                t1[e] = mn_bf16_head(r1);
                t2[e] = r1 - t1[e];
            }
            *reinterpret_cast<u32x4*>(dst + o8 * 8) = u32x4{mn_pack_bf16x2(t0[0], t0[1]), mn_pack_bf16x2(t0[2], t0[3]), mn_pack_bf16x2(t0[4], t0[5]), mn_pack_bf16x2(t0[6], t0[7])};
            *reinterpret_cast<u32x4*>(dst + 64 * C1B_LD + o8 * 8) = u32x4{mn_pack_bf16x2(t1[0], t1[1]), mn_pack_bf16x2(t1[2], t1[3]), mn_pack_bf16x2(t1[4], t1[5]), mn_pack_bf16x2(t1[6], t1[7])};
            *reinterpret_cast<u32x4*>(dst + 128 * C1B_LD + o8 * 8) = u32x4{mn_pack_bf16x2(t2[0], t2[1]), mn_pack_bf16x2(t2[2], t2[3]), mn_pack_bf16x2(t2[4], t2[5]), mn_pack_bf16x2(t2[6], t2[7])};
        }
    };
    // (one tile buffer, two blocks per CU: the other block's MFMAs fill this block's expand phase.  A double-buffered tile with one block per CU -- the next
    //  chunk expanded in program order before the current chunk's MFMAs -- measured 156 us against 113 us)
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        const int buf = 0;
        if (chunk) __syncthreads();          // the tile is consumed
        expand(chunk, 0);
        __syncthreads();
        f32x4 acc[4][MT];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const uint16_t* tile = im + buf * TB;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint16_t* br = tile + (q * 16 + j) * C1B_LD + ks * 32 + kg * 8;
                const u32x4 b0 = *reinterpret_cast<const u32x4*>(br), b1 = *reinterpret_cast<const u32x4*>(br + 64 * C1B_LD), b2 = *reinterpret_cast<const u32x4*>(br + 128 * C1B_LD);
                // the six term products, smallest first; MT independent accumulators between two MFMAs on the same one
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[q][t] = mn_mfma_bf16(wa[0][ks][t], b2, acc[q][t]);
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[q][t] = mn_mfma_bf16(wa[2][ks][t], b0, acc[q][t]);
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[q][t] = mn_mfma_bf16(wa[1][ks][t], b1, acc[q][t]);
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[q][t] = mn_mfma_bf16(wa[0][ks][t], b1, acc[q][t]);
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[q][t] = mn_mfma_bf16(wa[1][ks][t], b0, acc[q][t]);
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[q][t] = mn_mfma_bf16(wa[0][ks][t], b0, acc[q][t]);
            }
        // D[row = channel 4 kg + r][col j of n-tile q = pixel 4 j + q]: a float4 of 4 consecutive pixels per channel
        const int pix = chunk * 64 + 4 * j;
        // (opaque per chunk: the 16 MT per-channel constants of the epilogue -- bias, and the BatchNorm's four -- are re-read from LDS; hoisted out of the chunk loop
        //  they cost 80 registers the kernel does not have: 40 scratch reloads per chunk in the first fused build)
        const int ch0 = (int)mn_opaque((uint32_t)(wave * 16 * MT));
        if (pix < npix) {
            const uint32_t prow = fd_div(pix, p.fd_w);
            const int pcol = pix - prow * p.W;
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + t * 16 + kg * 4 + r;
                    if (m < p.O) {
                        const float bb = bs[ch0 + t * 16 + kg * 4 + r];
                        float* dst = p.y + (((int64_t)n * p.O + m) * p.H + row0 + (int)prow) * p.W + pcol;
                        float4 v = make_float4(acc[0][t][r] + bb, acc[1][t][r] + bb, acc[2][t][r] + bb, acc[3][t][r] + bb);
                        if (EPI) {
                            // BatchNorm + activation on the accumulators, expression for expression k_bns_apply<0, 1> / k_qa_fwd<1, 0, 0>: codes + pass bits leave, y does not
                            const float4 c = bc[ch0 + t * 16 + kg * 4 + r];
                            const float yv[4] = {v.x, v.y, v.z, v.w};
                            uint32_t code = 0u, mk = 0u;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float zh = (yv[q] - c.x) * c.y;
                                const float z = zh * c.z + c.w;
                                if (EPI == 1) {
                                    code |= (z < 0.f ? 0xFFu : 0x01u) << (8 * q);
                                    mk |= (fabsf(z) < 1.f) ? (1u << q) : 0u;          // (-1 < z && z < 1, NaN included, as ONE compare with a source modifier)
                                } else {
                                    const float a = qa_relu(z);
                                    code |= qa_code(a, p.qs) << (8 * q);
                                    const float tq = a * 0.1f;
                                    const bool pos = z > 0.f;
                                    mk |= (pos ? (1u << q) : 0u) | ((pos && tq >= 0.f && tq <= 1.f) ? (16u << q) : 0u);
                                }
                            }
                            const int64_t e = (((int64_t)n * p.O + m) * p.H + row0 + (int)prow) * p.W + pcol;
                            *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(p.codes) + e) = code;
                            p.mask4[e >> 2] = (uint8_t)mk;
                            continue;
                        }
                        if (p.relu) { v.x = qa_relu(v.x); v.y = qa_relu(v.y); v.z = qa_relu(v.z); v.w = qa_relu(v.w); }
                        if (p.mm) {
                            lo = fminf(lo, fminf(fminf(v.x, v.y), fminf(v.z, v.w)));
                            hi = fmaxf(hi, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
                            mnan |= (int)((v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w));
                        }
                        *reinterpret_cast<float4*>(dst) = v;
                    }
                }
        }
    }
    if (p.mm) {
        if (mnan) lo = hi = NAN;
        lo = block_reduce(lo, OpMinF(), INFINITY, xs);
        hi = block_reduce(hi, OpMaxF(), -INFINITY, xs);
        if (tid == 0) { p.mm[blockIdx.x] = lo; p.mm[gridDim.x + blockIdx.x] = hi; }
    }
}

// backward-weight: block z accumulates dw over its share of (image, strip) tiles for the 64*MT... out-channels of its channel block.
// wave w owns out-channels [m0, m0 + 16*MT); the five 16-wide tiles of the k axis cover K <= 80.
// BN = 1: the layer is followed by BatchNorm2d + BinaryActivation and nothing else consumes d loss / d y (the first layer has no
// backward-data): dy = gamma*invstd*(dz - sum_dz/n - zhat*sum_dzzhat/n) with dz = da*[|z| < 1] is formed from (da, y) while they
// stream in -- expression for expression what k_bns_apply<1> computes -- so the full-size dy tensor is never written or re-read.
// The gy rows reach the MFMA through a wave-private LDS image: lane (row j, k-group kq) of an A fragment wants 4 pixels of channel
// row j, i.e. a fragment-direct load touches 16 different cache lines per 16 lanes and every line twice (measured: 2x the designed
// HBM traffic, L1 tag rate bound).  Instead 8 lanes read one 128-byte line (32 pixels) of a row, the BatchNorm fold is applied on
// those registers, the step is written to the wave's [16 MT rows][32 pixels] image (rows padded to 160 B: conflict-free b128
// fragment reads) and read back as fragments -- no block barrier inside a tile, the next step's loads fly during the MFMAs.
#define C1_RSA 160
// DZ (with BN): the rows are dz alone (the clip-STE / ReLU masks applied, NOT the BatchNorm backward) and im2col column K is a column of ones, so that D[o][K] =
// sum dz: the one-pass backward of the first block (k_c1_bn_final below finishes it from the Gram data of x).
template <int MT, int BN, int DZ = 0>
__global__ __launch_bounds__(256, 2) void k_c1_wgrad(const C1Params p) {
    HIP_DYNAMIC_SHARED(float, smem)
    float* xs = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kq = lane >> 4;
    unsigned char* gsm = reinterpret_cast<unsigned char*>(smem) + p.xs_bytes + wave * (16 * MT * (C1_RSA + 32));
    float* ctab = reinterpret_cast<float*>(gsm + 16 * MT * C1_RSA);       // BN: [16 MT rows][8] = mean, invstd, gamma, beta, k1, k2 (kept out of the register file)
    uint32_t b = blockIdx.x;
    const int z = b % p.Z;
    const int cblk = b / p.Z;
    const int m0 = (cblk * 4 + wave) * 16 * MT;
    int koff[5];
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) koff[nt] = c1_koff(p, nt * 16 + j);     // B[pixel][k = nt*16 + j]
    const int one_nt = (DZ && (p.K & 15) == j) ? (p.K >> 4) : -1;          // DZ: this lane's column of n-tile one_nt is the column of ones

    // staging roles: row sr + 8 i of the wave's 16 MT rows, pixels 4 sq .. 4 sq + 3 of the 32-pixel step
    constexpr int NR = 2 * MT;
    const int sr = lane >> 3, sq = lane & 7;
    uint32_t roff[NR];          // element offset of the lane's quad inside image 0 / strip 0 (planner: the tensor has < 2^31 elements)
    float dbs[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int m = m0 + sr + 8 * i;
        const int mc = m < p.O ? m : p.O - 1;              // rows beyond O: clamped, their dw rows are never read
        roff[i] = (uint32_t)mc * (uint32_t)(p.H * p.W) + 4u * sq;
        dbs[i] = 0.f;
    }
    if (BN == 1 || BN == 2) {
        for (int r = lane; r < 16 * MT; r += 64) {
            const int m = m0 + r;
            const int mc = m < p.O ? m : p.O - 1;
            if (BN == 2) {        // QaCh rows 2 .. 5: mean, invstd, gamma, beta; row 8: gi (slot 6)
                ctab[r * 8 + 0] = p.chan[2 * p.O + mc]; ctab[r * 8 + 1] = p.chan[3 * p.O + mc]; ctab[r * 8 + 2] = p.chan[4 * p.O + mc]; ctab[r * 8 + 3] = p.chan[5 * p.O + mc];
            } else {
                ctab[r * 8 + 0] = p.save[mc]; ctab[r * 8 + 1] = p.save[p.O + mc]; ctab[r * 8 + 2] = p.gamma[mc]; ctab[r * 8 + 3] = p.beta[mc];
            }
            ctab[r * 8 + 4] = p.training ? p.sums[mc] / p.n_f : 0.f;
            ctab[r * 8 + 5] = p.training ? p.sums[p.O + mc] / p.n_f : 0.f;
            ctab[r * 8 + 6] = BN == 2 ? p.chan[8 * p.O + mc] : 0.f; ctab[r * 8 + 7] = 0.f;
            if (BN == 1 && DZ && p.interval) {          // the sign's clip-STE |z| < 1 as one interval of y per channel (as the DoReFa block below)
                const float mean = ctab[r * 8 + 0], invstd = ctab[r * 8 + 1], ga = ctab[r * 8 + 2], be = ctab[r * 8 + 3];
                if (fabsf(mean) <= 1.0e9f && fabsf(invstd) <= 1.0e9f && fabsf(ga) <= 1.0e9f && fabsf(be) <= 1.0e9f && ga != 0.f && invstd > 0.f) {
                    const QaInterval iv = qa_mask_interval(0x7f7fffff, [&](float y) { return ((y - mean) * invstd) * ga + be; }, [](int32_t k) { return mn_keyf(k); }, 1, true);
                    ctab[r * 8 + 2] = iv.lo; ctab[r * 8 + 3] = iv.hi; ctab[r * 8 + 7] = 1.f;
                }
            }
            if (BN == 2 && p.interval) {
                // the ReLU mask and the quantizer's clamp test as ONE interval of y per channel (qa_mask_interval, common.h): the per-element z, relu, 0.1 a and their
                // selects (10 of ~21 VALU instructions per element of this VALU-bound fold) become two compares.  A channel whose constants are not finite (or
                // gamma == 0: z is constant, an overflowing zhat would make it NaN) keeps the element-wise form (slot 7 = 0).
                const float mean = ctab[r * 8 + 0], invstd = ctab[r * 8 + 1], ga = ctab[r * 8 + 2], be = ctab[r * 8 + 3];
                if (fabsf(mean) <= 1.0e9f && fabsf(invstd) <= 1.0e9f && fabsf(ga) <= 1.0e9f && fabsf(be) <= 1.0e9f && ga != 0.f && invstd > 0.f) {
                    const QaInterval iv = qa_mask_interval(0x7f7fffff, [&](float y) { return ((y - mean) * invstd) * ga + be; }, [](int32_t k) { return mn_keyf(k); }, p.quant);
                    ctab[r * 8 + 2] = iv.lo; ctab[r * 8 + 3] = iv.hi; ctab[r * 8 + 7] = 1.f;
                }
            }
        }
        MN_WAVE_SYNC();
    }
    f32x4 acc[MT][5];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int nt = 0; nt < 5; ++nt) acc[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ntiles = p.N * p.strips, npix = p.R * p.W, nsteps = npix >> 5;     // 32 pixels per step (R*W % 32 == 0)
    for (int tile = z; tile < ntiles; tile += p.Z) {
        const int n = tile / p.strips, strip = tile - n * p.strips, row0 = strip * p.R;
        __syncthreads();                      // previous tile's patch fully consumed
        c1_stage(p, xs, n, row0);
        __syncthreads();
        const uint32_t tbase = (uint32_t)n * (uint32_t)(p.O * p.H * p.W) + (uint32_t)(row0 * p.W);      // uniform; the strip's pixels are contiguous in a plane
        float4 ra[NR], rb[NR];
        auto fetch = [&](int st) {            // unconditional: the caller clamps st
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const uint32_t off = tbase + roff[i] + (uint32_t)st * 32u;
                if (BN == 3) { ra[i] = *reinterpret_cast<const float4*>(p.da + off); rb[i].x = mn_u2f((uint32_t)p.mask4[off >> 2]); }
                else if (BN) { ra[i] = *reinterpret_cast<const float4*>(p.da + off); rb[i] = *reinterpret_cast<const float4*>(p.yb + off); }
                else ra[i] = *reinterpret_cast<const float4*>(p.gy + off);
            }
        };
        fetch(0);
        for (int st = 0; st < nsteps; ++st) {
            // registers -> LDS image (BatchNorm + sign backward applied on the way: expression for expression k_bns_apply<1>)
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                float r[4] = {ra[i].x, ra[i].y, ra[i].z, ra[i].w};
                if (BN == 3) {          // the forward's pass bits: dz = the gradient where the bit is set
                    const uint32_t mk = mn_f2u(rb[i].x) >> p.mask_shift;
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[e] = (mk >> e) & 1u ? r[e] : 0.f;
                } else if (BN) {
                    const float yv[4] = {rb[i].x, rb[i].y, rb[i].z, rb[i].w};
                    const float4 c0 = *reinterpret_cast<const float4*>(ctab + (sr + 8 * i) * 8);        // mean, invstd, gamma, beta
                    const float2 c1 = *reinterpret_cast<const float2*>(ctab + (sr + 8 * i) * 8 + 4);    // k1, k2
                    const float cgi_ = BN == 2 ? ctab[(sr + 8 * i) * 8 + 6] : c0.z * c0.y;
                    const bool ivl = (BN == 2 || DZ) && ctab[(sr + 8 * i) * 8 + 7] != 0.f;          // this row's masks are an interval of y: c0.z = lo, c0.w = hi
                    if (DZ && ivl) {
                        // one-pass backward: dz = the masked gradient; the quantizer's STE factor ((g s) / s) * 0.1 is applied as 0.1 by k_c1_bn_final (linear; the
                        // s / s round trip is dropped: <= 2^-23 relative per element) -- 4 VALU instructions per element instead of ~20 in a kernel that was VALU-bound
#pragma unroll
                        for (int e = 0; e < 4; ++e) r[e] = (yv[e] >= c0.z && yv[e] <= c0.w) ? r[e] : 0.f;
                    } else
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float zh = (yv[e] - c0.x) * c0.y;
                        float dz;
                        if (ivl) {
                            const float d = p.quant ? dorefa_ste_core_m(r[e], p.qs, p.qs_inv) : r[e];
                            dz = (yv[e] >= c0.z && yv[e] <= c0.w) ? d : 0.f;
                        } else {
                            const float zz = zh * c0.z + c0.w;
                            if (BN == 2) dz = qa_dz_m(r[e], qa_relu(zz), zz, p.qs, p.qs_inv, p.quant);     // expression for expression k_qa_apply<1, 0>
                            else dz = (zz > -1.f && zz < 1.f) ? r[e] : 0.f;
                        }
                        r[e] = DZ ? ((BN == 2 && p.quant) ? dz * 10.f : dz) : cgi_ * (dz - c1.x - zh * c1.y);          // (DZ: a row without interval, k_c1_bn_final scales by 0.1)
                    }
                }
                if (!DZ) dbs[i] += (r[0] + r[1]) + (r[2] + r[3]);
                *reinterpret_cast<float4*>(gsm + (sr + 8 * i) * C1_RSA + 16 * sq) = make_float4(r[0], r[1], r[2], r[3]);
            }
            fetch(st + 1 < nsteps ? st + 1 : st);
            MN_WAVE_SYNC();
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                // A[i = channel j][k = kq]: float4 = pixels st*32 + half*16 + 4kq + e, e = MFMA step
                float4 ga[MT];
#pragma unroll
                for (int t = 0; t < MT; ++t) ga[t] = *reinterpret_cast<const float4*>(gsm + (t * 16 + j) * C1_RSA + half * 64 + 16 * kq);
                const int pix = st * 32 + half * 16 + 4 * kq;
                const uint32_t prow = fd_div(pix, p.fd_w);
                const int pb = (int)prow * p.PW + (pix - (int)prow * p.W);
#pragma unroll
                for (int nt = 0; nt < 5; ++nt) {
                    const float* src = xs + pb + koff[nt];
                    float b0 = src[0], b1 = src[1], b2 = src[2], b3 = src[3];
                    if (DZ && nt == one_nt) b0 = b1 = b2 = b3 = 1.f;
#pragma unroll
                    for (int t = 0; t < MT; ++t) {
                        acc[t][nt] = MN_MFMA_F32(ga[t].x, b0, acc[t][nt]);
                        acc[t][nt] = MN_MFMA_F32(ga[t].y, b1, acc[t][nt]);
                        acc[t][nt] = MN_MFMA_F32(ga[t].z, b2, acc[t][nt]);
                        acc[t][nt] = MN_MFMA_F32(ga[t].w, b3, acc[t][nt]);
                    }
                }
            }
            MN_WAVE_SYNC();                   // the image is rewritten by the next step
        }
    }
    // D[row = channel 4kq + r][col = k = nt*16 + j]
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int nt = 0; nt < 5; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + t * 16 + kq * 4 + r;
                p.part[((int64_t)z * p.Opad + m) * 80 + nt * 16 + j] = acc[t][nt][r];
            }
    if (!DZ && p.want_db) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            float v = dbs[i];
            v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);      // the eight pixel chunks of the row
            if (sq == 0) p.dbpart[(int64_t)z * p.Opad + m0 + sr + 8 * i] = v;
        }
    }
}
// wp[k][Opad] = w[m][k] (k = (c, r, s) in OIHW order), zero padded
__global__ __launch_bounds__(256) void k_c1_pack(const float* __restrict__ w, float* __restrict__ wp, int O, int K, int Opad, int Kp) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < Kp * Opad; i += gridDim.x * 256) {
        const int k = i / Opad, m = i - k * Opad;
        wp[i] = (k < K && m < O) ? w[(int64_t)m * K + k] : 0.f;
    }
}
// fixed-order fp64 reduction of the Z partial tiles: dw[m][k], dbias[m].  A block owns 64 consecutive outputs (coalesced 256-B
// rows of every partial tile); its four waves sum z = w, w+4, ... and the four sums are combined in wave order through LDS --
// the same order on every run (deterministic).  Outputs are indexed over the padded [Opad][80] tile, then dbias.
__global__ __launch_bounds__(256) void k_c1_reduce(const float* __restrict__ part, const float* __restrict__ dbpart, float* __restrict__ dw, float* __restrict__ db,
                                                   int Z, int O, int K, int Opad) {
    __shared__ double red[4][64];
    const int og = threadIdx.x & 63, zg = threadIdx.x >> 6;
    const int64_t ntile = (int64_t)Opad * 80, total = ntile + (db ? Opad : 0);
    for (int64_t base = (int64_t)blockIdx.x * 64; base < total; base += (int64_t)gridDim.x * 64) {
        const int64_t i = base + og;
        double s = 0.0;
        // eight loads in flight, added in the order z = zg, zg + 4, ... as before (same sums bit for bit; one load at a time the loop is pure latency: 19.5 us for 15 k outputs)
        const float* src = i < ntile ? part + i : dbpart + (i - ntile);
        const int64_t zs = i < ntile ? ntile : (int64_t)Opad;
        if (i < total) {
            int z = zg;
            for (; z + 28 < Z; z += 32) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(z + 4 * u) * zs];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += (double)v[u];
            }
            for (; z + 12 < Z; z += 16) {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = src[(int64_t)(z + 4 * u) * zs];
#pragma unroll
                for (int u = 0; u < 4; ++u) s += (double)v[u];
            }
            for (; z < Z; z += 4) s += (double)src[(int64_t)z * zs];
        }
        __syncthreads();
        red[zg][og] = s;
        __syncthreads();
        if (zg == 0 && i < total) {
            const double v = ((red[0][og] + red[1][og]) + red[2][og]) + red[3][og];
            if (i < ntile) {
                const int m = (int)(i / 80), k = (int)(i - (int64_t)m * 80);
                if (m < O && k < K) dw[(int64_t)m * K + k] = (float)v;
            } else {
                const int m = (int)(i - ntile);
                if (m < O) db[m] = (float)v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ one-pass backward of the first block (round 5)
// The BatchNorm backward is LINEAR in dz = (masked) da: with f_k(pixel) the im2col row of x (f_K = 1), A[o][k] = sum dz[o] f_k, S1 = A[o][K], P[k] = sum f_k,
// G = sum f f^T and y[o] = w[o,:] . f + b[o]:
//     S2 = sum dz zhat  = invstd (w[o,:] . A[o,:] + (b - mean) S1)
//     B[o][k] = sum zhat f_k = invstd ((w G)[o][k] + (b - mean) P[k])
//     dw[o][k] = gamma invstd (A[o][k] - S1 P[k] / n - (S2 / n) B[o][k]),   dgamma = S2,   dbeta = S1
// so ONE pass over (da, y) -- k_c1_wgrad<.., DZ 1> -- replaces the sums pass (k_bns_partial<1> / k_qa_partial<1, 0>: 98 us on nin_gc at batch 256) AND the fold in
// the backward-weight's operand load; G, P (76 x 76 numbers) come from x alone (k_c1_xgram: the c3 block's idea, iao_bnfuse.hip, applied to the image).
// Conditioning (round 6): the variance is the cancelling form w (G - P P^T / n) w^T.  With G accumulated in fp32 per block its error is the fp32 error of G times
// mean(f)^2 / var(f) (un-normalised 0..255 images: large) times |w|^2 lambda_max / var_y (difference filters: large).  So the kernel accumulates the Gram data of
// SHIFTED features f_k - c[channel(k)], c = the mean of the first <= 256 pixels of each input-channel plane of image 0 (any constant near the mean does; every block
// computes the same one), which takes the mean out of the cancellation; k_c1_gram_unshift then rebuilds the raw G, P the consumers are written for in fp64:
// G = G' + c P'^T + P' c^T + n c c^T, P = P' + n c -- consistent with the centred data to fp64 rounding, so G - P P^T / n recovers it.
#define C1G_TILE (15 * 256)          // floats per partial: the 15 tile pairs ta <= tb of the 80 x 80 Gram matrix
__global__ __launch_bounds__(256) void k_c1_xgram(const C1Params p, float* __restrict__ gpart, float* __restrict__ shift) {
    HIP_DYNAMIC_SHARED(float, smem)
    float* xs = smem;
    float* red = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + p.xs_bytes);          // [4 waves][C1G_TILE]
    const int tid = threadIdx.x, lane = tid & 63, wave = mn_uniform(tid >> 6), j = lane & 15, kq = lane >> 4;
    int koff[5];
#pragma unroll
    for (int nt = 0; nt < 5; ++nt) koff[nt] = c1_koff(p, nt * 16 + j);
    const int one_nt = ((p.K & 15) == j) ? (p.K >> 4) : -1;
    // the feature K = 1: four floats of 1.0 behind the patch (inside its 64-byte pad, never staged over), addressed instead of the pixel's taps -- a select on the LDS
    // ADDRESS; selecting the loaded VALUE made every one of the 20 reads of a pixel group a branch of its own (28 us against 13 us of matrix work)
    const int ones_off = p.C * p.CS;
    if (tid < 4) xs[ones_off + tid] = 1.f;
    __shared__ float csh[80], cred[16];
    {
        const int hw = p.H * p.W, ns = hw < 256 ? hw : 256;
        for (int c = 0; c < p.C; ++c) {          // (C <= 76; CIFAR: 3)
            const float v = tid < ns ? p.x[(int64_t)c * hw + tid] : 0.f;
            const float sum = block_reduce(v, OpAddF(), 0.f, cred);
            if (tid == 0) { csh[c] = sum / (float)ns; if (blockIdx.x == 0) shift[c] = csh[c]; }
        }
        __syncthreads();
    }
    f32x4 acc[15];
#pragma unroll
    for (int i = 0; i < 15; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int ntiles = p.N * p.strips, ngroups = (p.R * p.W) >> 4;          // 16 pixels per group (R W % 32 == 0)
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n = tile / p.strips, strip = tile - n * p.strips, row0 = strip * p.R;
        __syncthreads();
        c1_stage(p, xs, n, row0);
        __syncthreads();
        for (int i = tid; i < p.C * p.CS; i += 256) xs[i] -= csh[i / p.CS];          // the shifted features (the zero padding becomes -c: a shift of EVERY feature value)
        __syncthreads();
        for (int gi = wave; gi < ngroups; gi += 4) {
            // MFMA step e contracts the pixels 4 kq + e: the lane's value of feature tile t is at once A[i = j][kq] and B[kq][j]
            const int pix = gi * 16 + 4 * kq;
            const uint32_t prow = fd_div(pix, p.fd_w);
            const int pb = (int)prow * p.PW + (pix - (int)prow * p.W);
            float v[5][4];
#pragma unroll
            for (int nt = 0; nt < 5; ++nt) {
                const float* src = xs + ((nt == one_nt) ? ones_off : pb + koff[nt]);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[nt][e] = src[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {          // (15 independent accumulators between two MFMAs on the same one)
                int pr = 0;
#pragma unroll
                for (int ta = 0; ta < 5; ++ta)
#pragma unroll
                    for (int tb = ta; tb < 5; ++tb) {
                        acc[pr] = MN_MFMA_F32(v[ta][e], v[tb][e], acc[pr]);
                        ++pr;
                    }
            }
        }
    }
    // the four waves' tiles summed in wave order: D[row 4 kq + r][col j]
#pragma unroll
    for (int pr = 0; pr < 15; ++pr)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave * C1G_TILE + pr * 256 + (4 * kq + r) * 16 + j] = acc[pr][r];
    __syncthreads();
    for (int i = tid; i < C1G_TILE; i += 256)
        gpart[(int64_t)blockIdx.x * C1G_TILE + i] = ((red[i] + red[C1G_TILE + i]) + red[2 * C1G_TILE + i]) + red[3 * C1G_TILE + i];
}
// gram[a][b] (fp64, 80 x 80; row / column K = the feature sums P, gram[K][K] = n): fixed-order sum of the Z partials, both triangles written
#define C1G_ZG 16          // z-groups of the final sums: 16 waves per block, each partial row read by one of them (4 groups: 8.1 us for 512 partials, pure latency)
__global__ __launch_bounds__(64 * C1G_ZG) void k_c1_xgram_final(const float* __restrict__ gpart, int Z, double* __restrict__ gram) {
    __shared__ double red[C1G_ZG][64];
    const int og = threadIdx.x & 63, zg = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + og;
    double s = 0.0;
    if (i < C1G_TILE) {
        const float* src = gpart + i;
        int z = zg;
        for (; z + 7 * C1G_ZG < Z; z += 8 * C1G_ZG) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(z + C1G_ZG * u) * C1G_TILE];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (double)v[u];
        }
        for (; z < Z; z += C1G_ZG) s += (double)src[(int64_t)z * C1G_TILE];
    }
    red[zg][og] = s;
    __syncthreads();
    if (zg == 0 && i < C1G_TILE) {
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < C1G_ZG; ++q) v += red[q][og];
        int pr = i >> 8, ta = 0;
        while (pr >= 5 - ta) { pr -= 5 - ta; ++ta; }
        const int tb = ta + pr, a = ta * 16 + ((i >> 4) & 15), b = tb * 16 + (i & 15);
        gram[a * 80 + b] = v;
        if (ta != tb) gram[b * 80 + a] = v;
    }
}
// gram (shifted features, from k_c1_xgram_final) -> the raw Gram data in place (fp64): one block
__global__ __launch_bounds__(256) void k_c1_gram_unshift(double* __restrict__ gram, const float* __restrict__ shift, int K, int T) {
    __shared__ double P[80], c[80];
    const int t = threadIdx.x;
    if (t < 80) { P[t] = t < K ? gram[K * 80 + t] : 0.0; c[t] = t < K ? (double)shift[t / T] : 0.0; }
    __syncthreads();
    const double n = gram[K * 80 + K];
    for (int i = t; i < 80 * 80; i += 256) {
        const int a = i / 80, b = i - a * 80;
        if (a < K && b < K) gram[i] += c[a] * P[b] + P[a] * c[b] + n * c[a] * c[b];
        else if (a == K && b < K) gram[i] = P[b] + n * c[b];
        else if (b == K && a < K) gram[i] = P[a] + n * c[a];
    }
}
// the end of the one-pass backward: block o sums its partial row A[o][0 .. 79] in the fixed order of k_c1_reduce and does the per-channel algebra above in fp64
struct C1BnFin {
    const float* part; int Z, O, K, Opad;
    const float* w; const float* bias;
    const float* save; const float* gamma;      // BatchNorm + sign: mean, invstd [2][O]; gamma
    const float* chan;                          // or the DoReFa block's [9][O] constants (rows 2 .. 4: mean, invstd, gamma; row 8: gamma * invstd)
    const double* gram; double n;
    double scale;                               // of the rows the backward-weight contracted (0.1: the quantizer's STE factor; 1)
    float* dw; float* dbias; float* dgamma; float* dbeta;
};
#define C1F_ZG 12          // z-groups of the partial-row sums (960 threads: the 256 partial rows of nin_gc's first layer are 3 rounds of 8 loads in flight)
__global__ __launch_bounds__(80 * C1F_ZG) void k_c1_bn_final(const C1BnFin f) {
    __shared__ double red[C1F_ZG][80];
    __shared__ double sA[80], sw[80];
    const int o = blockIdx.x, t = threadIdx.x, col = t % 80, zg = t / 80;
    {
        const float* src = f.part + (int64_t)o * 80 + col;
        const int64_t zs = (int64_t)f.Opad * 80;
        double s = 0.0;
        int z = zg;
        for (; z + 7 * C1F_ZG < f.Z; z += 8 * C1F_ZG) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(z + C1F_ZG * u) * zs];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (double)v[u];
        }
        for (; z < f.Z; z += C1F_ZG) s += (double)src[(int64_t)z * zs];
        red[zg][col] = s;
    }
    if (t < 80) sw[t] = t < f.K ? (double)f.w[(int64_t)o * f.K + t] : 0.0;
    __syncthreads();
    if (t < 80) {
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < C1F_ZG; ++q) v += red[q][t];
        sA[t] = v * f.scale;
    }
    __syncthreads();
    if (t >= f.K) return;
    float mean_f, invstd_f, cgi_f;
    if (f.chan) { mean_f = f.chan[2 * f.O + o]; invstd_f = f.chan[3 * f.O + o]; cgi_f = f.chan[8 * f.O + o]; }
    else { mean_f = f.save[o]; invstd_f = f.save[f.O + o]; cgi_f = f.gamma[o] * invstd_f; }
    const double invstd = (double)invstd_f, cgi = (double)cgi_f, n = f.n;
    const double bm = (f.bias ? (double)f.bias[o] : 0.0) - (double)mean_f;
    const double S1 = sA[f.K];
    double dot = 0.0, wg = 0.0;
    {
        int k = 0;
        for (; k + 15 <= f.K; k += 15) {          // 15 Gram loads in flight (one at a time the loop is pure latency), added in order
            double gv[15];
#pragma unroll
            for (int u = 0; u < 15; ++u) gv[u] = f.gram[(k + u) * 80 + t];
#pragma unroll
            for (int u = 0; u < 15; ++u) { dot += sw[k + u] * sA[k + u]; wg += sw[k + u] * gv[u]; }
        }
        for (; k < f.K; ++k) { dot += sw[k] * sA[k]; wg += sw[k] * f.gram[k * 80 + t]; }
    }
    const double S2 = invstd * (dot + bm * S1);
    const double P = f.gram[f.K * 80 + t];
    const double B = invstd * (wg + bm * P);
    f.dw[(int64_t)o * f.K + t] = (float)(cgi * (sA[t] - S1 * P / n - (S2 / n) * B));
    if (t == 0) {
        if (f.dgamma) f.dgamma[o] = (float)S2;
        if (f.dbeta) f.dbeta[o] = (float)S1;
        if (f.dbias) {          // sum dy = -gamma invstd (S2 / n) sum zhat: zero but for the rounding of the saved mean
            double wp = 0.0;
            for (int k = 0; k < f.K; ++k) wp += sw[k] * f.gram[f.K * 80 + k];
            f.dbias[o] = (float)(-cgi * (S2 / n) * (invstd * (wp + n * bm)));
        }
    }
}

// training-mode BatchNorm statistics of y = conv(x, w) + b WITHOUT a pass over y: mean[o] = w[o,:] . P / n + b[o], sum of squared deviations = w (G - P P^T / n) w^T
// (fp64; the expressions behind it are k_bns_final_fwd's)
struct C1Stats {
    const float* w; const float* bias; const double* gram; int O, K; double n;
    float eps, momentum; float* running_mean; float* running_var; float* save;
};
__global__ __launch_bounds__(128) void k_c1_gram_stats(const C1Stats f) {
    __shared__ double sw[80], st[80];
    const int o = blockIdx.x, t = threadIdx.x;
    if (t < 80) sw[t] = t < f.K ? (double)f.w[(int64_t)o * f.K + t] : 0.0;
    __syncthreads();
    if (t < 80) {
        double acc = 0.0;
        if (t < f.K) {
            const double pt = f.gram[f.K * 80 + t] / f.n;
            int k = 0;
            for (; k + 15 <= f.K; k += 15) {
                double gv[15], pv[15];
#pragma unroll
                for (int u = 0; u < 15; ++u) { gv[u] = f.gram[(k + u) * 80 + t]; pv[u] = f.gram[f.K * 80 + k + u]; }
#pragma unroll
                for (int u = 0; u < 15; ++u) acc += sw[k + u] * (gv[u] - pv[u] * pt);
            }
            for (; k < f.K; ++k) acc += sw[k] * (f.gram[k * 80 + t] - f.gram[f.K * 80 + k] * pt);
        }
        st[t] = sw[t] * acc;
    }
    __syncthreads();
    if (t == 0) {
        double ss = 0.0, wp = 0.0;
        for (int k = 0; k < f.K; ++k) { ss += st[k]; wp += sw[k] * f.gram[f.K * 80 + k]; }
        if (ss < 0.0) ss = 0.0;
        const double mean = wp / f.n + (f.bias ? (double)f.bias[o] : 0.0);
        const float var_b = (float)(ss / f.n);
        f.save[o] = (float)mean;
        f.save[f.O + o] = 1.0f / sqrtf(var_b + f.eps);
        if (f.running_mean) f.running_mean[o] = (1.f - f.momentum) * f.running_mean[o] + f.momentum * (float)mean;
        if (f.running_var) f.running_var[o] = (1.f - f.momentum) * f.running_var[o] + f.momentum * (float)(ss / (f.n - 1.0));
    }
}

// ------------------------------------------------------------------------------------------------ host side
struct C1Plan {
    C1Params p;
    int MT, grid_f, grid_w, cblks;
    size_t lds;
    int64_t wp_bytes, off_db, ws_bytes_f, ws_bytes_w;
};
// which: 0 forward, 2 backward-weight (different strip heights; everything else is common)
static int plan_c1(const mn_conv_geom* g, C1Plan* pl, int which = 0) {
    if (g->groups != 1 || g->stride_h != 1 || g->stride_w != 1 || g->dil_h != 1 || g->dil_w != 1 || g->in_shuffle > 1) return 0;
    if (2 * g->pad_h != g->KH - 1 || 2 * g->pad_w != g->KW - 1) return 0;       // "same": Ho = H, Wo = W
    const int K = g->C * g->KH * g->KW;
    if (K > 76 || g->W % 4 || g->W < 4) return 0;
    if ((int64_t)g->N * g->O * g->H * g->W >= ((int64_t)1 << 31)) return 0;       // 32-bit element offsets in the backward-weight
    C1Params& p = pl->p;
    p.N = g->N; p.C = g->C; p.H = g->H; p.W = g->W; p.O = g->O; p.KH = g->KH; p.KW = g->KW; p.ph = g->pad_h; p.pw = g->pad_w;
    p.K = K; p.KS = (K + 3) / 4;
    pl->MT = g->O > 192 ? 4 : (g->O > 128 ? 3 : (g->O > 64 ? 2 : 1));        // 16 MT out-channels per wave, four waves
    const int per_blk = 64 * pl->MT;                      // out-channels per block (4 waves)
    pl->cblks = (g->O + per_blk - 1) / per_blk;
    p.Opad = pl->cblks * per_blk;
    int R = g->H;                                          // strip height: halve while that keeps 16-pixel steps and fills the chip
    // strips: the forward likes one block slot per (image, strip) -- 512 = 2 per CU; the backward-weight one tile per block (measured on L1:
    // forward 131 -> 123 us, backward-weight 141 -> 128 us against 1024 blocks)
    int64_t want_blocks = which == 2 ? 256 : 512;
    while (R % 2 == 0 && ((R / 2) * g->W) % 64 == 0 && (int64_t)g->N * (g->H / R) * pl->cblks < want_blocks) R /= 2;
    if ((R * g->W) % 32) return 0;
    p.R = R; p.strips = g->H / R;
    p.PR = R + g->KH - 1; p.PW = g->W + g->KW - 1 + 3;     // + 3: the 4-wide reads of the last pixel quad stay inside the row
    p.CS = p.PR * p.PW;
    p.xs_bytes = (int)(((size_t)g->C * p.CS * 4 + 64 + 15) / 16 * 16);
    pl->lds = (size_t)p.xs_bytes + (size_t)4 * 16 * pl->MT * (C1_RSA + 32);   // forward uses the patch only
    if (pl->lds > 80 * 1024) return 0;          // two blocks per CU
    p.fd_w = make_fastdiv((uint32_t)g->W);
    const int64_t nbf = (int64_t)g->N * p.strips * pl->cblks;
    if (nbf > 0x7fffffff) return 0;
    pl->grid_f = (int)nbf;
    const int ntiles = g->N * p.strips;
    int Z = 512 / pl->cblks;
    if (Z > ntiles) Z = ntiles;
    if (Z < 1) Z = 1;
    p.Z = Z;
    pl->grid_w = Z * pl->cblks;
    pl->wp_bytes = (int64_t)C1_KS * 4 * p.Opad * 4;
    pl->ws_bytes_f = pl->wp_bytes;
    const int64_t part_bytes = (int64_t)Z * p.Opad * 80 * 4;
    pl->off_db = (part_bytes + 255) / 256 * 256;
    pl->ws_bytes_w = pl->off_db + (int64_t)Z * p.Opad * 4;
    return 1;
}
int c1_supported(const mn_conv_geom* g, int which) {
    C1Plan pl;
    return (which == 0 || which == 2) && plan_c1(g, &pl, which);
}
int64_t c1_ws_bytes(const mn_conv_geom* g, int which) {
    C1Plan pl;
    if (!plan_c1(g, &pl, which)) return 0;
    return which == 0 ? pl.ws_bytes_f : (which == 2 ? pl.ws_bytes_w : 0);
}
int c1_fwd_mm_count(const mn_conv_geom* g) {
    C1Plan pl;
    return plan_c1(g, &pl) ? pl.grid_f : 0;
}
int c1_fwd(const mn_conv_geom* g, const float* x, const float* w, const float* bias, float* y, void* ws, int64_t ws_bytes, hipStream_t s) {
    return c1_fwd_act(g, x, w, bias, y, 0, nullptr, ws, ws_bytes, s);
}
static size_t c1b_lds_bytes(const C1Params& p) {          // patch | tile (weight rows before the loop) | k table | bias | BatchNorm constants (fused epilogue)
    return (size_t)p.xs_bytes + (size_t)3 * 64 * C1B_LD * 2 + C1B_KP * 4 + 64 * 4 * 4 + 64 * 4 * 16;
}
// the fused first block: act 1 = BatchNorm + sign -> int8 codes, act 2 = BatchNorm + ReLU + the next conv's a-bit DoReFa quantizer -> uint8 codes; mask4 for the backward
int c1_fwd_bnact(const mn_conv_geom* g, const float* x, const float* w, const float* bias, const float* save, const float* gamma, const float* beta, int act, int a_bits,
                 void* codes, uint8_t* mask4, hipStream_t s) {
    C1Plan pl;
    if (!plan_c1(g, &pl) || (((uintptr_t)codes) & 3)) MN_FAIL(MN_ENOTSUP, "mn_conv2d_first_bnact_fwd: geometry not covered by the first-layer kernels");
    if (!x || !w || !save || !gamma || !beta || !codes || !mask4 || (act != 1 && act != 2) || (act == 2 && (a_bits < 2 || a_bits > 8)))
        MN_FAIL(MN_EINVAL, "mn_conv2d_first_bnact_fwd: bad arguments");
    C1Params& p = pl.p;
    p.x = x; p.bias = bias; p.y = nullptr; p.gy = nullptr; p.part = nullptr; p.dbpart = nullptr; p.want_db = 0; p.da = nullptr;
    p.relu = 0; p.mm = nullptr; p.wp = w;
    p.bn_save = save; p.bn_gamma = gamma; p.bn_beta = beta; p.codes = codes; p.mask4 = mask4; p.qs = act == 2 ? dorefa_scale(a_bits) : 1.f;
    const size_t lds_b = c1b_lds_bytes(p);
    if (lds_b > 80 * 1024) MN_FAIL(MN_ENOTSUP, "mn_conv2d_first_bnact_fwd: image strip too large for the fused kernel");
    mn_set_last_kernel("k_c1b_fwd<%d, %d>", pl.MT, act);
    { const double ny = (double)g->N * g->O * g->H * g->W; mn_prof_bytes(1.25 * ny + 4.0 * g->N * g->C * g->H * g->W); }
    mn_prof_begin(s);
#define C1B_LAUNCH(MT_, E_) { raise_lds_limit((const void*)k_c1b_fwd<MT_, E_>, lds_b); hipLaunchKernelGGL((k_c1b_fwd<MT_, E_>), dim3(pl.grid_f), dim3(256), lds_b, s, p); }
    if (act == 1) { if (pl.MT == 4) C1B_LAUNCH(4, 1) else if (pl.MT == 3) C1B_LAUNCH(3, 1) else if (pl.MT == 2) C1B_LAUNCH(2, 1) else C1B_LAUNCH(1, 1) }
    else { if (pl.MT == 4) C1B_LAUNCH(4, 2) else if (pl.MT == 3) C1B_LAUNCH(3, 2) else if (pl.MT == 2) C1B_LAUNCH(2, 2) else C1B_LAUNCH(1, 2) }
#undef C1B_LAUNCH
    mn_prof_end(s);
    MN_CHECK_LAUNCH("mn_conv2d_first_bnact_fwd");
    return MN_OK;
}
int c1_fwd_act(const mn_conv_geom* g, const float* x, const float* w, const float* bias, float* y, int relu, float* mm, void* ws, int64_t ws_bytes, hipStream_t s) {
    C1Plan pl;
    if (!plan_c1(g, &pl) || !aligned16(y)) MN_FAIL(MN_ENOTSUP, "mn_conv2d_fwd(first-layer): geometry not covered");
    if (!ws || ws_bytes < pl.ws_bytes_f || !aligned16(ws)) MN_FAIL(MN_ENOSPC, "mn_conv2d_fwd(first-layer): workspace too small");
    C1Params& p = pl.p;
    p.x = x; p.bias = bias; p.y = y; p.gy = nullptr; p.part = nullptr; p.dbpart = nullptr; p.want_db = 0; p.da = nullptr;
    p.relu = relu; p.mm = mm; p.codes = nullptr; p.mask4 = nullptr;
    const size_t lds_b = c1b_lds_bytes(p);
    if (lds_b <= 80 * 1024) {          // three-term bf16 forward (reads the weights as they are: no pack launch)
        p.wp = w;
        mn_set_last_kernel("k_c1b_fwd<%d>", pl.MT);
        mn_prof_begin(s);
        if (pl.MT == 4) { raise_lds_limit((const void*)k_c1b_fwd<4>, lds_b); hipLaunchKernelGGL(k_c1b_fwd<4>, dim3(pl.grid_f), dim3(256), lds_b, s, p); }
        else if (pl.MT == 3) { raise_lds_limit((const void*)k_c1b_fwd<3>, lds_b); hipLaunchKernelGGL(k_c1b_fwd<3>, dim3(pl.grid_f), dim3(256), lds_b, s, p); }
        else if (pl.MT == 2) { raise_lds_limit((const void*)k_c1b_fwd<2>, lds_b); hipLaunchKernelGGL(k_c1b_fwd<2>, dim3(pl.grid_f), dim3(256), lds_b, s, p); }
        else { raise_lds_limit((const void*)k_c1b_fwd<1>, lds_b); hipLaunchKernelGGL(k_c1b_fwd<1>, dim3(pl.grid_f), dim3(256), lds_b, s, p); }
        mn_prof_end(s);
        MN_CHECK_LAUNCH("mn_conv2d_fwd(first-layer)");
        return MN_OK;
    }
    float* wp = (float*)ws;
    hipLaunchKernelGGL(k_c1_pack, dim3(mn_grid_for((int64_t)C1_KS * 4 * p.Opad, 256, 256)), dim3(256), 0, s, w, wp, p.O, p.K, p.Opad, C1_KS * 4);
    p.wp = wp;
    mn_set_last_kernel("k_c1_fwd<%d>", pl.MT);
    mn_prof_begin(s);
    if (pl.MT == 4) { raise_lds_limit((const void*)k_c1_fwd<4>, pl.lds); hipLaunchKernelGGL(k_c1_fwd<4>, dim3(pl.grid_f), dim3(256), pl.lds, s, p); }
    else if (pl.MT == 3) { raise_lds_limit((const void*)k_c1_fwd<3>, pl.lds); hipLaunchKernelGGL(k_c1_fwd<3>, dim3(pl.grid_f), dim3(256), pl.lds, s, p); }
    else if (pl.MT == 2) { raise_lds_limit((const void*)k_c1_fwd<2>, pl.lds); hipLaunchKernelGGL(k_c1_fwd<2>, dim3(pl.grid_f), dim3(256), pl.lds, s, p); }
    else { raise_lds_limit((const void*)k_c1_fwd<1>, pl.lds); hipLaunchKernelGGL(k_c1_fwd<1>, dim3(pl.grid_f), dim3(256), pl.lds, s, p); }
    mn_prof_end(s);
    MN_CHECK_LAUNCH("mn_conv2d_fwd(first-layer)");
    return MN_OK;
}
template <int MT>
static void c1_launch_wgrad_mt(const C1Plan& pl, const C1Params& p, hipStream_t s) {
    if (p.da && p.chan) { raise_lds_limit((const void*)k_c1_wgrad<MT, 2>, pl.lds); hipLaunchKernelGGL((k_c1_wgrad<MT, 2>), dim3(pl.grid_w), dim3(256), pl.lds, s, p); }
    else if (p.da) { raise_lds_limit((const void*)k_c1_wgrad<MT, 1>, pl.lds); hipLaunchKernelGGL((k_c1_wgrad<MT, 1>), dim3(pl.grid_w), dim3(256), pl.lds, s, p); }
    else { raise_lds_limit((const void*)k_c1_wgrad<MT, 0>, pl.lds); hipLaunchKernelGGL((k_c1_wgrad<MT, 0>), dim3(pl.grid_w), dim3(256), pl.lds, s, p); }
}
template <int MT>
static void c1_launch_wgrad_dz_mt(const C1Plan& pl, const C1Params& p, hipStream_t s) {
    if (p.mask4) { raise_lds_limit((const void*)k_c1_wgrad<MT, 3, 1>, pl.lds); hipLaunchKernelGGL((k_c1_wgrad<MT, 3, 1>), dim3(pl.grid_w), dim3(256), pl.lds, s, p); }
    else if (p.chan) { raise_lds_limit((const void*)k_c1_wgrad<MT, 2, 1>, pl.lds); hipLaunchKernelGGL((k_c1_wgrad<MT, 2, 1>), dim3(pl.grid_w), dim3(256), pl.lds, s, p); }
    else { raise_lds_limit((const void*)k_c1_wgrad<MT, 1, 1>, pl.lds); hipLaunchKernelGGL((k_c1_wgrad<MT, 1, 1>), dim3(pl.grid_w), dim3(256), pl.lds, s, p); }
}
static void c1_launch_wgrad_dz(const C1Plan& pl, const C1Params& p, hipStream_t s) {
    if (pl.MT == 4) c1_launch_wgrad_dz_mt<4>(pl, p, s);
    else if (pl.MT == 3) c1_launch_wgrad_dz_mt<3>(pl, p, s);
    else if (pl.MT == 2) c1_launch_wgrad_dz_mt<2>(pl, p, s);
    else c1_launch_wgrad_dz_mt<1>(pl, p, s);
}
static void c1_launch_wgrad(const C1Plan& pl, const C1Params& p, hipStream_t s) {
    if (pl.MT == 4) c1_launch_wgrad_mt<4>(pl, p, s);
    else if (pl.MT == 3) c1_launch_wgrad_mt<3>(pl, p, s);
    else if (pl.MT == 2) c1_launch_wgrad_mt<2>(pl, p, s);
    else c1_launch_wgrad_mt<1>(pl, p, s);
}
// gy == nullptr: the BatchNorm+sign variant (da, yb, save, gamma, beta, sums as for mn_bnsign_bwd)
int c1_bwd_weight_bn(const mn_conv_geom* g, const float* gy, const float* da, const float* yb, const float* save, const float* gamma, const float* beta,
                     const float* sums, int training, const float* x, float* dw, float* dbias, void* ws, int64_t ws_bytes, hipStream_t s);
int c1_bwd_weight(const mn_conv_geom* g, const float* gy, const float* x, float* dw, float* dbias, void* ws, int64_t ws_bytes, hipStream_t s) {
    return c1_bwd_weight_bn(g, gy, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, x, dw, dbias, ws, ws_bytes, s);
}
static int c1_bwd_weight_any(const mn_conv_geom* g, const float* gy, const float* da, const float* yb, const float* save, const float* gamma, const float* beta,
                             const float* chan, int quant, float qs, const float* sums, int training, const float* x, float* dw, float* dbias, void* ws,
                             int64_t ws_bytes, hipStream_t s);
int c1_bwd_weight_bn(const mn_conv_geom* g, const float* gy, const float* da, const float* yb, const float* save, const float* gamma, const float* beta,
                     const float* sums, int training, const float* x, float* dw, float* dbias, void* ws, int64_t ws_bytes, hipStream_t s) {
    return c1_bwd_weight_any(g, gy, da, yb, save, gamma, beta, nullptr, 0, 1.f, sums, training, x, dw, dbias, ws, ws_bytes, s);
}
// the DoReFa block: dq, y, chan [9][O] (mn_qa_chan_from_save), sums of mn_qa_bwd_sums
int c1_bwd_weight_qa(const mn_conv_geom* g, const float* dq, const float* yb, const float* chan, int quant, int a_bits, const float* sums, int training,
                     const float* x, float* dw, float* dbias, void* ws, int64_t ws_bytes, hipStream_t s) {
    if (!chan || a_bits < 2 || a_bits > 8) MN_FAIL(MN_EINVAL, "mn_conv2d_bwd_weight_first_qa: bad arguments");
    return c1_bwd_weight_any(g, nullptr, dq, yb, nullptr, nullptr, nullptr, chan, quant, dorefa_scale(a_bits), sums, training, x, dw, dbias, ws, ws_bytes, s);
}
static int c1_bwd_weight_any(const mn_conv_geom* g, const float* gy, const float* da, const float* yb, const float* save, const float* gamma, const float* beta,
                             const float* chan, int quant, float qs, const float* sums, int training, const float* x, float* dw, float* dbias, void* ws,
                             int64_t ws_bytes, hipStream_t s) {
    C1Plan pl;
    if (!plan_c1(g, &pl, 2) || (gy && !aligned16(gy)) || (da && (!aligned16(da) || !aligned16(yb)))) MN_FAIL(MN_ENOTSUP, "mn_conv2d_bwd_weight(first-layer): geometry not covered");
    if (!gy && (!da || !yb || !sums || (!chan && (!save || !gamma || !beta)))) MN_FAIL(MN_EINVAL, "mn_conv2d_bwd_weight(first-layer, bn): null argument");
    if (!ws || ws_bytes < pl.ws_bytes_w || !aligned16(ws)) MN_FAIL(MN_ENOSPC, "mn_conv2d_bwd_weight(first-layer): workspace too small");
    C1Params& p = pl.p;
    p.x = x; p.gy = gy; p.part = (float*)ws; p.dbpart = (float*)((char*)ws + pl.off_db); p.want_db = dbias != nullptr;
    p.wp = nullptr; p.bias = nullptr; p.y = nullptr;
    p.da = gy ? nullptr : da; p.yb = yb; p.save = save; p.gamma = gamma; p.beta = beta; p.sums = sums; p.training = training;
    p.n_f = (float)g->N * (float)(g->H * g->W);
    p.chan = gy ? nullptr : chan; p.quant = quant; p.qs = qs; p.qs_inv = mn_qa_inv(qs); p.interval = mn_qa_interval();
    mn_set_last_kernel("k_c1_wgrad<%d, %d>", pl.MT, p.da ? (p.chan ? 2 : 1) : 0);
    { const double ny = (double)g->N * g->O * g->H * g->W; mn_prof_bytes((p.da ? 8.0 : 4.0) * ny + 4.0 * g->N * g->C * g->H * g->W); }
    mn_prof_begin(s);
    c1_launch_wgrad(pl, p, s);
    mn_prof_end(s);
    const int64_t total = (int64_t)p.O * p.K + (dbias ? p.O : 0);
    hipLaunchKernelGGL(k_c1_reduce, dim3(mn_grid_for((int64_t)p.Opad * 81, 64, 2048)), dim3(256), 0, s, (const float*)p.part, (const float*)p.dbpart, dw, dbias, p.Z, p.O, p.K, p.Opad);
    MN_CHECK_LAUNCH("mn_conv2d_bwd_weight(first-layer)");
    return MN_OK;
}

// ---- the one-pass backward of the first block (training mode): Gram data of x, then dz-only backward-weight + per-channel finish
static int c1_xgram_grid(const C1Plan& pl) {
    const int ntiles = pl.p.N * pl.p.strips;
    return ntiles < 512 ? ntiles : 512;
}
int64_t c1_xgram_ws_bytes(const mn_conv_geom* g) {
    C1Plan pl;
    if (!plan_c1(g, &pl, 0)) return 0;          // the forward's strips: two blocks per CU
    return (int64_t)c1_xgram_grid(pl) * C1G_TILE * 4 + 512;          // + the shift constants (<= 76 floats) behind the partials
}
int c1_xgram(const mn_conv_geom* g, const float* x, double* gram, void* ws, int64_t ws_bytes, hipStream_t s) {
    C1Plan pl;
    if (!plan_c1(g, &pl, 0) || !c1_supported(g, 2)) MN_FAIL(MN_ENOTSUP, "mn_conv2d_first_xgram: geometry not covered by the first-layer kernels");
    const int Zg = c1_xgram_grid(pl);
    if (!ws || ws_bytes < (int64_t)Zg * C1G_TILE * 4 + 512 || !aligned16(ws)) MN_FAIL(MN_ENOSPC, "mn_conv2d_first_xgram: workspace too small");
    float* shift = (float*)ws + (int64_t)Zg * C1G_TILE;
    C1Params& p = pl.p;
    p.x = x;
    const size_t lds_b = (size_t)p.xs_bytes + (size_t)4 * C1G_TILE * 4;
    mn_set_last_kernel("k_c1_xgram");
    mn_prof_bytes(4.0 * g->N * g->C * g->H * g->W);
    mn_prof_begin(s);
    raise_lds_limit((const void*)k_c1_xgram, lds_b);
    hipLaunchKernelGGL(k_c1_xgram, dim3(Zg), dim3(256), lds_b, s, p, (float*)ws, shift);
    mn_prof_end(s);
    hipLaunchKernelGGL(k_c1_xgram_final, dim3(C1G_TILE / 64), dim3(64 * C1G_ZG), 0, s, (const float*)ws, Zg, gram);
    hipLaunchKernelGGL(k_c1_gram_unshift, dim3(1), dim3(256), 0, s, gram, (const float*)shift, p.K, p.KH * p.KW);
    MN_CHECK_LAUNCH("mn_conv2d_first_xgram");
    return MN_OK;
}
static int c1_bwd_first_any(const mn_conv_geom* g, const float* da, const float* yb, const uint8_t* mask4, double mask_scale, const float* save, const float* gamma,
                            const float* beta, const float* chan, int quant, int a_bits, const float* w, const float* bias, const double* gram, const float* x, float* dw,
                            float* dbias, float* dgamma, float* dbeta, void* ws, int64_t ws_bytes, hipStream_t s);
int c1_bwd_first_gram(const mn_conv_geom* g, const float* da, const float* yb, const float* save, const float* gamma, const float* beta, const float* chan, int quant,
                      int a_bits, const float* w, const float* bias, const double* gram, const float* x, float* dw, float* dbias, float* dgamma, float* dbeta, void* ws,
                      int64_t ws_bytes, hipStream_t s) {
    return c1_bwd_first_any(g, da, yb, nullptr, 1.0, save, gamma, beta, chan, quant, a_bits, w, bias, gram, x, dw, dbias, dgamma, dbeta, ws, ws_bytes, s);
}
// the fused first block's backward: mask4 = the forward's pass bits (quant: their high nibble, rows scaled by 0.1)
int c1_bwd_first_mask(const mn_conv_geom* g, const float* da, const uint8_t* mask4, int quant, const float* save, const float* gamma, const float* w, const float* bias,
                      const double* gram, const float* x, float* dw, float* dbias, float* dgamma, float* dbeta, void* ws, int64_t ws_bytes, hipStream_t s) {
    if (!mask4 || !save || !gamma) MN_FAIL(MN_EINVAL, "mn_conv2d_bwd_first_mask_gram: null argument");
    return c1_bwd_first_any(g, da, nullptr, mask4, quant ? 0.1 : 1.0, save, gamma, gamma, nullptr, quant, 0, w, bias, gram, x, dw, dbias, dgamma, dbeta, ws, ws_bytes, s);
}
static int c1_bwd_first_any(const mn_conv_geom* g, const float* da, const float* yb, const uint8_t* mask4, double mask_scale, const float* save, const float* gamma,
                            const float* beta, const float* chan, int quant, int a_bits, const float* w, const float* bias, const double* gram, const float* x, float* dw,
                            float* dbias, float* dgamma, float* dbeta, void* ws, int64_t ws_bytes, hipStream_t s) {
    C1Plan pl;
    if (!plan_c1(g, &pl, 2) || !aligned16(da) || (yb && !aligned16(yb))) MN_FAIL(MN_ENOTSUP, "mn_conv2d_bwd_first_gram: geometry not covered by the first-layer kernels");
    if (!da || (!yb && !mask4) || !w || !gram || !x || !dw || (!chan && (!save || !gamma || !beta))) MN_FAIL(MN_EINVAL, "mn_conv2d_bwd_first_gram: null argument");
    if (chan && (a_bits < 2 || a_bits > 8)) MN_FAIL(MN_EINVAL, "mn_conv2d_bwd_first_gram: activation bits out of range");
    if (!ws || ws_bytes < pl.ws_bytes_w || !aligned16(ws)) MN_FAIL(MN_ENOSPC, "mn_conv2d_bwd_first_gram: workspace too small");
    C1Params& p = pl.p;
    const float qs = chan ? dorefa_scale(a_bits) : 1.f;
    p.x = x; p.gy = nullptr; p.part = (float*)ws; p.dbpart = nullptr; p.want_db = 0;
    p.wp = nullptr; p.bias = nullptr; p.y = nullptr;
    p.da = da; p.yb = yb; p.save = save; p.gamma = gamma; p.beta = beta; p.sums = nullptr; p.training = 0;          // (k1, k2 of the fold are not used: DZ)
    p.n_f = (float)g->N * (float)(g->H * g->W);
    p.chan = chan; p.quant = quant; p.qs = qs; p.qs_inv = mn_qa_inv(qs); p.interval = mn_qa_interval();
    p.mask4 = const_cast<uint8_t*>(mask4); p.mask_shift = (mask4 && quant) ? 4 : 0;
    mn_set_last_kernel("k_c1_wgrad<%d, %d, 1>", pl.MT, mask4 ? 3 : (chan ? 2 : 1));
    { const double ny = (double)g->N * g->O * g->H * g->W; mn_prof_bytes((mask4 ? 4.25 : 8.0) * ny + 4.0 * g->N * g->C * g->H * g->W); }
    mn_prof_begin(s);
    c1_launch_wgrad_dz(pl, p, s);
    mn_prof_end(s);
    C1BnFin f;
    f.part = p.part; f.Z = p.Z; f.O = p.O; f.K = p.K; f.Opad = p.Opad; f.w = w; f.bias = bias; f.save = save; f.gamma = gamma; f.chan = chan; f.gram = gram;
    f.scale = mask4 ? mask_scale : ((chan && quant) ? 0.1 : 1.0);
    f.n = (double)g->N * (double)(g->H * g->W); f.dw = dw; f.dbias = dbias; f.dgamma = dgamma; f.dbeta = dbeta;
    hipLaunchKernelGGL(k_c1_bn_final, dim3(p.O), dim3(80 * C1F_ZG), 0, s, f);
    MN_CHECK_LAUNCH("mn_conv2d_bwd_first_gram");
    return MN_OK;
}
int c1_gram_bnstats(const mn_conv_geom* g, const float* w, const float* bias, const double* gram, float eps, float momentum, float* running_mean, float* running_var,
                    float* save, hipStream_t s) {
    if (!c1_supported(g, 2)) MN_FAIL(MN_ENOTSUP, "mn_conv2d_first_gram_bnstats: geometry not covered by the first-layer kernels");
    if (!w || !gram || !save) MN_FAIL(MN_EINVAL, "mn_conv2d_first_gram_bnstats: null argument");
    C1Stats f;
    f.w = w; f.bias = bias; f.gram = gram; f.O = g->O; f.K = g->C * g->KH * g->KW; f.n = (double)g->N * (double)(g->H * g->W);
    f.eps = eps; f.momentum = momentum; f.running_mean = running_mean; f.running_var = running_var; f.save = save;
    hipLaunchKernelGGL(k_c1_gram_stats, dim3(g->O), dim3(128), 0, s, f);
    MN_CHECK_LAUNCH("mn_conv2d_first_gram_bnstats");
    return MN_OK;
}
