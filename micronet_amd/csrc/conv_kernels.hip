// Fake-quantized convolution forward / backward-data / backward-weight for gfx950 (CDNA4).
//
// Contraction engine: v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate) -- the operands are the
// reference's own fp32 values (de-quantised codes, real-valued gradients), every product is the
// exact fp32 product the reference's conv forms, only the summation order differs, so results sit
// within ~1e-6 rel of the CPU path.  Layout choices (MI355X_MICROARCH.md / cdna_hip_programming.md):
//   * A[i=pixel][k=channel] / B[k=channel][j=out-channel] fragments need ONE fp32 per lane with
//     lane = 16*k + i, so the NCHW activation tile is staged in LDS in its natural
//     [channel][row][col] order (coalesced float4 HBM reads, no transpose) and read back with
//     ds_read_b32: the 16 lanes of a k-group read 16 consecutive pixels (conflict-free);
//   * D[i=pixel][j=out-channel]: each lane ends up with 4 consecutive pixels of one output channel
//     -> one float4 (dwordx4) store per accumulator tile, 64 B contiguous per channel;
//   * the activation quantizer (DoReFa clamp/round, IAO scale/round/clamp) is applied ONCE per
//     element while the tile travels HBM -> registers -> LDS (fused prologue); zero padding is
//     written after quantisation, as the reference pads the quantised tensor;
//   * backward-data is the same kernel run on gy with flipped weights (zero-insertion for
//     stride > 1) and the clip-STE mask of the activation quantizer fused into the epilogue;
//   * backward-weight is D[i=out-ch][j=in-ch(tap)] with K = pixels, split over the grid into
//     per-block partial tiles that a second kernel sums in a fixed order (deterministic, no atomics).
// A block is 256 threads = 4 waves (one per SIMD); 128-pixel tiles give >=2048 workgroups per
// layer at batch 256, i.e. >=8 per CU across the 8 XCDs.
#include "common.h"
#include "qgemm.h"

// ------------------------------------------------------------------------------------------------
// activation-quantizer descriptors: Pro / pro_apply / make_pro live in common.h
#define EPI_PLAIN 0
#define EPI_BIAS 1
#define EPI_STE 2   // multiply by d actq(x)/dx, x read from aux (same shape as the output)

// ------------------------------------------------------------------------------------------------
// tile geometry shared by the MFMA kernels: a tile is TP consecutive output pixels = whole output
// rows of NI images (NI > 1 only when an image has fewer than TP pixels).
struct TileGeom {
    int TP, NI, TR, tpi;        // pixels/tile, images/tile, out rows per image per tile, tiles per image (>=1)
    int PR, PW, HALO, CS;       // input patch rows, cols (multiple of 4), left halo, LDS channel stride (floats)
    int num_ptiles;
    FastDiv fd_wo, fd_tr, fd_pwq, fd_pr, fd_ni, fd_tpi, fd_ppi;   // fd_ppi: pixels of one image inside a tile (TR*Wo)
    int tpq_shift;                                               // log2(TP / 4)
};
// "forward-style" view: out[n][g*Mg+m][oy][ox] = sum_{c,r,s} Q(in')[n][g*Kc+c][oy*Sh+r*Dh-ph][ox*Sw+s*Dw-pw] * W
// where in' is `in` with U-1 zeros inserted between samples (U > 1 only for strided backward-data).
struct ConvView {
    const float* in;
    int N, Cin_total, Hin, Win, U, Hv, Wv, Kc, G;
    int Cout_total, Ho, Wo, Mg;
    int KH, KW, Sh, Sw, Dh, Dw, ph, pw;
    int vec_in;   // rows of `in` are 16 B aligned and Win % 4 == 0
};
static inline int roundup(int a, int b) { return (a + b - 1) / b * b; }

// returns 0 if the geometry cannot be tiled
static int make_tile_geom(const ConvView& v, int TP, int cs_mod32, TileGeom* t) {
    if (v.Wo < 4 || TP % v.Wo != 0 || v.pw > 8 || v.pw < 0 || v.ph < 0) return 0;
    int rpt = TP / v.Wo;
    t->TP = TP;
    if (rpt >= v.Ho) {
        if (rpt % v.Ho) return 0;
        t->NI = rpt / v.Ho; t->TR = v.Ho; t->tpi = 1;
        t->num_ptiles = (v.N + t->NI - 1) / t->NI;
    } else {
        if (v.Ho % rpt) return 0;
        t->NI = 1; t->TR = rpt; t->tpi = v.Ho / rpt;
        t->num_ptiles = v.N * t->tpi;
    }
    t->PR = (t->TR - 1) * v.Sh + (v.KH - 1) * v.Dh + 1;
    t->HALO = roundup(v.pw, 4);
    int max_vc = (v.Wo - 1) * v.Sw + (v.KW - 1) * v.Dw - v.pw;   // last virtual input column touched
    if (max_vc < 0) max_vc = 0;
    t->PW = roundup(t->HALO + max_vc + 1, 4);
    int cs = t->NI * t->PR * t->PW;                 // multiple of 4
    while ((cs & 31) != cs_mod32) cs += 4;          // bank spreading between channels (see kernels)
    t->CS = cs;
    t->fd_wo = make_fastdiv(v.Wo); t->fd_tr = make_fastdiv(t->TR); t->fd_pwq = make_fastdiv(t->PW / 4);
    t->fd_pr = make_fastdiv(t->PR); t->fd_ni = make_fastdiv(t->NI); t->fd_tpi = make_fastdiv(t->tpi);
    t->fd_ppi = make_fastdiv(t->TR * v.Wo);
    t->tpq_shift = 0;
    while ((4 << t->tpq_shift) < TP) ++t->tpq_shift;
    if ((4 << t->tpq_shift) != TP) return 0;
    return 1;
}

// stage CK channels [c0, c0+CK) of group g of the input patch of one tile into LDS: xs[cl][img][prow][pcol]
__device__ __forceinline__ void stage_patch(float* __restrict__ xs, const ConvView& v, const TileGeom& t, const Pro& pro,
                                            int g, int c0, int CK, int n0, int row0) {
    const int PWQ = t.PW >> 2;
    const int nq = CK * t.NI * t.PR * PWQ;
    float sc = 1.f, zp = 0.f;
    if (pro.mode == MN_ACTQ_IAO) { sc = pro.qp[0]; zp = pro.qp[1]; }
    for (int q = threadIdx.x; q < nq; q += blockDim.x) {
        const uint32_t t1 = fd_div(q, t.fd_pwq);
        const int pq = q - t1 * PWQ;
        const uint32_t t2 = fd_div(t1, t.fd_pr);
        const int prow = t1 - t2 * t.PR;
        const uint32_t cl = fd_div(t2, t.fd_ni);
        const int img = t2 - cl * t.NI;
        const int c = c0 + (int)cl, n = n0 + img;
        const int vr = row0 + prow, vc0 = pq * 4 - t.HALO;
        float e[4] = {0.f, 0.f, 0.f, 0.f};
        if (c < v.Kc && n < v.N && vr >= 0 && vr < v.Hv) {
            if (v.U == 1) {
                const float* rowp = v.in + (((int64_t)n * v.Cin_total + (int64_t)g * v.Kc + c) * v.Hin + vr) * v.Win;
                if (v.vec_in && vc0 >= 0 && vc0 + 3 < v.Win) {
                    const float4 f = *reinterpret_cast<const float4*>(rowp + vc0);
                    e[0] = pro_apply(pro, f.x, sc, zp); e[1] = pro_apply(pro, f.y, sc, zp);
                    e[2] = pro_apply(pro, f.z, sc, zp); e[3] = pro_apply(pro, f.w, sc, zp);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int vc = vc0 + j;
                        if (vc >= 0 && vc < v.Win) e[j] = pro_apply(pro, rowp[vc], sc, zp);
                    }
                }
            } else if (vr % v.U == 0) {
                const float* rowp = v.in + (((int64_t)n * v.Cin_total + (int64_t)g * v.Kc + c) * v.Hin + vr / v.U) * v.Win;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int vc = vc0 + j;
                    if (vc >= 0 && vc < v.Wv && vc % v.U == 0) e[j] = pro_apply(pro, rowp[vc / v.U], sc, zp);
                }
            }
        }
        *reinterpret_cast<float4*>(xs + (int64_t)cl * t.CS + (img * t.PR + prow) * t.PW + pq * 4) = make_float4(e[0], e[1], e[2], e[3]);
    }
}
// LDS offset (floats, within one channel) of the top-left tap of output pixel `pidx` of the tile
__device__ __forceinline__ int pixel_patch_offset(const ConvView& v, const TileGeom& t, int pidx) {
    const uint32_t fr = fd_div(pidx, t.fd_wo);
    const int ow = pidx - fr * v.Wo;
    const uint32_t img = fd_div(fr, t.fd_tr);
    const int orow = fr - img * t.TR;
    return ((int)img * t.PR + orow * v.Sh) * t.PW + ow * v.Sw + (t.HALO - v.pw);
}

// ------------------------------------------------------------------------------------------------
// weight re-layout for the MFMA kernels (tiny; runs once per call)
//   forward : wp[g][tap][c (Kcp)][m (Mgpad)]           = w[g*Mg+m][c][r][s]
//   bwd-data: wp[g][tap'][m (Mgp4)][c (Cgpad)]         = w[g*Mg+m][c][KH-1-r'][KW-1-s']   (roles of c and m swap)
__global__ __launch_bounds__(256) void k_pack_weights(const float* __restrict__ w, float* __restrict__ wp, int G, int Mg, int Cg,
                                                      int KH, int KW, int Kp, int Jp, int transpose_flip) {
    // output index space: [G][T][Kp][Jp]; forward: K = c, J = m ; bwd-data: K = m, J = c
    const int T = KH * KW;
    const int64_t total = (int64_t)G * T * Kp * Jp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int j = (int)(i % Jp);
        int64_t r1 = i / Jp;
        int k = (int)(r1 % Kp);
        int64_t r2 = r1 / Kp;
        int tap = (int)(r2 % T);
        int g = (int)(r2 / T);
        int m, c, r, s;
        if (!transpose_flip) { c = k; m = j; r = tap / KW; s = tap % KW; }
        else { m = k; c = j; r = KH - 1 - tap / KW; s = KW - 1 - tap % KW; }
        float val = 0.f;
        if (m < Mg && c < Cg) val = w[((((int64_t)g * Mg + m) * Cg + c) * KH + r) * KW + s];
        wp[i] = val;
    }
}

// ------------------------------------------------------------------------------------------------
// MFMA forward-style kernel (forward, and backward-data on gy with flipped weights)
struct FwdParams {
    ConvView v;
    TileGeom t;
    Pro pro;
    const float* wp;     // packed weights [G][T][Kcp][Mgpad]
    float* out;
    const float* bias;   // EPI_BIAS
    const float* aux;    // EPI_STE: x
    Pro ste;             // EPI_STE: which activation quantizer to differentiate
    int epi;
    int Kcp, Mgpad, CK, ck_shift, num_mblk;
};

template <int MT>
__global__ __launch_bounds__(256) void k_conv_mfma(const FwdParams p) {
    HIP_DYNAMIC_SHARED(float, smem)
    const ConvView& v = p.v;
    const TileGeom& t = p.t;
    constexpr int TMB = 16 * MT;
    float* xs = smem;
    float* wsm = smem + p.CK * t.CS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kk = lane >> 4, l15 = lane & 15;
    const int T = v.KH * v.KW;

    uint32_t b = blockIdx.x;
    const int pt = b % t.num_ptiles; b /= t.num_ptiles;
    const int mblk = b % p.num_mblk;
    const int g = b / p.num_mblk;
    const uint32_t timg = fd_div(pt, t.fd_tpi);
    const int n0 = (int)timg * t.NI;
    const int oh0 = (pt - (int)timg * t.tpi) * t.TR;
    const int row0 = oh0 * v.Sh - v.ph;

    int pixoffA[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) pixoffA[i] = pixel_patch_offset(v, t, (wave * 2 + i) * 16 + l15) + kk * t.CS;

    f32x4 acc[2][MT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[i][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    const float* wg = p.wp + (int64_t)g * T * p.Kcp * p.Mgpad + (int64_t)mblk * TMB;
    for (int c0 = 0; c0 < p.Kcp; c0 += p.CK) {
        stage_patch(xs, v, t, p.pro, g, c0, p.CK, n0, row0);
        {   // weights chunk: wsm[tap][cl][TMB]
            constexpr int MQ = TMB / 4;
            const int nq = T * p.CK * MQ;
            for (int q = tid; q < nq; q += 256) {
                const int mq = q % MQ;
                const int r1 = q / MQ;
                const int cl = r1 & (p.CK - 1);      // CK is a power of two
                const int tap = r1 >> p.ck_shift;
                const float4 f = *reinterpret_cast<const float4*>(wg + ((int64_t)tap * p.Kcp + c0 + cl) * p.Mgpad + mq * 4);
                *reinterpret_cast<float4*>(wsm + (tap * p.CK + cl) * TMB + mq * 4) = f;
            }
        }
        __syncthreads();
        int tap = 0;
        for (int r = 0; r < v.KH; ++r) {
            for (int s = 0; s < v.KW; ++s, ++tap) {
                const float* xa = xs + r * v.Dh * t.PW + s * v.Dw;
                const float* wb = wsm + (tap * p.CK + kk) * TMB + l15;
                for (int cl = 0; cl < p.CK; cl += 4) {
                    const float a0 = xa[cl * t.CS + pixoffA[0]];
                    const float a1 = xa[cl * t.CS + pixoffA[1]];
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const float bv = wb[cl * TMB + m * 16];
                        acc[0][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, acc[0][m], 0, 0, 0);
                        acc[1][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv, acc[1][m], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }

    // epilogue: lane holds pixels (lane>>4)*4 .. +3 of output channel l15 of each 16x16 tile
    float sc = 1.f, zp = 0.f, lo = 0.f, hi = 0.f;
    if (p.epi == EPI_STE && p.ste.mode == MN_ACTQ_IAO) { sc = p.ste.qp[0]; zp = p.ste.qp[1]; lo = p.ste.qp[2]; hi = p.ste.qp[3]; }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pidx = (wave * 2 + i) * 16 + kk * 4;
        const uint32_t fr = fd_div(pidx, t.fd_wo);
        const int ow = pidx - fr * v.Wo;
        const uint32_t img = fd_div(fr, t.fd_tr);
        const int orow = fr - img * t.TR;
        const int n = n0 + (int)img;
        if (n >= v.N) continue;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int mo = mblk * TMB + m * 16 + l15;
            if (mo >= v.Mg) continue;
            const int64_t off = (((int64_t)n * v.Cout_total + (int64_t)g * v.Mg + mo) * v.Ho + oh0 + orow) * v.Wo + ow;
            float o0 = acc[i][m][0], o1 = acc[i][m][1], o2 = acc[i][m][2], o3 = acc[i][m][3];
            if (p.epi == EPI_BIAS) {
                if (p.bias) { const float bb = p.bias[g * v.Mg + mo]; o0 += bb; o1 += bb; o2 += bb; o3 += bb; }
            } else if (p.epi == EPI_STE) {
                const float4 xv = *reinterpret_cast<const float4*>(p.aux + off);
                if (p.ste.mode == MN_ACTQ_DOREFA) {
                    o0 = dorefa_act_grad(o0, xv.x, p.ste.s); o1 = dorefa_act_grad(o1, xv.y, p.ste.s);
                    o2 = dorefa_act_grad(o2, xv.z, p.ste.s); o3 = dorefa_act_grad(o3, xv.w, p.ste.s);
                } else if (p.ste.mode == MN_ACTQ_IAO) {
                    o0 = iao_fq_grad(o0, xv.x, sc, zp, lo, hi, p.ste.qmin, p.ste.qmax);
                    o1 = iao_fq_grad(o1, xv.y, sc, zp, lo, hi, p.ste.qmin, p.ste.qmax);
                    o2 = iao_fq_grad(o2, xv.z, sc, zp, lo, hi, p.ste.qmin, p.ste.qmax);
                    o3 = iao_fq_grad(o3, xv.w, sc, zp, lo, hi, p.ste.qmin, p.ste.qmax);
                }
            }
            *reinterpret_cast<float4*>(p.out + off) = make_float4(o0, o1, o2, o3);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// MFMA backward-weight: D[i = out-channel][j = (in-channel tile, tap)] , K = output pixels
struct WgradParams {
    ConvView v;          // forward view of x (prologue = activation quantizer)
    TileGeom t;
    Pro pro;
    const float* gy;     // [N][G*Mg][Ho][Wo]
    float* part;         // [Z][G][T][Mgw][Cgw]
    int Z, nmb, ncb, WM, WJ, MTW, CTW, TMW, TCW, Mgw, Cgw, GS;
};
template <int MPW, int JPW>
__global__ __launch_bounds__(256) void k_wgrad_mfma(const WgradParams p) {
    HIP_DYNAMIC_SHARED(float, smem)
    const ConvView& v = p.v;
    const TileGeom& t = p.t;
    float* gs = smem;                                  // [TMW][GS]
    float* xs = gs + p.TMW * p.GS;                     // [TCW][CS]
    int* poff = reinterpret_cast<int*>(xs + p.TCW * t.CS);   // [TP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kk = lane >> 4, l15 = lane & 15;
    const int T = v.KH * v.KW, JT = p.CTW * T;

    uint32_t b = blockIdx.x;
    const int z = b % p.Z; b /= p.Z;
    const int cb = b % p.ncb; b /= p.ncb;
    const int mb = b % p.nmb;
    const int g = b / p.nmb;
    const int wm = wave % p.WM, wj = wave / p.WM;

    int aoff[MPW], boff[JPW];
    bool jvalid[JPW];
#pragma unroll
    for (int mi = 0; mi < MPW; ++mi) aoff[mi] = ((wm * MPW + mi) * 16 + l15) * p.GS + kk;
#pragma unroll
    for (int jj = 0; jj < JPW; ++jj) {
        const int j = wj + jj * p.WJ;
        jvalid[jj] = j < JT;
        const int ct = jvalid[jj] ? j / T : 0, tap = jvalid[jj] ? j % T : 0;
        const int r = tap / v.KW, s = tap % v.KW;
        boff[jj] = (ct * 16 + l15) * t.CS + r * v.Dh * t.PW + s * v.Dw;
    }
    for (int i = tid; i < t.TP; i += 256) poff[i] = pixel_patch_offset(v, t, i);

    f32x4 acc[MPW][JPW];
#pragma unroll
    for (int mi = 0; mi < MPW; ++mi)
#pragma unroll
        for (int jj = 0; jj < JPW; ++jj) acc[mi][jj] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int HoWo = v.Ho * v.Wo;
    const int px_per_img = t.TR * v.Wo;             // pixels of one image inside a tile
    const int TPQ = t.TP >> 2;
    for (int pt = z; pt < t.num_ptiles; pt += p.Z) {
        const uint32_t timg = fd_div(pt, t.fd_tpi);
        const int n0 = (int)timg * t.NI;
        const int oh0 = (pt - (int)timg * t.tpi) * t.TR;
        const int row0 = oh0 * v.Sh - v.ph;
        // gy tile: gs[m][pixel]; within one image the tile's pixels are contiguous in memory
        for (int q = tid; q < p.TMW * TPQ; q += 256) {
            const int m = q >> t.tpq_shift, pq = q - (m << t.tpq_shift);
            const int pix = pq * 4;
            const uint32_t img = fd_div(pix, t.fd_ppi);
            const int rem = pix - img * px_per_img;
            const int n = n0 + (int)img, mo = mb * p.TMW + m;
            float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n < v.N && mo < v.Mg)
                f = *reinterpret_cast<const float4*>(p.gy + ((int64_t)n * v.Cout_total + (int64_t)g * v.Mg + mo) * HoWo + oh0 * v.Wo + rem);
            *reinterpret_cast<float4*>(gs + m * p.GS + pix) = f;
        }
        stage_patch(xs, v, t, p.pro, g, cb * p.TCW, p.TCW, n0, row0);
        __syncthreads();
        for (int p0 = 0; p0 < t.TP; p0 += 4) {
            const int po = poff[p0 + kk];
            float a[MPW];
#pragma unroll
            for (int mi = 0; mi < MPW; ++mi) a[mi] = gs[aoff[mi] + p0];
#pragma unroll
            for (int jj = 0; jj < JPW; ++jj) {
                if (!jvalid[jj]) continue;     // wave-uniform
                const float bv = xs[boff[jj] + po];
#pragma unroll
                for (int mi = 0; mi < MPW; ++mi) acc[mi][jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi], bv, acc[mi][jj], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // partial tiles: rows = out-channel (lane>>4)*4 + r, col = in-channel l15
#pragma unroll
    for (int jj = 0; jj < JPW; ++jj) {
        if (!jvalid[jj]) continue;
        const int j = wj + jj * p.WJ;
        const int ct = j / T, tap = j % T;
#pragma unroll
        for (int mi = 0; mi < MPW; ++mi) {
            const int mrow = mb * p.TMW + (wm * MPW + mi) * 16 + kk * 4;
            const int ccol = cb * p.TCW + ct * 16 + l15;
            float* dst = p.part + ((((int64_t)z * v.G + g) * T + tap) * p.Mgw + mrow) * p.Cgw + ccol;
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[(int64_t)r * p.Cgw] = acc[mi][jj][r];
        }
    }
}
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ part, float* __restrict__ dw, int Z, int G, int Mg, int Cg,
                                                      int KH, int KW, int Mgw, int Cgw) {
    const int T = KH * KW;
    const int64_t total = (int64_t)G * Mg * Cg * T;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int tap = (int)(i % T);
        int64_t r1 = i / T;
        const int c = (int)(r1 % Cg);
        const int64_t o = r1 / Cg;
        const int g = (int)(o / Mg), m = (int)(o % Mg);
        float s = 0.f;
        for (int z = 0; z < Z; ++z) s += part[((((int64_t)z * G + g) * T + tap) * Mgw + m) * Cgw + c];
        dw[i] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// generic direct kernels (any geometry): correctness fallback for shapes the tiler rejects
struct DirectParams {
    int N, C, H, W, O, KH, KW, Sh, Sw, ph, pw, Dh, Dw, G, Ho, Wo, Cg, Mg;
    Pro pro;
};
__global__ __launch_bounds__(256) void k_conv_direct_fwd(const DirectParams p, const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y) {
    const int64_t total = (int64_t)p.N * p.O * p.Ho * p.Wo;
    float sc = 1.f, zp = 0.f;
    if (p.pro.mode == MN_ACTQ_IAO) { sc = p.pro.qp[0]; zp = p.pro.qp[1]; }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ow = (int)(i % p.Wo);
        int64_t r1 = i / p.Wo;
        const int oh = (int)(r1 % p.Ho);
        r1 /= p.Ho;
        const int o = (int)(r1 % p.O);
        const int n = (int)(r1 / p.O);
        const int g = o / p.Mg;
        float acc = 0.f;
        for (int c = 0; c < p.Cg; ++c) {
            const float* xc = x + ((int64_t)n * p.C + g * p.Cg + c) * p.H * p.W;
            const float* wc = w + ((int64_t)o * p.Cg + c) * p.KH * p.KW;
            for (int r = 0; r < p.KH; ++r) {
                const int ih = oh * p.Sh + r * p.Dh - p.ph;
                if (ih < 0 || ih >= p.H) continue;
                for (int s = 0; s < p.KW; ++s) {
                    const int iw = ow * p.Sw + s * p.Dw - p.pw;
                    if (iw < 0 || iw >= p.W) continue;
                    acc = fmaf(pro_apply(p.pro, xc[ih * p.W + iw], sc, zp), wc[r * p.KW + s], acc);
                }
            }
        }
        y[i] = bias ? acc + bias[o] : acc;
    }
}
__global__ __launch_bounds__(256) void k_conv_direct_bwd_data(const DirectParams p, const float* __restrict__ gy, const float* __restrict__ w,
                                                              const float* __restrict__ x, float* __restrict__ dx) {
    const int64_t total = (int64_t)p.N * p.C * p.H * p.W;
    float sc = 1.f, zp = 0.f, lo = 0.f, hi = 0.f;
    if (p.pro.mode == MN_ACTQ_IAO) { sc = p.pro.qp[0]; zp = p.pro.qp[1]; lo = p.pro.qp[2]; hi = p.pro.qp[3]; }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int iw = (int)(i % p.W);
        int64_t r1 = i / p.W;
        const int ih = (int)(r1 % p.H);
        r1 /= p.H;
        const int c = (int)(r1 % p.C);
        const int n = (int)(r1 / p.C);
        const int g = c / p.Cg, cl = c % p.Cg;
        float acc = 0.f;
        for (int m = 0; m < p.Mg; ++m) {
            const int o = g * p.Mg + m;
            const float* gyo = gy + ((int64_t)n * p.O + o) * p.Ho * p.Wo;
            const float* wo = w + ((int64_t)o * p.Cg + cl) * p.KH * p.KW;
            for (int r = 0; r < p.KH; ++r) {
                const int th = ih + p.ph - r * p.Dh;
                if (th < 0 || th % p.Sh) continue;
                const int oh = th / p.Sh;
                if (oh >= p.Ho) continue;
                for (int s = 0; s < p.KW; ++s) {
                    const int tw = iw + p.pw - s * p.Dw;
                    if (tw < 0 || tw % p.Sw) continue;
                    const int ow = tw / p.Sw;
                    if (ow >= p.Wo) continue;
                    acc = fmaf(gyo[oh * p.Wo + ow], wo[r * p.KW + s], acc);
                }
            }
        }
        if (p.pro.mode == MN_ACTQ_DOREFA) acc = dorefa_act_grad(acc, x[i], p.pro.s);
        else if (p.pro.mode == MN_ACTQ_IAO) acc = iao_fq_grad(acc, x[i], sc, zp, lo, hi, p.pro.qmin, p.pro.qmax);
        dx[i] = acc;
    }
}
// one workgroup per weight element
__global__ __launch_bounds__(256) void k_conv_direct_bwd_weight(const DirectParams p, const float* __restrict__ gy, const float* __restrict__ x,
                                                                float* __restrict__ dw) {
    __shared__ double scd[16];
    float sc = 1.f, zp = 0.f;
    if (p.pro.mode == MN_ACTQ_IAO) { sc = p.pro.qp[0]; zp = p.pro.qp[1]; }
    int64_t i = blockIdx.x;
    const int s = (int)(i % p.KW); i /= p.KW;
    const int r = (int)(i % p.KH); i /= p.KH;
    const int cl = (int)(i % p.Cg);
    const int o = (int)(i / p.Cg);
    const int g = o / p.Mg;
    const int64_t total = (int64_t)p.N * p.Ho * p.Wo;
    double acc = 0.0;
    for (int64_t k = threadIdx.x; k < total; k += blockDim.x) {
        const int ow = (int)(k % p.Wo);
        int64_t r1 = k / p.Wo;
        const int oh = (int)(r1 % p.Ho);
        const int n = (int)(r1 / p.Ho);
        const int ih = oh * p.Sh + r * p.Dh - p.ph, iw = ow * p.Sw + s * p.Dw - p.pw;
        if (ih < 0 || ih >= p.H || iw < 0 || iw >= p.W) continue;
        const float xv = pro_apply(p.pro, x[(((int64_t)n * p.C + g * p.Cg + cl) * p.H + ih) * p.W + iw], sc, zp);
        acc += (double)(gy[(((int64_t)n * p.O + o) * p.Ho + oh) * p.Wo + ow] * xv);
    }
    acc = block_reduce(acc, OpAddD(), 0.0, scd);
    if (threadIdx.x == 0) dw[blockIdx.x] = (float)acc;
}
// dbias[o] = sum over (n, pixels) of gy ; one workgroup per channel, fp64 accumulation (order independent)
__global__ __launch_bounds__(256) void k_bias_grad(const float* __restrict__ gy, float* __restrict__ db, int N, int O, int HW) {
    __shared__ double scd[16];
    const int o = blockIdx.x;
    double acc = 0.0;
    const int vec = (HW % 4 == 0) && aligned16(gy);
    for (int n = 0; n < N; ++n) {
        const float* p = gy + ((int64_t)n * O + o) * HW;
        if (vec) {
            for (int j = threadIdx.x; j < HW / 4; j += blockDim.x) {
                const float4 f = reinterpret_cast<const float4*>(p)[j];
                acc += ((double)f.x + (double)f.y) + ((double)f.z + (double)f.w);
            }
        } else {
            for (int j = threadIdx.x; j < HW; j += blockDim.x) acc += (double)p[j];
        }
    }
    acc = block_reduce(acc, OpAddD(), 0.0, scd);
    if (threadIdx.x == 0) db[o] = (float)acc;
}

// ------------------------------------------------------------------------------------------------
// host side: planning
static int out_dim(int in, int k, int s, int p, int d) { return (in + 2 * p - d * (k - 1) - 1) / s + 1; }
static int pow2_floor(int v) { int r = 1; while (r * 2 <= v) r *= 2; return r; }

static int check_geom(const mn_conv_geom* g, const char* what) {
    if (!g) MN_FAIL(MN_EINVAL, "%s: null geometry", what);
    if (g->N <= 0 || g->C <= 0 || g->H <= 0 || g->W <= 0 || g->O <= 0 || g->KH <= 0 || g->KW <= 0 || g->stride_h <= 0 ||
        g->stride_w <= 0 || g->dil_h <= 0 || g->dil_w <= 0 || g->pad_h < 0 || g->pad_w < 0 || g->groups <= 0 ||
        g->C % g->groups || g->O % g->groups)
        MN_FAIL(MN_EINVAL, "%s: invalid geometry", what);
    if (out_dim(g->H, g->KH, g->stride_h, g->pad_h, g->dil_h) <= 0 || out_dim(g->W, g->KW, g->stride_w, g->pad_w, g->dil_w) <= 0)
        MN_FAIL(MN_EINVAL, "%s: empty output", what);
    return MN_OK;
}
static const int LDS_CAP_FWD = 64 * 1024;
static const int LDS_CAP_WGRAD = 64 * 1024;

struct FwdPlan { FwdParams p; int MT; size_t lds; int64_t wp_floats; int grid; };
// view: forward (which == 0) or backward-data (which == 1) expressed as a forward-style contraction
static int plan_fwd_view(const mn_conv_geom* g, int which, FwdPlan* pl) {
    ConvView v;
    const int Ho = out_dim(g->H, g->KH, g->stride_h, g->pad_h, g->dil_h), Wo = out_dim(g->W, g->KW, g->stride_w, g->pad_w, g->dil_w);
    const int Cg = g->C / g->groups, Mg = g->O / g->groups;
    v.N = g->N; v.G = g->groups; v.KH = g->KH; v.KW = g->KW; v.Dh = g->dil_h; v.Dw = g->dil_w;
    if (which == 0) {
        v.Cin_total = g->C; v.Hin = g->H; v.Win = g->W; v.U = 1; v.Hv = g->H; v.Wv = g->W; v.Kc = Cg;
        v.Cout_total = g->O; v.Ho = Ho; v.Wo = Wo; v.Mg = Mg; v.Sh = g->stride_h; v.Sw = g->stride_w; v.ph = g->pad_h; v.pw = g->pad_w;
    } else {
        if (g->stride_h != g->stride_w) return 0;
        v.Cin_total = g->O; v.Hin = Ho; v.Win = Wo; v.U = g->stride_h; v.Hv = (Ho - 1) * v.U + 1; v.Wv = (Wo - 1) * v.U + 1; v.Kc = Mg;
        v.Cout_total = g->C; v.Ho = g->H; v.Wo = g->W; v.Mg = Cg; v.Sh = 1; v.Sw = 1;
        v.ph = (g->KH - 1) * g->dil_h - g->pad_h; v.pw = (g->KW - 1) * g->dil_w - g->pad_w;
        if (v.ph < 0 || v.pw < 0) return 0;
    }
    v.in = nullptr; v.vec_in = (v.Win % 4 == 0);
    TileGeom t;
    if (!make_tile_geom(v, 128, 16, &t)) return 0;
    const int T = v.KH * v.KW;
    int TMB = v.Mg <= 16 ? 16 : (v.Mg <= 32 ? 32 : (v.Mg <= 64 ? 64 : 128));
    const int Kcp = roundup(v.Kc, 4);
    int CK = 4;
    while (CK < 32 && Kcp % (CK * 2) == 0) CK *= 2;
    size_t lds;
    for (;;) {
        lds = ((size_t)CK * t.CS + (size_t)T * CK * TMB) * sizeof(float);
        if (lds <= (size_t)LDS_CAP_FWD) break;
        if (CK > 4) CK /= 2;
        else if (TMB > 16) TMB /= 2;
        else return 0;
    }
    FwdParams& p = pl->p;
    p.v = v; p.t = t; p.Kcp = Kcp; p.CK = CK;
    p.ck_shift = 0;
    while ((1 << p.ck_shift) < CK) ++p.ck_shift;
    p.num_mblk = (v.Mg + TMB - 1) / TMB;
    p.Mgpad = p.num_mblk * TMB;
    pl->MT = TMB / 16;
    pl->lds = lds;
    pl->wp_floats = (int64_t)v.G * T * Kcp * p.Mgpad;
    const int64_t grid = (int64_t)t.num_ptiles * p.num_mblk * v.G;
    if (grid > 0x7fffffff) return 0;
    pl->grid = (int)grid;
    return 1;
}

struct WgradPlan { WgradParams p; int MPW; size_t lds; int64_t part_floats; int grid; };
static int plan_wgrad(const mn_conv_geom* g, WgradPlan* pl) {
    ConvView v;
    const int Ho = out_dim(g->H, g->KH, g->stride_h, g->pad_h, g->dil_h), Wo = out_dim(g->W, g->KW, g->stride_w, g->pad_w, g->dil_w);
    const int Cg = g->C / g->groups, Mg = g->O / g->groups;
    v.in = nullptr; v.N = g->N; v.G = g->groups; v.KH = g->KH; v.KW = g->KW; v.Dh = g->dil_h; v.Dw = g->dil_w;
    v.Cin_total = g->C; v.Hin = g->H; v.Win = g->W; v.U = 1; v.Hv = g->H; v.Wv = g->W; v.Kc = Cg;
    v.Cout_total = g->O; v.Ho = Ho; v.Wo = Wo; v.Mg = Mg; v.Sh = g->stride_h; v.Sw = g->stride_w; v.ph = g->pad_h; v.pw = g->pad_w;
    v.vec_in = (v.Win % 4 == 0);
    const int T = v.KH * v.KW;
    int MTW = (Mg + 15) / 16;
    MTW = MTW >= 8 ? 8 : (MTW > 4 ? 8 : (MTW > 2 ? 4 : (MTW > 1 ? 2 : 1)));
    int WM, MPW, WJ, JTmax;
    for (;;) {
        WM = MTW < 4 ? MTW : 4; MPW = MTW / WM; WJ = 4 / WM;
        JTmax = (MPW == 2 ? 9 : 13) * WJ;
        if (T <= JTmax) break;
        if (MTW == 1) return 0;
        MTW /= 2;
    }
    int CTWmax = (Cg + 15) / 16;
    if (CTWmax > JTmax / T) CTWmax = JTmax / T;
    WgradParams& p = pl->p;
    p.MTW = MTW; p.WM = WM; p.WJ = WJ; p.TMW = MTW * 16;
    p.nmb = (Mg + p.TMW - 1) / p.TMW;
    p.Mgw = p.nmb * p.TMW;
    // largest pixel tile whose gy tile + input patch fit in LDS; shrink the channel block before the pixel tile
    TileGeom t;
    int ok = 0;
    size_t lds = 0;
    for (int pass = 0; pass < 2 && !ok; ++pass) {
        for (int TP = 128; TP >= 16 && !ok; TP /= 2) {
            if (!make_tile_geom(v, TP, 4, &t)) continue;
            p.GS = TP + 4;
            for (int ctw = CTWmax; ctw >= 1; --ctw) {
                lds = ((size_t)p.TMW * p.GS + (size_t)ctw * 16 * t.CS + TP) * sizeof(float);
                if (lds > (size_t)LDS_CAP_WGRAD) continue;
                if (pass == 0 && ctw < (CTWmax < 4 ? CTWmax : 4)) break;   // first pass: insist on a decent channel block
                p.CTW = ctw; ok = 1;
                break;
            }
        }
    }
    if (!ok) return 0;
    p.TCW = p.CTW * 16;
    p.ncb = (Cg + p.TCW - 1) / p.TCW;
    p.Cgw = p.ncb * p.TCW;
    p.v = v; p.t = t;
    const int base = v.G * p.nmb * p.ncb;
    int Z = 512 / base;
    if (Z < 1) Z = 1;
    if (Z > t.num_ptiles) Z = t.num_ptiles;
    p.Z = Z;
    pl->MPW = MPW;
    pl->lds = lds;
    pl->part_floats = (int64_t)Z * v.G * T * p.Mgw * p.Cgw;
    pl->grid = base * Z;
    return 1;
}

extern "C" int mn_conv2d_mfma_supported(const mn_conv_geom* g, int which) {
    if (check_geom(g, "mn_conv2d_mfma_supported") != MN_OK) return 0;
    if (which == 0 || which == 1) { FwdPlan pl; return plan_fwd_view(g, which, &pl); }
    if (which == 2) { WgradPlan pl; return plan_wgrad(g, &pl); }
    return 0;
}
extern "C" int64_t mn_conv2d_ws_bytes(const mn_conv_geom* g, int which, int algo) {
    if (check_geom(g, "mn_conv2d_ws_bytes") != MN_OK) return -1;
    if (algo == MN_ALGO_DIRECT) return 0;
    if (which < 0 || which > 2) return -1;
    int64_t q = (algo == MN_ALGO_AUTO || algo == MN_ALGO_QGEMM) ? qg_ws_bytes(g, which) : 0;
    if (algo == MN_ALGO_AUTO) { const int64_t c1 = c1_ws_bytes(g, which); q = q > c1 ? q : c1; }
    int64_t m = 0;
    if (algo == MN_ALGO_AUTO || algo == MN_ALGO_MFMA) {
        if (which == 0 || which == 1) { FwdPlan pl; m = plan_fwd_view(g, which, &pl) ? pl.wp_floats * 4 : 0; }
        else { WgradPlan pl; m = plan_wgrad(g, &pl) ? pl.part_floats * 4 : 0; }
    }
    return q > m ? q : m;
}
extern "C" int mn_conv2d_first_supported(const mn_conv_geom* g, int which) {
    if (check_geom(g, "mn_conv2d_first_supported") != MN_OK) return 0;
    return c1_supported(g, which);
}
extern "C" int mn_conv2d_qgemm_supported(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, int which) {
    if (check_geom(g, "mn_conv2d_qgemm_supported") != MN_OK) return 0;
    return qg_supported(g, aq, wq, which);
}

static DirectParams make_direct(const mn_conv_geom* g, const Pro& pro) {
    DirectParams d;
    d.N = g->N; d.C = g->C; d.H = g->H; d.W = g->W; d.O = g->O; d.KH = g->KH; d.KW = g->KW; d.Sh = g->stride_h; d.Sw = g->stride_w;
    d.ph = g->pad_h; d.pw = g->pad_w; d.Dh = g->dil_h; d.Dw = g->dil_w; d.G = g->groups;
    d.Ho = out_dim(g->H, g->KH, g->stride_h, g->pad_h, g->dil_h); d.Wo = out_dim(g->W, g->KW, g->stride_w, g->pad_w, g->dil_w);
    d.Cg = g->C / g->groups; d.Mg = g->O / g->groups; d.pro = pro;
    return d;
}
template <int MT>
static void launch_fwd(const FwdPlan& pl, hipStream_t s) {
    hipLaunchKernelGGL(k_conv_mfma<MT>, dim3(pl.grid), dim3(256), pl.lds, s, pl.p);
}
static int run_fwd_plan(FwdPlan& pl, hipStream_t s, const char* what) {
    mn_set_last_kernel("k_conv_mfma<%d>", pl.MT);
    mn_prof_begin(s);
    switch (pl.MT) {
        case 1: launch_fwd<1>(pl, s); break;
        case 2: launch_fwd<2>(pl, s); break;
        case 4: launch_fwd<4>(pl, s); break;
        case 8: launch_fwd<8>(pl, s); break;
        default: MN_FAIL(MN_EINVAL, "%s: bad MT", what);
    }
    mn_prof_end(s);
    MN_CHECK_LAUNCH(what);
    return MN_OK;
}

extern "C" int mn_conv2d_fwd(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, const float* x, const float* w,
                             const float* bias, float* y, void* ws, int64_t ws_bytes, int algo, mn_stream_t stream) {
    int rc = check_geom(g, "mn_conv2d_fwd");
    if (rc) return rc;
    if (!x || !w || !y) MN_FAIL(MN_EINVAL, "mn_conv2d_fwd: null tensor");
    Pro pro;
    if ((rc = make_pro(aq, &pro, 0, "mn_conv2d_fwd"))) return rc;
    hipStream_t s = (hipStream_t)stream;
    {
        const double nx = (double)g->N * g->C * g->H * g->W, nw = (double)g->O * (g->C / g->groups) * g->KH * g->KW;
        const double ny = (double)g->N * g->O * out_dim(g->H, g->KH, g->stride_h, g->pad_h, g->dil_h) * out_dim(g->W, g->KW, g->stride_w, g->pad_w, g->dil_w);
        mn_prof_bytes((pro.mode == MN_ACTQ_SIGN8 ? 1.0 : 4.0) * nx + 4.0 * (ny + nw));
    }
    if (algo == MN_ALGO_QGEMM || (algo == MN_ALGO_AUTO && qg_supported(g, aq, wq, 0) && aligned16(x) && aligned16(y)))
        return qg_fwd(g, aq, wq, x, w, bias, y, ws, ws_bytes, s);
    if (algo == MN_ALGO_AUTO && pro.mode == MN_ACTQ_NONE && c1_supported(g, 0) && aligned16(y) && ws && ws_bytes >= c1_ws_bytes(g, 0))
        return c1_fwd(g, x, w, bias, y, ws, ws_bytes, s);          // un-quantised first layer: real fp32 operands
    if (g->in_shuffle > 1) MN_FAIL(MN_ENOTSUP, "mn_conv2d_fwd: in_shuffle is only available on the code-domain kernels");
    if (pro.mode == MN_ACTQ_SIGN8 || pro.mode == MN_ACTQ_CODE8) MN_FAIL(MN_ENOTSUP, "mn_conv2d_fwd: int8 sign / activation codes are only read by the code-domain kernels");
    FwdPlan pl;
    const int can = (algo != MN_ALGO_DIRECT) && plan_fwd_view(g, 0, &pl) && aligned16(x) && aligned16(y);
    if (algo == MN_ALGO_MFMA && !can) MN_FAIL(MN_ENOTSUP, "mn_conv2d_fwd: geometry not supported by the MFMA tiler");
    if (can) {
        if (!ws || ws_bytes < pl.wp_floats * 4 || !aligned16(ws)) MN_FAIL(MN_ENOSPC, "mn_conv2d_fwd: workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)pl.wp_floats * 4);
        float* wp = (float*)ws;
        hipLaunchKernelGGL(k_pack_weights, dim3(mn_grid_for(pl.wp_floats, 256, 1024)), dim3(256), 0, s, w, wp, g->groups, g->O / g->groups,
                           g->C / g->groups, g->KH, g->KW, pl.p.Kcp, pl.p.Mgpad, 0);
        pl.p.v.in = x; pl.p.wp = wp; pl.p.out = y; pl.p.bias = bias; pl.p.aux = nullptr; pl.p.epi = EPI_BIAS; pl.p.pro = pro; pl.p.ste = pro;
        return run_fwd_plan(pl, s, "mn_conv2d_fwd(mfma)");
    }
    DirectParams d = make_direct(g, pro);
    mn_set_last_kernel("k_conv_direct_fwd");
    const int64_t total = (int64_t)d.N * d.O * d.Ho * d.Wo;
    hipLaunchKernelGGL(k_conv_direct_fwd, dim3(mn_grid_for(total, 256, 65535)), dim3(256), 0, s, d, x, w, bias, y);
    MN_CHECK_LAUNCH("mn_conv2d_fwd(direct)");
    return MN_OK;
}

static int fwd_act_first(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq) {          // real operands on the first-layer kernels (an image, <= 76 taps)
    return (!aq || aq->mode == MN_ACTQ_NONE) && (!wq || wq->mode == MN_WQ_REAL) && g->in_shuffle <= 1 && c1_supported(g, 0);
}
extern "C" int64_t mn_conv2d_fwd_act_mm_count(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq) {
    if (!g || check_geom(g, "mn_conv2d_fwd_act_mm_count")) return 0;
    if (fwd_act_first(g, aq, wq)) return c1_fwd_mm_count(g);
    if (!qg_supported(g, aq, wq, 0)) return 0;
    return qg_fwd_act_mm_count(g);
}
extern "C" int mn_conv2d_fwd_act(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, const float* x, const float* w, const float* bias, float* y, int relu,
                                 float* mm, void* ws, int64_t ws_bytes, mn_stream_t stream) {
    int rc = check_geom(g, "mn_conv2d_fwd_act");
    if (rc) return rc;
    if (!x || !w || !y) MN_FAIL(MN_EINVAL, "mn_conv2d_fwd_act: null tensor");
    if (fwd_act_first(g, aq, wq) && aligned16(y)) {
        const double nx = (double)g->N * g->C * g->H * g->W, ny = (double)g->N * g->O * g->H * g->W;
        mn_prof_bytes(4.0 * (nx + ny));
        return c1_fwd_act(g, x, w, bias, y, relu, mm, ws, ws_bytes, (hipStream_t)stream);
    }
    if (!qg_fwd_act_mm_count(g) || !qg_supported(g, aq, wq, 0) || !aligned16(x) || !aligned16(y))
        MN_FAIL(MN_ENOTSUP, "mn_conv2d_fwd_act: pointwise code-domain layers and first-layer (image) convolutions only");
    {
        const double nx = (double)g->N * g->C * g->H * g->W, ny = (double)g->N * g->O * g->H * g->W;
        mn_prof_bytes(4.0 * (nx + ny));
    }
    return qg_fwd_act(g, aq, wq, x, w, bias, y, relu, mm, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int mn_conv2d_bwd_data_add_supported(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq) {
    if (!g || check_geom(g, "mn_conv2d_bwd_data_add_supported")) return 0;
    return qd_iao_dx_add_supported(g, aq, wq);
}
extern "C" int mn_conv2d_bwd_data(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, const float* gy, const float* w,
                                  const float* x, float* dx, void* ws, int64_t ws_bytes, int algo, mn_stream_t stream) {
    int rc = check_geom(g, "mn_conv2d_bwd_data");
    if (rc) return rc;
    if (!gy || !w || !dx) MN_FAIL(MN_EINVAL, "mn_conv2d_bwd_data: null tensor");
    Pro ste;
    if ((rc = make_pro(aq, &ste, 1, "mn_conv2d_bwd_data"))) return rc;
    if (ste.mode == MN_ACTQ_SIGN8 || ste.mode == MN_ACTQ_CODE8) ste.mode = MN_ACTQ_NONE;      // codes carry no STE here (it lives in mn_bnsign_bwd / mn_qa_bwd_*); x is not read
    if (ste.mode != MN_ACTQ_NONE && !x) MN_FAIL(MN_EINVAL, "mn_conv2d_bwd_data: x required for the clip-STE epilogue");
    hipStream_t s = (hipStream_t)stream;
    if (aq && aq->dx_add) {          // only the dense IAO kernel adds a tensor in its store: anything else must refuse, not drop it
        if ((algo != MN_ALGO_AUTO && algo != MN_ALGO_QGEMM) || !qd_iao_dx_add_supported(g, aq, wq) || !ws || ws_bytes < qd_iao_ws_bytes(g, 1) || !aligned16(gy) || !aligned16(dx) || !aligned16(x))
            MN_FAIL(MN_ENOTSUP, "mn_conv2d_bwd_data: mn_actq.dx_add is not available for this layer (mn_conv2d_bwd_data_add_supported)");
        return qd_iao_bwd_data(g, aq, wq, gy, w, x, dx, ws, ws_bytes, s);
    }
    {
        const double nx = (double)g->N * g->C * g->H * g->W, nw = (double)g->O * (g->C / g->groups) * g->KH * g->KW;
        const double ny = (double)g->N * g->O * out_dim(g->H, g->KH, g->stride_h, g->pad_h, g->dil_h) * out_dim(g->W, g->KW, g->stride_w, g->pad_w, g->dil_w);
        mn_prof_bytes(4.0 * (ny + nx + nw + (ste.mode != MN_ACTQ_NONE ? nx : 0.0)));
    }
    if (algo == MN_ALGO_QGEMM || (algo == MN_ALGO_AUTO && qg_supported(g, aq, wq, 1) && aligned16(gy) && aligned16(dx) &&
                                  (ste.mode == MN_ACTQ_NONE || aligned16(x))))
        return qg_bwd_data(g, aq, wq, gy, w, x, dx, ws, ws_bytes, s);
    if (g->in_shuffle > 1) MN_FAIL(MN_ENOTSUP, "mn_conv2d_bwd_data: in_shuffle is only available on the code-domain kernels");
    FwdPlan pl;
    const int can = (algo != MN_ALGO_DIRECT) && plan_fwd_view(g, 1, &pl) && aligned16(gy) && aligned16(dx) && (ste.mode == MN_ACTQ_NONE || aligned16(x));
    if (algo == MN_ALGO_MFMA && !can) MN_FAIL(MN_ENOTSUP, "mn_conv2d_bwd_data: geometry not supported by the MFMA tiler");
    if (can) {
        if (!ws || ws_bytes < pl.wp_floats * 4 || !aligned16(ws)) MN_FAIL(MN_ENOSPC, "mn_conv2d_bwd_data: workspace too small");
        float* wp = (float*)ws;
        hipLaunchKernelGGL(k_pack_weights, dim3(mn_grid_for(pl.wp_floats, 256, 1024)), dim3(256), 0, s, w, wp, g->groups, g->O / g->groups,
                           g->C / g->groups, g->KH, g->KW, pl.p.Kcp, pl.p.Mgpad, 1);
        Pro none; none.mode = MN_ACTQ_NONE; none.s = 1.f; none.qmin = none.qmax = 0.f; none.qp = nullptr;
        pl.p.v.in = gy; pl.p.wp = wp; pl.p.out = dx; pl.p.bias = nullptr; pl.p.aux = x; pl.p.pro = none; pl.p.ste = ste;
        pl.p.epi = ste.mode == MN_ACTQ_NONE ? EPI_PLAIN : EPI_STE;
        return run_fwd_plan(pl, s, "mn_conv2d_bwd_data(mfma)");
    }
    DirectParams d = make_direct(g, ste);
    mn_set_last_kernel("k_conv_direct_bwd_data");
    const int64_t total = (int64_t)d.N * d.C * d.H * d.W;
    hipLaunchKernelGGL(k_conv_direct_bwd_data, dim3(mn_grid_for(total, 256, 65535)), dim3(256), 0, s, d, gy, w, x, dx);
    MN_CHECK_LAUNCH("mn_conv2d_bwd_data(direct)");
    return MN_OK;
}

extern "C" int mn_conv2d_bnh_supported(const mn_conv_geom* g, const mn_wq* wq) {
    if (check_geom(g, "mn_conv2d_bnh_supported") != MN_OK) return 0;
    return pwd_supported(g, wq) && pws_wgrad_supported(g);
}
extern "C" int mn_conv2d_bwd_data_bnh(const mn_conv_geom* g, const mn_wq* wq, const float* da, const uint8_t* h, const float* chan, const float* sums,
                                      int training, const float* w, float* dx, void* ws, int64_t ws_bytes, mn_stream_t stream) {
    int rc = check_geom(g, "mn_conv2d_bwd_data_bnh");
    if (rc) return rc;
    if (!da || !h || !chan || !sums || !w || !dx) MN_FAIL(MN_EINVAL, "mn_conv2d_bwd_data_bnh: null tensor");
    return pwd_bwd_data_bnh(g, wq, da, h, chan, sums, training, w, dx, ws, ws_bytes, (hipStream_t)stream);
}
extern "C" int mn_conv2d_bwd_weight_bnh(const mn_conv_geom* g, const float* da, const uint8_t* h, const float* chan, const float* sums, int training,
                                        const int8_t* x, float* dw, float* dbias, void* ws, int64_t ws_bytes, mn_stream_t stream) {
    int rc = check_geom(g, "mn_conv2d_bwd_weight_bnh");
    if (rc) return rc;
    if (!da || !h || !chan || !sums || !x || !dw) MN_FAIL(MN_EINVAL, "mn_conv2d_bwd_weight_bnh: null tensor");
    return pws_bwd_weight_bnh(g, da, h, chan, sums, training, x, dw, dbias, ws, ws_bytes, (hipStream_t)stream);
}
// the same two behind a block whose output is MAX-POOLED (2x2 / stride 2): dpool = d loss / d pooled output [N][O][H/2][W/2], own = the block's sign output
extern "C" int mn_conv2d_bnh_pool_supported(const mn_conv_geom* g, const mn_wq* wq) {
    if (check_geom(g, "mn_conv2d_bnh_pool_supported") != MN_OK) return 0;
    return pwd_supported(g, wq) && pws_wgrad_staged(g) && !(g->H & 1) && !(g->W & 3);
}
extern "C" int mn_conv2d_bwd_data_bnh_pool(const mn_conv_geom* g, const mn_wq* wq, const float* dpool, const uint8_t* h, const int8_t* own, const float* chan,
                                           const float* sums, int training, const float* w, float* dx, void* ws, int64_t ws_bytes, mn_stream_t stream) {
    int rc = check_geom(g, "mn_conv2d_bwd_data_bnh_pool");
    if (rc) return rc;
    if (!dpool || !h || !own || !chan || !sums || !w || !dx) MN_FAIL(MN_EINVAL, "mn_conv2d_bwd_data_bnh_pool: null tensor");
    return pwd_bwd_data_bnh(g, wq, dpool, h, chan, sums, training, w, dx, ws, ws_bytes, (hipStream_t)stream, own);
}
extern "C" int mn_conv2d_bwd_weight_bnh_pool(const mn_conv_geom* g, const float* dpool, const uint8_t* h, const int8_t* own, const float* chan, const float* sums,
                                             int training, const int8_t* x, float* dw, float* dbias, void* ws, int64_t ws_bytes, mn_stream_t stream) {
    int rc = check_geom(g, "mn_conv2d_bwd_weight_bnh_pool");
    if (rc) return rc;
    if (!dpool || !h || !own || !chan || !sums || !x || !dw) MN_FAIL(MN_EINVAL, "mn_conv2d_bwd_weight_bnh_pool: null tensor");
    return pws_bwd_weight_bnh(g, dpool, h, chan, sums, training, x, dw, dbias, ws, ws_bytes, (hipStream_t)stream, own);
}
// both gradients of the block in one launch (qgemm_pwb.hip)
extern "C" int mn_conv2d_bwd_bnh_supported(const mn_conv_geom* g, const mn_wq* wq, int pooled) {
    if (check_geom(g, "mn_conv2d_bwd_bnh_supported") != MN_OK) return 0;
    return pwb_supported(g, wq, pooled);
}
extern "C" int64_t mn_conv2d_bwd_bnh_ws_bytes(const mn_conv_geom* g) {
    if (check_geom(g, "mn_conv2d_bwd_bnh_ws_bytes") != MN_OK) return -1;
    return pwb_ws_bytes(g);
}
extern "C" int mn_conv2d_bwd_bnh(const mn_conv_geom* g, const mn_wq* wq, const float* da, const uint8_t* h, const int8_t* own, const float* chan, const float* sums,
                                 int training, const float* w, const int8_t* x, float* dx, float* dw, float* dbias, void* ws, int64_t ws_bytes, mn_stream_t stream) {
    int rc = check_geom(g, "mn_conv2d_bwd_bnh");
    if (rc) return rc;
    if (!wq || !da || !h || !chan || !sums || !w || !x || !dx || !dw) MN_FAIL(MN_EINVAL, "mn_conv2d_bwd_bnh: null tensor");
    return pwb_bwd_bnh(g, wq, da, h, own, chan, sums, training, w, x, dx, dw, dbias, ws, ws_bytes, (hipStream_t)stream);
}
extern "C" int mn_conv2d_bwd_bnh_up_splits(const mn_conv_geom* g, const mn_wq* wq, int pooled, int64_t up_k) {
    if (check_geom(g, "mn_conv2d_bwd_bnh_up_splits") != MN_OK) return 0;
    if (up_k < 1 || up_k > 254 || !pwb_supported(g, wq, pooled)) return 0;          // (a stash byte of 255 marks "no element passes": the upstream conv's K must stay below)
    return pwb_up_splits(g);
}
extern "C" int mn_conv2d_bwd_bnh_up(const mn_conv_geom* g, const mn_wq* wq, const float* da, const uint8_t* h, const int8_t* own, const float* chan, const float* sums,
                                    int training, const float* w, const int8_t* x, float* dx, float* dw, float* dbias, void* ws, int64_t ws_bytes,
                                    const uint8_t* up_h, const float* up_chan, double* up_part, mn_stream_t stream) {
    int rc = check_geom(g, "mn_conv2d_bwd_bnh_up");
    if (rc) return rc;
    if (!wq || !da || !h || !chan || !sums || !w || !x || !dx || !dw || !up_h || !up_chan || !up_part) MN_FAIL(MN_EINVAL, "mn_conv2d_bwd_bnh_up: null tensor");
    return pwb_bwd_bnh_up(g, wq, da, h, own, chan, sums, training, w, x, dx, dw, dbias, ws, ws_bytes, up_h, up_chan, up_part, (hipStream_t)stream);
}
extern "C" int mn_conv2d_bwd_bnh_up9_splits(const mn_conv_geom* g, const mn_wq* wq, int64_t up_k) {
    if (check_geom(g, "mn_conv2d_bwd_bnh_up9_splits") != MN_OK) return 0;
    if (up_k < 1 || up_k > 254 || !pwb_supported(g, wq, 0)) return 0;
    return pwb_up9_splits(g);
}
extern "C" int mn_conv2d_bwd_bnh_up9(const mn_conv_geom* g, const mn_wq* wq, const float* da, const uint8_t* h, const float* chan, const float* sums, int training,
                                     const float* w, const int8_t* x, float* dx, float* dw, float* dbias, void* ws, int64_t ws_bytes, const uint8_t* up_h,
                                     const float* up_chan, double* up_part, mn_stream_t stream) {
    int rc = check_geom(g, "mn_conv2d_bwd_bnh_up9");
    if (rc) return rc;
    if (!wq || !da || !h || !chan || !sums || !w || !x || !dx || !dw || !up_h || !up_chan || !up_part) MN_FAIL(MN_EINVAL, "mn_conv2d_bwd_bnh_up9: null tensor");
    return pwb_bwd_bnh_up9(g, wq, da, h, chan, sums, training, w, x, dx, dw, dbias, ws, ws_bytes, up_h, up_chan, up_part, (hipStream_t)stream);
}
extern "C" int mn_conv2d_bwd_codes(const mn_conv_geom* g, const mn_wq* wq, const float* gy, const float* w, const void* x_codes, int x_bits, float* dx, float* dw,
                                   float* dbias, void* ws, int64_t ws_bytes, mn_stream_t stream) {
    int rc = check_geom(g, "mn_conv2d_bwd_codes");
    if (rc) return rc;
    if (!wq || !gy || !w || !x_codes || !dx || !dw || x_bits < 0 || x_bits > 8) MN_FAIL(MN_EINVAL, "mn_conv2d_bwd_codes: null tensor / bad bit width");
    return pwb_bwd_plain(g, wq, gy, w, x_codes, x_bits, dx, dw, dbias, ws, ws_bytes, (hipStream_t)stream);
}
extern "C" int mn_conv2d_bwd_qa(const mn_conv_geom* g, const mn_wq* wq, const float* dq, const void* stash, int stash_bits, const float* chan, const float* sums,
                                int out_bits, int quant, int training, const float* w, const uint8_t* x_codes, int x_bits, float* dx, float* dw, float* dbias, void* ws,
                                int64_t ws_bytes, mn_stream_t stream) {
    int rc = check_geom(g, "mn_conv2d_bwd_qa");
    if (rc) return rc;
    if (!wq || !dq || !stash || !chan || !sums || !w || !x_codes || !dx || !dw) MN_FAIL(MN_EINVAL, "mn_conv2d_bwd_qa: null tensor");
    return pwb_bwd_qa(g, wq, dq, stash, stash_bits, chan, sums, out_bits, quant, training, w, x_codes, x_bits, dx, dw, dbias, ws, ws_bytes, (hipStream_t)stream);
}
extern "C" int mn_conv2d_bwd_codes_up(const mn_conv_geom* g, const mn_wq* wq, const float* gy, const float* w, const void* x_codes, int x_bits, float* dx, float* dw,
                                      float* dbias, void* ws, int64_t ws_bytes, const void* up_stash, const float* up_chan, int up_quant, double* up_part,
                                      mn_stream_t stream) {
    int rc = check_geom(g, "mn_conv2d_bwd_codes_up");
    if (rc) return rc;
    if (!wq || !gy || !w || !x_codes || !dx || !dw || !up_stash || !up_chan || !up_part) MN_FAIL(MN_EINVAL, "mn_conv2d_bwd_codes_up: null tensor");
    return pwb_bwd_plain_up(g, wq, gy, w, x_codes, x_bits, dx, dw, dbias, ws, ws_bytes, up_stash, up_chan, up_quant, up_part, (hipStream_t)stream);
}
extern "C" int mn_conv2d_bwd_qa_up(const mn_conv_geom* g, const mn_wq* wq, const float* dq, const void* stash, const float* chan, const float* sums, int out_bits,
                                   int quant, int training, const float* w, const uint8_t* x_codes, int x_bits, float* dx, float* dw, float* dbias, void* ws,
                                   int64_t ws_bytes, const void* up_stash, const float* up_chan, int up_quant, double* up_part, mn_stream_t stream) {
    int rc = check_geom(g, "mn_conv2d_bwd_qa_up");
    if (rc) return rc;
    if (!wq || !dq || !stash || !chan || !sums || !w || !x_codes || !dx || !dw || !up_stash || !up_chan || !up_part) MN_FAIL(MN_EINVAL, "mn_conv2d_bwd_qa_up: null tensor");
    return pwb_bwd_qa_up(g, wq, dq, stash, chan, sums, out_bits, quant, training, w, x_codes, x_bits, dx, dw, dbias, ws, ws_bytes, up_stash, up_chan, up_quant, up_part,
                         (hipStream_t)stream);
}
extern "C" int mn_conv2d_bwd_weight_first_bn(const mn_conv_geom* g, const float* da, const float* y, const float* save, const float* gamma, const float* beta,
                                             const float* sums, int training, const float* x, float* dw, float* dbias, void* ws, int64_t ws_bytes,
                                             mn_stream_t stream) {
    int rc = check_geom(g, "mn_conv2d_bwd_weight_first_bn");
    if (rc) return rc;
    if (!x || !dw) MN_FAIL(MN_EINVAL, "mn_conv2d_bwd_weight_first_bn: null tensor");
    if (!c1_supported(g, 2)) MN_FAIL(MN_ENOTSUP, "mn_conv2d_bwd_weight_first_bn: geometry not covered by the first-layer kernels");
    return c1_bwd_weight_bn(g, nullptr, da, y, save, gamma, beta, sums, training, x, dw, dbias, ws, ws_bytes, (hipStream_t)stream);
}
extern "C" int mn_conv2d_bwd_weight_first_qa(const mn_conv_geom* g, const float* dq, const float* y, const float* chan, const float* sums, int a_bits,
                                             int quant, int training, const float* x, float* dw, float* dbias, void* ws, int64_t ws_bytes, mn_stream_t stream) {
    int rc = check_geom(g, "mn_conv2d_bwd_weight_first_qa");
    if (rc) return rc;
    if (!x || !dw) MN_FAIL(MN_EINVAL, "mn_conv2d_bwd_weight_first_qa: null tensor");
    if (!c1_supported(g, 2)) MN_FAIL(MN_ENOTSUP, "mn_conv2d_bwd_weight_first_qa: geometry not covered by the first-layer kernels");
    return c1_bwd_weight_qa(g, dq, y, chan, quant, a_bits, sums, training, x, dw, dbias, ws, ws_bytes, (hipStream_t)stream);
}
extern "C" int64_t mn_conv2d_first_xgram_ws_bytes(const mn_conv_geom* g) {
    if (!g || check_geom(g, "mn_conv2d_first_xgram_ws_bytes")) return 0;
    return c1_xgram_ws_bytes(g);
}
extern "C" int mn_conv2d_first_xgram(const mn_conv_geom* g, const float* x, double* gram, void* ws, int64_t ws_bytes, mn_stream_t stream) {
    int rc = check_geom(g, "mn_conv2d_first_xgram");
    if (rc) return rc;
    if (!x || !gram || (((uintptr_t)gram) & 7)) MN_FAIL(MN_EINVAL, "mn_conv2d_first_xgram: null / misaligned tensor");
    return c1_xgram(g, x, gram, ws, ws_bytes, (hipStream_t)stream);
}
extern "C" int mn_conv2d_first_bnact_fwd(const mn_conv_geom* g, const float* x, const float* w, const float* bias, const float* save, const float* gamma,
                                        const float* beta, int act, int a_bits, void* codes, uint8_t* mask4, mn_stream_t stream) {
    int rc = check_geom(g, "mn_conv2d_first_bnact_fwd");
    if (rc) return rc;
    if (!c1_supported(g, 0) || !c1_supported(g, 2)) MN_FAIL(MN_ENOTSUP, "mn_conv2d_first_bnact_fwd: geometry not covered by the first-layer kernels");
    return c1_fwd_bnact(g, x, w, bias, save, gamma, beta, act, a_bits, codes, mask4, (hipStream_t)stream);
}
extern "C" int mn_conv2d_bwd_first_mask_gram(const mn_conv_geom* g, const float* da, const uint8_t* mask4, int quant, const float* save, const float* gamma,
                                             const float* w, const float* bias, const double* gram, const float* x, float* dw, float* dbias, float* dgamma,
                                             float* dbeta, void* ws, int64_t ws_bytes, mn_stream_t stream) {
    int rc = check_geom(g, "mn_conv2d_bwd_first_mask_gram");
    if (rc) return rc;
    return c1_bwd_first_mask(g, da, mask4, quant, save, gamma, w, bias, gram, x, dw, dbias, dgamma, dbeta, ws, ws_bytes, (hipStream_t)stream);
}
extern "C" int mn_conv2d_first_gram_bnstats(const mn_conv_geom* g, const float* w, const float* bias, const double* gram, float eps, float momentum,
                                           float* running_mean, float* running_var, float* save, mn_stream_t stream) {
    int rc = check_geom(g, "mn_conv2d_first_gram_bnstats");
    if (rc) return rc;
    return c1_gram_bnstats(g, w, bias, gram, eps, momentum, running_mean, running_var, save, (hipStream_t)stream);
}
extern "C" int mn_conv2d_bwd_first_bn_gram(const mn_conv_geom* g, const float* da, const float* y, const float* save, const float* gamma, const float* beta,
                                           const float* w, const float* bias, const double* gram, const float* x, float* dw, float* dbias, float* dgamma,
                                           float* dbeta, void* ws, int64_t ws_bytes, mn_stream_t stream) {
    int rc = check_geom(g, "mn_conv2d_bwd_first_bn_gram");
    if (rc) return rc;
    if (!save || !gamma || !beta) MN_FAIL(MN_EINVAL, "mn_conv2d_bwd_first_bn_gram: null tensor");
    return c1_bwd_first_gram(g, da, y, save, gamma, beta, nullptr, 0, 0, w, bias, gram, x, dw, dbias, dgamma, dbeta, ws, ws_bytes, (hipStream_t)stream);
}
extern "C" int mn_conv2d_bwd_first_qa_gram(const mn_conv_geom* g, const float* dq, const float* y, const float* chan, int a_bits, int quant, const float* w,
                                           const float* bias, const double* gram, const float* x, float* dw, float* dbias, float* dgamma, float* dbeta, void* ws,
                                           int64_t ws_bytes, mn_stream_t stream) {
    int rc = check_geom(g, "mn_conv2d_bwd_first_qa_gram");
    if (rc) return rc;
    if (!chan) MN_FAIL(MN_EINVAL, "mn_conv2d_bwd_first_qa_gram: null tensor");
    return c1_bwd_first_gram(g, dq, y, nullptr, nullptr, nullptr, chan, quant, a_bits, w, bias, gram, x, dw, dbias, dgamma, dbeta, ws, ws_bytes, (hipStream_t)stream);
}
extern "C" int mn_conv2d_bwd_weight(const mn_conv_geom* g, const mn_actq* aq, const float* gy, const float* x, float* dw,
                                    float* dbias, void* ws, int64_t ws_bytes, int algo, mn_stream_t stream) {
    int rc = check_geom(g, "mn_conv2d_bwd_weight");
    if (rc) return rc;
    if (!gy || !x || !dw) MN_FAIL(MN_EINVAL, "mn_conv2d_bwd_weight: null tensor");
    Pro pro;
    if ((rc = make_pro(aq, &pro, 0, "mn_conv2d_bwd_weight"))) return rc;
    hipStream_t s = (hipStream_t)stream;
    {
        const double nx = (double)g->N * g->C * g->H * g->W, nw = (double)g->O * (g->C / g->groups) * g->KH * g->KW;
        const double ny = (double)g->N * g->O * out_dim(g->H, g->KH, g->stride_h, g->pad_h, g->dil_h) * out_dim(g->W, g->KW, g->stride_w, g->pad_w, g->dil_w);
        mn_prof_bytes((pro.mode == MN_ACTQ_SIGN8 ? 1.0 : 4.0) * nx + 4.0 * (ny + nw));
    }
    const int Ho = out_dim(g->H, g->KH, g->stride_h, g->pad_h, g->dil_h), Wo = out_dim(g->W, g->KW, g->stride_w, g->pad_w, g->dil_w);
    if (algo == MN_ALGO_AUTO && pro.mode == MN_ACTQ_NONE && c1_supported(g, 2) && aligned16(gy) && ws && ws_bytes >= c1_ws_bytes(g, 2))
        return c1_bwd_weight(g, gy, x, dw, dbias, ws, ws_bytes, s);   // first layer: K = Cin*KH*KW <= 76, exact fp32 MFMA
    if (algo == MN_ALGO_QGEMM || (algo == MN_ALGO_AUTO && qg_supported(g, aq, nullptr, 2) && (aligned16(x) || pro.mode == MN_ACTQ_CODE8) && aligned16(gy)))
        return qg_bwd_weight(g, aq, gy, x, dw, dbias, ws, ws_bytes, s);
    if (g->in_shuffle > 1) MN_FAIL(MN_ENOTSUP, "mn_conv2d_bwd_weight: in_shuffle is only available on the code-domain kernels");
    if (pro.mode == MN_ACTQ_SIGN8 || pro.mode == MN_ACTQ_CODE8) MN_FAIL(MN_ENOTSUP, "mn_conv2d_bwd_weight: int8 sign / activation codes are only read by the code-domain kernels");
    WgradPlan pl;
    const int can = (algo != MN_ALGO_DIRECT) && plan_wgrad(g, &pl) && aligned16(x) && aligned16(gy);
    if (algo == MN_ALGO_MFMA && !can) MN_FAIL(MN_ENOTSUP, "mn_conv2d_bwd_weight: geometry not supported by the MFMA tiler");
    if (can) {
        if (!ws || ws_bytes < pl.part_floats * 4 || !aligned16(ws)) MN_FAIL(MN_ENOSPC, "mn_conv2d_bwd_weight: workspace too small");
        pl.p.v.in = x; pl.p.gy = gy; pl.p.part = (float*)ws; pl.p.pro = pro;
        mn_set_last_kernel(pl.MPW == 2 ? "k_wgrad_mfma<2, 9>" : "k_wgrad_mfma<1, 13>");
        mn_prof_begin(s);
        if (pl.MPW == 2) hipLaunchKernelGGL((k_wgrad_mfma<2, 9>), dim3(pl.grid), dim3(256), pl.lds, s, pl.p);
        else hipLaunchKernelGGL((k_wgrad_mfma<1, 13>), dim3(pl.grid), dim3(256), pl.lds, s, pl.p);
        mn_prof_end(s);
        const int64_t total = (int64_t)g->O * (g->C / g->groups) * g->KH * g->KW;
        hipLaunchKernelGGL(k_wgrad_reduce, dim3(mn_grid_for(total, 256, 2048)), dim3(256), 0, s, (const float*)ws, dw, pl.p.Z, g->groups,
                           g->O / g->groups, g->C / g->groups, g->KH, g->KW, pl.p.Mgw, pl.p.Cgw);
    } else {
        DirectParams d = make_direct(g, pro);
        mn_set_last_kernel("k_conv_direct_bwd_weight");
        const int64_t total = (int64_t)g->O * d.Cg * g->KH * g->KW;
        if (total > 0x7fffffff) MN_FAIL(MN_ENOTSUP, "mn_conv2d_bwd_weight(direct): too many weights");
        hipLaunchKernelGGL(k_conv_direct_bwd_weight, dim3((unsigned)total), dim3(256), 0, s, d, gy, x, dw);
    }
    if (dbias) hipLaunchKernelGGL(k_bias_grad, dim3((unsigned)g->O), dim3(256), 0, s, gy, dbias, g->N, g->O, Ho * Wo);
    MN_CHECK_LAUNCH("mn_conv2d_bwd_weight");
    return MN_OK;
}
