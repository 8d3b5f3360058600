// Code-domain k x k convolution (any kernel size / stride / dilation / groups) for gfx950: forward and backward-data.
//
// Same arithmetic as the pointwise kernels (qgemm_kernels.hip): integer codes, exact in bf16, contracted on
// v_mfma_f32_16x16x32_bf16; a real-valued streamed operand is written as three exact bf16 terms and all-zero terms are
// skipped.  What differs is the data path: the taps re-read every input pixel KH*KW times, so the input patch of a
// 256-pixel output tile is staged ONCE in LDS -- transposed on the way in from NCHW rows (coalesced float4 loads of 8
// channels x 4 pixels per thread) to channel-innermost records [position][term][CC channels] of bf16 codes, so that the B
// fragment of (pixel, tap) is one aligned ds_read_b128 of 8 consecutive channels.  Records are padded by 16 B: the 16
// pixels of a fragment read then fall on distinct bank quads.  K is ordered tap-major inside a channel chunk
// (k = tap*CC + c), the weight codes are packed in the same order and staged per chunk as A fragments.
// Pixel <-> MFMA column permutation as in the pointwise kernel: MFMA q takes pixel 4j+q from lane j, so the epilogue stores
// float4 = 4 consecutive pixels per out-channel (256-B runs per 16 lanes).
// Backward-data = the same kernel on gy with tap-flipped, transposed codes (stride 1 only; strided convolutions fall back
// to the fp32-MFMA kernels), gy pre-scaled by the per-channel weight scale and split in three terms, clip-STE epilogue.
#include "qgemm_dev.h"

struct KkParams {
    const float* in;          // x (fwd) / gy (bwd-data)   [N][G*Kc][Hin][Win]
    float* out;               // y / dx                     [N][G*Mr][Ho][Wo]
    uint8_t* out8;            // QG_EPI_H8: the byte stash, same shape
    const uint16_t* wc;       // codes [G][Mpad][T][Cgp]
    const float* rowscale;    // [G][Mpad] or null
    const float* kscale;      // [G][Cgp]  or null (bwd-data: weight scale of the contraction channel)
    const float* bias;
    const float* aux;         // bwd-data STE: x
    Pro pro, ste;
    int N, Cin_total, Hin, Win, Kc, G, Cout_total, Ho, Wo, Mr;
    int KH, KW, T, Sh, Sw, Dh, Dw, ph, pw;
    int Cgp, CC, cc_shift, nck, KS, LDW, PSB;      // PSB: bytes per patch position record
    int NI, TR, tpi, PR, PWp, npos, num_ptiles, Mpad, num_mblk, epi, W4;
    FastDiv fd_wo, fd_tr, fd_tpi, fd_pwp, fd_pr, fd_w4, fd_o8;
    float ascale;
    ChanMap in_map, out_map;   // channel shuffle folded into the input (fwd) / output (bwd-data) addressing
};

// PL = number of term planes kept in LDS.  XMODE NONE (a real-valued stream): PL = 3 stores the three terms side by side
// (backward-data: gy always needs them); PL = 1 stores one term at a time -- the forward of wbwtab, whose +-1 activations
// are exact in the first term: a block re-stages and re-contracts terms 2 and 3 only if it met an inexact element.
template <int NT, int XMODE, int PL>
__global__ __launch_bounds__(256, 2) void k_kk(const KkParams p) {
    HIP_DYNAMIC_SHARED(float, smem)
    constexpr int MB = 16 * NT;
    constexpr int NTERM = PL;
    constexpr bool ADAPT = (XMODE == MN_ACTQ_NONE) && PL == 1;
    char* patch = reinterpret_cast<char*>(smem);                              // [npos][PSB]
    uint16_t* wsm = reinterpret_cast<uint16_t*>(patch + (size_t)p.npos * p.PSB);   // [MB][LDW]
    float* rs = reinterpret_cast<float*>(wsm + MB * p.LDW);
    float* bs = rs + MB;
    int* toff = reinterpret_cast<int*>(bs + MB);                             // [T] byte offset of a tap inside the patch
    int* tflag = toff + p.T;                                                 // [nck][2] any non-zero second / third term in the chunk's patch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kg = lane >> 4;

    uint32_t b = blockIdx.x;
    const int pt = b % p.num_ptiles; b /= p.num_ptiles;
    const int mblk = b % p.num_mblk;
    const int g = b / p.num_mblk;
    const uint32_t timg = fd_div(pt, p.fd_tpi);
    const int n0 = (int)timg * p.NI;
    const int oh0 = (pt - (int)timg * p.tpi) * p.TR;
    const int row0 = oh0 * p.Sh - p.ph;

    float sc = 1.f, zp = 0.f;
    if (XMODE == MN_ACTQ_IAO) { sc = p.pro.qp[0]; zp = p.pro.qp[1]; }
    {
        float as = p.ascale;
        if (XMODE == MN_ACTQ_IAO) as = sc;
        for (int i = tid; i < MB; i += 256) {
            const int m = mblk * MB + i;
            rs[i] = p.rowscale ? p.rowscale[g * p.Mpad + m] * as : 1.f;
            bs[i] = (p.bias && m < p.Mr) ? p.bias[g * p.Mr + m] : 0.f;
        }
        for (int t = tid; t < p.T; t += 256) toff[t] = ((t / p.KW) * p.Dh * p.PWp + (t % p.KW) * p.Dw) * p.PSB;
        if (XMODE == MN_ACTQ_NONE) for (int t = tid; t < 2 * p.nck; t += 256) tflag[t] = 0;
    }

    // patch byte offset of tap (0,0) of this lane's four pixels
    int pos0[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int pix = wave * 64 + 4 * j + q;
        const uint32_t fr = fd_div(pix, p.fd_wo);
        const int ocol = pix - fr * p.Wo;
        const uint32_t img = fd_div(fr, p.fd_tr);
        const int orow = fr - img * p.TR;
        pos0[q] = (((int)img * p.PR + orow * p.Sh) * p.PWp + ocol * p.Sw) * p.PSB;
    }

    f32x4 acc[4][NT];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[q][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int o8n = p.CC >> 3;
    const int64_t plane = (int64_t)p.Hin * p.Win;
    for (int ck = 0; ck < p.nck; ++ck) {
        __syncthreads();                             // the previous chunk's fragments are consumed (first pass: flags / tables visible)
        // ---- weights of this chunk: wsm[row][k], k = tap*CC + cc
        {
            const int k8n = (p.KS * 32) >> 3;
            const uint16_t* wg = p.wc + ((int64_t)g * p.Mpad + mblk * MB) * p.T * p.Cgp + ck * p.CC;
            for (int q = tid; q < MB * k8n; q += 256) {
                const int row = q / k8n, k8 = q - row * k8n;
                const int k = k8 * 8, tap = k >> p.cc_shift, cc = k & (p.CC - 1);
                u32x4 v = u32x4{0u, 0u, 0u, 0u};
                if (tap < p.T) v = *reinterpret_cast<const u32x4*>(wg + ((int64_t)row * p.T + tap) * p.Cgp + cc);
                *reinterpret_cast<u32x4*>(wsm + row * p.LDW + k) = v;
            }
        }
        // ---- zero the halo records (zero padding is code 0 in every scheme)
        {
            const int nrec = p.npos * o8n * NTERM;
            for (int q = tid; q < nrec; q += 256) {
                const int sub = q % (o8n * NTERM);
                const int pos = q / (o8n * NTERM);
                const uint32_t t1 = fd_div(pos, p.fd_pwp);
                const int pcol = pos - t1 * p.PWp;
                const uint32_t slot = fd_div(t1, p.fd_pr);
                const int prow = t1 - slot * p.PR;
                const int ir = row0 + prow, ic = pcol - p.pw;
                if (ir < 0 || ir >= p.Hin || ic < 0 || ic >= p.Win || n0 + (int)slot >= p.N)
                    *reinterpret_cast<u32x4*>(patch + (size_t)pos * p.PSB + sub * 16) = u32x4{0u, 0u, 0u, 0u};
            }
        }
        // ---- stage the interior: item = (slot, channel octet, patch row, input column quad)
        auto stage_interior = [&](int term) {
            const int nitem = p.NI * o8n * p.PR * p.W4;
            unsigned any1 = 0u, any2 = 0u;
            for (int it = tid; it < nitem; it += 256) {
                const uint32_t t1 = fd_div(it, p.fd_w4);
                const int iq = it - t1 * p.W4;
                const uint32_t t2 = fd_div(t1, p.fd_pr);
                const int prow = t1 - t2 * p.PR;
                const uint32_t slot = fd_div(t2, p.fd_o8);
                const int o8 = t2 - slot * o8n;
                const int ir = row0 + prow, n = n0 + (int)slot;
                if (ir < 0 || ir >= p.Hin || n >= p.N) continue;
                const int c0 = ck * p.CC + o8 * 8;
                const int64_t soff = (int64_t)n * p.Cin_total * plane + (int64_t)ir * p.Win + iq * 4;
                float v[8][4];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (c0 + jj < p.Kc) f = mn_ld_x4<XMODE>(p.in, soff + (int64_t)chan_phys(p.in_map, g * p.Kc + c0 + jj) * plane);
                    v[jj][0] = f.x; v[jj][1] = f.y; v[jj][2] = f.z; v[jj][3] = f.w;
                }
                if (XMODE == MN_ACTQ_NONE && p.kscale) {
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) {
                        const float ksv = p.kscale[g * p.Cgp + c0 + jj];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[jj][e] *= ksv;
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int pcol = iq * 4 + e + p.pw;
                    if (pcol >= p.PWp) continue;
                    char* rec = patch + (size_t)(((int)slot * p.PR + prow) * p.PWp + pcol) * p.PSB + o8 * 16;
                    if (NTERM == 1) {
                        float c8[8];
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) {
                            c8[jj] = act_code<XMODE>(v[jj][e], p.pro, sc, zp);
                            if (ADAPT) {          // term 0: the bf16 head (packing truncates); terms 1, 2: the remainders
                                float r = c8[jj] - mn_bf16_head(c8[jj]);
                                if (term == 0) any1 |= mn_f2u(r) << 1;
                                else { if (term == 2) r = r - mn_bf16_head(r); c8[jj] = r; }
                            }
                        }
                        *reinterpret_cast<u32x4*>(rec) = u32x4{mn_pack_bf16x2(c8[0], c8[1]), mn_pack_bf16x2(c8[2], c8[3]),
                                                              mn_pack_bf16x2(c8[4], c8[5]), mn_pack_bf16x2(c8[6], c8[7])};
                    } else {
                        float t0[8], t1_[8], t2_[8];
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) {
                            const float x0 = v[jj][e];
                            t0[jj] = mn_bf16_head(x0);
                            const float r1 = x0 - t0[jj];
                            t1_[jj] = mn_bf16_head(r1);
                            t2_[jj] = r1 - t1_[jj];
                            any1 |= mn_f2u(r1) << 1;
                            any2 |= mn_f2u(t2_[jj]) << 1;
                        }
                        *reinterpret_cast<u32x4*>(rec) = u32x4{mn_pack_bf16x2(t0[0], t0[1]), mn_pack_bf16x2(t0[2], t0[3]),
                                                              mn_pack_bf16x2(t0[4], t0[5]), mn_pack_bf16x2(t0[6], t0[7])};
                        *reinterpret_cast<u32x4*>(rec + p.CC * 2) = u32x4{mn_pack_bf16x2(t1_[0], t1_[1]), mn_pack_bf16x2(t1_[2], t1_[3]),
                                                                         mn_pack_bf16x2(t1_[4], t1_[5]), mn_pack_bf16x2(t1_[6], t1_[7])};
                        *reinterpret_cast<u32x4*>(rec + p.CC * 4) = u32x4{mn_pack_bf16x2(t2_[0], t2_[1]), mn_pack_bf16x2(t2_[2], t2_[3]),
                                                                         mn_pack_bf16x2(t2_[4], t2_[5]), mn_pack_bf16x2(t2_[6], t2_[7])};
                    }
                }
            }
            if (NTERM == 3 || (ADAPT && term == 0)) {
                if (any1) tflag[2 * ck] = 1;
                if (any2) tflag[2 * ck + 1] = 1;
            }
        };
        stage_interior(0);
        __syncthreads();
        // NTERM == 3 (backward-data: a real-valued gradient) contracts all three planes unconditionally.  An earlier version skipped the second /
        // third plane when a chunk's flags said they were all zero; with run-time flags the kernel was intermittently wrong under load at batch 256
        // (tests/test_gpu_determinism.py), and a real-valued gradient practically never has an all-zero remainder plane anyway.
        constexpr int use1 = NTERM == 3, use2 = NTERM == 3;
        // ---- contraction over this chunk's K = T*CC
        auto contract = [&]() {
        for (int ks = 0; ks < p.KS; ++ks) {
            const int kl = ks * 32 + kg * 8;
            const int tap = kl >> p.cc_shift, choff = (kl & (p.CC - 1)) * 2;
            const bool tv = tap < p.T;
            const int to = tv ? toff[tap] + choff : 0;
            u32x4 b0[4], b1[4], b2[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const char* rec = patch + pos0[q] + to;
                b0[q] = tv ? *reinterpret_cast<const u32x4*>(rec) : u32x4{0u, 0u, 0u, 0u};
                if (NTERM == 3) {
                    b1[q] = (tv && use1) ? *reinterpret_cast<const u32x4*>(rec + p.CC * 2) : u32x4{0u, 0u, 0u, 0u};
                    b2[q] = (tv && use2) ? *reinterpret_cast<const u32x4*>(rec + p.CC * 4) : u32x4{0u, 0u, 0u, 0u};
                }
            }
            const uint16_t* wk = wsm + j * p.LDW + kl;
            // term-outer: 4*NT independent accumulators between two MFMAs on the same one (a dependent MFMA stalls the issue)
            u32x4 av[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) av[t] = *reinterpret_cast<const u32x4*>(wk + t * 16 * p.LDW);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q][t] = mn_mfma_bf16(av[t], b0[q], acc[q][t]);
            if (NTERM == 3) {
                if (use1) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[q][t] = mn_mfma_bf16(av[t], b1[q], acc[q][t]);
                }
                if (use2) {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[q][t] = mn_mfma_bf16(av[t], b2[q], acc[q][t]);
                }
            }
        }
        };
        contract();
        if (ADAPT && tflag[2 * ck]) {          // block-uniform: real-valued x, add the contributions of its second and third term
            for (int term = 1; term <= 2; ++term) {
                __syncthreads();
                stage_interior(term);
                __syncthreads();
                contract();
            }
        }
    }

    // ---- epilogue: lane (j, kg) holds out-channels 4kg..4kg+3 of pixels 4j..4j+3 of the wave's 64-pixel span
    float ste_sc = 1.f, ste_zp = 0.f, ste_lo = 0.f, ste_hi = 0.f;
    if (p.epi == QG_EPI_STE && p.ste.mode == MN_ACTQ_IAO) { ste_sc = p.ste.qp[0]; ste_zp = p.ste.qp[1]; ste_lo = p.ste.qp[2]; ste_hi = p.ste.qp[3]; }
    const int pix = wave * 64 + 4 * j;
    const uint32_t fr = fd_div(pix, p.fd_wo);
    const int ocol = pix - fr * p.Wo;
    const uint32_t img = fd_div(fr, p.fd_tr);
    const int orow = fr - img * p.TR;
    const int n = n0 + (int)img;
    if (n >= p.N) return;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ml = t * 16 + kg * 4 + r;
            const int m = mblk * MB + ml;
            if (m >= p.Mr) continue;
            const int64_t off = (((int64_t)n * p.Cout_total + chan_phys(p.out_map, g * p.Mr + m)) * p.Ho + oh0 + orow) * p.Wo + ocol;
            float o0 = acc[0][t][r], o1 = acc[1][t][r], o2 = acc[2][t][r], o3 = acc[3][t][r];
            if (p.epi == QG_EPI_H8) {
                // acc has the parity of the non-zero weights that meet a non-zero input: per pixel class at the image border (3x3, padding 1;
                // p.bias = nnz9 [9][O], stash_nnz_quad in common.h): (acc + nnz) / 2 is an exact integer in [0, K]
                const int ch = g * p.Mr + m, rowa = oh0 + orow;
                const float* nb = p.bias + (rowa == 0 ? 0 : (rowa == p.Ho - 1 ? 6 : 3)) * (int64_t)p.Cout_total + ch;
                const float m1 = nb[p.Cout_total];
                const float b0 = ocol == 0 ? nb[0] : m1, b3 = ocol + 4 == p.Wo ? nb[2 * (int64_t)p.Cout_total] : m1;
                *reinterpret_cast<uint32_t*>(p.out8 + off) = (uint32_t)((o0 + b0) * 0.5f) | ((uint32_t)((o1 + m1) * 0.5f) << 8) |
                                                             ((uint32_t)((o2 + m1) * 0.5f) << 16) | ((uint32_t)((o3 + b3) * 0.5f) << 24);
                continue;
            }
            if (p.epi == QG_EPI_SCALE_BIAS) {
                const float a_ = rs[ml], b_ = bs[ml];
                o0 = o0 * a_ + b_; o1 = o1 * a_ + b_; o2 = o2 * a_ + b_; o3 = o3 * a_ + b_;
            } else if (p.epi == QG_EPI_STE) {
                const float4 xv = *reinterpret_cast<const float4*>(p.aux + off);
                if (p.ste.mode == MN_ACTQ_DOREFA) {
                    o0 = dorefa_act_grad(o0, xv.x, p.ste.s); o1 = dorefa_act_grad(o1, xv.y, p.ste.s);
                    o2 = dorefa_act_grad(o2, xv.z, p.ste.s); o3 = dorefa_act_grad(o3, xv.w, p.ste.s);
                } else if (p.ste.mode == MN_ACTQ_IAO) {
                    o0 = iao_fq_grad(o0, xv.x, ste_sc, ste_zp, ste_lo, ste_hi, p.ste.qmin, p.ste.qmax);
                    o1 = iao_fq_grad(o1, xv.y, ste_sc, ste_zp, ste_lo, ste_hi, p.ste.qmin, p.ste.qmax);
                    o2 = iao_fq_grad(o2, xv.z, ste_sc, ste_zp, ste_lo, ste_hi, p.ste.qmin, p.ste.qmax);
                    o3 = iao_fq_grad(o3, xv.w, ste_sc, ste_zp, ste_lo, ste_hi, p.ste.qmin, p.ste.qmax);
                }
            }
            *reinterpret_cast<float4*>(p.out + off) = make_float4(o0, o1, o2, o3);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
static const size_t KK_LDS_CAP = 128 * 1024;   // one block per CU in the worst case (large kernels on small images); typical layers use 25-60 KB
static int kk_out_dim(int in, int k, int s, int pd, int d) { return (in + 2 * pd - d * (k - 1) - 1) / s + 1; }

struct KkPlan {
    KkParams p;
    PackParams pk;
    int NT, xmode, planes;
    size_t lds;
    int grid, pack_grid;
    int64_t off_codes, off_scale, ws_bytes;
};
// which: 0 forward, 1 backward-data (as a forward-style contraction over gy with flipped taps)
static int plan_kk(const mn_conv_geom* g, int which, int xmode, KkPlan* pl) {
    KkParams& p = pl->p;
    const int Ho = kk_out_dim(g->H, g->KH, g->stride_h, g->pad_h, g->dil_h), Wo = kk_out_dim(g->W, g->KW, g->stride_w, g->pad_w, g->dil_w);
    const int Cg = g->C / g->groups, Mg = g->O / g->groups;
    p.N = g->N; p.G = g->groups; p.KH = g->KH; p.KW = g->KW; p.T = g->KH * g->KW; p.Dh = g->dil_h; p.Dw = g->dil_w;
    if (g->in_shuffle > 1 && g->C % g->in_shuffle) return 0;
    p.in_map = make_chanmap(which == 0 ? g->in_shuffle : 0, g->C);
    p.out_map = make_chanmap(which == 1 ? g->in_shuffle : 0, g->C);
    if (which == 0) {
        p.Cin_total = g->C; p.Hin = g->H; p.Win = g->W; p.Kc = Cg; p.Cout_total = g->O; p.Ho = Ho; p.Wo = Wo; p.Mr = Mg;
        p.Sh = g->stride_h; p.Sw = g->stride_w; p.ph = g->pad_h; p.pw = g->pad_w;
    } else {
        if (g->stride_h != 1 || g->stride_w != 1) return 0;
        p.Cin_total = g->O; p.Hin = Ho; p.Win = Wo; p.Kc = Mg; p.Cout_total = g->C; p.Ho = g->H; p.Wo = g->W; p.Mr = Cg;
        p.Sh = 1; p.Sw = 1; p.ph = (g->KH - 1) * g->dil_h - g->pad_h; p.pw = (g->KW - 1) * g->dil_w - g->pad_w;
        if (p.ph < 0 || p.pw < 0) return 0;
        xmode = MN_ACTQ_NONE;
    }
    if (p.Win % 4 || p.Wo % 4 || p.T > 64) return 0;
    // 256-pixel tiles made of whole output rows
    const int TP = 256;
    if (TP % p.Wo) return 0;
    const int rpt = TP / p.Wo;
    if (rpt >= p.Ho) {
        if (rpt % p.Ho) return 0;
        p.NI = rpt / p.Ho; p.TR = p.Ho; p.tpi = 1;
        p.num_ptiles = (p.N + p.NI - 1) / p.NI;
    } else {
        if (p.Ho % rpt) return 0;
        p.NI = 1; p.TR = rpt; p.tpi = p.Ho / rpt;
        p.num_ptiles = p.N * p.tpi;
    }
    p.PR = (p.TR - 1) * p.Sh + (p.KH - 1) * p.Dh + 1;
    p.PWp = (p.Wo - 1) * p.Sw + (p.KW - 1) * p.Dw + 1;
    p.npos = p.NI * p.PR * p.PWp;
    const int nterm = which == 1 ? 3 : 1;      // backward-data keeps the three terms of gy side by side
    p.CC = (p.Kc <= 16 || nterm == 3) ? 16 : 32;
    p.cc_shift = p.CC == 16 ? 4 : 5;
    p.nck = (p.Kc + p.CC - 1) / p.CC;
    p.Cgp = p.nck * p.CC;
    p.KS = (p.T * p.CC + 31) / 32;
    p.LDW = p.KS * 32 + 8;
    p.PSB = nterm * p.CC * 2 + 16;
    p.W4 = p.Win / 4;
    int NT = p.Mr > 32 ? 4 : (p.Mr > 16 ? 2 : 1);
    size_t lds;
    for (;;) {
        lds = (size_t)p.npos * p.PSB + (size_t)16 * NT * p.LDW * 2 + (size_t)2 * 16 * NT * 4 + (size_t)p.T * 4 + (size_t)p.nck * 8;
        lds = (lds + 15) / 16 * 16;
        if (lds <= KK_LDS_CAP) break;
        if (NT == 1) return 0;
        NT /= 2;
    }
    pl->NT = NT; pl->lds = lds; pl->xmode = xmode; pl->planes = nterm;
    p.num_mblk = (p.Mr + 16 * NT - 1) / (16 * NT);
    p.Mpad = p.num_mblk * 16 * NT;
    p.fd_wo = make_fastdiv(p.Wo); p.fd_tr = make_fastdiv(p.TR); p.fd_tpi = make_fastdiv(p.tpi); p.fd_pwp = make_fastdiv(p.PWp);
    p.fd_pr = make_fastdiv(p.PR); p.fd_w4 = make_fastdiv(p.W4); p.fd_o8 = make_fastdiv(p.CC / 8);
    const int64_t nb = (int64_t)p.num_ptiles * p.num_mblk * p.G;
    if (nb > 0x7fffffff) return 0;
    pl->grid = (int)nb;
    pl->off_codes = 0;
    const int64_t code_bytes = (int64_t)p.G * p.Mpad * p.T * p.Cgp * 2;
    pl->off_scale = (code_bytes + 255) / 256 * 256;
    const int64_t nscale = which == 0 ? (int64_t)p.G * p.Mpad : (int64_t)p.G * p.Cgp;
    pl->ws_bytes = pl->off_scale + nscale * 4;
    PackParams& k = pl->pk;
    k.G = g->groups; k.Mg = Mg; k.Cg = Cg; k.T = p.T; k.KW = g->KW; k.transpose = which == 1;
    k.Mpad = which == 0 ? p.Mpad : 0; k.Cgp = which == 0 ? p.Cgp : 0;
    k.Cpad = which == 1 ? p.Mpad : 0; k.Mgp = which == 1 ? p.Cgp : 0;
    pl->pack_grid = k.G * (which == 0 ? k.Mpad : k.Mgp);
    return 1;
}

template <int NT, int XMODE, int PL>
static void launch_kk1(const KkPlan& pl, hipStream_t s) {
    raise_lds_limit((const void*)k_kk<NT, XMODE, PL>, pl.lds);
    hipLaunchKernelGGL((k_kk<NT, XMODE, PL>), dim3(pl.grid), dim3(256), pl.lds, s, pl.p);
}
template <int NT>
static void launch_kk(const KkPlan& pl, hipStream_t s) {
    if (pl.planes == 3) launch_kk1<NT, MN_ACTQ_NONE, 3>(pl, s);
    else if (pl.xmode == MN_ACTQ_DOREFA) launch_kk1<NT, MN_ACTQ_DOREFA, 1>(pl, s);
    else if (pl.xmode == MN_ACTQ_IAO) launch_kk1<NT, MN_ACTQ_IAO, 1>(pl, s);
    else if (pl.xmode == MN_ACTQ_SIGN8) launch_kk1<NT, MN_ACTQ_SIGN8, 1>(pl, s);
    else launch_kk1<NT, MN_ACTQ_NONE, 1>(pl, s);
}
static int run_kk(const KkPlan& pl, hipStream_t s, const char* what) {
    mn_set_last_kernel("k_kk<%d, %d, %d>", pl.NT, pl.planes == 3 ? 0 : pl.xmode, pl.planes);
    mn_prof_begin(s);
    switch (pl.NT) {
        case 1: launch_kk<1>(pl, s); break;
        case 2: launch_kk<2>(pl, s); break;
        case 4: launch_kk<4>(pl, s); break;
        default: MN_FAIL(MN_EINVAL, "%s: bad NT", what);
    }
    mn_prof_end(s);
    MN_CHECK_LAUNCH(what);
    return MN_OK;
}

int kk_fwd(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, const float* x, const float* w, const float* bias, float* y,
           void* ws, int64_t ws_bytes, hipStream_t s) {
    KkPlan pl;
    if (aq && aq->mode == MN_ACTQ_CODE8) MN_FAIL(MN_ENOTSUP, "mn_conv2d_fwd: activation codes are read by mn_qconv_bnq_fwd_stash only");
    if (qd_iao_supported(g, aq, wq, 0) && ws_bytes >= qd_iao_ws_bytes(g, 0) && aligned16(x) && aligned16(y))          // dense IAO layers (the ResNets): qgemm_dense.hip
        return qd_iao_fwd(g, aq, wq, x, w, bias, y, ws, ws_bytes, s);
    if (!wq_codeable(wq) || !aq_codeable(aq, 0) || !plan_kk(g, 0, aq ? aq->mode : MN_ACTQ_NONE, &pl) || !aligned16(x) || !aligned16(y))
        MN_FAIL(MN_ENOTSUP, "mn_conv2d_fwd(qgemm): geometry / quantizer combination not covered");
    if (!ws || ws_bytes < pl.ws_bytes || !aligned16(ws)) MN_FAIL(MN_ENOSPC, "mn_conv2d_fwd(qgemm kxk): workspace too small (%lld < %lld)", (long long)ws_bytes, (long long)pl.ws_bytes);
    Pro pro;
    int rc = make_pro(aq, &pro, 0, "mn_conv2d_fwd(qgemm)");
    if (rc) return rc;
    fill_pack(pl.pk, wq, w, ws, pl.off_codes, pl.off_scale);
    qg_launch_pack(pl.pk, pl.pack_grid, s);
    KkParams& p = pl.p;
    p.in = x; p.out = y; p.out8 = nullptr; p.wc = pl.pk.codes; p.rowscale = pl.pk.scale_out; p.kscale = nullptr; p.bias = bias; p.aux = nullptr;
    p.pro = pro; p.ste = pro; p.epi = QG_EPI_SCALE_BIAS; p.ascale = pro.mode == MN_ACTQ_DOREFA ? pro.s : 1.f;
    return run_kk(pl, s, "mn_conv2d_fwd(qgemm kxk)");
}
int kk_h8_supported(const mn_conv_geom* g, const mn_wq* wq) {
    KkPlan pl;
    if (!g || !wq || wq->mode != MN_WQ_TERNARY || !plan_kk(g, 0, MN_ACTQ_SIGN8, &pl)) return 0;
    if (g->KH != 3 || g->KW != 3 || g->stride_h != 1 || g->stride_w != 1 || g->pad_h != 1 || g->pad_w != 1 || g->dil_h != 1 || g->dil_w != 1) return 0;
    return (g->C / g->groups) * 9 <= 254 && g->H >= 2 && g->W % 4 == 0;
}
int64_t kk_h8_ws_bytes(const mn_conv_geom* g) { KkPlan pl; return plan_kk(g, 0, MN_ACTQ_SIGN8, &pl) ? pl.ws_bytes : 0; }
int kk_h8_mpad(const mn_conv_geom* g) { KkPlan pl; return plan_kk(g, 0, MN_ACTQ_SIGN8, &pl) ? pl.pk.Mpad : 0; }
int kk_fwd_h8(const mn_conv_geom* g, const mn_wq* wq, const int8_t* x, const float* w, const float* nnzf, uint8_t* h, void* ws, int64_t ws_bytes,
              hipStream_t s, KkH8Info* info) {
    KkPlan pl;
    if (!kk_h8_supported(g, wq) || !plan_kk(g, 0, MN_ACTQ_SIGN8, &pl) || (((uintptr_t)x) & 3) || (((uintptr_t)h) & 3))
        MN_FAIL(MN_ENOTSUP, "mn_qconv_bnsign_fwd_stash(k x k): geometry / quantizer combination not covered");
    if (!ws || ws_bytes < pl.ws_bytes || !aligned16(ws)) MN_FAIL(MN_ENOSPC, "mn_qconv_bnsign_fwd_stash(k x k): workspace too small");
    mn_actq aq; aq.mode = MN_ACTQ_SIGN8; aq.bits = 8; aq.q_type = 0; aq.flags = 0; aq.qp = nullptr; aq.codes = nullptr; aq.stats = nullptr; aq.dx_add = nullptr; aq.ste_mask = nullptr;
    Pro pro;
    int rc = make_pro(&aq, &pro, 0, "mn_qconv_bnsign_fwd_stash(k x k)");
    if (rc) return rc;
    fill_pack(pl.pk, wq, w, ws, pl.off_codes, pl.off_scale);
    qg_launch_pack(pl.pk, pl.pack_grid, s);
    KkParams& p = pl.p;
    p.in = reinterpret_cast<const float*>(x); p.out = nullptr; p.out8 = h; p.wc = pl.pk.codes; p.rowscale = pl.pk.scale_out; p.kscale = nullptr;
    p.bias = nnzf; p.aux = nullptr; p.pro = pro; p.ste = pro; p.epi = QG_EPI_H8; p.ascale = 1.f;
    info->codes = pl.pk.codes; info->rowscale = pl.pk.scale_out; info->Mpad = pl.pk.Mpad; info->Kp = p.T * p.Cgp; info->K = p.T * p.Kc;
    { const double nx = (double)g->N * g->C * g->H * g->W, ny = (double)g->N * g->O * p.Ho * p.Wo; mn_prof_bytes(nx + ny); }
    return run_kk(pl, s, "mn_qconv_bnsign_fwd_stash(k x k)");
}
int kk_bwd_data(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, const float* gy, const float* w, const float* x, float* dx,
                void* ws, int64_t ws_bytes, hipStream_t s) {
    // dense layers without a clip-STE epilogue (activation codes: the STE lives in mn_qa_bwd_* / mn_qr_bwd_*): the staged-patch kernel of qgemm_dense.hip
    if ((!aq || aq->mode == MN_ACTQ_NONE || aq->mode == MN_ACTQ_CODE8 || aq->mode == MN_ACTQ_SIGN8) && qd_dgrad_native(g, wq) && ws_bytes >= qd_dgrad_ws_bytes(g) &&
        aligned16(gy) && aligned16(dx))
        return qd_bwd_data(g, wq, gy, w, dx, ws, ws_bytes, s);
    if (qd_iao_supported(g, aq, wq, 1) && ws_bytes >= qd_iao_ws_bytes(g, 1) && aligned16(gy) && aligned16(dx) && aligned16(x))
        return qd_iao_bwd_data(g, aq, wq, gy, w, x, dx, ws, ws_bytes, s);
    KkPlan pl;
    if (!wq_codeable(wq) || !plan_kk(g, 1, MN_ACTQ_NONE, &pl) || !aligned16(gy) || !aligned16(dx))
        MN_FAIL(MN_ENOTSUP, "mn_conv2d_bwd_data(qgemm): geometry / quantizer combination not covered");
    if (!ws || ws_bytes < pl.ws_bytes || !aligned16(ws)) MN_FAIL(MN_ENOSPC, "mn_conv2d_bwd_data(qgemm kxk): workspace too small");
    Pro ste;
    int rc = make_pro(aq, &ste, 1, "mn_conv2d_bwd_data(qgemm)");
    if (rc) return rc;
    if (ste.mode == MN_ACTQ_SIGN8 || ste.mode == MN_ACTQ_CODE8) ste.mode = MN_ACTQ_NONE;      // the clip-STE lives in mn_bnsign_bwd / mn_qa_bwd_*
    if (ste.mode != MN_ACTQ_NONE && (!x || !aligned16(x))) MN_FAIL(MN_EINVAL, "mn_conv2d_bwd_data(qgemm): x required (16 B aligned) for the clip-STE epilogue");
    if (ste.mode == MN_ACTQ_NONE && k3s_dgrad_supported(g, wq)) return k3s_bwd_data(g, wq, gy, w, dx, s);      // 3x3, ternary / binary weights: staged-image kernel
    fill_pack(pl.pk, wq, w, ws, pl.off_codes, pl.off_scale);
    qg_launch_pack(pl.pk, pl.pack_grid, s);
    Pro none; none.mode = MN_ACTQ_NONE; none.s = 1.f; none.qmin = none.qmax = 0.f; none.qp = nullptr;
    KkParams& p = pl.p;
    p.in = gy; p.out = dx; p.out8 = nullptr; p.wc = pl.pk.codes; p.rowscale = nullptr; p.kscale = pl.pk.scale_out; p.bias = nullptr; p.aux = x;
    p.pro = none; p.ste = ste; p.epi = ste.mode == MN_ACTQ_NONE ? QG_EPI_PLAIN : QG_EPI_STE; p.ascale = 1.f;
    return run_kk(pl, s, "mn_conv2d_bwd_data(qgemm kxk)");
}

// ------------------------------------------------------------------------------------------------
// k x k backward-weight:  dwq[g][m][c][tap] = sum over output pixels of gy[m][pixel] * code_x[c][pixel shifted by tap].
// K = output pixels.  Both operands are pixel-contiguous in NCHW, so fragments are 8 consecutive pixels of one row:
//   A: gy rows (three exact bf16 terms) staged as [term][m][tile pixel];
//   B: the activation codes staged as KW column-shifted copies [s][c][patch row][ocol] = code_x[c][row][ocol + s*Dw - pw],
//      so that the fragment of tap (r, s) starts on a 16-byte boundary for every s (a ds_read_b128 must be aligned);
// every tap is one 16-column N tile (16 input channels per block).  The four waves split the tile's 32-pixel K-steps; at
// the end of the block's tile range they sum their accumulators through LDS in wave order (deterministic) and write ONE
// partial tile; a second kernel reduces the Z partials in fp64.  dbias is accumulated from the staged gy values.
struct KwParams {
    const float* gy;
    const float* x;
    float* part;      // [Z][G][Mgw][Cgw*T]
    float* dbpart;    // [Z][G][Mgw]
    Pro pro;
    int N, C, H, W, O, Ho, Wo, Cg, Mg, G, KH, KW, T, Sh, Dh, Dw, ph, pw;
    int NI, TR, tpi, PR, num_ptiles, GS, XCS, EQ, PWL, nmb, ncb, Z, Mgw, Cgw, want_db;
    FastDiv fd_wo, fd_tr, fd_tpi, fd_pr, fd_eq, fd_ppi;
    ChanMap in_map;
};
#define KW_TP 128     // output pixels per tile (4 K-steps of 32 pixels: one per wave)
#define KW_CC 16      // input channels per block (one c-tile)
#define KW_XPF 5      // x items (float4) a thread prefetches into registers per tile (nin_gc 3x3 layers: 4-5 per thread)

template <int MT, int NTL, int XMODE>
__global__ __launch_bounds__(256, 2) void k_kk_wgrad(const KwParams p) {
    HIP_DYNAMIC_SHARED(float, smem)
    constexpr int MTt = 16 * MT, NDB = MTt / 8, TPQ = KW_TP / 4;
    uint16_t* gt = reinterpret_cast<uint16_t*>(smem);                 // [3][MTt][GS]; after the last tile: float red[MTt][16*T]
    const size_t gt_bytes = (size_t)3 * MTt * p.GS * 2, red_bytes = (size_t)MTt * KW_CC * p.T * 4;
    uint16_t* xc = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(smem) + (gt_bytes > red_bytes ? gt_bytes : red_bytes));   // [KW][16][XCS]
    int* ntoff = reinterpret_cast<int*>(xc + (size_t)p.KW * KW_CC * p.XCS);      // [T] element offset of a tap inside xc
    int* xflag = ntoff + p.T;                                        // [2], alternating per tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kg = lane >> 4;
    uint32_t b = blockIdx.x;
    const int z = b % p.Z; b /= p.Z;
    const int cb = b % p.ncb; b /= p.ncb;
    const int mb = b % p.nmb;
    const int g = b / p.nmb;
    float sc = 1.f, zp = 0.f;
    if (XMODE == MN_ACTQ_IAO) { sc = p.pro.qp[0]; zp = p.pro.qp[1]; }

    for (int tap = tid; tap < p.T; tap += 256) {
        const int r = tap / p.KW, s_ = tap - r * p.KW;
        ntoff[tap] = s_ * KW_CC * p.XCS + r * p.Dh * p.Wo;
    }
    if (tid < 2) xflag[tid] = 0;

    f32x4 acc[MT][NTL];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NTL; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbacc[NDB];          // rows (tid >> 5) + 8i of the block's gy slab: the staging always gives a thread the same rows
#pragma unroll
    for (int i = 0; i < NDB; ++i) dbacc[i] = 0.f;

    const int HoWo = p.Ho * p.Wo, px_per_img = p.TR * p.Wo;
    const int64_t xplane = (int64_t)p.H * p.W;
    // channels beyond Cg are skipped altogether: they only feed dw columns that are never read
    const int ccv = (p.Cg - cb * KW_CC) < KW_CC ? (p.Cg - cb * KW_CC) : KW_CC;
    const int nitem = ccv * p.NI * p.PR * p.EQ;

    struct XItem { int dst, ic0; int64_t src; };      // dst < 0: no item; src < 0: zero padding; else element offset into x
    auto x_item = [&](int it, int n0, int row0) {
        XItem r; r.dst = -1; r.ic0 = 0; r.src = -1;
        if (it < nitem) {
            const uint32_t t1 = fd_div(it, p.fd_eq);
            const int eq = it - t1 * p.EQ;
            const uint32_t t2 = fd_div(t1, p.fd_pr);
            const int prow = t1 - t2 * p.PR;
            const int cl = (int)t2 / p.NI, slot = (int)t2 - cl * p.NI;
            const int c = cb * KW_CC + cl, n = n0 + slot, ir = row0 + prow, ic0 = eq * 4 - p.PWL;
            r.dst = cl * p.XCS + (slot * p.PR + prow) * p.Wo;
            r.ic0 = ic0;
            if (n < p.N && ir >= 0 && ir < p.H && ic0 >= 0 && ic0 + 3 < p.W)
                r.src = ((int64_t)n * p.C + chan_phys(p.in_map, g * p.Cg + c)) * xplane + (int64_t)ir * p.W + ic0;
        }
        return r;
    };
    unsigned inx = 0u;
    auto x_scatter = [&](const XItem& xi, float4 f, int term) {
        if (xi.dst < 0) return;
        float v[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (XMODE == MN_ACTQ_NONE) {
                float r = v[e] - mn_bf16_head(v[e]);
                if (term == 0) { inx |= mn_f2u(r) << 1; }
                else { if (term == 2) r = r - mn_bf16_head(r); v[e] = r; }
            } else {
                v[e] = act_code<XMODE>(v[e], p.pro, sc, zp);    // a zero-padded element quantises to code 0 in both schemes
            }
        }
        uint16_t* rowp = xc + xi.dst;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint16_t hb = (uint16_t)(mn_f2u(v[e]) >> 16);
            for (int s_ = 0; s_ < p.KW; ++s_) {
                const int ocol = xi.ic0 + e - s_ * p.Dw + p.pw;
                if (ocol >= 0 && ocol < p.Wo) rowp[(size_t)s_ * KW_CC * p.XCS + ocol] = hb;
            }
        }
    };
    auto tile_origin = [&](int pt, int& n0, int& oh0, int& row0) {
        const uint32_t timg = fd_div(pt, p.fd_tpi);
        n0 = (int)timg * p.NI;
        oh0 = (pt - (int)timg * p.tpi) * p.TR;
        row0 = oh0 * p.Sh - p.ph;
    };

    float4 rg[NDB], rx[KW_XPF];
    auto fetch = [&](int pt) {            // global loads of one tile, left in flight
        int n0, oh0, row0;
        tile_origin(pt, n0, oh0, row0);
        const int pix = (tid & (TPQ - 1)) * 4;
        const uint32_t img = fd_div(pix, p.fd_ppi);
        const int rem = pix - img * px_per_img;
        const int n = n0 + (int)img;
#pragma unroll
        for (int i = 0; i < NDB; ++i) {
            const int m = (tid >> 5) + 8 * i, mo = mb * MTt + m;
            rg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n < p.N && mo < p.Mg) rg[i] = *reinterpret_cast<const float4*>(p.gy + ((int64_t)n * p.O + (int64_t)g * p.Mg + mo) * HoWo + oh0 * p.Wo + rem);
        }
#pragma unroll
        for (int u = 0; u < KW_XPF; ++u) {
            const XItem xi = x_item(tid + u * 256, n0, row0);
            rx[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (xi.src >= 0) rx[u] = mn_ld_x4<XMODE>(p.x, xi.src);
        }
    };
    auto commit = [&](int pt, int par) {   // registers -> LDS
        int n0, oh0, row0;
        tile_origin(pt, n0, oh0, row0);
        const int pix = (tid & (TPQ - 1)) * 4;
#pragma unroll
        for (int i = 0; i < NDB; ++i) {
            const int m = (tid >> 5) + 8 * i;
            const float v[4] = {rg[i].x, rg[i].y, rg[i].z, rg[i].w};
            float t0[4], t1[4], t2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                t0[e] = mn_bf16_head(v[e]);
                const float r1 = v[e] - t0[e];
                t1[e] = mn_bf16_head(r1);
                t2[e] = r1 - t1[e];
            }
            dbacc[i] += (v[0] + v[1]) + (v[2] + v[3]);
            uint16_t* d = gt + m * p.GS + pix;
            *reinterpret_cast<u32x2*>(d) = u32x2{mn_pack_bf16x2(t0[0], t0[1]), mn_pack_bf16x2(t0[2], t0[3])};
            *reinterpret_cast<u32x2*>(d + MTt * p.GS) = u32x2{mn_pack_bf16x2(t1[0], t1[1]), mn_pack_bf16x2(t1[2], t1[3])};
            *reinterpret_cast<u32x2*>(d + 2 * MTt * p.GS) = u32x2{mn_pack_bf16x2(t2[0], t2[1]), mn_pack_bf16x2(t2[2], t2[3])};
        }
        inx = 0u;
#pragma unroll
        for (int u = 0; u < KW_XPF; ++u) x_scatter(x_item(tid + u * 256, n0, row0), rx[u], 0);
        for (int it = tid + KW_XPF * 256; it < nitem; it += 256) {      // patches larger than the prefetch window
            const XItem xi = x_item(it, n0, row0);
            x_scatter(xi, xi.src >= 0 ? mn_ld_x4<XMODE>(p.x, xi.src) : make_float4(0.f, 0.f, 0.f, 0.f), 0);
        }
        if (XMODE == MN_ACTQ_NONE && inx) xflag[par] = 1;
    };
    auto restage_x_term = [&](int pt, int term) {     // slow path of a real-valued x
        int n0, oh0, row0;
        tile_origin(pt, n0, oh0, row0);
        for (int it = tid; it < nitem; it += 256) {
            const XItem xi = x_item(it, n0, row0);
            x_scatter(xi, xi.src >= 0 ? mn_ld_x4<XMODE>(p.x, xi.src) : make_float4(0.f, 0.f, 0.f, 0.f), term);
        }
    };
    auto contract = [&]() {
        const int pi = wave * 32 + kg * 8;      // this wave's K-step of the tile
        const uint32_t fr = fd_div(pi, p.fd_wo);
        const int ocol0 = pi - fr * p.Wo;
        const uint32_t slot = fd_div(fr, p.fd_tr);
        const int orow = fr - slot * p.TR;
        const uint16_t* xb = xc + (size_t)j * p.XCS + ((int)slot * p.PR + orow * p.Sh) * p.Wo + ocol0;
        u32x4 a[MT][3];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int term = 0; term < 3; ++term)
                a[mi][term] = *reinterpret_cast<const u32x4*>(gt + (term * MTt + mi * 16 + j) * p.GS + pi);
        // three taps at a time, term-outer: 3*MT independent accumulators between two MFMAs on the same one
#pragma unroll
        for (int n0 = 0; n0 < NTL; n0 += 3) {
            u32x4 bf[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) bf[u] = (n0 + u < NTL && n0 + u < p.T) ? *reinterpret_cast<const u32x4*>(xb + ntoff[n0 + u]) : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    if (n0 + u < NTL && n0 + u < p.T) {
#pragma unroll
                        for (int mi = 0; mi < MT; ++mi) acc[mi][n0 + u] = mn_mfma_bf16(a[mi][term], bf[u], acc[mi][n0 + u]);
                    }
                }
        }
    };

    int par = 0;
    if (z < p.num_ptiles) fetch(z);
    for (int pt = z; pt < p.num_ptiles; pt += p.Z, par ^= 1) {
        __syncthreads();                       // previous tile consumed; tables / flags visible
        commit(pt, par);
        __syncthreads();
        const int inexact = (XMODE == MN_ACTQ_NONE) ? xflag[par] : 0;
        if (XMODE == MN_ACTQ_NONE && tid == 0) xflag[par ^ 1] = 0;
        if (pt + p.Z < p.num_ptiles) fetch(pt + p.Z);       // in flight during the MFMA phase
        contract();
        if (inexact) {                         // block-uniform: a real-valued x, contract its second and third term as well
            for (int term = 1; term <= 2; ++term) {
                __syncthreads();
                restage_x_term(pt, term);
                __syncthreads();
                contract();
            }
        }
    }
    // ---- sum the four waves' accumulators in wave order through LDS, then one coalesced partial-tile write
    float* red = reinterpret_cast<float*>(smem);
    const int RW = KW_CC * p.T;                // columns of the block's dw tile: (c_local, tap)
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int ni = 0; ni < NTL; ++ni) {
                    if (ni < p.T) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int idx = (mi * 16 + kg * 4 + r) * RW + j * p.T + ni;
                            red[idx] = (w == 0) ? acc[mi][ni][r] : red[idx] + acc[mi][ni][r];
                        }
                    }
                }
        }
    }
    __syncthreads();
    for (int i = tid; i < MTt * RW; i += 256) {
        const int m = i / RW, col = i - m * RW;
        p.part[(((int64_t)z * p.G + g) * p.Mgw + mb * MTt + m) * ((int64_t)p.Cgw * p.T) + (int64_t)cb * RW + col] = red[i];
    }
    if (p.want_db && cb == 0) {
#pragma unroll
        for (int i = 0; i < NDB; ++i) {
            float v = dbacc[i];
            v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
            if ((tid & 31) == 0) p.dbpart[((int64_t)z * p.G + g) * p.Mgw + mb * MTt + (tid >> 5) + 8 * i] = v;
        }
    }
}

struct KwPlan {
    KwParams p;
    int MT, NTL;
    size_t lds;
    int grid;
    int64_t off_db, ws_bytes;
};
static int plan_kk_wgrad(const mn_conv_geom* g, KwPlan* pl) {
    KwParams& p = pl->p;
    p.N = g->N; p.C = g->C; p.H = g->H; p.W = g->W; p.O = g->O; p.G = g->groups; p.KH = g->KH; p.KW = g->KW; p.T = g->KH * g->KW;
    p.Ho = kk_out_dim(g->H, g->KH, g->stride_h, g->pad_h, g->dil_h); p.Wo = kk_out_dim(g->W, g->KW, g->stride_w, g->pad_w, g->dil_w);
    p.Cg = g->C / g->groups; p.Mg = g->O / g->groups; p.Sh = g->stride_h; p.Dh = g->dil_h; p.Dw = g->dil_w; p.ph = g->pad_h; p.pw = g->pad_w;
    if (g->stride_w != 1 || p.W % 4 || p.Wo % 8 || KW_TP % p.Wo) return 0;
    if (g->in_shuffle > 1 && g->C % g->in_shuffle) return 0;
    p.in_map = make_chanmap(g->in_shuffle, g->C);
    if (p.T <= 9) pl->NTL = 9; else if (p.T <= 25) pl->NTL = 25; else return 0;
    pl->MT = (pl->NTL == 9 && p.Mg > 16) ? 2 : 1;
    const int MTt = 16 * pl->MT;
    const int rpt = KW_TP / p.Wo;
    if (rpt >= p.Ho) {
        if (rpt % p.Ho) return 0;
        p.NI = rpt / p.Ho; p.TR = p.Ho; p.tpi = 1; p.num_ptiles = (p.N + p.NI - 1) / p.NI;
    } else {
        if (p.Ho % rpt) return 0;
        p.NI = 1; p.TR = rpt; p.tpi = p.Ho / rpt; p.num_ptiles = p.N * p.tpi;
    }
    p.PR = (p.TR - 1) * p.Sh + (p.KH - 1) * p.Dh + 1;
    p.GS = KW_TP + 8;
    p.XCS = p.NI * p.PR * p.Wo + 8;
    p.PWL = qg_roundup(p.pw, 4);
    const int max_ic = p.Wo - 1 + (p.KW - 1) * p.Dw - p.pw;          // last input column any tap touches
    p.EQ = (max_ic + p.PWL) / 4 + 1;
    const size_t gt_bytes = (size_t)3 * MTt * p.GS * 2, red_bytes = (size_t)MTt * KW_CC * p.T * 4;
    size_t lds = (gt_bytes > red_bytes ? gt_bytes : red_bytes) + (size_t)p.KW * KW_CC * p.XCS * 2 + (size_t)p.T * 4 + 16;
    lds = (lds + 15) / 16 * 16;
    if (lds > KK_LDS_CAP) return 0;
    pl->lds = lds;
    p.nmb = (p.Mg + MTt - 1) / MTt; p.ncb = (p.Cg + KW_CC - 1) / KW_CC;
    p.Mgw = p.nmb * MTt; p.Cgw = p.ncb * KW_CC;
    const int base = p.G * p.nmb * p.ncb;
    int Z = 512 / base;
    if (Z > p.num_ptiles / 2) Z = p.num_ptiles / 2;
    if (Z < 1) Z = 1;
    p.Z = Z;
    p.fd_wo = make_fastdiv(p.Wo); p.fd_tr = make_fastdiv(p.TR); p.fd_tpi = make_fastdiv(p.tpi); p.fd_pr = make_fastdiv(p.PR);
    p.fd_eq = make_fastdiv(p.EQ); p.fd_ppi = make_fastdiv(p.TR * p.Wo);
    const int64_t nb = (int64_t)base * Z;
    if (nb > 0x7fffffff) return 0;
    pl->grid = (int)nb;
    const int64_t part_bytes = (int64_t)Z * p.G * p.Mgw * p.Cgw * p.T * 4;
    pl->off_db = (part_bytes + 255) / 256 * 256;
    pl->ws_bytes = pl->off_db + (int64_t)Z * p.G * p.Mgw * 4;
    return 1;
}
template <int MT, int NTL>
static void launch_kw(const KwPlan& pl, int xmode, hipStream_t s) {
    if (xmode == MN_ACTQ_DOREFA) {
        raise_lds_limit((const void*)k_kk_wgrad<MT, NTL, MN_ACTQ_DOREFA>, pl.lds);
        hipLaunchKernelGGL((k_kk_wgrad<MT, NTL, MN_ACTQ_DOREFA>), dim3(pl.grid), dim3(256), pl.lds, s, pl.p);
    } else if (xmode == MN_ACTQ_IAO) {
        raise_lds_limit((const void*)k_kk_wgrad<MT, NTL, MN_ACTQ_IAO>, pl.lds);
        hipLaunchKernelGGL((k_kk_wgrad<MT, NTL, MN_ACTQ_IAO>), dim3(pl.grid), dim3(256), pl.lds, s, pl.p);
    } else if (xmode == MN_ACTQ_SIGN8) {
        raise_lds_limit((const void*)k_kk_wgrad<MT, NTL, MN_ACTQ_SIGN8>, pl.lds);
        hipLaunchKernelGGL((k_kk_wgrad<MT, NTL, MN_ACTQ_SIGN8>), dim3(pl.grid), dim3(256), pl.lds, s, pl.p);
    } else {
        raise_lds_limit((const void*)k_kk_wgrad<MT, NTL, MN_ACTQ_NONE>, pl.lds);
        hipLaunchKernelGGL((k_kk_wgrad<MT, NTL, MN_ACTQ_NONE>), dim3(pl.grid), dim3(256), pl.lds, s, pl.p);
    }
}

int kk_supported(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, int which) {
    KkPlan pl;
    if (which >= 0 && which <= 2 && qd_iao_supported(g, aq, wq, which)) return 1;
    if (which == 0) return wq_codeable(wq) && aq_codeable(aq, 0) && plan_kk(g, 0, aq ? aq->mode : MN_ACTQ_NONE, &pl);
    if (which == 1) return wq_codeable(wq) && (plan_kk(g, 1, MN_ACTQ_NONE, &pl) ||
                                               ((!aq || aq->mode == MN_ACTQ_NONE || aq->mode == MN_ACTQ_CODE8 || aq->mode == MN_ACTQ_SIGN8) && qd_dgrad_native(g, wq)));
    if (which == 2 && aq && aq->mode == MN_ACTQ_CODE8) return k3s_wgrad_code8_supported(g, aq->bits) || qd_wgrad_supported(g, aq->bits);
    if (which == 2) { KwPlan kw; return aq_codeable(aq, 1) && (plan_kk_wgrad(g, &kw) || (aq && aq->mode == MN_ACTQ_SIGN8 && k3s_wgrad_supported(g))); }
    return 0;
}
int64_t kk_ws_bytes(const mn_conv_geom* g, int which) {
    KkPlan pl;
    if (which == 0) {   // the plan (hence Mpad / Cgp) depends on the activation mode: take the larger
        int64_t a = plan_kk(g, 0, MN_ACTQ_NONE, &pl) ? pl.ws_bytes : 0;
        int64_t b = plan_kk(g, 0, MN_ACTQ_DOREFA, &pl) ? pl.ws_bytes : 0;
        const int64_t c = qd_iao_ws_bytes(g, 0);
        if (c > a) a = c;
        return a > b ? a : b;
    }
    if (which == 1) { const int64_t a = plan_kk(g, 1, MN_ACTQ_NONE, &pl) ? pl.ws_bytes : 0, b = qd_dgrad_ws_bytes(g); return a > b ? a : b; }
    if (which == 2) {
        KwPlan kw;
        const int64_t a = plan_kk_wgrad(g, &kw) ? kw.ws_bytes : 0, b = k3s_wgrad_ws_bytes(g), c = qd_wgrad_ws_bytes(g), d = qd_iao_ws_bytes(g, 2);
        int64_t m = a > b ? a : b;
        if (c > m) m = c;
        return m > d ? m : d;
    }
    return 0;
}
int kk_bwd_weight(const mn_conv_geom* g, const mn_actq* aq, const float* gy, const float* x, float* dw, float* dbias, void* ws,
                  int64_t ws_bytes, hipStream_t s) {
    if (aq && aq->mode == MN_ACTQ_SIGN8 && k3s_wgrad_supported(g) && ws_bytes >= k3s_wgrad_ws_bytes(g))
        return k3s_bwd_weight(g, gy, (const int8_t*)x, dw, dbias, ws, ws_bytes, s);      // 3x3 on sign codes: wave-private streaming kernel
    if (aq && aq->mode == MN_ACTQ_CODE8 && !dbias && qd_wgrad_supported(g, aq->bits) && ws_bytes >= qd_wgrad_ws_bytes(g))      // dense layers: qgemm_dense.hip
        return qd_bwd_weight(g, gy, (const uint8_t*)x, dorefa_scale(aq->bits), dw, ws, ws_bytes, s);
    if (qd_iao_supported(g, aq, nullptr, 2) && ws_bytes >= qd_iao_ws_bytes(g, 2) && aligned16(gy) && aligned16(x))          // dense IAO layers
        return qd_iao_bwd_weight(g, aq, gy, x, dw, dbias, ws, ws_bytes, s);
    if (aq && aq->mode == MN_ACTQ_CODE8) {        // k-bit activation codes: only the wave-private 3x3 kernel reads them
        if (!k3s_wgrad_code8_supported(g, aq->bits) || ws_bytes < k3s_wgrad_ws_bytes(g)) MN_FAIL(MN_ENOTSUP, "mn_conv2d_bwd_weight(code8): geometry / bits not covered");
        return k3s_bwd_weight_code8(g, gy, (const uint8_t*)x, aq->bits, dorefa_scale(aq->bits), dw, dbias, ws, ws_bytes, s);
    }
    KwPlan pl;
    if (!aq_codeable(aq, 1) || !plan_kk_wgrad(g, &pl) || !aligned16(gy) || !aligned16(x))
        MN_FAIL(MN_ENOTSUP, "mn_conv2d_bwd_weight(qgemm): geometry / quantizer combination not covered");
    if (!ws || ws_bytes < pl.ws_bytes || !aligned16(ws)) MN_FAIL(MN_ENOSPC, "mn_conv2d_bwd_weight(qgemm kxk): workspace too small");
    Pro pro;
    int rc = make_pro(aq, &pro, 0, "mn_conv2d_bwd_weight(qgemm)");
    if (rc) return rc;
    KwParams& p = pl.p;
    p.gy = gy; p.x = x; p.part = (float*)ws; p.dbpart = (float*)((char*)ws + pl.off_db); p.pro = pro; p.want_db = dbias != nullptr;
    mn_set_last_kernel("k_kk_wgrad<%d, %d, %d>", pl.MT, pl.NTL, pro.mode);
    mn_prof_begin(s);
    if (pl.NTL == 9 && pl.MT == 2) launch_kw<2, 9>(pl, pro.mode, s);
    else if (pl.NTL == 9) launch_kw<1, 9>(pl, pro.mode, s);
    else launch_kw<1, 25>(pl, pro.mode, s);
    mn_prof_end(s);
    qg_launch_wgrad_reduce(p.part, p.dbpart, dw, dbias, p.Z, p.G, p.Mg, p.Cg * p.T, p.Mgw, p.Cgw * p.T,
                           pro.mode == MN_ACTQ_DOREFA ? pro.s : 1.f, pro.mode == MN_ACTQ_IAO ? pro.qp : (const float*)nullptr, s);
    MN_CHECK_LAUNCH("mn_conv2d_bwd_weight(qgemm kxk)");
    return MN_OK;
}
