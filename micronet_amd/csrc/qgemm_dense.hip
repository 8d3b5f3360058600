// Dense (groups = 1) code-domain convolutions for gfx950: the 3 x 3 / padding 1 (stride 1 and 2) and 1 x 1 / stride 2 layers of the
// reference's CIFAR ResNets (models/resnet.py:7-65, 122-182) under the k-bit DoReFa scheme (wqaq/dorefa/quantize.py:107-122):
//
//   forward          acc[n][o][oh][ow] = sum over (c, r, s) of wcode[o][c][r][s] * j[n][c][oh*S + r - P][ow*S + s - P]      (exact integers)
//   backward-data    dq[n][c][ih][iw]  = sum over (o, r, s) of wcode[o][c][r][s] * (gy[n][o][oh][ow] / n_w),  ih = oh*S + r - P ...
//   backward-weight  dw[o][c][r][s]    = s_a * sum over (n, oh, ow) of gy[n][o][oh][ow] * j[n][c][oh*S + r - P][ow*S + s - P]
//
// with Cin, Cout multiples of 64 (64 ... 512) and small images (W = 4 ... 32): K = 9 Cin = 576 ... 4608 -- unlike the grouped layers of nin_gc
// (qgemm_k3s.hip, K = 144) these are bound by the matrix cores, not by HBM.  All three are implicit GEMMs on v_mfma_f32_16x16x32_bf16 with
// everything global in the reference's NCHW layout (so the streaming kernels of qact_kernels.hip consume / produce the same tensors):
//
//   * k_qd_fwd: a block (4 waves) owns 64 MF output pixels x 64 output channels and walks the input channels in chunks of 64.  The chunk's
//     input patch ((TH - 1) S + 3 rows x (Wo - 1) S + 3 columns per image, one-pixel zero frame) is staged ONCE per chunk as bf16 codes,
//     transposed to [pixel slot][64 c] (144-byte slots: conflict-free b128 reads): the nine taps are nine constant slot offsets, so one
//     staged element feeds 9 x 64 MACs.  The weights arrive pre-packed in fragment order (k_qd_pack), 8 KB per (tap, chunk) step, through a
//     two-slot LDS ring filled one step ahead; the next chunk's patch is fetched into registers while the current one is contracted.  A wave
//     computes 16 MF pixels x 64 channels: per K-step of 32, MF A + 4 B fragment reads feed 4 MF MFMAs.  M = pixels: a lane ends with 4
//     consecutive pixels of one channel = one 8-byte (int16) or 16-byte (int32) store of the NCHW stash.
//   * the statistics of the stash (k_qd_stats) are exact integer sums per channel, in the partial layout k_qa_stats_prep reads.
#include "qgemm_dev.h"

#include <stdlib.h>

#define QD_RS 144              // forward patch: LDS bytes per pixel slot (64 bf16 + pad)
#define QD_WSTEP 8192          // bytes of packed weights per forward step (one tap x 64 input channels x 64 output channels)
#define QDF_UPT 6              // staging units (4 channels x 4 pixels) per thread and chunk

// Order of the staging units (channel quad q, image, patch row pr, row dword d) over the threads.  A unit writes 8 bytes (4 channels of one pixel) per pixel into
// slot-major LDS images whose slot stride is 144 or 80 bytes: the bank pair of a write is (4 or 20) * slot + 2 q mod 32 = 16 (d & 1) + 8 (pr & 1) + 2 q + const.
// The low four bits of the unit index are therefore (q & 3, d & 1, pr & 1) -- rows of one dword, W = 4: (q & 3, pr & 3) -- so that the 16 lanes of an LDS write
// group fall on 16 distinct bank pairs (the plain row-major order put all of them on two: 8-way conflicts, 2-3 k cycles per chunk); the global loads still read
// 64 contiguous bytes per channel plane and instruction.
struct QdUnits { int mode, A, B, NI, PH, NQ, nunits; FastDiv fd_a, fd_b, fd_ni; };
static QdUnits qd_make_units(int W4, int PH, int NI, int NQ) {
    QdUnits u;
    u.mode = W4 >= 2 ? 0 : 1; u.A = W4 >= 2 ? W4 / 2 : 1; u.B = W4 >= 2 ? (PH + 1) / 2 : (PH + 3) / 4; u.NI = NI; u.PH = PH; u.NQ = NQ;
    u.nunits = 16 * u.A * u.B * NI * (NQ / 4);
    u.fd_a = make_fastdiv((uint32_t)u.A); u.fd_b = make_fastdiv((uint32_t)u.B); u.fd_ni = make_fastdiv((uint32_t)NI);
    return u;
}
__device__ __forceinline__ bool qd_unit(const QdUnits& m, int u, int& q, int& img, int& pr, int& d) {
    const int ql = u & 3;
    uint32_t r = (uint32_t)u >> 4;
    int dl = 0, prl;
    if (m.mode == 0) { dl = (u >> 2) & 1; prl = (u >> 3) & 1; } else prl = (u >> 2) & 3;
    const uint32_t r1 = fd_div(r, m.fd_a);
    const int dh = (int)r - (int)r1 * m.A;
    const uint32_t r2 = fd_div(r1, m.fd_b);
    const int prh = (int)r1 - (int)r2 * m.B;
    const uint32_t qh = fd_div(r2, m.fd_ni);
    img = (int)r2 - (int)qh * m.NI;
    q = 4 * (int)qh + ql;
    d = m.mode == 0 ? 2 * dh + dl : 0;
    pr = m.mode == 0 ? 2 * prh + prl : 4 * prh + prl;
    return u < m.nunits && pr < m.PH && q < m.NQ;
}

// ------------------------------------------------------------------------------------------------ weight codes in fragment order
// orient 0 (forward):        [cot][chunk = c / 64][tap][ks = 0, 1][nf = 0..3][lane][e]: o = 64 cot + 16 nf + (lane & 15), c = 64 chunk + 32 ks + 8 (lane >> 4) + e
// orient 1 (backward-data):  [cit][chunk = o / 32][tap][nf][lane][e]:                   c = 64 cit + 16 nf + (lane & 15), o = 32 chunk + 8 (lane >> 4) + e
// code = rint(w * (2^bits - 1)): the integer 2k - n of a DoReFa weight (2k - n) / n (wqaq/dorefa/quantize.py:68-72); exact in bf16
// IAO weights (wqaq/iao/quantize.py:227-239, symmetric): w = code * scale[o] -> code = rint(w / scale[o]) (wsc != nullptr; stride: floats between channels, 0 per layer)
struct QdPackParams { const float* w; uint16_t* out; int O, C, T; float wn; int orient; int64_t ngroups; const float* wsc; int wsc_stride; };
__device__ __forceinline__ void qd_pack_group(const QdPackParams& p, int64_t gi) {
    const int lane = (int)(gi & 63);
    int64_t t = gi >> 6;
    const int nf = (int)(t & 3); t >>= 2;
    int o0, c0, tap, ostep, cstep;
    if (p.orient == 0) {
        const int ks = (int)(t & 1); t >>= 1;
        tap = (int)(t % p.T); t /= p.T;
        const int nch = p.C / 64;
        const int chunk = (int)(t % nch), cot = (int)(t / nch);
        o0 = cot * 64 + nf * 16 + (lane & 15); c0 = chunk * 64 + ks * 32 + (lane >> 4) * 8; ostep = 0; cstep = 1;
    } else {
        tap = (int)(t % p.T); t /= p.T;
        const int nch = p.O / 32;
        const int chunk = (int)(t % nch), cit = (int)(t / nch);
        c0 = cit * 64 + nf * 16 + (lane & 15); o0 = chunk * 32 + (lane >> 4) * 8; ostep = 1; cstep = 0;
    }
    uint32_t h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int o = o0 + e * ostep, c = c0 + e * cstep;
        const float v = p.w[((int64_t)o * p.C + c) * p.T + tap];
        h[e] = mn_f2u(p.wsc ? rintf(v / p.wsc[(int64_t)o * p.wsc_stride]) : rintf(v * p.wn)) >> 16;
    }
    *reinterpret_cast<u32x4*>(p.out + gi * 8) = u32x4{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
}
// orient 2 (forward on v_mfma_i32_16x16x64_i8): [cot][chunk = c / 64][tap][nf = 0..3][lane][e = 0..15] signed bytes: o = 64 cot + 16 nf + (lane & 15),
// c = 64 chunk + 16 (lane >> 4) + e; a group = 16 codes = 16 bytes
__device__ __forceinline__ void qd_pack_group8(const QdPackParams& p, int64_t gi) {
    const int lane = (int)(gi & 63);
    int64_t t = gi >> 6;
    const int nf = (int)(t & 3); t >>= 2;
    const int tap = (int)(t % p.T); t /= p.T;
    const int nch = p.C / 64;
    const int chunk = (int)(t % nch), cot = (int)(t / nch);
    const int o = cot * 64 + nf * 16 + (lane & 15), c0 = chunk * 64 + (lane >> 4) * 16;
    const float isc = p.wsc ? p.wsc[(int64_t)o * p.wsc_stride] : 1.f;
    uint32_t d[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float v = p.w[((int64_t)o * p.C + c0 + e) * p.T + tap];
        const int code = (int)(p.wsc ? rintf(v / isc) : rintf(v * p.wn));
        d[e >> 2] |= ((uint32_t)code & 0xffu) << (8 * (e & 3));
    }
    *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned char*>(p.out) + gi * 16) = u32x4{d[0], d[1], d[2], d[3]};
}
__global__ __launch_bounds__(256) void k_qd_pack(const QdPackParams p) {
    const int64_t gi = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gi >= p.ngroups) return;
    if (p.orient == 2) qd_pack_group8(p, gi); else qd_pack_group(p, gi);
}
// both orientations of up to QD_PACK_MAX weight tensors in ONE launch (mn_qd_pack_multi: once per training step, right after the weight quantizer)
#define QD_PACK_MAX 32
struct QdPackTable {
    const float* w[QD_PACK_MAX]; uint16_t* outf[QD_PACK_MAX]; uint16_t* outd[QD_PACK_MAX]; const float* wsc[QD_PACK_MAX];
    int O[QD_PACK_MAX], C[QD_PACK_MAX], T[QD_PACK_MAX], wsc_stride[QD_PACK_MAX], blk0[QD_PACK_MAX + 1];
    int n; float wn; int fwd8;          // fwd8: the forward image in the int8 order (orient 2)
};
// A block owns a 16-row slice of one (64 o, 64 c) tile of one tensor: the slice's 16 rows of 64 T contiguous floats are read coalesced (all of a thread's 16-byte
// loads in flight at once), turned into codes ONCE (int16 in LDS, [o][c][tap]) and written out in every fragment order wanted as whole 16-byte lanes -- one nf of the
// forward image ([cot][chunk][tap][nf] blocks of 1 KB int8 / 2 x 1 KB bf16) and two of the four lane groups of one 32-row chunk of the backward-data image.
// (History: the first version gathered 4-byte elements at strides of T and C T floats per lane: 139 us for resnet18's 11 M weights; whole tiles by 256 and then 1024
// threads: 77-105 us -- ~300 blocks, one per CU, each serialising 147 KB of loads behind one another; slices give 1200 blocks of 256 threads and ~5 blocks per CU.)
#define QD_PACK_THREADS 256
#define QD_PACK_ROWS 16
#define QD_PACK_MAXQ 9          // float4 loads per thread: 16 rows x 64 c x 9 taps / 4 / 256
__global__ __launch_bounds__(QD_PACK_THREADS) void k_qd_pack_multi(const QdPackTable t) {
    HIP_DYNAMIC_SHARED(float, smem)
    int16_t* codes = reinterpret_cast<int16_t*>(smem);          // [16 o][64 c][T]
    int e = 0;
    while (e + 1 < t.n && (int)blockIdx.x >= t.blk0[e + 1]) ++e;
    const int O = t.O[e], Cn = t.C[e], T = t.T[e];
    const int sl = (int)blockIdx.x - t.blk0[e];
    const int tile = sl >> 2, sub = sl & 3;                      // sub: rows 16 sub .. 16 sub + 15 of the tile (the forward image's nf)
    const int ncit = Cn / 64, cot = tile / ncit, cit = tile - cot * ncit;
    const float* w = t.w[e];
    const float* wsc = t.wsc[e];
    const int row_len = 64 * T, total = QD_PACK_ROWS * row_len;
    const int o_base = cot * 64 + sub * 16;
    const int tid = threadIdx.x;
    if ((((uintptr_t)w) & 15) == 0) {          // a row of the slice starts at a multiple of 64 floats: every quad of it is one aligned 16-byte load
        const int nq = total / 4;
        float4 v[QD_PACK_MAXQ];
#pragma unroll
        for (int u = 0; u < QD_PACK_MAXQ; ++u) {
            const int i = tid + QD_PACK_THREADS * u;
            if (i < nq) {
                const int row = (4 * i) / row_len, col = 4 * i - row * row_len;
                v[u] = *reinterpret_cast<const float4*>(w + ((int64_t)(o_base + row) * Cn + cit * 64) * T + col);
            }
        }
#pragma unroll
        for (int u = 0; u < QD_PACK_MAXQ; ++u) {
            const int i = tid + QD_PACK_THREADS * u;
            if (i < nq) {
                int c0, c1, c2, c3;
                if (wsc) {
                    const int row = (4 * i) / row_len;
                    const float sc = wsc[(int64_t)(o_base + row) * t.wsc_stride[e]];
                    c0 = (int)rintf(v[u].x / sc); c1 = (int)rintf(v[u].y / sc); c2 = (int)rintf(v[u].z / sc); c3 = (int)rintf(v[u].w / sc);
                } else {
                    c0 = (int)rintf(v[u].x * t.wn); c1 = (int)rintf(v[u].y * t.wn); c2 = (int)rintf(v[u].z * t.wn); c3 = (int)rintf(v[u].w * t.wn);
                }
                *reinterpret_cast<u32x2*>(codes + 4 * i) = u32x2{((uint32_t)c0 & 0xffffu) | ((uint32_t)c1 << 16), ((uint32_t)c2 & 0xffffu) | ((uint32_t)c3 << 16)};
            }
        }
    } else {
        for (int i = tid; i < total; i += QD_PACK_THREADS) {
            const int row = i / row_len, col = i - row * row_len;
            const int o = o_base + row;
            const float v = w[((int64_t)o * Cn + cit * 64) * T + col];
            codes[i] = (int16_t)(int)(wsc ? rintf(v / wsc[(int64_t)o * t.wsc_stride[e]]) : rintf(v * t.wn));
        }
    }
    __syncthreads();
    const int lane = tid & 63;
    if (t.outf[e]) {
        if (t.fwd8) {          // orient 2: [cot][chunk = cit][tap][nf][lane][16 signed bytes]: o = 16 nf + (lane & 15), c = 16 (lane >> 4) + b; this slice: nf = sub
            unsigned char* dst = reinterpret_cast<unsigned char*>(t.outf[e]) + (int64_t)(cot * ncit + cit) * T * 4096;
            for (int it = tid; it < T * 64; it += QD_PACK_THREADS) {
                const int tap = it >> 6;
                const int16_t* src = codes + ((lane & 15) * 64 + (lane >> 4) * 16) * T + tap;
                uint32_t d[4] = {0u, 0u, 0u, 0u};
#pragma unroll
                for (int b_ = 0; b_ < 16; ++b_) d[b_ >> 2] |= ((uint32_t)src[b_ * T] & 0xffu) << (8 * (b_ & 3));
                *reinterpret_cast<u32x4*>(dst + (int64_t)((tap * 4 + sub) * 64 + lane) * 16) = u32x4{d[0], d[1], d[2], d[3]};
            }
        } else {               // orient 0: [cot][chunk][tap][ks][nf][lane][8 bf16]: o = 16 nf + (lane & 15), c = 32 ks + 8 (lane >> 4) + b; this slice: nf = sub
            unsigned char* dst = reinterpret_cast<unsigned char*>(t.outf[e]) + (int64_t)(cot * ncit + cit) * T * 8192;
            for (int it = tid; it < T * 128; it += QD_PACK_THREADS) {
                const int tap = it >> 7, ks = (it >> 6) & 1;
                const int16_t* src = codes + ((lane & 15) * 64 + ks * 32 + (lane >> 4) * 8) * T + tap;
                uint32_t h[8];
#pragma unroll
                for (int b_ = 0; b_ < 8; ++b_) h[b_] = mn_f2u((float)src[b_ * T]) >> 16;
                *reinterpret_cast<u32x4*>(dst + (int64_t)(((tap * 2 + ks) * 4 + sub) * 64 + lane) * 16) =
                    u32x4{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
            }
        }
    }
    if (t.outd[e]) {           // orient 1: [cit][chunk = o / 32][tap][nf][lane][8 bf16]: c = 16 nf + (lane & 15), o = 32 chunk + 8 (lane >> 4) + b;
        const int nch = O / 32;          // this slice: chunk = 2 cot + (sub >> 1), the lane groups (lane >> 4) = 2 (sub & 1), 2 (sub & 1) + 1
        unsigned char* dst0 = reinterpret_cast<unsigned char*>(t.outd[e]) + ((int64_t)(cit * nch + cot * 2 + (sub >> 1)) * T) * 4096;
        for (int it = tid; it < T * 128; it += QD_PACK_THREADS) {
            const int tap = it >> 7, nf = (it >> 5) & 3, l5 = it & 31;
            const int lg = l5 >> 4, c = nf * 16 + (l5 & 15);          // local rows 8 lg .. 8 lg + 7 of the slice
            const int16_t* src = codes + ((lg * 8) * 64 + c) * T + tap;
            uint32_t h[8];
#pragma unroll
            for (int b_ = 0; b_ < 8; ++b_) h[b_] = mn_f2u((float)src[b_ * 64 * T]) >> 16;
            const int olane = ((2 * (sub & 1) + lg) << 4) | (l5 & 15);
            *reinterpret_cast<u32x4*>(dst0 + (int64_t)((tap * 4 + nf) * 64 + olane) * 16) = u32x4{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
        }
    }
}

// ------------------------------------------------------------------------------------------------ forward
struct QdfParams {
    const unsigned char* x;       // [N][C][H][W] activation codes
    const uint16_t* wpk;          // k_qd_pack orient 0
    void* stash;                  // [N][O][Ho][Wo] int16 / int32
    int N, C, H, W, O, Ho, Wo, HW, HoWo;
    int S, PAD, TAPS;             // stride, padding, taps (9: 3 x 3, 1: 1 x 1)
    int TH, NI, PH, PW, W4;       // output rows per tile and image, images per tile, patch rows / columns per image
    int tpi, ncot, nchunks, nitems, nunits, out32, wo_shift;
    FastDiv fd_w4, fd_ph, fd_ni, fd_th, fd_ncot, fd_tpi, fd_ipt;
    QdUnits units;
    // out32 == 2: the IAO layers -- x holds SIGNED codes (xsgn), the output is fp32 y = acc * (sa[0] * sw[o * sw_stride]) + bias[o] (wqaq/iao/quantize.py:492-507)
    int xsgn, sw_stride;
    const float *sa, *sw, *bias;
    // k_qd_fwd8 with a stash output: exact per-channel sums of acc and acc^2 over the block's items (every item of a block has the same channel tile: the grid
    // is a multiple of ncot) -> stats[((blockIdx / ncot) * O + channel) * 2 + {0, 1}]: the partial layout k_qa_stats_prep reads; nullptr: none (k_qd_stats instead)
    double* stats;
    // k_qd_fwd8, IAO layers: the extrema of acc per channel over the same items -> accmm[((blockIdx / ncot) * O + channel) * 2 + {0, 1}] = min, max (INT_MAX, INT_MIN
    // for a block without a valid pixel).  y = acc * al + bias is monotone in acc, so is the BatchNorm [+ ReLU] behind it: the range of THAT activation follows from
    // these without a pass over it (mn_bn_acc_prep); nullptr: none
    int32_t* accmm;
};

template <int MF, int TPS>
__global__ __launch_bounds__(256, 2) void k_qd_fwd(const QdfParams p) {
    HIP_DYNAMIC_SHARED(float, smem)
    unsigned char* wbuf = reinterpret_cast<unsigned char*>(smem);
    unsigned char* patch = wbuf + TPS * QD_WSTEP;          // ONE weight buffer of TPS taps (a step = TPS taps x 64 input channels)
    const int tid = threadIdx.x, lane = tid & 63, wave = mn_uniform(tid >> 6), j = lane & 15, kg = lane >> 4;
    if ((int)blockIdx.x >= p.nitems) return;
    {   // the frame columns (and anything staging never writes) stay zero for the whole kernel
        const int n16 = p.NI * p.PH * p.PW * (QD_RS / 16);
        for (int i = tid; i < n16; i += 256) *reinterpret_cast<u32x4*>(patch + 16 * i) = u32x4{0u, 0u, 0u, 0u};
    }
    // staging roles: a unit = channels 4q .. 4q + 3, patch row pr of image img, input pixels 4d .. 4d + 3 (order over the threads: QdUnits)
    int u_lds[QDF_UPT], u_goff[QDF_UPT], u_pi[QDF_UPT];
#pragma unroll
    for (int i = 0; i < QDF_UPT; ++i) {
        int q, img, pr, d;
        const bool v = qd_unit(p.units, tid + 256 * i, q, img, pr, d);
        u_lds[i] = v ? ((img * p.PH + pr) * p.PW + 4 * d + p.PAD) * QD_RS + 8 * q : -1;
        u_goff[i] = v ? 4 * q * p.HW + 4 * d : 0;
        u_pi[i] = pr | (img << 8);
    }
    uint32_t preg[QDF_UPT][4];
    uint32_t pok = 0u;
    auto tile_origin = [&](int item, int& n0, int& oh0, int& cot) {
        const uint32_t tile = fd_div((uint32_t)item, p.fd_ncot);
        cot = item - (int)tile * p.ncot;
        if (p.NI == 1) { const uint32_t n = fd_div(tile, p.fd_tpi); n0 = (int)n; oh0 = ((int)tile - (int)n * p.tpi) * p.TH; }
        else { n0 = (int)tile * p.NI; oh0 = 0; }
    };
    auto fetch_patch = [&](int item, int chunk) {
        int n0, oh0, cot;
        tile_origin(item, n0, oh0, cot);
        const int ih0 = oh0 * p.S - p.PAD;
        pok = 0u;
#pragma unroll
        for (int i = 0; i < QDF_UPT; ++i) {
            int n = n0 + (u_pi[i] >> 8), ih = ih0 + (u_pi[i] & 255);
            const bool ok = u_lds[i] >= 0 && n < p.N && ih >= 0 && ih < p.H;
            n = n < p.N ? n : p.N - 1;
            ih = ih < 0 ? 0 : (ih < p.H ? ih : p.H - 1);          // rows outside the image: any valid address (written as zeros)
            const uint32_t base = (uint32_t)((n * p.C + chunk * 64) * p.H + ih) * (uint32_t)p.W + (uint32_t)u_goff[i];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) preg[i][cc] = *reinterpret_cast<const uint32_t*>(p.x + base + (uint32_t)(cc * p.HW));
            pok |= (ok ? 1u : 0u) << i;
        }
    };
    auto commit_patch = [&]() {
#pragma unroll
        for (int i = 0; i < QDF_UPT; ++i) {
            if (u_lds[i] < 0) continue;
            const bool ok = (pok >> i) & 1u;
            const uint32_t flip = p.xsgn ? 0x80808080u : 0u;          // signed codes: byte ^ 0x80 = code + 128 as an unsigned byte
            const float off = p.xsgn ? 128.f : 0.f;
            const uint32_t r0 = preg[i][0] ^ flip, r1 = preg[i][1] ^ flip, r2 = preg[i][2] ^ flip, r3 = preg[i][3] ^ flip;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float f0 = (float)((r0 >> (8 * e)) & 0xffu) - off, f1 = (float)((r1 >> (8 * e)) & 0xffu) - off;
                const float f2 = (float)((r2 >> (8 * e)) & 0xffu) - off, f3 = (float)((r3 >> (8 * e)) & 0xffu) - off;
                const u32x2 v = ok ? u32x2{mn_pack_hi16(f0, f1), mn_pack_hi16(f2, f3)} : u32x2{0u, 0u};       // integers <= 255: exact in bf16
                *reinterpret_cast<u32x2*>(patch + u_lds[i] + e * QD_RS) = v;
            }
        }
    };
    // weights: the steps of an item read consecutive TPS x 8 KB blocks of the packed image.  A step is long (TPS x 32 MFMAs per wave), so ONE LDS buffer is enough:
    // barrier -> write this step's weights (loaded into registers during the previous step; at a chunk start also the chunk's patch) -> barrier -> issue the next
    // step's loads -> contract.  Two barriers per step, no write / read overlap to manage, 24 KB instead of 2 x 8.
    u32x4 wreg[2 * TPS];
    const int stride_items = (int)gridDim.x;
    const int nsteps_chunk = p.TAPS / TPS;
    const int steps_item = p.nchunks * nsteps_chunk;
    const int my_items = (p.nitems - (int)blockIdx.x + stride_items - 1) / stride_items;
    const int total_steps = my_items * steps_item;
    auto item_wbase = [&](int item) {
        const uint32_t tile = fd_div((uint32_t)item, p.fd_ncot);
        const int cot = item - (int)tile * p.ncot;
        return reinterpret_cast<const unsigned char*>(p.wpk) + (int64_t)cot * p.nchunks * p.TAPS * QD_WSTEP + tid * 16;
    };
    const unsigned char* wsrc = item_wbase((int)blockIdx.x);      // source of the NEXT weight load
    int w_item = (int)blockIdx.x, w_left = steps_item;            // its item, steps left in that item
    auto fetch_w = [&]() {
#pragma unroll
        for (int t = 0; t < 2 * TPS; ++t) wreg[t] = *reinterpret_cast<const u32x4*>(wsrc + t * 4096);
        wsrc += TPS * QD_WSTEP;
        if (--w_left == 0) { w_item += stride_items; w_left = steps_item; if (w_item < p.nitems) wsrc = item_wbase(w_item); }
    };
    auto commit_w = [&]() {
#pragma unroll
        for (int t = 0; t < 2 * TPS; ++t) *reinterpret_cast<u32x4*>(wbuf + t * 4096 + tid * 16) = wreg[t];
    };
    // A fragment bases: pixel tp = 16 MF wave + 16 mf + j of the tile -> the slot of its first tap
    int abase[MF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
        const int tp = wave * 16 * MF + mf * 16 + j;
        const int ow = tp & (p.Wo - 1), t = tp >> p.wo_shift;
        const uint32_t img = fd_div((uint32_t)t, p.fd_th);
        const int ohl = t - (int)img * p.TH;
        abase[mf] = (((int)img * p.PH + ohl * p.S) * p.PW + ow * p.S) * QD_RS + kg * 16;
    }
    fetch_patch((int)blockIdx.x, 0);
    fetch_w();
    int gs = 0;                               // steps done by this block
    for (int item = (int)blockIdx.x; item < p.nitems; item += stride_items) {
        f32x4 acc[MF][4];
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) acc[mf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int chunk = 0; chunk < p.nchunks; ++chunk) {
            int nitem = item, nchunk = chunk + 1;
            if (nchunk == p.nchunks) { nchunk = 0; nitem += stride_items; }
            const bool have_next = nitem < p.nitems;
            for (int st = 0; st < nsteps_chunk; ++st) {
                __syncthreads();                                   // every wave is done with the previous step (and, at a chunk start, with the previous patch / the zero fill)
                if (st == 0) commit_patch();
                commit_w();
                __syncthreads();
                if (gs + 1 < total_steps) fetch_w();               // the next step's weights: in flight during this step's MFMAs
                if (st == 0 && have_next) fetch_patch(nitem, nchunk);
                ++gs;
#pragma unroll
                for (int tis = 0; tis < TPS; ++tis) {
                    const int tap = st * TPS + tis;
                    const int r = TPS == 3 ? st : 0;
                    const int toff = (r * p.PW + (tap - 3 * r)) * QD_RS;
                    const unsigned char* wb = wbuf + tis * QD_WSTEP + lane * 16;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        u32x4 b[4], a[MF];
#pragma unroll
                        for (int nf = 0; nf < 4; ++nf) b[nf] = *reinterpret_cast<const u32x4*>(wb + (ks * 4 + nf) * 1024);
#pragma unroll
                        for (int mf = 0; mf < MF; ++mf) a[mf] = *reinterpret_cast<const u32x4*>(patch + abase[mf] + toff + ks * 64);
#pragma unroll
                        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                            for (int nf = 0; nf < 4; ++nf) acc[mf][nf] = mn_mfma_bf16(a[mf], b[nf], acc[mf][nf]);
                    }
                }
            }
        }
        __syncthreads();                      // the patch memory is idle: the epilogue borrows it
        // ---- epilogue.  D[row = pixel 4 kg + r][col = channel j].  The wave transposes its 16 MF pixels x 64 channels through its own piece of the (now idle)
        // patch memory -- rows [channel][16 MF pixels] padded by 8 bytes (conflict-free 8-byte writes) -- and stores 16 bytes per lane: a store instruction
        // writes whole contiguous rows of several channels (int16: 32 MF bytes per row; int32: two passes of 32 channels, 64 MF bytes per row).
        int n0, oh0, cot;
        tile_origin(item, n0, oh0, cot);
        {
            unsigned char* scr = patch + wave * (MF * 2048 + 64 * 8);
            const int tp0 = wave * 16 * MF;                               // first tile pixel of this wave; its pixels are consecutive in ONE image when
            const int ipt = p.TH * p.Wo;                                  // 16 MF <= pixels per image and tile, else whole images
            if (!p.out32) {
                constexpr int ROW = 32 * MF + 8;
#pragma unroll
                for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                    for (int nf = 0; nf < 4; ++nf) {
                        const int v0 = (int)acc[mf][nf][0], v1 = (int)acc[mf][nf][1], v2 = (int)acc[mf][nf][2], v3 = (int)acc[mf][nf][3];
                        *reinterpret_cast<u32x2*>(scr + (nf * 16 + j) * ROW + (mf * 16 + 4 * kg) * 2) =
                            u32x2{((uint32_t)v0 & 0xffffu) | ((uint32_t)v1 << 16), ((uint32_t)v2 & 0xffffu) | ((uint32_t)v3 << 16)};
                    }
                MN_WAVE_SYNC();
                constexpr int CPR = 2 * MF, RPI = 64 / CPR;                // 16-byte chunks per row, rows per store instruction
#pragma unroll
                for (int it = 0; it < 64 / RPI; ++it) {
                    const int co = it * RPI + lane / CPR, ch = lane % CPR;
                    const unsigned char* q = scr + co * ROW + ch * 16;
                    const u32x2 a = *reinterpret_cast<const u32x2*>(q), b = *reinterpret_cast<const u32x2*>(q + 8);
                    const int tp = tp0 + ch * 8;
                    const int im = (int)fd_div((uint32_t)tp, p.fd_ipt), lp = tp - im * ipt;
                    const int n = n0 + im;
                    if (n < p.N)
                        *reinterpret_cast<u32x4*>(reinterpret_cast<int16_t*>(p.stash) + (uint32_t)((n * p.O + cot * 64 + co) * p.HoWo + oh0 * p.Wo + lp)) = u32x4{a[0], a[1], b[0], b[1]};
                }
            } else {
                constexpr int ROW = 64 * MF + 8;
                float al[4] = {1.f, 1.f, 1.f, 1.f}, bi[4] = {0.f, 0.f, 0.f, 0.f};
                if (p.out32 == 2) {
                    const float sa = p.sa[0];
#pragma unroll
                    for (int nf = 0; nf < 4; ++nf) {
                        const int o = cot * 64 + nf * 16 + j;
                        al[nf] = sa * p.sw[(int64_t)o * p.sw_stride];
                        bi[nf] = p.bias ? p.bias[o] : 0.f;
                    }
                }
#pragma unroll
                for (int half = 0; half < 2; ++half) {
#pragma unroll
                    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                        for (int n2 = 0; n2 < 2; ++n2) {
                            const int nf = half * 2 + n2;
                            unsigned char* d = scr + (n2 * 16 + j) * ROW + (mf * 16 + 4 * kg) * 4;
                            uint32_t w4[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) w4[r] = p.out32 == 2 ? mn_f2u(acc[mf][nf][r] * al[nf] + bi[nf]) : (uint32_t)(int)acc[mf][nf][r];
                            *reinterpret_cast<u32x2*>(d) = u32x2{w4[0], w4[1]};
                            *reinterpret_cast<u32x2*>(d + 8) = u32x2{w4[2], w4[3]};
                        }
                    MN_WAVE_SYNC();
                    constexpr int CPR = 4 * MF, RPI = 64 / CPR;
#pragma unroll
                    for (int it = 0; it < 32 / RPI; ++it) {
                        const int cl = it * RPI + lane / CPR, ch = lane % CPR;
                        const unsigned char* q = scr + cl * ROW + ch * 16;
                        const u32x2 a = *reinterpret_cast<const u32x2*>(q), b = *reinterpret_cast<const u32x2*>(q + 8);
                        const int tp = tp0 + ch * 4;
                        const int im = (int)fd_div((uint32_t)tp, p.fd_ipt), lp = tp - im * ipt;
                        const int n = n0 + im;
                        if (n < p.N)
                            *reinterpret_cast<u32x4*>(reinterpret_cast<int32_t*>(p.stash) + (uint32_t)((n * p.O + cot * 64 + half * 32 + cl) * p.HoWo + oh0 * p.Wo + lp)) =
                                u32x4{a[0], a[1], b[0], b[1]};
                    }
                    MN_WAVE_SYNC();
                }
            }
        }
        if (item + stride_items < p.nitems) {
            __syncthreads();                  // every wave is done with its transposition scratch
            for (int i = tid; i < (4 * (MF * 2048 + 512)) / 16; i += 256) *reinterpret_cast<u32x4*>(patch + 16 * i) = u32x4{0u, 0u, 0u, 0u};      // the zero frame again
        }
    }
}

// ------------------------------------------------------------------------------------------------ forward on the int8 matrix cores
// The same organisation on v_mfma_i32_16x16x64_i8 (twice the bf16 rate, exact i32 accumulation for any K): the codes stay BYTES from HBM to the MFMA -- the
// patch is [pixel slot][64 c] signed bytes (80-byte slots: a lane's A fragment = 16 consecutive channels of its pixel, one conflict-free b128 read), the weights
// arrive as k_qd_pack orient 2 (4 KB per tap and chunk), one MFMA K-step = a whole 64-channel chunk of one tap.  Half the LDS bytes per MAC of the bf16 form.
// Requires codes in [-128, 127]: DoReFa activations of <= 7 bits (0 .. 127), IAO activations (signed), weight codes of <= 7 (DoReFa) / <= 8 (IAO) bits.
#define QD8_RS 80
#define QD8_WSTEP 4096
template <int MF, int TPS>
__global__ __launch_bounds__(256, 2) void k_qd_fwd8(const QdfParams p) {
    HIP_DYNAMIC_SHARED(float, smem)
    unsigned char* wbuf = reinterpret_cast<unsigned char*>(smem);
    unsigned char* patch = wbuf + TPS * QD8_WSTEP;
    const int tid = threadIdx.x, lane = tid & 63, wave = mn_uniform(tid >> 6), j = lane & 15, kg = lane >> 4;
    if ((int)blockIdx.x >= p.nitems) return;
    {
        const int n16 = p.NI * p.PH * p.PW * (QD8_RS / 16);
        for (int i = tid; i < n16; i += 256) *reinterpret_cast<u32x4*>(patch + 16 * i) = u32x4{0u, 0u, 0u, 0u};
    }
    int u_lds[QDF_UPT], u_goff[QDF_UPT], u_pi[QDF_UPT];
#pragma unroll
    for (int i = 0; i < QDF_UPT; ++i) {
        int q, img, pr, d;
        const bool v = qd_unit(p.units, tid + 256 * i, q, img, pr, d);
        u_lds[i] = v ? ((img * p.PH + pr) * p.PW + 4 * d + p.PAD) * QD8_RS + 4 * q : -1;
        u_goff[i] = v ? 4 * q * p.HW + 4 * d : 0;
        u_pi[i] = pr | (img << 8);
    }
    uint32_t preg[QDF_UPT][4];
    uint32_t pok = 0u;
    auto tile_origin = [&](int item, int& n0, int& oh0, int& cot) {
        const uint32_t tile = fd_div((uint32_t)item, p.fd_ncot);
        cot = item - (int)tile * p.ncot;
        if (p.NI == 1) { const uint32_t n = fd_div(tile, p.fd_tpi); n0 = (int)n; oh0 = ((int)tile - (int)n * p.tpi) * p.TH; }
        else { n0 = (int)tile * p.NI; oh0 = 0; }
    };
    auto fetch_patch = [&](int item, int chunk) {
        int n0, oh0, cot;
        tile_origin(item, n0, oh0, cot);
        const int ih0 = oh0 * p.S - p.PAD;
        pok = 0u;
#pragma unroll
        for (int i = 0; i < QDF_UPT; ++i) {
            int n = n0 + (u_pi[i] >> 8), ih = ih0 + (u_pi[i] & 255);
            const bool ok = u_lds[i] >= 0 && n < p.N && ih >= 0 && ih < p.H;
            n = n < p.N ? n : p.N - 1;
            ih = ih < 0 ? 0 : (ih < p.H ? ih : p.H - 1);
            const uint32_t base = (uint32_t)((n * p.C + chunk * 64) * p.H + ih) * (uint32_t)p.W + (uint32_t)u_goff[i];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) preg[i][cc] = *reinterpret_cast<const uint32_t*>(p.x + base + (uint32_t)(cc * p.HW));
            pok |= (ok ? 1u : 0u) << i;
        }
    };
    auto commit_patch = [&]() {
#pragma unroll
        for (int i = 0; i < QDF_UPT; ++i) {
            if (u_lds[i] < 0) continue;
            const bool ok = (pok >> i) & 1u;
            // 4 channels x 4 pixels of bytes, transposed: one dword (4 channels) per pixel
            const uint32_t lo01 = mn_perm(preg[i][1], preg[i][0], 0x05010400u), hi01 = mn_perm(preg[i][1], preg[i][0], 0x07030602u);
            const uint32_t lo23 = mn_perm(preg[i][3], preg[i][2], 0x05010400u), hi23 = mn_perm(preg[i][3], preg[i][2], 0x07030602u);
            const uint32_t o0 = mn_perm(lo23, lo01, 0x05040100u), o1 = mn_perm(lo23, lo01, 0x07060302u);
            const uint32_t o2 = mn_perm(hi23, hi01, 0x05040100u), o3 = mn_perm(hi23, hi01, 0x07060302u);
            unsigned char* d = patch + u_lds[i];
            *reinterpret_cast<uint32_t*>(d) = ok ? o0 : 0u;
            *reinterpret_cast<uint32_t*>(d + QD8_RS) = ok ? o1 : 0u;
            *reinterpret_cast<uint32_t*>(d + 2 * QD8_RS) = ok ? o2 : 0u;
            *reinterpret_cast<uint32_t*>(d + 3 * QD8_RS) = ok ? o3 : 0u;
        }
    };
    u32x4 wreg[TPS];
    const int stride_items = (int)gridDim.x;
    const int nsteps_chunk = p.TAPS / TPS;
    const int steps_item = p.nchunks * nsteps_chunk;
    const int my_items = (p.nitems - (int)blockIdx.x + stride_items - 1) / stride_items;
    const int total_steps = my_items * steps_item;
    auto item_wbase = [&](int item) {
        const uint32_t tile = fd_div((uint32_t)item, p.fd_ncot);
        const int cot = item - (int)tile * p.ncot;
        return reinterpret_cast<const unsigned char*>(p.wpk) + (int64_t)cot * p.nchunks * p.TAPS * QD8_WSTEP + tid * 16;
    };
    const unsigned char* wsrc = item_wbase((int)blockIdx.x);
    int w_item = (int)blockIdx.x, w_left = steps_item;
    auto fetch_w = [&]() {
#pragma unroll
        for (int t = 0; t < TPS; ++t) wreg[t] = *reinterpret_cast<const u32x4*>(wsrc + t * QD8_WSTEP);
        wsrc += TPS * QD8_WSTEP;
        if (--w_left == 0) { w_item += stride_items; w_left = steps_item; if (w_item < p.nitems) wsrc = item_wbase(w_item); }
    };
    auto commit_w = [&]() {
#pragma unroll
        for (int t = 0; t < TPS; ++t) *reinterpret_cast<u32x4*>(wbuf + t * QD8_WSTEP + tid * 16) = wreg[t];
    };
    int abase[MF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
        const int tp = wave * 16 * MF + mf * 16 + j;
        const int ow = tp & (p.Wo - 1), t = tp >> p.wo_shift;
        const uint32_t img = fd_div((uint32_t)t, p.fd_th);
        const int ohl = t - (int)img * p.TH;
        abase[mf] = (((int)img * p.PH + ohl * p.S) * p.PW + ow * p.S) * QD8_RS + kg * 16;
    }
    fetch_patch((int)blockIdx.x, 0);
    fetch_w();
    int gs = 0;
    double st1[4] = {0.0, 0.0, 0.0, 0.0}, st2[4] = {0.0, 0.0, 0.0, 0.0};          // this lane's share of sum acc, sum acc^2 of channels 16 nf + j (exact: < 2^53)
    int amn[4] = {INT_MAX, INT_MAX, INT_MAX, INT_MAX}, amx[4] = {INT_MIN, INT_MIN, INT_MIN, INT_MIN};          // ... and of their extrema (valid pixels only)
    for (int item = (int)blockIdx.x; item < p.nitems; item += stride_items) {
        i32x4 acc[MF][4];
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) acc[mf][nf] = i32x4{0, 0, 0, 0};
        for (int chunk = 0; chunk < p.nchunks; ++chunk) {
            int nitem = item, nchunk = chunk + 1;
            if (nchunk == p.nchunks) { nchunk = 0; nitem += stride_items; }
            const bool have_next = nitem < p.nitems;
            for (int st = 0; st < nsteps_chunk; ++st) {
                __syncthreads();
                if (st == 0) commit_patch();
                commit_w();
                __syncthreads();
                if (gs + 1 < total_steps) fetch_w();
                if (st == 0 && have_next) fetch_patch(nitem, nchunk);
                ++gs;
#pragma unroll
                for (int tis = 0; tis < TPS; ++tis) {
                    const int tap = st * TPS + tis;
                    const int r = TPS == 9 ? tis / 3 : (TPS == 3 ? st : 0);
                    const int toff = (r * p.PW + (tap - 3 * r)) * QD8_RS;
                    const unsigned char* wb = wbuf + tis * QD8_WSTEP + lane * 16;
                    u32x4 b[4], a[MF];
#pragma unroll
                    for (int nf = 0; nf < 4; ++nf) b[nf] = *reinterpret_cast<const u32x4*>(wb + nf * 1024);
#pragma unroll
                    for (int mf = 0; mf < MF; ++mf) a[mf] = *reinterpret_cast<const u32x4*>(patch + abase[mf] + toff);
#pragma unroll
                    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                        for (int nf = 0; nf < 4; ++nf) acc[mf][nf] = mn_mfma_i8(a[mf], b[nf], acc[mf][nf]);
                }
            }
        }
        __syncthreads();
        // ---- epilogue: as k_qd_fwd (transposition through the wave's piece of the idle patch memory, 16-byte stores), on exact i32 values
        int n0, oh0, cot;
        tile_origin(item, n0, oh0, cot);
        if (p.stats) {          // (pixels of images beyond N were staged as zeros: they add nothing)
#pragma unroll
            for (int nf = 0; nf < 4; ++nf) {
                int s1 = 0;
                double s2 = 0.0;
#pragma unroll
                for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const int v = acc[mf][nf][r]; s1 += v; const double dv = (double)v; s2 = fma(dv, dv, s2); }
                st1[nf] += (double)s1; st2[nf] += s2;
            }
        }
        if (p.accmm) {          // (a lane's four pixels 4 kg .. 4 kg + 3 of fragment mf lie in one image: the pixels per image and tile are a multiple of 4)
#pragma unroll
            for (int mf = 0; mf < MF; ++mf) {
                const int tpm = wave * 16 * MF + mf * 16 + 4 * kg;
                if (n0 + (int)fd_div((uint32_t)tpm, p.fd_ipt) < p.N) {
#pragma unroll
                    for (int nf = 0; nf < 4; ++nf)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const int v = acc[mf][nf][r]; amn[nf] = v < amn[nf] ? v : amn[nf]; amx[nf] = v > amx[nf] ? v : amx[nf]; }
                }
            }
        }
        {
            unsigned char* scr = wbuf + wave * (MF * 2048 + 64 * 8);          // weights and patch are both idle here: the scratch starts at the weight buffer
            const int tp0 = wave * 16 * MF;
            const int ipt = p.TH * p.Wo;
            if (!p.out32) {
                constexpr int ROW = 32 * MF + 8;
#pragma unroll
                for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                    for (int nf = 0; nf < 4; ++nf)
                        *reinterpret_cast<u32x2*>(scr + (nf * 16 + j) * ROW + (mf * 16 + 4 * kg) * 2) =
                            u32x2{((uint32_t)acc[mf][nf][0] & 0xffffu) | ((uint32_t)acc[mf][nf][1] << 16), ((uint32_t)acc[mf][nf][2] & 0xffffu) | ((uint32_t)acc[mf][nf][3] << 16)};
                MN_WAVE_SYNC();
                constexpr int CPR = 2 * MF, RPI = 64 / CPR;
#pragma unroll
                for (int it = 0; it < 64 / RPI; ++it) {
                    const int co = it * RPI + lane / CPR, ch = lane % CPR;
                    const unsigned char* q = scr + co * ROW + ch * 16;
                    const u32x2 a = *reinterpret_cast<const u32x2*>(q), b = *reinterpret_cast<const u32x2*>(q + 8);
                    const int tp = tp0 + ch * 8;
                    const int im = (int)fd_div((uint32_t)tp, p.fd_ipt), lp = tp - im * ipt;
                    const int n = n0 + im;
                    if (n < p.N)
                        *reinterpret_cast<u32x4*>(reinterpret_cast<int16_t*>(p.stash) + (uint32_t)((n * p.O + cot * 64 + co) * p.HoWo + oh0 * p.Wo + lp)) = u32x4{a[0], a[1], b[0], b[1]};
                }
            } else {
                constexpr int ROW = 64 * MF + 8;
                float al[4] = {1.f, 1.f, 1.f, 1.f}, bi[4] = {0.f, 0.f, 0.f, 0.f};
                if (p.out32 == 2) {
                    const float sa = p.sa[0];
#pragma unroll
                    for (int nf = 0; nf < 4; ++nf) {
                        const int o = cot * 64 + nf * 16 + j;
                        al[nf] = sa * p.sw[(int64_t)o * p.sw_stride];
                        bi[nf] = p.bias ? p.bias[o] : 0.f;
                    }
                }
#pragma unroll
                for (int half = 0; half < 2; ++half) {
#pragma unroll
                    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                        for (int n2 = 0; n2 < 2; ++n2) {
                            const int nf = half * 2 + n2;
                            unsigned char* d = scr + (n2 * 16 + j) * ROW + (mf * 16 + 4 * kg) * 4;
                            uint32_t w4[4];
#pragma unroll
                            for (int r = 0; r < 4; ++r) w4[r] = p.out32 == 2 ? mn_f2u((float)acc[mf][nf][r] * al[nf] + bi[nf]) : (uint32_t)acc[mf][nf][r];
                            *reinterpret_cast<u32x2*>(d) = u32x2{w4[0], w4[1]};
                            *reinterpret_cast<u32x2*>(d + 8) = u32x2{w4[2], w4[3]};
                        }
                    MN_WAVE_SYNC();
                    constexpr int CPR = 4 * MF, RPI = 64 / CPR;
#pragma unroll
                    for (int it = 0; it < 32 / RPI; ++it) {
                        const int cl = it * RPI + lane / CPR, ch = lane % CPR;
                        const unsigned char* q = scr + cl * ROW + ch * 16;
                        const u32x2 a = *reinterpret_cast<const u32x2*>(q), b = *reinterpret_cast<const u32x2*>(q + 8);
                        const int tp = tp0 + ch * 4;
                        const int im = (int)fd_div((uint32_t)tp, p.fd_ipt), lp = tp - im * ipt;
                        const int n = n0 + im;
                        if (n < p.N)
                            *reinterpret_cast<u32x4*>(reinterpret_cast<int32_t*>(p.stash) + (uint32_t)((n * p.O + cot * 64 + half * 32 + cl) * p.HoWo + oh0 * p.Wo + lp)) =
                                u32x4{a[0], a[1], b[0], b[1]};
                    }
                    MN_WAVE_SYNC();
                }
            }
        }
        if (item + stride_items < p.nitems) {
            __syncthreads();
            constexpr int over = 4 * (MF * 2048 + 512) - TPS * QD8_WSTEP;          // bytes of the patch the scratch covered: the zero frame again
            for (int i = tid; i < over / 16; i += 256) *reinterpret_cast<u32x4*>(patch + 16 * i) = u32x4{0u, 0u, 0u, 0u};
        }
    }
    if (p.stats) {
        __syncthreads();
        double* red = reinterpret_cast<double*>(smem);          // [wave][kg][64 channels][2]: 16 KB of the (idle) dynamic LDS
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            double* d = red + (((wave * 4 + kg) * 64) + nf * 16 + j) * 2;
            d[0] = st1[nf]; d[1] = st2[nf];
        }
        __syncthreads();
        if (tid < 64) {
            double a1 = 0.0, a2 = 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) { a1 += red[(q * 64 + tid) * 2]; a2 += red[(q * 64 + tid) * 2 + 1]; }
            const int sp = (int)blockIdx.x / p.ncot, cot = (int)blockIdx.x - sp * p.ncot;
            double* dst = p.stats + ((int64_t)sp * p.O + cot * 64 + tid) * 2;
            dst[0] = a1; dst[1] = a2;
        }
    }
    if (p.accmm) {
        __syncthreads();
        int* red = reinterpret_cast<int*>(smem);          // [wave][kg][64 channels][2]
#pragma unroll
        for (int nf = 0; nf < 4; ++nf) {
            int* d = red + (((wave * 4 + kg) * 64) + nf * 16 + j) * 2;
            d[0] = amn[nf]; d[1] = amx[nf];
        }
        __syncthreads();
        if (tid < 64) {
            int lo = INT_MAX, hi = INT_MIN;
#pragma unroll
            for (int q = 0; q < 16; ++q) { const int a = red[(q * 64 + tid) * 2], b = red[(q * 64 + tid) * 2 + 1]; lo = a < lo ? a : lo; hi = b > hi ? b : hi; }
            const int sp = (int)blockIdx.x / p.ncot, cot = (int)blockIdx.x - sp * p.ncot;
            int32_t* dst = p.accmm + ((int64_t)sp * p.O + cot * 64 + tid) * 2;
            dst[0] = lo; dst[1] = hi;
        }
    }
}

// exact integer sums of the stash per channel: part[(sp * O + c) * 2 + {0, 1}] = sum acc, sum acc^2 of split sp
template <int IN32>
__global__ __launch_bounds__(256) void k_qd_stats(const void* __restrict__ stash, int N, int O, int HW, double* __restrict__ part) {
    __shared__ double scd[16];
    const int c = blockIdx.x, sp = blockIdx.y, S = gridDim.y;
    const int HW8 = HW >> 3;
    const int64_t n8 = (int64_t)N * HW8;
    long long s1 = 0, s2 = 0;
    for (int64_t i = (int64_t)sp * 256 + threadIdx.x; i < n8; i += (int64_t)S * 256) {
        const int64_t n = i / HW8;
        const int64_t off = (n * O + c) * HW + (i - n * HW8) * 8;
        int v[8];
        if (IN32) {
            const u32x4 a = *reinterpret_cast<const u32x4*>(reinterpret_cast<const int32_t*>(stash) + off), b = *reinterpret_cast<const u32x4*>(reinterpret_cast<const int32_t*>(stash) + off + 4);
#pragma unroll
            for (int d = 0; d < 4; ++d) { v[d] = (int)a[d]; v[4 + d] = (int)b[d]; }
        } else {
            const u32x4 u = *reinterpret_cast<const u32x4*>(reinterpret_cast<const int16_t*>(stash) + off);
#pragma unroll
            for (int d = 0; d < 4; ++d) { v[2 * d] = (int)(int16_t)(u[d] & 0xffffu); v[2 * d + 1] = (int)(int16_t)(u[d] >> 16); }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1 += v[e]; s2 += (long long)v[e] * v[e]; }
    }
    const double r1 = block_reduce((double)s1, OpAddD(), 0.0, scd);          // partial sums are integers below 2^53: exact
    const double r2 = block_reduce((double)s2, OpAddD(), 0.0, scd);
    if (threadIdx.x == 0) { part[((int64_t)sp * O + c) * 2] = r1; part[((int64_t)sp * O + c) * 2 + 1] = r2; }
}

// ------------------------------------------------------------------------------------------------ host side: forward
struct QdfPlan { QdfParams p; int MF, grid, i8, TPS; size_t lds; int64_t off_part, off_scale, ws_bytes; int S_stats; };
static int qd_log2(int v) { int s = 0; while ((1 << s) < v) ++s; return (1 << s) == v ? s : -1; }
static int qd_geom_ok(const mn_conv_geom* g) {
    if (!g || g->groups != 1 || g->in_shuffle > 1 || g->dil_h != 1 || g->dil_w != 1 || g->C % 64 || g->O % 64 || g->N < 1) return 0;
    if (g->KH == 3 && g->KW == 3 && g->pad_h == 1 && g->pad_w == 1 && g->stride_h == g->stride_w && (g->stride_h == 1 || g->stride_h == 2)) { }
    else if (g->KH == 1 && g->KW == 1 && g->pad_h == 0 && g->pad_w == 0 && g->stride_h == 2 && g->stride_w == 2) { }
    else return 0;
    if (g->W % 4 || g->H % g->stride_h || g->W % g->stride_w) return 0;
    return 1;
}
static int plan_qdf(const mn_conv_geom* g, int out32, QdfPlan* pl, int i8 = 0) {
    if (!qd_geom_ok(g)) return 0;
    QdfParams& p = pl->p;
    const int RS = i8 ? QD8_RS : QD_RS;
    pl->i8 = i8;
    p.N = g->N; p.C = g->C; p.H = g->H; p.W = g->W; p.O = g->O; p.S = g->stride_h; p.PAD = g->pad_h; p.TAPS = g->KH * g->KW;
    p.Ho = (g->H + 2 * g->pad_h - g->KH) / p.S + 1; p.Wo = (g->W + 2 * g->pad_w - g->KW) / p.S + 1;
    p.HW = g->H * g->W; p.HoWo = p.Ho * p.Wo;
    p.wo_shift = qd_log2(p.Wo);
    if (p.wo_shift < 2 || p.Wo > 32 || p.HoWo % 8) return 0;
    if ((int64_t)g->N * g->C * p.HW >= ((int64_t)1 << 31) || (int64_t)g->N * g->O * p.HoWo >= ((int64_t)1 << 31)) return 0;          // 32-bit element offsets
    int MF = 0;
    for (int mf = 4; mf >= 1; mf >>= 1) {
        const int BM = 64 * mf;
        int TH, NI;
        if (p.HoWo >= BM) { if (BM % p.Wo) continue; TH = BM / p.Wo; if (p.Ho % TH) continue; NI = 1; }
        else { if (BM % p.HoWo) continue; NI = BM / p.HoWo; TH = p.Ho; }
        const int PH = (TH - 1) * p.S + g->KH, PW = g->W + 2 * g->pad_w;          // staging writes whole input rows
        if (PH > 255 || NI > 255) continue;
        const int64_t patch = (int64_t)NI * PH * PW * RS;
        const QdUnits units = qd_make_units(g->W / 4, PH, NI, 16);
        if (patch > 56 * 1024 || units.nunits > 256 * QDF_UPT || (!i8 && patch < 4 * (mf * 2048 + 512))) continue;      // (the epilogue borrows the patch memory)
        MF = mf; p.TH = TH; p.NI = NI; p.PH = PH; p.PW = PW; p.nunits = units.nunits; p.units = units;
        break;
    }
    if (!MF) return 0;
    pl->MF = MF;
    p.W4 = g->W / 4;
    p.tpi = p.NI == 1 ? p.Ho / p.TH : 1;
    const int ntiles = p.NI == 1 ? g->N * p.tpi : (g->N + p.NI - 1) / p.NI;
    p.ncot = g->O / 64; p.nchunks = g->C / 64; p.nitems = ntiles * p.ncot; p.out32 = out32;
    p.fd_w4 = make_fastdiv((uint32_t)p.W4); p.fd_ph = make_fastdiv((uint32_t)p.PH); p.fd_ni = make_fastdiv((uint32_t)p.NI);
    p.fd_th = make_fastdiv((uint32_t)p.TH); p.fd_ncot = make_fastdiv((uint32_t)p.ncot); p.fd_tpi = make_fastdiv((uint32_t)p.tpi);
    p.fd_ipt = make_fastdiv((uint32_t)(p.TH * p.Wo));
    int tgt = 512;
    pl->grid = p.nitems < tgt ? p.nitems : tgt;
    if (pl->grid > 512) pl->grid = 512;
    pl->grid -= pl->grid % p.ncot;          // every item of a block then has the block's channel tile (item % ncot == blockIdx % ncot): the epilogue statistics rely on it
    if (pl->grid < p.ncot) pl->grid = p.ncot;
    pl->TPS = p.TAPS == 9 ? 3 : 1;
    pl->lds = (size_t)pl->TPS * (i8 ? QD8_WSTEP : QD_WSTEP) + (size_t)p.NI * p.PH * p.PW * RS;
    if (i8 && pl->lds < (size_t)4 * (MF * 2048 + 512)) pl->lds = (size_t)4 * (MF * 2048 + 512);          // its epilogue scratch starts at the weight buffer
    if (i8 && pl->lds < 16384) pl->lds = 16384;                                                           // (and the 16 KB of the statistics hand-over)
    // workspace: packed weights | statistics partials [S][O][2] doubles | per-channel weight scale [O]
    const int64_t pack_bytes = ((int64_t)g->O * g->C * p.TAPS * 2 + 255) / 256 * 256;
    int S = (2048 + g->O - 1) / g->O;
    const int64_t maxS = ((int64_t)g->N * (p.HoWo / 8) + 255) / 256;
    if (S > maxS) S = (int)maxS;
    if (S > 32) S = 32;
    if (S < 1) S = 1;
    pl->S_stats = S;
    pl->off_part = pack_bytes;
    const int64_t nslots = S > 512 ? S : 512;          // (the int8 forward writes one partial per block of its channel tile: up to 512)
    pl->off_scale = pl->off_part + (nslots * g->O * 2 * 8 + 255) / 256 * 256;
    pl->ws_bytes = pl->off_scale + ((int64_t)g->O * 4 + 255) / 256 * 256;
    return 1;
}
static void qd_launch_pack(const float* w, uint16_t* out, int O, int C, int T, int w_bits, int orient, hipStream_t s, const float* wsc = nullptr, int wsc_stride = 0) {
    QdPackParams k;
    k.w = w; k.out = out; k.O = O; k.C = C; k.T = T; k.wn = (float)((1ll << w_bits) - 1); k.orient = orient; k.ngroups = (int64_t)O * C * T / (orient == 2 ? 16 : 8);
    k.wsc = wsc; k.wsc_stride = wsc_stride;
    hipLaunchKernelGGL(k_qd_pack, dim3((unsigned)((k.ngroups + 255) / 256)), dim3(256), 0, s, k);
}
template <int MF, int TPS>
static void qd_launch_fwd8_t(const QdfPlan& pl, hipStream_t s) {
    raise_lds_limit((const void*)k_qd_fwd8<MF, TPS>, pl.lds);
    hipLaunchKernelGGL((k_qd_fwd8<MF, TPS>), dim3(pl.grid), dim3(256), pl.lds, s, pl.p);
}
static void qd_launch_fwd(const QdfPlan& pl, hipStream_t s) {
    const QdfParams& p = pl.p;
    if (pl.i8) {
        if (pl.TPS == 9) { if (pl.MF == 4) qd_launch_fwd8_t<4, 9>(pl, s); else if (pl.MF == 2) qd_launch_fwd8_t<2, 9>(pl, s); else qd_launch_fwd8_t<1, 9>(pl, s); }
        else if (pl.TPS == 3) { if (pl.MF == 4) qd_launch_fwd8_t<4, 3>(pl, s); else if (pl.MF == 2) qd_launch_fwd8_t<2, 3>(pl, s); else qd_launch_fwd8_t<1, 3>(pl, s); }
        else { if (pl.MF == 4) qd_launch_fwd8_t<4, 1>(pl, s); else if (pl.MF == 2) qd_launch_fwd8_t<2, 1>(pl, s); else qd_launch_fwd8_t<1, 1>(pl, s); }
        return;
    }
    if (p.TAPS == 9) {
        if (pl.MF == 4) { raise_lds_limit((const void*)k_qd_fwd<4, 3>, pl.lds); hipLaunchKernelGGL((k_qd_fwd<4, 3>), dim3(pl.grid), dim3(256), pl.lds, s, p); }
        else if (pl.MF == 2) { raise_lds_limit((const void*)k_qd_fwd<2, 3>, pl.lds); hipLaunchKernelGGL((k_qd_fwd<2, 3>), dim3(pl.grid), dim3(256), pl.lds, s, p); }
        else { raise_lds_limit((const void*)k_qd_fwd<1, 3>, pl.lds); hipLaunchKernelGGL((k_qd_fwd<1, 3>), dim3(pl.grid), dim3(256), pl.lds, s, p); }
    } else {
        if (pl.MF == 4) { raise_lds_limit((const void*)k_qd_fwd<4, 1>, pl.lds); hipLaunchKernelGGL((k_qd_fwd<4, 1>), dim3(pl.grid), dim3(256), pl.lds, s, p); }
        else if (pl.MF == 2) { raise_lds_limit((const void*)k_qd_fwd<2, 1>, pl.lds); hipLaunchKernelGGL((k_qd_fwd<2, 1>), dim3(pl.grid), dim3(256), pl.lds, s, p); }
        else { raise_lds_limit((const void*)k_qd_fwd<1, 1>, pl.lds); hipLaunchKernelGGL((k_qd_fwd<1, 1>), dim3(pl.grid), dim3(256), pl.lds, s, p); }
    }
}
// the forward runs on the int8 matrix cores whenever the weight codes fit signed bytes (activation codes always do: DoReFa <= 7 bits unsigned, IAO signed)
static int qd_fwd_i8(const mn_wq* wq) {
    return wq && ((wq->mode == MN_WQ_DOREFA && wq->bits <= 7) || (wq->mode == MN_WQ_IAO && wq->bits <= 8));
}
// the stash of a dense layer is 32 bits wide when K * amax * wmax does not fit 16
int qd_stash32(const mn_conv_geom* g, const mn_wq* wq, int a_bits) {
    const int64_t K = (int64_t)g->C * g->KH * g->KW, wmax = (1ll << wq->bits) - 1, amax = (1ll << a_bits) - 1;
    return K * amax * wmax > 32767;
}
int qd_fwd_supported(const mn_conv_geom* g, const mn_wq* wq, int a_bits) {
    if (!wq || wq->mode != MN_WQ_DOREFA || wq->bits < 2 || wq->bits > 8 || a_bits < 2 || a_bits > 7 || !g) return 0;
    const int64_t K = (int64_t)g->C * g->KH * g->KW, wmax = (1ll << wq->bits) - 1, amax = (1ll << a_bits) - 1;
    const int i8 = qd_fwd_i8(wq);
    if (!i8 && K * amax * wmax >= (1ll << 24)) return 0;          // bf16 forward: the fp32 accumulation of integer products must stay exact (i32 always is)
    if (K * amax * wmax >= (1ll << 31)) return 0;
    QdfPlan pl;
    return plan_qdf(g, 0, &pl, i8);
}
int64_t qd_fwd_ws_bytes(const mn_conv_geom* g) { QdfPlan pl; return (plan_qdf(g, 0, &pl, 0) || plan_qdf(g, 0, &pl, 1)) ? pl.ws_bytes : 0; }          // (the layout does not depend on the variant)
// conv on activation codes -> stash (int16 / int32 by qd_stash32) + statistics partials; *parts / *nparts / *rowscale: what qa_launch_stats_prep reads
int qd_fwd_stash(const mn_conv_geom* g, const mn_wq* wq, const uint8_t* x, int a_bits, const float* w, void* stash, void* ws, int64_t ws_bytes, hipStream_t s,
                 const double** parts, int* nparts, float** rowscale) {
    QdfPlan pl;
    const int out32 = qd_stash32(g, wq, a_bits), i8 = qd_fwd_i8(wq);
    if (!qd_fwd_supported(g, wq, a_bits) || !plan_qdf(g, out32, &pl, i8) || (((uintptr_t)x) & 3) || !aligned16(stash) || !w) MN_FAIL(MN_ENOTSUP, "mn_qconv_bnq_fwd_stash(dense): geometry not covered");
    if (!ws || ws_bytes < pl.ws_bytes || !aligned16(ws)) MN_FAIL(MN_ENOSPC, "mn_qconv_bnq_fwd_stash(dense): workspace too small");
    QdfParams& p = pl.p;
    const uint16_t* wpk = reinterpret_cast<const uint16_t*>(wq->packed_fwd);
    if (!wpk) { qd_launch_pack(w, reinterpret_cast<uint16_t*>(ws), g->O, g->C, p.TAPS, wq->bits, i8 ? 2 : 0, s); wpk = reinterpret_cast<const uint16_t*>(ws); }
    p.x = x; p.wpk = wpk; p.stash = stash; p.xsgn = 0; p.sa = p.sw = p.bias = nullptr; p.sw_stride = 0;
    double* part = reinterpret_cast<double*>((char*)ws + pl.off_part);
    const int epi_stats = i8;
    p.stats = epi_stats ? part : nullptr;
    p.accmm = nullptr;
    mn_set_last_kernel(i8 ? "k_qd_fwd8<%d, %d>" : "k_qd_fwd<%d, %d>", pl.MF, pl.TPS);
    { const double nx = (double)g->N * g->C * p.HW, ny = (double)g->N * g->O * p.HoWo; mn_prof_bytes(nx * p.ncot + (out32 ? 4.0 : 2.0) * ny); mn_prof_flops(2.0 * ny * g->C * p.TAPS); }
    mn_prof_begin(s);
    qd_launch_fwd(pl, s);
    mn_prof_end(s);
    if (!epi_stats) {
        const dim3 sgrid((unsigned)g->O, (unsigned)pl.S_stats);
        if (out32) hipLaunchKernelGGL(k_qd_stats<1>, sgrid, dim3(256), 0, s, (const void*)stash, (int)g->N, (int)g->O, p.HoWo, part);
        else hipLaunchKernelGGL(k_qd_stats<0>, sgrid, dim3(256), 0, s, (const void*)stash, (int)g->N, (int)g->O, p.HoWo, part);
    }
    *parts = part; *nparts = epi_stats ? pl.grid / p.ncot : pl.S_stats; *rowscale = reinterpret_cast<float*>((char*)ws + pl.off_scale);
    MN_CHECK_LAUNCH("mn_qconv_bnq_fwd_stash(dense)");
    return MN_OK;
}

// The fp32 gradient operand of the two backward kernels as bf16 TERMS: 3 = exact (three truncation terms carry all 24 significant bits), 2 (default, round 5) =
// round-to-nearest hi + round-to-nearest remainder: |error| <= 2^-18 |gy| per element (3.8e-6; measured parity in DESIGN / profiles/parity_r05.json), one third fewer
// matrix passes and LDS plane reads.  MN_QD_TERMS=3 restores the exact split.
static int qd_terms() { return mn_grad_terms(); }
// ================================================================================================ backward-data
//   dq[n][c][ih][iw] = (1 / n_w) * sum over (o, r, s) of wcode[o][c][r][s] * gy[n][o][oh][ow],   ih = oh S + r - P, iw = ow S + s - P
// The same organisation with the roles of the channel axes swapped: the staged patch is gy (fp32, three exact bf16 terms: three LDS planes of
// [pixel slot][32 o] with 80-byte slots, one-pixel zero frame), the contraction walks the OUTPUT channels in chunks of 32 (one MFMA K), a step is
// one kernel row (3 taps, 12 KB of weight fragments), a wave owns 16 MF gy-domain pixels x 64 input channels.
//   S = 1: dq pixel (ih, iw) meets gy pixel (ih + 1 - r, iw + 1 - s): tap (r, s) is the patch offset (2 - r, 2 - s) from the window's corner.
//   S = 2: a gy-domain CELL (oh', ow') owns the four dq pixels (2 oh' + ph, 2 ow' + pw); tap row r belongs to ph = (r != 1) and meets gy row
//          oh' + (r == 0), columns alike: four accumulator sets [ph][pw], 1 + 2 + 2 + 4 = 9 taps per cell -- no multiplication by the zeros of a
//          dilated gy; the two column parities of a row are interleaved in registers: 8 consecutive dq pixels per lane and channel.
//          1 x 1 / stride 2 (the shortcut): only (ph, pw) = (0, 0) receives anything; the other three quarters are written as zeros.
// No clip-STE epilogue: the consumer of dq (mn_qa_bwd_* / mn_qr_bwd_*) applies the quantizer's STE where it recomputes the activation.
#define QDD_RS 80
#define QDD_UPT_OF(mf) ((mf) == 4 ? 3 : 2)
struct QddParams {
    const float* gy;              // [N][O][Hg][Wg]
    const uint16_t* wpk;          // k_qd_pack orient 1
    float* dx;                    // [N][C][S Hg][S Wg]
    float wscale;
    const float* wsc;             // IAO: per-output-channel weight scale (stride wsc_stride floats, 0: per layer), folded into gy before the split; nullptr: none
    int wsc_stride;
    const float* ste_x;           // IAO: the conv's fp32 input -- the activation quantizer's clip-STE (ref 163-168, 232) is applied while dx is stored; nullptr: none
    const float* ste_qp;          // {scale, zero point, lo, hi} on the device
    const unsigned char* ste_mask; // IAO: the same decisions as bits (mn_actq.ste_mask, written by the forward's k_qd_iao_codes): bit e of byte i = element 8 i + e passes; replaces ste_x
    const float* dx_add;          // nullable: added to dx in the store, after the clip-STE (mn_actq.dx_add: the identity shortcut's gradient of a residual block)
    float ste_qmin, ste_qmax;
    int N, C, Hg, Wg, O, HWg;
    int TAPS, TPS, NSTEP, WSB;    // taps, taps per step, steps per chunk, bytes of weights per step
    int TH, NI, PH, PW, W4, TS;   // TS: bytes per term plane
    int tpi, ncit, nchunks, nitems, nunits, w_shift;
    FastDiv fd_w4, fd_ph, fd_ni, fd_th, fd_ncit, fd_tpi;
    QdUnits units;
};

template <int MF, int S, int TAPS, int NT>
__global__ __launch_bounds__(256, 2) void k_qd_dgrad(const QddParams p) {
    HIP_DYNAMIC_SHARED(float, smem)
    unsigned char* wbuf = reinterpret_cast<unsigned char*>(smem);
    unsigned char* patch = wbuf + 2 * p.WSB;
    const int tid = threadIdx.x, lane = tid & 63, wave = mn_uniform(tid >> 6), j = lane & 15, kg = lane >> 4;
    constexpr int TPS = TAPS == 9 ? 3 : 1, NSTEP = TAPS == 9 ? 3 : 1;
    constexpr int UPT = QDD_UPT_OF(MF);          // staging units per thread (a third one only for the 256-pixel tile)
    if ((int)blockIdx.x >= p.nitems) return;
    for (int i = tid; i < (NT * p.TS) / 16; i += 256) *reinterpret_cast<u32x4*>(patch + 16 * i) = u32x4{0u, 0u, 0u, 0u};
    // staging roles: a unit = output channels 4q .. 4q + 3 of the chunk, patch row pr of image img, pixels 4d .. 4d + 3 (order over the threads: QdUnits)
    int u_lds[UPT], u_goff[UPT], u_pi[UPT];
#pragma unroll
    for (int i = 0; i < UPT; ++i) {
        int q, img, pr, d;
        const bool v = qd_unit(p.units, tid + 256 * i, q, img, pr, d);
        u_lds[i] = v ? ((img * p.PH + pr) * p.PW + 4 * d + 1) * QDD_RS + 8 * q : -1;
        u_goff[i] = v ? 4 * q * p.HWg + 4 * d : 0;
        u_pi[i] = pr | (img << 8);
    }
    float4 preg[UPT][4];
    float psc[UPT][4];
    int u_q[UPT];
#pragma unroll
    for (int i = 0; i < UPT; ++i) {
        int q, img, pr, d;
        qd_unit(p.units, tid + 256 * i, q, img, pr, d);
        u_q[i] = q;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) psc[i][cc] = 1.f;
    }
    uint32_t pok = 0u;
    auto tile_origin = [&](int item, int& n0, int& oh0, int& cit) {
        const uint32_t tile = fd_div((uint32_t)item, p.fd_ncit);
        cit = item - (int)tile * p.ncit;
        if (p.NI == 1) { const uint32_t n = fd_div(tile, p.fd_tpi); n0 = (int)n; oh0 = ((int)tile - (int)n * p.tpi) * p.TH; }
        else { n0 = (int)tile * p.NI; oh0 = 0; }
    };
    auto fetch_patch = [&](int item, int chunk) {
        int n0, oh0, cit;
        tile_origin(item, n0, oh0, cit);
        pok = 0u;
#pragma unroll
        for (int i = 0; i < UPT; ++i) {
            int n = n0 + (u_pi[i] >> 8), oh = oh0 - 1 + (u_pi[i] & 255);
            const bool ok = u_lds[i] >= 0 && n < p.N && oh >= 0 && oh < p.Hg;
            n = n < p.N ? n : p.N - 1;
            oh = oh < 0 ? 0 : (oh < p.Hg ? oh : p.Hg - 1);
            const uint32_t base = (uint32_t)((n * p.O + chunk * 32) * p.Hg + oh) * (uint32_t)p.Wg + (uint32_t)u_goff[i];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) preg[i][cc] = *reinterpret_cast<const float4*>(p.gy + base + (uint32_t)(cc * p.HWg));
            if (p.wsc && u_lds[i] >= 0) {
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) psc[i][cc] = p.wsc[(int64_t)(chunk * 32 + 4 * u_q[i] + cc) * p.wsc_stride];
            }
            pok |= (ok ? 1u : 0u) << i;
        }
    };
    auto commit_patch = [&]() {
#pragma unroll
        for (int i = 0; i < UPT; ++i) {
            if (u_lds[i] < 0) continue;
            const bool ok = (pok >> i) & 1u;
            if (NT == 2) {                                             // round-to-nearest hi + round-to-nearest remainder (qd_terms)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v[4];
#pragma unroll
                    for (int cc = 0; cc < 4; ++cc) {
                        const float raw = e == 0 ? preg[i][cc].x : e == 1 ? preg[i][cc].y : e == 2 ? preg[i][cc].z : preg[i][cc].w;
                        v[cc] = ok ? (p.wsc ? raw * psc[i][cc] : raw) : 0.f;
                    }
                    unsigned h0, l0, h1, l1;
                    mn_split2_bf16x2(v[0], v[1], h0, l0);
                    mn_split2_bf16x2(v[2], v[3], h1, l1);
                    unsigned char* d = patch + u_lds[i] + e * QDD_RS;
                    *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
                    *reinterpret_cast<u32x2*>(d + p.TS) = u32x2{l0, l1};
                }
                continue;
            }
            float t0[4][4], t1[4][4], t2[4][4];                    // [channel][pixel]
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const float v[4] = {preg[i][cc].x, preg[i][cc].y, preg[i][cc].z, preg[i][cc].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float vv = ok ? (p.wsc ? v[e] * psc[i][cc] : v[e]) : 0.f;
                    t0[cc][e] = mn_bf16_head(vv);
                    const float r1 = vv - t0[cc][e];
                    t1[cc][e] = mn_bf16_head(r1);
                    t2[cc][e] = r1 - t1[cc][e];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned char* d = patch + u_lds[i] + e * QDD_RS;
                *reinterpret_cast<u32x2*>(d) = u32x2{mn_pack_bf16x2(t0[0][e], t0[1][e]), mn_pack_bf16x2(t0[2][e], t0[3][e])};
                *reinterpret_cast<u32x2*>(d + p.TS) = u32x2{mn_pack_bf16x2(t1[0][e], t1[1][e]), mn_pack_bf16x2(t1[2][e], t1[3][e])};
                *reinterpret_cast<u32x2*>(d + 2 * p.TS) = u32x2{mn_pack_bf16x2(t2[0][e], t2[1][e]), mn_pack_bf16x2(t2[2][e], t2[3][e])};
            }
        }
    };
    u32x4 wreg[TPS];
    struct Pos { int item, chunk, step; };
    auto fetch_w = [&](const Pos& q) {
        const uint32_t tile = fd_div((uint32_t)q.item, p.fd_ncit);
        const int cit = q.item - (int)tile * p.ncit;
        const unsigned char* src = reinterpret_cast<const unsigned char*>(p.wpk) + (int64_t)((cit * p.nchunks + q.chunk) * NSTEP + q.step) * p.WSB;
#pragma unroll
        for (int t = 0; t < TPS; ++t) wreg[t] = *reinterpret_cast<const u32x4*>(src + t * 4096 + tid * 16);
    };
    auto commit_w = [&](int buf) {
#pragma unroll
        for (int t = 0; t < TPS; ++t) *reinterpret_cast<u32x4*>(wbuf + buf * p.WSB + t * 4096 + tid * 16) = wreg[t];
    };
    const int stride_items = (int)gridDim.x;
    auto advance = [&](Pos& q) {
        if (++q.step == NSTEP) { q.step = 0; if (++q.chunk == p.nchunks) { q.chunk = 0; q.item += stride_items; } }
    };
    // S = 1: the corner of the pixel's 3 x 3 window;  S = 2: the cell's own slot
    int abase[MF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
        const int tp = wave * 16 * MF + mf * 16 + j;
        const int ow = tp & (p.Wg - 1), t = tp >> p.w_shift;
        const uint32_t img = fd_div((uint32_t)t, p.fd_th);
        const int ohl = t - (int)img * p.TH;
        abase[mf] = (((int)img * p.PH + ohl + (S == 2 ? 1 : 0)) * p.PW + ow + (S == 2 ? 1 : 0)) * QDD_RS + kg * 16;
    }
    constexpr int NACC = S == 2 ? 4 : 1;
    Pos nxt{(int)blockIdx.x, 0, 0};
    fetch_patch(nxt.item, 0);
    fetch_w(nxt);
    __syncthreads();
    commit_patch();
    commit_w(0);
    advance(nxt);
    if (nxt.item < p.nitems) fetch_w(nxt);
    Pos la = nxt;
    advance(la);
    __syncthreads();
    int buf = 0;
    for (int item = (int)blockIdx.x; item < p.nitems; item += stride_items) {
        f32x4 acc[NACC][MF][4];
#pragma unroll
        for (int a_ = 0; a_ < NACC; ++a_)
#pragma unroll
            for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                for (int nf = 0; nf < 4; ++nf) acc[a_][mf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int chunk = 0; chunk < p.nchunks; ++chunk) {
            int nitem = item, nchunk = chunk + 1;
            if (nchunk == p.nchunks) { nchunk = 0; nitem += stride_items; }
            const bool have_next = nitem < p.nitems;
            if (have_next) fetch_patch(nitem, nchunk);
#pragma unroll
            for (int step = 0; step < NSTEP; ++step) {             // kernel row r = step (1 x 1: the single tap)
                if (nxt.item < p.nitems) commit_w(buf ^ 1);
                if (la.item < p.nitems) fetch_w(la);
                advance(nxt);
                advance(la);
                const unsigned char* wb = wbuf + buf * p.WSB + lane * 16;
#pragma unroll
                for (int tis = 0; tis < TPS; ++tis) {               // kernel column s = tis
                    const int toff = S == 1 ? ((2 - step) * p.PW + (2 - tis)) * QDD_RS : (TAPS == 9 ? ((step == 0 ? 1 : 0) * p.PW + (tis == 0 ? 1 : 0)) * QDD_RS : 0);
                    const int ai = (S == 2 && TAPS == 9) ? (step != 1 ? 2 : 0) + (tis != 1 ? 1 : 0) : 0;
                    u32x4 b[4];
#pragma unroll
                    for (int nf = 0; nf < 4; ++nf) b[nf] = *reinterpret_cast<const u32x4*>(wb + (tis * 4 + nf) * 1024);
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        u32x4 a[MF];
#pragma unroll
                        for (int mf = 0; mf < MF; ++mf) a[mf] = *reinterpret_cast<const u32x4*>(patch + t * p.TS + abase[mf] + toff);
#pragma unroll
                        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
                            for (int nf = 0; nf < 4; ++nf) {
                                if (NACC == 1) acc[0][mf][nf] = mn_mfma_bf16(a[mf], b[nf], acc[0][mf][nf]);
                                else {
#pragma unroll
                                    for (int a_ = 0; a_ < NACC; ++a_) if (a_ == ai) acc[a_][mf][nf] = mn_mfma_bf16(a[mf], b[nf], acc[a_][mf][nf]);
                                }
                            }
                    }
                }
                __syncthreads();
                buf ^= 1;
            }
            if (have_next) { commit_patch(); __syncthreads(); }
        }
        int n0, oh0, cit;
        tile_origin(item, n0, oh0, cit);
        float s_sc = 1.f, s_zp = 0.f, s_lo = 0.f, s_hi = 0.f;
        if (p.ste_x || p.ste_mask) { s_sc = p.ste_qp[0]; s_zp = p.ste_qp[1]; s_lo = p.ste_qp[2]; s_hi = p.ste_qp[3]; }
        const float s_inv = 1.0f / s_sc;
        auto ste4m = [&](float4 v, uint32_t m) {                   // iao_fq_grad_m with the two conditions read from the forward's bits
            return make_float4((m & 1u) ? mn_div_m(v.x * s_sc, s_sc, s_inv) : 0.f, (m & 2u) ? mn_div_m(v.y * s_sc, s_sc, s_inv) : 0.f,
                               (m & 4u) ? mn_div_m(v.z * s_sc, s_sc, s_inv) : 0.f, (m & 8u) ? mn_div_m(v.w * s_sc, s_sc, s_inv) : 0.f);
        };
        auto ste4 = [&](float4 v, const float* xp) {
            const float4 xv = *reinterpret_cast<const float4*>(xp);
            return make_float4(iao_fq_grad_m(v.x, xv.x, s_sc, s_inv, s_zp, s_lo, s_hi, p.ste_qmin, p.ste_qmax), iao_fq_grad_m(v.y, xv.y, s_sc, s_inv, s_zp, s_lo, s_hi, p.ste_qmin, p.ste_qmax),
                               iao_fq_grad_m(v.z, xv.z, s_sc, s_inv, s_zp, s_lo, s_hi, p.ste_qmin, p.ste_qmax), iao_fq_grad_m(v.w, xv.w, s_sc, s_inv, s_zp, s_lo, s_hi, p.ste_qmin, p.ste_qmax));
        };
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
            const int tp = wave * 16 * MF + mf * 16 + 4 * kg;
            const int ow = tp & (p.Wg - 1), t = tp >> p.w_shift;
            const uint32_t img = fd_div((uint32_t)t, p.fd_th);
            const int ohl = t - (int)img * p.TH;
            const int n = n0 + (int)img;
            if (n >= p.N) continue;
            if (S == 1) {
                const uint32_t base = (uint32_t)((n * p.C + cit * 64 + j) * p.Hg + oh0 + ohl) * (uint32_t)p.Wg + (uint32_t)ow;
#pragma unroll
                for (int nf = 0; nf < 4; ++nf) {
                    float4 v = make_float4(acc[0][mf][nf][0] * p.wscale, acc[0][mf][nf][1] * p.wscale, acc[0][mf][nf][2] * p.wscale, acc[0][mf][nf][3] * p.wscale);
                    if (p.ste_mask) { const uint32_t e0 = base + (uint32_t)(nf * 16 * p.HWg); v = ste4m(v, (uint32_t)p.ste_mask[e0 >> 3] >> (e0 & 4u)); }
                    else if (p.ste_x) v = ste4(v, p.ste_x + base + (uint32_t)(nf * 16 * p.HWg));
                    if (p.dx_add) { const float4 a = *reinterpret_cast<const float4*>(p.dx_add + base + (uint32_t)(nf * 16 * p.HWg)); v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
                    *reinterpret_cast<float4*>(p.dx + base + (uint32_t)(nf * 16 * p.HWg)) = v;
                }
            } else {
                const int WX = 2 * p.Wg;
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) {
                    const uint32_t base = (uint32_t)((n * p.C + cit * 64 + j) * 2 * p.Hg + 2 * (oh0 + ohl) + ph) * (uint32_t)WX + (uint32_t)(2 * ow);
#pragma unroll
                    for (int nf = 0; nf < 4; ++nf) {
                        const f32x4 e = acc[(NACC == 4 ? 2 * ph : 0)][mf][nf], o = acc[(NACC == 4 ? 2 * ph + 1 : 0)][mf][nf];
                        const uint32_t off = base + (uint32_t)(nf * 16 * 4 * p.HWg);
                        float* dst = p.dx + off;
                        float4 v0 = make_float4(e[0] * p.wscale, o[0] * p.wscale, e[1] * p.wscale, o[1] * p.wscale);
                        float4 v1 = make_float4(e[2] * p.wscale, o[2] * p.wscale, e[3] * p.wscale, o[3] * p.wscale);
                        if (p.ste_mask) { const uint32_t mb = p.ste_mask[off >> 3]; v0 = ste4m(v0, mb); v1 = ste4m(v1, mb >> 4); }
                        else if (p.ste_x) { v0 = ste4(v0, p.ste_x + off); v1 = ste4(v1, p.ste_x + off + 4); }
                        if (p.dx_add) {
                            const float4 a0 = *reinterpret_cast<const float4*>(p.dx_add + off), a1 = *reinterpret_cast<const float4*>(p.dx_add + off + 4);
                            v0.x += a0.x; v0.y += a0.y; v0.z += a0.z; v0.w += a0.w; v1.x += a1.x; v1.y += a1.y; v1.z += a1.z; v1.w += a1.w;
                        }
                        *reinterpret_cast<float4*>(dst) = v0;
                        *reinterpret_cast<float4*>(dst + 4) = v1;
                    }
                }
            }
        }
    }
}

struct QddPlan { QddParams p; int MF, S, grid; size_t lds; int64_t ws_bytes; };
static int plan_qdd(const mn_conv_geom* g, QddPlan* pl) {
    if (!qd_geom_ok(g)) return 0;
    QddParams& p = pl->p;
    const int S = g->stride_h;
    pl->S = S;
    p.N = g->N; p.C = g->C; p.O = g->O; p.Hg = g->H / S; p.Wg = g->W / S; p.HWg = p.Hg * p.Wg;
    p.TAPS = g->KH * g->KW; p.TPS = p.TAPS == 9 ? 3 : 1; p.NSTEP = p.TAPS == 9 ? 3 : 1; p.WSB = p.TPS * 4096;
    p.w_shift = qd_log2(p.Wg);
    if (p.w_shift < 2 || p.Wg > 32 || p.HWg % 8) return 0;
    if ((int64_t)g->N * g->C * g->H * g->W >= ((int64_t)1 << 31) || (int64_t)g->N * g->O * p.HWg >= ((int64_t)1 << 31)) return 0;
    int MF = 0;
    const int NTm = qd_terms();
    int mf0 = S == 2 ? 1 : 4;          // MF 4 (round 5): 256-pixel tiles -- an item's fixed costs (first patch, epilogue stores: ~45 % of an MF 2 item, s_memtime timeline) over twice the work
    for (int mf = mf0; mf >= 1; mf >>= 1) {
        const int BM = 64 * mf;
        int TH, NI;
        if (p.HWg >= BM) { if (BM % p.Wg) continue; TH = BM / p.Wg; if (p.Hg % TH) continue; NI = 1; }
        else { if (BM % p.HWg) continue; NI = BM / p.HWg; TH = p.Hg; }
        const int PH = TH + 2, PW = p.Wg + 2;
        if (PH > 255 || NI > 255) continue;
        const int64_t plane = ((int64_t)NI * PH * PW * QDD_RS + 255) / 256 * 256;
        const QdUnits units = qd_make_units(p.Wg / 4, PH, NI, 8);
        if (NTm * plane > 56 * 1024 || units.nunits > 256 * QDD_UPT_OF(mf)) continue;          // two blocks per CU: 2 x (term planes + 24 KB of weights) in 160 KB
        {   // ... and two blocks per CU there must be: a larger tile that leaves fewer than 512 items loses the overlap of the two blocks' phases (512 x 512 @ 4 x 4: MF 2 75 us, MF 1 61 us)
            const int64_t nt = NI == 1 ? (int64_t)g->N * (p.Hg / TH) : ((int64_t)g->N + NI - 1) / NI;
            if (mf > 1 && nt * (g->C / 64) < 512) continue;
        }
        MF = mf; p.TH = TH; p.NI = NI; p.PH = PH; p.PW = PW; p.TS = (int)plane; p.nunits = units.nunits; p.units = units;
        break;
    }
    if (!MF) return 0;
    pl->MF = MF;
    p.W4 = p.Wg / 4;
    p.tpi = p.NI == 1 ? p.Hg / p.TH : 1;
    const int ntiles = p.NI == 1 ? g->N * p.tpi : (g->N + p.NI - 1) / p.NI;
    p.ncit = g->C / 64; p.nchunks = g->O / 32; p.nitems = ntiles * p.ncit;
    p.fd_w4 = make_fastdiv((uint32_t)p.W4); p.fd_ph = make_fastdiv((uint32_t)p.PH); p.fd_ni = make_fastdiv((uint32_t)p.NI);
    p.fd_th = make_fastdiv((uint32_t)p.TH); p.fd_ncit = make_fastdiv((uint32_t)p.ncit); p.fd_tpi = make_fastdiv((uint32_t)p.tpi);
    int tgt = 512;
    pl->grid = p.nitems < tgt ? p.nitems : tgt;
    pl->lds = (size_t)2 * p.WSB + (size_t)NTm * p.TS;
    pl->ws_bytes = ((int64_t)g->O * g->C * p.TAPS * 2 + 255) / 256 * 256;
    return 1;
}
static void qd_launch_dgrad(const QddPlan& pl, hipStream_t s) {
    const QddParams& p = pl.p;
    const int NT = qd_terms();
#define QDD_LAUNCH(M_, S_, T_, N_) do { raise_lds_limit((const void*)k_qd_dgrad<M_, S_, T_, N_>, pl.lds); hipLaunchKernelGGL((k_qd_dgrad<M_, S_, T_, N_>), dim3(pl.grid), dim3(256), pl.lds, s, p); } while (0)
    if (pl.S == 2 && p.TAPS == 9) { if (NT == 2) QDD_LAUNCH(1, 2, 9, 2); else QDD_LAUNCH(1, 2, 9, 3); }
    else if (pl.S == 2) { if (NT == 2) QDD_LAUNCH(1, 2, 1, 2); else QDD_LAUNCH(1, 2, 1, 3); }
    else if (pl.MF == 4) { if (NT == 2) QDD_LAUNCH(4, 1, 9, 2); else QDD_LAUNCH(4, 1, 9, 3); }
    else if (pl.MF == 2) { if (NT == 2) QDD_LAUNCH(2, 1, 9, 2); else QDD_LAUNCH(2, 1, 9, 3); }
    else { if (NT == 2) QDD_LAUNCH(1, 1, 9, 2); else QDD_LAUNCH(1, 1, 9, 3); }
#undef QDD_LAUNCH
}
int qd_dgrad_supported(const mn_conv_geom* g, const mn_wq* wq) {
    QddPlan pl;
    return wq && wq->mode == MN_WQ_DOREFA && wq->bits >= 2 && wq->bits <= 8 && plan_qdd(g, &pl);
}
int qd_dgrad_native(const mn_conv_geom* g, const mn_wq* wq) { return qd_dgrad_supported(g, wq); }
int64_t qd_dgrad_ws_bytes(const mn_conv_geom* g) { QddPlan pl; return plan_qdd(g, &pl) ? pl.ws_bytes : 0; }
int qd_bwd_data(const mn_conv_geom* g, const mn_wq* wq, const float* gy, const float* w, float* dx, void* ws, int64_t ws_bytes, hipStream_t s) {
    QddPlan pl;
    if (!qd_dgrad_supported(g, wq) || !plan_qdd(g, &pl) || !aligned16(gy) || !aligned16(dx) || !w) MN_FAIL(MN_ENOTSUP, "mn_conv2d_bwd_data(dense): geometry not covered");
    if (!ws || ws_bytes < pl.ws_bytes || !aligned16(ws)) MN_FAIL(MN_ENOSPC, "mn_conv2d_bwd_data(dense): workspace too small");
    QddParams& p = pl.p;
    const uint16_t* wpk = reinterpret_cast<const uint16_t*>(wq->packed_bwd);
    if (!wpk) { qd_launch_pack(w, reinterpret_cast<uint16_t*>(ws), g->O, g->C, p.TAPS, wq->bits, 1, s); wpk = reinterpret_cast<const uint16_t*>(ws); }
    p.gy = gy; p.wpk = wpk; p.dx = dx; p.wscale = 1.0f / (float)((1ll << wq->bits) - 1); p.wsc = nullptr; p.wsc_stride = 0; p.ste_x = nullptr; p.ste_qp = nullptr; p.ste_mask = nullptr; p.dx_add = nullptr;
    mn_set_last_kernel("k_qd_dgrad<%d, %d, %d, %d>", pl.MF, pl.S, p.TAPS, qd_terms());
    { const double nx = (double)g->N * g->C * g->H * g->W, ny = (double)g->N * g->O * p.HWg; mn_prof_bytes(4.0 * ny * p.ncit + 4.0 * nx); mn_prof_flops(2.0 * ny * g->C * p.TAPS); }
    mn_prof_begin(s);
    qd_launch_dgrad(pl, s);
    mn_prof_end(s);
    MN_CHECK_LAUNCH("mn_conv2d_bwd_data(dense)");
    return MN_OK;
}

// ================================================================================================ backward-weight
//   dw[o][c][r][s] = s_a * sum over (n, oh, ow) of gy[n][o][oh][ow] * j[n][c][oh S + r - P][ow S + s - P]
// M = 64 output channels, N = taps x 64 input channels, K = gy-domain pixels (contiguous in NCHW for both operands: no transposition anywhere).
// A block (12 waves, one per CU) owns one (64 o, 64 c) pair and a range of pixel tiles (split-K).  Round 5: WAVE-SPECIALISED -- knock-out builds of the
// round-4 kernel (8 waves that all staged, read fragments and contracted in the same phase between two barriers) showed its time to be the SUM of its parts
// (fixed 22 us + matrix 16 + gy path 10 + patch 7 + A reads 7 + B reads 7 = 60 us on every resnet18 layer), not their maximum:
//   * waves 0-3 PRODUCERS: gy rows of 2 - 4 K-steps (32 pixels each) in flight in registers, split into NT bf16 term planes [o][32 px] (80-byte rows, two LDS
//     buffers); the input patch of the NEXT tile (64 channels, bf16, rows 8-byte aligned, zero frame) into the second patch buffer while the consumers work on
//     the current one; one barrier per K-step;
//   * waves 4-11 CONSUMERS (fragment reads + MFMA only): wave w contracts input-channel fragment (w & 3) against output-channel fragments 2 (w >> 2), + 1 for
//     all taps: 18 accumulator tiles; software-pipelined by kernel ROW: the B words of row r + 1 (or, behind the step's barrier, the A fragments and row 0 of
//     the next step) are read while row r is contracted.
// The B fragment of a tap = 2 x 4 consecutive gy-domain pixels shifted by the tap:
//   S = 1: patch [c][image][row][4 + W + 4]; per kernel row one aligned 8-byte read + its two neighbour dwords per half fragment, the three
//          column shifts by v_alignbyte (no shifted copies in LDS, no im2col);
//   S = 2: the patch rows are split by column parity, [c][image][row][even | odd][4 + W/2 + 4]: tap column 1 reads the even plane, column 2 the
//          odd plane, column 0 the odd plane one pixel to the left; 1 x 1 / stride 2: the even plane of the even rows only.
// Partial tiles [z][pair][tap][c][o] (o innermost: a lane's four accumulator rows are one 16-byte store) are summed in fp64 in fixed order by
// k_qd_wgrad_reduce (deterministic).
#define QDW_PUPT 20            // patch dwords a producer thread stages per tile
#define QDW_PCH 4              // ... in this many chunks
#ifndef QDW_PRIO
#define QDW_PRIO 2
#endif
#define QDW_DYP 5120           // bytes per gy term plane (64 rows x 80)
struct QdwParams {
    const float* gy;              // [N][O][Hg][Wg]
    const unsigned char* x;       // [N][C][S Hg][S Wg] codes
    float* part;
    int N, C, Hg, Wg, O, HWg, HX, WX, PAD;
    int TH, NI, PH, PWp, RB, CS, W4, BMt, nks;    // tile rows / images, patch rows, padded plane row (bf16 elements), bytes per patch row, bytes per channel, K-steps per tile
    int tpi, ntiles, tpz, Z, ncit, npairs, nunits, w_shift;
    FastDiv fd_w4, fd_ph, fd_ni, fd_th, fd_tpi, fd_np, fd_nks;
    int xsgn;                     // x holds signed codes (IAO)
    int pdb;                      // 1: two patch buffers (the next tile's patch is staged while the current one is contracted); 0: one (shapes whose patch exceeds 64 KB)
};

template <int S, int TAPS, int NT>
__global__ __launch_bounds__(768) void k_qd_wgrad(const QdwParams p) {
    HIP_DYNAMIC_SHARED(float, smem)
    unsigned char* dyb = reinterpret_cast<unsigned char*>(smem);          // [2][3][64][80]
    unsigned char* xp0 = dyb + 2 * 3 * QDW_DYP;                           // [2][64 CS]
    const int XPB = 64 * p.CS;
    const int tid = threadIdx.x, lane = tid & 63, wave = mn_uniform(tid >> 6), j = lane & 15, kg = lane >> 4;
    constexpr int KR = TAPS == 9 ? 3 : 1;
    const uint32_t z = fd_div(blockIdx.x, p.fd_np);
    const int pair = (int)blockIdx.x - (int)z * p.npairs;
    const int cot = pair / p.ncit, cit = pair - cot * p.ncit;
    const int t_begin = (int)z * p.tpz, t_end = (t_begin + p.tpz) < p.ntiles ? (t_begin + p.tpz) : p.ntiles;
    const int n = t_begin < t_end ? (t_end - t_begin) * p.nks : 0;         // K-steps of this block; step k = (tile t_begin + k / nks, ks = k % nks)
    auto tile_origin = [&](int tile, int& n0, int& oh0) {
        if (p.NI == 1) { const uint32_t nn = fd_div((uint32_t)tile, p.fd_tpi); n0 = (int)nn; oh0 = (tile - (int)nn * p.tpi) * p.TH; }
        else { n0 = tile * p.NI; oh0 = 0; }
    };
    if (wave < 4) {
        // ------------------------------------------------------------------------------------------------ producers (256 threads)
        // patch staging roles: thread (channel pc = tid >> 2, quarter pq = tid & 3) stages positions pos = pq + 4 i of its channel, pos = (img * PH + pr) * W4 + d: input
        // pixels 4d .. 4d + 3 of patch row pr.  Two words per unit, everything else is per thread or per tile: ug = the unit's global offset inside the tile's window
        // (img C HX WX + pr WX + 4 d), ul = LDS byte offset inside the channel | pr << 16 | img << 24 (0xffffffff: the thread has no such unit -- it loads its channel's
        // first word and writes zeros into a dump slot behind the patch buffers: no per-unit branches).
        MN_SETPRIO(QDW_PRIO);                                              // the staging waves win the VALU arbitration against the two MFMA waves of their SIMD
        const int pc = tid >> 2, pq = tid & 3;
        const int npos = p.NI * p.PH * p.W4;
        uint32_t ul[QDW_PUPT];
        int ug[QDW_PUPT];
#pragma unroll
        for (int i = 0; i < QDW_PUPT; ++i) {
            const int pos = pq + 4 * i;
            const uint32_t t0 = fd_div((uint32_t)pos, p.fd_w4);
            const int d = pos - (int)t0 * p.W4;
            const uint32_t img = fd_div(t0, p.fd_ph);
            const int pr = (int)t0 - (int)img * p.PH;
            // S = 1: pixels 4d.. at elements 4 + 4d ..;  S = 2: the even pixels (4d, 4d + 2) at even-plane elements 4 + 2d, the odd ones at the odd plane's
            const uint32_t lo = (uint32_t)(((int)img * p.PH + pr) * p.RB + (4 + (S == 2 ? 2 : 4) * d) * 2);
            ul[i] = pos < npos ? (lo | ((uint32_t)pr << 16) | (img << 24)) : 0xffffffffu;
            ug[i] = pos < npos ? ((int)img * p.C * p.HX + pr) * p.WX + 4 * d : 0;
        }
        const int pc_g = pc * p.HX * p.WX;
        const uint32_t pc_l = (uint32_t)(pc * p.CS);
        const uint32_t dump = (uint32_t)((1 + p.pdb) * XPB);          // 16 bytes behind the patch buffers (relative to xp0)
        // The units are staged in QDW_PCH chunks: chunk c of tile t + 1's patch is committed (into the other patch buffer) and chunk c of tile t + 2's patch fetched
        // (into the same registers) during one of tile t's step pairs -- the loads have a whole tile to land and no tile boundary carries the whole patch
        // (measured with s_memtime: 6 500 cycles per boundary when it did, 16 000 when its registers spilled).
        uint32_t preg[QDW_PUPT];
        uint32_t pok = 0u;
        const int nu = (npos + 3) >> 2;                                  // units per thread that exist (uniform)
        auto fetch_chunk = [&](int c, int tile) {
            int n0, oh0;
            tile_origin(tile, n0, oh0);
            const int ihb = oh0 * S - p.PAD;
            const int base = ((n0 * p.C + cit * 64) * p.HX + ihb) * p.WX + pc_g;          // (may point above the image for the halo rows: those units load the channel's first word)
#pragma unroll
            for (int i = c * (QDW_PUPT / QDW_PCH); i < (c + 1) * (QDW_PUPT / QDW_PCH); ++i) {
                if (i >= nu) break;                                   // (uniform)
                const uint32_t w = mn_opaque(ul[i]);                   // (opaque: nothing derived from a unit's words is hoisted out of the step loop)
                const int ih = ihb + (int)((w >> 16) & 255u), nn = n0 + (int)(w >> 24);
                const bool ok = w != 0xffffffffu && (unsigned)ih < (unsigned)p.HX && nn < p.N;
                const int off = ok ? base + (int)mn_opaque((uint32_t)ug[i]) : pc_g;
                preg[i] = *reinterpret_cast<const uint32_t*>(p.x + (uint32_t)off);
                pok = (pok & ~(1u << i)) | ((ok ? 1u : 0u) << i);
            }
        };
        auto commit_chunk = [&](int c, unsigned char* xp, uint32_t dump_rel) {
#pragma unroll
            for (int i = c * (QDW_PUPT / QDW_PCH); i < (c + 1) * (QDW_PUPT / QDW_PCH); ++i) {
                if (i >= nu) break;                                   // (uniform)
                const uint32_t w = mn_opaque(ul[i]);
                const bool valid = w != 0xffffffffu;
                const uint32_t lo = valid ? pc_l + (w & 0xfffu) : dump_rel;
                uint32_t v = ((pok >> i) & 1u) ? preg[i] : 0u;          // (signed codes: 0 ^ 0x80 - 128 = 0 as well)
                float f0, f1, f2, f3;
                if (p.xsgn) {                                         // (uniform) signed codes: byte ^ 0x80 - 128
                    v ^= 0x80808080u;
                    f0 = (float)(v & 0xffu) - 128.f; f1 = (float)((v >> 8) & 0xffu) - 128.f; f2 = (float)((v >> 16) & 0xffu) - 128.f; f3 = (float)(v >> 24) - 128.f;
                } else { f0 = (float)(v & 0xffu); f1 = (float)((v >> 8) & 0xffu); f2 = (float)((v >> 16) & 0xffu); f3 = (float)(v >> 24); }
                if (S == 1) *reinterpret_cast<u32x2*>(xp + lo) = u32x2{mn_pack_hi16(f0, f1), mn_pack_hi16(f2, f3)};
                else {
                    *reinterpret_cast<uint32_t*>(xp + lo) = mn_pack_hi16(f0, f2);
                    *reinterpret_cast<uint32_t*>(xp + (valid ? lo + p.PWp * 2 : lo + 4)) = mn_pack_hi16(f1, f3);
                }
            }
        };
        // gy staging role: rows sr and sr + 32, float4 sq of the K-step.  A register STAGE holds a pair of K-steps (2 x 2 float4); two stages: 2 - 4 K-steps
        // (16 - 32 KB per CU) in flight.  K-steps per tile are even (planner), so a tile's first step is always the first step of a stage.
        const int sr = tid >> 3, sq = tid & 7;
        struct GStage { float4 v[2][2]; int ok; };                   // [step of the pair][row half]; ok: bit h = step h's image exists
        GStage gs[2];
        // loads are unconditional (a conditional load makes the register set a phi: copies and a vmcnt(0) right behind the issue); past the block's range the
        // block's own last pair is read again (an L2 hit)
        const int nq = n >> 1;
        auto fetch_gy = [&](GStage& G, int q) {
            const int qq = q < nq ? q : nq - 1;
            G.ok = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int kk = 2 * qq + h;
                const uint32_t tl = fd_div((uint32_t)kk, p.fd_nks);
                const int ks = kk - (int)tl * p.nks;
                int n0, oh0;
                tile_origin(t_begin + (int)tl, n0, oh0);
                const int kp = ks * 32 + 4 * sq;
                const int col = kp & (p.Wg - 1), t = kp >> p.w_shift;
                const uint32_t img = fd_div((uint32_t)t, p.fd_th);
                const int ohl = t - (int)img * p.TH;
                int nn = n0 + (int)img;
                const bool ok = nn < p.N;
                G.ok |= (ok ? 1 : 0) << h;
                nn = ok ? nn : p.N - 1;
                const float* src = p.gy + (uint32_t)((nn * p.O + cot * 64 + sr) * p.Hg + oh0 + ohl) * (uint32_t)p.Wg + (uint32_t)col;
                G.v[h][0] = *reinterpret_cast<const float4*>(src);
                G.v[h][1] = *reinterpret_cast<const float4*>(src + (uint32_t)(32 * p.HWg));
            }
        };
        auto commit_gy = [&](const GStage& G, int h) {                 // step h of the pair -> gy buffer h (the step's parity)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float4 g4 = ((G.ok >> h) & 1) ? G.v[h][i] : make_float4(0.f, 0.f, 0.f, 0.f);
                unsigned char* d = dyb + h * 3 * QDW_DYP + (sr + 32 * i) * 80 + sq * 8;
                if (NT == 2) {
                    unsigned h0, l0, h1, l1;
                    mn_split2_bf16x2(g4.x, g4.y, h0, l0);
                    mn_split2_bf16x2(g4.z, g4.w, h1, l1);
                    *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
                    *reinterpret_cast<u32x2*>(d + QDW_DYP) = u32x2{l0, l1};
                } else {
                    const float v[4] = {g4.x, g4.y, g4.z, g4.w};
                    float t0[4], t1[4], t2[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        t0[e] = mn_bf16_head(v[e]);
                        const float r1 = v[e] - t0[e];
                        t1[e] = mn_bf16_head(r1);
                        t2[e] = r1 - t1[e];
                    }
                    *reinterpret_cast<u32x2*>(d) = u32x2{mn_pack_bf16x2(t0[0], t0[1]), mn_pack_bf16x2(t0[2], t0[3])};
                    *reinterpret_cast<u32x2*>(d + QDW_DYP) = u32x2{mn_pack_bf16x2(t1[0], t1[1]), mn_pack_bf16x2(t1[2], t1[3])};
                    *reinterpret_cast<u32x2*>(d + 2 * QDW_DYP) = u32x2{mn_pack_bf16x2(t2[0], t2[1]), mn_pack_bf16x2(t2[2], t2[3])};
                }
            }
        };
        if (n > 0) {
            fetch_gy(gs[0], 0);
            MN_SCHED_FENCE();
            fetch_gy(gs[1], 1);
            if (t_begin + 1 < t_end) {
#pragma unroll
                for (int c = 0; c < QDW_PCH; ++c) fetch_chunk(c, t_begin + 1);
            }
        }
        __syncthreads();                          // barrier Z0: the consumers have zero-filled the patch buffers (the frame stays zero)
        __syncthreads();                          // barrier Z:  ... and staged the FIRST tile's patch (512 idle threads instead of these 256: the prologue was 16 k cycles)
        if (n > 0) {
            int ks = 0, tile = t_begin;
            const int ppt = p.nks >> 1;                                  // step pairs per tile (1, 2 or 4)
            // Barrier k (behind the commit of step k into gy buffer k & 1) releases the consumers' reads of step k; buffer k & 1 is overwritten with step k + 2 behind
            // barrier k + 1, which the consumers pass with every read of step k complete.  Patch: buffer (t + 1) & 1 is free once the consumers passed the barrier of
            // tile t's first step (their last reads of tile t - 1 are complete) and must be complete at the barrier of tile t + 1's first step: its chunks are written
            // in the SECOND half of tile t's step pairs.  One patch buffer (pdb == 0): the whole patch between barrier X and the next tile's first barrier.
            for (int q0 = 0; q0 < nq; q0 += 2) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int q = q0 + u;
                    if (q >= nq) break;                               // (uniform)
                    if (!p.pdb && ks == 0 && q > 0) {
                        __syncthreads();                                // barrier X: the consumers' last reads of the single patch buffer are complete
#pragma unroll
                        for (int c = 0; c < QDW_PCH; ++c) {
                            commit_chunk(c, xp0, dump);
                            if (tile + 1 < t_end) fetch_chunk(c, tile + 1);
                        }
                    }
                    commit_gy(gs[u], 0);
                    __syncthreads();
                    if (p.pdb && tile + 1 < t_end) {
                        const uint32_t po = (uint32_t)((((tile + 1 - t_begin) & 1)) * XPB);
                        const int pit = ks >> 1;                         // this pair's index inside the tile
#pragma unroll
                        for (int c = 0; c < QDW_PCH; ++c) {
                            if (((c * ppt) >> 2) != pit) continue;      // (uniform: chunk c belongs to pair c ppt / 4)
                            commit_chunk(c, xp0 + po, dump - po);
                            if (tile + 2 < t_end) fetch_chunk(c, tile + 2);
                        }
                    }
                    commit_gy(gs[u], 1);
                    fetch_gy(gs[u], q + 2);
                    __syncthreads();
                    ks += 2;
                    if (ks == p.nks) { ks = 0; ++tile; }
                }
            }
        }
    } else {
        // ------------------------------------------------------------------------------------------------ consumers (512 threads)
        const int cwv = wave - 4, cf = cwv & 3, coh = cwv >> 2;
        f32x4 acc[2][TAPS];
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int t = 0; t < TAPS; ++t) acc[c2][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        struct AFrag { u32x4 a[2][NT]; };
        struct BRow { uint32_t w[2][5]; };                                // raw words of one kernel row: [half][..]
        AFrag aa[2];
        BRow bb[2];
        auto load_a = [&](AFrag& A, int buf) {
            const unsigned char* gb = dyb + buf * 3 * QDW_DYP + ((2 * coh) * 16 + j) * 80 + kg * 16;
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int t = 0; t < NT; ++t) A.a[c2][t] = *reinterpret_cast<const u32x4*>(gb + t * QDW_DYP + c2 * 16 * 80);
        };
        // this lane's two half fragments: gy-domain pixels kp .. kp + 3 of the tile (kp = 32 ks + 8 kg + 4 h), kernel row r.  The LDS offset of (step, half) is kept
        // incrementally: a K-step is 32 / Wg rows further down; past the tile image's last row the next image of the tile follows (PH patch rows per image)
        const int rk = 32 >> p.w_shift;                                   // gy rows per K-step (0: several K-steps per row cannot happen, Wg <= 32)
        int boff0[2], bohl0[2], boff[2], bohl[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kp = 8 * kg + 4 * h;
            const int col = kp & (p.Wg - 1), t = kp >> p.w_shift;
            const uint32_t img = fd_div((uint32_t)t, p.fd_th);
            bohl0[h] = t - (int)img * p.TH;
            boff0[h] = (cf * 16 + j) * p.CS + ((int)img * p.PH + bohl0[h] * S) * p.RB + (4 + col) * 2;
            boff[h] = boff0[h]; bohl[h] = bohl0[h];
        }
        auto next_step_b = [&](bool new_tile) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (new_tile) { boff[h] = boff0[h]; bohl[h] = bohl0[h]; continue; }
                int o = bohl[h] + rk, a = boff[h] + rk * S * p.RB;
                while (o >= p.TH) { o -= p.TH; a += (p.PH - p.TH * S) * p.RB; }          // (at most 32 / (Wg TH) images further: a short uniform-trip loop on 4 x 4 images)
                bohl[h] = o; boff[h] = a;
            }
        };
        auto load_b = [&](BRow& B, int par, int r) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned char* q = xp0 + (par & p.pdb) * XPB + boff[h] + r * p.RB;
                if (S == 1) {
                    const u32x2 c = *reinterpret_cast<const u32x2*>(q);
                    B.w[h][0] = *reinterpret_cast<const uint32_t*>(q - 4); B.w[h][1] = c[0]; B.w[h][2] = c[1]; B.w[h][3] = *reinterpret_cast<const uint32_t*>(q + 8);
                } else {
                    const u32x2 e = *reinterpret_cast<const u32x2*>(q);
                    B.w[h][0] = e[0]; B.w[h][1] = e[1];
                    if (TAPS == 9) {
                        const unsigned char* qo = q + p.PWp * 2;
                        const u32x2 o = *reinterpret_cast<const u32x2*>(qo);
                        B.w[h][2] = *reinterpret_cast<const uint32_t*>(qo - 4); B.w[h][3] = o[0]; B.w[h][4] = o[1];
                    }
                }
            }
        };
        auto mma = [&](const AFrag& A, const BRow& B, int r) {
            uint32_t lo[3][2], hi[3][2];                                   // [s][half]
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (S == 1) {
                    const uint32_t pv = B.w[h][0], c0 = B.w[h][1], c1 = B.w[h][2], nx = B.w[h][3];
                    lo[0][h] = mn_alignbyte(c0, pv, 2); hi[0][h] = mn_alignbyte(c1, c0, 2);
                    lo[1][h] = c0; hi[1][h] = c1;
                    lo[2][h] = mn_alignbyte(c1, c0, 2); hi[2][h] = mn_alignbyte(nx, c1, 2);
                } else {
                    lo[1][h] = B.w[h][0]; hi[1][h] = B.w[h][1];
                    if (TAPS == 9) {
                        const uint32_t pv = B.w[h][2], o0 = B.w[h][3], o1 = B.w[h][4];
                        lo[0][h] = mn_alignbyte(o0, pv, 2); hi[0][h] = mn_alignbyte(o1, o0, 2);
                        lo[2][h] = o0; hi[2][h] = o1;
                    }
                }
            }
            u32x4 b[3];
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_) b[s_] = u32x4{lo[s_][0], hi[s_][0], lo[s_][1], hi[s_][1]};
            // term-outer over the row's taps: MFMAs on the same accumulator are 6 instructions apart (2 with the tap outermost: dependency stalls)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int s_ = 0; s_ < 3; ++s_) {
                    if (TAPS == 1 && s_ != 1) continue;
                    const int ti = TAPS == 9 ? r * 3 + s_ : 0;
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2) acc[c2][ti] = mn_mfma_bf16(A.a[c2][t], b[s_], acc[c2][ti]);
                }
        };
        // the first tile's patch: thread (channel ctid >> 3, lane ctid & 7) stages positions pos = lane + 8 i of its channel (the producers' unit, see above)
        const int ctid = tid - 256, sc_ = ctid >> 3, sl_ = ctid & 7;
        const int npos_c = p.NI * p.PH * p.W4;
        uint32_t sreg[10];
        uint32_t sok = 0u;
        if (n > 0) {
            int n0, oh0;
            tile_origin(t_begin, n0, oh0);
            const int ihb = oh0 * S - p.PAD;
            const int cb_g = sc_ * p.HX * p.WX;
            const int base = ((n0 * p.C + cit * 64) * p.HX + ihb) * p.WX + cb_g;
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const int pos = sl_ + 8 * i;
                if (8 * i >= npos_c) break;                          // (uniform)
                const uint32_t t0 = fd_div((uint32_t)pos, p.fd_w4);
                const int d = pos - (int)t0 * p.W4;
                const uint32_t img = fd_div(t0, p.fd_ph);
                const int pr = (int)t0 - (int)img * p.PH;
                const int ih = ihb + pr, nn = n0 + (int)img;
                const bool ok = pos < npos_c && (unsigned)ih < (unsigned)p.HX && nn < p.N;
                const int off = ok ? base + ((int)img * p.C * p.HX + pr) * p.WX + 4 * d : cb_g;
                sreg[i] = *reinterpret_cast<const uint32_t*>(p.x + (uint32_t)off);
                sok |= (ok ? 1u : 0u) << i;
            }
        }
        for (int i = ctid; i < ((1 + p.pdb) * XPB) / 8; i += 512) *reinterpret_cast<u32x2*>(xp0 + 8 * i) = u32x2{0u, 0u};          // the zero frame of both patch buffers
        __syncthreads();                          // barrier Z0
        if (n > 0) {
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const int pos = sl_ + 8 * i;
                if (8 * i >= npos_c) break;                          // (uniform)
                if (pos >= npos_c) continue;
                const uint32_t t0 = fd_div((uint32_t)pos, p.fd_w4);
                const int d = pos - (int)t0 * p.W4;
                const uint32_t img = fd_div(t0, p.fd_ph);
                const int pr = (int)t0 - (int)img * p.PH;
                const uint32_t lo = (uint32_t)(sc_ * p.CS + ((int)img * p.PH + pr) * p.RB + (4 + (S == 2 ? 2 : 4) * d) * 2);
                uint32_t v = ((sok >> i) & 1u) ? sreg[i] : 0u;
                float f0, f1, f2, f3;
                if (p.xsgn) {
                    v ^= 0x80808080u;
                    f0 = (float)(v & 0xffu) - 128.f; f1 = (float)((v >> 8) & 0xffu) - 128.f; f2 = (float)((v >> 16) & 0xffu) - 128.f; f3 = (float)(v >> 24) - 128.f;
                } else { f0 = (float)(v & 0xffu); f1 = (float)((v >> 8) & 0xffu); f2 = (float)((v >> 16) & 0xffu); f3 = (float)(v >> 24); }
                if (S == 1) *reinterpret_cast<u32x2*>(xp0 + lo) = u32x2{mn_pack_hi16(f0, f1), mn_pack_hi16(f2, f3)};
                else {
                    *reinterpret_cast<uint32_t*>(xp0 + lo) = mn_pack_hi16(f0, f2);
                    *reinterpret_cast<uint32_t*>(xp0 + lo + p.PWp * 2) = mn_pack_hi16(f1, f3);
                }
            }
        }
        __syncthreads();                          // barrier Z
        if (n > 0) {
            int ks = 0, par = 0;
            __syncthreads();                      // barrier 0
            load_a(aa[0], 0);
            load_b(bb[0], 0, 0);
            for (int k0 = 0; k0 < n; k0 += 2) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int k = k0 + kk;
                    if (k >= n) break;                                // (uniform)
#pragma unroll
                    for (int r = 0; r < KR; ++r) {
                        const int ph = kk * KR + r;                    // (2 KR phases per unrolled pair of steps: the word sets alternate consistently)
                        const bool need_x = !p.pdb && ks == p.nks - 1 && k + 1 < n;          // single patch buffer: barrier X behind the tile's last patch reads
                        if (r + 1 < KR) {
                            load_b(bb[(ph + 1) & 1], par, r + 1);
                            if (r + 2 == KR && need_x) __syncthreads();
                        } else if (k + 1 < n) {
                            if (KR == 1 && need_x) __syncthreads();
                            __syncthreads();                          // barrier k + 1
                            if (++ks == p.nks) { ks = 0; par ^= 1; next_step_b(true); } else next_step_b(false);
                            load_a(aa[(kk + 1) & 1], (k + 1) & 1);
                            load_b(bb[(ph + 1) & 1], par, 0);
                        }
                        MN_SCHED_FENCE();
                        mma(aa[kk & 1], bb[ph & 1], r);
                    }
                }
            }
        }
        // D[row = o: 4 kg + r][col = c: j] -> part[tap][c][o]
        float* dst = p.part + ((int64_t)((int)z * p.npairs + pair) * TAPS) * 4096 + (cf * 16 + j) * 64 + (2 * coh) * 16 + 4 * kg;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int t = 0; t < TAPS; ++t) *reinterpret_cast<float4*>(dst + t * 4096 + c2 * 16) = make_float4(acc[c2][t][0], acc[c2][t][1], acc[c2][t][2], acc[c2][t][3]);
    }
}
// dw[o][c][tap] = scale * sum over z (fixed order, fp64) of part[z][pair][tap][c % 64][o % 64].  A block owns 64 consecutive (pair, tap, o, c) indices (one
// 256-byte row of every partial tile): thread (tx = 16 float4 columns, ty = 16 z residues) sums its z subset, the 16 subsets are added in order through LDS.
__device__ __forceinline__ void qd_wgrad_reduce_block(const float* __restrict__ part, float* __restrict__ dw, int O, int C, int T, int Z, float scale, uint32_t blk,
                                                      double (&red)[16][64]) {
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int npairs = (O / 64) * (C / 64);
    const int64_t tile = (int64_t)npairs * T * 4096;
    const int64_t idx0 = (int64_t)blk * 64 + 4 * tx;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4                                  // (independent loads: keep several in flight, the kernel is pure latency otherwise)
    for (int zz = ty; zz < Z; zz += 16) {
        const float4 v = *reinterpret_cast<const float4*>(part + (int64_t)zz * tile + idx0);
        s[0] += (double)v.x; s[1] += (double)v.y; s[2] += (double)v.z; s[3] += (double)v.w;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[ty][4 * tx + e] = s[e];
    __syncthreads();
    if (threadIdx.x < 64) {
        double a = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) a += red[k][threadIdx.x];
        const int64_t idx = (int64_t)blk * 64 + threadIdx.x;
        const int o = (int)(idx & 63), c = (int)((idx >> 6) & 63), t = (int)((idx >> 12) % T), pair = (int)((idx >> 12) / T);
        const int cot = pair / (C / 64), cit = pair - cot * (C / 64);
        dw[((int64_t)(cot * 64 + o) * C + cit * 64 + c) * T + t] = (float)(a * (double)scale);
    }
}
__global__ __launch_bounds__(256) void k_qd_wgrad_reduce(const float* __restrict__ part, float* __restrict__ dw, int O, int C, int T, int Z, float scale_c,
                                                         const float* __restrict__ scale_p) {
    __shared__ double red[16][64];
    qd_wgrad_reduce_block(part, dw, O, C, T, Z, scale_p ? scale_p[0] : scale_c, blockIdx.x, red);          // IAO: the activation scale lives on the device
}
struct QdwPlan { QdwParams p; int S, T, grid; size_t lds; int64_t ws_bytes; };
static int plan_qdw(const mn_conv_geom* g, QdwPlan* pl) {
    if (!qd_geom_ok(g)) return 0;
    QdwParams& p = pl->p;
    const int S = g->stride_h, T = g->KH * g->KW;
    pl->S = S; pl->T = T;
    p.N = g->N; p.C = g->C; p.O = g->O; p.HX = g->H; p.WX = g->W; p.Hg = g->H / S; p.Wg = g->W / S; p.HWg = p.Hg * p.Wg; p.PAD = g->pad_h;
    p.w_shift = qd_log2(p.Wg);
    if (p.w_shift < 2 || p.Wg > 32 || p.HWg % 8) return 0;
    if ((int64_t)g->N * g->C * g->H * g->W >= ((int64_t)1 << 31) || (int64_t)g->N * g->O * p.HWg >= ((int64_t)1 << 31)) return 0;
    int ok = 0;
    for (int pdb = 1; pdb >= 0 && !ok; --pdb)                 // two patch buffers + 30 KB of gy planes in 160 KB, else one
    for (int BMt = 256; BMt >= 64; BMt >>= 1) {          // (>= 64: an even number of K-steps per tile, the producers stage K-steps in pairs)
        int TH, NI;
        if (p.HWg >= BMt) { if (BMt % p.Wg) continue; TH = BMt / p.Wg; if (p.Hg % TH) continue; NI = 1; }
        else { if (BMt % p.HWg) continue; NI = BMt / p.HWg; TH = p.Hg; }
        const int PH = (TH - 1) * S + g->KH, PWp = p.Wg + 8, RB = PWp * 2 * S;
        if (PH > 255 || NI > 63) continue;
        int CS = NI * PH * RB;
        if (((CS / 8) & 1) == 0) CS += 8;                     // odd multiple of 8 bytes: the 16 channels of a fragment fall on distinct 8-byte bank groups
        const int nunits = 64 * NI * PH * (g->W / 4);
        if ((int64_t)(1 + pdb) * 64 * CS > 128 * 1024 || nunits > 256 * QDW_PUPT || CS > 4096 || g->W / 4 > 15) continue;          // (the packed unit word of the producers)
        p.BMt = BMt; p.TH = TH; p.NI = NI; p.PH = PH; p.PWp = PWp; p.RB = RB; p.CS = CS; p.nunits = nunits; p.nks = BMt / 32; p.pdb = pdb;
        ok = 1;
        break;
    }
    if (!ok) return 0;
    p.W4 = g->W / 4;
    p.tpi = p.NI == 1 ? p.Hg / p.TH : 1;
    p.ntiles = p.NI == 1 ? g->N * p.tpi : (g->N + p.NI - 1) / p.NI;
    p.ncit = g->C / 64; p.npairs = (g->O / 64) * p.ncit;
    int tgt = 256;
    int Z = tgt / p.npairs;
    if (Z > p.ntiles) Z = p.ntiles;
    if (Z < 1) Z = 1;
    p.tpz = (p.ntiles + Z - 1) / Z;
    Z = (p.ntiles + p.tpz - 1) / p.tpz;
    p.Z = Z;
    p.fd_w4 = make_fastdiv((uint32_t)p.W4); p.fd_ph = make_fastdiv((uint32_t)p.PH); p.fd_ni = make_fastdiv((uint32_t)p.NI);
    p.fd_th = make_fastdiv((uint32_t)p.TH); p.fd_tpi = make_fastdiv((uint32_t)p.tpi); p.fd_np = make_fastdiv((uint32_t)p.npairs); p.fd_nks = make_fastdiv((uint32_t)p.nks);
    pl->grid = p.npairs * Z;
    pl->lds = (size_t)2 * 3 * QDW_DYP + (size_t)(1 + p.pdb) * 64 * p.CS + 16;
    pl->ws_bytes = (int64_t)Z * p.npairs * T * 4096 * 4;
    return 1;
}
// ------------------------------------------------------------------------------------------------ backward-weight, stride 1 / 3 x 3: second organisation (round 6)
// Knock-out builds of k_qd_wgrad<1, 9, 2> on the resnet18 shapes (56 us; profiles/README.md, round 6) put its time at: launch + set-up 8 us, the 32 block-wide
// hand-overs of a block 10 us with NOTHING between them, the staging waves' own instruction stream 15 us (a branch per staged 4-code unit: ~250 branches per
// pair of K-steps), consumer LDS reads 7, MFMA 13, the partial tile's store 7 -- added up, not overlapped.  Same data flow, same partial-tile layout, but:
//   * a hand-over is 64 gy-domain pixels (two MFMA K-steps): 4 per 256-pixel tile instead of 8;
//   * the staging waves run straight-line code: a patch unit is 16 codes of one channel plane (one dwordx4: 4 or 5 units per thread and tile, committed in
//     the tile's four steps by static index), addresses from shifts (W, H W powers of two), no per-unit words or branches; gy as 4 float4 per thread and step;
//   * signed / unsigned codes, the term count and the steps per tile (SPT 4: 256-pixel tiles; 2: 128-pixel tiles where two 256-pixel patches do not fit, W = 4) are
//     template parameters;
//   * neighbouring patch rows share their zero pads (row stride W + 4 elements instead of W + 8): two 8 x 8 x 4-image patches fit beside the gy planes.
#define QW2_ROWB 144                          // bytes per gy term-plane row: 64 pixels bf16 + 16
#define QW2_PLANE (64 * QW2_ROWB)
#define QW2_UPT 5
struct Qdw2Params {
    const float* gy;              // [N][O][H][W]
    const unsigned char* x;       // [N][C][H][W] codes
    float* part;
    int N, C, O, H, W, HW, wsh, hwsh;
    int NI, TH, PH, RB, CS;       // images per tile, gy rows per tile image, patch rows per image, bytes per patch row / per channel
    int tpish, band;              // log2 tiles per image (NI == 1); band: the tile is a band of rows of one image (the halo rows hold data)
    int ntiles, tpz, Z, ncit, npairs, nu;
    FastDiv fd_np;
};
template <int NT, int XSGN, int SPT>
__global__ __launch_bounds__(768) void k_qd_wgrad2(const Qdw2Params p) {
    HIP_DYNAMIC_SHARED(float, smem)
    unsigned char* dyb = reinterpret_cast<unsigned char*>(smem);          // [2][NT][64][144]
    unsigned char* xp0 = dyb + 2 * NT * QW2_PLANE;                        // [2][64 CS]
    const int XPB = 64 * p.CS;
    const int tid = threadIdx.x, lane = tid & 63, wave = mn_uniform(tid >> 6), j = lane & 15, kg = lane >> 4;
    const uint32_t z = fd_div(blockIdx.x, p.fd_np);
    const int pair = (int)blockIdx.x - (int)z * p.npairs;
    const int cot = pair / p.ncit, cit = pair - cot * p.ncit;
    const int t_begin = (int)z * p.tpz, t_end = (t_begin + p.tpz) < p.ntiles ? (t_begin + p.tpz) : p.ntiles;
    const int ns = t_begin < t_end ? SPT * (t_end - t_begin) : 0;           // hand-over steps of this block (64 pixels each)
    auto tile_origin = [&](int T, int& n0, int& oh0) {
        if (p.NI == 1) { n0 = T >> p.tpish; oh0 = (T & ((1 << p.tpish) - 1)) * p.TH; }
        else { n0 = T * p.NI; oh0 = 0; }
    };
    if (wave < 4) {
        // ------------------------------------------------------------------------------------------------ producers
        const int pc = tid >> 2, pq = tid & 3;
        int u_g[QW2_UPT], u_l[QW2_UPT], u_m[QW2_UPT];                 // global offset inside the tile's window, LDS offset of the unit's first dword, patch row | image << 8
#pragma unroll
        for (int i = 0; i < QW2_UPT; ++i) {
            const int g16 = 16 * (pq + 4 * i);
            if (p.band) {
                const int pr = g16 >> p.wsh, col = g16 & (p.W - 1);
                u_g[i] = g16; u_l[i] = pc * p.CS + pr * p.RB + (4 + col) * 2; u_m[i] = pr;
            } else {
                const int m = g16 >> p.hwsh, q = g16 & (p.HW - 1), row = q >> p.wsh, col = q & (p.W - 1);
                u_g[i] = m * p.C * p.HW + q; u_l[i] = pc * p.CS + (m * p.PH + 1 + row) * p.RB + (4 + col) * 2; u_m[i] = 1 | (m << 8);
            }
        }
        int de[4];                                                    // LDS offset of dword e of a unit relative to its first (a unit spans 16 / W rows when W < 16)
#pragma unroll
        for (int e = 0; e < 4; ++e) de[e] = ((4 * e) >> p.wsh) * p.RB + ((4 * e) & (p.W - 1)) * 2;
        const int safe = (cit * 64 + pc) * p.HW;                      // a valid address for the units that lie outside the image / batch
        u32x4 preg[QW2_UPT];
        uint32_t pok = 0u;
        auto fetch_unit = [&](int i, int T) {
            int n0, oh0;
            tile_origin(T, n0, oh0);
            const int base = (p.band ? ((n0 * p.C + cit * 64) * p.H + oh0 - 1) * p.W : (n0 * p.C + cit * 64) * p.HW) + pc * p.HW;
            const int ih = oh0 - 1 + (u_m[i] & 255), nn = n0 + (u_m[i] >> 8);
            const bool ok = (unsigned)ih < (unsigned)p.H && nn < p.N;
            const int off = ok ? base + u_g[i] : safe;
            preg[i] = *reinterpret_cast<const u32x4*>(p.x + (uint32_t)off);
            pok = (pok & ~(1u << i)) | ((ok ? 1u : 0u) << i);
        };
        auto commit_unit = [&](int i, unsigned char* xp) {
            const uint32_t okm = 0u - ((pok >> i) & 1u);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t v = preg[i][e] & okm;                    // (a mask, not a select: no branch around the staged words)
                float f0, f1, f2, f3;
                if (XSGN) { f0 = (float)(int)(signed char)(v & 0xffu); f1 = (float)(int)(signed char)((v >> 8) & 0xffu); f2 = (float)(int)(signed char)((v >> 16) & 0xffu); f3 = (float)(int)(signed char)(v >> 24); }
                else { f0 = (float)(v & 0xffu); f1 = (float)((v >> 8) & 0xffu); f2 = (float)((v >> 16) & 0xffu); f3 = (float)(v >> 24); }
                *reinterpret_cast<u32x2*>(xp + u_l[i] + de[e]) = u32x2{mn_pack_hi16(f0, f1), mn_pack_hi16(f2, f3)};
            }
        };
        // gy: rows go + 16 i, float4 column gc of the step's 64 pixels
        const int go = tid >> 4, gc = tid & 15;
        float4 G[2][4];
        int gok[2] = {0, 0};
        auto fetch_gy = [&](int b, int s) {
            const int sc = s < ns ? s : ns - 1;                        // (past the block's range its own last step is read again: an L2 hit, never committed)
            int n0, oh0;
            tile_origin(t_begin + sc / SPT, n0, oh0);
            const int pt = 64 * (sc % SPT) + 4 * gc;
            int nn = n0 + (pt >> p.hwsh);
            const int q = pt & (p.HW - 1);
            gok[b] = nn < p.N;
            nn = nn < p.N ? nn : p.N - 1;
            const float* src = p.gy + (uint32_t)((nn * p.O + cot * 64 + go) * p.HW + oh0 * p.W + q);
#pragma unroll
            for (int i = 0; i < 4; ++i) G[b][i] = *reinterpret_cast<const float4*>(src + (uint32_t)(16 * i * p.HW));
        };
        auto commit_gy = [&](int b) {                                  // register stage b -> gy buffer b (the step's parity)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t gm = 0u - (uint32_t)gok[b];
                const float4 g4 = make_float4(mn_u2f(mn_f2u(G[b][i].x) & gm), mn_u2f(mn_f2u(G[b][i].y) & gm), mn_u2f(mn_f2u(G[b][i].z) & gm), mn_u2f(mn_f2u(G[b][i].w) & gm));
                unsigned char* d = dyb + b * NT * QW2_PLANE + (go + 16 * i) * QW2_ROWB + gc * 8;
                if (NT == 2) {
                    unsigned h0, l0, h1, l1;
                    mn_split2_bf16x2(g4.x, g4.y, h0, l0);
                    mn_split2_bf16x2(g4.z, g4.w, h1, l1);
                    *reinterpret_cast<u32x2*>(d) = u32x2{h0, h1};
                    *reinterpret_cast<u32x2*>(d + QW2_PLANE) = u32x2{l0, l1};
                } else {
                    const float v[4] = {g4.x, g4.y, g4.z, g4.w};
                    float t0[4], t1[4], t2[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        t0[e] = mn_bf16_head(v[e]);
                        const float r1 = v[e] - t0[e];
                        t1[e] = mn_bf16_head(r1);
                        t2[e] = r1 - t1[e];
                    }
                    *reinterpret_cast<u32x2*>(d) = u32x2{mn_pack_bf16x2(t0[0], t0[1]), mn_pack_bf16x2(t0[2], t0[3])};
                    *reinterpret_cast<u32x2*>(d + QW2_PLANE) = u32x2{mn_pack_bf16x2(t1[0], t1[1]), mn_pack_bf16x2(t1[2], t1[3])};
                    *reinterpret_cast<u32x2*>(d + 2 * QW2_PLANE) = u32x2{mn_pack_bf16x2(t2[0], t2[1]), mn_pack_bf16x2(t2[2], t2[3])};
                }
            }
        };
        if (ns > 0) {
            fetch_gy(0, 0);
            fetch_gy(1, 1);
#pragma unroll
            for (int i = 0; i < QW2_UPT; ++i) if (i < p.nu) fetch_unit(i, t_begin);
        }
        for (int i = tid; i < (2 * XPB) / 8; i += 768) *reinterpret_cast<u32x2*>(xp0 + 8 * i) = u32x2{0u, 0u};          // the zero frame of both patch buffers (all 12 waves)
        __syncthreads();                          // barrier Z
        if (ns > 0) {
#pragma unroll
            for (int i = 0; i < QW2_UPT; ++i) {
                if (i >= p.nu) break;                                 // (uniform)
                commit_unit(i, xp0);
                if (t_begin + 1 < t_end) fetch_unit(i, t_begin + 1);
            }
            commit_gy(0);
            fetch_gy(0, 2);
        }
        __syncthreads();                          // barrier 0: step 0 (and the first tile's patch) may be read
        // Barrier s (behind the commit of step s into gy buffer s & 1) releases the consumers' reads of step s; they arrive at barrier s + 1 with every read of step s
        // complete, so buffer s & 1 is overwritten with step s + 2 behind barrier s + 1.  Patch: tile t + 1's buffer is free once the consumers passed the barrier of
        // tile t's first step and must be complete at the barrier of tile t + 1's first step: units 0, 1 | 2 | 3 | 4 go with the four steps in between.
        for (int s0 = 0; s0 < ns; s0 += SPT) {
            const int tl = s0 / SPT;                                   // tile of steps s0 .. s0 + SPT - 1 (relative to t_begin)
#pragma unroll
            for (int ts = 0; ts < SPT; ++ts) {
                if (ts == 0 && s0 == 0) continue;                     // (step 0 went with the prologue)
                commit_gy(ts & 1);
                fetch_gy(ts & 1, s0 + ts + 2);
                const int ptl = ts == 0 ? tl : tl + 1;                // the tile whose patch this step completes a part of
                if (t_begin + ptl < t_end) {
                    unsigned char* xp = xp0 + (ptl & 1) * XPB;
                    const bool more = t_begin + ptl + 1 < t_end;
#pragma unroll
                    for (int i = 0; i < QW2_UPT; ++i) {
                        // SPT 4: units 0, 1 | 2 | 3 | 4 with steps 1, 2, 3, 0;  SPT 2: unit 0 | 1 with steps 1, 0
                        const int when = SPT == 4 ? (i < 2 ? 1 : i == 2 ? 2 : i == 3 ? 3 : 0) : (i == 0 ? 1 : 0);
                        if (when != ts || i >= (SPT == 4 ? QW2_UPT : 2)) continue;
                        if (i >= 4 && p.nu <= 4) continue;          // (uniform: only the 32-wide band has a fifth unit)
                        commit_unit(i, xp);
                        if (more) fetch_unit(i, t_begin + ptl + 1);
                    }
                }
                __syncthreads();                  // barrier s0 + ts
            }
        }
    } else {
        // ------------------------------------------------------------------------------------------------ consumers (fragment reads + MFMA)
        const int cwv = wave - 4, cf = cwv & 3, coh = cwv >> 2;
        f32x4 acc[2][9];
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int t = 0; t < 9; ++t) acc[c2][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        struct AFrag { u32x4 a[2][NT]; };
        struct BRow { uint32_t w[2][4]; };                                // raw words of one kernel row: [half][previous, two own, next]
        AFrag aa[2];
        BRow bb[2];
        auto load_a = [&](AFrag& A, int buf, int kk) {
            const unsigned char* gb = dyb + buf * NT * QW2_PLANE + ((2 * coh) * 16 + j) * QW2_ROWB + kk * 64 + kg * 16;
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int t = 0; t < NT; ++t) A.a[c2][t] = *reinterpret_cast<const u32x4*>(gb + t * QW2_PLANE + c2 * 16 * QW2_ROWB);
        };
        int boff[2];
        auto set_b = [&](int ss) {                                        // K-step ss (32 pixels) of the tile: this lane's pixels 32 ss + 8 kg + 4 h ..
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int P = 32 * ss + 8 * kg + 4 * h;
                const int img = P >> p.hwsh, q = P & (p.HW - 1), ohl = q >> p.wsh, col = q & (p.W - 1);
                boff[h] = (cf * 16 + j) * p.CS + (img * p.PH + ohl) * p.RB + (4 + col) * 2;
            }
        };
        auto load_b = [&](BRow& B, int par, int r) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned char* q = xp0 + par * XPB + boff[h] + r * p.RB;
                const u32x2 c = *reinterpret_cast<const u32x2*>(q);
                B.w[h][0] = *reinterpret_cast<const uint32_t*>(q - 4); B.w[h][1] = c[0]; B.w[h][2] = c[1]; B.w[h][3] = *reinterpret_cast<const uint32_t*>(q + 8);
            }
        };
        auto mma = [&](const AFrag& A, const BRow& B, int r) {
            u32x4 b[3];
            {
                uint32_t lo[3][2], hi[3][2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t pv = B.w[h][0], c0 = B.w[h][1], c1 = B.w[h][2], nx = B.w[h][3];
                    lo[0][h] = mn_alignbyte(c0, pv, 2); hi[0][h] = mn_alignbyte(c1, c0, 2);
                    lo[1][h] = c0; hi[1][h] = c1;
                    lo[2][h] = mn_alignbyte(c1, c0, 2); hi[2][h] = mn_alignbyte(nx, c1, 2);
                }
#pragma unroll
                for (int s_ = 0; s_ < 3; ++s_) b[s_] = u32x4{lo[s_][0], hi[s_][0], lo[s_][1], hi[s_][1]};
            }
#pragma unroll
            for (int t = 0; t < NT; ++t)                                   // term-outer: MFMAs on the same accumulator are 6 instructions apart
#pragma unroll
                for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2) acc[c2][r * 3 + s_] = mn_mfma_bf16(A.a[c2][t], b[s_], acc[c2][r * 3 + s_]);
        };
        for (int i = tid; i < (2 * XPB) / 8; i += 768) *reinterpret_cast<u32x2*>(xp0 + 8 * i) = u32x2{0u, 0u};
        __syncthreads();                          // barrier Z
        __syncthreads();                          // barrier 0
        if (ns > 0) {
            int ss = 0, par = 0;
            set_b(0);
            load_a(aa[0], 0, 0);
            load_b(bb[0], 0, 0);
            for (int s0 = 0; s0 < ns; s0 += 2) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {                              // step s0 + u reads gy buffer u
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int r = 0; r < 3; ++r) {
                            const int ph = kk * 3 + r;                    // (six phases per step: the two word sets alternate consistently)
                            if (r < 2) load_b(bb[(ph + 1) & 1], par, r + 1);
                            else if (kk == 0) {
                                ++ss;
                                set_b(ss);
                                load_a(aa[1], u, 1);
                                load_b(bb[(ph + 1) & 1], par, 0);
                            } else if (s0 + u + 1 < ns) {
                                __syncthreads();                          // barrier s0 + u + 1
                                if (++ss == 2 * SPT) { ss = 0; par ^= 1; }
                                set_b(ss);
                                load_a(aa[0], u ^ 1, 0);
                                load_b(bb[0], par, 0);
                            }
                            MN_SCHED_FENCE();
                            mma(aa[kk], bb[ph & 1], r);
                        }
                }
            }
        }
        float* dst = p.part + ((int64_t)((int)z * p.npairs + pair) * 9) * 4096 + (cf * 16 + j) * 64 + (2 * coh) * 16 + 4 * kg;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int t = 0; t < 9; ++t) *reinterpret_cast<float4*>(dst + t * 4096 + c2 * 16) = make_float4(acc[c2][t][0], acc[c2][t][1], acc[c2][t][2], acc[c2][t][3]);
    }
}
struct Qdw2Plan { Qdw2Params p; int SPT, grid; size_t lds; int64_t ws_bytes; };
static int plan_qdw2(const mn_conv_geom* g, int NT, Qdw2Plan* pl) {
    if (!qd_geom_ok(g) || g->stride_h != 1 || g->KH != 3 || g->KW != 3 || g->pad_h != 1) return 0;
    Qdw2Params& p = pl->p;
    p.N = g->N; p.C = g->C; p.O = g->O; p.H = g->H; p.W = g->W; p.HW = g->H * g->W;
    p.wsh = qd_log2(p.W); p.hwsh = qd_log2(p.HW);
    if (p.wsh < 2 || p.W > 32 || p.hwsh < 4) return 0;
    if ((int64_t)g->N * g->C * p.HW >= ((int64_t)1 << 31) || (int64_t)g->N * g->O * p.HW >= ((int64_t)1 << 31)) return 0;
    p.ncit = g->C / 64; p.npairs = (g->O / 64) * p.ncit;
    p.fd_np = make_fastdiv((uint32_t)p.npairs);
    for (int SPT = 4; SPT >= 2; SPT >>= 1) {
        const int TP = 64 * SPT, tsh = SPT == 4 ? 8 : 7;          // pixels per tile
        if (p.HW >= TP) { p.NI = 1; p.TH = TP / p.W; p.tpish = p.hwsh - tsh; }
        else { p.NI = TP / p.HW; p.TH = p.H; p.tpish = 0; }
        p.band = p.TH < p.H;
        if (p.band && p.W < 16) continue;                     // (a 16-code unit must lie inside one row of a band)
        p.PH = p.TH + 2; p.RB = (p.W + 4) * 2;                  // rows share their pads: [4 zeros][W codes] per row, 4 more zeros behind the channel's last row
        p.CS = p.NI * p.PH * p.RB + 8;
        if (((p.CS / 8) & 1) == 0) p.CS += 8;                 // odd multiple of 8 bytes: the 16 channels of a fragment fall on distinct 8-byte bank groups
        const int upc = p.band ? p.PH * p.W / 16 : TP / 16;   // 16-code units per channel and tile
        if (upc % 4 || upc / 4 > (SPT == 4 ? QW2_UPT : 2)) continue;
        p.nu = upc / 4;
        pl->lds = (size_t)2 * NT * QW2_PLANE + (size_t)2 * 64 * p.CS;
        if (pl->lds > 160 * 1024) continue;
        p.ntiles = p.NI == 1 ? g->N << p.tpish : (g->N + p.NI - 1) / p.NI;
        int Z = 256 / p.npairs;
        if (Z > p.ntiles) Z = p.ntiles;
        if (Z < 1) Z = 1;
        p.tpz = (p.ntiles + Z - 1) / Z;
        p.Z = (p.ntiles + p.tpz - 1) / p.tpz;
        pl->SPT = SPT;
        pl->grid = p.npairs * p.Z;
        pl->ws_bytes = (int64_t)p.Z * p.npairs * 9 * 4096 * 4;
        return 1;
    }
    return 0;
}
int qd_wgrad_supported(const mn_conv_geom* g, int a_bits) {
    if (a_bits < 2 || a_bits > 7) return 0;
    QdwPlan pl;
    return plan_qdw(g, &pl);
}
int64_t qd_wgrad_ws_bytes(const mn_conv_geom* g) {
    QdwPlan pl;
    Qdw2Plan p2;
    const int64_t a = plan_qdw(g, &pl) ? pl.ws_bytes : 0, b = plan_qdw2(g, qd_terms(), &p2) ? p2.ws_bytes : 0;
    return a > b ? a : b;
}
int qd_bwd_weight(const mn_conv_geom* g, const float* gy, const uint8_t* x, float ascale, float* dw, void* ws, int64_t ws_bytes, hipStream_t s) {
    return qd_bwd_weight_ex(g, gy, x, 0, ascale, nullptr, dw, ws, ws_bytes, s);
}
// xsgn: x holds signed codes; ascale_dev != nullptr: the activation scale is read from the device (IAO qparams snapshot)
int qd_bwd_weight_ex(const mn_conv_geom* g, const float* gy, const uint8_t* x, int xsgn, float ascale, const float* ascale_dev, float* dw, void* ws, int64_t ws_bytes,
                     hipStream_t s) {
    {
        Qdw2Plan p2;
        const int NT = qd_terms();
        if (plan_qdw2(g, NT, &p2) && aligned16(gy) && aligned16(x) && ws && ws_bytes >= p2.ws_bytes && aligned16(ws)) {
            Qdw2Params& q = p2.p;
            q.gy = gy; q.x = x; q.part = reinterpret_cast<float*>(ws);
            mn_set_last_kernel("k_qd_wgrad2<%d, %d, %d>", NT, xsgn ? 1 : 0, p2.SPT);
            { const double nx = (double)g->N * g->C * g->H * g->W, ny = (double)g->N * g->O * q.HW; mn_prof_bytes(4.0 * ny * q.ncit + nx * (g->O / 64) + (double)p2.ws_bytes); mn_prof_flops(2.0 * ny * g->C * 9); }
            mn_prof_begin(s);
#define QW2_LAUNCH(N_, X_, S_) do { raise_lds_limit((const void*)k_qd_wgrad2<N_, X_, S_>, p2.lds); hipLaunchKernelGGL((k_qd_wgrad2<N_, X_, S_>), dim3(p2.grid), dim3(768), p2.lds, s, q); } while (0)
#define QW2_LAUNCH_S(N_, X_) do { if (p2.SPT == 4) QW2_LAUNCH(N_, X_, 4); else QW2_LAUNCH(N_, X_, 2); } while (0)
            if (NT == 2) { if (xsgn) QW2_LAUNCH_S(2, 1); else QW2_LAUNCH_S(2, 0); }
            else { if (xsgn) QW2_LAUNCH_S(3, 1); else QW2_LAUNCH_S(3, 0); }
#undef QW2_LAUNCH_S
#undef QW2_LAUNCH
            mn_prof_end(s);
            const int total = q.npairs * 9 * 4096;
            hipLaunchKernelGGL(k_qd_wgrad_reduce, dim3((unsigned)(total / 64)), dim3(256), 0, s, (const float*)q.part, dw, (int)g->O, (int)g->C, 9, q.Z, ascale, ascale_dev);
            MN_CHECK_LAUNCH("mn_conv2d_bwd_weight(dense)");
            return MN_OK;
        }
    }
    QdwPlan pl;
    if (!plan_qdw(g, &pl) || !aligned16(gy) || (((uintptr_t)x) & 3)) MN_FAIL(MN_ENOTSUP, "mn_conv2d_bwd_weight(dense): geometry not covered");
    if (!ws || ws_bytes < pl.ws_bytes || !aligned16(ws)) MN_FAIL(MN_ENOSPC, "mn_conv2d_bwd_weight(dense): workspace too small");
    QdwParams& p = pl.p;
    p.gy = gy; p.x = x; p.part = reinterpret_cast<float*>(ws); p.xsgn = xsgn;
    const int NT = qd_terms();
    mn_set_last_kernel("k_qd_wgrad<%d, %d, %d>", pl.S, pl.T, NT);
    { const double nx = (double)g->N * g->C * g->H * g->W, ny = (double)g->N * g->O * p.HWg; mn_prof_bytes(4.0 * ny * p.ncit + nx * (g->O / 64) + (double)pl.ws_bytes); mn_prof_flops(2.0 * ny * g->C * pl.T); }
    mn_prof_begin(s);
#define QDW_LAUNCH(S_, T_, N_) do { raise_lds_limit((const void*)k_qd_wgrad<S_, T_, N_>, pl.lds); hipLaunchKernelGGL((k_qd_wgrad<S_, T_, N_>), dim3(pl.grid), dim3(768), pl.lds, s, p); } while (0)
    if (pl.S == 1) { if (NT == 2) QDW_LAUNCH(1, 9, 2); else QDW_LAUNCH(1, 9, 3); }
    else if (pl.T == 9) { if (NT == 2) QDW_LAUNCH(2, 9, 2); else QDW_LAUNCH(2, 9, 3); }
    else { if (NT == 2) QDW_LAUNCH(2, 1, 2); else QDW_LAUNCH(2, 1, 3); }
#undef QDW_LAUNCH
    mn_prof_end(s);
    const int total = p.npairs * pl.T * 4096;
    hipLaunchKernelGGL(k_qd_wgrad_reduce, dim3((unsigned)(total / 64)), dim3(256), 0, s, (const float*)p.part, dw, (int)g->O, (int)g->C, pl.T, p.Z, ascale, ascale_dev);
    MN_CHECK_LAUNCH("mn_conv2d_bwd_weight(dense)");
    return MN_OK;
}


// ================================================================================================ IAO layers (wqaq/iao/quantize.py:492-507 QuantConv2d)
// Symmetric per-tensor activation quantizer (codes in [-2^(b-1), 2^(b-1) - 1], value = code * sa) and symmetric per-channel / per-layer weight quantizer
// (w = wcode * sw[o]): y = sa sw[o] * sum wcode * code + bias.  The activation arrives as fp32 (the quantizer belongs to this conv in the reference), so each
// direction first writes the codes once -- k_qd_iao_codes, 4 B in / 1 B out per element -- into its workspace and then runs the dense kernels above on them:
// forward with the fp32 epilogue, backward-data with the per-channel scale folded into gy followed by the quantizer's clip-STE (k_qd_iao_ste, in place),
// backward-weight on the signed codes with the activation scale taken from the device snapshot.
// mask (nullable): one byte per thread = the clip-STE decisions of its 8 elements (iao_fq_grad's two conditions on the same v), read back by k_qd_dgrad's store
// (the pass is VALU-bound before it is HBM-bound: x / sc is Markstein's correctly rounded quotient -- mn_div_m, the float of the IEEE sequence for in-range operands --
// rha as copysign(floor(|v| + 0.5), v), and with zero_point == 0 (the symmetric quantizer: always) one rounded value serves the code and the clip test)
template <int ZP0>
__device__ __forceinline__ void qd_iao_codes_body(const float* __restrict__ x, signed char* __restrict__ codes, unsigned char* __restrict__ mask, int64_t n8, float sc, float zp,
                                                  float rlo, float rhi, float qmin, float qmax) {
    const float inv = 1.0f / sc;
    auto rha = [](float v) { return copysignf(floorf(fabsf(v) + 0.5f), v); };          // mn_rha's float except for the sign of a zero (seen by neither the byte nor the tests)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const float4 a = *reinterpret_cast<const float4*>(x + 8 * i), b = *reinterpret_cast<const float4*>(x + 8 * i + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        uint32_t m = 0u, lo = 0u, hi = 0u;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            // (+-inf and anything the quotient's fma chain would overflow on: clamped far outside the quantizer's range first -- the same code and a failed clip
            // test, as the IEEE division gives; a NaN fails the test either way and its byte is forced below)
            const float xc = fminf(fmaxf(v[e], -1.0e30f), 1.0e30f);
            const float q = mn_div_m(xc, sc, inv);
            const float vv = ZP0 ? q : q - zp, r = rha(vv);
            m |= ((r >= qmin && r <= qmax && !(vv > rhi || vv < rlo)) ? 1u : 0u) << e;
            const float rq = ZP0 ? r : rha(q);
            // clamp(rha(x / sc), qmin, qmax): the integer of iao_fq (zero_point == 0); NaN -> 0 (a byte cannot hold it)
            const float cc = (v[e] == v[e]) ? fminf(fmaxf(rq, qmin), qmax) : 0.f;
            const uint32_t byte = (uint32_t)(int)cc & 0xffu;
            if (e < 4) lo |= byte << (8 * e); else hi |= byte << (8 * (e - 4));
        }
        if (mask) mask[i] = (unsigned char)m;
        *reinterpret_cast<u32x2*>(codes + 8 * i) = u32x2{lo, hi};
    }
}
__global__ __launch_bounds__(256) void k_qd_iao_codes(const float* __restrict__ x, signed char* __restrict__ codes, unsigned char* __restrict__ mask, int64_t n8,
                                                      const float* __restrict__ qp, float qmin, float qmax) {
    const float sc = qp[0], zp = qp[1], rlo = qp[2], rhi = qp[3];
    if (zp == 0.f) qd_iao_codes_body<1>(x, codes, mask, n8, sc, zp, rlo, rhi, qmin, qmax);
    else qd_iao_codes_body<0>(x, codes, mask, n8, sc, zp, rlo, rhi, qmin, qmax);
}
__global__ __launch_bounds__(256) void k_qd_iao_ste(float* __restrict__ dx, const float* __restrict__ x, int64_t n4, const float* __restrict__ qp, float qmin, float qmax) {
    const float sc = qp[0], zp = qp[1], lo = qp[2], hi = qp[3];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 g = *reinterpret_cast<const float4*>(dx + 4 * i), v = *reinterpret_cast<const float4*>(x + 4 * i);
        *reinterpret_cast<float4*>(dx + 4 * i) = make_float4(iao_fq_grad(g.x, v.x, sc, zp, lo, hi, qmin, qmax), iao_fq_grad(g.y, v.y, sc, zp, lo, hi, qmin, qmax),
                                                             iao_fq_grad(g.z, v.z, sc, zp, lo, hi, qmin, qmax), iao_fq_grad(g.w, v.w, sc, zp, lo, hi, qmin, qmax));
    }
}
static int qd_iao_quant_ok(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, int need_w) {
    if (!g || !aq || aq->mode != MN_ACTQ_IAO || aq->q_type != 0 || aq->bits < 2 || aq->bits > 8 || !aq->qp) return 0;
    int64_t wmax = 127;
    if (need_w) {
        if (!wq || wq->mode != MN_WQ_IAO || wq->q_type != 0 || wq->bits < 2 || wq->bits > 8 || !wq->scale) return 0;
        wmax = (1ll << (wq->bits - 1)) - 1;
    }
    const int64_t K = (int64_t)g->C * g->KH * g->KW, amax = 1ll << (aq->bits - 1);
    if (need_w == 1 && !qd_fwd_i8(wq) && K * amax * wmax >= (1ll << 24)) return 0;          // bf16 forward only: the fp32 accumulation of integer products must stay exact
    if (((int64_t)g->N * g->C * g->H * g->W) % 8) return 0;
    return 1;
}
int qd_iao_supported(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, int which);
static int64_t qd_iao_codes_bytes(const mn_conv_geom* g) { return ((int64_t)g->N * g->C * g->H * g->W + 255) / 256 * 256; }
extern "C" int64_t mn_conv2d_iao_stats_rows(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq) {
    QdfPlan pl;
    if (!qd_iao_quant_ok(g, aq, wq, 1) || !qd_fwd_i8(wq) || !plan_qdf(g, 2, &pl, 1)) return 0;
    return pl.grid / pl.p.ncot;
}
extern "C" int64_t mn_conv2d_iao_codes_bytes(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq) {
    if (!g || !aq || !wq || !qd_iao_supported(g, aq, wq, 0) || !qd_iao_supported(g, aq, nullptr, 2)) return 0;
    return qd_iao_codes_bytes(g);
}
int qd_iao_supported(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, int which) {
    if (which == 0) { QdfPlan pl; return qd_iao_quant_ok(g, aq, wq, 1) && plan_qdf(g, 2, &pl, qd_fwd_i8(wq)); }
    if (which == 1) { QddPlan pl; return qd_iao_quant_ok(g, aq, wq, 2) && plan_qdd(g, &pl); }
    if (which == 2) { QdwPlan pl; return qd_iao_quant_ok(g, aq, nullptr, 0) && plan_qdw(g, &pl); }
    return 0;
}
int64_t qd_iao_ws_bytes(const mn_conv_geom* g, int which) {
    if (which == 0) { QdfPlan pl; return (plan_qdf(g, 2, &pl, 0) || plan_qdf(g, 2, &pl, 1)) ? qd_iao_codes_bytes(g) + pl.ws_bytes : 0; }
    if (which == 1) { QddPlan pl; return plan_qdd(g, &pl) ? pl.ws_bytes : 0; }
    if (which == 2) { QdwPlan pl; return plan_qdw(g, &pl) ? qd_iao_codes_bytes(g) + pl.ws_bytes : 0; }
    return 0;
}
static void qd_iao_launch_codes(const mn_conv_geom* g, const mn_actq* aq, const float* x, void* codes, void* mask, hipStream_t s) {
    const int64_t n8 = (int64_t)g->N * g->C * g->H * g->W / 8;
    const IaoRange r = iao_range(aq->bits, 0, 1);
    int64_t nb = (n8 + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_qd_iao_codes, dim3((unsigned)nb), dim3(256), 0, s, x, (signed char*)codes, (unsigned char*)mask, n8, aq->qp, r.qmin, r.qmax);
}
int qd_iao_fwd(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, const float* x, const float* w, const float* bias, float* y, void* ws, int64_t ws_bytes,
               hipStream_t s) {
    QdfPlan pl;
    const int i8 = qd_fwd_i8(wq);
    const int given = (aq->flags & MN_ACTQ_CODES_GIVEN) != 0;          // mn_actq.codes already holds this forward's codes (mn_bn_apply_codes wrote them): x is not read
    if (!qd_iao_quant_ok(g, aq, wq, 1) || !plan_qdf(g, 2, &pl, i8) || (!given && !aligned16(x)) || !aligned16(y) || !w) MN_FAIL(MN_ENOTSUP, "mn_conv2d_fwd(dense iao): geometry / quantizer not covered");
    if (given && !aq->codes) MN_FAIL(MN_EINVAL, "mn_conv2d_fwd(dense iao): MN_ACTQ_CODES_GIVEN without mn_actq.codes");
    const int64_t cb = qd_iao_codes_bytes(g);
    if (!ws || ws_bytes < cb + pl.ws_bytes || !aligned16(ws)) MN_FAIL(MN_ENOSPC, "mn_conv2d_fwd(dense iao): workspace too small");
    QdfParams& p = pl.p;
    void* cbuf = aq->codes ? aq->codes : ws;          // a caller-owned buffer keeps the codes for backward-weight
    if (!aligned16(cbuf)) MN_FAIL(MN_EINVAL, "mn_conv2d_fwd(dense iao): mn_actq.codes is not 16-byte aligned");
    if (!given) qd_iao_launch_codes(g, aq, x, cbuf, aq->codes ? aq->ste_mask : nullptr, s);          // (+ the clip-STE bits for backward-data when the caller keeps the codes)
    const uint16_t* wpk = reinterpret_cast<const uint16_t*>(wq->packed_fwd);
    if (!wpk) {
        qd_launch_pack(w, reinterpret_cast<uint16_t*>((char*)ws + cb), g->O, g->C, p.TAPS, wq->bits, i8 ? 2 : 0, s, wq->scale, wq->per_channel);
        wpk = reinterpret_cast<const uint16_t*>((char*)ws + cb);
    }
    p.stats = (i8 && aq->stats) ? reinterpret_cast<double*>(aq->stats) : nullptr;          // exact sums of acc for the BatchNorm behind (mn_bn_fwd_acc)
    p.accmm = (i8 && aq->acc_mm) ? reinterpret_cast<int32_t*>(aq->acc_mm) : nullptr;          // ... and its extrema (mn_bn_acc_prep)
    p.x = (const unsigned char*)cbuf; p.wpk = wpk; p.stash = y; p.xsgn = 1; p.sa = aq->qp; p.sw = wq->scale; p.sw_stride = wq->per_channel; p.bias = bias;
    mn_set_last_kernel(i8 ? "k_qd_fwd8<%d, %d>" : "k_qd_fwd<%d, %d>", pl.MF, pl.TPS);
    { const double nx = (double)g->N * g->C * p.HW, ny = (double)g->N * g->O * p.HoWo; mn_prof_bytes(nx * p.ncot + 4.0 * ny); mn_prof_flops(2.0 * ny * g->C * p.TAPS); }
    mn_prof_begin(s);
    qd_launch_fwd(pl, s);
    mn_prof_end(s);
    MN_CHECK_LAUNCH("mn_conv2d_fwd(dense iao)");
    return MN_OK;
}
int qd_iao_dx_add_supported(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq) {
    return g && aq && wq && qd_iao_supported(g, aq, wq, 1);
}
int qd_iao_bwd_data(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, const float* gy, const float* w, const float* x, float* dx, void* ws, int64_t ws_bytes,
                    hipStream_t s) {
    QddPlan pl;
    const unsigned char* smask = aq && aq->codes ? reinterpret_cast<const unsigned char*>(aq->ste_mask) : nullptr;          // the forward's clip-STE bits: x is not read
    if (!qd_iao_quant_ok(g, aq, wq, 2) || !plan_qdd(g, &pl) || !aligned16(gy) || !aligned16(dx) || (!smask && (!x || !aligned16(x))) || !w)
        MN_FAIL(MN_ENOTSUP, "mn_conv2d_bwd_data(dense iao): geometry / quantizer not covered");
    if (!ws || ws_bytes < pl.ws_bytes || !aligned16(ws)) MN_FAIL(MN_ENOSPC, "mn_conv2d_bwd_data(dense iao): workspace too small");
    QddParams& p = pl.p;
    const uint16_t* wpk = reinterpret_cast<const uint16_t*>(wq->packed_bwd);
    if (!wpk) { qd_launch_pack(w, reinterpret_cast<uint16_t*>(ws), g->O, g->C, p.TAPS, wq->bits, 1, s, wq->scale, wq->per_channel); wpk = reinterpret_cast<const uint16_t*>(ws); }
    p.gy = gy; p.wpk = wpk; p.dx = dx; p.wscale = 1.0f; p.wsc = wq->scale; p.wsc_stride = wq->per_channel;
    const IaoRange r = iao_range(aq->bits, 0, 1);
    p.ste_x = smask ? nullptr : x; p.ste_mask = smask; p.ste_qp = aq->qp; p.ste_qmin = r.qmin; p.ste_qmax = r.qmax;          // the quantizer's clip-STE rides the store of dx
    if (aq->dx_add && !aligned16(aq->dx_add)) MN_FAIL(MN_ENOTSUP, "mn_conv2d_bwd_data(dense iao): mn_actq.dx_add needs a 16-byte aligned tensor");
    p.dx_add = aq->dx_add;
    mn_set_last_kernel("k_qd_dgrad<%d, %d, %d, %d>", pl.MF, pl.S, p.TAPS, qd_terms());
    { const double nx = (double)g->N * g->C * g->H * g->W, ny = (double)g->N * g->O * p.HWg; mn_prof_bytes(4.0 * ny * p.ncit + (smask ? 4.125 : 8.0) * nx + (p.dx_add ? 4.0 * nx : 0.0)); mn_prof_flops(2.0 * ny * g->C * p.TAPS); }
    mn_prof_begin(s);
    qd_launch_dgrad(pl, s);
    mn_prof_end(s);
    MN_CHECK_LAUNCH("mn_conv2d_bwd_data(dense iao)");
    return MN_OK;
}
// dbias[o] = sum over (n, pixels) of gy: one block per channel, fp64 (the IAO convs of the ResNets carry no bias; QuantConv2d(bias=True) elsewhere does)
__global__ __launch_bounds__(256) void k_qd_bias_grad(const float* __restrict__ gy, float* __restrict__ db, int N, int O, int HW) {
    __shared__ double scd[16];
    const int o = blockIdx.x;
    double a = 0.0;
    for (int64_t i = threadIdx.x; i < (int64_t)N * HW; i += 256) { const int64_t n = i / HW; a += (double)gy[(n * O + o) * HW + (i - n * HW)]; }
    a = block_reduce(a, OpAddD(), 0.0, scd);
    if (threadIdx.x == 0) db[o] = (float)a;
}
int qd_iao_bwd_weight(const mn_conv_geom* g, const mn_actq* aq, const float* gy, const float* x, float* dw, float* dbias, void* ws, int64_t ws_bytes, hipStream_t s) {
    QdwPlan pl;
    if (!qd_iao_quant_ok(g, aq, nullptr, 0) || !plan_qdw(g, &pl) || !aligned16(x)) MN_FAIL(MN_ENOTSUP, "mn_conv2d_bwd_weight(dense iao): geometry / quantizer not covered");
    const int64_t cb = qd_iao_codes_bytes(g);
    if (!ws || ws_bytes < cb + pl.ws_bytes || !aligned16(ws)) MN_FAIL(MN_ENOSPC, "mn_conv2d_bwd_weight(dense iao): workspace too small");
    const void* cbuf = aq->codes ? aq->codes : ws;
    if (!aq->codes) qd_iao_launch_codes(g, aq, x, ws, nullptr, s);          // else: the forward of this step left them there
    if (dbias) hipLaunchKernelGGL(k_qd_bias_grad, dim3((unsigned)g->O), dim3(256), 0, s, gy, dbias, (int)g->N, (int)g->O, pl.p.HWg);
    return qd_bwd_weight_ex(g, gy, (const uint8_t*)cbuf, 1, 1.f, aq->qp, dw, (char*)ws + cb, ws_bytes - cb, s);
}

// ================================================================================================ weight codes of a whole net in one launch
extern "C" int64_t mn_qd_packed_bytes(const mn_conv_geom* g) {
    if (!g || !qd_geom_ok(g)) return 0;
    return ((int64_t)g->O * g->C * g->KH * g->KW * 2 + 255) / 256 * 256;
}
extern "C" int mn_qd_pack_multi(const float* const* w, void* const* out_fwd, void* const* out_bwd, const int64_t* O, const int64_t* Cin, const int64_t* taps,
                                const float* const* wscale, const int32_t* wscale_stride, int32_t count, int w_bits, mn_stream_t stream) {
    if (count <= 0) return MN_OK;
    if (!w || !out_fwd || !out_bwd || !O || !Cin || !taps || w_bits < 2 || w_bits > 8) MN_FAIL(MN_EINVAL, "mn_qd_pack_multi: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    for (int base = 0; base < count; base += QD_PACK_MAX) {
        QdPackTable t;
        t.n = count - base < QD_PACK_MAX ? count - base : QD_PACK_MAX;
        t.wn = (float)((1ll << w_bits) - 1);
        { mn_wq q; q.mode = wscale ? MN_WQ_IAO : MN_WQ_DOREFA; q.bits = w_bits; t.fwd8 = qd_fwd_i8(&q); }
        int blk = 0, max_t = 1;
        for (int i = 0; i < t.n; ++i) {
            const int k = base + i;
            if (!w[k] || O[k] % 64 || Cin[k] % 64 || (taps[k] != 9 && taps[k] != 1) || (out_fwd[k] && !aligned16(out_fwd[k])) || (out_bwd[k] && !aligned16(out_bwd[k])))
                MN_FAIL(MN_EINVAL, "mn_qd_pack_multi: tensor %d is not a dense-family weight (O, C multiples of 64; 9 or 1 taps) or its output is misaligned", k);
            t.w[i] = w[k]; t.outf[i] = (uint16_t*)out_fwd[k]; t.outd[i] = (uint16_t*)out_bwd[k]; t.wsc[i] = wscale ? wscale[k] : nullptr;
            t.wsc_stride[i] = (wscale && wscale_stride) ? wscale_stride[k] : 0;
            t.O[i] = (int)O[k]; t.C[i] = (int)Cin[k]; t.T[i] = (int)taps[k];
            t.blk0[i] = blk;
            blk += (int)((O[k] / 64) * (Cin[k] / 64)) * 4;          // four 16-row slices per (64 o, 64 c) tile
            if (taps[k] > max_t) max_t = (int)taps[k];
        }
        t.blk0[t.n] = blk;
        const size_t lds = (size_t)QD_PACK_ROWS * 64 * max_t * 2;
        raise_lds_limit((const void*)k_qd_pack_multi, lds);
        mn_set_last_kernel("k_qd_pack_multi");
        hipLaunchKernelGGL(k_qd_pack_multi, dim3((unsigned)blk), dim3(QD_PACK_THREADS), lds, s, t);
    }
    MN_CHECK_LAUNCH("mn_qd_pack_multi");
    return MN_OK;
}
