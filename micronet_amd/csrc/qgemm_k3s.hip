// 3 x 3 (stride 1, padding 1) backward-weight on packed sign activations, for gfx950.
//
//   dwq[g][m][c][r][s] = sum over (n, oh, ow) of gy[n][g*Mg + m][oh][ow] * a[n][g*Cg + c][oh + r - 1][ow + s - 1]
//
// (the weight gradient of the grouped 3x3 QuantConv2d layers of nin_gc, wbwtab/quantize.py:181-195 through autograd's
// conv2d backward; a = the int8 sign codes of BinaryActivation).  The contraction index is the output pixel, and both operands
// are pixel-contiguous in NCHW.  MFMA shape per group tile: M = 32 output channels (2 row tiles), N = 16 input channels, one
// N tile PER TAP (9 accumulator tiles per row tile), K = 32 output pixels per step = 32 / W whole image rows.
//
// Every WAVE is an independent worker: it streams its own K-steps (the four waves of a block interleave over the block's step
// range, so neighbouring rows of the same planes are read together), with NO block barrier in the loop:
//   * global -> registers, coalesced, two steps ahead: 8 lanes read one 128-byte line of a gy row; the input patch of a step
//     ((R + 2) image rows x W codes per channel) is one contiguous byte range per channel, read dword by dword;
//   * registers -> the wave's private LDS image: gy rows padded to 160 B (conflict-free b128 fragment reads); the codes re-encoded
//     as the HIGH byte of a bf16 (+-0.5 = 0x3F00 / 0xBF00, 0 = the zero padding) in rows of W + 4 bytes, the 4 bytes between two
//     rows staying zero: left / right padding of every row, and rows outside the image are written as zeros (top / bottom);
//   * fragments: gy as three exact bf16 terms (as k_pws_wgrad_s); for tap (r, s) a lane takes the three dwords around its 4-pixel
//     chunk in patch row (local row + r), shifts by s - 1 bytes (v_alignbyte) and spreads the four bytes into two bf16 pairs
//     (v_perm): 6 VALU instructions per B fragment, no shifted copies in LDS.
// The +-0.5 encoding makes the products half of the true ones -- an exact power of two, undone by the reduction kernel's scale.
// At the end the four waves add their tiles through LDS in wave order (deterministic) and write one partial tile per block in the
// layout of k_kk_wgrad; k_pw_wgrad_reduce sums the Z partials in fp64.
#include "qgemm_dev.h"

#include <stdlib.h>

#define K3_RSA 160            // LDS bytes per gy row (32 pixels fp32 + pad)
#define K3_NXL 6              // code dwords a lane stages per step (W = 32: 16 channels x 3 rows x 8 dwords = 384 = 6 x 64)

struct K3wParams {
    const float* gy;
    const char* x;
    float* part;      // [Z][G][Mgw][Cgw*9]
    float* dbpart;    // [Z][G][Mgw]
    int N, C, H, W, O, Cg, Mg, G, nmb, ncb, Z, Mgw, Cgw, want_db;
    int R, spi, nsteps, st_per_z, DPC, W4, CS, XB;    // rows per step, steps per image, dwords per channel patch, W/4, LDS bytes per channel / per wave
    FastDiv fd_spi, fd_dpc, fd_w4;
    ChanMap in_map;
};

// XENC 1: x holds k-bit activation codes j <= 7 (a_bits <= 3): the patch keeps the raw bytes (0 = the zero padding = code 0) and the B fragment is
// built by byte look-up -- v_perm with the CODES as the selector into an 8-entry table of bf16 bit patterns (high / low byte), then two
// v_perm to interleave: exact bf16 j, 8 VALU per fragment dword pair instead of 2.  The reduction multiplies by the quantizer's scale.
// XENC 2: codes j <= 255 (a_bits 4 .. 8): same patch of raw bytes, the fragment is bf16 j = the high half of (float)j (v_cvt_f32_ubyte x 4 + 2 v_perm per
// shifted dword).
template <int XENC>
__global__ __launch_bounds__(256, 2) void k_k3s_wgrad(const K3wParams p) {
    HIP_DYNAMIC_SHARED(float, smem)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kg = lane >> 4;
    unsigned char* gsm = reinterpret_cast<unsigned char*>(smem) + wave * (32 * K3_RSA + p.XB);     // this wave's image: gy rows, then the code patch
    unsigned char* xsm = gsm + 32 * K3_RSA;
    uint32_t b = blockIdx.x;
    const int z = b % p.Z; b /= p.Z;
    const int cb = b % p.ncb; b /= p.ncb;
    const int mb = b % p.nmb;
    const int g = b / p.nmb;
    const uint32_t HW = (uint32_t)(p.H * p.W);

    // the zero dwords around the patch rows (never overwritten)
    for (int i = lane; i < 16 * (p.R + 3); i += 64) {
        const int c = i / (p.R + 3), k = i - c * (p.R + 3);
        *reinterpret_cast<uint32_t*>(xsm + c * p.CS + k * (p.W + 4)) = 0u;
    }
    // staging roles.  gy: row (lane >> 3) + 8 i, pixels 4 (lane & 7) ...;  codes: dword (lane + 64 u) of the [16][DPC] patch
    const int sr = lane >> 3, sq = lane & 7;
    uint32_t goff[4];
    float dbacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = mb * 32 + sr + 8 * i;
        m = m < p.Mg ? m : p.Mg - 1;
        goff[i] = (uint32_t)(g * p.Mg + m) * HW + 4u * sq;
        dbacc[i] = 0.f;
    }
    const int ccv = (p.Cg - cb * 16) < 16 ? (p.Cg - cb * 16) : 16;      // channels beyond Cg: zero codes
    uint32_t xoff[K3_NXL];      // byte offset of the dword inside image n, without its row
    int xlds[K3_NXL], xpr[K3_NXL];      // LDS byte offset (-1: no item); patch row
#pragma unroll
    for (int u = 0; u < K3_NXL; ++u) {
        const int idx = lane + 64 * u;
        const uint32_t c = fd_div(idx, p.fd_dpc);
        const int d = idx - (int)c * p.DPC;
        const uint32_t pr = fd_div(d, p.fd_w4);
        const int col4 = d - (int)pr * p.W4;
        const int cc = (int)c < ccv ? (int)c : ccv - 1;
        xoff[u] = (uint32_t)chan_phys(p.in_map, g * p.Cg + cb * 16 + cc) * HW + 4u * col4;
        xlds[u] = (idx < 16 * p.DPC) ? (int)c * p.CS + 4 + (int)pr * (p.W + 4) + 4 * col4 : -1;
        xpr[u] = ((int)c < ccv) ? (int)pr : -1000;      // a channel beyond Cg never has a valid row
    }
    f32x4 acc[2][9];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[mi][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    struct Stage { float4 gv[4]; uint32_t xv[K3_NXL]; int oh0; };
    Stage s0, s1;
    const int st0 = z * p.st_per_z;
    const int nblk = ((st0 + p.st_per_z) < p.nsteps ? (st0 + p.st_per_z) : p.nsteps) - st0;     // steps of this block
    const int nw = nblk > wave ? (nblk - wave + 3) / 4 : 0;                                         // ... of this wave: st0 + wave + 4 t
    // loads are unconditional (clamped step, clamped rows): steps past the range are never contracted
    auto fetch = [&](Stage& S, int t) {
        int st = st0 + wave + 4 * (t < nw ? t : (nw > 0 ? nw - 1 : 0));      // past the range: this wave's own last step again (an L2 hit, not the neighbour's data)
        st = st < p.nsteps ? st : p.nsteps - 1;
        const uint32_t n = fd_div(st, p.fd_spi);
        const int oh0 = (st - (int)n * p.spi) * p.R;
        S.oh0 = oh0;
        const uint32_t go = n * (uint32_t)p.O * HW + (uint32_t)(oh0 * p.W);
#pragma unroll
        for (int i = 0; i < 4; ++i) S.gv[i] = *reinterpret_cast<const float4*>(p.gy + (go + goff[i]));
        const uint32_t xo = n * (uint32_t)p.C * HW;
#pragma unroll
        for (int u = 0; u < K3_NXL; ++u) {
            int ir = oh0 - 1 + (xpr[u] < 0 ? 0 : xpr[u]);
            ir = ir < 0 ? 0 : (ir < p.H ? ir : p.H - 1);          // rows outside the image: any valid address (written as zeros)
            S.xv[u] = *reinterpret_cast<const uint32_t*>(p.x + (xo + xoff[u] + (uint32_t)(ir * p.W)));
        }
    };
    auto commit = [&](Stage& S, bool valid) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dbacc[i] += valid ? (S.gv[i].x + S.gv[i].y) + (S.gv[i].z + S.gv[i].w) : 0.f;
            *reinterpret_cast<float4*>(gsm + (sr + 8 * i) * K3_RSA + 16 * sq) = S.gv[i];
        }
#pragma unroll
        for (int u = 0; u < K3_NXL; ++u) {
            const int ir = S.oh0 - 1 + xpr[u];
            const uint32_t enc = (ir >= 0 && ir < p.H) ? (XENC ? S.xv[u] : ((S.xv[u] & 0x80808080u) | 0x3F3F3F3Fu)) : 0u;
            if (xlds[u] >= 0) *reinterpret_cast<uint32_t*>(xsm + xlds[u]) = enc;
        }
    };
    // fragment geometry of this lane: chunk a = pixels 4 kg .. + 3, chunk b = pixels 16 + 4 kg .. + 3 of the step
    int xa, xb;
    {
        const int pa = 4 * kg, pb = 16 + 4 * kg;
        const int ra = pa / p.W, rb = pb / p.W;
        xa = j * p.CS + 4 + ra * (p.W + 4) + (pa - ra * p.W);
        xb = j * p.CS + 4 + rb * (p.W + 4) + (pb - rb * p.W);
    }
    const int xrs = p.W + 4;
    auto contract = [&]() {
        u32x4 a0[2], a1[2], a2[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const unsigned char* row = gsm + (mi * 16 + j) * K3_RSA + 16 * kg;
            const float4 ga = *reinterpret_cast<const float4*>(row), gb = *reinterpret_cast<const float4*>(row + 64);
            const float v[8] = {ga.x, ga.y, ga.z, ga.w, gb.x, gb.y, gb.z, gb.w};
            float t0[8], t1[8], t2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                t0[e] = mn_bf16_head(v[e]);
                const float r1 = v[e] - t0[e];
                t1[e] = mn_bf16_head(r1);
                t2[e] = r1 - t1[e];
            }
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                a0[mi][d] = mn_pack_bf16x2(t0[2 * d], t0[2 * d + 1]);
                a1[mi][d] = mn_pack_bf16x2(t1[2 * d], t1[2 * d + 1]);
                a2[mi][d] = mn_pack_bf16x2(t2[2 * d], t2[2 * d + 1]);
            }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const unsigned char* qa = xsm + xa + r * xrs;
            const unsigned char* qb = xsm + xb + r * xrs;
            const uint32_t ap = *reinterpret_cast<const uint32_t*>(qa - 4), ac = *reinterpret_cast<const uint32_t*>(qa), an = *reinterpret_cast<const uint32_t*>(qa + 4);
            const uint32_t bp = *reinterpret_cast<const uint32_t*>(qb - 4), bc = *reinterpret_cast<const uint32_t*>(qb), bn = *reinterpret_cast<const uint32_t*>(qb + 4);
            u32x4 bf[3];
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_) {
                const uint32_t ua = s_ == 0 ? mn_alignbyte(ac, ap, 3) : (s_ == 1 ? ac : mn_alignbyte(an, ac, 1));
                const uint32_t ub = s_ == 0 ? mn_alignbyte(bc, bp, 3) : (s_ == 1 ? bc : mn_alignbyte(bn, bc, 1));
                if (XENC == 0) {
                    bf[s_] = u32x4{mn_perm(0u, ua, 0x010c000cu), mn_perm(0u, ua, 0x030c020cu), mn_perm(0u, ub, 0x010c000cu), mn_perm(0u, ub, 0x030c020cu)};
                } else if (XENC == 2) {
                    bf[s_] = u32x4{mn_pack_hi16((float)(ua & 0xffu), (float)((ua >> 8) & 0xffu)), mn_pack_hi16((float)((ua >> 16) & 0xffu), (float)(ua >> 24)),
                                   mn_pack_hi16((float)(ub & 0xffu), (float)((ub >> 8) & 0xffu)), mn_pack_hi16((float)((ub >> 16) & 0xffu), (float)(ub >> 24))};
                } else {          // bf16(j), j = 0..7: 0000 3F80 4000 4040 4080 40A0 40C0 40E0
                    const uint32_t ha = mn_perm(0x40404040u, 0x40403F00u, ua), la = mn_perm(0xE0C0A080u, 0x40008000u, ua);
                    const uint32_t hb = mn_perm(0x40404040u, 0x40403F00u, ub), lb = mn_perm(0xE0C0A080u, 0x40008000u, ub);
                    bf[s_] = u32x4{mn_perm(ha, la, 0x05010400u), mn_perm(ha, la, 0x07030602u), mn_perm(hb, lb, 0x05010400u), mn_perm(hb, lb, 0x07030602u)};
                }
            }
            // term-outer over the three taps of this kernel row: 6 independent accumulators between two MFMAs on the same one
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) acc[mi][r * 3 + s_] = mn_mfma_bf16(a0[mi], bf[s_], acc[mi][r * 3 + s_]);
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) acc[mi][r * 3 + s_] = mn_mfma_bf16(a1[mi], bf[s_], acc[mi][r * 3 + s_]);
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) acc[mi][r * 3 + s_] = mn_mfma_bf16(a2[mi], bf[s_], acc[mi][r * 3 + s_]);
        }
    };
    fetch(s0, 0);
    fetch(s1, 1);
    MN_WAVE_SYNC();                       // the zero dwords are in place
    for (int t = 0; t < nw; t += 2) {
        commit(s0, true);
        fetch(s0, t + 2);
        MN_WAVE_SYNC();
        contract();
        MN_WAVE_SYNC();
        commit(s1, t + 1 < nw);
        fetch(s1, t + 3);
        MN_WAVE_SYNC();
        if (t + 1 < nw) contract();
        MN_WAVE_SYNC();
    }
    // ---- the four waves add their tiles through LDS in wave order, then one coalesced partial-tile write
    float* red = smem;                    // [32][144]
    float* dbs = smem + 32 * 144;         // [32]
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int idx = (mi * 16 + kg * 4 + r) * 144 + j * 9 + t;
                        red[idx] = (w == 0) ? acc[mi][t][r] : red[idx] + acc[mi][t][r];
                    }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = dbacc[i];
                v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 1, 64);
                if (sq == 0) dbs[sr + 8 * i] = (w == 0) ? v : dbs[sr + 8 * i] + v;
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < 32 * 144; i += 256) {
        const int m = i / 144, col = i - m * 144;
        p.part[(((int64_t)z * p.G + g) * p.Mgw + mb * 32 + m) * ((int64_t)p.Cgw * 9) + (int64_t)cb * 144 + col] = red[i];
    }
    if (p.want_db && cb == 0 && tid < 32) p.dbpart[((int64_t)z * p.G + g) * p.Mgw + mb * 32 + tid] = dbs[tid];
}

struct K3wPlan { K3wParams p; int grid; size_t lds; int64_t off_db, ws_bytes; };
static int plan_k3s(const mn_conv_geom* g, K3wPlan* pl) {
    if (g->KH != 3 || g->KW != 3 || g->stride_h != 1 || g->stride_w != 1 || g->pad_h != 1 || g->pad_w != 1 || g->dil_h != 1 || g->dil_w != 1) return 0;
    if (g->W != 8 && g->W != 16 && g->W != 32) return 0;
    const int R = 32 / g->W;
    if (g->H % R) return 0;
    if (g->in_shuffle > 1 && g->C % g->in_shuffle) return 0;
    const int64_t HW = (int64_t)g->H * g->W;
    if ((int64_t)g->N * g->O * HW >= ((int64_t)1 << 31) || (int64_t)g->N * g->C * HW >= ((int64_t)1 << 31)) return 0;     // 32-bit element offsets
    K3wParams& p = pl->p;
    p.N = g->N; p.C = g->C; p.H = g->H; p.W = g->W; p.O = g->O; p.G = g->groups; p.Cg = g->C / g->groups; p.Mg = g->O / g->groups;
    p.in_map = make_chanmap(g->in_shuffle, g->C);
    p.R = R; p.spi = g->H / R; p.nsteps = g->N * p.spi;
    p.W4 = g->W / 4; p.DPC = (R + 2) * p.W4; p.CS = (R + 2) * (g->W + 4) + 4; p.XB = (16 * p.CS + 15) / 16 * 16;
    p.nmb = (p.Mg + 31) / 32; p.ncb = (p.Cg + 15) / 16; p.Mgw = p.nmb * 32; p.Cgw = p.ncb * 16;
    const int base = p.G * p.nmb * p.ncb;
    int Z = 512 / base;
    while (Z > 1 && p.nsteps / Z < 32 && base * Z > 256) Z /= 2;
    if (Z > p.nsteps / 4) Z = p.nsteps / 4;
    if (Z < 1) Z = 1;
    p.Z = Z;
    p.st_per_z = (p.nsteps + Z - 1) / Z;
    p.fd_spi = make_fastdiv((uint32_t)p.spi); p.fd_dpc = make_fastdiv((uint32_t)p.DPC); p.fd_w4 = make_fastdiv((uint32_t)p.W4);
    const int64_t nb = (int64_t)base * Z;
    if (nb > 0x7fffffff) return 0;
    pl->grid = (int)nb;
    size_t lds = (size_t)4 * (32 * K3_RSA + p.XB), red = (size_t)(32 * 144 + 32) * 4;
    pl->lds = lds > red ? lds : red;
    const int64_t part_bytes = (int64_t)Z * p.G * p.Mgw * p.Cgw * 9 * 4;
    pl->off_db = (part_bytes + 255) / 256 * 256;
    pl->ws_bytes = pl->off_db + (int64_t)Z * p.G * p.Mgw * 4;
    return 1;
}
int k3s_wgrad_supported(const mn_conv_geom* g) { K3wPlan pl; return plan_k3s(g, &pl); }
int64_t k3s_wgrad_ws_bytes(const mn_conv_geom* g) { K3wPlan pl; return plan_k3s(g, &pl) ? pl.ws_bytes : 0; }
int k3s_bwd_weight(const mn_conv_geom* g, const float* gy, const int8_t* x, float* dw, float* dbias, void* ws, int64_t ws_bytes, hipStream_t s) {
    K3wPlan pl;
    if (!plan_k3s(g, &pl) || !aligned16(gy) || (((uintptr_t)x) & 3)) MN_FAIL(MN_ENOTSUP, "mn_conv2d_bwd_weight(sign 3x3): geometry not covered");
    if (!ws || ws_bytes < pl.ws_bytes || !aligned16(ws)) MN_FAIL(MN_ENOSPC, "mn_conv2d_bwd_weight(sign 3x3): workspace too small");
    K3wParams& p = pl.p;
    p.gy = gy; p.x = (const char*)x; p.part = (float*)ws; p.dbpart = (float*)((char*)ws + pl.off_db); p.want_db = dbias != nullptr;
    mn_set_last_kernel("k_k3s_wgrad");
    { const double nx = (double)g->N * g->C * g->H * g->W, ny = (double)g->N * g->O * g->H * g->W; mn_prof_bytes(4.0 * ny + nx); }
    mn_prof_begin(s);
    raise_lds_limit((const void*)k_k3s_wgrad<0>, pl.lds);
    hipLaunchKernelGGL(k_k3s_wgrad<0>, dim3(pl.grid), dim3(256), pl.lds, s, p);
    mn_prof_end(s);
    // the codes were contracted as +-0.5: the reduction doubles (exact)
    qg_launch_wgrad_reduce(p.part, p.dbpart, dw, dbias, p.Z, p.G, p.Mg, p.Cg * 9, p.Mgw, p.Cgw * 9, 2.f, nullptr, s);
    MN_CHECK_LAUNCH("mn_conv2d_bwd_weight(sign 3x3)");
    return MN_OK;
}
// 3 x 3 backward-weight on k-bit activation codes (bytes): dw = s * sum gy * j;  a_bits <= 3: the 8-entry look-up (XENC 1), else the conversion (XENC 2)
int k3s_wgrad_code8_supported(const mn_conv_geom* g, int a_bits) { K3wPlan pl; return a_bits >= 2 && a_bits <= 8 && plan_k3s(g, &pl); }
int k3s_bwd_weight_code8(const mn_conv_geom* g, const float* gy, const uint8_t* x, int a_bits, float ascale, float* dw, float* dbias, void* ws, int64_t ws_bytes, hipStream_t s) {
    K3wPlan pl;
    if (!plan_k3s(g, &pl) || !aligned16(gy) || (((uintptr_t)x) & 3)) MN_FAIL(MN_ENOTSUP, "mn_conv2d_bwd_weight(code8 3x3): geometry not covered");
    if (!ws || ws_bytes < pl.ws_bytes || !aligned16(ws)) MN_FAIL(MN_ENOSPC, "mn_conv2d_bwd_weight(code8 3x3): workspace too small");
    K3wParams& p = pl.p;
    p.gy = gy; p.x = (const char*)x; p.part = (float*)ws; p.dbpart = (float*)((char*)ws + pl.off_db); p.want_db = dbias != nullptr;
    const bool lut = a_bits <= 3;
    mn_set_last_kernel(lut ? "k_k3s_wgrad<1>" : "k_k3s_wgrad<2>");
    { const double nx = (double)g->N * g->C * g->H * g->W, ny = (double)g->N * g->O * g->H * g->W; mn_prof_bytes(4.0 * ny + nx); }
    mn_prof_begin(s);
    if (lut) {
        raise_lds_limit((const void*)k_k3s_wgrad<1>, pl.lds);
        hipLaunchKernelGGL(k_k3s_wgrad<1>, dim3(pl.grid), dim3(256), pl.lds, s, p);
    } else {
        raise_lds_limit((const void*)k_k3s_wgrad<2>, pl.lds);
        hipLaunchKernelGGL(k_k3s_wgrad<2>, dim3(pl.grid), dim3(256), pl.lds, s, p);
    }
    mn_prof_end(s);
    qg_launch_wgrad_reduce(p.part, p.dbpart, dw, dbias, p.Z, p.G, p.Mg, p.Cg * 9, p.Mgw, p.Cgw * 9, ascale, nullptr, s);
    MN_CHECK_LAUNCH("mn_conv2d_bwd_weight(code8 3x3)");
    return MN_OK;
}

// ================================================================================================
// 3 x 3 (stride 1, padding 1) backward-data of a ternary / binary-weight layer:
//
//   dx[n][g*Cg + c][ih][iw] = sum over (m, r, s) of  t[m][c][r][s] * (alpha[m] * gy[n][g*Mg + m][ih + 1 - r][iw + 1 - s])
//
// (w = alpha[m] * t with t in {-1, 0, +1}: wbwtab/quantize.py:96-150; autograd's conv2d backward-data.)  The contraction runs over the
// gy channels m, but memory is pixel-contiguous: a B fragment (8 channels of ONE pixel per lane) needs gy transposed.  k_kk does that
// with 2-byte LDS stores (12 per float4).  Here a block stages a whole stage (NI images of one group) once:
//   * 256 threads read the 32 gy rows coalesced (a wave = 1 KB of one row per load), scale by alpha[m], split into three exact bf16
//     terms and write them TRANSPOSED as [pixel slot][32 k] with 8-byte stores: the contraction order k is a permutation of m
//     (k = 4 (m % 8) + m / 8) chosen so that the four rows a thread holds are adjacent in k;
//   * the image sits in LDS with a one-pixel zero frame ((H + 2) x (W + 2) slots, rows padded to 80 B): the nine taps of an output
//     pixel are nine constant slot offsets, border handling costs nothing;
//   * each wave owns 32-pixel tiles of the stage: per tile 2 (16-pixel halves) x 3 (terms) independent accumulators, 9 MFMAs each;
//     the weight codes of the nine taps are A fragments held in registers for the whole kernel (same k permutation);
//   * the next stage's gy rows are loaded into registers while the current one is contracted.
// D[row = channel 4 kg + r][col = pixel j]: 16 lanes store 64 contiguous bytes of one dx row.
#define K3D_RS 80             // LDS bytes per pixel slot and term (32 bf16 + pad: b128 reads of 16 consecutive slots at most 2-way)
#define K3D_MAXF4 8           // float4 a thread stages per stage: 32 rows x SP pixels / 256 threads / 4, SP <= 256

struct K3dParams {
    const float* gy;
    const float* w;           // fake-quantised weights [O][Cg][3][3]
    float* dx;
    float wn;                 // 0: ternary / binary weights (code = sign, scale = max |w| of the row);  n = 2^w_bits - 1: DoReFa weights (2k - n) / n
                              // (wqaq/dorefa/quantize.py:68-72): code = rint(w * n), scale = 1 / n
    int N, C, H, W, O, Cg, Mg, G, ncb, Zb;
    int NI, SP, HW, IS, TS, nstages, WP;      // images per stage, pixels per stage, slots per image, bytes per term plane, W + 2
    FastDiv fd_hw4, fd_w4, fd_w;
    ChanMap out_map;
};

__global__ __launch_bounds__(256, 2) void k_k3s_dgrad(const K3dParams p) {
    HIP_DYNAMIC_SHARED(float, smem)
    unsigned char* lds = reinterpret_cast<unsigned char*>(smem);
    float* alpha = reinterpret_cast<float*>(lds + 3 * p.TS);          // [32]
    const int tid = threadIdx.x, lane = tid & 63, wave = mn_uniform(tid >> 6), j = lane & 15, kg = lane >> 4;
    uint32_t b = blockIdx.x;
    const int z = b % p.Zb; b /= p.Zb;
    const int cb = b % p.ncb;
    const int g = b / p.ncb;
    const int wrow = p.Cg * 9;                                         // floats per weight row

    // the group's weights go through LDS first (coalesced; the fragment build below reads them 72 times per lane), then the image region is
    // zeroed: the zero frame stays for the whole kernel.  alpha[m] = max |w[m][..]| (every non-zero weight of a ternary / binary row)
    float* wl = reinterpret_cast<float*>(lds);
    const int nwl = p.Mg * wrow;
    for (int i = tid; i < nwl; i += 256) wl[i] = p.w[(int64_t)g * p.Mg * wrow + i];
    __syncthreads();
    float my_alpha = 0.f;
    {
        const int m = tid >> 3, part = tid & 7;
        float a = 0.f;
        if (m < p.Mg) for (int k = part; k < wrow; k += 8) a = fmaxf(a, fabsf(wl[m * wrow + k]));
        a = fmaxf(a, __shfl_xor(a, 4, 64)); a = fmaxf(a, __shfl_xor(a, 2, 64)); a = fmaxf(a, __shfl_xor(a, 1, 64));
        my_alpha = p.wn > 0.f ? 1.0f / p.wn : a;
    }
    // A fragments: code t[m(k)][c = cb*16 + j][tap] for k = 8 kg + e  ->  m = 2 kg + e / 4 + 8 (e % 4)
    u32x4 wa[9];
    {
        const int c = cb * 16 + j;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            uint32_t h16[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int m = 2 * kg + (e >> 2) + 8 * (e & 3);
                float v = 0.f;
                if (m < p.Mg && c < p.Cg) v = wl[m * wrow + c * 9 + t];
                h16[e] = p.wn > 0.f ? (mn_f2u(rintf(v * p.wn)) >> 16) : (v > 0.f ? 0x3F80u : (v < 0.f ? 0xBF80u : 0u));
            }
            wa[t] = u32x4{h16[0] | (h16[1] << 16), h16[2] | (h16[3] << 16), h16[4] | (h16[5] << 16), h16[6] | (h16[7] << 16)};
        }
    }
    __syncthreads();                                                   // everybody is done with the weight image
    for (int i = tid; i < (3 * p.TS) / 16; i += 256) *reinterpret_cast<u32x4*>(lds + 16 * i) = u32x4{0u, 0u, 0u, 0u};
    if ((tid & 7) == 0) alpha[tid >> 3] = my_alpha;
    // staging roles: pair u of this thread = (chunk, sr): rows m = sr + 8 i, pixels 4 chunk .. of the stage
    const int cps = p.SP >> 2;                                         // chunks per stage
    const int npair = (p.SP * 2) >> 8;                                 // pairs per thread (1 or 2)
    int s_slot[2];                                                     // LDS byte offset of the pair's first pixel slot (+ 8 sr)
    uint32_t s_goff[2][4];                                             // element offset inside the stage's first image plane set
    float s_al[2][4];
    __syncthreads();                                                   // alpha visible
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int pi = tid + 256 * u;
        const int chunk = pi % cps, sr = pi / cps;                     // sr < 8 when u < npair
        const uint32_t img = fd_div(chunk, p.fd_hw4);
        const int q = chunk - (int)img * (p.HW >> 2);
        const uint32_t row = fd_div(q, p.fd_w4);
        const int col = (q - (int)row * (p.W >> 2)) * 4;
        s_slot[u] = ((int)img * p.IS + ((int)row + 1) * p.WP + col + 1) * K3D_RS + 8 * (sr & 7);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int m = (sr & 7) + 8 * i;
            s_al[u][i] = m < p.Mg ? alpha[m] : 0.f;
            m = m < p.Mg ? m : p.Mg - 1;
            s_goff[u][i] = (img * (uint32_t)p.O + (uint32_t)(g * p.Mg + m)) * (uint32_t)p.HW + 4u * q;
        }
    }
    float4 rg[2][4];
    auto fetch = [&](int st) {                                         // unconditional; images past N are clamped (never stored)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u < npair) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int64_t base = (int64_t)st * p.NI * p.O * p.HW;
                    const int64_t lim = ((int64_t)p.N * p.O - 1) * p.HW + p.HW - 4;          // last valid quad
                    int64_t off = base + s_goff[u][i];
                    off = off < lim ? off : lim;
                    rg[u][i] = *reinterpret_cast<const float4*>(p.gy + off);
                }
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u < npair) {
                float t0[4][4], t1[4][4], t2[4][4];                    // [i][e]
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v[4] = {rg[u][i].x * s_al[u][i], rg[u][i].y * s_al[u][i], rg[u][i].z * s_al[u][i], rg[u][i].w * s_al[u][i]};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        t0[i][e] = mn_bf16_head(v[e]);
                        const float r1 = v[e] - t0[i][e];
                        t1[i][e] = mn_bf16_head(r1);
                        t2[i][e] = r1 - t1[i][e];
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {                          // pixel e of the chunk: k = 4 sr + i, i = 0..3 -> one 8-byte store per term
                    unsigned char* d = lds + s_slot[u] + e * K3D_RS;
                    *reinterpret_cast<u32x2*>(d) = u32x2{mn_pack_bf16x2(t0[0][e], t0[1][e]), mn_pack_bf16x2(t0[2][e], t0[3][e])};
                    *reinterpret_cast<u32x2*>(d + p.TS) = u32x2{mn_pack_bf16x2(t1[0][e], t1[1][e]), mn_pack_bf16x2(t1[2][e], t1[3][e])};
                    *reinterpret_cast<u32x2*>(d + 2 * p.TS) = u32x2{mn_pack_bf16x2(t2[0][e], t2[1][e]), mn_pack_bf16x2(t2[2][e], t2[3][e])};
                }
            }
        }
    };
    const int ntile = p.SP >> 5;                                       // 32-pixel tiles per stage
    int st = z;
    if (st < p.nstages) fetch(st);
    for (; st < p.nstages; st += p.Zb) {
        __syncthreads();                                               // previous stage fully contracted
        commit();
        __syncthreads();
        fetch(st + p.Zb < p.nstages ? st + p.Zb : st);                 // in flight during the MFMAs
        for (int tile = wave; tile < ntile; tile += 4) {
            f32x4 acc[2][3];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int t = 0; t < 3; ++t) acc[hh][t] = f32x4{0.f, 0.f, 0.f, 0.f};
            int sb[2];                                                 // byte offset of pixel (hh, j)'s own slot + this lane's k chunk
            int opix[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int sp = tile * 32 + hh * 16 + j;                // pixel of the stage
                const uint32_t img = fd_div(sp >> 2, p.fd_hw4);
                const int pp = sp - (int)img * p.HW;
                const uint32_t row = fd_div(pp, p.fd_w);
                const int col = pp - (int)row * p.W;
                sb[hh] = ((int)img * p.IS + ((int)row + 1) * p.WP + col + 1) * K3D_RS + 16 * kg;
                opix[hh] = sp;
            }
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int s_ = 0; s_ < 3; ++s_) {
                    const int toff = ((1 - r) * p.WP + (1 - s_)) * K3D_RS;        // input pixel (ih + 1 - r, iw + 1 - s)
#pragma unroll
                    for (int t = 0; t < 3; ++t)
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh) {
                            const u32x4 bf = *reinterpret_cast<const u32x4*>(lds + t * p.TS + sb[hh] + toff);
                            acc[hh][t] = mn_mfma_bf16(wa[r * 3 + s_], bf, acc[hh][t]);
                        }
                }
            // D[row = channel 4 kg + rr][col = pixel j]
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const uint32_t img = fd_div(opix[hh] >> 2, p.fd_hw4);
                const int pp = opix[hh] - (int)img * p.HW;
                const int n = st * p.NI + (int)img;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int c = cb * 16 + 4 * kg + rr;
                    if (n < p.N && c < p.Cg)
                        p.dx[((int64_t)n * p.C + chan_phys(p.out_map, g * p.Cg + c)) * p.HW + pp] = (acc[hh][0][rr] + acc[hh][1][rr]) + acc[hh][2][rr];
                }
            }
        }
    }
}

struct K3dPlan { K3dParams p; int grid; size_t lds; };
static int plan_k3d(const mn_conv_geom* g, const mn_wq* wq, K3dPlan* pl) {
    if (!wq || !(wq->mode == MN_WQ_TERNARY || (wq->mode == MN_WQ_DOREFA && wq->bits >= 2 && wq->bits <= 8))) return 0;
    if (g->KH != 3 || g->KW != 3 || g->stride_h != 1 || g->stride_w != 1 || g->pad_h != 1 || g->pad_w != 1 || g->dil_h != 1 || g->dil_w != 1) return 0;
    if (g->in_shuffle > 1 && g->C % g->in_shuffle) return 0;
    const int HW = g->H * g->W, Mg = g->O / g->groups, Cg = g->C / g->groups;
    if (g->W % 4 || HW % 32 || Mg > 32 || Mg < 1) return 0;
    if ((int64_t)g->N * g->O * HW >= ((int64_t)1 << 31)) return 0;
    K3dParams& p = pl->p;
    int NI = 1;
    while (NI * HW < 128) NI *= 2;
    const int SP = NI * HW;
    if (SP > 256 || SP % 128) return 0;          // a thread stages at most K3D_MAXF4 float4
    p.N = g->N; p.C = g->C; p.H = g->H; p.W = g->W; p.O = g->O; p.Cg = Cg; p.Mg = Mg; p.G = g->groups;
    p.NI = NI; p.SP = SP; p.HW = HW; p.WP = g->W + 2; p.IS = (g->H + 2) * (g->W + 2);
    p.TS = (NI * p.IS * K3D_RS + 255) / 256 * 256;
    pl->lds = (size_t)3 * p.TS + 128;
    if (pl->lds < (size_t)Mg * Cg * 9 * 4) pl->lds = (size_t)Mg * Cg * 9 * 4;          // the prologue's weight image
    if (pl->lds > 80 * 1024) return 0;
    p.nstages = (g->N + NI - 1) / NI;
    p.ncb = (Cg + 15) / 16;
    const int base = p.G * p.ncb;
    int tgt = 512;
    int Zb = tgt / base;
    if (Zb > p.nstages) Zb = p.nstages;
    if (Zb < 1) Zb = 1;
    p.Zb = Zb;
    const int64_t nb = (int64_t)base * Zb;
    if (nb > 0x7fffffff) return 0;
    pl->grid = (int)nb;
    p.fd_hw4 = make_fastdiv((uint32_t)(HW / 4)); p.fd_w4 = make_fastdiv((uint32_t)(g->W / 4)); p.fd_w = make_fastdiv((uint32_t)g->W);
    p.out_map = make_chanmap(g->in_shuffle, g->C);
    return 1;
}
int k3s_dgrad_supported(const mn_conv_geom* g, const mn_wq* wq) { K3dPlan pl; return plan_k3d(g, wq, &pl); }
int k3s_bwd_data(const mn_conv_geom* g, const mn_wq* wq, const float* gy, const float* w, float* dx, hipStream_t s) {
    K3dPlan pl;
    if (!plan_k3d(g, wq, &pl) || !aligned16(gy) || !w || !dx) MN_FAIL(MN_ENOTSUP, "mn_conv2d_bwd_data(3x3 ternary): geometry not covered");
    K3dParams& p = pl.p;
    p.gy = gy; p.w = w; p.dx = dx; p.wn = wq->mode == MN_WQ_DOREFA ? (float)((1ll << wq->bits) - 1) : 0.f;
    mn_set_last_kernel("k_k3s_dgrad");
    { const double nx = (double)g->N * g->C * g->H * g->W, ny = (double)g->N * g->O * g->H * g->W; mn_prof_bytes(4.0 * ny + 4.0 * nx); }
    mn_prof_begin(s);
    raise_lds_limit((const void*)k_k3s_dgrad, pl.lds);
    hipLaunchKernelGGL(k_k3s_dgrad, dim3(pl.grid), dim3(256), pl.lds, s, p);
    mn_prof_end(s);
    MN_CHECK_LAUNCH("mn_conv2d_bwd_data(3x3 ternary)");
    return MN_OK;
}

// ================================================================================================
// 3 x 3 (stride 1, padding 1) FORWARD of a ternary / binary-weight layer on sign codes, writing the byte stash:
//
//   acc[n][g*Mg + m][oh][ow] = sum over (c, r, s) of t[m][c][r][s] * a[n][g*Cg + c][oh + r - 1][ow + s - 1]      (exact integer)
//   h = (acc + nnz[class(oh, ow)][m]) / 2 as one byte;   per-channel partial sums of acc and acc^2 (the BatchNorm batch statistics)
//
// (wbwtab/quantize.py:181-195 with binary activations; the block's BatchNorm + sign then only need h: mn_qconv_bnsign_fwd_stash.)
// Same organisation as k_k3s_dgrad: a block stages NI images of one group in LDS with a zero frame -- here the codes themselves,
// transposed to [pixel slot][16 c] as bf16 +-0.5 (sign byte over 0x3F as the high byte; one v_perm per dword; the weight codes are
// +-2) under the channel permutation k = 4 (c % 4) + c / 4 that makes a thread's four channels adjacent (8-byte stores); a K-step is two
// taps x 16 channels, 5 K-steps cover the 9 taps; the weights are B fragments in registers for the whole kernel.  M = pixels:
// D[row = pixel 4 kg + r][col = channel j] leaves every lane with 4 consecutive pixels of one channel = one dword of h.  The statistics
// are accumulated in integer registers over the block's stages and leave as one fp64 partial per (block, channel) in the layout
// k_pws_stats_prep reads.
#define K3F_RS 48             // LDS bytes per pixel slot (16 bf16 + pad)

struct K3fParams {
    const char* x;
    const float* w;           // fake-quantised weights [O][Cg][3][3]
    const float* nnz9;        // [9][O]
    unsigned char* h;
    int16_t* h16;             // XENC 1: the 16-bit stash of acc
    int32_t* h32;             // XENC 2: the 32-bit stash of acc
    float wn;                 // XENC 1: DoReFa weights (2k - n) / n: the integer code is rint(w * wn), wn = 2^w_bits - 1
    double* part;             // [Zb][G*Mg][2]
    int N, C, H, W, O, Cg, Mg, G, Zb;
    int NI, SP, HW, IS, TS, nstages, WP;
    FastDiv fd_hw4, fd_w4, fd_w;
    ChanMap in_map;
};

// XENC 0: sign codes x ternary weights -> byte stash h (above).  XENC 1: k-bit activation codes j in [0, 127] x DoReFa weight codes: the LDS image
// holds bf16 128 + j (the frame and the padding slots hold 128 = code 0), acc' - 128 * sum of the row's weight codes is the exact integer acc,
// stored as int16; statistics in 64-bit integers.  XENC 2: codes j in [0, 255] (or an accumulator beyond int16): the image holds bf16 j itself (the
// staging threads convert: v_cvt_f32_ubyte, high half), frame and padding are 0, no row constant, |acc| < 2^24 (planner) leaves as a 32-bit stash.
template <int XENC>
__global__ __launch_bounds__(256, 2) void k_k3s_fwd(const K3fParams p) {
    HIP_DYNAMIC_SHARED(float, smem)
    unsigned char* lds = reinterpret_cast<unsigned char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = mn_uniform(tid >> 6), j = lane & 15, kg = lane >> 4;
    uint32_t b = blockIdx.x;
    const int z = b % p.Zb;
    const int g = b / p.Zb;
    const int wrow = p.Cg * 9;

    // the group's weights through LDS (coalesced), fragments from there; then the image region is zeroed (the zero frame stays)
    float* wl = reinterpret_cast<float*>(lds);
    for (int i = tid; i < p.Mg * wrow; i += 256) wl[i] = p.w[(int64_t)g * p.Mg * wrow + i];
    __syncthreads();
    // B fragments: 2 * t[m = mt*16 + j][c(k)][tap(ks, kg)] for k = 8 kg + e: tap = 2 ks + (kg >> 1), position 8 (kg & 1) + e -> c = (pos >> 2) + 4 (pos & 3)
    u32x4 wb[5][2];
    StashNnz zn[2];
    float wsum[2] = {0.f, 0.f};          // XENC 1: 128 * sum of the channel's weight codes
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = mt * 16 + j;
        const int co = g * p.Mg + (m < p.Mg ? m : p.Mg - 1);
        if (XENC == 0) {
            zn[mt].v0 = p.nnz9[co]; zn[mt].v1 = p.nnz9[p.O + co]; zn[mt].v2 = p.nnz9[2 * p.O + co]; zn[mt].v3 = p.nnz9[3 * p.O + co];
            zn[mt].v4 = p.nnz9[4 * p.O + co]; zn[mt].v5 = p.nnz9[5 * p.O + co]; zn[mt].v6 = p.nnz9[6 * p.O + co]; zn[mt].v7 = p.nnz9[7 * p.O + co];
            zn[mt].v8 = p.nnz9[8 * p.O + co];
        } else if (XENC == 1 && m < p.Mg) {
            float sm = 0.f;
            for (int i = 0; i < wrow; ++i) sm += rintf(wl[m * wrow + i] * p.wn);
            wsum[mt] = 128.f * sm;
        }
#pragma unroll
        for (int ks = 0; ks < 5; ++ks) {
            const int tap = 2 * ks + (kg >> 1);
            uint32_t h16[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int pos = 8 * (kg & 1) + e, c = (pos >> 2) + 4 * (pos & 3);
                float v = 0.f;
                if (m < p.Mg && c < p.Cg && tap < 9) v = wl[m * wrow + c * 9 + tap];
                if (XENC == 0) h16[e] = v > 0.f ? 0x4000u : (v < 0.f ? 0xC000u : 0u);
                else h16[e] = mn_f2u(rintf(v * p.wn)) >> 16;               // an integer |code| <= 255: exact in bf16
            }
            wb[ks][mt] = u32x4{h16[0] | (h16[1] << 16), h16[2] | (h16[3] << 16), h16[4] | (h16[5] << 16), h16[6] | (h16[7] << 16)};
        }
    }
    __syncthreads();
    {
        const uint32_t fill = XENC == 1 ? 0x43004300u : 0u;          // code 0 (the zero padding) is bf16 128 under XENC 1
        for (int i = tid; i < p.TS / 16; i += 256) *reinterpret_cast<u32x4*>(lds + 16 * i) = u32x4{fill, fill, fill, fill};
    }
    // tap offset (bytes) of this lane's half of every K-step; tap 9 (the padding half of the last step) reads tap 8's slot against zero weights
    int toff[5];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        int tap = 2 * ks + (kg >> 1);
        tap = tap < 9 ? tap : 8;
        const int r = tap / 3, s_ = tap - 3 * r;
        toff[ks] = ((r - 1) * p.WP + (s_ - 1)) * K3F_RS + 16 * (kg & 1);
    }
    // staging role: chunk (4 pixels) x channel quartet sr: channels sr + 4 i
    const int cps = p.SP >> 2;
    const bool sact = tid < cps * 4;
    const int chunk = tid % cps, sr = (tid / cps) & 3;
    int s_slot;
    uint32_t s_goff[4];
    bool s_cv[4];
    {
        const uint32_t img = fd_div(chunk, p.fd_hw4);
        const int q = chunk - (int)img * (p.HW >> 2);
        const uint32_t row = fd_div(q, p.fd_w4);
        const int col = (q - (int)row * (p.W >> 2)) * 4;
        s_slot = ((int)img * p.IS + ((int)row + 1) * p.WP + col + 1) * K3F_RS + 8 * sr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = sr + 4 * i;
            s_cv[i] = c < p.Cg;
            s_goff[i] = (img * (uint32_t)p.C + (uint32_t)chan_phys(p.in_map, g * p.Cg + (s_cv[i] ? c : p.Cg - 1))) * (uint32_t)p.HW + 4u * q;
        }
    }
    uint32_t rg[4];
    auto fetch = [&](int st) {
        const int64_t lim = ((int64_t)p.N * p.C - 1) * p.HW + p.HW - 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int64_t off = (int64_t)st * p.NI * p.C * p.HW + s_goff[i];
            off = off < lim ? off : lim;
            rg[i] = *reinterpret_cast<const uint32_t*>(p.x + off);
        }
    };
    auto commit = [&]() {
        if (!sact) return;
        uint32_t en[4];
        if (XENC == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) en[i] = s_cv[i] ? ((rg[i] & 0x80808080u) | 0x3F3F3F3Fu) : 0u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t se = 0x000c000cu | ((uint32_t)e << 8) | ((uint32_t)(4 + e) << 24);      // [0, b.byte e, 0, a.byte e]
                *reinterpret_cast<u32x2*>(lds + s_slot + e * K3F_RS) = u32x2{mn_perm(en[1], en[0], se), mn_perm(en[3], en[2], se)};
            }
        } else if (XENC == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) en[i] = s_cv[i] ? rg[i] : 0u;
#pragma unroll
            for (int e = 0; e < 4; ++e)          // slot of pixel e: channels sr, sr + 4 | sr + 8, sr + 12 as bf16 j
                *reinterpret_cast<u32x2*>(lds + s_slot + e * K3F_RS) = u32x2{mn_pack_hi16((float)((en[0] >> (8 * e)) & 0xffu), (float)((en[1] >> (8 * e)) & 0xffu)),
                                                                             mn_pack_hi16((float)((en[2] >> (8 * e)) & 0xffu), (float)((en[3] >> (8 * e)) & 0xffu))};
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) en[i] = s_cv[i] ? rg[i] : 0u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t se = 0x0c000c00u | (uint32_t)e | ((uint32_t)(4 + e) << 16);            // [b.byte e, 0, a.byte e, 0], then the 0x43 high bytes
                *reinterpret_cast<u32x2*>(lds + s_slot + e * K3F_RS) = u32x2{mn_perm(en[1], en[0], se) | 0x43004300u, mn_perm(en[3], en[2], se) | 0x43004300u};
            }
        }
    };
    long long s1[2] = {0, 0}, s2[2] = {0, 0};
    const int ntile = p.SP >> 5;
    int st = z;
    __syncthreads();
    if (st < p.nstages) fetch(st);
    for (; st < p.nstages; st += p.Zb) {
        __syncthreads();
        commit();
        __syncthreads();
        fetch(st + p.Zb < p.nstages ? st + p.Zb : st);
        for (int tile = wave; tile < ntile; tile += 4) {
            f32x4 acc[2][2];
#pragma unroll
            for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[pt][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            int sb[2];
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const int sp = tile * 32 + pt * 16 + j;
                const uint32_t img = fd_div(sp >> 2, p.fd_hw4);
                const int pp = sp - (int)img * p.HW;
                const uint32_t row = fd_div(pp, p.fd_w);
                sb[pt] = ((int)img * p.IS + ((int)row + 1) * p.WP + (pp - (int)row * p.W) + 1) * K3F_RS;
            }
#pragma unroll
            for (int ks = 0; ks < 5; ++ks)
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) {
                    const u32x4 af = *reinterpret_cast<const u32x4*>(lds + sb[pt] + toff[ks]);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) acc[pt][mt] = mn_mfma_bf16(af, wb[ks][mt], acc[pt][mt]);
                }
            // D[row = pixel 4 kg + rr][col = channel j]: this lane's quad = stage pixels tile*32 + pt*16 + 4 kg .. + 3
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const int sp0 = tile * 32 + pt * 16 + 4 * kg;
                const uint32_t img = fd_div(sp0 >> 2, p.fd_hw4);
                const int pp = sp0 - (int)img * p.HW;
                const uint32_t row = fd_div(pp, p.fd_w);
                const int col4 = (pp - (int)row * p.W) >> 2;
                const int n = st * p.NI + (int)img;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int m = mt * 16 + j;
                    if (n < p.N && m < p.Mg) {
                        if (XENC == 0) {
                            float nz[4];
                            stash_nnz_quad(zn[mt], (int)row, col4, p.H, p.W >> 2, nz);
                            uint32_t hb = 0u;
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr) {
                                const float a = acc[pt][mt][rr];
                                hb |= (uint32_t)((a + nz[rr]) * 0.5f) << (8 * rr);
                                const int ai = (int)a;
                                s1[mt] += ai; s2[mt] += ai * ai;
                            }
                            *reinterpret_cast<uint32_t*>(p.h + ((int64_t)n * p.O + g * p.Mg + m) * p.HW + pp) = hb;
                        } else {
                            int ai[4];
#pragma unroll
                            for (int rr = 0; rr < 4; ++rr) {
                                ai[rr] = (int)(acc[pt][mt][rr] - wsum[mt]);
                                s1[mt] += ai[rr]; s2[mt] += (long long)ai[rr] * ai[rr];
                            }
                            if (XENC == 2) *reinterpret_cast<u32x4*>(p.h32 + ((int64_t)n * p.O + g * p.Mg + m) * p.HW + pp) = u32x4{(uint32_t)ai[0], (uint32_t)ai[1], (uint32_t)ai[2], (uint32_t)ai[3]};
                            else *reinterpret_cast<u32x2*>(p.h16 + ((int64_t)n * p.O + g * p.Mg + m) * p.HW + pp) =
                                u32x2{((uint32_t)ai[0] & 0xffffu) | ((uint32_t)ai[1] << 16), ((uint32_t)ai[2] & 0xffffu) | ((uint32_t)ai[3] << 16)};
                        }
                    }
                }
            }
        }
    }
    // statistics: the four pixel groups of a wave by shuffles, the four waves through LDS in wave order
    __syncthreads();
    long long* red = reinterpret_cast<long long*>(lds);          // [4 waves][2 mt][16][2]
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        long long a1 = s1[mt], a2 = s2[mt];
        a1 += __shfl_xor(a1, 16, 64); a1 += __shfl_xor(a1, 32, 64);
        a2 += __shfl_xor(a2, 16, 64); a2 += __shfl_xor(a2, 32, 64);
        if (kg == 0) { red[((wave * 2 + mt) * 16 + j) * 2] = a1; red[((wave * 2 + mt) * 16 + j) * 2 + 1] = a2; }
    }
    __syncthreads();
    if (tid < 32) {
        const int mt = tid >> 4, jj = tid & 15, m = mt * 16 + jj;
        if (m < p.Mg) {
            long long t1 = 0, t2 = 0;
            for (int w_ = 0; w_ < 4; ++w_) { t1 += red[((w_ * 2 + mt) * 16 + jj) * 2]; t2 += red[((w_ * 2 + mt) * 16 + jj) * 2 + 1]; }
            double* dst = p.part + ((int64_t)z * p.G * p.Mg + g * p.Mg + m) * 2;
            dst[0] = (double)t1; dst[1] = (double)t2;
        }
    }
}

struct K3fPlan { K3fParams p; int grid; size_t lds; };
static int plan_k3f(const mn_conv_geom* g, const mn_wq* wq, K3fPlan* pl, int xenc = 0) {
    if (!wq || (xenc ? wq->mode != MN_WQ_DOREFA : wq->mode != MN_WQ_TERNARY)) return 0;
    if (g->KH != 3 || g->KW != 3 || g->stride_h != 1 || g->stride_w != 1 || g->pad_h != 1 || g->pad_w != 1 || g->dil_h != 1 || g->dil_w != 1) return 0;
    if (g->in_shuffle > 1 && g->C % g->in_shuffle) return 0;
    const int HW = g->H * g->W, Mg = g->O / g->groups, Cg = g->C / g->groups;
    if (g->W % 4 || HW % 32 || g->H < 2 || Mg > 32 || Cg > 16) return 0;
    if ((int64_t)g->N * g->O * HW >= ((int64_t)1 << 31) || (int64_t)g->N * g->C * HW >= ((int64_t)1 << 31)) return 0;
    K3fParams& p = pl->p;
    int NI = 1;
    while (NI * HW < 128) NI *= 2;
    const int SP = NI * HW;
    if (SP > 256 || SP % 128) return 0;
    p.N = g->N; p.C = g->C; p.H = g->H; p.W = g->W; p.O = g->O; p.Cg = Cg; p.Mg = Mg; p.G = g->groups;
    p.NI = NI; p.SP = SP; p.HW = HW; p.WP = g->W + 2; p.IS = (g->H + 2) * (g->W + 2);
    p.TS = (NI * p.IS * K3F_RS + 255) / 256 * 256;
    pl->lds = (size_t)p.TS > 1024 ? (size_t)p.TS : 1024;
    if (pl->lds < (size_t)Mg * Cg * 9 * 4) pl->lds = (size_t)Mg * Cg * 9 * 4;          // the prologue's weight image
    if (pl->lds > 64 * 1024) return 0;
    p.nstages = (g->N + NI - 1) / NI;
    int tgt = 512;
    int Zb = tgt / p.G;
    if (Zb > p.nstages) Zb = p.nstages;
    if (Zb < 1) Zb = 1;
    p.Zb = Zb;
    const int64_t nb = (int64_t)p.G * Zb;
    if (nb > 0x7fffffff) return 0;
    pl->grid = (int)nb;
    p.fd_hw4 = make_fastdiv((uint32_t)(HW / 4)); p.fd_w4 = make_fastdiv((uint32_t)(g->W / 4)); p.fd_w = make_fastdiv((uint32_t)g->W);
    p.in_map = make_chanmap(g->in_shuffle, g->C);
    return 1;
}
int k3s_fwd_supported(const mn_conv_geom* g, const mn_wq* wq) { K3fPlan pl; return plan_k3f(g, wq, &pl); }
int k3s_fwd_parts(const mn_conv_geom* g, const mn_wq* wq) { K3fPlan pl; return plan_k3f(g, wq, &pl) ? pl.p.Zb : 0; }     // partial rows per channel
int k3s_fwd_h8(const mn_conv_geom* g, const mn_wq* wq, const int8_t* x, const float* w, const float* nnz9, uint8_t* h, double* part, hipStream_t s) {
    K3fPlan pl;
    if (!plan_k3f(g, wq, &pl) || (((uintptr_t)x) & 3) || (((uintptr_t)h) & 3) || !w || !nnz9 || !part) MN_FAIL(MN_ENOTSUP, "mn_qconv_bnsign_fwd_stash(3x3): geometry not covered");
    K3fParams& p = pl.p;
    p.x = (const char*)x; p.w = w; p.nnz9 = nnz9; p.h = h; p.h16 = nullptr; p.h32 = nullptr; p.wn = 1.f; p.part = part;
    mn_set_last_kernel("k_k3s_fwd");
    { const double nx = (double)g->N * g->C * g->H * g->W, ny = (double)g->N * g->O * g->H * g->W; mn_prof_bytes(nx + ny); }
    mn_prof_begin(s);
    raise_lds_limit((const void*)k_k3s_fwd<0>, pl.lds);
    hipLaunchKernelGGL(k_k3s_fwd<0>, dim3(pl.grid), dim3(256), pl.lds, s, p);
    mn_prof_end(s);
    MN_CHECK_LAUNCH("mn_qconv_bnsign_fwd_stash(3x3)");
    return MN_OK;
}
// the same 3 x 3 forward on k-bit activation codes x DoReFa weights, writing the 16-bit stash of acc + the statistics partials
int k3s_fwd16_supported(const mn_conv_geom* g, const mn_wq* wq) { K3fPlan pl; return plan_k3f(g, wq, &pl, 1) && pl.lds >= 4 * 2 * 16 * 2 * 8; }
int k3s_fwd16_parts(const mn_conv_geom* g, const mn_wq* wq) { K3fPlan pl; return plan_k3f(g, wq, &pl, 1) ? pl.p.Zb : 0; }
int k3s_fwd_h16(const mn_conv_geom* g, const mn_wq* wq, const uint8_t* x, const float* w, int16_t* h16, int wide, double* part, hipStream_t s) {
    K3fPlan pl;
    if (!plan_k3f(g, wq, &pl, 1) || (((uintptr_t)x) & 3) || (((uintptr_t)h16) & (wide ? 15 : 7)) || !w || !part) MN_FAIL(MN_ENOTSUP, "mn_qconv_bnq_fwd_stash(3x3): geometry not covered");
    K3fParams& p = pl.p;
    p.x = (const char*)x; p.w = w; p.nnz9 = nullptr; p.h = nullptr; p.h16 = wide ? nullptr : h16; p.h32 = wide ? reinterpret_cast<int32_t*>(h16) : nullptr; p.part = part;
    p.wn = (float)((1ll << wq->bits) - 1);
    mn_set_last_kernel(wide ? "k_k3s_fwd<2>" : "k_k3s_fwd<1>");
    { const double nx = (double)g->N * g->C * g->H * g->W, ny = (double)g->N * g->O * g->H * g->W; mn_prof_bytes(nx + (wide ? 4.0 : 2.0) * ny); }
    mn_prof_begin(s);
    if (wide) {
        raise_lds_limit((const void*)k_k3s_fwd<2>, pl.lds);
        hipLaunchKernelGGL(k_k3s_fwd<2>, dim3(pl.grid), dim3(256), pl.lds, s, p);
    } else {
        raise_lds_limit((const void*)k_k3s_fwd<1>, pl.lds);
        hipLaunchKernelGGL(k_k3s_fwd<1>, dim3(pl.grid), dim3(256), pl.lds, s, p);
    }
    mn_prof_end(s);
    MN_CHECK_LAUNCH("mn_qconv_bnq_fwd_stash(3x3)");
    return MN_OK;
}
