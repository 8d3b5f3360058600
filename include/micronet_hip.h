/*
 * libmicronet_hip -- C ABI of the MI355X (gfx950) fake-quantized conv hot path.
 *
 * This is the drop-in boundary under micronet's Python `torch.nn.Module` surface
 * (reference repo 666DZY666/micronet, paths relative to
 * micronet/compression/quantization/).  The reference has no native layer: each
 * entry point below replaces the chain of ATen ops that the cited reference lines
 * launch.  Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch); the library
 *     never allocates, frees or retains memory; scratch is passed in as `ws`;
 *   - tensors are contiguous fp32, activations NCHW, weights OIHW;
 *   - `stream` is a hipStream_t (passed as void*); all work is enqueued on it,
 *     nothing synchronises with the host;
 *   - return value: 0 on success, negative errno-style code otherwise
 *     (MN_EINVAL bad argument, MN_ENOTSUP unsupported shape/algo, MN_EHIP launch
 *     failure); `mn_last_error()` gives a thread-local message.
 */
#ifndef MICRONET_HIP_H
#define MICRONET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MN_OK 0
#define MN_EINVAL (-22)
#define MN_ENOTSUP (-95)
#define MN_EHIP (-5)
#define MN_ENOSPC (-28) /* workspace too small */

typedef void* mn_stream_t;

int mn_version(void);
const char* mn_last_error(void);
/* name (as rocprofv3 prints it, without the argument list) of the main kernel the last mn_conv2d_* call on this thread launched */
const char* mn_last_kernel(void);
/* measurement hook: the next mn_conv2d_* call on this thread records `start_event` / `stop_event` (hipEvent_t, may be NULL)
 * on its stream immediately around its main kernel launch (not around the small weight-pack / partial-reduce helpers). */
void mn_profile_next(void* start_event, void* stop_event);
/* library-kept measurement: while enabled (per calling thread), EVERY main kernel launch of the library (conv, BN+sign, pool) is
 * bracketed by two pooled HIP events on its stream.  mn_profile_collect -- call it once the stream is idle -- aggregates the
 * recorded spans by kernel name into `out` (at most `cap` entries), clears them and returns the number of entries.
 * `bytes` = the kernel's designed HBM bytes (each operand read or written once, int8 codes counted as 1 byte); `flops` = its algorithmic FLOPs
 * (2 x MACs; 0 for the streaming kernels: only the matrix-bound dense convolutions report it). */
typedef struct mn_prof_entry {
    char name[96];
    int64_t launches;
    double total_ms;
    double bytes;
    double flops;
} mn_prof_entry;
void mn_profile_enable(int on);
int mn_profile_collect(mn_prof_entry* out, int cap);
/* 1 if the library was built as the CPU SIMT emulation used by the unit tests, 0 for the gfx950 build */
int mn_is_emulation(void);
/* bf16 terms the dense (ResNet-family) backward kernels carry the fp32 gradient operand in: 2 (default: hi = rne(g), lo = rne(g - hi), |error| <= 2^-18 |g|) or 3
 * (MN_GRAD_TERMS=3 in the environment: the exact truncation split).  The nin_gc kernels (pointwise, grouped 3x3, first block) always use the exact three terms.
 * bench.py reports it (config.grad_terms) and prices the matrix-core passes with it. */
int mn_dense_grad_terms(void);

/* ------------------------------------------------------------------ DoReFa
 * wqaq/dorefa/quantize.py */
/* Round.forward 13-16: out = sign(v) * floor(|v| + 0.5) in fp32 */
int mn_round_half_away(const float* v, float* out, int64_t n, mn_stream_t stream);
/* ActivationQuantizer.forward 36-46 (a_bits in 2..31): y = rha(clamp(0.1x,0,1)/s)*s */
int mn_dorefa_act_fwd(const float* x, float* y, int64_t n, int a_bits, mn_stream_t stream);
/* its autograd backward: dx = ((g*s)/s) * [0 <= 0.1x <= 1] * 0.1 */
int mn_dorefa_act_bwd(const float* g, const float* x, float* dx, int64_t n, int a_bits, mn_stream_t stream);
/* WeightQuantizer.forward 61-73 over the whole tensor (global max of |tanh w|).
 * ws: >= mn_dorefa_w_ws_floats(n) floats; ws[0] receives M = max|tanh w|. */
int64_t mn_dorefa_w_ws_floats(int64_t n);
int mn_dorefa_w_fwd(const float* w, float* qw, int64_t n, int w_bits, float* ws, mn_stream_t stream);
/* backward incl. the path through the global max (ties share equally) */
int mn_dorefa_w_bwd(const float* g, const float* w, float* dw, int64_t n, int w_bits, float* ws, mn_stream_t stream);

/* y = tanh(x) exactly as the weight-quantizer kernels evaluate it: the CORRECTLY ROUNDED fp32 value (fp64 evaluation, one rounding).  The reference runs
 * torch.tanh on the CPU = Intel MKL VML vsTanh (HA) in a MKL build of torch: closed source, host-CPU dependent in the last bit (0.05 % ... 1.5 % of inputs differ
 * from the correctly rounded value by one ulp depending on the host) -- it cannot be restated; where such a difference straddles a rounding boundary a weight
 * code differs by one step (<= 2 in 10^5).  tests/golden/tanh_device_vs_cpu.json pins the kernels' function on inputs where the two differed. */
int mn_tanh_f32(const float* x, float* y, int64_t n, mn_stream_t stream);

/* The same quantizer over count <= 32 weight tensors in ONE launch per phase (host arrays of device pointers / element counts; nothing is allocated,
 * graph-capturable): a training step quantizes every conv's weights, the calls above are 2 + 3 launches of ~5 us per layer.  Bit-identical to them.
 * ws[i]: >= mn_dorefa_w_ws_floats(n[i]) floats each. */
int mn_dorefa_w_fwd_multi(const float* const* w, float* const* qw, float* const* ws, const int64_t* n, int32_t count, int w_bits, mn_stream_t stream);
int mn_dorefa_w_bwd_multi(const float* const* g, const float* const* w, float* const* dw, float* const* ws, const int64_t* n, int32_t count, int w_bits,
                          mn_stream_t stream);
/* the same with tanh(w) cached for the step (th[i]: n[i] floats per tensor, written by the forward): the backward takes the FORWARD's ws and th of the same
 * weights and skips the absmax pass; bit-identical results */
int mn_dorefa_w_fwd_multi_cached(const float* const* w, float* const* qw, float* const* ws, float* const* th, const int64_t* n, int32_t count, int w_bits,
                                 mn_stream_t stream);
int mn_dorefa_w_bwd_multi_cached(const float* const* g, const float* const* w, float* const* dw, float* const* ws, float* const* th, const int64_t* n,
                                 int32_t count, int w_bits, mn_stream_t stream);

/* ------------------------------------------------------------------ WbWtAb
 * wbwtab/quantize.py */
/* BinaryActivation.forward 13-19 / backward 22-36 */
int mn_binact_fwd(const float* x, float* y, int64_t n, mn_stream_t stream);
int mn_binact_bwd(const float* g, const float* x, float* dx, int64_t n, mn_stream_t stream);
/* WeightQuantizer W==3 branch 132-146 (+ Ternary 55-75). w: [O][K] rows (K = Cin/g*kh*kw).
 * stats: [O][4] = {alpha, thr, cnt, sum|w|>thr} written by fwd, read by bwd. */
int mn_ternary_w_fwd(const float* w, float* qw, float* stats, int64_t O, int64_t K, mn_stream_t stream);
int mn_ternary_w_bwd(const float* g, const float* w, const float* stats, float* dw, int64_t O, int64_t K,
                     mn_stream_t stream);
/* The same quantizer over n <= 32 weight tensors in ONE launch (host arrays of device pointers / row counts / row lengths; nothing is
 * allocated, graph-capturable): a training step quantizes every conv's weights, each call above is a ~5 us launch. */
int mn_ternary_w_fwd_multi(const float* const* w, float* const* qw, float* const* stats, const int64_t* O, const int64_t* K, int32_t n,
                           mn_stream_t stream);
int mn_ternary_w_bwd_multi(const float* const* g, const float* const* w, float* const* stats, float* const* dw, const int64_t* O, const int64_t* K,
                           int32_t n, mn_stream_t stream);
/* WeightQuantizer W==2 branch 121-130 incl. meancenter_clamp_convparams 98-102, which
 * MUTATES w in place (w -= mean over the Cin axis; clamp to [-1,1]).  w: [O][C][R] with R = kh*kw.
 * alpha: [O] written by fwd, read by bwd. */
int mn_binary_w_fwd(float* w_inplace, float* qw, float* alpha, int64_t O, int64_t C, int64_t R, mn_stream_t stream);
int mn_binary_w_bwd(const float* g, const float* w, const float* alpha, float* dw, int64_t O, int64_t K,
                    mn_stream_t stream);
/* ... over n <= 32 weight tensors in ONE launch each way (wbwtab/quantize.py:98-102, 121-130 once per conv of the net: a W = 2 nin_gc step quantizes 7 of them):
 * w[i] is [O[i]][C[i]][R[i]] (R = KH * KW <= 256) and is mutated in place like the single-tensor call; same arithmetic per tensor. */
int mn_binary_w_fwd_multi(float* const* w, float* const* qw, float* const* alpha, const int64_t* O, const int64_t* C, const int64_t* R, int32_t n, mn_stream_t stream);
int mn_binary_w_bwd_multi(const float* const* g, const float* const* w, float* const* alpha, float* const* dw, const int64_t* O, const int64_t* C, const int64_t* R,
                          int32_t n, mn_stream_t stream);

/* ------------------------------------------------------------------ IAO
 * wqaq/iao/quantize.py */
/* ObserverBase.forward 23-36 + update_range (MinMax 62-74, MovingAverage 101-113).
 * x viewed as [rows][cols]: rows == 1 is level 'L' (whole tensor), rows == O is 'C'/'FC'.
 * obs_kind: 0 = running min/max, 1 = EMA(momentum).  first != 0 copies (num_flag == 0).
 * min_val/max_val: [rows] module buffers, updated in place.
 * ws: >= mn_iao_observe_ws_floats(rows, cols) floats. */
int64_t mn_iao_observe_ws_floats(int64_t rows, int64_t cols);
int mn_iao_observe(const float* x, int64_t rows, int64_t cols, int obs_kind, int first, double momentum,
                   float* min_val, float* max_val, float* ws, mn_stream_t stream);
/* update_qparams (symmetric 293-305 / asymmetric 310-321) + a snapshot for the kernels.
 * update != 0: recompute scale/zero_point from min_val/max_val and store them ([rows] each);
 * update == 0: keep scale/zero_point (eval / qaft).  Always writes qp: [rows][4] =
 * {scale, zero_point, lo, hi} with (lo,hi) the clip-STE bounds of Round.forward 148-157. */
int mn_iao_qparams(const float* min_val, const float* max_val, int64_t rows, int bits, int q_type, int is_act,
                   int update, float* scale, float* zero_point, float* qp, mn_stream_t stream);
/* Quantizer.forward 227-239 with the snapshot: y = (clamp(rha(x/s - zp), qmin, qmax) + zp) * s */
int mn_iao_fq_fwd(const float* x, float* y, int64_t rows, int64_t cols, const float* qp, int bits, int q_type,
                  int is_act, mn_stream_t stream);
/* backward: dx = ((g*s)/s) * [qmin <= r <= qmax] * [lo <= v <= hi] */
int mn_iao_fq_bwd(const float* g, const float* x, float* dx, int64_t rows, int64_t cols, const float* qp, int bits,
                  int q_type, int is_act, mn_stream_t stream);
/* Fake-quant fused with the activation behind it: QuantReLU / QuantLeakyReLU / QuantSigmoid.forward (1196-1199, 1240-1243, 1279-1282):
 * y = act(Q(x)), per-tensor quantizer snapshot qp = {scale, zero_point, lo, hi}; act: 1 relu, 2 leaky_relu(slope), 3 sigmoid.
 * backward: dx = Q'(x) * act'(Q(x)) * g (threshold / leaky / sigmoid backward of ATen, then the clip-STE of Round.backward 163-168). */
int mn_iao_fq_act_fwd(const float* x, float* y, int64_t n, const float* qp, int bits, int q_type, int act, float slope, mn_stream_t stream);
int mn_iao_fq_act_bwd(const float* g, const float* x, float* dx, int64_t n, const float* qp, int bits, int q_type, int act, float slope,
                      mn_stream_t stream);
/* Fake-quant fused with average pooling: QuantAvgPool2d.forward (1401-1411) for kernel k x k, stride k, no padding (H % k == W % k == 0), and
 * QuantAdaptiveAvgPool2d((1, 1)) as k == H == W (1433-1436).  x: [planes][H][W], y: [planes][H/k][W/k]. */
int mn_iao_fq_avgpool_supported(int64_t H, int64_t W, int64_t k);
int mn_iao_fq_avgpool_fwd(const float* x, float* y, int64_t planes, int64_t H, int64_t W, int64_t k, const float* qp, int bits, int q_type,
                          mn_stream_t stream);
int mn_iao_fq_avgpool_bwd(const float* g, const float* x, float* dx, int64_t planes, int64_t H, int64_t W, int64_t k, const float* qp, int bits,
                          int q_type, mn_stream_t stream);
/* Fake-quant fused with the 2 x 2 / stride-2 max-pool behind it: QuantMaxPool2d.forward (1347-1359) = max_pool2d(Q(x), 2, 2).  x / dx: [planes][H][W] (even H,
 * W % 8 == 0), y / gy / idx: [planes][H/2][W/2]; idx = the argmax of every window (0..3, ATen's tie / NaN rule); mm (nullable): 2 * mn_iao_fq_maxpool2x2_mm_count
 * floats = per-block (min, max) of y for mn_iao_observe_partials (the observer of the layer that reads y).  Backward: pool scatter, then the quantizer's clip-STE
 * (Round.backward 163-168), then -- relu_mask != 0, x being the output of a ReLU -- that ReLU's mask [x > 0]. */
int mn_iao_fq_maxpool2x2_supported(int64_t H, int64_t W);
int64_t mn_iao_fq_maxpool2x2_mm_count(int64_t planes, int64_t H, int64_t W);
int mn_iao_fq_maxpool2x2_fwd(const float* x, int64_t planes, int64_t H, int64_t W, const float* qp, int bits, int q_type, float* y, uint8_t* idx, float* mm,
                             mn_stream_t stream);
int mn_iao_fq_maxpool2x2_bwd(const float* gy, const uint8_t* idx, const float* x, int64_t planes, int64_t H, int64_t W, const float* qp, int bits, int q_type,
                             int relu_mask, float* dx, mn_stream_t stream);
/* Streaming helpers of the BN-fused blocks (n % 4 == 0, 16-byte aligned): mn_add_relu_mask: out = (a [+ b]) * [x > 0] (b, x nullable) -- the sum of
 * QuantBNFuseConv2d's two input gradients (quantised path 947-955, statistics path 843-855) with the ReLU mask of the block in front, or a ReLU backward alone;
 * mn_relu_mm: y = relu(x) (in place allowed) + per-block (min, max) of y in mm (nullable, 2 * mn_relu_mm_count(n) floats) for mn_iao_observe_partials. */
int mn_add_relu_mask(const float* a, const float* b, const float* x, float* out, int64_t n, mn_stream_t stream);
int64_t mn_relu_mm_count(int64_t n);
int mn_relu_mm(const float* x, float* y, int64_t n, float* mm, mn_stream_t stream);
/* HistogramObserver.forward (116-139), the PTQ percentile calibrator: cur = k-th smallest |x| (k 1-based = int(percentile * n), exact --
 * a radix select on the bit patterns, bit-identical to torch.kthvalue); max_val = cur when first != 0 else (1 - momentum) * max_val +
 * momentum * cur, on the device.  out (nullable) receives cur.  ws: mn_kth_abs_ws_bytes() bytes, 4-byte aligned. */
int64_t mn_kth_abs_ws_bytes(void);
int mn_hist_observe(const float* x, int64_t n, int64_t k, int first, double momentum, float* max_val, float* out, void* ws, mn_stream_t stream);
/* QuantAdd.forward 1487-1492: union of two observer ranges */
int mn_iao_union_range(const float* min_a, const float* max_a, const float* min_b, const float* max_b,
                       float* min_out, float* max_out, mn_stream_t stream);
/* QuantAdd (wqaq/iao/quantize.py:1484-1498: out = Q(res) + Q(shortcut) with ONE quantizer whose range is the union of the two inputs' observed ranges) in three
 * launches: mn_iao_qadd_observe = observer_res(res), observer_shortcut(shortcut) (per-tensor, obs_kind 0 running min / max, 1 moving average; first_*: the
 * observer's first call), union -> (min_out, max_out) = the shared quantizer's observer, and its qparams (update != 0: scale / zero_point recomputed, as in
 * training; 0: taken as they are) -> qp {scale, zero_point, lo, hi}; ws: mn_iao_qadd_ws_floats() floats.  mn_iao_qadd_fwd: out = fq(res) + fq(shortcut);
 * mn_iao_qadd_bwd: both clip-STE gradients from one read of g.  relu != 0: the ReLU a ResNet block applies to the sum (models/resnet.py:63) in the same
 * pass (backward: the gradient passes where the recomputed sum is > 0).  n % 4 == 0, 16-byte aligned tensors.  Bit-identical to the separate entry points. */
/* The IAO weight quantizers of a whole net (per-channel observers: one row per output channel) in ONE launch per direction: for every row of every tensor the
 * observer update (obs_kind 0 running min / max, 1 moving average; first[i]: tensor i's observer is at its first call), scale / zero_point, the
 * {scale, zero_point, lo, hi} snapshot qp[i][rows][4] and the fake-quantised row (wqaq/iao/quantize.py:15-36, 293-321, 227-239, weights: activation_weight_flag 0).
 * Backward: the clip-STE of every row from the forward's qp.  count <= 32.  Bit-identical to mn_iao_observe + mn_iao_qparams + mn_iao_fq_fwd / _bwd per tensor. */
int mn_iao_w_fwd_multi(const float* const* w, float* const* qw, float* const* min_val, float* const* max_val, float* const* scale, float* const* zero_point,
                       float* const* qp, const int64_t* rows, const int64_t* cols, const int32_t* first, int32_t count, int obs_kind, double momentum, int bits,
                       int q_type, mn_stream_t stream);
int mn_iao_w_bwd_multi(const float* const* g, const float* const* w, float* const* dw, float* const* qp, const int64_t* rows, const int64_t* cols, int32_t count,
                       int bits, int q_type, mn_stream_t stream);
int64_t mn_iao_qadd_ws_floats(void);
int64_t mn_iao_qadd_mm_count(int64_t n);
int mn_iao_qadd_fwd_mm(const float* res, const float* shortcut, float* out, int64_t n, const float* qp, int bits, int q_type, int relu, float* mm, mn_stream_t stream);
/* observer update (as mn_iao_observe with rows == 1) from the (min, max) partials a producing kernel left: mm[0 .. count) minima, mm[count .. 2 count) maxima */
int mn_iao_observe_partials(const float* mm, int64_t count, int obs_kind, int first, double momentum, float* min_val, float* max_val, mn_stream_t stream);
/* the same followed by the per-tensor quantizer's update_qparams (wqaq/iao/quantize.py:293-321) in the same launch: scale / zero_point recomputed, qp [4] written */
int mn_iao_observe_partials_qparams(const float* mm, int64_t count, int obs_kind, int first, double momentum, float* min_val, float* max_val, int bits, int q_type,
                                    int is_act, float* scale, float* zero_point, float* qp, mn_stream_t stream);
int mn_iao_qadd_observe(const float* res, const float* shortcut, int64_t n, int obs_kind, int first_res, int first_shortcut, double momentum, float* min_res,
                        float* max_res, float* min_shortcut, float* max_shortcut, float* min_out, float* max_out, int bits, int q_type, int update, float* scale,
                        float* zero_point, float* qp, float* ws, mn_stream_t stream);
/* the same bookkeeping from the (min, max) partials the producers of res / shortcut left behind (mn_bn2d_fwd_mm, mn_bnrelu_fwd_mm, mn_iao_qadd_fwd_mm): ONE launch, neither
 * tensor is read; bit-identical to mn_iao_qadd_observe (min / max are exact and order-free) */
int mn_iao_qadd_observe_partials(const float* mm_res, int64_t count_res, const float* mm_shortcut, int64_t count_shortcut, int obs_kind, int first_res,
                                 int first_shortcut, double momentum, float* min_res, float* max_res, float* min_shortcut, float* max_shortcut, float* min_out,
                                 float* max_out, int bits, int q_type, int update, float* scale, float* zero_point, float* qp, mn_stream_t stream);
int mn_iao_qadd_fwd(const float* res, const float* shortcut, float* out, int64_t n, const float* qp, int bits, int q_type, int relu, mn_stream_t stream);
int mn_iao_qadd_bwd(const float* g, const float* res, const float* shortcut, float* dres, float* dshortcut, int64_t n, const float* qp, int bits, int q_type,
                    int relu, mn_stream_t stream);
/* QuantBNFuseConv2d.forward 853-855: per-channel mean and UNBIASED variance of o[N][C][HW] over (N,HW).
 * stats: [2][C] = mean, var.  ws: >= mn_bn_stats_ws_floats(N, C, HW) floats. */
int64_t mn_bn_stats_ws_floats(int64_t N, int64_t C, int64_t HW);
int mn_bn_stats_fwd(const float* o, int64_t N, int64_t C, int64_t HW, float* stats, float* ws, mn_stream_t stream);
/* backward: do = dmean/n + dvar * 2 (o - mean)/(n - 1) */
int mn_bn_stats_bwd(const float* o, const float* stats, const float* dmean, const float* dvar, float* d_o,
                    int64_t N, int64_t C, int64_t HW, mn_stream_t stream);

/* ------------------------------------------------------------------ convolution
 * F.conv2d call sites: dorefa 113-121, wbwtab 186-194, iao 498-506 / 843-851 / 947-993;
 * F.linear: dorefa 198, iao 1156 (use H = W = 1, 1x1 kernel). */
typedef struct mn_conv_geom {
    int32_t N, C, H, W;      /* input  [N][C][H][W] */
    int32_t O, KH, KW;       /* weight [O][C/groups][KH][KW] */
    int32_t stride_h, stride_w, pad_h, pad_w, dil_h, dil_w, groups;
    int32_t in_shuffle;      /* > 1: the conv reads its input through a ShuffleNet channel shuffle with this many groups
                                (models/nin_gc.py:4-15 `channel_shuffle`, called in front of the conv at 53-56): logical input
                                channel cl lives at physical channel (cl % s) * (C / s) + cl / s of x; dx is written through the
                                same map.  Folding the permutation into the addressing removes a full copy of the activation
                                tensor in forward and backward.  Code-domain kernels only (else MN_ENOTSUP). 0 / 1: none. */
} mn_conv_geom;

/* activation quantizer fused into the conv prologue (fwd, bwd_weight) and into the
 * clip-STE epilogue of bwd_data */
#define MN_ACTQ_NONE 0
#define MN_ACTQ_DOREFA 1
#define MN_ACTQ_IAO 2
#define MN_ACTQ_SIGN8 3 /* `x` is NOT fp32: it points to int8 codes in {-1, +1}, laid out NCHW like x -- the packed output of
                          mn_bnsign_fwd_i8 / mn_maxpool2x2_sign8_fwd (the BinaryActivation of wbwtab/quantize.py:13-19 stored in
                          one byte per element: a quarter of the HBM traffic of the fp32 +-1 tensor).  fwd and bwd_weight read the
                          codes directly (4-byte aligned rows: H*W % 4 == 0); bwd_data ignores x (the clip-STE of the sign lives
                          in mn_bnsign_bwd).  Code-domain kernels only (else MN_ENOTSUP). */
#define MN_ACTQ_CODE8 4 /* `x` is NOT fp32: it points to the k-bit activation CODES j of the quantizer (uint8 in [0, 2^bits - 1], bits <= 8, NCHW like x;
                          value = j * s, s = 1 / (2^bits - 1): DoReFa ActivationQuantizer, wqaq/dorefa/quantize.py:43-45) as written by mn_qa_fwd -- one byte
                          per element instead of four, and the quantizer is not re-evaluated.  Supported by bwd_weight (dw = s * sum gy * j) and bwd_data
                          (x ignored: the clip-STE lives in mn_qa_bwd_*) where mn_qconv_bnq_supported says so; the forward on codes is
                          mn_qconv_bnq_fwd_stash. */
#define MN_ACTQ_X_IS_CODE 1 /* flags: optional hint that with MN_ACTQ_NONE x holds small integers exact in bf16 (the +-1 of
                              wbwtab's BinaryActivation, wbwtab/quantize.py:13-19).  Never required: real-valued x is split
                              into exact bf16 terms on the fly and all-zero terms are skipped. */
#define MN_ACTQ_CODES_GIVEN 2 /* flags, MN_ACTQ_IAO forward of a dense layer: mn_actq.codes (and .ste_mask) ALREADY hold this forward's activation codes -- written by
                                 mn_bn_apply_codes, the BatchNorm [+ ReLU] in front fused with this conv's quantizer (models/resnet.py:17-29 under
                                 wqaq/iao/quantize.py:492-507) -- so mn_conv2d_fwd quantises nothing and never reads x (pass any aligned non-NULL pointer). */
typedef struct mn_actq {
    int32_t mode;    /* MN_ACTQ_* */
    int32_t bits;
    int32_t q_type;  /* iao: 0 symmetric, 1 asymmetric */
    int32_t flags;   /* MN_ACTQ_X_IS_CODE */
    const float* qp; /* iao: device {scale, zero_point, lo, hi} (per-tensor) */
    void* codes;     /* optional, MN_ACTQ_IAO on a dense layer (mn_conv2d_iao_codes_bytes(g, aq, wq) > 0): that many bytes owned by the caller.  mn_conv2d_fwd
                        writes the activation's signed codes there instead of into its workspace, and a later mn_conv2d_bwd_weight handed the SAME buffer (same x,
                        same qp) reads them instead of quantising x again.  NULL: every call quantises for itself. */
    void* stats;     /* optional, MN_ACTQ_IAO forward of a dense layer on the int8 matrix cores (mn_conv2d_iao_stats_rows(g, aq, wq) = R > 0): R * O * 2 doubles owned
                        by the caller.  mn_conv2d_fwd leaves the exact per-channel sums of the integer accumulator there -- stats[(r * O + o) * 2 + {0, 1}] = sum acc,
                        sum acc^2 over the pixels of partial r -- from its epilogue: the BatchNorm behind the conv (models/resnet.py:17-29) then needs no statistics
                        pass of its own (mn_bn_fwd_acc).  NULL: none. */
    const float* dx_add; /* optional, mn_conv2d_bwd_data of a dense MN_ACTQ_IAO layer (mn_conv2d_bwd_data_add_supported): a tensor shaped like dx that is ADDED to dx in the
                        store -- after the quantizer's clip-STE: dx = STE(W^T gy) + dx_add.  The gradient a residual block's identity shortcut carries to the same input
                        (models/resnet.py:60-65 with QuantAdd, wqaq/iao/quantize.py:1484-1498) then needs no add kernel of its own.  NULL: none.  A call that cannot honour
                        it fails with MN_ENOTSUP (never silently drops it). */
    void* ste_mask;  /* optional, together with `codes` (same layers): mn_conv2d_iao_codes_bytes(g, aq, wq) / 8 bytes owned by the caller.  mn_conv2d_fwd then also leaves the
                        quantizer's clip-STE decision of every element there -- bit e of byte i: does the gradient of element 8 i + e pass (Round.backward and the clamp,
                        wqaq/iao/quantize.py:163-168, 232) -- and a later mn_conv2d_bwd_data handed the SAME buffer (same x, same qp) applies the STE from those bits instead
                        of reading the fp32 x again (4 bytes per element -> 1 bit; bit-identical dx).  NULL: backward-data reads x. */
    void* acc_mm;    /* optional, same layers and row count R as `stats`: R * O * 2 int32 owned by the caller.  mn_conv2d_fwd leaves the per-channel extrema of its integer
                        accumulator there -- acc_mm[(r * O + o) * 2 + {0, 1}] = min acc, max acc over the (valid) pixels of partial r.  y = al[o] * acc + bias[o] and the
                        BatchNorm [+ ReLU] behind it are monotone in acc per channel, so the (min, max) of THAT activation -- what the next layer's observer wants
                        (wqaq/iao/quantize.py:23-36) -- follow without a pass over it: mn_bn_acc_prep.  NULL: none. */
} mn_actq;

/* How the (already fake-quantised, fp32 OIHW) weight tensor factors into integer codes x per-channel scale.  The
 * code-domain kernels (MN_ALGO_QGEMM) contract bf16-exact integer codes on v_mfma_f32_16x16x32_bf16 and apply the scale
 * outside the sum; the descriptor tells the pack kernel how to recover the codes from w:
 *   MN_WQ_TERNARY  w[o,:] = t * alpha[o], t in {-1,0,+1}   (wbwtab/quantize.py:121-146)   alpha[o] := max|w[o,:]|
 *   MN_WQ_DOREFA   w = (2k - n)/n, n = 2^bits - 1           (wqaq/dorefa/quantize.py:68-72)
 *   MN_WQ_IAO      w = code * scale[o]                       (wqaq/iao/quantize.py:227-239); scale = the quantizer's buffer
 *   MN_WQ_REAL     arbitrary fp32 weights: the code-domain kernels do not apply (fp32-MFMA kernels are used) */
#define MN_WQ_REAL 0
#define MN_WQ_TERNARY 1
#define MN_WQ_DOREFA 2
#define MN_WQ_IAO 3
typedef struct mn_wq {
    int32_t mode;         /* MN_WQ_* */
    int32_t bits;
    int32_t q_type;       /* iao: 0 symmetric, 1 asymmetric */
    int32_t per_channel;  /* iao: stride, in floats, between the scales of consecutive out-channels: 0 = one scale for the
                             tensor, 1 = a dense [O] vector, 4 = the rows of an mn_iao_qparams snapshot */
    const float* scale;   /* iao: device pointer */
    const void* packed_fwd; /* optional (NULL: the kernels pack the codes themselves, once per call): the weight codes in the fragment order of the dense */
    const void* packed_bwd; /* family's forward / backward-data kernels, written by mn_qd_pack_multi for THIS `w` (same step, same quantizer state) -- or, for a
                               POINTWISE (1 x 1, stride 1) layer of the code kernels, the images mn_qg_pack_multi wrote (the two families never share a geometry) */
} mn_wq;

#define MN_ALGO_AUTO 0
#define MN_ALGO_DIRECT 1 /* generic VALU kernels: any geometry */
#define MN_ALGO_MFMA 2   /* implicit-GEMM on v_mfma_f32_16x16x4_f32; MN_ENOTSUP if the shape does not tile */
#define MN_ALGO_QGEMM 3  /* code-domain kernels on v_mfma_f32_16x16x32_bf16 (exact integer codes; real operands as exact
                            3-term bf16 splits); MN_ENOTSUP if geometry / quantizer combination is not covered.
                            MN_ALGO_AUTO tries QGEMM, then MFMA, then DIRECT. */

/* which: 0 fwd, 1 bwd_data, 2 bwd_weight.  Bytes of `ws` the call needs for `algo` (AUTO: enough for whichever is chosen). */
int64_t mn_conv2d_ws_bytes(const mn_conv_geom* g, int which, int algo);
/* 1 if MN_ALGO_MFMA supports this geometry for `which` */
int mn_conv2d_mfma_supported(const mn_conv_geom* g, int which);
/* 1 if the first-layer kernels (real fp32 operands, groups 1, stride 1, "same" padding, Cin*KH*KW <= 76: the un-quantised first
 * convolution of the DoReFa / WbWtAb nets, wqaq/dorefa/quantize.py:206, wbwtab/quantize.py:251) cover `which` (0 fwd, 2 bwd_weight);
 * MN_ALGO_AUTO uses them when no activation quantizer is fused. */
int mn_conv2d_first_supported(const mn_conv_geom* g, int which);
/* 1 if MN_ALGO_QGEMM supports this geometry and quantizer combination for `which` (aq / wq may be NULL = none / real) */
int mn_conv2d_qgemm_supported(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, int which);
int64_t mn_conv2d_iao_codes_bytes(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq);   /* size of mn_actq.codes for this layer; 0: not used */
int mn_conv2d_bwd_data_add_supported(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq);   /* 1: mn_conv2d_bwd_data(algo = MN_ALGO_AUTO or MN_ALGO_QGEMM) honours mn_actq.dx_add for this layer */
int64_t mn_conv2d_iao_stats_rows(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq);   /* partial rows R of mn_actq.stats for this layer; 0: not available */
/* y[N][O][Ho][Wo] = conv2d(actq(x), w, bias); w are the (already fake-quantised) fp32 weights, wq says how they factor
 * (NULL = MN_WQ_REAL) */
int mn_conv2d_fwd(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, const float* x, const float* w,
                  const float* bias, float* y, void* ws, int64_t ws_bytes, int algo, mn_stream_t stream);
/* y = relu?(conv2d(actq(x), w, bias)) for pointwise (1 x 1, stride 1) code-domain layers, with -- mm != NULL -- the per-wave (min, max) of everything stored left in
 * mm[0 .. count) / mm[count .. 2 count), count = mn_conv2d_fwd_act_mm_count(g, aq, wq) (0: layer not covered): the block `relu(bn(conv(x)))` of the reference's nets
 * after the IAO rewrite folded the BatchNorm into the conv (models/nin_gc.py:53-59 with bn = nn.Identity, wqaq/iao/quantize.py:1567-1624); the NEXT layer's
 * observer (wqaq/iao/quantize.py:23-36) is then updated by mn_iao_observe_partials without a pass of its own over the activation.  Also covered: the first layer
 * of a net (aq = NULL / MN_ACTQ_NONE, wq = NULL: real operands on the first-layer kernels, mn_conv2d_first_supported). */
int64_t mn_conv2d_fwd_act_mm_count(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq);
int mn_conv2d_fwd_act(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, const float* x, const float* w, const float* bias, float* y, int relu, float* mm,
                      void* ws, int64_t ws_bytes, mn_stream_t stream);
/* dx = conv2d_backward_data(gy, w) * d actq(x)/dx  (x may be NULL when aq->mode == NONE) */
int mn_conv2d_bwd_data(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, const float* gy, const float* w,
                       const float* x, float* dx, void* ws, int64_t ws_bytes, int algo, mn_stream_t stream);
/* dw = conv2d_backward_weight(gy, actq(x)); dbias = sum gy (dbias may be NULL) */
int mn_conv2d_bwd_weight(const mn_conv_geom* g, const mn_actq* aq, const float* gy, const float* x, float* dw,
                         float* dbias, void* ws, int64_t ws_bytes, int algo, mn_stream_t stream);

/* ------------------------------------------------------------------ BatchNorm2d + BinaryActivation, fused
 * models/nin_gc.py:53-59 `relu(bn(conv(x)))` with the ReLU replaced by BinaryActivation (wbwtab/quantize.py:79-94, 319-322).
 * y, a, da, dy: [N][C][HW] fp32 (HW % 4 == 0, 16-byte aligned).  save: [2][C] = mean, invstd (written by fwd, read by bwd).
 * training != 0: batch statistics; running_mean / running_var (nullable) are updated in place with `momentum` and the
 * unbiased variance, as nn.BatchNorm2d does.  training == 0: the running statistics normalise, nothing is updated.
 * ws: >= mn_bnsign_ws_floats(C) floats, 8-byte aligned.  The outputs must not alias the inputs (`a` != `y`, `dy` != `da`, `dy` != `y`): the apply pass of a
 * training call finishes the statistics itself (every block re-reads the channel's partial sums and its first element of y). */
int64_t mn_bnsign_ws_floats(int64_t C);
int mn_bnsign_fwd(const float* y, int64_t N, int64_t C, int64_t HW, const float* gamma, const float* beta, float eps, float momentum,
                  int training, float* running_mean, float* running_var, float* save, float* a, float* ws, mn_stream_t stream);
/* same, but `a` is written as int8 sign codes {-1,+1} (one byte per element, NCHW like y): the packed activation the
 * code-domain conv kernels read with MN_ACTQ_SIGN8.  NaN inputs map to +1 (no NaN in int8). */
int mn_bnsign_fwd_i8(const float* y, int64_t N, int64_t C, int64_t HW, const float* gamma, const float* beta, float eps, float momentum,
                     int training, float* running_mean, float* running_var, float* save, int8_t* a, float* ws, mn_stream_t stream);
/* nn.MaxPool2d(kernel_size=2, stride=2) between binary-activation blocks (models/nin_gc.py:88,119) on int8 sign codes:
 * a [planes][H][W] -> out [planes][H/2][W/2] (even H, W % 8 == 0).  Backward routes each output gradient to the first
 * maximum of its window in row-major order (what ATen's max_pool2d backward does): din [planes][H][W] fp32 (W % 4 == 0). */
int mn_maxpool2x2_sign8_fwd(const int8_t* a, int64_t planes, int64_t H, int64_t W, int8_t* out, mn_stream_t stream);
int mn_maxpool2x2_sign8_bwd(const float* dout, const int8_t* a, int64_t planes, int64_t H, int64_t W, float* din, mn_stream_t stream);
/* dy = d loss / d y given da = d loss / d a (clip-STE of the sign through the BatchNorm backward); dgamma / dbeta nullable */
int mn_bnsign_bwd(const float* da, const float* y, const float* save, const float* gamma, const float* beta, int64_t N, int64_t C,
                  int64_t HW, int training, float* dy, float* dgamma, float* dbeta, float* ws, mn_stream_t stream);

/* nn.MaxPool2d(2, 2) on fp32 activations (models/nin_gc.py:88,119 in the DoReFa / IAO nets): forward also writes the argmax of every window
 * as one byte (0..3, ATen's tie / NaN rule), the backward scatters from it.  x / dx: [planes][H][W], y / gy / idx: [planes][H/2][W/2]. */
int mn_maxpool2x2_f32_supported(int64_t H, int64_t W);
int mn_maxpool2x2_f32_fwd(const float* x, int64_t planes, int64_t H, int64_t W, float* y, uint8_t* idx, mn_stream_t stream);
int mn_maxpool2x2_f32_bwd(const float* gy, const uint8_t* idx, int64_t planes, int64_t H, int64_t W, float* dx, mn_stream_t stream);
/* nn.AvgPool2d whose window covers the whole image (the tail of nin / nin_gc, models/nin_gc.py:139): x [planes][HW] -> y [planes] = sum / HW; backward gy / HW. */
int mn_avgpool_global_fwd(const float* x, int64_t planes, int64_t HW, float* y, mn_stream_t stream);
int mn_avgpool_global_bwd(const float* gy, int64_t planes, int64_t HW, float* dx, mn_stream_t stream);
/* BatchNorm2d + ReLU fused the same way (relu(batch_norm(y)) of the ConvBNReLU blocks in the DoReFa / IAO nets, models/nin_gc.py:53-59;
 * backward mask z > 0): same arguments as mn_bnsign_fwd / mn_bnsign_bwd, fp32 output. */
int mn_bnrelu_fwd(const float* y, int64_t N, int64_t C, int64_t HW, const float* gamma, const float* beta, float eps, float momentum,
                  int training, float* running_mean, float* running_var, float* save, float* a, float* ws, mn_stream_t stream);
int mn_bnrelu_bwd(const float* da, const float* y, const float* save, const float* gamma, const float* beta, int64_t N, int64_t C,
                  int64_t HW, int training, float* dy, float* dgamma, float* dbeta, float* ws, mn_stream_t stream);
/* mn_bnrelu_fwd that also leaves per-block (min, max) of its output in mm (2 * mn_bnrelu_mm_count(N, C, HW) floats): the IAO observer of the layer that reads
 * the activation (wqaq/iao/quantize.py:23-36) is then updated by mn_iao_observe_partials without a pass of its own over the tensor. */
int64_t mn_bnrelu_mm_count(int64_t N, int64_t C, int64_t HW);
int mn_bnrelu_fwd_mm(const float* y, int64_t N, int64_t C, int64_t HW, const float* gamma, const float* beta, float eps, float momentum,
                     int training, float* running_mean, float* running_var, float* save, float* a, float* ws, float* mm, mn_stream_t stream);
/* plain nn.BatchNorm2d (no activation behind it: the BatchNorms in front of a residual add, models/resnet.py:21-29) on the same streaming kernels: same
 * arguments as mn_bnrelu_fwd / _bwd. */
int mn_bn2d_fwd(const float* y, int64_t N, int64_t C, int64_t HW, const float* gamma, const float* beta, float eps, float momentum,
                int training, float* running_mean, float* running_var, float* save, float* a, float* ws, mn_stream_t stream);
/* ... + per-block (min, max) of the output, mm: 2 * mn_bnrelu_mm_count(N, C, HW) floats */
int mn_bn2d_fwd_mm(const float* y, int64_t N, int64_t C, int64_t HW, const float* gamma, const float* beta, float eps, float momentum, int training,
                   float* running_mean, float* running_var, float* save, float* a, float* ws, float* mm, mn_stream_t stream);
/* BatchNorm2d [+ ReLU] in TRAINING mode behind a dense IAO conv whose forward left the exact sums of its integer accumulator (mn_actq.stats): y = al[c] * acc + cb[c]
 * with al[c] = sa[0] * sw[c * sw_stride] (the conv epilogue's fp32 scale) and cb = the conv bias (nullable), so mean and variance of y follow in fp64 from
 * (sum acc, sum acc^2) -- ONE streaming pass (normalise [+ ReLU] [+ per-block (min, max) -> mm, nullable]) instead of statistics pass + apply pass.  act: 1 ReLU
 * (mn_bnrelu_fwd), 2 none (mn_bn2d_fwd).  save / running statistics / a as mn_bnsign_fwd; the backward is the ordinary mn_bnrelu_bwd / mn_bn2d_bwd. */
int mn_bn_fwd_acc(const float* y, int64_t N, int64_t C, int64_t HW, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                  float* running_var, float* save, float* a, float* mm, int act, const double* stats, int64_t rows, const float* sa, const float* sw,
                  int64_t sw_stride, const float* conv_bias, mn_stream_t stream);
int mn_bn2d_bwd(const float* da, const float* y, const float* save, const float* gamma, const float* beta, int64_t N, int64_t C,
                int64_t HW, int training, float* dy, float* dgamma, float* dbeta, float* ws, mn_stream_t stream);
/* The BatchNorm [+ ReLU] behind a dense IAO conv and the activation quantizer of the NEXT dense IAO conv as one streaming pass (models/resnet.py:17-29: conv -> bn -> relu
 * -> conv under wqaq/iao/quantize.py:492-507), in three calls:
 *   mn_bn_acc_prep     per channel, no pass over y: batch statistics from the conv's exact accumulator sums (the arithmetic of mn_bn_fwd_acc: save, running statistics)
 *                      and, from the accumulator's extrema (mn_actq.acc_mm), the (min, max) of a = act(bn(y)) over that channel -> mm[c], mm[C + c]: a partials buffer
 *                      of count C for mn_iao_observe_partials[_qparams] -- the observer sees exactly the extrema a pass over `a` would have found (every step of
 *                      acc -> y -> bn -> relu is monotone in fp32), before `a` exists;
 *   (the caller updates the quantizer: mn_iao_observe_partials_qparams -> qp)
 *   mn_bn_apply_codes  ONE pass over y: a = act(bn(y)) with the finished `save`, then the symmetric `bits`-bit quantizer qp exactly as the dense conv's own code pass
 *                      evaluates it -> signed codes (1 byte per element, layout of y) + clip-STE bits (1 bit per element): what mn_conv2d_fwd consumes under
 *                      MN_ACTQ_CODES_GIVEN and mn_conv2d_bwd_data / _bwd_weight through mn_actq.codes / .ste_mask.  fp32 `a` is never written (5.1 B per element
 *                      instead of 8 + 5.1).  HW % 8 == 0; codes: N * C * HW bytes rounded up to 256, mask: an eighth of that.
 *   mn_bn_apply        the same normalisation written as fp32 (a consumer outside the fused path; the values the codes were taken from).
 * act: 1 ReLU, 2 none.  The backward is the ordinary mn_bnrelu_bwd / mn_bn2d_bwd on (da, y). */
int mn_bn_acc_prep(int64_t N, int64_t C, int64_t HW, const float* gamma, const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                   float* save, int act, const double* stats, const int32_t* acc_mm, int64_t rows, const float* sa, const float* sw, int64_t sw_stride,
                   const float* conv_bias, float* mm, mn_stream_t stream);
int mn_bn_apply_codes(const float* y, int64_t N, int64_t C, int64_t HW, const float* gamma, const float* beta, const float* save, int act, const float* qp, int bits,
                      int8_t* codes, uint8_t* ste_mask, mn_stream_t stream);
int mn_bn_apply(const float* y, int64_t N, int64_t C, int64_t HW, const float* gamma, const float* beta, const float* save, int act, float* a, mn_stream_t stream);
/* The END of an IAO residual block in one pass: out = [relu] (Q(res) + Q(shortcut)) with res = bn(res_y) and shortcut = sc_x itself (sc_save NULL: identity) or bn(sc_x)
 * (the down-sampling blocks), one shared per-tensor quantizer qp (models/resnet.py:21-29, 60-65 with QuantAdd, wqaq/iao/quantize.py:1484-1498) -- instead of one
 * BatchNorm apply pass per side + mn_iao_qadd_fwd.  *_save = {mean, invstd} [2][C] as mn_bn_acc_prep finishes them (which also hands the QuantAdd's two input observers
 * their ranges: mn_iao_qadd_observe_partials).  mm (nullable): 2 * mn_bnrelu_mm_count(N, C, HW) floats, per-block (min, max) of out.  bits_res / bits_sc: N * C * HW / 8
 * bytes each; bit e of byte i = element 8 i + e: (the ReLU passes) and (the quantizer's clip-STE passes that input) -- what mn_iao_qadd_bn_bwd reads.  HW % 8 == 0.
 * mn_iao_qadd_bn_bwd: backward of ONE BatchNorm side from g = d loss / d out: dy (+ dgamma, dbeta) of that BatchNorm, the gradient of its output formed from (g, bits)
 * on the way; d_other (nullable, with bits_other): the gradient of the other input -- the identity shortcut's -- from the same read of g.  Bit-identical to
 * mn_iao_qadd_bwd followed by mn_bn2d_bwd.  ws: mn_bnsign_ws_floats(C) floats. */
int mn_iao_qadd_bn_fwd(const float* res_y, const float* res_save, const float* res_gamma, const float* res_beta, const float* sc_x, const float* sc_save,
                       const float* sc_gamma, const float* sc_beta, int64_t N, int64_t C, int64_t HW, const float* qp, int bits, int q_type, int relu, float* out, float* mm,
                       uint8_t* bits_res, uint8_t* bits_sc, mn_stream_t stream);
int mn_iao_qadd_bn_bwd(const float* g, const float* y, const float* save, const float* gamma, const float* beta, int64_t N, int64_t C, int64_t HW, const float* qp,
                       const uint8_t* bits, const uint8_t* bits_other, float* dy, float* d_other, float* dgamma, float* dbeta, float* ws, mn_stream_t stream);
/* The tail of the reference's nets in one launch per direction: pooled[n][c] = mean over the image of relu(batch_norm(y)) -- BatchNorm2d -> ReLU -> AvgPool2d over the whole
 * map (models/nin_gc.py:136-147), TRAINING mode (batch statistics, running statistics updated with the unbiased variance, save = {mean, invstd} [2][C]); one block per
 * channel: meant for the few-channel classifier map (N x 10 x 8 x 8), where the step pays launches, not bytes.  HW % 4 == 0.  _bwd: dy, dgamma, dbeta from d pooled. */
int mn_bnrelu_gap_supported(int64_t N, int64_t C, int64_t HW);          /* HW / 4 a power of two <= 64 and N * HW <= 16384: a channel lives in one block's registers */
int mn_bnrelu_gap_fwd(const float* y, int64_t N, int64_t C, int64_t HW, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                      float* running_var, float* save, float* pooled, mn_stream_t stream);
int mn_bnrelu_gap_bwd(const float* dpool, const float* y, const float* save, const float* gamma, const float* beta, int64_t N, int64_t C, int64_t HW, float* dy, float* dgamma,
                      float* dbeta, mn_stream_t stream);
/* nn.CrossEntropyLoss() (mean reduction, ignore_index; the criterion of the reference's main.py, wqaq/dorefa/main.py:87-92) on logits [N][K] and int64 targets: the loss
 * AND its gradient d loss / d logits (for an incoming gradient of 1) in one launch; mn_scale_by: out = a * scalar[0] (the backward: the incoming scalar gradient). */
int mn_cross_entropy_fwd(const float* logits, const int64_t* target, int64_t N, int64_t K, int64_t ignore_index, float* loss, float* dlogits, mn_stream_t stream);
int mn_scale_by(const float* a, const float* scalar, float* out, int64_t n, mn_stream_t stream);
/* the two halves of mn_bnsign_bwd for a consumer that forms dy itself: mn_bnsign_bwd_sums = dgamma, dbeta and sums [2][C] =
 * {sum dz, sum dz*zhat}; mn_conv2d_bwd_weight_first_bn = backward-weight (+ dbias) of the first-layer convolution
 * (mn_conv2d_first_supported) whose output y went through BatchNorm2d + BinaryActivation: dy is formed from (da, y, save, gamma,
 * beta, sums) while the operands stream in -- the first layer has no backward-data, so dy is needed nowhere else. */
int mn_bnsign_bwd_sums(const float* da, const float* y, const float* save, const float* gamma, const float* beta, int64_t N, int64_t C,
                       int64_t HW, float* dgamma, float* dbeta, float* sums, float* ws, mn_stream_t stream);
int mn_conv2d_bwd_weight_first_bn(const mn_conv_geom* g, const float* da, const float* y, const float* save, const float* gamma, const float* beta,
                                  const float* sums, int training, const float* x, float* dw, float* dbias, void* ws, int64_t ws_bytes,
                                  mn_stream_t stream);
/* the same for the first block of a DoReFa net (BatchNorm2d + ReLU + the next conv's k-bit activation quantizer, mn_qa_*): dq = the gradient handed to
 * mn_qa_bwd_apply, y = the first conv's output, chan = the [9][O] constants of mn_qa_chan_from_save, sums = mn_qa_bwd_sums' output; dy (what
 * mn_qa_bwd_apply(in_f32 = 1, pool = 0) would write) is formed in registers, expression for expression. */
int mn_conv2d_bwd_weight_first_qa(const mn_conv_geom* g, const float* dq, const float* y, const float* chan, const float* sums, int a_bits, int quant,
                                  int training, const float* x, float* dw, float* dbias, void* ws, int64_t ws_bytes, mn_stream_t stream);

/* One-pass backward of the first block in training mode (round 5; replaces mn_bnsign_bwd_sums + mn_conv2d_bwd_weight_first_bn, resp. mn_qa_bwd_sums +
 * mn_conv2d_bwd_weight_first_qa: ONE pass over (da, y) instead of two).  The BatchNorm backward is linear in dz (= da with the clip-STE / ReLU / quantizer masks):
 * with f_k(pixel) the im2col row of x and f_K = 1,  A[o][k] = sum dz f_k,  S1 = sum dz,  S2 = sum dz zhat = invstd (w[o,:] . A[o,:] + (b - mean) S1),
 * B[o][k] = sum zhat f_k = invstd ((w G)[o][k] + (b - mean) P[k]),  dw = gamma invstd (A - S1 P / n - (S2 / n) B),  dgamma = S2,  dbeta = S1,
 * where G = sum f f^T and P = sum f depend on x alone: mn_conv2d_first_xgram writes them as gram [80][80] doubles (row / column K = P, gram[K][K] = n; K =
 * C KH KW <= 76).  Geometry: mn_conv2d_first_supported(g, 2); ws of the backward: mn_conv2d_ws_bytes(g, 2, 0) bytes; w, bias (nullable): the convolution's
 * parameters (torch/nn/modules/conv.py via wbwtab/quantize.py:251, dorefa/quantize.py:206: the first layer is not quantised).  dbias (nullable) = sum dy, zero
 * but for the rounding of the saved mean.  Results agree with the two-pass path to fp32 rounding (not bit for bit: a different summation order). */
/* ... and the FORWARD statistics of the BatchNorm behind that convolution from the same Gram data, without a pass over y (mn_conv2d_first_gram_bnstats: save =
 * {mean, invstd} [2][O] and the running update of nn.BatchNorm2d in training mode: mean = w . P / n + b, biased variance = w (G - P P^T / n) w^T / n); mn_bnsign_apply
 * = the apply pass of mn_bnsign_fwd (out8 = 0) / mn_bnsign_fwd_i8 (out8 = 1) with those statistics given. */
int mn_conv2d_first_gram_bnstats(const mn_conv_geom* g, const float* w, const float* bias, const double* gram, float eps, float momentum, float* running_mean,
                                 float* running_var, float* save, mn_stream_t stream);
int mn_bnsign_apply(const float* y, int64_t N, int64_t C, int64_t HW, const float* gamma, const float* beta, const float* save, void* a, int out8, mn_stream_t stream);
/* The FUSED first block (statistics known before the convolution runs, so its epilogue can normalise): y = conv(x, w) + b is never written.
 *   mn_conv2d_first_bnact_fwd   act 1: codes = int8 sign(bn(y)) (models/nin_gc.py:53-59 with wbwtab/quantize.py:11-36); act 2: codes = the a_bits DoReFa activation
 *                               code of relu(bn(y)) (wqaq/dorefa/quantize.py:36-46), uint8.  mask4 [N][O][H W / 4]: bit e of the low nibble = the backward's pass
 *                               test for pixel 4 i + e (act 1: |z| < 1, act 2: z > 0), high nibble (act 2): ... and 0 <= 0.1 relu(z) <= 1 (the quantizer's clamp).
 *   mn_conv2d_bwd_first_mask_gram  the one-pass backward on (da, mask4) instead of (da, y); quant != 0: da is the gradient w.r.t. the quantised activation (high
 *                               nibble, rows scaled by 0.1 -- the quantizer's STE).  save / gamma: the BatchNorm's, as given to the forward. */
/* the unfused forward of the DoReFa first block that leaves the same pass nibbles: mn_qa_fwd(in_f32 = 1, pool = 0, codes) + mask4 (the fused act 2 epilogue is
 * VALU-bound by the quantizer's rounding -- measured slower than conv + this pass -- so the default DoReFa first block is: conv, this, mn_conv2d_bwd_first_mask_gram) */
int mn_qa_fwd_f32_mask(const float* y, const float* chan, int64_t N, int64_t C, int64_t H, int64_t W, int a_bits, uint8_t* codes, uint8_t* mask4, mn_stream_t stream);
int mn_conv2d_first_bnact_fwd(const mn_conv_geom* g, const float* x, const float* w, const float* bias, const float* save, const float* gamma, const float* beta,
                              int act, int a_bits, void* codes, uint8_t* mask4, mn_stream_t stream);
int mn_conv2d_bwd_first_mask_gram(const mn_conv_geom* g, const float* da, const uint8_t* mask4, int quant, const float* save, const float* gamma, const float* w,
                                  const float* bias, const double* gram, const float* x, float* dw, float* dbias, float* dgamma, float* dbeta, void* ws,
                                  int64_t ws_bytes, mn_stream_t stream);
int64_t mn_conv2d_first_xgram_ws_bytes(const mn_conv_geom* g);
int mn_conv2d_first_xgram(const mn_conv_geom* g, const float* x, double* gram, void* ws, int64_t ws_bytes, mn_stream_t stream);
int mn_conv2d_bwd_first_bn_gram(const mn_conv_geom* g, const float* da, const float* y, const float* save, const float* gamma, const float* beta, const float* w,
                                const float* bias, const double* gram, const float* x, float* dw, float* dbias, float* dgamma, float* dbeta, void* ws,
                                int64_t ws_bytes, mn_stream_t stream);
int mn_conv2d_bwd_first_qa_gram(const mn_conv_geom* g, const float* dq, const float* y, const float* chan, int a_bits, int quant, const float* w, const float* bias,
                                const double* gram, const float* x, float* dw, float* dbias, float* dgamma, float* dbeta, void* ws, int64_t ws_bytes,
                                mn_stream_t stream);

/* ------------------------------------------------------------------ conv + BatchNorm2d + BinaryActivation, fused, on packed signs
 * The whole W/A-binary block of the reference -- `relu(bn(conv(x)))` with the ReLU replaced by BinaryActivation
 * (models/nin_gc.py:53-59; wbwtab/quantize.py:79-94, 181-195) -- for pointwise (1x1, stride 1) convolutions whose input is
 * already packed sign codes (`x`: int8 {-1,+1}, NCHW, as MN_ACTQ_SIGN8) and whose weights factor as codes x scale (`wq`,
 * `w` = the fake-quantised fp32 weights as for mn_conv2d_fwd).  The conv output y is NEVER written: it is recomputed on the
 * matrix cores wherever it is needed.
 *   fwd: batch statistics of y (training) or the running ones (eval) -> save [2][O] = mean, invstd; running_* updated like
 *        nn.BatchNorm2d; a = sign(bn(y)) as int8 codes [N][O][H][W].
 *   bwd: given da = d loss / d a (fp32): dy = d loss / d y (what mn_conv2d_bwd_data / _bwd_weight consume), dgamma, dbeta
 *        (nullable) -- the clip-STE of the sign through the BatchNorm backward, y recomputed from x.
 * ws: >= mn_qconv_bnsign_ws_bytes(g) bytes, 16-byte aligned.  MN_ENOTSUP if mn_qconv_bnsign_supported(g, wq) == 0. */
int mn_qconv_bnsign_supported(const mn_conv_geom* g, const mn_wq* wq);
int64_t mn_qconv_bnsign_ws_bytes(const mn_conv_geom* g);
int mn_qconv_bnsign_fwd(const mn_conv_geom* g, const mn_wq* wq, const int8_t* x, const float* w, const float* bias, const float* gamma,
                        const float* beta, float eps, float momentum, int training, float* running_mean, float* running_var,
                        float* save, int8_t* a, void* ws, int64_t ws_bytes, mn_stream_t stream);
int mn_qconv_bnsign_bwd(const mn_conv_geom* g, const mn_wq* wq, const int8_t* x, const float* w, const float* bias, const float* gamma,
                        const float* beta, const float* save, const float* da, int training, float* dy, float* dgamma, float* dbeta,
                        void* ws, int64_t ws_bytes, mn_stream_t stream);

/* forward that additionally STASHES the convolution result in one byte per element: h = (acc + nnz[o]) / 2 (uint8 [N][O][H][W]; y =
 * alpha[o]*acc + bias, acc an integer with the parity of nnz[o] = number of non-zero weight codes of channel o) and keeps the per-channel
 * constants `chan` [8][O] (thresholds of sign(z) and |z| < 1 in the integer domain, zhat = acc*A + B, gamma*invstd, nnz).  With them the
 * BatchNorm+sign backward needs neither y nor the convolution: mn_bnh_bwd_sums (dgamma, dbeta, sums [2][O]) and mn_bnh_bwd_apply (dy)
 * stream (da, h) once each; `own` != NULL: da is the gradient of the 2x2 max-pool behind the block, own = the block's output codes. */
/* num_batches_tracked (nullable, int64 on the device): nn.BatchNorm2d's forward counter (torch/nn/modules/batchnorm.py: += 1 per training
 * forward), incremented by the same launch that forms the statistics -- one tiny kernel per block less than a separate add. */
int mn_qconv_bnsign_fwd_stash(const mn_conv_geom* g, const mn_wq* wq, const int8_t* x, const float* w, const float* bias, const float* gamma,
                              const float* beta, float eps, float momentum, int training, float* running_mean, float* running_var,
                              int64_t* num_batches_tracked, float* save, int8_t* a, uint8_t* h, float* chan, void* ws, int64_t ws_bytes,
                              mn_stream_t stream);
/* ... of a pointwise block whose output goes through nn.MaxPool2d(2, 2) (models/nin_gc.py:88,119), training mode: the sign pass also writes the POOLED codes a_pool
 * [N][O][H/2][W/2] -- the window's maximum is +1 iff one of its four elements is -- so mn_maxpool2x2_sign8_fwd's pass over `a` is not launched (round 6).  H even, W a
 * multiple of 16; every other output as mn_qconv_bnsign_fwd_stash. */
int mn_qconv_bnsign_fwd_stash_pool_supported(const mn_conv_geom* g, const mn_wq* wq);
int mn_qconv_bnsign_fwd_stash_pool(const mn_conv_geom* g, const mn_wq* wq, const int8_t* x, const float* w, const float* bias, const float* gamma, const float* beta,
                                   float eps, float momentum, int training, float* running_mean, float* running_var, int64_t* num_batches_tracked, float* save,
                                   int8_t* a, int8_t* a_pool, uint8_t* h, float* chan, void* ws, int64_t ws_bytes, mn_stream_t stream);
/* The stash forward also covers k x k convolutions with ternary / binary weights on sign codes (nin_gc's grouped 3x3 layers,
 * models/nin_gc.py:88-119): the code-domain k x k kernel writes h instead of y, the batch statistics and the sign are streamed from h
 * (one byte per element each).  These two queries answer for both kinds of block; the stash forward needs the workspace they name. */
int mn_qconv_bnsign_stash_supported(const mn_conv_geom* g, const mn_wq* wq);
int64_t mn_qconv_bnsign_stash_ws_bytes(const mn_conv_geom* g);
/* rows of `chan` the stash forward fills (and mn_bnh_bwd_* read): 8 for a pointwise block; 17 for a 3x3 / padding 1 block, where the taps
 * outside the image meet zeros, so acc has the parity of a PER-PIXEL-CLASS count (top / middle / bottom row x left / middle / right
 * column): row 7 is -1 and rows 8..16 hold nnz[3 rc + cc][o]; h = (acc + nnz[class][o]) / 2 */
int mn_qconv_bnsign_stash_chan_rows(const mn_conv_geom* g);
int mn_bnh_bwd_sums(const float* da, const uint8_t* h, const int8_t* own, const float* chan, int64_t N, int64_t C, int64_t H, int64_t W,
                    float* dgamma, float* dbeta, float* sums, float* ws, mn_stream_t stream);
int mn_bnh_bwd_apply(const float* da, const uint8_t* h, const int8_t* own, const float* chan, const float* sums, int64_t N, int64_t C, int64_t H,
                     int64_t W, int training, float* dy, mn_stream_t stream);
/* the finish of mn_bnh_bwd_sums alone, on partial sums part [C][splits][2] (doubles) that the producer of d a left (mn_conv2d_bwd_bnh_up): fixed-order sum over the
 * splits -> sums [2][C], dgamma, dbeta (nullable) */
int mn_bnh_bwd_sums_final(const double* part, int32_t splits, int64_t N, int64_t C, int64_t H, int64_t W, float* dgamma, float* dbeta, float* sums,
                          mn_stream_t stream);
/* ... and the consumers of that dy can form it themselves: backward-data / backward-weight of the block's pointwise convolution whose incoming
 * gradient is the BatchNorm+sign backward of (da, h) -- evaluated in registers while da and h stream in, so dy is never written or re-read
 * (x = the block's input codes; chan, sums as above; needs mn_conv2d_bnh_supported). */
int mn_conv2d_bnh_supported(const mn_conv_geom* g, const mn_wq* wq);
int mn_conv2d_bwd_data_bnh(const mn_conv_geom* g, const mn_wq* wq, const float* da, const uint8_t* h, const float* chan, const float* sums,
                           int training, const float* w, float* dx, void* ws, int64_t ws_bytes, mn_stream_t stream);
int mn_conv2d_bwd_weight_bnh(const mn_conv_geom* g, const float* da, const uint8_t* h, const float* chan, const float* sums, int training,
                             const int8_t* x, float* dw, float* dbias, void* ws, int64_t ws_bytes, mn_stream_t stream);
/* ... and behind a block whose output goes through a 2x2 / stride-2 max-pool (models/nin_gc.py:88,119): the pool's backward is folded in as well.  dpool =
 * d loss / d (pooled output) [N][O][H/2][W/2] (8-byte aligned), own = the block's own sign output [N][O][H][W]: a window's gradient goes to its first +1 in scan
 * order (ATen's max_pool2d backward), exactly what mn_bnh_bwd_apply(own != NULL) writes -- that full-size dy is then never written or read. */
int mn_conv2d_bnh_pool_supported(const mn_conv_geom* g, const mn_wq* wq);
int mn_conv2d_bwd_data_bnh_pool(const mn_conv_geom* g, const mn_wq* wq, const float* dpool, const uint8_t* h, const int8_t* own, const float* chan,
                                const float* sums, int training, const float* w, float* dx, void* ws, int64_t ws_bytes, mn_stream_t stream);
int mn_conv2d_bwd_weight_bnh_pool(const mn_conv_geom* g, const float* dpool, const uint8_t* h, const int8_t* own, const float* chan, const float* sums,
                                  int training, const int8_t* x, float* dw, float* dbias, void* ws, int64_t ws_bytes, mn_stream_t stream);
/* BOTH gradients of such a block in ONE launch (round 6): backward-data and backward-weight of the pointwise convolution read (da, h) once and rebuild dy once
 * (wbwtab/quantize.py:11-36, 181-195 + autograd's conv backward; the two calls above read them once each).  own == NULL: da = d loss / d a [N][O][H][W]
 * (16-byte aligned); own != NULL: da is the pooled gradient as in the *_pool calls.  Covers groups of 128 -> 128 channels with H*W a multiple of 32 (every
 * pointwise layer of models/nin_gc.py): ask mn_conv2d_bwd_bnh_supported; workspace mn_conv2d_bwd_bnh_ws_bytes (16-byte aligned); dbias nullable. */
int mn_conv2d_bwd_bnh_supported(const mn_conv_geom* g, const mn_wq* wq, int pooled);
int64_t mn_conv2d_bwd_bnh_ws_bytes(const mn_conv_geom* g);
int mn_conv2d_bwd_bnh(const mn_conv_geom* g, const mn_wq* wq, const float* da, const uint8_t* h, const int8_t* own, const float* chan, const float* sums,
                      int training, const float* w, const int8_t* x, float* dx, float* dw, float* dbias, void* ws, int64_t ws_bytes, mn_stream_t stream);
/* ... and, when the block IN FRONT of this one is a BatchNorm+sign block on a pointwise conv too (wbwtab/quantize.py:11-36 behind models/nin_gc.py:18-59's
 * ConvBNReLU chain), the per-channel sums of THAT block's BatchNorm backward as a by-product: this launch's dx is that block's d a, and its dx waves accumulate
 * sum dz, sum dz zhat (dz = dx [clip-STE mask from the upstream byte stash up_h], up_chan = the upstream [8][C] constants) into up_part [C][splits][2] doubles --
 * the upstream block then calls mn_bnh_bwd_sums_final instead of mn_bnh_bwd_sums (no pass over (d a, h)).  splits: mn_conv2d_bwd_bnh_up_splits (0: not
 * covered; up_k = K of the upstream conv, <= 254).  Same masks as mn_bnh_bwd_sums; the sums agree to fp32 rounding (different summation order). */
int mn_conv2d_bwd_bnh_up_splits(const mn_conv_geom* g, const mn_wq* wq, int pooled, int64_t up_k);
int mn_conv2d_bwd_bnh_up(const mn_conv_geom* g, const mn_wq* wq, const float* da, const uint8_t* h, const int8_t* own, const float* chan, const float* sums,
                         int training, const float* w, const int8_t* x, float* dx, float* dw, float* dbias, void* ws, int64_t ws_bytes,
                         const uint8_t* up_h, const float* up_chan, double* up_part, mn_stream_t stream);
/* ... and when the block in front is a 3x3 / padding-1 BatchNorm+sign block (models/nin_gc.py:62-147: layers 4 | 5 and 7 | 8): its stash offset depends on the pixel's
 * border class, so up_chan = its [17][C] constants (rows 8..16: nnz of the nine classes, mn_qconv_bnsign_stash_chan_rows == 17) and the dx waves run mn_bnh_bwd_sums'
 * test on acc = 2 h - nnz(pixel) itself.  Un-pooled consumers, W a power of two >= 8; splits: mn_conv2d_bwd_bnh_up9_splits (0: not covered). */
int mn_conv2d_bwd_bnh_up9_splits(const mn_conv_geom* g, const mn_wq* wq, int64_t up_k);
int mn_conv2d_bwd_bnh_up9(const mn_conv_geom* g, const mn_wq* wq, const float* da, const uint8_t* h, const float* chan, const float* sums, int training,
                          const float* w, const int8_t* x, float* dx, float* dw, float* dbias, void* ws, int64_t ws_bytes, const uint8_t* up_h, const float* up_chan,
                          double* up_part, mn_stream_t stream);
/* (the k-bit counterparts of the *_up call: mn_conv2d_bwd_codes_up / mn_conv2d_bwd_qa_up below) */
/* The same one-launch backward for the k-bit (DoReFa) blocks, wqaq/dorefa/quantize.py:36-46, 107-122 + autograd's conv backward (same geometry, same two queries
 * with pooled = 0, same workspace):
 *   mn_conv2d_bwd_codes: dx (NO clip-STE: the producing block applies it where it recomputes the activation) and dw = s_x * sum gy * j, dbias from a plain fp32
 *     gradient gy [N][O][H][W]; x = the input's activation codes j (bytes) of x_bits bits -- x_bits == 0: sign codes (+-1 bytes, wbwtab), dw = sum gy * x.
 *   mn_conv2d_bwd_qa: the gradient is the BatchNorm + ReLU + next-quantizer backward of (dq, stash), formed while they stream in -- what mn_qa_bwd_apply would
 *     write as fp32 dy (un-pooled blocks): dq = d loss / d (the block's output; quant != 0: its out_bits-quantised output, the clip-STE is applied here), stash =
 *     the conv's integer accumulator (int16, stash_bits == 32: int32; mn_qconv_bnq_fwd_stash), chan [9][O], sums [2][O] (mn_qa_bwd_sums). */
int mn_conv2d_bwd_codes(const mn_conv_geom* g, const mn_wq* wq, const float* gy, const float* w, const void* x_codes, int x_bits, float* dx, float* dw, float* dbias,
                        void* ws, int64_t ws_bytes, mn_stream_t stream);
int mn_conv2d_bwd_qa(const mn_conv_geom* g, const mn_wq* wq, const float* dq, const void* stash, int stash_bits, const float* chan, const float* sums, int out_bits,
                     int quant, int training, const float* w, const uint8_t* x_codes, int x_bits, float* dx, float* dw, float* dbias, void* ws, int64_t ws_bytes,
                     mn_stream_t stream);
/* ... with the sums of the BatchNorm backward of the k-bit block IN FRONT as a by-product (round 6; the k-bit counterpart of mn_conv2d_bwd_bnh_up): this dx is that
 * block's d q, so the launch also accumulates sum dz and sum dz zhat per input channel -- dz = clip-STE(dx) under the ReLU / clamp masks of the upstream block's 16-bit
 * stash up_stash [N][C][H][W] (up_chan = its [9][C] constants, up_quant: dx is w.r.t. its x_bits-quantised output), mn_qa_bwd_sums' arithmetic element by element --
 * into up_part [C][splits][2] doubles (splits: mn_conv2d_bwd_bnh_up_splits(g, wq, 0, 1)); the upstream block then calls mn_qa_bwd_sums_final instead of
 * mn_qa_bwd_sums (no pass over (dq, stash)).  16-bit stashes on both sides (stash_bits == 16). */
int mn_conv2d_bwd_codes_up(const mn_conv_geom* g, const mn_wq* wq, const float* gy, const float* w, const void* x_codes, int x_bits, float* dx, float* dw, float* dbias,
                           void* ws, int64_t ws_bytes, const void* up_stash, const float* up_chan, int up_quant, double* up_part, mn_stream_t stream);
int mn_conv2d_bwd_qa_up(const mn_conv_geom* g, const mn_wq* wq, const float* dq, const void* stash, const float* chan, const float* sums, int out_bits, int quant,
                        int training, const float* w, const uint8_t* x_codes, int x_bits, float* dx, float* dw, float* dbias, void* ws, int64_t ws_bytes,
                        const void* up_stash, const float* up_chan, int up_quant, double* up_part, mn_stream_t stream);
/* same backward when a 2x2 / stride-2 max-pool (models/nin_gc.py:88,119) sits behind the block: `dpool` = d loss / d pool(a),
 * [N][O][H/2][W/2] fp32, `a_own` = the block's own output codes (what mn_qconv_bnsign_fwd wrote); the pool's backward (gradient to
 * the first maximum of each window) is applied while the gradient is read -- a quarter of the bytes, no full-size da tensor. */
int mn_qconv_bnsign_bwd_pooled(const mn_conv_geom* g, const mn_wq* wq, const int8_t* x, const float* w, const float* bias, const float* gamma,
                               const float* beta, const float* save, const float* dpool, const int8_t* a_own, int training, float* dy,
                               float* dgamma, float* dbeta, void* ws, int64_t ws_bytes, mn_stream_t stream);

/* ------------------------------------------------------------------ conv + BatchNorm2d + ReLU + the next layer's k-bit activation quantizer, fused
 * The block of the reference's DoReFa nets: `relu(bn(conv(x)))` (models/nin_gc.py:53-59) whose output only feeds the ActivationQuantizer of the next
 * QuantConv2d (wqaq/dorefa/quantize.py:36-46, 107-122), possibly through a 2x2 max-pool (models/nin_gc.py:88,119).  Input: activation codes j (uint8,
 * MN_ACTQ_CODE8); weights: DoReFa codes (wq->mode == MN_WQ_DOREFA, `w` = the fake-quantised fp32 weights).  The conv result y = alpha * acc + bias with
 * acc an EXACT integer is never written as fp32 (a_bits_in 2 .. 8, wq->bits 2 .. 8):
 *   mn_qconv_bnq_fwd_stash   conv on codes -> acc as a 16- / 32-bit stash (mn_qconv_bnq_stash_bits) [N][O][H][W] + exact batch statistics -> save [2][O] (mean, invstd), running
 *                            statistics / num_batches_tracked like nn.BatchNorm2d, chan [MN_QA_NCH][O] (the per-channel constants the streaming
 *                            kernels below read).  ws: mn_qconv_bnq_ws_bytes(g).
 *   mn_qa_fwd                stash (in_f32 == 0) or fp32 y (in_f32 == 1: the block behind the un-quantised first conv) -> a = relu(bn(y)) -> [2x2 max-pool]
 *                            -> codes of the a_bits quantizer (codes != NULL) and / or the fp32 activation itself (act_f32 != NULL: foreign consumers)
 *   mn_qa_bwd_sums / _apply  dq = d loss / d (pooled) activation -> dgamma, dbeta, sums [2][C]; dy [N][C][H][W] (what conv backward consumes).
 *                            quant != 0: dq is the gradient w.r.t. the QUANTISED activation and the quantizer's clip-STE is applied here;
 *                            quant == 0: dq is the gradient w.r.t. the activation (consumer applied its own STE, or is not quantised).
 * Backward of the conv itself: mn_conv2d_bwd_data (no STE epilogue) and mn_conv2d_bwd_weight with aq->mode == MN_ACTQ_CODE8. */
#define MN_QA_NCH 9
int mn_qconv_bnq_supported(const mn_conv_geom* g, const mn_wq* wq, int a_bits_in);
/* Dense layers (groups == 1, C and O multiples of 64, 3 x 3 stride 1 / 2 or 1 x 1 stride 2: the ResNets): the weight codes of every such layer of a net, in
 * both fragment orders, in ONE launch -- once per training step, right after the weight quantizer, instead of one launch per conv call and direction.
 * w[i]: the fake-quantised fp32 weights [O][C][taps]; out_fwd[i] / out_bwd[i]: mn_qd_packed_bytes(g) bytes each (either may be NULL), handed to the conv
 * entry points through mn_wq.packed_fwd / packed_bwd.  wscale == NULL: DoReFa codes rint(w (2^bits - 1)); else IAO codes rint(w / wscale[i][o * stride]). */
int64_t mn_qd_packed_bytes(const mn_conv_geom* g);
int mn_qd_pack_multi(const float* const* w, void* const* out_fwd, void* const* out_bwd, const int64_t* O, const int64_t* Cin, const int64_t* taps,
                     const float* const* wscale, const int32_t* wscale_stride, int32_t count, int w_bits, mn_stream_t stream);
/* The pointwise counterpart of mn_qd_pack_multi: the weight-code images of every pointwise layer of a net -- which[i] = 0: the forward image of the fused
 * sign / k-bit kernels (mn_qconv_bnsign_fwd*, mn_qconv_bnq_fwd_stash, mn_conv2d_fwd on sign codes), 1: the transposed image of backward-data without a clip-STE
 * epilogue (mn_conv2d_bwd_data on sign / k-bit codes, mn_conv2d_bwd_data_bnh[_pool]) -- in ONE launch per step.  out[i]: mn_qg_packed_bytes(g, which) bytes,
 * 16-byte aligned (0: this geometry is not packed ahead), handed to those entry points through mn_wq.packed_fwd / packed_bwd; `w` = the fake-quantised weights. */
int64_t mn_qg_packed_bytes(const mn_conv_geom* g, int which);
int mn_qg_pack_multi(int32_t count, const mn_conv_geom* const* g, const mn_wq* const* wq, const float* const* w, const int32_t* which, void* const* out,
                     mn_stream_t stream);
/* width of the stash mn_qconv_bnq_fwd_stash writes for this layer: 16, or 32 when K * (2^a - 1) * (2^w - 1) exceeds 32767 -- a DENSE layer (groups == 1, C and O
 * multiples of 64: the 3 x 3 stride 1 / 2 and 1 x 1 stride 2 convolutions of the reference's ResNets, models/resnet.py:7-65) or a grouped / pointwise layer of
 * nin_gc at more than 4 bits (W8A8, the reference's CPU configuration: wqaq/dorefa/main.py:135,189-190), and for every layer read through 8-bit activation codes;
 * 0 when unsupported.  A 32-bit stash is passed through the same `stash` pointer (16-byte aligned) and read by mn_qa_* / mn_qr_* with in_kind == 2.  Dense layers
 * write [N][O][Ho][Wo] (stride 2: half the input size).  The wide grouped kernels contract bf16 codes with fp32 accumulation: K * (2^a - 1) * (2^w - 1) < 2^24. */
int mn_qconv_bnq_stash_bits(const mn_conv_geom* g, const mn_wq* wq, int a_bits_in);
int64_t mn_qconv_bnq_ws_bytes(const mn_conv_geom* g);
int mn_qconv_bnq_fwd_stash(const mn_conv_geom* g, const mn_wq* wq, const uint8_t* x_codes, int a_bits_in, const float* w, const float* bias, const float* gamma,
                           const float* beta, float eps, float momentum, int training, float* running_mean, float* running_var,
                           int64_t* num_batches_tracked, float* save, int16_t* stash, float* chan, void* ws, int64_t ws_bytes, mn_stream_t stream);
/* statistics half of mn_bnrelu_fwd: save [2][C] = {mean, invstd} of fp32 y (training: batch statistics + running update like nn.BatchNorm2d; eval: the
 * running statistics); ws: mn_bnsign_ws_floats(C) */
int mn_bn_save_stats(const float* y, int64_t N, int64_t C, int64_t HW, float eps, float momentum, int training, float* running_mean, float* running_var,
                     float* save, float* ws, mn_stream_t stream);
int mn_qa_supported(int64_t H, int64_t W, int pool);
int64_t mn_qa_ws_floats(int64_t C);
/* chan from the (mean, invstd) a BatchNorm over fp32 y saved (mn_bnrelu_fwd's `save`): the first block of a net */
int mn_qa_chan_from_save(const float* save, const float* gamma, const float* beta, int64_t C, float* chan, mn_stream_t stream);
/* in_f32: 0 = the int16 stash, 1 = fp32 y, 2 = the int32 stash (mn_qconv_bnq_stash_bits) */
int mn_qa_fwd(int in_f32, const void* in, const float* chan, int64_t N, int64_t C, int64_t H, int64_t W, int a_bits, int pool, uint8_t* codes, float* act_f32,
              mn_stream_t stream);
int mn_qa_bwd_sums(int in_f32, const void* in, const float* chan, const float* dq, int64_t N, int64_t C, int64_t H, int64_t W, int a_bits, int pool, int quant,
                   float* dgamma, float* dbeta, float* sums, float* ws, mn_stream_t stream);
/* its finish alone, on partial sums part [C][splits][2] (doubles) the producer of dq left (mn_conv2d_bwd_qa_up / mn_conv2d_bwd_codes_up) */
int mn_qa_bwd_sums_final(const double* part, int32_t splits, int64_t C, float* dgamma, float* dbeta, float* sums, mn_stream_t stream);
int mn_qa_bwd_apply(int in_f32, const void* in, const float* chan, const float* sums, const float* dq, int64_t N, int64_t C, int64_t H, int64_t W, int a_bits,
                    int pool, int quant, int training, float* dy, mn_stream_t stream);
/* the two calls above as TWO launches instead of three: the apply pass sums the partial rows itself (same order: bit-identical statistics) and writes dgamma, dbeta, sums */
int mn_qa_bwd(int in_f32, const void* in, const float* chan, const float* dq, int64_t N, int64_t C, int64_t H, int64_t W, int a_bits, int pool, int quant, int training,
              float* dgamma, float* dbeta, float* sums, float* dy, float* ws, mn_stream_t stream);

/* ------------------------------------------------------------------ the END of a residual block (k-bit DoReFa ResNets)
 * models/resnet.py:60-65 -- relu(add(residual_function(x), shortcut(x))) -- whose residual branch ends in a dense QuantConv2d + BatchNorm2d
 * (wqaq/dorefa/quantize.py:107-122) and whose consumers are the next block's QuantConv2d(s) and / or its identity shortcut:
 *     u = bn(y) + res,   a = relu(u)   ->   codes of the a_bits quantizer (codes != NULL) and / or the fp32 activation (act_f32 != NULL), ONE pass
 *   in_kind   0: y is the int16 stash of mn_qconv_bnq_fwd_stash, 2: its int32 stash (mn_qconv_bnq_stash_bits), 1: fp32 y; chan as for mn_qa_*
 *   res_kind  0: no residual (a block that must emit codes AND fp32 at once: the stem), 1: `res` = fp32 [N][C][H][W] (identity shortcut),
 *             2 / 3: `res` = the int16 / int32 stash of the shortcut conv, res_chan = its chan table: res = bn_s(y_s) is evaluated on the fly
 *   mn_qr_bwd_sums   du = (STE(dq) [+ STE(dq2)] [+ g_f32]) * [u > 0] -> du [N][C][H][W] (this IS the gradient of an identity shortcut), dgamma / dbeta /
 *                    sums [2][C] of the main BatchNorm and (res_kind >= 2) dgamma_s / dbeta_s / sums_s of the shortcut's.  dq, dq2: gradients w.r.t. the
 *                    QUANTISED activation from the convs that read the codes (clip-STE of wqaq/dorefa/quantize.py:36-46 applied here, per consumer, as
 *                    autograd does); g_f32: gradient w.r.t. the activation itself.  ws: mn_qr_ws_floats(C) floats.
 *   mn_qr_bwd_apply  dy = gamma invstd (du - sum_du / n - zhat sum_du_zhat / n) for the residual branch's conv and (res_kind >= 2) dy_s for the shortcut conv. */
int64_t mn_qr_ws_floats(int64_t C);
int mn_qr_fwd(int in_kind, const void* in, const float* chan, int res_kind, const void* res, const float* res_chan, int64_t N, int64_t C, int64_t H, int64_t W,
              int a_bits, uint8_t* codes, float* act_f32, mn_stream_t stream);
int mn_qr_bwd_sums(int in_kind, const void* in, const float* chan, int res_kind, const void* res, const float* res_chan, const float* dq, const float* dq2,
                   const float* g_f32, int64_t N, int64_t C, int64_t H, int64_t W, int a_bits, float* du, float* dgamma, float* dbeta, float* sums,
                   float* dgamma_s, float* dbeta_s, float* sums_s, float* ws, mn_stream_t stream);
int mn_qr_bwd_apply(int in_kind, const void* in, const float* chan, const float* sums, int res_kind, const void* res, const float* res_chan, const float* sums_s,
                    const float* du, int64_t N, int64_t C, int64_t H, int64_t W, int training, float* dy, float* dy_s, mn_stream_t stream);
int mn_qr_bwd(int in_kind, const void* in, const float* chan, int res_kind, const void* res, const float* res_chan, const float* dq, const float* dq2, const float* g_f32,
              int64_t N, int64_t C, int64_t H, int64_t W, int a_bits, int training, float* du, float* dgamma, float* dbeta, float* sums, float* dgamma_s, float* dbeta_s,
              float* sums_s, float* dy, float* dy_s, float* ws, mn_stream_t stream);          /* mn_qr_bwd_sums + mn_qr_bwd_apply, the final sums inside the apply pass */

/* ------------------------------------------------------------------ QuantLinear with few outputs (O <= 64: the classifier of the ResNets, 512 -> 10)
 * y[n][o] = bias[o] + sum_c Q_a(x[n][c]) * w[o][c]  (wqaq/dorefa/quantize.py:192-199, wqaq/iao/quantize.py:1150-1157; F.linear call sites dorefa 198, iao 1156):
 * x [N][C] fp32, `w` = the already fake-quantised weight [O][C], aq = the activation quantizer evaluated in registers (MN_ACTQ_NONE / _DOREFA / _IAO);
 * bwd_data applies its clip-STE (x required unless MN_ACTQ_NONE); bwd_weight sums over n in a fixed order (deterministic), dbias may be NULL. */
int mn_qlinear_supported(int64_t N, int64_t C, int64_t O);
int mn_qlinear_fwd(const mn_actq* aq, const float* x, const float* w, const float* bias, float* y, int64_t N, int64_t C, int64_t O, mn_stream_t stream);
int mn_qlinear_bwd_data(const mn_actq* aq, const float* gy, const float* w, const float* x, float* dx, int64_t N, int64_t C, int64_t O, mn_stream_t stream);
int mn_qlinear_bwd_weight(const mn_actq* aq, const float* gy, const float* x, float* dw, float* dbias, int64_t N, int64_t C, int64_t O, mn_stream_t stream);

/* ------------------------------------------------------------------ classifier conv of a binary net
 * The LAST conv of the WbWtAb nets keeps fp32 weights (the rewrite skips it, wbwtab/quantize.py:251) but reads the +-1 output of the
 * previous block (models/nin_gc.py: 1024 -> 10, 1x1).  With O <= 16 this is a per-pixel dot product over C sign codes:
 *   fwd:      y[n][o][p] = bias[o] + sum_c w[o][c] * a[n][c][p]        (a: int8 codes [N][C][HW], w: [O][C], y: [N][O][HW] fp32)
 *   bwd_data: dx[n][c][p] = sum_o w[o][c] * gy[n][o][p]
 * (backward-weight is mn_conv2d_bwd_weight with MN_ACTQ_SIGN8).  Needs O <= 16, HW % 4 == 0. */
int mn_signconv1x1_small_supported(int64_t C, int64_t HW, int64_t O);
int mn_signconv1x1_small_fwd(const int8_t* a, const float* w, const float* bias, float* y, int64_t N, int64_t C, int64_t HW, int64_t O, mn_stream_t stream);
int mn_conv1x1_small_bwd_data(const float* gy, const float* w, float* dx, int64_t N, int64_t C, int64_t HW, int64_t O, mn_stream_t stream);
/* The same layer in a DoReFa net (wqaq/dorefa/quantize.py:107-122; every conv but the first is quantised, so the classifier reads the k-bit
 * activation codes of the block in front): codes uint8 j in [0, 2^a_bits - 1], w the fake-quantised weights:
 *   y[n][o][p] = bias[o] + s * sum_c w[o][c] * j[n][c][p],   s = 1 / (2^a_bits - 1)
 * backward-data is mn_conv1x1_small_bwd_data (gradient w.r.t. the QUANTISED activation), backward-weight mn_conv2d_bwd_weight with MN_ACTQ_CODE8. */
int mn_codeconv1x1_small_fwd(const uint8_t* codes, int a_bits, const float* w, const float* bias, float* y, int64_t N, int64_t C, int64_t HW, int64_t O,
                             mn_stream_t stream);

/* BatchNorm folding of QuantBNFuseConv2d (wqaq/iao/quantize.py:900-956) in one launch (forward) / one launch (backward):
 *   w_f[o][:] = w[o][:] * (gamma[o] / sqrt(var_w[o] + eps));   bias_f[o] = beta[o] + (bias[o] - mean[o]) * (gamma[o] / sqrt(var_b[o] + eps))
 * (bias NULL: beta - mean * k_b).  w: [O][K] (K = Cin/g * KH * KW).  Backward: any output pointer may be NULL; dvar_b / dvar_w are returned separately
 * (the caller adds them when both are the batch variance). */
int mn_iao_bnfold_fwd(const float* w, const float* bias, const float* gamma, const float* beta, const float* mean, const float* var_b, const float* var_w,
                      float eps, int64_t O, int64_t K, float* wf, float* bf, mn_stream_t stream);
int mn_iao_bnfold_bwd(const float* dwf, const float* dbf, const float* w, const float* bias, const float* gamma, const float* mean, const float* var_b,
                      const float* var_w, float eps, int64_t O, int64_t K, float* dw, float* dbias, float* dgamma, float* dbeta, float* dmean,
                      float* dvar_b, float* dvar_w, mn_stream_t stream);

/* ------------------------------------------------------------------ QuantBNFuseConv2d in training mode without the statistics convolution
 * wqaq/iao/quantize.py:837-994 for POINTWISE grouped layers (1 x 1, stride 1, <= 128 channels per group; g->in_shuffle honoured).  The raw convolution of
 * 843-851 is linear, and the block needs only the per-channel mean / unbiased variance of its output (853-855) and their gradient; both follow from the
 * per-channel sums sx[c] = sum_p x[c,p] and the Gram matrix gram[g][c][c'] = sum_p x[c,p] x[c',p] of each group's input channels (logical, i.e. post-shuffle,
 * channel order), accumulated on the matrix cores in fp32 over <= 2048 pixels per partial and combined in fp64:
 *   mn_iaobf_gram      x -> gram [G][Cg][Cg], sx [G * Cg] (fp64).  Also the FIRST layer of a net (k x k, stride 1, "same" padding, groups 1, Cg := Cin * KH * KW <= 128,
 *                      W % 4 == 0; nin_gc: 5 x 5 on RGB): the Gram matrix of the im2col matrix, gathered from the image without materialising it (G = 1).
 *   mn_iaobf_gram_stats  statistics of the raw convolution's output from the Gram data, no pass over any activation: stats [2][O] = mean[o] = W[o,:] . x_bar + b[o]
 *                      and the unbiased var[o] = W[o,:] S W[o,:]^T / (n - 1) (S = gram - n x_bar x_bar^T), plus vc [O][Cg] = W S for the backward.
 *   mn_iaobf_prep_fwd  ONE launch for: running statistics from stats_in [2][O] (856-879; first_bn: the copy of the first forward of a non-pretrained net; stats_in =
 *                      mn_iaobf_gram_stats' output, or mn_bn_stats_fwd of a materialised raw output for other geometries), the fold w_f = w * gamma / sqrt(var + eps),
 *                      bias_f = beta + (bias - mean) * ... (881-901), the per-channel weight observer + update_qparams + fake-quant (945 with 15-36 / 62-74 / 101-113,
 *                      293-321, 227-239).  Outputs: stats [2][O] = mean, var; kfold [O]; bias_f [O]; qw [O][K] = the fake-quantised folded weights; qp [O][4].
 *   mn_iaobf_prep_bwd  ONE launch for: the weight quantizer's clip-STE on dwq (gradient w.r.t. qw), the fold's backward (dgamma, dbeta, dbias), dmean / dvar and
 *                      coef [4][O] = {dmean / n, 2 dvar / (n - 1), dmean, dvar}; with vc != NULL (pointwise: mn_iaobf_gram_stats' vc, and sx) also the raw
 *                      convolution's weight gradient dmean x_bar + coef1 vc, so that dw is complete; with vc == NULL dw holds the quantised path only and the caller
 *                      adds the raw conv's backward-weight of d y_raw = coef0 + coef1 (y - mean).
 *   mn_iaobf_bwd_data  dx = STE_x(W_q^T gy) + W^T d y_raw, the second term evaluated as M (x - x_bar) + v with M = W^T diag(coef1) W inside the same kernel
 *                      (no raw convolution output exists); relu_mask != 0: x is the output of a ReLU whose backward mask [x > 0] is applied to dx here.
 *                      gy must already carry the block's own ReLU mask.  aq: the activation quantizer snapshot (MN_ACTQ_IAO); wqp = qp of prep_fwd. */
int mn_iaobf_gram_supported(const mn_conv_geom* g);
int64_t mn_iaobf_gram_ws_bytes(const mn_conv_geom* g);
int mn_iaobf_gram(const mn_conv_geom* g, const float* x, double* gram, double* sx, void* ws, int64_t ws_bytes, mn_stream_t stream);
int mn_iaobf_gram_stats(const float* w, const float* bias, const double* gram, const double* sx, int64_t O, int64_t Cg, int64_t groups, double n, float* stats, float* vc,
                        mn_stream_t stream);
int mn_iaobf_prep_fwd(const float* w, const float* bias, const float* gamma, const float* beta, int64_t O, int64_t K, const float* stats_in, float eps, float momentum,
                      int first_bn, float* running_mean, float* running_var,
                      int w_bits, int w_qtype, int w_obs_kind, int first_w, double momentum_w, float* wmin, float* wmax, float* wscale, float* wzp,
                      float* stats, float* kfold, float* bias_f, float* qw, float* qp, mn_stream_t stream);
int mn_iaobf_prep_bwd(const float* dwq, const float* dbf, const float* w, const float* bias, const float* gamma, const float* stats, const float* qp,
                      int64_t O, int64_t K, int64_t groups, const float* vc, const double* sx, double n, float eps, int w_bits, int w_qtype, float* dw,
                      float* dbias, float* dgamma, float* dbeta, float* coef, mn_stream_t stream);
int mn_iaobf_bwd_data_supported(const mn_conv_geom* g);
int64_t mn_iaobf_bwd_data_ws_bytes(const mn_conv_geom* g);
int mn_iaobf_bwd_data(const mn_conv_geom* g, const mn_actq* aq, const float* gy, const float* x, const float* w, const float* qw, const float* wqp,
                      const float* coef, const double* sx, int relu_mask, float* dx, void* ws, int64_t ws_bytes, mn_stream_t stream);

/* The same block for the grouped 3 x 3 layers of nin_gc (models/nin_gc.py:74-79: 3 x 3 / stride 1 / padding 1, 16 input and 32 output channels per group, 8 x 8 or
 * 16 x 16 maps, N * H * W a multiple of 256) whose input lies on a symmetric <= 8-bit quantizer grid (the output of QuantMaxPool2d, wqaq/iao/quantize.py:1347-1359:
 * every value = code * xgrid[0]) -- csrc/iao_g3.hip, persistent image-resident kernels.  The reference's dataflow (raw conv 843-851 -> batch statistics 853-855 -> fold
 * -> quantised conv 947-955, and its autograd) with the raw output never written in the forward:
 *   mn_iaobf_g3_stats       stats[2][O] = batch mean / unbiased variance of conv2d(x, w, bias)                                  (feeds mn_iaobf_prep_fwd's stats_in)
 *   mn_iaobf_g3_fwd         out = [relu](conv2d(Q_a(x), qw, bias_f)) + the (min, max) partials of out (mm: 2 * mn_iaobf_g3_mm_count floats, nullable)
 *   mn_iaobf_g3_bwd_weight  dw (+)= xqp[0] * conv2d_backward_weight(a * [mask > 0], codes_xqp(x)); dbias = sum of the masked a (nullable).  Called twice: (a = d out,
 *                           mask = the block's rectified output or NULL when the consumer already masked, xqp = the activation quantizer) and (a = d y_raw,
 *                           xqp = the input's grid, accumulate = 1 into mn_iaobf_prep_bwd's dw)
 *   mn_iaobf_g3_dyraw       dy = coef[0][o] + coef[1][o] (conv2d(x, w, bias) - stats[0][o])   (coef of mn_iaobf_prep_bwd: the gradient of the batch statistics)
 *   mn_iaobf_g3_bwd_data    dx = clip-STE_a(conv2d_backward_data(gy * [mask > 0], qw)) + conv2d_backward_data(dy, w)   [* [x > 0] when relu_in]
 * g->in_shuffle is honoured (x read / dx written through the channel shuffle).  ws: mn_iaobf_g3_ws_bytes(g) bytes. */
int mn_iaobf_g3_supported(const mn_conv_geom* g);
int64_t mn_iaobf_g3_ws_bytes(const mn_conv_geom* g);
int64_t mn_iaobf_g3_mm_count(const mn_conv_geom* g);
int mn_iaobf_g3_stats(const mn_conv_geom* g, const float* x, const float* xgrid, int grid_bits, const float* w, const float* bias, float* stats, void* ws,
                      int64_t ws_bytes, mn_stream_t stream);
int mn_iaobf_g3_fwd(const mn_conv_geom* g, const float* x, const float* aqp, int a_bits, const float* qw, const float* wqp, const float* bias_f, int relu, float* out,
                    float* mm, mn_stream_t stream);
int mn_iaobf_g3_dyraw(const mn_conv_geom* g, const float* x, const float* xgrid, int grid_bits, const float* w, const float* bias, const float* stats, const float* coef,
                      float* dy, mn_stream_t stream);
int mn_iaobf_g3_bwd_weight(const mn_conv_geom* g, const float* a, const float* mask, const float* x, const float* xqp, int x_bits, int accumulate, float* dw,
                           float* dbias, void* ws, int64_t ws_bytes, mn_stream_t stream);
int mn_iaobf_g3_bwd_data(const mn_conv_geom* g, const float* gy, const float* mask, const float* dy, const float* x, const float* aqp, int a_bits, const float* qw,
                         const float* wqp, const float* w, int relu_in, float* dx, mn_stream_t stream);

/* The same block for a pointwise layer with a THIN output (1 x 1 / stride 1 / groups 1, O <= 16, C a multiple of 64: the classifier layer of nin_gc,
 * models/nin_gc.py:83, 1024 -> 10) -- csrc/iao_thin.hip, HBM-bound streaming kernels on the vector units (20 flop per input byte: not matrix-core work).  The
 * caller runs the reference's dataflow: raw conv (aqp = NULL) -> mn_bn_stats_fwd -> mn_iaobf_prep_fwd -> quantised conv (aqp = the symmetric activation quantizer's
 * {scale, zp, lo, hi}); backward: both backward-weights (accumulate = 1 for the second), the two-path backward-data.  wt / qwt: mn_iaobf_thin_pack of w / qw
 * ([C][16], transposed and zero-padded).  g->in_shuffle is honoured. */
int mn_iaobf_thin_supported(const mn_conv_geom* g);
int64_t mn_iaobf_thin_mm_count(const mn_conv_geom* g);
int mn_iaobf_thin_pack(const float* w, int64_t O, int64_t C, float* wt, mn_stream_t stream);
int mn_iaobf_thin_fwd(const mn_conv_geom* g, const float* x, const float* aqp, int a_bits, const float* wt, const float* bias, int relu, float* out, float* mm,
                      mn_stream_t stream);
int mn_iaobf_thin_bwd_weight(const mn_conv_geom* g, const float* a, const float* x, const float* aqp, int a_bits, int accumulate, float* dw, float* dbias,
                             mn_stream_t stream);
int mn_iaobf_thin_bwd_data(const mn_conv_geom* g, const float* gy, const float* dy, const float* x, const float* aqp, int a_bits, const float* qwt, const float* wt,
                           int relu_in, float* dx, mn_stream_t stream);

/* ------------------------------------------------------------------ input pipeline of the training loop
 * <scheme>/main.py:203-210: transforms.Compose([RandomCrop(32, padding=4), RandomHorizontalFlip(), ToTensor(), Normalize(mean, std)]) applied to a batch
 * gathered from the uint8 dataset resident in device memory.  images: uint8 [n_images][H][W][C] (HWC, torchvision's CIFAR10.data); index [B]: the
 * sample of each output image; ox / oy [B]: crop offsets in [0, 2 * pad]; flip [B]: 0 / 1; mean / std: HOST arrays of C floats;
 * out: fp32 [B][C][H][W] = ((pixel / 255) - mean[c]) / std[c] with pixel = 0 in the padding -- bit-identical to the CPU transforms for the same draws. */
int mn_cifar_augment(const uint8_t* images, int64_t n_images, const int32_t* index, const int32_t* ox, const int32_t* oy, const uint8_t* flip, int64_t B,
                     int64_t H, int64_t W, int64_t C, int pad, const float* mean, const float* std, float* out, mn_stream_t stream);

/* ------------------------------------------------------------------ optimizer step of the training loop
 * <scheme>/main.py: optimizer.step() with torch.optim.Adam, one parameter group per tensor (wqaq/dorefa/main.py:308-315).
 * One launch per MN_ADAM_MAX_TENSORS tensors; `tensors` is a HOST array (device pointers inside), copied into the kernel
 * arguments.  Math = torch.optim.Adam (amsgrad off, L2 weight decay added to the gradient). `step` counts from 1. */
#define MN_ADAM_MAX_TENSORS 32
typedef struct mn_adam_tensor {
    float* p;          /* parameter, updated in place */
    const float* g;    /* gradient */
    float* m;          /* exp_avg, updated in place */
    float* v;          /* exp_avg_sq, updated in place */
    int64_t n;         /* elements */
    float lr, weight_decay;
} mn_adam_tensor;
int mn_adam_step(const mn_adam_tensor* tensors, int count, int step, float beta1, float beta2, float eps, mn_stream_t stream);
/* same update, but the step count (>= 1) is read from DEVICE memory by the kernel: the launch can then be captured in a HIP
 * graph and replayed while a device-side counter advances (the caller increments *step_dev before the launch).
 * hyper_dev (nullable): device array [count][2] = {lr, weight_decay} per tensor, read by the kernel INSTEAD of tensors[i].lr /
 * .weight_decay -- a replayed graph then follows the training loop's learning-rate schedule (the reference edits
 * param_group['lr'] every epoch: wbwtab/main.py:62-66 adjust_learning_rate) by refreshing that small array between replays. */
int mn_adam_step_dev(const mn_adam_tensor* tensors, int count, const int32_t* step_dev, const float* hyper_dev, float beta1, float beta2, float eps,
                     mn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MICRONET_HIP_H */
