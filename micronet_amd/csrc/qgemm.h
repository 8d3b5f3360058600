// Internal interface between the C-ABI entry points (conv_kernels.hip) and the code-domain kernels (qgemm_kernels.hip).
#pragma once
#include "common.h"

// which: 0 fwd, 1 bwd_data, 2 bwd_weight.  All return MN_OK / negative code; *_supported return 0/1 and never set the error.
int qg_supported(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, int which);
int64_t qg_ws_bytes(const mn_conv_geom* g, int which);
int qg_fwd(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, const float* x, const float* w, const float* bias, float* y,
           void* ws, int64_t ws_bytes, hipStream_t s);
int qg_bwd_data(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, const float* gy, const float* w, const float* x, float* dx,
                void* ws, int64_t ws_bytes, hipStream_t s);
// pointwise forward with the ReLU behind the conv and per-wave (min, max) partials of the result in the epilogue (mm: 2 * qg_fwd_act_mm_count(g) floats, nullable)
int qg_fwd_act_mm_count(const mn_conv_geom* g);
int qg_fwd_act(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, const float* x, const float* w, const float* bias, float* y, int relu, float* mm,
               void* ws, int64_t ws_bytes, hipStream_t s);
// first-layer convolution, real fp32 operands, K = Cin*KH*KW <= 76 (conv_first.hip); which: 0 fwd, 2 bwd_weight
int c1_supported(const mn_conv_geom* g, int which);
int64_t c1_ws_bytes(const mn_conv_geom* g, int which);
int c1_fwd(const mn_conv_geom* g, const float* x, const float* w, const float* bias, float* y, void* ws, int64_t ws_bytes, hipStream_t s);
// ... with the block's ReLU and per-block (min, max) partials of the stored result (mm: 2 * c1_fwd_mm_count(g) floats, nullable)
int c1_fwd_mm_count(const mn_conv_geom* g);
int c1_fwd_act(const mn_conv_geom* g, const float* x, const float* w, const float* bias, float* y, int relu, float* mm, void* ws, int64_t ws_bytes, hipStream_t s);
int c1_bwd_weight_bn(const mn_conv_geom* g, const float* gy, const float* da, const float* yb, const float* save, const float* gamma, const float* beta,
                     const float* sums, int training, const float* x, float* dw, float* dbias, void* ws, int64_t ws_bytes, hipStream_t s);
int c1_bwd_weight(const mn_conv_geom* g, const float* gy, const float* x, float* dw, float* dbias, void* ws, int64_t ws_bytes, hipStream_t s);
int c1_bwd_weight_qa(const mn_conv_geom* g, const float* dq, const float* yb, const float* chan, int quant, int a_bits, const float* sums, int training,
                     const float* x, float* dw, float* dbias, void* ws, int64_t ws_bytes, hipStream_t s);
// dense IAO layers (qgemm_dense.hip): the backward-data that adds mn_actq.dx_add in its store, reached directly from mn_conv2d_bwd_data
int qd_iao_dx_add_supported(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq);
int64_t qd_iao_ws_bytes(const mn_conv_geom* g, int which);
int qd_iao_bwd_data(const mn_conv_geom* g, const mn_actq* aq, const mn_wq* wq, const float* gy, const float* w, const float* x, float* dx, void* ws, int64_t ws_bytes,
                    hipStream_t s);
// one-pass backward of the first block: Gram data of x's im2col rows (gram: 80 x 80 doubles), then dz-only backward-weight + per-channel finish
int64_t c1_xgram_ws_bytes(const mn_conv_geom* g);
int c1_xgram(const mn_conv_geom* g, const float* x, double* gram, void* ws, int64_t ws_bytes, hipStream_t s);
int c1_fwd_bnact(const mn_conv_geom* g, const float* x, const float* w, const float* bias, const float* save, const float* gamma, const float* beta, int act, int a_bits,
                 void* codes, uint8_t* mask4, hipStream_t s);
int c1_bwd_first_mask(const mn_conv_geom* g, const float* da, const uint8_t* mask4, int quant, const float* save, const float* gamma, const float* w, const float* bias,
                      const double* gram, const float* x, float* dw, float* dbias, float* dgamma, float* dbeta, void* ws, int64_t ws_bytes, hipStream_t s);
int c1_gram_bnstats(const mn_conv_geom* g, const float* w, const float* bias, const double* gram, float eps, float momentum, float* running_mean, float* running_var,
                    float* save, hipStream_t s);
int c1_bwd_first_gram(const mn_conv_geom* g, const float* da, const float* yb, const float* save, const float* gamma, const float* beta, const float* chan, int quant,
                      int a_bits, const float* w, const float* bias, const double* gram, const float* x, float* dw, float* dbias, float* dgamma, float* dbeta, void* ws,
                      int64_t ws_bytes, hipStream_t s);
// pointwise convolution on int8 sign codes, fused BatchNorm + sign epilogues (qgemm_sign.hip)
int pws_supported(const mn_conv_geom* g, const mn_wq* wq);
int64_t pws_ws_bytes(const mn_conv_geom* g);
int pws_fwd(const mn_conv_geom* g, const mn_wq* wq, const int8_t* x, const float* w, const float* bias, float* y, void* ws, int64_t ws_bytes, hipStream_t s);
int pws_wgrad_supported(const mn_conv_geom* g);
int64_t pws_wgrad_ws_bytes(const mn_conv_geom* g);
int pws_bwd_weight(const mn_conv_geom* g, const float* gy, const int8_t* x, float* dw, float* dbias, void* ws, int64_t ws_bytes, hipStream_t s);
// BatchNorm+sign backward folded into the consumers of dy (pointwise, sign-code activations): see k_pwd / k_pws_wgrad
int pwd_supported(const mn_conv_geom* g, const mn_wq* wq);
int64_t pwd_ws_bytes(const mn_conv_geom* g);
int pwd_bwd_data_bnh(const mn_conv_geom* g, const mn_wq* wq, const float* da, const uint8_t* h, const float* chan, const float* sums, int training,
                     const float* w, float* dx, void* ws, int64_t ws_bytes, hipStream_t s, const int8_t* own = nullptr);
int pws_bwd_weight_bnh(const mn_conv_geom* g, const float* da, const uint8_t* h, const float* chan, const float* sums, int training, const int8_t* x,
                       float* dw, float* dbias, void* ws, int64_t ws_bytes, hipStream_t s, const int8_t* own = nullptr);
int pws_wgrad_staged(const mn_conv_geom* g);          // the LDS-staged backward-weight kernel covers this geometry (the pooled BatchNorm fold lives there only)
int qg_bwd_weight(const mn_conv_geom* g, const mn_actq* aq, const float* gy, const float* x, float* dw, float* dbias, void* ws,
                  int64_t ws_bytes, hipStream_t s);
// backward-data AND backward-weight of a pointwise binary block in one kernel (qgemm_pwb.hip): (da, h) read once
int pwb_supported(const mn_conv_geom* g, const mn_wq* wq, int pooled);
int64_t pwb_ws_bytes(const mn_conv_geom* g);
int pwb_up_splits(const mn_conv_geom* g);
int pwb_bwd_bnh_up(const mn_conv_geom* g, const mn_wq* wq, const float* da, const uint8_t* h, const int8_t* own, const float* chan, const float* sums, int training,
                   const float* w, const int8_t* x, float* dx, float* dw, float* dbias, void* ws, int64_t ws_bytes, const uint8_t* up_h, const float* up_chan,
                   double* up_part, hipStream_t s);
int pwb_up9_splits(const mn_conv_geom* g);
int pwb_bwd_bnh_up9(const mn_conv_geom* g, const mn_wq* wq, const float* da, const uint8_t* h, const float* chan, const float* sums, int training, const float* w,
                    const int8_t* x, float* dx, float* dw, float* dbias, void* ws, int64_t ws_bytes, const uint8_t* up_h, const float* up_chan, double* up_part,
                    hipStream_t s);
int pwb_bwd_plain_up(const mn_conv_geom* g, const mn_wq* wq, const float* gy, const float* w, const void* x, int x_bits, float* dx, float* dw, float* dbias, void* ws,
                     int64_t ws_bytes, const void* up_stash, const float* up_chan, int up_quant, double* up_part, hipStream_t s);
int pwb_bwd_qa_up(const mn_conv_geom* g, const mn_wq* wq, const float* dq, const void* stash, const float* chan, const float* sums, int out_bits, int quant, int training,
                  const float* w, const uint8_t* x, int x_bits, float* dx, float* dw, float* dbias, void* ws, int64_t ws_bytes, const void* up_stash, const float* up_chan,
                  int up_quant, double* up_part, hipStream_t s);
int pwb_bwd_bnh(const mn_conv_geom* g, const mn_wq* wq, const float* da, const uint8_t* h, const int8_t* own, const float* chan, const float* sums, int training,
                const float* w, const int8_t* x, float* dx, float* dw, float* dbias, void* ws, int64_t ws_bytes, hipStream_t s);
int pwb_bwd_plain(const mn_conv_geom* g, const mn_wq* wq, const float* gy, const float* w, const void* x, int x_bits, float* dx, float* dw, float* dbias, void* ws,
                  int64_t ws_bytes, hipStream_t s);
int pwb_bwd_qa(const mn_conv_geom* g, const mn_wq* wq, const float* dq, const void* stash, int stash_bits, const float* chan, const float* sums, int out_bits, int quant,
               int training, const float* w, const uint8_t* x, int x_bits, float* dx, float* dw, float* dbias, void* ws, int64_t ws_bytes, hipStream_t s);
