// Internal (C++ linkage) entry points of the backward pass with the element type stated: UNCL_BF16 (training) or UNCL_F32 (the
// fp32 parity mode of uncl_gen_backward).  The exported uncl_* functions of the same names are their bf16 forms.
#pragma once
#include "common.h"

int bwd_outc_backward(int dtype, const float* g_out, const float* x_out, const void* g_upx, const void* up_x, const float* w, void* G_up, float* gw, float* gb, long long P, int last_act, float slope, int accumulate, void* workspace /* 1024*33 floats */, void* stream);
int bwd_ssr_backward(int dtype, const void* g_cat, const void* x2, void* G_x2, void* G_x1, int N, int H, int W, int C, int H1, int W1, float slope, int accumulate_x2, void* stream);
int bwd_pool_backward(int dtype, const void* g_pool, const void* x, void* G_x, int N, int H, int W, int C, float slope, int accumulate, void* stream);
int bwd_upconv2x2_dgrad_handoff(const void* gy, const void* wt, const void* mask, float slope, void* gx, int N, int H, int W, int Cin, int Cout, const void* carry_in, void* carry_out, int pc, void* stream);
int bwd_pool_backward_handoff(int dtype, const void* g_pool, const void* x, void* G_x, int N, int H, int W, int C, float slope, int accumulate, const void* carry_in, void* carry_out, int pc, void* stream);
int bwd_gelu_forward(int dtype, const void* z, void* h, long long n, void* stream);
int bwd_gelu_backward(int dtype, const void* g_h, const void* z, void* g_z, long long n, void* stream);
int bwd_scale_rows(int dtype, const void* x, const float* scale, void* y, int N, long long per, void* stream);
int bwd_mask_minus(int dtype, const void* g, const void* x, const void* pe, void* out, int N, long long per, float slope, void* stream);
int bwd_sum_samples(int dtype, const void* g, float* out, int N, long long per, int accumulate, void* stream);
int bwd_head_handoff(int dtype, void* g, const void* mask, float slope, const void* carry_in, void* carry_out, long long npix, int C, int prev_ch, void* stream);
int bwd_mix_heads(int dtype, const void* x, const void* prev, void* out, long long npix, int C, int prev_ch, void* stream);
int bwd_mix_heads_clip(int dtype, const void* x, void* out, long long npix, long long frame_pix, int C, int prev_ch, void* stream);
int bwd_gcn_maxrel_backward(int dtype, const void* g_out, const void* x, const int32_t* idx, float* g_x_f32, void* g_x_bf16, int N, int n, int C, int k, void* stream);
int bwd_conv_in_c1_wgrad(int dtype, const void* G, const float* x, float* gw, float* gb, int N, int H, int W, int accumulate, void* workspace, void* stream);

// conv_igemm.hip: bf16 1x1 convolution with the graph block's GELU fused into its store (gmode 1 forward + pre-activation to zbuf,
// 2 data gradient x gelu'(zbuf)); > 0 (UNCL_GELU_NOT_FUSED) = not taken, run the two launches
constexpr int UNCL_GELU_NOT_FUSED = 1;
int uncl_conv1x1_gelu(const uncl_conv_desc* d, void* zbuf, int gmode, void* stream);

// fp32 parity mode (bwd_f32.hip): deterministic plain-fp32 forms of the matrix-core backward kernels
int bwd_wgrad_f32(const uncl_conv_desc* d, const void* gy, float* dw, hipStream_t s);
int bwd_upconv2x2_wgrad_f32(const void* x, const void* gy, float* dw, int N, int H, int W, int C, int Cout, hipStream_t s);
int bwd_upconv2x2_dgrad_f32(const void* gy, const void* wt, const void* mask, float slope, void* gx, int N, int H, int W, int Cin,
                            int Cout, hipStream_t s);
int bwd_mask_acc_f32(const void* src, const void* mask, float slope, void* dst, int accumulate, long long n, hipStream_t s);
int bwd_colsum_f32(const void* x, long long rows, int C, int ld, float* out, int accumulate, hipStream_t s);
int bwd_maxpool2_f32(const void* x, void* y, int N, int H, int W, int C, hipStream_t s);

// generator InstanceNorm (inorm.hip)
int bwd_inorm_forward(int dtype, void* x, void* zhat, float* rstd, const void* res, int res_b0, int N, int P, int C, float slope,
                      hipStream_t s);
int bwd_inorm_backward(int dtype, void* g, const void* zhat, const float* rstd, int N, int P, int C, hipStream_t s);
// BatchNorm2d training mode (inorm.hip); scratch = N * C * 16 + C * 8 bytes, 8-byte aligned
int bwd_bnorm_forward(int dtype, void* x, void* zhat, float* rstd_rep, const float* gamma, const float* beta, float* rmean, float* rvar,
                      float momentum, const void* res, int res_b0, int N, int P, int C, float slope, void* scratch, hipStream_t s);
int bwd_bnorm_backward(int dtype, void* g, const void* zhat, const float* rstd_rep, const float* gamma, float* g_gamma, float* g_beta,
                       int accumulate, int N, int P, int C, void* scratch, hipStream_t s);
int bwd_maxpool2(int dtype, const void* x, void* y, int N, int H, int W, int C, hipStream_t s);
int bwd_outc_forward(int dtype, const void* up, const float* w, const float* b, float* out, long long P, int act, hipStream_t s);

// wgrad.hip: true while a scratch buffer for deterministic weight gradients is set on this thread (uncl_wgrad_set_scratch)
bool uncl_wgrad_deterministic();
