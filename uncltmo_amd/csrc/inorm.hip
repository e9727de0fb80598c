// Generator InstanceNorm (unet_norm = 'instance_norm', unet_parts.py:20-29,34-37: nn.InstanceNorm2d(out_ch) -- no affine, no
// running statistics, eps 1e-5 -- between every 3x3 convolution and its activation), forward and backward, NHWC.
//
// One workgroup owns one (sample, 8-channel group): statistics are per (n, c) over H x W, so nothing crosses workgroups and
// every reduction is a fixed-order tree (deterministic).  Two-pass variance (mean first, then sum (x - mean)^2) like
// torch's CPU kernel, fp32 sums per thread, fp64 across the workgroup.  The passes re-read the workgroup's own 16-byte
// column of every pixel: HBM-bound, ~3 reads + 1-2 writes per element (the convolution before it cannot know the mean).
//
//   forward : z = conv(x) + b  ->  zhat = (z - mean) * rstd  ->  a = act(zhat) (+ residual)      [a in place of z]
//             training also keeps zhat (the activation loses its negative half) and rstd
//   backward: g_z = rstd * (g - mean(g) - zhat * mean(g * zhat)),  g = dL/dzhat (already masked by the activation derivative)
#include "bwd_internal.h"

namespace {

template <typename T>
__device__ __forceinline__ void ld8n(const T* p, float* f) {
  if constexpr (sizeof(T) == 2) {
    Elem<T>::unpack(*reinterpret_cast<const typename Elem<T>::vec*>(p), f);
  } else {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[i] = a[i]; f[4 + i] = b[i]; }
  }
}
template <typename T>
__device__ __forceinline__ void st8n(T* p, const float* f) {
  if constexpr (sizeof(T) == 2) {
    *reinterpret_cast<typename Elem<T>::vec*>(p) = Elem<T>::pack(f);
  } else {
    *reinterpret_cast<f32x4*>(p) = f32x4{f[0], f[1], f[2], f[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{f[4], f[5], f[6], f[7]};
  }
}

// sum of eight per-thread values over the 256 threads of the workgroup, fixed order, fp64; result in out[8] for everybody
__device__ __forceinline__ void block_sum8(const float* v, double* out, double (*red)[8]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    double t = (double)v[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if (lane == 0) red[wave][c] = t;
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 8; ++c) out[c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
  __syncthreads();
}

// grid (C / 8, N).  x: (N, P, C) in/out (a = act(zhat) [+ res]); zhat_out (optional) same shape; rstd_out (optional) [N][C]
template <typename T>
__global__ __launch_bounds__(256) void inorm_fwd_kernel(T* __restrict__ x, T* __restrict__ zhat_out, float* __restrict__ rstd_out,
                                                        const T* __restrict__ res, int res_b0, int P, int C, float slope,
                                                        float eps) {
  __shared__ double red[4][8];
  const int n = blockIdx.y, c0 = blockIdx.x * 8;
  T* xb = x + (size_t)n * P * C + c0;
  float s[8];
  double tot[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) s[c] = 0.f;
  for (int p = threadIdx.x; p < P; p += 256) {
    float f[8];
    ld8n(xb + (size_t)p * C, f);
#pragma unroll
    for (int c = 0; c < 8; ++c) s[c] += f[c];
  }
  block_sum8(s, tot, red);
  float mean[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) { mean[c] = (float)(tot[c] / P); s[c] = 0.f; }
  for (int p = threadIdx.x; p < P; p += 256) {
    float f[8];
    ld8n(xb + (size_t)p * C, f);
#pragma unroll
    for (int c = 0; c < 8; ++c) { const float d = f[c] - mean[c]; s[c] = fmaf(d, d, s[c]); }
  }
  block_sum8(s, tot, red);
  float rstd[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) rstd[c] = 1.f / sqrtf((float)(tot[c] / P) + eps);     // biased variance, like F.instance_norm
  if (rstd_out && threadIdx.x < 8) rstd_out[(size_t)n * C + c0 + threadIdx.x] = rstd[threadIdx.x];
  for (int p = threadIdx.x; p < P; p += 256) {
    float f[8], a[8];
    ld8n(xb + (size_t)p * C, f);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      f[c] = (f[c] - mean[c]) * rstd[c];
      a[c] = f[c] > 0.f ? f[c] : slope * f[c];
    }
    if (zhat_out) st8n(zhat_out + ((size_t)n * P + p) * C + c0, f);
    if (res) {
      float r[8];
      ld8n(res + ((size_t)(res_b0 ? 0 : n) * P + p) * C + c0, r);
#pragma unroll
      for (int c = 0; c < 8; ++c) a[c] += r[c];
    }
    st8n(xb + (size_t)p * C, a);
  }
}

// g (N, P, C) in place: dL/dzhat -> dL/dz
template <typename T>
__global__ __launch_bounds__(256) void inorm_bwd_kernel(T* __restrict__ g, const T* __restrict__ zhat, const float* __restrict__ rstd,
                                                        int P, int C) {
  __shared__ double red[4][8];
  const int n = blockIdx.y, c0 = blockIdx.x * 8;
  T* gb = g + (size_t)n * P * C + c0;
  const T* zb = zhat + (size_t)n * P * C + c0;
  float s1[8], s2[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) { s1[c] = 0.f; s2[c] = 0.f; }
  for (int p = threadIdx.x; p < P; p += 256) {
    float a[8], z[8];
    ld8n(gb + (size_t)p * C, a);
    ld8n(zb + (size_t)p * C, z);
#pragma unroll
    for (int c = 0; c < 8; ++c) { s1[c] += a[c]; s2[c] = fmaf(a[c], z[c], s2[c]); }
  }
  double t1[8], t2[8];
  block_sum8(s1, t1, red);
  block_sum8(s2, t2, red);
  float m1[8], m2[8], rs[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) { m1[c] = (float)(t1[c] / P); m2[c] = (float)(t2[c] / P); rs[c] = rstd[(size_t)n * C + c0 + c]; }
  for (int p = threadIdx.x; p < P; p += 256) {
    float a[8], z[8];
    ld8n(gb + (size_t)p * C, a);
    ld8n(zb + (size_t)p * C, z);
#pragma unroll
    for (int c = 0; c < 8; ++c) a[c] = rs[c] * (a[c] - m1[c] - z[c] * m2[c]);
    st8n(gb + (size_t)p * C, a);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// BatchNorm2d in TRAINING mode (unet_norm = 'batch_norm', unet_parts.py:20-21, 34-35, 72-73: nn.BatchNorm2d(out_ch), affine, eps
// 1e-5, momentum 0.1): statistics per channel over (N, H, W).  Three launches per direction: fp64 partial sums per (sample,
// 8-channel group) -- fixed-order trees, deterministic --, one thread per channel to finish them (and to update the running
// statistics: momentum, UNBIASED variance, like the module), then the element-wise pass.
//   forward : zhat = (z - mean) * rstd, y = gamma zhat + beta, a = act(y) (+ residual)         [a in place of z; zhat, rstd kept]
//   backward: g = dL/dy (masked by the activation derivative already)
//             dgamma = sum g zhat, dbeta = sum g, g_z = rstd gamma (g - mean(g) - zhat mean(g zhat))
// Eval mode never comes here: the host folds the running statistics into the convolutions (uncltmo_amd/generator.py).
// ------------------------------------------------------------------------------------------------------------------
// grid (C / 8, N): part[((n * C) + c) * 2 + {0, 1}] = sum_p v0, sum_p v1 with (v0, v1) = (x, x^2) or (g, g * zhat)
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void bn_partial_kernel(const T* __restrict__ x, const T* __restrict__ zhat, double* __restrict__ part,
                                                         int P, int C) {
  __shared__ double red[4][8];
  const int n = blockIdx.y, c0 = blockIdx.x * 8;
  const T* xb = x + (size_t)n * P * C + c0;
  const T* zb = BWD ? zhat + (size_t)n * P * C + c0 : nullptr;
  double s1[8], s2[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) { s1[c] = 0.0; s2[c] = 0.0; }
  for (int p = threadIdx.x; p < P; p += 256) {
    float f[8], z[8];
    ld8n(xb + (size_t)p * C, f);
    if (BWD) ld8n(zb + (size_t)p * C, z);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      s1[c] += (double)f[c];
      s2[c] += BWD ? (double)f[c] * (double)z[c] : (double)f[c] * (double)f[c];
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      double t = k ? s2[c] : s1[c];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
      if (lane == 0) red[wave][c] = t;
    }
    __syncthreads();
    if (threadIdx.x < 8)
      part[((size_t)n * C + c0 + threadIdx.x) * 2 + k] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    __syncthreads();
  }
}

// forward: one thread per channel.  coef[c] = mean, rstd_rep[n][c] = 1 / sqrt(var + eps) for every sample (the backward pass and
// the element-wise kernel index it like InstanceNorm's per-sample values); running statistics updated in place.
__global__ void bn_finalize_fwd_kernel(const double* __restrict__ part, int N, int P, int C, float eps, float momentum,
                                       float* __restrict__ coef, float* __restrict__ rstd_rep, float* __restrict__ rmean,
                                       float* __restrict__ rvar) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int n = 0; n < N; ++n) { s1 += part[((size_t)n * C + c) * 2]; s2 += part[((size_t)n * C + c) * 2 + 1]; }
  const double M = (double)N * (double)P;
  const double mean = s1 / M;
  double var = s2 / M - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  coef[c] = (float)mean;
  for (int n = 0; n < N; ++n) rstd_rep[(size_t)n * C + c] = rstd;
  if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
  if (rvar) rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)(M > 1.0 ? var * M / (M - 1.0) : var);
}

// backward: coef[c] = mean(g), coef[C + c] = mean(g zhat); dbeta / dgamma written or accumulated
__global__ void bn_finalize_bwd_kernel(const double* __restrict__ part, int N, int P, int C, float* __restrict__ coef,
                                       float* __restrict__ g_gamma, float* __restrict__ g_beta, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int n = 0; n < N; ++n) { s1 += part[((size_t)n * C + c) * 2]; s2 += part[((size_t)n * C + c) * 2 + 1]; }
  const double M = (double)N * (double)P;
  coef[c] = (float)(s1 / M);
  coef[C + c] = (float)(s2 / M);
  if (g_beta) g_beta[c] = (accumulate ? g_beta[c] : 0.f) + (float)s1;
  if (g_gamma) g_gamma[c] = (accumulate ? g_gamma[c] : 0.f) + (float)s2;
}

// grid (C / 8, N), forward element-wise pass
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(T* __restrict__ x, T* __restrict__ zhat_out, const float* __restrict__ coef,
                                                           const float* __restrict__ rstd_rep, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const T* __restrict__ res, int res_b0,
                                                           int P, int C, float slope) {
  const int n = blockIdx.y, c0 = blockIdx.x * 8;
  T* xb = x + (size_t)n * P * C + c0;
  float mean[8], rstd[8], ga[8], be[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) { mean[c] = coef[c0 + c]; rstd[c] = rstd_rep[(size_t)n * C + c0 + c]; ga[c] = gamma[c0 + c]; be[c] = beta[c0 + c]; }
  for (int p = threadIdx.x; p < P; p += 256) {
    float f[8], a[8];
    ld8n(xb + (size_t)p * C, f);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      f[c] = (f[c] - mean[c]) * rstd[c];
      const float y = fmaf(ga[c], f[c], be[c]);
      a[c] = y > 0.f ? y : slope * y;
    }
    if (zhat_out) st8n(zhat_out + ((size_t)n * P + p) * C + c0, f);
    if (res) {
      float r[8];
      ld8n(res + ((size_t)(res_b0 ? 0 : n) * P + p) * C + c0, r);
#pragma unroll
      for (int c = 0; c < 8; ++c) a[c] += r[c];
    }
    st8n(xb + (size_t)p * C, a);
  }
}

// grid (C / 8, N), backward element-wise pass: g (dL/dy) -> dL/dz in place
template <typename T>
__global__ __launch_bounds__(256) void bn_apply_bwd_kernel(T* __restrict__ g, const T* __restrict__ zhat, const float* __restrict__ coef,
                                                           const float* __restrict__ rstd_rep, const float* __restrict__ gamma, int P, int C) {
  const int n = blockIdx.y, c0 = blockIdx.x * 8;
  T* gb = g + (size_t)n * P * C + c0;
  const T* zb = zhat + (size_t)n * P * C + c0;
  float m1[8], m2[8], sc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) { m1[c] = coef[c0 + c]; m2[c] = coef[C + c0 + c]; sc[c] = rstd_rep[(size_t)n * C + c0 + c] * gamma[c0 + c]; }
  for (int p = threadIdx.x; p < P; p += 256) {
    float a[8], z[8];
    ld8n(gb + (size_t)p * C, a);
    ld8n(zb + (size_t)p * C, z);
#pragma unroll
    for (int c = 0; c < 8; ++c) a[c] = sc[c] * (a[c] - m1[c] - z[c] * m2[c]);
    st8n(gb + (size_t)p * C, a);
  }
}

// MaxPool2d(2) copy (the 16-bit path pools in the conv epilogue, which a norm between conv and activation rules out)
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C) {
  const int Hp = H / 2, Wp = W / 2, VC = C / 8;
  const size_t total = (size_t)N * Hp * Wp * VC;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int v = (int)(i % VC);
    size_t r = i / VC;
    const int px = (int)(r % Wp); r /= Wp;
    const int py = (int)(r % Hp);
    const int n = (int)(r / Hp);
    float m[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float f[8];
      ld8n(x + (((size_t)n * H + 2 * py + (q >> 1)) * W + 2 * px + (q & 1)) * C + v * 8, f);
#pragma unroll
      for (int c = 0; c < 8; ++c) m[c] = q == 0 ? f[c] : fmaxf(m[c], f[c]);
    }
    st8n(y + (((size_t)n * Hp + py) * Wp + px) * C + v * 8, m);
  }
}

// outconv (1x1, 32 -> 1) + last activation on a stored feature map (fused into the last conv's epilogue when there is no norm)
template <typename T>
__global__ __launch_bounds__(256) void outc_fwd_kernel(const T* __restrict__ up, const float* __restrict__ w, const float* __restrict__ b,
                                                       float* __restrict__ out, size_t P, int act) {
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < P; p += (size_t)gridDim.x * 256) {
    float s = b[0];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      float f[8];
      ld8n(up + p * 32 + v * 8, f);
#pragma unroll
      for (int c = 0; c < 8; ++c) s = fmaf(f[c], w[v * 8 + c], s);
    }
    out[p] = uncl_act(s, act);
  }
}

inline int nb(size_t n, int cap = 8192) {
  const size_t b = (n + 255) / 256;
  return (int)(b < (size_t)cap ? (b ? b : 1) : (size_t)cap);
}

}  // namespace

#define DISPATCH_T(dtype, CALL)                                             \
  do {                                                                      \
    if ((dtype) == UNCL_F32) { using T = float; CALL; }                      \
    else if ((dtype) == UNCL_F16) { using T = f16_t; CALL; }                 \
    else { using T = bf16_t; CALL; }                                         \
  } while (0)

int bwd_inorm_forward(int dtype, void* x, void* zhat, float* rstd, const void* res, int res_b0, int N, int P, int C, float slope,
                      hipStream_t s) {
  if (!x || N <= 0 || P <= 0 || C % 8 != 0) return UNCL_ERR_ARG;
  DISPATCH_T(dtype, hipLaunchKernelGGL(inorm_fwd_kernel<T>, dim3(C / 8, N), dim3(256), 0, s, (T*)x, (T*)zhat, rstd, (const T*)res,
                                       res_b0, P, C, slope, 1e-5f));
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

int bwd_inorm_backward(int dtype, void* g, const void* zhat, const float* rstd, int N, int P, int C, hipStream_t s) {
  if (!g || !zhat || !rstd || N <= 0 || P <= 0 || C % 8 != 0) return UNCL_ERR_ARG;
  DISPATCH_T(dtype, hipLaunchKernelGGL(inorm_bwd_kernel<T>, dim3(C / 8, N), dim3(256), 0, s, (T*)g, (const T*)zhat, rstd, P, C));
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

// scratch: doubles [N][C][2] then floats [2 C] (8-byte aligned, N * C * 16 + C * 8 bytes)
int bwd_bnorm_forward(int dtype, void* x, void* zhat, float* rstd_rep, const float* gamma, const float* beta, float* rmean, float* rvar,
                      float momentum, const void* res, int res_b0, int N, int P, int C, float slope, void* scratch, hipStream_t s) {
  if (!x || !rstd_rep || !gamma || !beta || !scratch || N <= 0 || P <= 0 || C % 8 != 0) return UNCL_ERR_ARG;
  double* part = reinterpret_cast<double*>(scratch);
  float* coef = reinterpret_cast<float*>(part + (size_t)N * C * 2);
  DISPATCH_T(dtype, hipLaunchKernelGGL((bn_partial_kernel<T, false>), dim3(C / 8, N), dim3(256), 0, s, (const T*)x, (const T*)nullptr, part, P, C));
  hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3((C + 63) / 64), dim3(64), 0, s, part, N, P, C, 1e-5f, momentum, coef, rstd_rep, rmean, rvar);
  DISPATCH_T(dtype, hipLaunchKernelGGL(bn_apply_fwd_kernel<T>, dim3(C / 8, N), dim3(256), 0, s, (T*)x, (T*)zhat, coef, rstd_rep, gamma, beta,
                                       (const T*)res, res_b0, P, C, slope));
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

int bwd_bnorm_backward(int dtype, void* g, const void* zhat, const float* rstd_rep, const float* gamma, float* g_gamma, float* g_beta,
                       int accumulate, int N, int P, int C, void* scratch, hipStream_t s) {
  if (!g || !zhat || !rstd_rep || !gamma || !scratch || N <= 0 || P <= 0 || C % 8 != 0) return UNCL_ERR_ARG;
  double* part = reinterpret_cast<double*>(scratch);
  float* coef = reinterpret_cast<float*>(part + (size_t)N * C * 2);
  DISPATCH_T(dtype, hipLaunchKernelGGL((bn_partial_kernel<T, true>), dim3(C / 8, N), dim3(256), 0, s, (const T*)g, (const T*)zhat, part, P, C));
  hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3((C + 63) / 64), dim3(64), 0, s, part, N, P, C, coef, g_gamma, g_beta, accumulate);
  DISPATCH_T(dtype, hipLaunchKernelGGL(bn_apply_bwd_kernel<T>, dim3(C / 8, N), dim3(256), 0, s, (T*)g, (const T*)zhat, coef, rstd_rep, gamma, P, C));
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

int bwd_maxpool2(int dtype, const void* x, void* y, int N, int H, int W, int C, hipStream_t s) {
  if (!x || !y || C % 8 != 0) return UNCL_ERR_ARG;
  DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool2_kernel<T>, dim3(nb((size_t)N * (H / 2) * (W / 2) * (C / 8))), dim3(256), 0, s,
                                       (const T*)x, (T*)y, N, H, W, C));
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

int bwd_outc_forward(int dtype, const void* up, const float* w, const float* b, float* out, long long P, int act, hipStream_t s) {
  if (!up || !w || !b || !out || P <= 0) return UNCL_ERR_ARG;
  DISPATCH_T(dtype, hipLaunchKernelGGL(outc_fwd_kernel<T>, dim3(nb((size_t)P)), dim3(256), 0, s, (const T*)up, w, b, out, (size_t)P,
                                       act));
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

// Stand-alone forms for tests and for callers outside the generator: x (N, H*W, C) NHWC in `dtype`.
extern "C" int uncl_inorm_act(void* x, void* zhat, float* rstd, int dtype, int N, int HW, int C, float slope, void* stream) {
  if (dtype != UNCL_F32 && !uncl_is_h16(dtype)) return UNCL_ERR_ARG;
  return bwd_inorm_forward(dtype, x, zhat, rstd, nullptr, 0, N, HW, C, slope, reinterpret_cast<hipStream_t>(stream));
}
extern "C" int uncl_inorm_backward(void* g, const void* zhat, const float* rstd, int dtype, int N, int HW, int C, void* stream) {
  if (dtype != UNCL_F32 && !uncl_is_h16(dtype)) return UNCL_ERR_ARG;
  return bwd_inorm_backward(dtype, g, zhat, rstd, N, HW, C, reinterpret_cast<hipStream_t>(stream));
}

// BatchNorm2d (training mode) + activation on an NHWC tensor, and its backward, stand-alone: x (N, HW, C) in `dtype`, in place;
// zhat (same shape) and rstd ([N][C], the per-channel value repeated per sample) are kept for the backward; gamma / beta /
// running_mean / running_var fp32 [C]; scratch = uncl_bnorm_scratch_bytes(N, C) bytes of device memory.
extern "C" size_t uncl_bnorm_scratch_bytes(int N, int C) { return (size_t)N * C * 16 + (size_t)C * 8 + 64; }
extern "C" int uncl_bnorm_act(void* x, void* zhat, float* rstd, const float* gamma, const float* beta, float* running_mean,
                              float* running_var, float momentum, int dtype, int N, int HW, int C, float slope, void* scratch,
                              void* stream) {
  if (dtype != UNCL_F32 && !uncl_is_h16(dtype)) return UNCL_ERR_ARG;
  return bwd_bnorm_forward(dtype, x, zhat, rstd, gamma, beta, running_mean, running_var, momentum, nullptr, 0, N, HW, C, slope, scratch,
                           reinterpret_cast<hipStream_t>(stream));
}
// g: dL/dy already multiplied by the activation derivative, in place -> dL/dz; g_gamma / g_beta [C] written (accumulate = 0) or added
extern "C" int uncl_bnorm_backward(void* g, const void* zhat, const float* rstd, const float* gamma, float* g_gamma, float* g_beta,
                                   int accumulate, int dtype, int N, int HW, int C, void* scratch, void* stream) {
  if (dtype != UNCL_F32 && !uncl_is_h16(dtype)) return UNCL_ERR_ARG;
  return bwd_bnorm_backward(dtype, g, zhat, rstd, gamma, g_gamma, g_beta, accumulate, N, HW, C, scratch, reinterpret_cast<hipStream_t>(stream));
}
