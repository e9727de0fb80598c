// Structural loss (models/struct_loss.py:46-104), forward and backward, fp32, HBM-bound.
//
// Per pyramid level: every 5x5 window w of `fake` (a) and of `hdr_input` (b) is normalised by its own box-filter
// mean and std,  u = (w - mu) / (sqrt(max(var,0) + 1e-5) + 1e-5),  and the loss is the MSE between the two
// normalised window stacks.  The reference materialises both (N,1,H-4,W-4,25) stacks with `unfold`; here a
// workgroup keeps a (32+8)^2 input tile in LDS and never writes the stacks.
//
// Backward (d loss / d a): with per-window scalars ia = 1/(sd_a + eps), ib, mu_a, mu_b, D0 = sum_t d_t,
// D1 = sum_t d_t (a_t - mu_a), d_t = 2 (u_t - v_t), the contribution of window o to pixel i is
//   A_o a_i + B_o b_i + C_o,   A = 2 ia^2 - ia^2 [var>0] D1 / (25 sd),   B = -2 ia ib,
//   C = -2 ia^2 mu_a + 2 ia ib mu_b - ia D0 / 25 + ia^2 [var>0] D1 mu_a / (25 sd)
// so the gradient is three 5x5 "full" box sums of coefficient maps — again tiled through LDS.
// Between levels the images are halved with bicubic (A = -0.75) interpolation, align_corners = False:
// out[o] = sum_k w_k in[clamp(2o - 1 + k)], w = (-3/32, 19/32, 19/32, -3/32) per axis.
#include "common.h"

namespace {

constexpr int WS = 5;
constexpr float EPS2 = 1e-5f;
constexpr int TS = 32;            // output tile edge
constexpr int WT = TS + WS - 1;   // windows needed by the backward tile (36)
constexpr int IT = WT + WS - 1;   // input pixels needed by those windows (40)

__device__ __forceinline__ void window_stats(const float* s, int stride, float& mu, float& sd, float& var) {
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int dy = 0; dy < WS; ++dy)
#pragma unroll
    for (int dx = 0; dx < WS; ++dx) {
      const float v = s[dy * stride + dx];
      s1 += v;
      s2 = fmaf(v, v, s2);
    }
  mu = s1 * (1.f / 25.f);
  var = s2 * (1.f / 25.f) - mu * mu;
  sd = sqrtf(fmaxf(var, 0.f) + EPS2);
}

// forward: per-workgroup partial sums of the squared differences
__global__ __launch_bounds__(256) void struct_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ partial, int H, int W, int tiles_x,
                                                         int tiles_y) {
  __shared__ float sa[(TS + 4) * (TS + 4)], sb[(TS + 4) * (TS + 4)];
  __shared__ float red[4];
  constexpr int L = TS + 4;
  const int n = blockIdx.y;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int y0 = ty * TS, x0 = tx * TS;
  const int Ho = H - 4, Wo = W - 4;
  const float* an = a + (size_t)n * H * W;
  const float* bn = b + (size_t)n * H * W;
  for (int i = threadIdx.x; i < L * L; i += 256) {
    const int ly = i / L, lx = i - ly * L;
    const int gy = min(y0 + ly, H - 1), gx = min(x0 + lx, W - 1);
    sa[i] = an[(size_t)gy * W + gx];
    sb[i] = bn[(size_t)gy * W + gx];
  }
  __syncthreads();
  float acc = 0.f;
  for (int i = threadIdx.x; i < TS * TS; i += 256) {
    const int ly = i / TS, lx = i - ly * TS;
    if (y0 + ly < Ho && x0 + lx < Wo) {
      float mua, sda, va, mub, sdb, vb;
      window_stats(sa + ly * L + lx, L, mua, sda, va);
      window_stats(sb + ly * L + lx, L, mub, sdb, vb);
      const float ia = 1.f / (sda + EPS2), ib = 1.f / (sdb + EPS2);
#pragma unroll
      for (int dy = 0; dy < WS; ++dy)
#pragma unroll
        for (int dx = 0; dx < WS; ++dx) {
          const float u = (sa[(ly + dy) * L + lx + dx] - mua) * ia;
          const float v = (sb[(ly + dy) * L + lx + dx] - mub) * ib;
          const float d = u - v;
          acc = fmaf(d, d, acc);
        }
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[(size_t)n * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void sum_partials_kernel(const float* __restrict__ partial, int count, double scale, float* __restrict__ out,
                                    int accumulate) {
  // one wave; fixed order -> deterministic
  double s = 0.0;
  for (int i = threadIdx.x; i < count; i += 64) s += (double)partial[i];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + (float)(s * scale);
}

// backward: grad_a (+)= scale * d(sum of squared differences)/da
__global__ __launch_bounds__(256) void struct_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ ga, const float* __restrict__ upstream,
                                                         float scale, int H, int W, int tiles_x, int accumulate) {
  __shared__ float sa[IT * IT], sb[IT * IT];
  __shared__ float cA[WT * WT], cB[WT * WT], cC[WT * WT];
  const int n = blockIdx.y;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int y0 = ty * TS, x0 = tx * TS;    // output (pixel) tile origin
  const int Ho = H - 4, Wo = W - 4;
  const float* an = a + (size_t)n * H * W;
  const float* bn = b + (size_t)n * H * W;
  // windows o in [y0-4, y0+TS) x [x0-4, x0+TS) cover the tile; they read pixels [y0-4, y0+TS+4)
  for (int i = threadIdx.x; i < IT * IT; i += 256) {
    const int ly = i / IT, lx = i - ly * IT;
    const int gy = min(max(y0 - 4 + ly, 0), H - 1), gx = min(max(x0 - 4 + lx, 0), W - 1);
    sa[i] = an[(size_t)gy * W + gx];
    sb[i] = bn[(size_t)gy * W + gx];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < WT * WT; i += 256) {
    const int wy = i / WT, wx = i - wy * WT;
    const int oy = y0 - 4 + wy, ox = x0 - 4 + wx;   // window origin in the image
    float A = 0.f, B = 0.f, C = 0.f;
    if (oy >= 0 && oy < Ho && ox >= 0 && ox < Wo) {
      float mua, sda, va, mub, sdb, vb;
      window_stats(sa + wy * IT + wx, IT, mua, sda, va);
      window_stats(sb + wy * IT + wx, IT, mub, sdb, vb);
      const float ia = 1.f / (sda + EPS2), ib = 1.f / (sdb + EPS2);
      float D0 = 0.f, D1 = 0.f;
#pragma unroll
      for (int dy = 0; dy < WS; ++dy)
#pragma unroll
        for (int dx = 0; dx < WS; ++dx) {
          const float da = sa[(wy + dy) * IT + wx + dx] - mua;
          const float d = 2.f * (da * ia - (sb[(wy + dy) * IT + wx + dx] - mub) * ib);
          D0 += d;
          D1 = fmaf(d, da, D1);
        }
      const float k = (va > 0.f) ? ia * ia * D1 / (25.f * sda) : 0.f;
      A = 2.f * ia * ia - k;
      B = -2.f * ia * ib;
      C = -2.f * ia * ia * mua + 2.f * ia * ib * mub - ia * D0 * (1.f / 25.f) + k * mua;
    }
    cA[i] = A; cB[i] = B; cC[i] = C;
  }
  __syncthreads();
  const float s = scale * (upstream ? upstream[0] : 1.f);
  for (int i = threadIdx.x; i < TS * TS; i += 256) {
    const int ly = i / TS, lx = i - ly * TS;
    const int gy = y0 + ly, gx = x0 + lx;
    if (gy < H && gx < W) {
      float SA = 0.f, SB = 0.f, SC = 0.f;
      // windows with origin (gy - dy, gx - dx), dy,dx in 0..4  ->  local (ly + 4 - dy, lx + 4 - dx)
#pragma unroll
      for (int dy = 0; dy < WS; ++dy)
#pragma unroll
        for (int dx = 0; dx < WS; ++dx) {
          const int j = (ly + 4 - dy) * WT + lx + 4 - dx;
          SA += cA[j]; SB += cB[j]; SC += cC[j];
        }
      const float av = sa[(ly + 4) * IT + lx + 4], bv = sb[(ly + 4) * IT + lx + 4];
      const float g = s * (av * SA + bv * SB + SC);
      float* dst = ga + (size_t)n * H * W + (size_t)gy * W + gx;
      *dst = accumulate ? *dst + g : g;
    }
  }
}

__device__ __forceinline__ float bicubic_w(int k) { return (k == 0 || k == 3) ? (-3.f / 32.f) : (19.f / 32.f); }

__global__ void bicubic_half_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, int N, int H, int W) {
  const int Ho = H / 2, Wo = W / 2;
  const size_t total = (size_t)N * Ho * Wo;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % Wo), oy = (int)((i / Wo) % Ho), n = (int)(i / ((size_t)Wo * Ho));
    const float* p = in + (size_t)n * H * W;
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
      const int iy = min(max(2 * oy - 1 + ky, 0), H - 1);
      float row = 0.f;
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const int ix = min(max(2 * ox - 1 + kx, 0), W - 1);
        row = fmaf(bicubic_w(kx), p[(size_t)iy * W + ix], row);
      }
      acc = fmaf(bicubic_w(ky), row, acc);
    }
    out[i] = acc;
  }
}

// gin[i] += sum over (o, k) with clamp(2o - 1 + k) == i of w_k * gout[o]   (both axes)
__global__ void bicubic_half_bwd_kernel(const float* __restrict__ gout, float* __restrict__ gin, int N, int H, int W) {
  const int Ho = H / 2, Wo = W / 2;
  const size_t total = (size_t)N * H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ix = (int)(i % W), iy = (int)((i / W) % H), n = (int)(i / ((size_t)W * H));
    const float* g = gout + (size_t)n * Ho * Wo;
    float acc = 0.f;
    for (int oy = max(iy / 2 - 1, 0); oy <= min(iy / 2 + 1, Ho - 1); ++oy) {
      float wy = 0.f;
#pragma unroll
      for (int ky = 0; ky < 4; ++ky)
        if (min(max(2 * oy - 1 + ky, 0), H - 1) == iy) wy += bicubic_w(ky);
      if (wy == 0.f) continue;
      for (int ox = max(ix / 2 - 1, 0); ox <= min(ix / 2 + 1, Wo - 1); ++ox) {
        float wx = 0.f;
#pragma unroll
        for (int kx = 0; kx < 4; ++kx)
          if (min(max(2 * ox - 1 + kx, 0), W - 1) == ix) wx += bicubic_w(kx);
        if (wx != 0.f) acc = fmaf(wy * wx, g[(size_t)oy * Wo + ox], acc);
      }
    }
    gin[i] += acc;
  }
}

}  // namespace

extern "C" size_t uncl_struct_loss_workspace_bytes(int N, int H, int W, int levels) {
  size_t img = 0, part = 0;
  int h = H, w = W;
  for (int l = 0; l < levels; ++l) {
    part += (size_t)N * ((h - 4 + TS - 1) / TS) * ((w - 4 + TS - 1) / TS);
    if (l > 0) img += (size_t)N * h * w * 3;  // a_l, b_l, grad_l
    h /= 2; w /= 2;
  }
  return (img + part) * sizeof(float) + 256;
}

// fake, hdr: fp32 (N,H,W).  weights: host array of `levels` pyramid weights.  loss_out: device fp32 scalar
// (= sum_l weights[l] * MSE_l).  If grad_fake != NULL it receives d loss_out / d fake times upstream[0]
// (upstream: device fp32 scalar or NULL for 1).  Replaces StructLoss.forward + its autograd backward
// (models/struct_loss.py:23-104).
extern "C" int uncl_struct_loss(const float* fake, const float* hdr, const float* weights_host, int levels, float* loss_out,
                                float* grad_fake, const float* upstream, int N, int H, int W, void* workspace,
                                void* stream) {
  if (!fake || !hdr || !weights_host || !loss_out || !workspace || N <= 0 || levels <= 0 || levels > 6) return UNCL_ERR_ARG;
  if ((H >> (levels - 1)) < WS || (W >> (levels - 1)) < WS) return UNCL_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  float* ws = reinterpret_cast<float*>(workspace);
  const float* a[6]; const float* b[6]; float* g[6]; float* part[6];
  int hh[6], ww[6];
  a[0] = fake; b[0] = hdr; g[0] = grad_fake; hh[0] = H; ww[0] = W;
  for (int l = 1; l < levels; ++l) {
    hh[l] = hh[l - 1] / 2; ww[l] = ww[l - 1] / 2;
    const size_t sz = (size_t)N * hh[l] * ww[l];
    float* al = ws; ws += sz;
    float* bl = ws; ws += sz;
    g[l] = ws; ws += sz;
    a[l] = al; b[l] = bl;
    const int blocks = (int)((sz + 255) / 256 < 2048 ? (sz + 255) / 256 : 2048);
    hipLaunchKernelGGL(bicubic_half_fwd_kernel, dim3(blocks), dim3(256), 0, st, a[l - 1], al, N, hh[l - 1], ww[l - 1]);
    hipLaunchKernelGGL(bicubic_half_fwd_kernel, dim3(blocks), dim3(256), 0, st, b[l - 1], bl, N, hh[l - 1], ww[l - 1]);
  }
  for (int l = 0; l < levels; ++l) {
    const int tx = (ww[l] - 4 + TS - 1) / TS, ty = (hh[l] - 4 + TS - 1) / TS;
    part[l] = ws; ws += (size_t)N * tx * ty;
    hipLaunchKernelGGL(struct_fwd_kernel, dim3(tx * ty, N), dim3(256), 0, st, a[l], b[l], part[l], hh[l], ww[l], tx, ty);
    const double denom = (double)N * (hh[l] - 4) * (ww[l] - 4) * 25.0;
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, st, part[l], N * tx * ty, (double)weights_host[l] / denom,
                       loss_out, l > 0 ? 1 : 0);
  }
  UNCL_CHECK_LAUNCH();
  if (grad_fake != nullptr) {
    // coarse to fine: g_l = upstream * w_l * dMSE_l/da_l  +  bicubic^T(g_{l+1})
    for (int l = levels - 1; l >= 0; --l) {
      const int tx = (ww[l] + TS - 1) / TS, ty = (hh[l] + TS - 1) / TS;
      const double denom = (double)N * (hh[l] - 4) * (ww[l] - 4) * 25.0;
      hipLaunchKernelGGL(struct_bwd_kernel, dim3(tx * ty, N), dim3(256), 0, st, a[l], b[l], g[l], upstream,
                         (float)((double)weights_host[l] / denom), hh[l], ww[l], tx, 0);
      if (l + 1 < levels) {
        const size_t sz = (size_t)N * hh[l] * ww[l];
        const int blocks = (int)((sz + 255) / 256 < 2048 ? (sz + 255) / 256 : 2048);
        hipLaunchKernelGGL(bicubic_half_bwd_kernel, dim3(blocks), dim3(256), 0, st, g[l + 1], g[l], N, hh[l], ww[l]);
      }
    }
    UNCL_CHECK_LAUNCH();
  }
  return UNCL_OK;
}

extern "C" int uncl_bicubic_half(const float* in, float* out, int N, int H, int W, void* stream) {
  if (!in || !out || N <= 0 || H < 2 || W < 2) return UNCL_ERR_ARG;
  const size_t sz = (size_t)N * (H / 2) * (W / 2);
  const int blocks = (int)((sz + 255) / 256 < 2048 ? (sz + 255) / 256 : 2048);
  hipLaunchKernelGGL(bicubic_half_fwd_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), in, out, N, H, W);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}
