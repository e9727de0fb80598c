// PatchGAN discriminator forward (models/Discriminator.py:129-167 NLayerDiscriminator, models/Blocks.py:6-36
// Conv2dBlock), fp32 throughout, NHWC activations:
//   conv(4,2,1)+bias+LReLU(0.2)  ->  (n_layers-1) x [conv(4,2,1) -> InstanceNorm(eps 1e-5, no affine) -> LReLU]
//   -> [conv(4,1,1) -> InstanceNorm -> LReLU] -> conv(4,1,1)+bias  (one-channel patch map)
// The published configuration (ndf 16, n_layers 3, instance norm) is 0.40 GFLOP per 256x256 frame and is not reachable
// from the trainers: a forward-parity module.  Direct fp32 VALU convolutions from LDS tiles keep
// it comparable with the CPU reference at 1e-4; no MFMA, no reshaping into GEMMs.
#include "bwd_internal.h"

namespace {

constexpr int PT = 8;        // output tile: PT x PT pixels
constexpr int COB = 32;      // output channels per workgroup (8 per wave)
constexpr int CIB = 16;      // input channels per LDS stage

// x: (N,H,W,Cin) fp32 NHWC; w: reference layout (Cout,Cin,4,4); y: (N,Ho,Wo,Cout); pad 1.
// workgroup = (PT x PT pixels) x COB channels; wave g owns channels 8g..8g+7 of the block, lane = pixel
template <int S>
__global__ __launch_bounds__(256) void patch_conv4_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y, int H, int W,
                                                          int Cin, int Ho, int Wo, int Cout, int tiles_x, float slope,
                                                          int apply_act) {
  constexpr int IT = (PT - 1) * S + 4;       // input tile edge
  constexpr int ITP = IT + 1;                // +1 column: the stride-S pixel pattern spreads over the banks
  __shared__ float sX[CIB * IT * ITP];
  __shared__ float sW[COB * CIB * 16];
  const int n = blockIdx.z, cb = blockIdx.y * COB;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int py = lane >> 3, px = lane & 7;
  const int oy = ty * PT + py, ox = tx * PT + px;
  const int iy0 = ty * PT * S - 1, ix0 = tx * PT * S - 1;
  float acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = 0.f;
  for (int c0 = 0; c0 < Cin; c0 += CIB) {
    const int nc = min(CIB, Cin - c0);
    __syncthreads();
    for (int i = threadIdx.x; i < nc * IT * IT; i += 256) {
      const int ci = i % nc, p = i / nc;          // consecutive threads read consecutive channels of one pixel
      const int ly = p / IT, lx = p - ly * IT;
      const int gy = iy0 + ly, gx = ix0 + lx;
      float v = 0.f;
      if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = x[(((size_t)n * H + gy) * W + gx) * Cin + c0 + ci];
      sX[(ci * IT + ly) * ITP + lx] = v;
    }
    for (int i = threadIdx.x; i < COB * nc * 16; i += 256) {
      const int t = i & 15, ci = (i >> 4) % nc, co = (i >> 4) / nc;
      sW[(co * CIB + ci) * 16 + t] = (cb + co < Cout) ? w[((size_t)(cb + co) * Cin + c0 + ci) * 16 + t] : 0.f;
    }
    __syncthreads();
    for (int ci = 0; ci < nc; ++ci) {
#pragma unroll
      for (int ky = 0; ky < 4; ++ky) {
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
          const float xv = sX[(ci * IT + py * S + ky) * ITP + px * S + kx];
#pragma unroll
          for (int c = 0; c < 8; ++c) acc[c] = fmaf(xv, sW[((g * 8 + c) * CIB + ci) * 16 + ky * 4 + kx], acc[c]);
        }
      }
    }
  }
  if (oy < Ho && ox < Wo) {
    float* o = y + (((size_t)n * Ho + oy) * Wo + ox) * Cout + cb + g * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (cb + g * 8 + c < Cout) {
        float v = acc[c] + (bias ? bias[cb + g * 8 + c] : 0.f);
        if (apply_act) v = v > 0.f ? v : slope * v;
        o[c] = v;
      }
    }
  }
}

// InstanceNorm2d (biased variance, eps, no affine) + LeakyReLU, in place on (N,HW,C) fp32.
// One workgroup per (sample, 32-channel group): two passes (mean, then centred second moment) as torch does.
__global__ __launch_bounds__(256) void inorm_lrelu_kernel(float* __restrict__ y, int HW, int C, float eps, float slope) {
  __shared__ float red[8][33];
  __shared__ float s_mean[32], s_rstd[32];
  const int n = blockIdx.y, c0 = blockIdx.x * 32;
  const int c = threadIdx.x & 31, r = threadIdx.x >> 5;   // 8 pixel lanes per channel
  float* base = y + (size_t)n * HW * C + c0 + c;
  const bool on = c0 + c < C;
  float s = 0.f;
  if (on)
    for (int p = r; p < HW; p += 8) s += base[(size_t)p * C];
  red[r][c] = s;
  __syncthreads();
  if (r == 0) {
    float t = 0.f;
    for (int i = 0; i < 8; ++i) t += red[i][c];
    s_mean[c] = t / (float)HW;
  }
  __syncthreads();
  const float m = s_mean[c];
  s = 0.f;
  if (on)
    for (int p = r; p < HW; p += 8) {
      const float d = base[(size_t)p * C] - m;
      s = fmaf(d, d, s);
    }
  __syncthreads();
  red[r][c] = s;
  __syncthreads();
  if (r == 0) {
    float t = 0.f;
    for (int i = 0; i < 8; ++i) t += red[i][c];
    s_rstd[c] = rsqrtf(t / (float)HW + eps);
  }
  __syncthreads();
  const float rs = s_rstd[c];
  if (on)
    for (int p = r; p < HW; p += 8) {
      const float v = (base[(size_t)p * C] - m) * rs;
      base[(size_t)p * C] = v > 0.f ? v : slope * v;
    }
}

int conv4(const float* x, const float* w, const float* b, float* y, int N, int H, int W, int Cin, int Cout, int S, int act,
          hipStream_t st) {
  const int Ho = (H + 2 - 4) / S + 1, Wo = (W + 2 - 4) / S + 1;
  const int tx = (Wo + PT - 1) / PT, ty = (Ho + PT - 1) / PT;
  dim3 grid(tx * ty, (Cout + COB - 1) / COB, N);
  if (S == 2)
    hipLaunchKernelGGL(patch_conv4_kernel<2>, grid, dim3(256), 0, st, x, w, b, y, H, W, Cin, Ho, Wo, Cout, tx, 0.2f, act);
  else
    hipLaunchKernelGGL(patch_conv4_kernel<1>, grid, dim3(256), 0, st, x, w, b, y, H, W, Cin, Ho, Wo, Cout, tx, 0.2f, act);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

}  // namespace

// ---- backward (fp32, deterministic: one owner per output element, fixed-order sums) ---------------------------------------
// g_x[n,iy,ix,ci] = sum_{co,ky,kx} g_y[n,oy,ox,co] w[co,ci,ky,kx],  oy * S - 1 + ky = iy,  ox * S - 1 + kx = ix
__global__ __launch_bounds__(256) void patch_dgrad4_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                           float* __restrict__ gx, int N, int H, int W, int Cin, int Ho, int Wo,
                                                           int Cout, int S) {
  const size_t total = (size_t)N * H * W * Cin;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ci = (int)(i % Cin);
    size_t r = i / Cin;
    const int ix = (int)(r % W); r /= W;
    const int iy = (int)(r % H);
    const int n = (int)(r / H);
    float s = 0.f;
    for (int ky = 0; ky < 4; ++ky) {
      const int ty = iy + 1 - ky;
      if (ty < 0 || ty % S != 0) continue;
      const int oy = ty / S;
      if (oy >= Ho) continue;
      for (int kx = 0; kx < 4; ++kx) {
        const int tx = ix + 1 - kx;
        if (tx < 0 || tx % S != 0) continue;
        const int ox = tx / S;
        if (ox >= Wo) continue;
        const float* g = gy + (((size_t)n * Ho + oy) * Wo + ox) * Cout;
        const float* wp = w + (size_t)ci * 16 + ky * 4 + kx;
        for (int co = 0; co < Cout; ++co) s = fmaf(g[co], wp[(size_t)co * Cin * 16], s);
      }
    }
    gx[i] = s;
  }
}

// gw[co][ci][ky][kx] = sum_{n,oy,ox} g_y[n,oy,ox,co] x[n, oy S - 1 + ky, ox S - 1 + kx, ci]; one workgroup per (co, ci)
__global__ __launch_bounds__(256) void patch_wgrad4_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                           float* __restrict__ gw, int N, int H, int W, int Cin, int Ho, int Wo,
                                                           int Cout, int S) {
  __shared__ float red[256][17];
  const int co = blockIdx.x, ci = blockIdx.y;
  float acc[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) acc[t] = 0.f;
  const int P = N * Ho * Wo;
  for (int p = threadIdx.x; p < P; p += 256) {
    const int n = p / (Ho * Wo), r = p - n * Ho * Wo;
    const int oy = r / Wo, ox = r - oy * Wo;
    const float g = gy[(size_t)p * Cout + co];
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
      const int iy = oy * S - 1 + ky;
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const int ix = ox * S - 1 + kx;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
          acc[ky * 4 + kx] = fmaf(g, x[(((size_t)n * H + iy) * W + ix) * Cin + ci], acc[ky * 4 + kx]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 16; ++t) red[threadIdx.x][t] = acc[t];
  __syncthreads();
  if (threadIdx.x < 16) {
    double s = 0.0;
    for (int i = 0; i < 256; ++i) s += (double)red[i][threadIdx.x];
    gw[((size_t)co * Cin + ci) * 16 + threadIdx.x] = (float)s;
  }
}

// g *= (a > 0 ? 1 : slope)  (LeakyReLU derivative through its own output: the sign is preserved)
__global__ __launch_bounds__(256) void lrelu_mask_kernel(float* __restrict__ g, const float* __restrict__ a, size_t n, float slope) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) g[i] = a[i] > 0.f ? g[i] : slope * g[i];
}

inline int nbp(size_t n) {
  const size_t b = (n + 255) / 256;
  return (int)(b < 8192 ? (b ? b : 1) : 8192);
}

// spatial size after layer i (0-based) of the n_layers + 2 convolutions, for an H x H input
static int patch_d_size(int H, int n_layers, int layer) {
  int h = H;
  for (int i = 0; i <= layer; ++i) h = (h + 2 - 4) / (i < n_layers ? 2 : 1) + 1;
  return h;
}

extern "C" size_t uncl_patch_d_workspace_bytes(int N, int H, int ndf, int n_layers) {
  if (N <= 0 || H < 16 || ndf <= 0 || n_layers < 1 || n_layers > 5) return 0;
  // two ping-pong buffers, each large enough for the biggest activation (the first layer's output)
  const int h0 = patch_d_size(H, n_layers, 0);
  return 2 * (((size_t)N * h0 * h0 * ndf * 4 + 255) & ~(size_t)255);
}

// x: fp32 (N,H,H) one-channel frames.  w[i], i = 0 .. n_layers+1: reference-layout conv weights (Cout,Cin,4,4);
// b_first (ndf) and b_last (1) are the biases of the first and last convolution (the blocks in between have none).
// out: (N, Ho, Ho) patch logits with Ho = uncl_patch_d_out_size(H, n_layers).
extern "C" int uncl_patch_d_out_size(int H, int n_layers) { return patch_d_size(H, n_layers, n_layers + 1); }

extern "C" int uncl_patch_d_forward(const float* x, const float* const* w, const float* b_first, const float* b_last, float* out,
                                    int N, int H, int ndf, int n_layers, void* workspace, void* stream) {
  if (!x || !w || !b_first || !b_last || !out || !workspace || N <= 0 || H < 16 || ndf <= 0 || ndf % 8 != 0) return UNCL_ERR_ARG;
  if (n_layers < 1 || n_layers > 5) return UNCL_ERR_ARG;
  for (int i = 0; i < n_layers + 2; ++i)
    if (!w[i]) return UNCL_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int h0 = patch_d_size(H, n_layers, 0);
  const size_t half = ((size_t)N * h0 * h0 * ndf * 4 + 255) & ~(size_t)255;
  float* buf[2] = {reinterpret_cast<float*>(workspace), reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + half)};
  int rc;
  // layer 0: conv(4,2,1) + bias + LeakyReLU
  if ((rc = conv4(x, w[0], b_first, buf[0], N, H, H, 1, ndf, 2, 1, st)) != UNCL_OK) return rc;
  int h = h0, cin = ndf, cur = 0, mult = 1;
  for (int i = 1; i <= n_layers; ++i) {
    mult = 1 << i;
    if (mult > 8) mult = 8;
    const int cout = ndf * mult, s = i < n_layers ? 2 : 1;
    const int ho = (h + 2 - 4) / s + 1;
    if ((size_t)N * ho * ho * cout * 4 > half) return UNCL_ERR_ARG;
    if ((rc = conv4(buf[cur], w[i], nullptr, buf[cur ^ 1], N, h, h, cin, cout, s, 0, st)) != UNCL_OK) return rc;
    hipLaunchKernelGGL(inorm_lrelu_kernel, dim3((cout + 31) / 32, N), dim3(256), 0, st, buf[cur ^ 1], ho * ho, cout, 1e-5f, 0.2f);
    UNCL_CHECK_LAUNCH();
    h = ho; cin = cout; cur ^= 1;
  }
  return conv4(buf[cur], w[n_layers + 1], b_last, out, N, h, h, cin, 1, 1, 0, st);
}

// ---- training form: the forward keeps a_0 and, per normalised block, zhat / a / rstd; the backward returns every parameter
// ---- gradient (reference layouts) and the input gradient.  Arena layout: see patch_train_layout().
namespace {
struct PdLayer { size_t a, z, r; int h, c; };     // byte offsets of the activation, its normalised pre-activation and 1/std
struct PdLayout { PdLayer L[8]; size_t total; };
PdLayout patch_train_layout(int N, int H, int ndf, int n_layers) {
  PdLayout p;
  size_t o = 0;
  auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
  int c = ndf;
  for (int i = 0; i <= n_layers; ++i) {
    int mult = 1 << i;
    if (mult > 8) mult = 8;
    c = ndf * mult;
    const int h = patch_d_size(H, n_layers, i);
    p.L[i].h = h; p.L[i].c = c;
    p.L[i].a = take((size_t)N * h * h * c * 4);
    p.L[i].z = i > 0 ? take((size_t)N * h * h * c * 4) : 0;
    p.L[i].r = i > 0 ? take((size_t)N * c * 4) : 0;
  }
  p.total = o;
  return p;
}
}  // namespace

extern "C" size_t uncl_patch_d_train_bytes(int N, int H, int ndf, int n_layers) {
  if (N <= 0 || H < 16 || ndf <= 0 || n_layers < 1 || n_layers > 5) return 0;
  const PdLayout p = patch_train_layout(N, H, ndf, n_layers);
  // + two gradient buffers of the largest activation for the backward pass
  return p.total + 2 * (((size_t)N * p.L[0].h * p.L[0].h * p.L[0].c * 4 + 255) & ~(size_t)255);
}

extern "C" int uncl_patch_d_forward_train(const float* x, const float* const* w, const float* b_first, const float* b_last, float* out,
                                          int N, int H, int ndf, int n_layers, void* arena, void* stream) {
  if (!x || !w || !b_first || !b_last || !out || !arena || N <= 0 || H < 16 || ndf <= 0 || ndf % 8 != 0) return UNCL_ERR_ARG;
  if (n_layers < 1 || n_layers > 5) return UNCL_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const PdLayout p = patch_train_layout(N, H, ndf, n_layers);
  char* base = reinterpret_cast<char*>(arena);
  int rc;
  if ((rc = conv4(x, w[0], b_first, reinterpret_cast<float*>(base + p.L[0].a), N, H, H, 1, ndf, 2, 1, st)) != UNCL_OK) return rc;
  for (int i = 1; i <= n_layers; ++i) {
    const PdLayer& q = p.L[i];
    const PdLayer& pr = p.L[i - 1];
    float* a = reinterpret_cast<float*>(base + q.a);
    if ((rc = conv4(reinterpret_cast<const float*>(base + pr.a), w[i], nullptr, a, N, pr.h, pr.h, pr.c, q.c, i < n_layers ? 2 : 1, 0,
                    st)) != UNCL_OK)
      return rc;
    if ((rc = bwd_inorm_forward(UNCL_F32, a, base + q.z, reinterpret_cast<float*>(base + q.r), nullptr, 0, N, q.h * q.h, q.c, 0.2f,
                                st)) != UNCL_OK)
      return rc;
  }
  const PdLayer& l = p.L[n_layers];
  return conv4(reinterpret_cast<const float*>(base + l.a), w[n_layers + 1], b_last, out, N, l.h, l.h, l.c, 1, 1, 0, st);
}

// g_out: (N,Ho,Ho); gw[i]: reference-layout weight gradients (overwritten); gb_first (ndf), gb_last (1); g_x (N,H,H) or NULL
extern "C" int uncl_patch_d_backward(const float* x, const float* const* w, const float* g_out, float* const* gw, float* gb_first,
                                     float* gb_last, float* g_x, int N, int H, int ndf, int n_layers, void* arena, void* stream) {
  if (!x || !w || !g_out || !gw || !gb_first || !gb_last || !arena || N <= 0 || n_layers < 1 || n_layers > 5) return UNCL_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const PdLayout p = patch_train_layout(N, H, ndf, n_layers);
  char* base = reinterpret_cast<char*>(arena);
  const size_t gsz = ((size_t)N * p.L[0].h * p.L[0].h * p.L[0].c * 4 + 255) & ~(size_t)255;
  float* gbuf[2] = {reinterpret_cast<float*>(base + p.total), reinterpret_cast<float*>(base + p.total + gsz)};
  const int ho = patch_d_size(H, n_layers, n_layers + 1);
  int rc, cur = 0;
  // last conv (stride 1, bias): input a_n
  {
    const PdLayer& l = p.L[n_layers];
    hipLaunchKernelGGL(patch_wgrad4_kernel, dim3(1, l.c), dim3(256), 0, st, g_out, reinterpret_cast<const float*>(base + l.a),
                       gw[n_layers + 1], N, l.h, l.h, l.c, ho, ho, 1, 1);
    if ((rc = bwd_colsum_f32(g_out, (long long)N * ho * ho, 1, 1, gb_last, 0, st)) != UNCL_OK) return rc;
    hipLaunchKernelGGL(patch_dgrad4_kernel, dim3(nbp((size_t)N * l.h * l.h * l.c)), dim3(256), 0, st, g_out, w[n_layers + 1], gbuf[cur],
                       N, l.h, l.h, l.c, ho, ho, 1, 1);
  }
  for (int i = n_layers; i >= 1; --i) {
    const PdLayer& q = p.L[i];
    const PdLayer& pr = p.L[i - 1];
    const int S = i < n_layers ? 2 : 1;
    const size_t ne = (size_t)N * q.h * q.h * q.c;
    // through LeakyReLU (sign of a = sign of zhat) and the norm: dL/da -> dL/dzhat -> dL/dz
    hipLaunchKernelGGL(lrelu_mask_kernel, dim3(nbp(ne)), dim3(256), 0, st, gbuf[cur], reinterpret_cast<const float*>(base + q.a), ne, 0.2f);
    if ((rc = bwd_inorm_backward(UNCL_F32, gbuf[cur], base + q.z, reinterpret_cast<const float*>(base + q.r), N, q.h * q.h, q.c, st)) !=
        UNCL_OK)
      return rc;
    hipLaunchKernelGGL(patch_wgrad4_kernel, dim3(q.c, pr.c), dim3(256), 0, st, gbuf[cur], reinterpret_cast<const float*>(base + pr.a),
                       gw[i], N, pr.h, pr.h, pr.c, q.h, q.h, q.c, S);
    hipLaunchKernelGGL(patch_dgrad4_kernel, dim3(nbp((size_t)N * pr.h * pr.h * pr.c)), dim3(256), 0, st, gbuf[cur], w[i], gbuf[cur ^ 1], N,
                       pr.h, pr.h, pr.c, q.h, q.h, q.c, S);
    cur ^= 1;
  }
  // first conv (stride 2, bias, LeakyReLU): input x (one channel)
  {
    const PdLayer& l = p.L[0];
    const size_t ne = (size_t)N * l.h * l.h * l.c;
    hipLaunchKernelGGL(lrelu_mask_kernel, dim3(nbp(ne)), dim3(256), 0, st, gbuf[cur], reinterpret_cast<const float*>(base + l.a), ne, 0.2f);
    hipLaunchKernelGGL(patch_wgrad4_kernel, dim3(l.c, 1), dim3(256), 0, st, gbuf[cur], x, gw[0], N, H, H, 1, l.h, l.h, l.c, 2);
    if ((rc = bwd_colsum_f32(gbuf[cur], (long long)N * l.h * l.h, l.c, l.c, gb_first, 0, st)) != UNCL_OK) return rc;
    if (g_x)
      hipLaunchKernelGGL(patch_dgrad4_kernel, dim3(nbp((size_t)N * H * H)), dim3(256), 0, st, gbuf[cur], w[0], g_x, N, H, H, 1, l.h, l.h,
                         l.c, 2);
  }
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}
