// SimpleDiscriminator forward + backward (models/Discriminator.py:87-126), fp32 throughout.
//
//   h1 = lrelu(conv4x4 s2 (x; w0,b0))      (N,127,127,16) NHWC
//   h2 = lrelu(conv4x4 s2 (h1; w2,b2))     (N,62,62,32)   NHWC
//   fea = conv1x1(h2; w4,b4)               (N,62,62)
//   out = <fea, wl>                        (N)       tail Linear(3844 -> 1, no bias)
//   fea_final = [mean(fea), mean(gauss local variance of fea)]   (uncl_gauss_stats on fea)
//
// The network is 71.5 MFLOP per frame and is evaluated 7 times per training step on <= a few dozen frames: it is
// latency / LDS bound, not MFMA work; direct fp32 VALU convolutions keep it bit-comparable with the CPU reference.
#include "common.h"

namespace {

constexpr int H0 = 256, H1 = 127, H2 = 62;
constexpr int C1 = 16, C2 = 32;
constexpr float SLOPE = 0.2f;

__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : SLOPE * v; }

// ---- conv1: one thread = one output pixel, all 16 channels
__global__ __launch_bounds__(256) void d_conv1_kernel(const float* __restrict__ x, const float* __restrict__ w0,
                                                      const float* __restrict__ b0, float* __restrict__ h1, int N) {
  __shared__ float sw[16 * C1 + C1];  // [tap][co], then bias
  for (int i = threadIdx.x; i < 16 * C1; i += 256) {
    const int co = i % C1, tap = i / C1;
    sw[i] = w0[co * 16 + tap];  // reference layout (co, 1, 4, 4)
  }
  if (threadIdx.x < C1) sw[16 * C1 + threadIdx.x] = b0[threadIdx.x];
  __syncthreads();
  const size_t total = (size_t)N * H1 * H1;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ox = (int)(i % H1), oy = (int)((i / H1) % H1), n = (int)(i / ((size_t)H1 * H1));
    const float* xp = x + ((size_t)n * H0 + 2 * oy) * H0 + 2 * ox;
    float acc[C1];
#pragma unroll
    for (int c = 0; c < C1; ++c) acc[c] = sw[16 * C1 + c];
#pragma unroll
    for (int ky = 0; ky < 4; ++ky)
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const float v = xp[ky * H0 + kx];
#pragma unroll
        for (int c = 0; c < C1; ++c) acc[c] = fmaf(v, sw[(ky * 4 + kx) * C1 + c], acc[c]);
      }
    float* o = h1 + i * C1;
#pragma unroll
    for (int c = 0; c < C1; c += 4)
      *reinterpret_cast<f32x4*>(o + c) = f32x4{lrelu(acc[c]), lrelu(acc[c + 1]), lrelu(acc[c + 2]), lrelu(acc[c + 3])};
  }
}

// ---- conv2 + 1x1 head: workgroup = 8x8 output pixels x 32 channels; wave g owns channels 8g..8g+7
__global__ __launch_bounds__(256) void d_conv2_head_kernel(const float* __restrict__ h1, const float* __restrict__ w2,
                                                           const float* __restrict__ b2, const float* __restrict__ w4,
                                                           const float* __restrict__ b4, const float* __restrict__ wl,
                                                           float* __restrict__ h2, float* __restrict__ fea,
                                                           float* __restrict__ partial) {
  extern __shared__ float sm[];
  float* sw = sm;                     // [k = tap*16 + ci][co]   8192 floats
  float* sa = sm + 256 * C2;          // [ci][18*18]             5184 floats
  float* sp = sa + C1 * 324;          // [4][64] head partials
  const int n = blockIdx.y;
  const int ty = blockIdx.x / 8, tx = blockIdx.x % 8;
  const int y0 = ty * 8, x0 = tx * 8;
  for (int i = threadIdx.x; i < 256 * C2; i += 256) {
    const int co = i % C2, k = i / C2, ci = k % C1, tap = k / C1;
    sw[i] = w2[(co * C1 + ci) * 16 + tap];  // reference layout (co, ci, 4, 4)
  }
  for (int i = threadIdx.x; i < C1 * 324; i += 256) {
    const int ci = i / 324, r = i % 324, ly = r / 18, lx = r % 18;
    const int gy = min(2 * y0 + ly, H1 - 1), gx = min(2 * x0 + lx, H1 - 1);
    sa[i] = h1[(((size_t)n * H1 + gy) * H1 + gx) * C1 + ci];
  }
  __syncthreads();
  const int p = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int py = p >> 3, px = p & 7;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = b2[g * 8 + j];
  for (int tap = 0; tap < 16; ++tap) {
    const int ky = tap >> 2, kx = tap & 3;
    const float* ap = sa + (2 * py + ky) * 18 + 2 * px + kx;
#pragma unroll
    for (int ci = 0; ci < C1; ++ci) {
      const float a = ap[ci * 324];
      const f32x4 wa = *reinterpret_cast<const f32x4*>(sw + (tap * C1 + ci) * C2 + g * 8);
      const f32x4 wb = *reinterpret_cast<const f32x4*>(sw + (tap * C1 + ci) * C2 + g * 8 + 4);
      acc[0] = fmaf(a, wa[0], acc[0]); acc[1] = fmaf(a, wa[1], acc[1]);
      acc[2] = fmaf(a, wa[2], acc[2]); acc[3] = fmaf(a, wa[3], acc[3]);
      acc[4] = fmaf(a, wb[0], acc[4]); acc[5] = fmaf(a, wb[1], acc[5]);
      acc[6] = fmaf(a, wb[2], acc[6]); acc[7] = fmaf(a, wb[3], acc[7]);
    }
  }
  const int oy = y0 + py, ox = x0 + px;
  const bool valid = oy < H2 && ox < H2;
  float head = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    acc[j] = lrelu(acc[j]);
    head = fmaf(acc[j], w4[g * 8 + j], head);
  }
  if (valid) {
    float* o = h2 + (((size_t)n * H2 + oy) * H2 + ox) * C2 + g * 8;
    *reinterpret_cast<f32x4*>(o) = f32x4{acc[0], acc[1], acc[2], acc[3]};
    *reinterpret_cast<f32x4*>(o + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
  }
  sp[g * 64 + p] = head;
  __syncthreads();
  if (threadIdx.x < 64) {
    float f = ((sp[p] + sp[64 + p]) + (sp[128 + p] + sp[192 + p])) + b4[0];
    float lo = 0.f, sf = 0.f;
    if (valid) {
      fea[((size_t)n * H2 + oy) * H2 + ox] = f;
      lo = f * wl[oy * H2 + ox];
      sf = f;
    }
    lo = wave_sum(lo);
    sf = wave_sum(sf);
    if (threadIdx.x == 0) {
      partial[((size_t)n * 64 + blockIdx.x) * 2] = lo;
      partial[((size_t)n * 64 + blockIdx.x) * 2 + 1] = sf;
    }
  }
}

__global__ void d_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double lo = 0.0;
  for (int t = 0; t < 64; ++t) lo += (double)partial[((size_t)n * 64 + t) * 2];
  out[n] = (float)lo;
}

// ---- backward ---------------------------------------------------------------------------------------------
// g_fea[n][p] = g_out[n] * wl[p] + g_f1[n] / 3844 + g_var[n][p]      (g_var: gradient of the variance feature)
// g_h2pre = g_fea * w4[co] * lrelu'(h2);   accumulates gw4, gb4, gwl partials
__global__ __launch_bounds__(256) void d_head_bwd_kernel(const float* __restrict__ g_out, const float* __restrict__ g_f1,
                                                         const float* __restrict__ g_var, const float* __restrict__ wl,
                                                         const float* __restrict__ w4, const float* __restrict__ h2,
                                                         const float* __restrict__ fea, float* __restrict__ g_h2pre,
                                                         float* __restrict__ gw4b4_partial /* [blocks][33] */, int N) {
  __shared__ float red[4][C2 + 1];
  const size_t total = (size_t)N * H2 * H2;
  float a4[C2 + 1];
#pragma unroll
  for (int c = 0; c <= C2; ++c) a4[c] = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int pp = (int)(i % (H2 * H2)), n = (int)(i / (H2 * H2));
    float gf = (g_out ? g_out[n] * wl[pp] : 0.f) + (g_f1 ? g_f1[n] * (1.f / (H2 * H2)) : 0.f) + (g_var ? g_var[i] : 0.f);
    const float* hp = h2 + i * C2;
    float* gp = g_h2pre + i * C2;
#pragma unroll
    for (int c = 0; c < C2; c += 4) {
      const f32x4 hv = *reinterpret_cast<const f32x4*>(hp + c);
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        a4[c + r] = fmaf(gf, hv[r], a4[c + r]);
        o[r] = gf * w4[c + r] * (hv[r] > 0.f ? 1.f : SLOPE);
      }
      *reinterpret_cast<f32x4*>(gp + c) = o;
    }
    a4[C2] += gf;
  }
#pragma unroll
  for (int c = 0; c <= C2; ++c) {
    const float s = wave_sum(a4[c]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][c] = s;
  }
  __syncthreads();
  if (threadIdx.x <= C2)
    gw4b4_partial[(size_t)blockIdx.x * (C2 + 1) + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// gwl[p] = sum_n g_out[n] * fea[n][p]
__global__ void d_wl_bwd_kernel(const float* __restrict__ g_out, const float* __restrict__ fea, float* __restrict__ gwl,
                                int N, int accumulate) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= H2 * H2) return;
  float s = 0.f;
  for (int n = 0; n < N; ++n) s = fmaf(g_out[n], fea[(size_t)n * H2 * H2 + p], s);
  gwl[p] = accumulate ? gwl[p] + s : s;
}

// conv2 data gradient: g_h1pre[n][y][x][ci] = lrelu'(h1) * sum_{ky,kx,co: (y-ky),(x-kx) even, in range} g_h2pre[(y-ky)/2][(x-kx)/2][co] w2[co][ci][ky][kx]
__global__ __launch_bounds__(256) void d_conv2_dgrad_kernel(const float* __restrict__ g_h2pre, const float* __restrict__ w2,
                                                            const float* __restrict__ h1, float* __restrict__ g_h1pre,
                                                            int N) {
  __shared__ float sw[16 * C2 * C1];  // [tap][co][ci]
  for (int i = threadIdx.x; i < 16 * C2 * C1; i += 256) {
    const int ci = i % C1, co = (i / C1) % C2, tap = i / (C1 * C2);
    sw[i] = w2[(co * C1 + ci) * 16 + tap];
  }
  __syncthreads();
  const size_t total = (size_t)N * H1 * H1;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int x = (int)(i % H1), y = (int)((i / H1) % H1), n = (int)(i / ((size_t)H1 * H1));
    float acc[C1];
#pragma unroll
    for (int c = 0; c < C1; ++c) acc[c] = 0.f;
    for (int ky = (y & 1); ky < 4; ky += 2) {
      const int oy = (y - ky) >> 1;
      if (y - ky < 0 || oy >= H2) continue;
      for (int kx = (x & 1); kx < 4; kx += 2) {
        const int ox = (x - kx) >> 1;
        if (x - kx < 0 || ox >= H2) continue;
        const float* gp = g_h2pre + (((size_t)n * H2 + oy) * H2 + ox) * C2;
        const float* wp = sw + (ky * 4 + kx) * C2 * C1;
        for (int co = 0; co < C2; ++co) {
          const float gv = gp[co];
#pragma unroll
          for (int c = 0; c < C1; ++c) acc[c] = fmaf(gv, wp[co * C1 + c], acc[c]);
        }
      }
    }
    const float* hp = h1 + i * C1;
    float* o = g_h1pre + i * C1;
#pragma unroll
    for (int c = 0; c < C1; ++c) o[c] = acc[c] * (hp[c] > 0.f ? 1.f : SLOPE);
  }
}

// conv2 weight / bias gradient.  Workgroup = one 8x8 block of output pixels of one frame; thread = one (tap, ci) column
// (256 of them) accumulating all 32 output channels; partial sums per workgroup, reduced in a fixed order afterwards.
__global__ __launch_bounds__(256) void d_conv2_wgrad_kernel(const float* __restrict__ g_h2pre, const float* __restrict__ h1,
                                                            float* __restrict__ partial /* [wg][8192 + 32] */) {
  __shared__ float sg[64 * C2];          // [pixel][co]
  __shared__ float sa[C1 * 324];         // [ci][18*18]
  const int n = blockIdx.y;
  const int ty = blockIdx.x / 8, tx = blockIdx.x % 8;
  const int y0 = ty * 8, x0 = tx * 8;
  for (int i = threadIdx.x; i < 64 * C2; i += 256) {
    const int p = i / C2, co = i % C2;
    const int oy = y0 + (p >> 3), ox = x0 + (p & 7);
    sg[i] = (oy < H2 && ox < H2) ? g_h2pre[(((size_t)n * H2 + oy) * H2 + ox) * C2 + co] : 0.f;
  }
  for (int i = threadIdx.x; i < C1 * 324; i += 256) {
    const int ci = i / 324, r = i % 324, ly = r / 18, lx = r % 18;
    const int gy = min(2 * y0 + ly, H1 - 1), gx = min(2 * x0 + lx, H1 - 1);
    sa[i] = h1[(((size_t)n * H1 + gy) * H1 + gx) * C1 + ci];
  }
  __syncthreads();
  const int tap = threadIdx.x >> 4, ci = threadIdx.x & 15;
  const int ky = tap >> 2, kx = tap & 3;
  float acc[C2];
#pragma unroll
  for (int c = 0; c < C2; ++c) acc[c] = 0.f;
  for (int p = 0; p < 64; ++p) {
    const float a = sa[ci * 324 + (2 * (p >> 3) + ky) * 18 + 2 * (p & 7) + kx];
#pragma unroll
    for (int c = 0; c < C2; c += 4) {
      const f32x4 gv = *reinterpret_cast<const f32x4*>(sg + p * C2 + c);
      acc[c] = fmaf(a, gv[0], acc[c]); acc[c + 1] = fmaf(a, gv[1], acc[c + 1]);
      acc[c + 2] = fmaf(a, gv[2], acc[c + 2]); acc[c + 3] = fmaf(a, gv[3], acc[c + 3]);
    }
  }
  float* out = partial + ((size_t)n * 64 + blockIdx.x) * (8192 + 32);
#pragma unroll
  for (int c = 0; c < C2; ++c) out[(c * C1 + ci) * 16 + tap] = acc[c];   // reference layout (co, ci, 4, 4)
  if (threadIdx.x < C2) {
    float b = 0.f;
    for (int p = 0; p < 64; ++p) b += sg[p * C2 + threadIdx.x];
    out[8192 + threadIdx.x] = b;
  }
}

// conv1 weight / bias gradient.  Workgroup = 16x32 output pixels of one frame; thread = (tap, co).
__global__ __launch_bounds__(256) void d_conv1_wgrad_kernel(const float* __restrict__ g_h1pre, const float* __restrict__ x,
                                                            float* __restrict__ partial /* [wg][256 + 16] */) {
  __shared__ float sx[34 * 66];
  __shared__ float sg[512 * 17];         // [pixel][co], padded row
  const int n = blockIdx.y;
  const int ty = blockIdx.x / 4, tx = blockIdx.x % 4;
  const int y0 = ty * 16, x0 = tx * 32;
  for (int i = threadIdx.x; i < 34 * 66; i += 256) {
    const int ly = i / 66, lx = i % 66;
    const int gy = min(2 * y0 + ly, H0 - 1), gx = min(2 * x0 + lx, H0 - 1);
    sx[i] = x[((size_t)n * H0 + gy) * H0 + gx];
  }
  for (int i = threadIdx.x; i < 512 * C1; i += 256) {
    const int p = i / C1, co = i % C1;
    const int oy = y0 + (p >> 5), ox = x0 + (p & 31);
    sg[p * 17 + co] = (oy < H1 && ox < H1) ? g_h1pre[(((size_t)n * H1 + oy) * H1 + ox) * C1 + co] : 0.f;
  }
  __syncthreads();
  const int tap = threadIdx.x >> 4, co = threadIdx.x & 15;
  const int ky = tap >> 2, kx = tap & 3;
  float acc = 0.f, bsum = 0.f;
  for (int p = 0; p < 512; ++p) {
    const float gv = sg[p * 17 + co];
    acc = fmaf(gv, sx[(2 * (p >> 5) + ky) * 66 + 2 * (p & 31) + kx], acc);
    bsum += gv;
  }
  float* out = partial + ((size_t)n * 32 + blockIdx.x) * (256 + 16);
  out[co * 16 + tap] = acc;
  if (tap == 0) out[256 + co] = bsum;
}

// out[j] (+)= sum over `count` partial rows of width `width`: 32 columns x 8 row lanes per workgroup, fp64, fixed order
__global__ __launch_bounds__(256) void d_partial_sum_kernel(const float* __restrict__ partial, int count, int width, int off0,
                                                            int n0, float* __restrict__ dst0, int n1, float* __restrict__ dst1,
                                                            int accumulate) {
  __shared__ double red[8][33];
  const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + c;
  double s = 0.0;
  if (j < n0 + n1) {
    const float* p = partial + off0 + j;
#pragma unroll 8
    for (int r = rg; r < count; r += 8) s += (double)p[(size_t)r * width];
  }
  red[rg][c] = s;
  __syncthreads();
  if (rg == 0 && j < n0 + n1) {
    double t = 0.0;
    for (int i = 0; i < 8; ++i) t += red[i][c];
    float* d = j < n0 ? dst0 + j : dst1 + (j - n0);
    *d = accumulate ? *d + (float)t : (float)t;
  }
}

// g_x[n][y][x] = sum_{ky,kx,co} g_h1pre[(y-ky)/2][(x-kx)/2][co] w0[co][ky][kx]
__global__ __launch_bounds__(256) void d_conv1_dgrad_kernel(const float* __restrict__ g_h1pre, const float* __restrict__ w0,
                                                            float* __restrict__ g_x, int N, int accumulate) {
  __shared__ float sw[16 * C1];  // [tap][co]
  for (int i = threadIdx.x; i < 16 * C1; i += 256) sw[i] = w0[(i % C1) * 16 + i / C1];
  __syncthreads();
  const size_t total = (size_t)N * H0 * H0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int x = (int)(i % H0), y = (int)((i / H0) % H0), n = (int)(i / ((size_t)H0 * H0));
    float acc = 0.f;
    for (int ky = (y & 1); ky < 4; ky += 2) {
      const int oy = (y - ky) >> 1;
      if (y - ky < 0 || oy >= H1) continue;
      for (int kx = (x & 1); kx < 4; kx += 2) {
        const int ox = (x - kx) >> 1;
        if (x - kx < 0 || ox >= H1) continue;
        const float* gp = g_h1pre + (((size_t)n * H1 + oy) * H1 + ox) * C1;
        const float* wp = sw + (ky * 4 + kx) * C1;
#pragma unroll
        for (int c = 0; c < C1; ++c) acc = fmaf(gp[c], wp[c], acc);
      }
    }
    g_x[i] = accumulate ? g_x[i] + acc : acc;
  }
}

__global__ void d_w4_final_kernel(const float* __restrict__ partial, int blocks, float* __restrict__ gw4, float* __restrict__ gb4,
                                  int accumulate) {
  const int c = threadIdx.x;
  if (c > C2) return;
  double s = 0.0;
  for (int b = 0; b < blocks; ++b) s += (double)partial[(size_t)b * (C2 + 1) + c];
  float* dst = c < C2 ? gw4 + c : gb4;
  *dst = accumulate ? *dst + (float)s : (float)s;
}

}  // namespace

extern "C" size_t uncl_simple_d_workspace_bytes(int N) {
  // h1, h2, fea, partial(fwd), g_h2pre, g_h1pre, head partials
  const size_t f = (size_t)N * H1 * H1 * C1 + (size_t)N * H2 * H2 * C2 + (size_t)N * H2 * H2 + (size_t)N * 64 * 2;
  const size_t b = (size_t)N * H2 * H2 * C2 + (size_t)N * H1 * H1 * C1 + (size_t)256 * (C2 + 1) +
                   (size_t)N * 64 * (8192 + 32) + (size_t)N * 32 * (256 + 16);
  return (f + b) * sizeof(float) + 1024;
}

namespace {
struct DBufs {
  float *h1, *h2, *fea, *partial, *g_h2pre, *g_h1pre, *w4part, *w2part, *w0part;
};
DBufs d_bufs(void* workspace, int N) {
  DBufs b;
  float* p = reinterpret_cast<float*>(workspace);
  b.h1 = p; p += (size_t)N * H1 * H1 * C1;
  b.h2 = p; p += (size_t)N * H2 * H2 * C2;
  b.fea = p; p += (size_t)N * H2 * H2;
  b.partial = p; p += (size_t)N * 64 * 2;
  b.g_h2pre = p; p += (size_t)N * H2 * H2 * C2;
  b.g_h1pre = p; p += (size_t)N * H1 * H1 * C1;
  b.w4part = p; p += (size_t)256 * (C2 + 1);
  b.w2part = p; p += (size_t)N * 64 * (8192 + 32);
  b.w0part = p;
  return b;
}
}  // namespace

// x: fp32 (N,256,256).  params in reference layout: w0 (16,1,4,4) b0 (16) w2 (32,16,4,4) b2 (32) w4 (1,32,1,1) b4 (1)
// wl (1,3844).  out: (N) logits.  fea_final: (N,2).  The workspace keeps h1/h2/fea for uncl_simple_d_backward.
// gs_workspace: scratch for uncl_gauss_stats (uncl_gauss_stats_workspace_bytes(N, 62, 1)).
extern "C" int uncl_simple_d_forward(const float* x, const float* w0, const float* b0, const float* w2, const float* b2,
                                     const float* w4, const float* b4, const float* wl, float* out, float* fea_final,
                                     int N, void* workspace, void* gs_workspace, void* stream) {
  if (!x || !w0 || !b0 || !w2 || !b2 || !w4 || !b4 || !wl || !out || !workspace || N <= 0) return UNCL_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  DBufs b = d_bufs(workspace, N);
  const size_t t1 = (size_t)N * H1 * H1;
  hipLaunchKernelGGL(d_conv1_kernel, dim3((unsigned)((t1 + 255) / 256 < 4096 ? (t1 + 255) / 256 : 4096)), dim3(256), 0, st, x,
                     w0, b0, b.h1, N);
  const size_t lds = (256 * C2 + C1 * 324 + 256) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(d_conv2_head_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return UNCL_ERR_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL(d_conv2_head_kernel, dim3(64, N), dim3(256), lds, st, b.h1, w2, b2, w4, b4, wl, b.h2, b.fea, b.partial);
  hipLaunchKernelGGL(d_final_kernel, dim3((N + 63) / 64), dim3(64), 0, st, b.partial, out, N);
  UNCL_CHECK_LAUNCH();
  if (fea_final != nullptr) {
    if (gs_workspace == nullptr) return UNCL_ERR_ARG;
    return uncl_gauss_stats(b.fea, UNCL_F32, fea_final, N, H2, H2, 1, gs_workspace, stream);
  }
  return UNCL_OK;
}

// Backward of the forward just run on `workspace`.  g_out (N) and g_f1 (N) may be NULL; g_var (N,62,62) is the
// gradient that arrives at `fea` through the variance feature (or NULL).  Parameter gradients are written (or
// accumulated) in reference layout when gw0 != NULL; g_x (N,256,256) is produced when g_x != NULL.
extern "C" int uncl_simple_d_backward(const float* x, const float* w0, const float* w2, const float* w4, const float* wl,
                                      const float* g_out, const float* g_f1, const float* g_var, float* gw0, float* gb0,
                                      float* gw2, float* gb2, float* gw4, float* gb4, float* gwl, float* g_x, int accumulate,
                                      int N, void* workspace, void* stream) {
  if (!x || !w0 || !w2 || !w4 || !wl || !workspace || N <= 0) return UNCL_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  DBufs b = d_bufs(workspace, N);
  const size_t t2 = (size_t)N * H2 * H2, t1 = (size_t)N * H1 * H1, t0 = (size_t)N * H0 * H0;
  const int hb = (int)((t2 + 255) / 256 < 256 ? (t2 + 255) / 256 : 256);
  hipLaunchKernelGGL(d_head_bwd_kernel, dim3(hb), dim3(256), 0, st, g_out, g_f1, g_var, wl, w4, b.h2, b.fea, b.g_h2pre,
                     b.w4part, N);
  const bool params = gw0 != nullptr;
  if (params) {
    if (!gb0 || !gw2 || !gb2 || !gw4 || !gb4 || !gwl) return UNCL_ERR_ARG;
    hipLaunchKernelGGL(d_w4_final_kernel, dim3(1), dim3(64), 0, st, b.w4part, hb, gw4, gb4, accumulate);
    if (g_out != nullptr)
      hipLaunchKernelGGL(d_wl_bwd_kernel, dim3((H2 * H2 + 255) / 256), dim3(256), 0, st, g_out, b.fea, gwl, N, accumulate);
    else if (!accumulate)
      (void)hipMemsetAsync(gwl, 0, H2 * H2 * sizeof(float), st);
    hipLaunchKernelGGL(d_conv2_wgrad_kernel, dim3(64, N), dim3(256), 0, st, b.g_h2pre, b.h1, b.w2part);
    hipLaunchKernelGGL(d_partial_sum_kernel, dim3((8192 + 32 + 31) / 32), dim3(256), 0, st, b.w2part, N * 64, 8192 + 32, 0, 8192, gw2,
                       32, gb2, accumulate);
  }
  hipLaunchKernelGGL(d_conv2_dgrad_kernel, dim3((unsigned)((t1 + 255) / 256 < 4096 ? (t1 + 255) / 256 : 4096)), dim3(256), 0,
                     st, b.g_h2pre, w2, b.h1, b.g_h1pre, N);
  if (params) {
    hipLaunchKernelGGL(d_conv1_wgrad_kernel, dim3(32, N), dim3(256), 0, st, b.g_h1pre, x, b.w0part);
    hipLaunchKernelGGL(d_partial_sum_kernel, dim3((256 + 16 + 31) / 32), dim3(256), 0, st, b.w0part, N * 32, 256 + 16, 0, 256, gw0, 16, gb0, accumulate);
  }
  if (g_x != nullptr)
    hipLaunchKernelGGL(d_conv1_dgrad_kernel, dim3((unsigned)((t0 + 255) / 256 < 4096 ? (t0 + 255) / 256 : 4096)), dim3(256), 0,
                       st, b.g_h1pre, w0, g_x, N, 0);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}
