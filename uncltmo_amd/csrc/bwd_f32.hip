// fp32 parity mode of the generator backward pass: the pieces that are matrix-core kernels in the bf16 training path
// (weight gradients, the 2x2 transposed conv's gradients, the 3x3 data gradient's masked / accumulating store) as plain,
// DETERMINISTIC fp32 kernels.  They exist so that a whole trainer step can be compared with the CPU oracle at fp32 tolerances
// (loss scalars 1e-4, gradients 1e-3 rel-L2) -- speed is not their job: every output element is owned by
// one thread group and summed in a fixed order (no atomics), operands are read straight from global memory.
//
// Same tensors, layouts and packed weight-gradient layout ([group][tap][Cout][Cin]) as wgrad.hip / upconv2x2.hip, so the
// unpacking into the reference parameter layout is shared.
#include "bwd_internal.h"

namespace {

inline int nblocks_(size_t n, int cap = 8192) {
  const size_t b = (n + 255) / 256;
  return (int)(b < (size_t)cap ? (b ? b : 1) : (size_t)cap);
}

struct WgF {
  const float* src0;   // conv input (N, s0H, s0W, s0C) -- for the concat source: the skip x2
  const float* src1;   // concat source: the up-sampled map x1 (N, s1H, s1W, C), replicate-padded to the skip's extent
  const float* gy;     // output gradient (N, Hout, Wout, gy_ld); up == 1: (N, 2H, 2W, gy_ld)
  float* dw;           // [group][tap][Cout][Cin], accumulated into
  int N, H, W, Cin, Cout, pad, ks, concat, s0H, s0W, s0C, s1H, s1W, gy_ld, Hout, Wout, up;
};

// dw[g][tap][co][ci] += sum_p gy[p'][g Cout + co] * X[p + tap][g Cin + ci]
// grid: (Cin/32, Cout/32, taps * groups); 256 threads = 8 pixel lanes x 32 output channels, 32 input channels per thread
__global__ __launch_bounds__(256) void wgrad_f32_kernel(const WgF a) {
  __shared__ float red[8][32][33];
  const int co = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
  const int taps = a.up ? 4 : a.ks * a.ks;
  const int tap = (int)blockIdx.z % taps, grp = (int)blockIdx.z / taps;
  const int ty = a.up ? (tap >> 1) : tap / a.ks, tx = a.up ? (tap & 1) : tap % a.ks;
  float acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = 0.f;
  // concat source [x2 | x1 | x2^2 | sqrt(x2 + 1e-8)] (unet_parts.py:319-322): a 32-channel tile lies in one segment
  const int C = a.concat ? a.Cin / 4 : a.Cin;
  const int seg = a.concat ? ci0 / C : 0, cseg = a.concat ? ci0 - seg * C : ci0;
  const int dy1 = (a.s0H - a.s1H) >> 1, dx1 = (a.s0W - a.s1W) >> 1;
  const int PH = a.up ? a.H : a.Hout, PW = a.up ? a.W : a.Wout;
  const long long P = (long long)a.N * PH * PW;
  for (long long p = pl; p < P; p += 8) {
    const int n = (int)(p / ((long long)PH * PW));
    const int r = (int)(p - (long long)n * PH * PW);
    const int py = r / PW, px = r - py * PW;
    int iy, ix;
    size_t gidx;
    if (a.up) {       // x pixel (py, px), gradient pixel (2 py + dy, 2 px + dx)
      iy = py; ix = px;
      gidx = (((size_t)n * 2 * a.H + 2 * py + ty) * 2 * a.W + 2 * px + tx) * a.gy_ld;
    } else {          // output pixel (py, px), input pixel (py + ty - pad, px + tx - pad): zero outside
      iy = py + ty - a.pad; ix = px + tx - a.pad;
      gidx = (((size_t)n * a.Hout + py) * a.Wout + px) * a.gy_ld;
      if ((unsigned)iy >= (unsigned)a.H || (unsigned)ix >= (unsigned)a.W) continue;
    }
    const float g = a.gy[gidx + grp * a.Cout + co0 + co];
    const float* xp;
    if (seg == 1) {
      const int sy = min(max(iy - dy1, 0), a.s1H - 1), sx = min(max(ix - dx1, 0), a.s1W - 1);
      xp = a.src1 + (((size_t)n * a.s1H + sy) * a.s1W + sx) * C + cseg;
    } else {
      xp = a.src0 + (((size_t)n * a.s0H + iy) * a.s0W + ix) * a.s0C + grp * a.Cin + cseg;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      f32x4 v = *reinterpret_cast<const f32x4*>(xp + 4 * q);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = v[e];
        if (seg == 2) x = x * x;
        else if (seg == 3) x = sqrtf(x + 1e-8f);
        acc[4 * q + e] = fmaf(g, x, acc[4 * q + e]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 32; ++i) red[pl][co][i] = acc[i];
  __syncthreads();
  // fixed-order sum over the eight pixel lanes: thread = (co, 4 input channels)
  for (int e = threadIdx.x; e < 32 * 32; e += 256) {
    const int c = e >> 5, i = e & 31;
    float s = 0.f;
#pragma unroll
    for (int l = 0; l < 8; ++l) s += red[l][c][i];
    float* dst = a.dw + (((size_t)grp * taps + tap) * a.Cout + co0 + c) * a.Cin + ci0 + i;
    *dst += s;
  }
}

// gx[n,y,x][ci] = (sum_{tap,co} gy[n,2y+dy,2x+dx][co] * wt[tap][ci][co]) * (mask > 0 ? 1 : slope)
__global__ __launch_bounds__(256) void upconv2x2_dgrad_f32_kernel(const float* __restrict__ gy, const float* __restrict__ wt,
                                                                  const float* __restrict__ mask, float slope,
                                                                  float* __restrict__ gx, int N, int H, int W, int Cin, int Cout) {
  const size_t total = (size_t)N * H * W * Cin;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ci = (int)(i % Cin);
    size_t r = i / Cin;
    const int x = (int)(r % W); r /= W;
    const int y = (int)(r % H);
    const int n = (int)(r / H);
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float* g = gy + (((size_t)n * 2 * H + 2 * y + (t >> 1)) * 2 * W + 2 * x + (t & 1)) * Cout;
      const float* w = wt + ((size_t)t * Cin + ci) * Cout;
      for (int c = 0; c < Cout; c += 4) {
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + c), wv = *reinterpret_cast<const f32x4*>(w + c);
        s = fmaf(gv[0], wv[0], s); s = fmaf(gv[1], wv[1], s); s = fmaf(gv[2], wv[2], s); s = fmaf(gv[3], wv[3], s);
      }
    }
    if (mask) s = mask[i] > 0.f ? s : slope * s;
    gx[i] = s;
  }
}

// dst (+)= src * (mask > 0 ? 1 : slope): the gradient-mode store of the 3x3 data gradient (conv3x3_pipe's epilogue in bf16)
__global__ __launch_bounds__(256) void mask_acc_f32_kernel(const float* __restrict__ src, const float* __restrict__ mask, float slope,
                                                           float* __restrict__ dst, int accumulate, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
    if (mask) {
      const f32x4 m = reinterpret_cast<const f32x4*>(mask)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = m[e] > 0.f ? v[e] : slope * v[e];
    }
    if (accumulate) {
      const f32x4 o = reinterpret_cast<const f32x4*>(dst)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += o[e];
    }
    reinterpret_cast<f32x4*>(dst)[i] = v;
  }
}

// out[c] (+)= sum_rows x[row][c] (bias gradients), one workgroup per 32 columns, fixed order, fp64 partials
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ x, long long rows, int C, int ld, float* __restrict__ out,
                                                         int accumulate) {
  __shared__ double red[8][33];
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double s = 0.0;
  if (c < C)
    for (long long r = rg; r < rows; r += 8) s += (double)x[(size_t)r * ld + c];
  red[rg][cl] = s;
  __syncthreads();
  if (rg == 0 && c < C) {
    double t = 0.0;
    for (int i = 0; i < 8; ++i) t += red[i][cl];
    out[c] = accumulate ? out[c] + (float)t : (float)t;
  }
}

// MaxPool2d(2) copy of an NHWC tensor (the fp32 forward pools inside the next conv's loader; the backward pass wants the
// pooled tensor itself as the weight gradient's input)
__global__ __launch_bounds__(256) void maxpool2_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C) {
  const int Hp = H / 2, Wp = W / 2;
  const size_t total = (size_t)N * Hp * Wp * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    size_t r = i / C;
    const int px = (int)(r % Wp); r /= Wp;
    const int py = (int)(r % Hp);
    const int n = (int)(r / Hp);
    const float* b = x + (((size_t)n * H + 2 * py) * W + 2 * px) * C + c;
    y[i] = fmaxf(fmaxf(b[0], b[C]), fmaxf(b[(size_t)W * C], b[(size_t)W * C + C]));
  }
}

}  // namespace

int bwd_wgrad_f32(const uncl_conv_desc* d, const void* gy, float* dw, hipStream_t s) {
  if (!d || !gy || !dw || d->Cin % 32 != 0 || d->Cout % 32 != 0 || !d->src0) return UNCL_ERR_ARG;
  if (d->src_mode != UNCL_SRC_PLAIN && d->src_mode != UNCL_SRC_CONCAT_SSR) return UNCL_ERR_ARG;
  WgF a;
  a.src0 = (const float*)d->src0; a.src1 = (const float*)d->src1; a.gy = (const float*)gy; a.dw = dw;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.ks = d->ksize; a.pad = d->ksize == 3 ? d->pad : 0;
  a.concat = d->src_mode == UNCL_SRC_CONCAT_SSR;
  a.s0H = d->src0_H; a.s0W = d->src0_W; a.s0C = d->src0_C; a.s1H = d->src1_H; a.s1W = d->src1_W;
  if (a.concat && (d->src1 == nullptr || d->Cin != 4 * d->src0_C || (d->src0_C % 32) != 0)) return UNCL_ERR_ARG;
  a.Hout = d->H + 2 * a.pad - (d->ksize - 1); a.Wout = d->W + 2 * a.pad - (d->ksize - 1);
  a.gy_ld = d->out_C > 0 ? d->out_C : d->Cout;
  a.up = 0;
  const int groups = (d->ksize == 1 && d->z_mode == UNCL_Z_GROUPS && d->groups > 1) ? d->groups : 1;
  hipLaunchKernelGGL(wgrad_f32_kernel, dim3(d->Cin / 32, d->Cout / 32, d->ksize * d->ksize * groups), dim3(256), 0, s, a);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

int bwd_upconv2x2_wgrad_f32(const void* x, const void* gy, float* dw, int N, int H, int W, int C, int Cout, hipStream_t s) {
  if (!x || !gy || !dw || C % 32 != 0 || Cout % 32 != 0) return UNCL_ERR_ARG;
  WgF a = {};
  a.src0 = (const float*)x; a.src1 = nullptr; a.gy = (const float*)gy; a.dw = dw;
  a.N = N; a.H = H; a.W = W; a.Cin = C; a.Cout = Cout; a.ks = 1; a.pad = 0; a.concat = 0;
  a.s0H = H; a.s0W = W; a.s0C = C; a.gy_ld = Cout; a.Hout = H; a.Wout = W; a.up = 1;
  hipLaunchKernelGGL(wgrad_f32_kernel, dim3(C / 32, Cout / 32, 4), dim3(256), 0, s, a);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

int bwd_upconv2x2_dgrad_f32(const void* gy, const void* wt, const void* mask, float slope, void* gx, int N, int H, int W, int Cin,
                            int Cout, hipStream_t s) {
  if (!gy || !wt || !gx || Cout % 4 != 0) return UNCL_ERR_ARG;
  hipLaunchKernelGGL(upconv2x2_dgrad_f32_kernel, dim3(nblocks_((size_t)N * H * W * Cin)), dim3(256), 0, s, (const float*)gy,
                     (const float*)wt, (const float*)mask, slope, (float*)gx, N, H, W, Cin, Cout);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

int bwd_mask_acc_f32(const void* src, const void* mask, float slope, void* dst, int accumulate, long long n, hipStream_t s) {
  if (!src || !dst || n <= 0 || n % 4 != 0) return UNCL_ERR_ARG;
  hipLaunchKernelGGL(mask_acc_f32_kernel, dim3(nblocks_((size_t)n / 4)), dim3(256), 0, s, (const float*)src, (const float*)mask, slope,
                     (float*)dst, accumulate, (size_t)n / 4);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

int bwd_colsum_f32(const void* x, long long rows, int C, int ld, float* out, int accumulate, hipStream_t s) {
  if (!x || !out || rows <= 0 || C <= 0) return UNCL_ERR_ARG;
  hipLaunchKernelGGL(colsum_f32_kernel, dim3((C + 31) / 32), dim3(256), 0, s, (const float*)x, rows, C, ld, out, accumulate);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

int bwd_maxpool2_f32(const void* x, void* y, int N, int H, int W, int C, hipStream_t s) {
  if (!x || !y) return UNCL_ERR_ARG;
  hipLaunchKernelGGL(maxpool2_f32_kernel, dim3(nblocks_((size_t)N * (H / 2) * (W / 2) * C)), dim3(256), 0, s, (const float*)x, (float*)y,
                     N, H, W, C);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}
