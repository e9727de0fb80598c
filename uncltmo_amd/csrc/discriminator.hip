// SimpleDiscriminator forward + backward (models/Discriminator.py:87-126), fp32 throughout.
//
//   h1 = lrelu(conv4x4 s2 (x; w0,b0))      (N,127,127,16) NHWC
//   h2 = lrelu(conv4x4 s2 (h1; w2,b2))     (N,62,62,32)   NHWC
//   fea = conv1x1(h2; w4,b4)               (N,62,62)
//   out = <fea, wl>                        (N)       tail Linear(3844 -> 1, no bias)
//   fea_final = [mean(fea), mean(gauss local variance of fea)]   (uncl_gauss_stats on fea)
//
// The network is 71.5 MFLOP per frame and is evaluated 7 times per training step on <= a few dozen frames.  Everything stays
// fp32 (parity with the CPU reference to 1e-4): the one-channel first layer (forward, weight and data gradient) is direct
// VALU convolution; the 16 -> 32 second layer (88 % of the FLOPs), its weight gradient and its data gradient run on the fp32
// matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation) from register-resident weights / accumulators.
#include "common.h"

namespace {

constexpr int H0 = 256, H1 = 127, H2 = 62;
constexpr int C1 = 16, C2 = 32;
constexpr float SLOPE = 0.2f;

__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : SLOPE * v; }

// ---- conv1: one thread = two horizontally adjacent output pixels, all 16 channels (every weight vector read from LDS feeds
// both pixels; the 4 x 6 input window is shared)
__global__ __launch_bounds__(256) void d_conv1_kernel(const float* __restrict__ x, const float* __restrict__ w0,
                                                      const float* __restrict__ b0, float* __restrict__ h1, int N) {
  __shared__ __attribute__((aligned(16))) float sw[16 * C1 + C1];  // [tap][co], then bias
  for (int i = threadIdx.x; i < 16 * C1; i += 256) {
    const int co = i % C1, tap = i / C1;
    sw[i] = w0[co * 16 + tap];  // reference layout (co, 1, 4, 4)
  }
  if (threadIdx.x < C1) sw[16 * C1 + threadIdx.x] = b0[threadIdx.x];
  __syncthreads();
  constexpr int HP = (H1 + 1) / 2;                         // pixel pairs per output row (the last pair of a row is half empty)
  const size_t total = (size_t)N * H1 * HP;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int pxp = (int)(i % HP), oy = (int)((i / HP) % H1), n = (int)(i / ((size_t)HP * H1));
    const int ox = 2 * pxp;
    const bool two = ox + 1 < H1;
    const float* xp = x + ((size_t)n * H0 + 2 * oy) * H0 + 2 * ox;
    float in[4][6];
#pragma unroll
    for (int ky = 0; ky < 4; ++ky)
#pragma unroll
      for (int kx = 0; kx < 6; ++kx) in[ky][kx] = (kx < 4 || two) ? xp[ky * H0 + kx] : 0.f;
    float acc[2][C1];
#pragma unroll
    for (int c = 0; c < C1; ++c) acc[0][c] = acc[1][c] = sw[16 * C1 + c];
#pragma unroll
    for (int ky = 0; ky < 4; ++ky)
#pragma unroll
      for (int kx = 0; kx < 4; ++kx) {
        const float v0 = in[ky][kx], v1 = in[ky][kx + 2];
#pragma unroll
        for (int c4 = 0; c4 < C1; c4 += 4) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(sw + (ky * 4 + kx) * C1 + c4);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[0][c4 + e] = fmaf(v0, wv[e], acc[0][c4 + e]);
            acc[1][c4 + e] = fmaf(v1, wv[e], acc[1][c4 + e]);
          }
        }
      }
    float* o = h1 + (((size_t)n * H1 + oy) * H1 + ox) * C1;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if (p == 1 && !two) break;
#pragma unroll
      for (int c = 0; c < C1; c += 4)
        *reinterpret_cast<f32x4*>(o + p * C1 + c) =
            f32x4{lrelu(acc[p][c]), lrelu(acc[p][c + 1]), lrelu(acc[p][c + 2]), lrelu(acc[p][c + 3])};
    }
  }
}

// ---- conv2 + 1x1 head on the matrix cores (exact fp32 products: v_mfma_f32_32x32x2_f32).  The layer is a
// (N*3844 pixels) x (32 channels) x (K = 16 taps * 16 channels) GEMM; a VALU form (one thread = one pixel x 8 channels, 102 us
// for 32 frames) spent its time re-staging the 32 KB weight matrix for every 64-pixel tile and reading LDS once per 2.7
// FMAs; this one takes 40 us.  A workgroup is two waves that keep the
// WHOLE weight matrix in registers (lane (co, half) holds w[co][tap][8*half + j]: 128 values) and walk (frame, 8x8 tile) work
// items persistently; each item stages its 18 x 18 x 16 input patch (rows padded to 20 floats: conflict-free b128 reads) and
// every wave runs 128 MFMAs for its 32 pixels.  The 8x8 tiles define the partial-sum layout the tail reduction expects.
__global__ __launch_bounds__(128) void d_conv2_head_mfma_kernel(const float* __restrict__ h1, const float* __restrict__ w2,
                                                                const float* __restrict__ b2, const float* __restrict__ w4,
                                                                const float* __restrict__ b4, const float* __restrict__ wl,
                                                                float* __restrict__ h2, float* __restrict__ fea,
                                                                float* __restrict__ partial, int n_items) {
  constexpr int PP = 20;                       // floats per staged pixel (16 channels + 4 of padding)
  __shared__ __attribute__((aligned(16))) float sa[18 * 18 * PP];
  __shared__ float sred[2][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  // A operand of MFMA (tap, j): row co = lr, k-slot = channel 8*lh + j of that tap
  float wreg[16][8];
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int j = 0; j < 8; ++j) wreg[t][j] = w2[(lr * C1 + 8 * lh + j) * 16 + t];
  float binit[16], w4r[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int co = 8 * (i >> 2) + 4 * lh + (i & 3);        // accumulator register i of this lane = output channel co
    binit[i] = b2[co];
    w4r[i] = w4[co];
  }
  const float b4v = b4[0];
  const int py = 4 * wave + (lr >> 3), px = lr & 7;        // this lane's pixel inside the 8x8 tile
  const float* bbase = sa + ((2 * py) * 18 + 2 * px) * PP + 8 * lh;
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int n = item >> 6, tile = item & 63;
    const int y0 = (tile >> 3) * 8, x0 = (tile & 7) * 8;
    __syncthreads();                                       // the previous item's readers are done with the patch
    for (int v = tid; v < 18 * 18 * 4; v += 128) {
      const int pix = v >> 2, c4 = v & 3, ly = pix / 18, lx = pix - ly * 18;
      const int gy = min(2 * y0 + ly, H1 - 1), gx = min(2 * x0 + lx, H1 - 1);
      *reinterpret_cast<f32x4*>(sa + pix * PP + c4 * 4) =
          *reinterpret_cast<const f32x4*>(h1 + (((size_t)n * H1 + gy) * H1 + gx) * C1 + c4 * 4);
    }
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = binit[i];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const float* bp = bbase + ((t >> 2) * 18 + (t & 3)) * PP;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(bp), v1 = *reinterpret_cast<const f32x4*>(bp + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[t][j], v0[j], acc, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[t][4 + j], v1[j], acc, 0, 0, 0);
    }
    const int oy = y0 + py, ox = x0 + px;
    const bool valid = oy < H2 && ox < H2;
    float head = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      acc[i] = lrelu(acc[i]);
      head = fmaf(acc[i], w4r[i], head);
    }
    if (valid) {
      float* o = h2 + (((size_t)n * H2 + oy) * H2 + ox) * C2 + 4 * lh;
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4)
        *reinterpret_cast<f32x4*>(o + 8 * q4) = f32x4{acc[4 * q4], acc[4 * q4 + 1], acc[4 * q4 + 2], acc[4 * q4 + 3]};
    }
    head += __shfl_xor(head, 32, 64);                      // the two channel halves of the pixel
    const float f = head + b4v;
    float lo = 0.f, sf = 0.f;
    if (valid && lh == 0) {
      fea[((size_t)n * H2 + oy) * H2 + ox] = f;
      lo = f * wl[oy * H2 + ox];
      sf = f;
    }
    lo = wave_sum(lo);
    sf = wave_sum(sf);
    if (lane == 0) { sred[wave][0] = lo; sred[wave][1] = sf; }
    __syncthreads();
    if (tid < 2) partial[((size_t)n * 64 + tile) * 2 + tid] = sred[0][tid] + sred[1][tid];
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
//
// With y = 2a + py, x = 2b + px the four parity classes of (y, x) all read the SAME 2x2 neighbourhood {a-1, a} x {b-1, b} of
// the gradient map and differ only in the weights (ky = py + 2u, kx = px + 2v).  That makes it a GEMM per position (a, b):
// rows = (px, ci) = 32, K = (u, v, co) = 128, one weight matrix per py -- on the fp32 matrix cores (v_mfma_f32_32x32x2_f32)
// with both weight matrices resident in registers (128 values per lane) and every B fragment (gradient values) shared by the
// two py.  Persistent workgroups of four waves walk (frame, 4 rows of a, 32 columns of b) items: wave = row a, lane = b.
__global__ __launch_bounds__(256) void d_conv2_dgrad_mfma_kernel(const float* __restrict__ g_h2pre, const float* __restrict__ w2,
                                                                 const float* __restrict__ h1, float* __restrict__ g_h1pre,
                                                                 int n_items) {
  constexpr int GP = 36;                        // floats per staged gradient pixel (32 channels + 4 of padding)
  __shared__ __attribute__((aligned(16))) float sg[5 * 33 * GP];   // rows a0-1 .. a0+3, columns b0-1 .. b0+31
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  // A operand: row lr = (px = lr >> 4, ci = lr & 15); MFMA (uv, j) of parity py: k-slot = (u, v, co = 16 lh + j)
  float wreg[2][4][16];
#pragma unroll
  for (int py = 0; py < 2; ++py)
#pragma unroll
    for (int uv = 0; uv < 4; ++uv)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int ky = py + 2 * (uv >> 1), kx = (lr >> 4) + 2 * (uv & 1);
        wreg[py][uv][j] = w2[((16 * lh + j) * C1 + (lr & 15)) * 16 + ky * 4 + kx];
      }
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    // 64 x 64 positions per frame = 16 row blocks x 2 column halves
    const int n = item >> 5, rb = (item >> 1) & 15, half = item & 1;
    const int a0 = rb * 4, b0 = half * 32;
    __syncthreads();
    for (int v = tid; v < 5 * 33 * 8; v += 256) {
      const int pix = v >> 3, c4 = v & 7, ly = pix / 33, lx = pix - ly * 33;
      const int oy = a0 - 1 + ly, ox = b0 - 1 + lx;
      f32x4 g = {0.f, 0.f, 0.f, 0.f};
      if ((unsigned)oy < (unsigned)H2 && (unsigned)ox < (unsigned)H2)
        g = *reinterpret_cast<const f32x4*>(g_h2pre + (((size_t)n * H2 + oy) * H2 + ox) * C2 + c4 * 4);
      *reinterpret_cast<f32x4*>(sg + pix * GP + c4 * 4) = g;
    }
    __syncthreads();
    f32x16 acc[2];
#pragma unroll
    for (int py = 0; py < 2; ++py)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[py][i] = 0.f;
    // this lane's position: row a = a0 + wave (staged row wave + 1), column b = b0 + lr (staged column lr + 1)
#pragma unroll
    for (int uv = 0; uv < 4; ++uv) {
      const float* gp = sg + ((wave + 1 - (uv >> 1)) * 33 + (lr + 1 - (uv & 1))) * GP + 16 * lh;
#pragma unroll
      for (int j4 = 0; j4 < 4; ++j4) {
        const f32x4 gv = *reinterpret_cast<const f32x4*>(gp + 4 * j4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[0][uv][4 * j4 + e], gv[e], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wreg[1][uv][4 * j4 + e], gv[e], acc[1], 0, 0, 0);
        }
      }
    }
    // D register 4q+e of this lane = row 8q + 4lh + e = (px = q >> 1, ci = 8 (q & 1) + 4 lh + e) of position (a, b)
    const int a = a0 + wave, bcol = b0 + lr;
#pragma unroll
    for (int py = 0; py < 2; ++py) {
      const int y = 2 * a + py;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int x = 2 * bcol + (q >> 1);
        if (y < H1 && x < H1) {
          const size_t off = (((size_t)n * H1 + y) * H1 + x) * C1 + 8 * (q & 1) + 4 * lh;
          const f32x4 hv = *reinterpret_cast<const f32x4*>(h1 + off);
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = acc[py][4 * q + e] * (hv[e] > 0.f ? 1.f : SLOPE);
          *reinterpret_cast<f32x4*>(g_h1pre + off) = o;
        }
      }
    }
  }
}

// conv2 weight / bias gradient on the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32 products): dW[co][tap*16+ci] is a 32 x 256 matrix
// reduced over ALL pixels, so persistent workgroups keep their share of it in accumulators (wave w owns taps 4w .. 4w+3 =
// two 32-column blocks) while they walk (frame, 8x8 tile) items -- per item 8 KB of gradient and a 26 KB input patch are
// staged, then every k-step (two pixels) is one A read (g^T), two B reads and two MFMAs per wave.  One partial per workgroup
// instead of one per tile: the fixed-order reduction that follows reads grid_x rows, not N*64.
__global__ __launch_bounds__(256) void d_conv2_wgrad_mfma_kernel(const float* __restrict__ g_h2pre, const float* __restrict__ h1,
                                                                 float* __restrict__ partial /* [grid][8192 + 32] */, int n_items) {
  constexpr int PP = 20;
  __shared__ __attribute__((aligned(16))) float sg[64 * C2];            // [pixel][co]
  __shared__ __attribute__((aligned(16))) float sa[18 * 18 * PP];       // [py][px][ci (+4 pad)]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  f32x16 acc[2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
  float bsum = 0.f;
  // B operand of column block b: column lr -> tap 4*wave + 2*b + (lr >> 4), channel lr & 15
  int boff[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int tap = 4 * wave + 2 * b + (lr >> 4);
    boff[b] = ((tap >> 2) * 18 + (tap & 3)) * PP + (lr & 15);
  }
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int n = item >> 6, tile = item & 63;
    const int y0 = (tile >> 3) * 8, x0 = (tile & 7) * 8;
    __syncthreads();
    for (int v = tid; v < 64 * C2 / 4; v += 256) {
      const int p = v >> 3, c4 = v & 7;
      const int oy = y0 + (p >> 3), ox = x0 + (p & 7);
      f32x4 g = {0.f, 0.f, 0.f, 0.f};
      if (oy < H2 && ox < H2) g = *reinterpret_cast<const f32x4*>(g_h2pre + (((size_t)n * H2 + oy) * H2 + ox) * C2 + c4 * 4);
      *reinterpret_cast<f32x4*>(sg + p * C2 + c4 * 4) = g;
    }
    for (int v = tid; v < 18 * 18 * 4; v += 256) {
      const int pix = v >> 2, c4 = v & 3, ly = pix / 18, lx = pix - ly * 18;
      const int gy = min(2 * y0 + ly, H1 - 1), gx = min(2 * x0 + lx, H1 - 1);
      *reinterpret_cast<f32x4*>(sa + pix * PP + c4 * 4) =
          *reinterpret_cast<const f32x4*>(h1 + (((size_t)n * H1 + gy) * H1 + gx) * C1 + c4 * 4);
    }
    __syncthreads();
#pragma unroll 8
    for (int i = 0; i < 32; ++i) {
      const int p = 2 * i + lh;                                        // this lane's pixel of the k-step
      const float a = sg[p * C2 + lr];
      const float* ap = sa + ((2 * (p >> 3)) * 18 + 2 * (p & 7)) * PP;
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, ap[boff[0]], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, ap[boff[1]], acc[1], 0, 0, 0);
    }
    if (tid < C2) {
      float b = 0.f;
      for (int p = 0; p < 64; ++p) b += sg[p * C2 + tid];
      bsum += b;
    }
  }
  float* out = partial + (size_t)blockIdx.x * (8192 + 32);
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int tap = 4 * wave + 2 * b + (lr >> 4), ci = lr & 15;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = 8 * (i >> 2) + 4 * lh + (i & 3);
      out[(co * C1 + ci) * 16 + tap] = acc[b][i];                      // reference layout (co, ci, 4, 4)
    }
  }
  if (tid < C2) out[8192 + tid] = bsum;
}

// conv1 weight / bias gradient.  Workgroup = 16x32 output pixels of one frame; thread = (tap, co).
__global__ __launch_bounds__(256) void d_conv1_wgrad_kernel(const float* __restrict__ g_h1pre, const float* __restrict__ x,
                                                            float* __restrict__ partial /* [wg][256 + 16] */) {
  __shared__ float sx[34 * 66];
  __shared__ float sg[512 * 17];         // [pixel][co], padded row
  const int n = blockIdx.y;
  const int ty = blockIdx.x / 4, tx = blockIdx.x % 4;
  const int y0 = ty * 16, x0 = tx * 32;
  // every load of a thread is requested before its first LDS write (a load -> store loop waited for each of its 32 + 9 loads in turn:
  // 54 us per launch, whatever the batch)
  {
    constexpr int NX = (34 * 66 + 255) / 256;      // 9 image samples per thread
    float vx[NX];
#pragma unroll
    for (int u = 0; u < NX; ++u) {
      const int i = min(threadIdx.x + 256 * u, 34 * 66 - 1);
      const int ly = i / 66, lx = i % 66;
      const int gy = min(2 * y0 + ly, H0 - 1), gx = min(2 * x0 + lx, H0 - 1);
      vx[u] = x[((size_t)n * H0 + gy) * H0 + gx];
    }
    f32x4 vg[8];                                   // 512 pixels x 4 channel quads = 8 float4 per thread
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int v = threadIdx.x + 256 * u, p = v >> 2, q4 = v & 3;
      const int oy = y0 + (p >> 5), ox = x0 + (p & 31);
      vg[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (oy < H1 && ox < H1) vg[u] = *reinterpret_cast<const f32x4*>(g_h1pre + (((size_t)n * H1 + oy) * H1 + ox) * C1 + q4 * 4);
    }
#pragma unroll
    for (int u = 0; u < NX; ++u)
      if (threadIdx.x + 256 * u < 34 * 66) sx[threadIdx.x + 256 * u] = vx[u];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int v = threadIdx.x + 256 * u, p = v >> 2, q4 = v & 3;
#pragma unroll
      for (int e = 0; e < 4; ++e) sg[p * 17 + q4 * 4 + e] = vg[u][e];
    }
  }
  __syncthreads();
  const int tap = threadIdx.x >> 4, co = threadIdx.x & 15;
  const int ky = tap >> 2, kx = tap & 3;
  float acc = 0.f, bsum = 0.f;
  // rows outside, columns unrolled: both LDS addresses are a row base plus a compile-time offset (the flat loop rebuilt them from p
  // every iteration -- ten instructions per multiply-add); same order of additions
  for (int py = 0; py < 16; ++py) {
    const float* gr = sg + py * 32 * 17 + co;
    const float* xr = sx + (2 * py + ky) * 66 + kx;
#pragma unroll
    for (int px = 0; px < 32; ++px) {
      const float gv = gr[px * 17];
      acc = fmaf(gv, xr[2 * px], acc);
      bsum += gv;
    }
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
  const size_t t1 = (size_t)N * H1 * ((H1 + 1) / 2);     // pixel pairs
  hipLaunchKernelGGL(d_conv1_kernel, dim3((unsigned)((t1 + 255) / 256 < 4096 ? (t1 + 255) / 256 : 4096)), dim3(256), 0, st, x,
                     w0, b0, b.h1, N);
  {
    // persistent two-wave workgroups (four per CU at 232 VGPRs), each walking its share of the (frame, tile) items
    const int items = 64 * N;
    hipLaunchKernelGGL(d_conv2_head_mfma_kernel, dim3(items < 1024 ? items : 1024), dim3(128), 0, st, b.h1, w2, b2, w4, b4, wl,
                       b.h2, b.fea, b.partial, items);
  }
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
    hipLaunchKernelGGL(d_partial_sum_kernel, dim3(2), dim3(256), 0, st, b.w4part, hb, C2 + 1, 0, C2, gw4, 1, gb4, accumulate);
    if (g_out != nullptr)
      hipLaunchKernelGGL(d_wl_bwd_kernel, dim3((H2 * H2 + 255) / 256), dim3(256), 0, st, g_out, b.fea, gwl, N, accumulate);
    else if (!accumulate)
      (void)hipMemsetAsync(gwl, 0, H2 * H2 * sizeof(float), st);
    const int w2rows = N * 64 < 512 ? N * 64 : 512;      // one partial per persistent workgroup
    hipLaunchKernelGGL(d_conv2_wgrad_mfma_kernel, dim3(w2rows), dim3(256), 0, st, b.g_h2pre, b.h1, b.w2part, N * 64);
    hipLaunchKernelGGL(d_partial_sum_kernel, dim3((8192 + 32 + 31) / 32), dim3(256), 0, st, b.w2part, w2rows, 8192 + 32, 0, 8192, gw2,
                       32, gb2, accumulate);
  }
  {
    const int items = N * 32;     // (frame, 16 row blocks, 2 column halves)
    hipLaunchKernelGGL(d_conv2_dgrad_mfma_kernel, dim3(items < 512 ? items : 512), dim3(256), 0, st, b.g_h2pre, w2, b.h1, b.g_h1pre,
                       items);
  }
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
