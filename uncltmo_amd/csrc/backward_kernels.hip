// Element-wise / gather kernels of the generator's backward pass (bf16 activations and gradients, fp32 math).
// Every gradient tensor that is stored is already multiplied by the activation derivative of the layer that
// produced the corresponding activation ("pre-activation gradient"), so consumers never need a second mask pass.
#include "common.h"

namespace {

// eight consecutive elements as floats: one 16-byte access for the 16-bit types, two for fp32 (the parity mode of the backward
// pass keeps activations and gradients in fp32)
template <typename T>
__device__ __forceinline__ void ld8(const T* p, float* f) {
  if constexpr (sizeof(T) == 2) {
    Elem<T>::unpack(*reinterpret_cast<const typename Elem<T>::vec*>(p), f);
  } else {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[i] = a[i]; f[4 + i] = b[i]; }
  }
}
template <typename T>
__device__ __forceinline__ void st8(T* p, const float* f) {
  if constexpr (sizeof(T) == 2) {
    *reinterpret_cast<typename Elem<T>::vec*>(p) = Elem<T>::pack(f);
  } else {
    *reinterpret_cast<f32x4*>(p) = f32x4{f[0], f[1], f[2], f[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{f[4], f[5], f[6], f[7]};
  }
}

// ---- outconv (1x1 to one channel) + sigmoid backward (Unet_singleFrame.py:207-209)
// g_pre[p] = g_out[p] s (1 - s);  G_up[p][c] = (g_upx[p][c] + g_pre[p] w[c]) * [up_x[p][c] > 0 ? 1 : slope]
// partial[block][33] = sum_p g_pre[p] up_x[p][c] (c < 32), sum_p g_pre[p]
template <typename T>
__global__ __launch_bounds__(256) void outc_bwd_kernel(const float* __restrict__ g_out, const float* __restrict__ x_out,
                                                       const T* __restrict__ g_upx, const T* __restrict__ up_x,
                                                       const float* __restrict__ w, T* __restrict__ G_up,
                                                       float* __restrict__ partial, size_t P, int last_act, float slope) {
  __shared__ float red[4][33];
  // thread = (pixel, 8-channel vector): 4 threads per pixel; the vector index is the same for every element a thread visits
  // (the grid stride is a multiple of 4), so its eight weight-gradient sums live in registers
  float aw[8], ab = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) aw[c] = 0.f;
  const int v = (int)(threadIdx.x & 3);
  float wv[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) wv[c] = w[v * 8 + c];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < P * 4; i += (size_t)gridDim.x * 256) {
    const size_t p = i >> 2;
    const float s = x_out[p];
    // derivative of the last activation expressed through its output s (Unet_singleFrame.py:207-212, Blocks.py:85-91)
    float gp = g_out[p];
    if (last_act == UNCL_ACT_SIGMOID) gp *= s * (1.f - s);
    else if (last_act == UNCL_ACT_TANH) gp *= 1.f - s * s;
    else if (last_act == UNCL_ACT_MSIG) gp *= 3.f * s * (1.f - s);
    float u[8], gu[8], o[8];
    ld8(up_x + p * 32 + v * 8, u);
    if (g_upx) ld8(g_upx + p * 32 + v * 8, gu);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float t = (g_upx ? gu[c] : 0.f) + gp * wv[c];
      o[c] = u[c] > 0.f ? t : slope * t;
      aw[c] = fmaf(gp, u[c], aw[c]);
    }
    if (v == 0) ab += gp;
    st8(G_up + p * 32 + v * 8, o);
  }
  // sum over the 16 lanes of a wave that share the vector index (xor 4, 8, 16, 32), lanes 0..3 hold the results
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float t = aw[c];
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) t += __shfl_xor(t, o, 64);
    if ((threadIdx.x & 63) < 4) red[threadIdx.x >> 6][v * 8 + c] = t;
  }
  const float tb = wave_sum(ab);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][32] = tb;
  __syncthreads();
  if (threadIdx.x < 33)
    partial[(size_t)blockIdx.x * 33 + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// out[c] (+)= sum over `rows` partial rows, fp64, fixed order: CPB columns x (256 / CPB) row lanes per workgroup, every lane's loads
// in flight before its first addition, the lanes' sums added in lane order by one thread per column.  (Rounds 1 - 4a: 32 columns x 8
// row lanes with one dependent load per addition -- the 1024-row sum of the last layer's 33 gradient sums was a 32 - 40 us launch of
// two workgroups on the critical path of every frame; 4 columns x 64 lanes: 16 rows per lane.)
// `out_tail`: columns >= tail_from go to out_tail[c - tail_from] (the 33 sums of the last layer: 32 weights, 1 bias).
template <int CPB>
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ partial, int rows, int cols, float* __restrict__ out,
                                                          int accumulate, float* __restrict__ out_tail, int tail_from) {
  constexpr int LANES = 256 / CPB;
  __shared__ double red[LANES][CPB + 1];
  const int cl = threadIdx.x % CPB, rg = threadIdx.x / CPB;
  const int c = blockIdx.x * CPB + cl;
  double s = 0.0;
  if (c < cols) {
    constexpr int U = 8;
    for (int r = rg; r < rows; r += LANES * U) {
      float v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = (r + LANES * u < rows) ? partial[(size_t)(r + LANES * u) * cols + c] : 0.f;
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (r + LANES * u < rows) s += (double)v[u];
    }
  }
  red[rg][cl] = s;
  __syncthreads();
  if (rg == 0 && c < cols) {
    double t = 0.0;
    for (int i = 0; i < LANES; ++i) t += red[i][cl];
    float* o = (out_tail != nullptr && c >= tail_from) ? out_tail + (c - tail_from) : out + c;
    *o = accumulate ? *o + (float)t : (float)t;
  }
}

// ---- skip-concat backward (unet_parts.py:319-322, 292-298)
// g_cat: (N,H,W,4C).  G_x2 = (g0 + 2 x2 g2 + g3 / (2 sqrt(x2 + eps))) * relu'(x2)   [written or accumulated]
// G_x1 (N,H1,W1,C) = g1 with the replicate-padded border folded back (clamped positions accumulate)
template <typename T>
__global__ __launch_bounds__(256) void ssr_bwd_kernel(const T* __restrict__ g_cat, const T* __restrict__ x2,
                                                      T* __restrict__ G_x2, T* __restrict__ G_x1, int N, int H, int W,
                                                      int C, int H1, int W1, float slope, int accumulate_x2) {
  const int VC = C / 8;
  const size_t total = (size_t)N * H * W * VC;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int v = (int)(i % VC);
    const size_t p = i / VC;
    const T* gc = g_cat + p * 4 * C + v * 8;
    float g0[8], g2[8], g3[8], xv[8], o[8];
    ld8(gc, g0);
    ld8(gc + 2 * C, g2);
    ld8(gc + 3 * C, g3);
    ld8(x2 + p * C + v * 8, xv);
    if (accumulate_x2) ld8(G_x2 + p * C + v * 8, o);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float t = g0[c] + 2.f * xv[c] * g2[c] + g3[c] * 0.5f / sqrtf(xv[c] + 1e-8f);
      const float m = xv[c] > 0.f ? t : slope * t;
      o[c] = accumulate_x2 ? o[c] + m : m;
    }
    st8(G_x2 + p * C + v * 8, o);
  }
  // x1 gradient: one thread per (x1 pixel, vector) gathers the skip-resolution pixels that were clamped onto it
  const int dy = (H - H1) >> 1, dx = (W - W1) >> 1;
  const size_t total1 = (size_t)N * H1 * W1 * VC;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total1; i += (size_t)gridDim.x * 256) {
    const int v = (int)(i % VC);
    size_t r = i / VC;
    const int sx = (int)(r % W1); r /= W1;
    const int sy = (int)(r % H1);
    const int n = (int)(r / H1);
    const int y_lo = sy == 0 ? 0 : sy + dy, y_hi = sy == H1 - 1 ? H - 1 : sy + dy;
    const int x_lo = sx == 0 ? 0 : sx + dx, x_hi = sx == W1 - 1 ? W - 1 : sx + dx;
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    for (int y = y_lo; y <= y_hi; ++y)
      for (int x = x_lo; x <= x_hi; ++x) {
        float g1[8];
        ld8(g_cat + (((size_t)n * H + y) * W + x) * 4 * C + C + v * 8, g1);
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] += g1[c];
      }
    st8(G_x1 + (((size_t)n * H1 + sy) * W1 + sx) * C + v * 8, acc);
  }
}

// ---- 2x2 max-pool backward: the gradient of each pooled pixel goes to the first maximum of its window
// (cin / cout / pc: the recurrent hand-off of a clip's head channels on g_pool, folded in -- what head_handoff_kernel without a mask
// does to the first pc channels of every pooled pixel, with the same rounding: the head gradient leaves to cout and is replaced by
// cin's; this kernel is the only reader of g_pool after that point)
template <typename T>
__global__ __launch_bounds__(256) void pool_bwd_kernel(const T* __restrict__ g_pool, const T* __restrict__ x,
                                                       T* __restrict__ G_x, int N, int H, int W, int C, float slope,
                                                       int accumulate, const T* __restrict__ cin, T* __restrict__ cout, int pc) {
  const int VC = C / 8, Hp = H / 2, Wp = W / 2;
  const size_t total = (size_t)N * Hp * Wp * VC;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int v = (int)(i % VC);
    size_t r = i / VC;
    const int px = (int)(r % Wp); r /= Wp;
    const int py = (int)(r % Hp);
    const int n = (int)(r / Hp);
    float g[8], xs[4][8];
    const size_t ppix = ((size_t)n * Hp + py) * Wp + px;
    ld8(g_pool + ppix * C + v * 8, g);
    if (v == 0 && (cin != nullptr || cout != nullptr)) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (c < pc) {
          float hx = g[c];
          if (cout) { cout[ppix * pc + c] = (T)hx; hx = 0.f; }
          if (cin) hx += (float)cin[ppix * pc + c];
          g[c] = (float)(T)hx;          // (the separate kernel stored it in the gradient's type before this one read it)
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      ld8(x + (((size_t)n * H + 2 * py + (q >> 1)) * W + 2 * px + (q & 1)) * C + v * 8, xs[q]);
    int best[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      int b = 0;
      float m = xs[0][c];
#pragma unroll
      for (int q = 1; q < 4; ++q)
        if (xs[q][c] > m) { m = xs[q][c]; b = q; }
      best[c] = b;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      T* dst = G_x + (((size_t)n * H + 2 * py + (q >> 1)) * W + 2 * px + (q & 1)) * C + v * 8;
      float o[8];
      if (accumulate) ld8(dst, o);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float t = best[c] == q ? (xs[q][c] > 0.f ? g[c] : slope * g[c]) : 0.f;
        o[c] = accumulate ? o[c] + t : t;
      }
      st8(dst, o);
    }
  }
}

// ---- GELU forward / backward on [rows][C] with an optional per-sample scale (DropPath) on the backward
template <typename T>
__global__ void gelu_fwd_kernel(const T* __restrict__ z, T* __restrict__ h, size_t nvec) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    float f[8];
    ld8(z + i * 8, f);
#pragma unroll
    for (int c = 0; c < 8; ++c) f[c] = uncl_gelu(f[c]);
    st8(h + i * 8, f);
  }
}
template <typename T>
__global__ void gelu_bwd_kernel(const T* __restrict__ g_h, const T* __restrict__ z, T* __restrict__ g_z, size_t nvec) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    float g[8], f[8];
    ld8(g_h + i * 8, g);
    ld8(z + i * 8, f);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float x = f[c];
      const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
      const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
      g[c] *= cdf + x * pdf;
    }
    st8(g_z + i * 8, g);
  }
}

// y = x * scale[n] (per-sample DropPath factor) on [N][per] bf16; scale == NULL -> copy
template <typename T>
__global__ void scale_rows_kernel(const T* __restrict__ x, const float* __restrict__ scale, T* __restrict__ y,
                                  size_t per_vec, size_t nvec) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    float f[8];
    ld8(x + i * 8, f);
    const float s = scale ? scale[i / per_vec] : 1.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) f[c] *= s;
    st8(y + i * 8, f);
  }
}

// G = g * [(x - pe) > 0 ? 1 : slope]   (x = relu(v) + pos_embed, Unet_singleFrame.py:94); pe broadcast over samples
template <typename T>
__global__ void mask_minus_kernel(const T* __restrict__ g, const T* __restrict__ x, const T* __restrict__ pe,
                                  T* __restrict__ out, size_t per_vec, size_t nvec, float slope) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    float gg[8], xx[8], pp[8];
    ld8(g + i * 8, gg);
    ld8(x + i * 8, xx);
    ld8(pe + (i % per_vec) * 8, pp);
#pragma unroll
    for (int c = 0; c < 8; ++c) gg[c] = (xx[c] - pp[c]) > 0.f ? gg[c] : slope * gg[c];
    st8(out + i * 8, gg);
  }
}

// sum over samples: out[e] = sum_n g[n][e]   (pos_embed gradient), fp32 out
template <typename T>
__global__ void sum_samples_kernel(const T* __restrict__ g, float* __restrict__ out, int N, size_t per, int accumulate) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < per; e += (size_t)gridDim.x * 256) {
    float s = accumulate ? out[e] : 0.f;
    for (int n = 0; n < N; ++n) s += (float)g[(size_t)n * per + e];
    out[e] = s;
  }
}

// ---- recurrent hand-off of the video generator (Unet.py:244,270): from the second frame on, the first pc = C/32
// channels entering a down / up stage are the previous frame's.  Backward, for the gradient g of such a mixed input:
//   carry_out[pix][c] = g[pix][c], g[pix][c] = 0      (c < pc; the head gradient belongs to the previous frame)
//   g[pix][c] += carry_in[pix][c]                      (c < pc; what the NEXT frame's consumer sent back to this frame)
// and then, if mask is given, the activation derivative of the layer that produced this frame's tensor on all channels.
template <typename T>
__global__ void head_handoff_kernel(T* __restrict__ g, const T* __restrict__ mask, float slope,
                                    const T* __restrict__ cin, T* __restrict__ cout, size_t npix, int C, int pc) {
  const int VC = mask ? C / 8 : 1;  // without a mask only the first vector of every pixel changes
  const size_t total = npix * VC;
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
    const size_t pix = t / VC;
    const int v = (int)(t - pix * VC);
    float f[8];
    ld8(g + pix * C + v * 8, f);
    if (v == 0) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (c < pc) {
          float x = f[c];
          if (cout) { cout[pix * pc + c] = (T)x; x = 0.f; }
          if (cin) x += (float)cin[pix * pc + c];
          f[c] = x;
        }
      }
    }
    if (mask) {
      float m[8];
      ld8(mask + pix * C + v * 8, m);
#pragma unroll
      for (int c = 0; c < 8; ++c) f[c] = m[c] > 0.f ? f[c] : slope * f[c];
    }
    st8(g + pix * C + v * 8, f);
  }
}

// out = x with the first pc channels of every pixel taken from prev (the mixed tensor a stage of frame k > 0 consumed)
template <typename T>
// (own / lag: a whole clip in one launch -- the first `own` pixels, frame 0, keep their own heads; pixel p >= own takes them from
// pixel p - lag of `prev`, i.e. from the frame before it when prev == x and lag == own == one frame's pixels)
__global__ void mix_heads_kernel(const T* __restrict__ x, const T* __restrict__ prev, T* __restrict__ out,
                                 size_t npix, int C, int pc, size_t own, size_t lag) {
  const int VC = C / 8;
  const size_t total = npix * VC;
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
    const size_t pix = t / VC;
    const int v = (int)(t - pix * VC);
    float a[8];
    ld8(x + pix * C + v * 8, a);
    if (v == 0 && pix >= own) {
      float b[8];
      ld8(prev + (pix - lag) * C, b);
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (c < pc) a[c] = b[c];
    }
    st8(out + pix * C + v * 8, a);
  }
}

// ---- max-relative backward (torch_vertex.py:22-29): out[2c] = x_c, out[2c+1] = max_k (x_c[nbr_k] - x_c)
// g_x[i][c] = g[i][2c] - g[i][2c+1];  g_x[nbr*(i,c)][c] += g[i][2c+1]   (fp32 atomics into a zeroed fp32 buffer)
template <typename T>
__global__ __launch_bounds__(256) void maxrel_bwd_kernel(const T* __restrict__ g_out, const T* __restrict__ x,
                                                         const int32_t* __restrict__ idx, float* __restrict__ g_x, int N, int n,
                                                         int C, int k) {
  const int VC = C / 8;
  const size_t total = (size_t)N * n * VC;
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
    const int cv = (int)(t % VC);
    const size_t node = t / VC;
    const size_t b = node / n;
    float xi[8], m[8], g0[16];
    int arg[8];
    ld8(x + node * C + cv * 8, xi);
    ld8(g_out + node * 2 * C + (size_t)cv * 16, g0);
    ld8(g_out + node * 2 * C + (size_t)cv * 16 + 8, g0 + 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) { m[e] = -INFINITY; arg[e] = 0; }
    for (int r = 0; r < k; ++r) {
      const int j = idx[node * k + r];
      float xj[8];
      ld8(x + (b * n + j) * C + cv * 8, xj);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = xj[e] - xi[e];
        if (d > m[e]) { m[e] = d; arg[e] = j; }   // first maximum, like torch.max
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float gx = g0[2 * e], gr = g0[2 * e + 1];
      atomicAdd(g_x + node * C + cv * 8 + e, gx - gr);
      atomicAdd(g_x + (b * n + arg[e]) * C + cv * 8 + e, gr);
    }
  }
}

// 16-bit fast path: ONE launch, no global atomics.  A workgroup owns (sample, 32-channel slice): the slice of x and the sample's
// graph go to LDS, the two contributions of every (node, channel) are LDS atomics into an fp32 image [n][33] (unit = (8-channel
// group, node) with the node fastest: a wave's self-terms land on 64 different banks), and the image leaves as 16-bit rows.
// The global-atomic form above took 94 us at N = 32 (2.4 M fp32 atomics on 4.7 MB) plus a memset and a conversion launch.
constexpr int MRB_CH = 32, MRB_PITCH = MRB_CH + 1;
static size_t maxrel_bwd_lds_bytes(int n, int k, size_t es) {
  return (size_t)n * MRB_PITCH * 4 + (size_t)n * MRB_CH * es + (size_t)n * k * 4;
}
template <typename T>
__global__ __launch_bounds__(256) void maxrel_bwd_lds_kernel(const T* __restrict__ g_out, const T* __restrict__ x,
                                                             const int32_t* __restrict__ idx, T* __restrict__ g_x, int n, int C,
                                                             int k) {
  extern __shared__ __attribute__((aligned(16))) char mrb_smem[];
  float* acc = reinterpret_cast<float*>(mrb_smem);                                        // [n][33]
  T* sx = reinterpret_cast<T*>(mrb_smem + (size_t)n * MRB_PITCH * 4);                     // [n][32]
  int* sidx = reinterpret_cast<int*>(mrb_smem + (size_t)n * MRB_PITCH * 4 + (size_t)n * MRB_CH * sizeof(T));   // [n][k]
  const int tid = threadIdx.x;
  const size_t b = blockIdx.x;
  const int c0 = blockIdx.y * MRB_CH;
  const int units = n * (MRB_CH / 8);
  for (int u = tid; u < units; u += 256) {
    const int q = u / n, node = u - q * n;
    float v[8];
    ld8(x + (b * n + node) * C + c0 + q * 8, v);
    st8(sx + node * MRB_CH + q * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[node * MRB_PITCH + q * 8 + e] = 0.f;
  }
  for (int u = tid; u < n * k; u += 256) sidx[u] = min(max(idx[b * n * k + u], 0), n - 1);      // (an index outside the graph must not leave the LDS image)
  __syncthreads();
  for (int u = tid; u < units; u += 256) {
    const int q = u / n, node = u - q * n;
    float xi[8], m[8], g0[16];
    int arg[8];
    ld8(sx + node * MRB_CH + q * 8, xi);
    const T* gp = g_out + (b * n + node) * 2 * C + (size_t)(c0 + q * 8) * 2;
    ld8(gp, g0);
    ld8(gp + 8, g0 + 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) { m[e] = -INFINITY; arg[e] = 0; }
    for (int r = 0; r < k; ++r) {
      const int j = sidx[node * k + r];
      float xj[8];
      ld8(sx + j * MRB_CH + q * 8, xj);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = xj[e] - xi[e];
        if (d > m[e]) { m[e] = d; arg[e] = j; }   // first maximum, like torch.max
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float gx = g0[2 * e], gr = g0[2 * e + 1];
      atomicAdd(acc + node * MRB_PITCH + q * 8 + e, gx - gr);
      atomicAdd(acc + arg[e] * MRB_PITCH + q * 8 + e, gr);
    }
  }
  __syncthreads();
  for (int u = tid; u < units; u += 256) {
    const int q = u / n, node = u - q * n;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = acc[node * MRB_PITCH + q * 8 + e];
    st8(g_x + (b * n + node) * C + c0 + q * 8, v);
  }
}

// the same without atomics (fp32 parity mode: bit-reproducible): one thread per (sample, channel) walks the nodes in order and
// adds into its own column of an LDS image [n][64]
template <typename T>
__global__ __launch_bounds__(64) void maxrel_bwd_det_kernel(const T* __restrict__ g_out, const T* __restrict__ x,
                                                            const int32_t* __restrict__ idx, T* __restrict__ g_x, int n, int C, int k) {
  extern __shared__ float acc_lds[];                 // [n][64]
  const int b = blockIdx.x, c = blockIdx.y * 64 + threadIdx.x, t = threadIdx.x;
  for (int j = 0; j < n; ++j) acc_lds[j * 64 + t] = 0.f;
  for (int i = 0; i < n; ++i) {
    const size_t node = (size_t)b * n + i;
    const float xi = (float)x[node * C + c];
    float m = -INFINITY;
    int arg = 0;
    for (int r = 0; r < k; ++r) {
      const int j = idx[node * k + r];
      const float d = (float)x[((size_t)b * n + j) * C + c] - xi;
      if (d > m) { m = d; arg = j; }                 // first maximum, like torch.max
    }
    const float gx = (float)g_out[node * 2 * C + 2 * c], gr = (float)g_out[node * 2 * C + 2 * c + 1];
    acc_lds[i * 64 + t] += gx - gr;
    acc_lds[arg * 64 + t] += gr;
  }
  for (int j = 0; j < n; ++j) g_x[((size_t)b * n + j) * C + c] = (T)acc_lds[j * 64 + t];
}

template <typename T>
__global__ void f32_to_bf16_kernel(const float* __restrict__ x, T* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = (T)x[i];
}

// ---- first layer (Cin = 1) weight / bias gradient: dw[co][tap] = sum_p G[p][co] x[p + tap]
template <typename T>
__global__ __launch_bounds__(256) void conv_in_wgrad_kernel(const T* __restrict__ G, const float* __restrict__ x,
                                                            float* __restrict__ partial, int N, int H, int W) {
  // G: (N, H-2, W-2, 32).  thread = (pixel, 8-channel vector); partial[block][32*10]
  __shared__ float red[4][320];
  const int Ho = H - 2, Wo = W - 2;
  float acc[8][10];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int t = 0; t < 10; ++t) acc[c][t] = 0.f;
  const int v = threadIdx.x & 3;
  const size_t P = (size_t)N * Ho * Wo;
  // two waves per SIMD (512 workgroups: the partial-sum workspace) cannot cover a load -> 80 FMAs chain by themselves: the next
  // pixel's ten loads are requested before this pixel's arithmetic (same sums in the same order; 71 -> see DESIGN 3.3)
  // pixel index -> (n, oy, ox): the three 64-bit divisions of the straightforward form were more instructions than the 80 FMAs they
  // feed; below 2^24 pixels (N <= 260 at 254 x 254) two reciprocal multiplies are exact (x < 2^24, divisor < 2^16)
  const bool small = P < ((size_t)1 << 24) && Wo < 65536 && Ho < 65536;
  const unsigned mw = (unsigned)(((unsigned long long)1 << 32) / (unsigned)Wo) + 1u, mh = (unsigned)(((unsigned long long)1 << 32) / (unsigned)Ho) + 1u;
  auto load_px = [&](size_t p, float (&g)[8], float (&in)[9]) __attribute__((always_inline)) {
    int ox, oy, n;
    if (small) {
      const unsigned pp = (unsigned)p, row = __umulhi(pp, mw);
      ox = (int)(pp - row * (unsigned)Wo);
      n = (int)__umulhi(row, mh);
      oy = (int)(row - (unsigned)n * (unsigned)Ho);
    } else {
      ox = (int)(p % Wo); oy = (int)((p / Wo) % Ho); n = (int)(p / ((size_t)Wo * Ho));
    }
    ld8(G + p * 32 + v * 8, g);
#pragma unroll
    for (int t = 0; t < 9; ++t) in[t] = x[((size_t)n * H + oy + t / 3) * W + ox + t % 3];
  };
  const size_t stride = (size_t)gridDim.x * 64;
  size_t p = (size_t)blockIdx.x * 64 + (threadIdx.x >> 2);
  float gn[8], inn[9];
  if (p < P) load_px(p, gn, inn);
  while (p < P) {
    float g[8], in[9];
#pragma unroll
    for (int c = 0; c < 8; ++c) g[c] = gn[c];
#pragma unroll
    for (int t = 0; t < 9; ++t) in[t] = inn[t];
    p += stride;
    if (p < P) load_px(p, gn, inn);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[c][t] = fmaf(g[c], in[t], acc[c][t]);
      acc[c][9] += g[c];
    }
  }
  // lanes with equal v hold the same channels: reduce over the 16 pixel-lanes of a wave sharing v (stride 4)
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int t = 0; t < 10; ++t) {
      float s = acc[c][t];
#pragma unroll
      for (int o = 32; o >= 4; o >>= 1) s += __shfl_xor(s, o, 64);
      if ((threadIdx.x & 63) < 4) red[threadIdx.x >> 6][(v * 8 + c) * 10 + t] = s;
    }
  __syncthreads();
  for (int i = threadIdx.x; i < 320; i += 256)
    partial[(size_t)blockIdx.x * 320 + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

__global__ void conv_in_final_kernel(const float* __restrict__ tot, float* __restrict__ gw, float* __restrict__ gb, int accumulate) {
  const int i = threadIdx.x + blockIdx.x * blockDim.x;
  if (i >= 320) return;
  const int co = i / 10, t = i % 10;
  float* dst = t < 9 ? gw + co * 9 + t : gb + co;
  *dst = accumulate ? *dst + tot[i] : tot[i];
}

inline int nblocks(size_t n, int cap = 4096) { return (int)((n + 255) / 256 < (size_t)cap ? (n + 255) / 256 : cap); }

}  // namespace

#include "bwd_internal.h"

template <typename T>
static int outc_backward_t(const float* g_out, const float* x_out, const void* g_upx, const void* up_x, const float* w,
                                  void* G_up, float* gw, float* gb, long long P, int last_act, float slope, int accumulate,
                                  void* workspace /* 1024*33 floats */, void* stream) {
  if (!g_out || !x_out || !up_x || !w || !G_up || !gw || !gb || !workspace || P <= 0) return UNCL_ERR_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int blocks = nblocks((size_t)P * 4, 1024);
  hipLaunchKernelGGL(outc_bwd_kernel<T>, dim3(blocks), dim3(256), 0, s, g_out, x_out, (const T*)g_upx, (const T*)up_x, w,
                     (T*)G_up, (float*)workspace, (size_t)P, last_act, slope);
  // the 33 sums go straight to gw[32], gb[1] (round 4: one launch instead of three; same fp64 sums, rounded to fp32 once as before)
  hipLaunchKernelGGL(reduce_rows_kernel<4>, dim3(9), dim3(256), 0, s, (const float*)workspace, blocks, 33, gw, accumulate, gb, 32);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}
int bwd_outc_backward(int dtype, const float* g_out, const float* x_out, const void* g_upx, const void* up_x, const float* w,
                                  void* G_up, float* gw, float* gb, long long P, int last_act, float slope, int accumulate,
                                  void* workspace /* 1024*33 floats */, void* stream) {
  return dtype == UNCL_F32 ? outc_backward_t<float>(g_out, x_out, g_upx, up_x, w, G_up, gw, gb, P, last_act, slope, accumulate, workspace, stream) : outc_backward_t<bf16_t>(g_out, x_out, g_upx, up_x, w, G_up, gw, gb, P, last_act, slope, accumulate, workspace, stream);
}
extern "C" int uncl_outc_backward(const float* g_out, const float* x_out, const void* g_upx, const void* up_x, const float* w,
                                  void* G_up, float* gw, float* gb, long long P, int last_act, float slope, int accumulate,
                                  void* workspace /* 1024*33 floats */, void* stream) {
  return bwd_outc_backward(UNCL_BF16, g_out, x_out, g_upx, up_x, w, G_up, gw, gb, P, last_act, slope, accumulate, workspace, stream);
}

template <typename T>
static int ssr_backward_t(const void* g_cat, const void* x2, void* G_x2, void* G_x1, int N, int H, int W, int C, int H1,
                                 int W1, float slope, int accumulate_x2, void* stream) {
  if (!g_cat || !x2 || !G_x2 || !G_x1 || C % 8 != 0) return UNCL_ERR_ARG;
  hipLaunchKernelGGL(ssr_bwd_kernel<T>, dim3(nblocks((size_t)N * H * W * (C / 8))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     (const T*)g_cat, (const T*)x2, (T*)G_x2, (T*)G_x1, N, H, W, C, H1, W1, slope, accumulate_x2);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}
int bwd_ssr_backward(int dtype, const void* g_cat, const void* x2, void* G_x2, void* G_x1, int N, int H, int W, int C, int H1,
                                 int W1, float slope, int accumulate_x2, void* stream) {
  return dtype == UNCL_F32 ? ssr_backward_t<float>(g_cat, x2, G_x2, G_x1, N, H, W, C, H1, W1, slope, accumulate_x2, stream) : ssr_backward_t<bf16_t>(g_cat, x2, G_x2, G_x1, N, H, W, C, H1, W1, slope, accumulate_x2, stream);
}
extern "C" int uncl_ssr_backward(const void* g_cat, const void* x2, void* G_x2, void* G_x1, int N, int H, int W, int C, int H1,
                                 int W1, float slope, int accumulate_x2, void* stream) {
  return bwd_ssr_backward(UNCL_BF16, g_cat, x2, G_x2, G_x1, N, H, W, C, H1, W1, slope, accumulate_x2, stream);
}

template <typename T>
static int pool_backward_t(const void* g_pool, const void* x, void* G_x, int N, int H, int W, int C, float slope,
                                  int accumulate, void* stream, const void* cin = nullptr, void* cout = nullptr, int pc = 0) {
  if (!g_pool || !x || !G_x || C % 8 != 0 || pc < 0 || pc > 8) return UNCL_ERR_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (!accumulate && ((H & 1) || (W & 1))) {
    // odd sizes: the last row / column is outside every window and must read as zero
    if (hipMemsetAsync(G_x, 0, (size_t)N * H * W * C * sizeof(T), s) != hipSuccess) return UNCL_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(pool_bwd_kernel<T>, dim3(nblocks((size_t)N * (H / 2) * (W / 2) * (C / 8))), dim3(256), 0, s, (const T*)g_pool,
                     (const T*)x, (T*)G_x, N, H, W, C, slope, accumulate, (const T*)cin, (T*)cout, pc);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}
int bwd_pool_backward(int dtype, const void* g_pool, const void* x, void* G_x, int N, int H, int W, int C, float slope,
                                  int accumulate, void* stream) {
  return dtype == UNCL_F32 ? pool_backward_t<float>(g_pool, x, G_x, N, H, W, C, slope, accumulate, stream) : pool_backward_t<bf16_t>(g_pool, x, G_x, N, H, W, C, slope, accumulate, stream);
}
// the same with the head hand-off of a clip (bwd_head_handoff without a mask) applied to g_pool's first pc channels as they are read;
// carries: (N * H/2 * W/2, pc) in the gradient's type, either may be NULL
int bwd_pool_backward_handoff(int dtype, const void* g_pool, const void* x, void* G_x, int N, int H, int W, int C, float slope,
                              int accumulate, const void* carry_in, void* carry_out, int pc, void* stream) {
  return dtype == UNCL_F32 ? pool_backward_t<float>(g_pool, x, G_x, N, H, W, C, slope, accumulate, stream, carry_in, carry_out, pc)
                           : pool_backward_t<bf16_t>(g_pool, x, G_x, N, H, W, C, slope, accumulate, stream, carry_in, carry_out, pc);
}
extern "C" int uncl_pool_backward(const void* g_pool, const void* x, void* G_x, int N, int H, int W, int C, float slope,
                                  int accumulate, void* stream) {
  return bwd_pool_backward(UNCL_BF16, g_pool, x, G_x, N, H, W, C, slope, accumulate, stream);
}

template <typename T>
static int gelu_forward_t(const void* z, void* h, long long n, void* stream) {
  if (!z || !h || n <= 0 || n % 8 != 0) return UNCL_ERR_ARG;
  hipLaunchKernelGGL(gelu_fwd_kernel<T>, dim3(nblocks((size_t)n / 8)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     (const T*)z, (T*)h, (size_t)n / 8);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}
int bwd_gelu_forward(int dtype, const void* z, void* h, long long n, void* stream) {
  return dtype == UNCL_F32 ? gelu_forward_t<float>(z, h, n, stream) : gelu_forward_t<bf16_t>(z, h, n, stream);
}
extern "C" int uncl_gelu_forward(const void* z, void* h, long long n, void* stream) {
  return bwd_gelu_forward(UNCL_BF16, z, h, n, stream);
}
template <typename T>
static int gelu_backward_t(const void* g_h, const void* z, void* g_z, long long n, void* stream) {
  if (!g_h || !z || !g_z || n <= 0 || n % 8 != 0) return UNCL_ERR_ARG;
  hipLaunchKernelGGL(gelu_bwd_kernel<T>, dim3(nblocks((size_t)n / 8)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     (const T*)g_h, (const T*)z, (T*)g_z, (size_t)n / 8);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}
int bwd_gelu_backward(int dtype, const void* g_h, const void* z, void* g_z, long long n, void* stream) {
  return dtype == UNCL_F32 ? gelu_backward_t<float>(g_h, z, g_z, n, stream) : gelu_backward_t<bf16_t>(g_h, z, g_z, n, stream);
}
extern "C" int uncl_gelu_backward(const void* g_h, const void* z, void* g_z, long long n, void* stream) {
  return bwd_gelu_backward(UNCL_BF16, g_h, z, g_z, n, stream);
}
template <typename T>
static int scale_rows_t(const void* x, const float* scale, void* y, int N, long long per, void* stream) {
  if (!x || !y || N <= 0 || per <= 0 || per % 8 != 0) return UNCL_ERR_ARG;
  hipLaunchKernelGGL(scale_rows_kernel<T>, dim3(nblocks((size_t)N * per / 8)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     (const T*)x, scale, (T*)y, (size_t)per / 8, (size_t)N * per / 8);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}
int bwd_scale_rows(int dtype, const void* x, const float* scale, void* y, int N, long long per, void* stream) {
  return dtype == UNCL_F32 ? scale_rows_t<float>(x, scale, y, N, per, stream) : scale_rows_t<bf16_t>(x, scale, y, N, per, stream);
}
extern "C" int uncl_scale_rows(const void* x, const float* scale, void* y, int N, long long per, void* stream) {
  return bwd_scale_rows(UNCL_BF16, x, scale, y, N, per, stream);
}
template <typename T>
static int mask_minus_t(const void* g, const void* x, const void* pe, void* out, int N, long long per, float slope,
                               void* stream) {
  if (!g || !x || !pe || !out || per % 8 != 0) return UNCL_ERR_ARG;
  hipLaunchKernelGGL(mask_minus_kernel<T>, dim3(nblocks((size_t)N * per / 8)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     (const T*)g, (const T*)x, (const T*)pe, (T*)out, (size_t)per / 8, (size_t)N * per / 8, slope);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}
int bwd_mask_minus(int dtype, const void* g, const void* x, const void* pe, void* out, int N, long long per, float slope,
                               void* stream) {
  return dtype == UNCL_F32 ? mask_minus_t<float>(g, x, pe, out, N, per, slope, stream) : mask_minus_t<bf16_t>(g, x, pe, out, N, per, slope, stream);
}
extern "C" int uncl_mask_minus(const void* g, const void* x, const void* pe, void* out, int N, long long per, float slope,
                               void* stream) {
  return bwd_mask_minus(UNCL_BF16, g, x, pe, out, N, per, slope, stream);
}
template <typename T>
static int sum_samples_t(const void* g, float* out, int N, long long per, int accumulate, void* stream) {
  if (!g || !out || N <= 0 || per <= 0) return UNCL_ERR_ARG;
  hipLaunchKernelGGL(sum_samples_kernel<T>, dim3(nblocks((size_t)per)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     (const T*)g, out, N, (size_t)per, accumulate);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}
int bwd_sum_samples(int dtype, const void* g, float* out, int N, long long per, int accumulate, void* stream) {
  return dtype == UNCL_F32 ? sum_samples_t<float>(g, out, N, per, accumulate, stream) : sum_samples_t<bf16_t>(g, out, N, per, accumulate, stream);
}
extern "C" int uncl_sum_samples(const void* g, float* out, int N, long long per, int accumulate, void* stream) {
  return bwd_sum_samples(UNCL_BF16, g, out, N, per, accumulate, stream);
}
template <typename T>
static int head_handoff_t(void* g, const void* mask, float slope, const void* carry_in, void* carry_out, long long npix,
                                 int C, int prev_ch, void* stream) {
  if (!g || npix <= 0 || C % 8 != 0 || prev_ch <= 0 || prev_ch > 8 || prev_ch > C) return UNCL_ERR_ARG;
  if (!mask && !carry_in && !carry_out) return UNCL_OK;
  const size_t total = (size_t)npix * (mask ? C / 8 : 1);
  hipLaunchKernelGGL(head_handoff_kernel<T>, dim3(nblocks(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), (T*)g,
                     (const T*)mask, slope, (const T*)carry_in, (T*)carry_out, (size_t)npix, C, prev_ch);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}
int bwd_head_handoff(int dtype, void* g, const void* mask, float slope, const void* carry_in, void* carry_out, long long npix,
                                 int C, int prev_ch, void* stream) {
  return dtype == UNCL_F32 ? head_handoff_t<float>(g, mask, slope, carry_in, carry_out, npix, C, prev_ch, stream) : head_handoff_t<bf16_t>(g, mask, slope, carry_in, carry_out, npix, C, prev_ch, stream);
}
extern "C" int uncl_head_handoff(void* g, const void* mask, float slope, const void* carry_in, void* carry_out, long long npix,
                                 int C, int prev_ch, void* stream) {
  return bwd_head_handoff(UNCL_BF16, g, mask, slope, carry_in, carry_out, npix, C, prev_ch, stream);
}
template <typename T>
static int mix_heads_t(const void* x, const void* prev, void* out, long long npix, int C, int prev_ch, void* stream, long long own = 0,
                       long long lag = 0) {
  if (!x || !prev || !out || npix <= 0 || C % 8 != 0 || prev_ch <= 0 || prev_ch > 8 || own < 0 || lag < 0 || lag > own) return UNCL_ERR_ARG;
  hipLaunchKernelGGL(mix_heads_kernel<T>, dim3(nblocks((size_t)npix * (C / 8))), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     (const T*)x, (const T*)prev, (T*)out, (size_t)npix, C, prev_ch, (size_t)own, (size_t)lag);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}
int bwd_mix_heads(int dtype, const void* x, const void* prev, void* out, long long npix, int C, int prev_ch, void* stream) {
  return dtype == UNCL_F32 ? mix_heads_t<float>(x, prev, out, npix, C, prev_ch, stream) : mix_heads_t<bf16_t>(x, prev, out, npix, C, prev_ch, stream);
}
// a clip's frames one behind the other in x (npix pixels in all, frame_pix per frame): out = x with every frame's head channels taken
// from the frame before it, frame 0 keeping its own -- one launch instead of a copy of frame 0 and a mix of the rest
int bwd_mix_heads_clip(int dtype, const void* x, void* out, long long npix, long long frame_pix, int C, int prev_ch, void* stream) {
  return dtype == UNCL_F32 ? mix_heads_t<float>(x, x, out, npix, C, prev_ch, stream, frame_pix, frame_pix)
                           : mix_heads_t<bf16_t>(x, x, out, npix, C, prev_ch, stream, frame_pix, frame_pix);
}
extern "C" int uncl_mix_heads(const void* x, const void* prev, void* out, long long npix, int C, int prev_ch, void* stream) {
  return bwd_mix_heads(UNCL_BF16, x, prev, out, npix, C, prev_ch, stream);
}
// g_x_f32: fp32 scratch of N n C floats for the global-atomic form (zeroed here); g_x_bf16 receives the result
template <typename T>
static int gcn_maxrel_backward_t(const void* g_out, const void* x, const int32_t* idx, float* g_x_f32, void* g_x_bf16, int N,
                                        int n, int C, int k, void* stream) {
  if (!g_out || !x || !idx || !g_x_f32 || !g_x_bf16 || C % 8 != 0) return UNCL_ERR_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // gather form (one owner per element, fixed order): the fp32 parity mode, and bf16 passes in deterministic mode
  if ((sizeof(T) == 4 || uncl_wgrad_deterministic()) && C % 64 == 0 && (size_t)n * 64 * 4 <= 64 * 1024) {
    hipLaunchKernelGGL(maxrel_bwd_det_kernel<T>, dim3(N, C / 64), dim3(64), (size_t)n * 64 * 4, s, (const T*)g_out, (const T*)x, idx,
                       (T*)g_x_bf16, n, C, k);
    UNCL_CHECK_LAUNCH();
    return UNCL_OK;
  }
  // 16-bit passes: per-(sample, channel slice) workgroups accumulating in LDS (UNCL_MAXREL_BWD_LDS=0: the global-atomic form)
  static const int lds_on = [] { const char* e = getenv("UNCL_MAXREL_BWD_LDS"); return e ? atoi(e) : 1; }();
  if (lds_on && sizeof(T) == 2 && C % MRB_CH == 0 && maxrel_bwd_lds_bytes(n, k, sizeof(T)) <= 64 * 1024) {
    hipLaunchKernelGGL(maxrel_bwd_lds_kernel<T>, dim3(N, C / MRB_CH), dim3(256), maxrel_bwd_lds_bytes(n, k, sizeof(T)), s,
                       (const T*)g_out, (const T*)x, idx, (T*)g_x_bf16, n, C, k);
    UNCL_CHECK_LAUNCH();
    return UNCL_OK;
  }
  if (hipMemsetAsync(g_x_f32, 0, (size_t)N * n * C * 4, s) != hipSuccess) return UNCL_ERR_LAUNCH;
  hipLaunchKernelGGL(maxrel_bwd_kernel<T>, dim3(nblocks((size_t)N * n * (C / 8))), dim3(256), 0, s, (const T*)g_out, (const T*)x,
                     idx, g_x_f32, N, n, C, k);
  hipLaunchKernelGGL(f32_to_bf16_kernel<T>, dim3(nblocks((size_t)N * n * C)), dim3(256), 0, s, g_x_f32, (T*)g_x_bf16, (size_t)N * n * C);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}
int bwd_gcn_maxrel_backward(int dtype, const void* g_out, const void* x, const int32_t* idx, float* g_x_f32, void* g_x_bf16, int N,
                                        int n, int C, int k, void* stream) {
  return dtype == UNCL_F32 ? gcn_maxrel_backward_t<float>(g_out, x, idx, g_x_f32, g_x_bf16, N, n, C, k, stream) : gcn_maxrel_backward_t<bf16_t>(g_out, x, idx, g_x_f32, g_x_bf16, N, n, C, k, stream);
}
extern "C" int uncl_gcn_maxrel_backward(const void* g_out, const void* x, const int32_t* idx, float* g_x_f32, void* g_x_bf16, int N,
                                        int n, int C, int k, void* stream) {
  return bwd_gcn_maxrel_backward(UNCL_BF16, g_out, x, idx, g_x_f32, g_x_bf16, N, n, C, k, stream);
}
// G: (N,H-2,W-2,32) bf16 masked gradient of inc.conv.conv's output; gw: (32,1,3,3), gb: (32).  workspace: 512*320+320 floats
template <typename T>
static int conv_in_c1_wgrad_t(const void* G, const float* x, float* gw, float* gb, int N, int H, int W, int accumulate,
                                     void* workspace, void* stream) {
  if (!G || !x || !gw || !gb || !workspace) return UNCL_ERR_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t P = (size_t)N * (H - 2) * (W - 2);
  const int blocks = (int)((P + 63) / 64 < 512 ? (P + 63) / 64 : 512);
  float* part = (float*)workspace;
  hipLaunchKernelGGL(conv_in_wgrad_kernel<T>, dim3(blocks), dim3(256), 0, s, (const T*)G, x, part, N, H, W);
  float* tot = part + (size_t)512 * 320;
  hipLaunchKernelGGL(reduce_rows_kernel<4>, dim3(80), dim3(256), 0, s, (const float*)part, blocks, 320, tot, 0, nullptr, 0);
  hipLaunchKernelGGL(conv_in_final_kernel, dim3(5), dim3(64), 0, s, (const float*)tot, gw, gb, accumulate);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}
int bwd_conv_in_c1_wgrad(int dtype, const void* G, const float* x, float* gw, float* gb, int N, int H, int W, int accumulate,
                                     void* workspace, void* stream) {
  return dtype == UNCL_F32 ? conv_in_c1_wgrad_t<float>(G, x, gw, gb, N, H, W, accumulate, workspace, stream) : conv_in_c1_wgrad_t<bf16_t>(G, x, gw, gb, N, H, W, accumulate, workspace, stream);
}
extern "C" int uncl_conv_in_c1_wgrad(const void* G, const float* x, float* gw, float* gb, int N, int H, int W, int accumulate,
                                     void* workspace, void* stream) {
  return bwd_conv_in_c1_wgrad(UNCL_BF16, G, x, gw, gb, N, H, W, accumulate, workspace, stream);
}
