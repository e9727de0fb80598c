// Window-statistics reductions (HBM-bound, fp32 accumulation, deterministic two-stage sums).
//
// uncl_gauss_stats: per (sample, channel): mean(x) and mean of the 11x11 sigma-1.5 Gaussian local variance
//   var = G*(x^2) - (G*x)^2 over the 'valid' region.  Replaces ContrastExtracter / compute_contrast followed by
//   adaptive_avg_pool2d / .mean(dim=[-1,-2]) (Unet.py:101-123,274-278; Discriminator.py:50-83,122-124;
//   GanTrainerImg.py:24-56,308-313,361-367).
#include "common.h"

namespace {

constexpr int GW = 11;

struct GaussW {
  float g[GW];
};

// CPT channels per thread (8 for NHWC bf16/fp32 with C % 8 == 0, 1 for single-channel fp32 images)
template <typename T, int CPT>
__global__ __launch_bounds__(256) void gauss_stats_kernel(const T* __restrict__ x, float* __restrict__ partial, int H,
                                                          int W, int C, int band_rows, int n_bands, GaussW gw) {
  // grid: (n_bands, C / CPT, N); thread = column
  extern __shared__ float sv[];  // [2][CPT][256] vertical-pass results of the current output row
  const int col = threadIdx.x;
  const int band = blockIdx.x, cg = blockIdx.y, n = blockIdx.z;
  const int Ho = H - (GW - 1), Wo = W - (GW - 1);
  const int y0 = band * band_rows;
  const int y1 = min(y0 + band_rows, Ho);  // output rows [y0, y1)
  const bool incol = col < W;
  const T* base = x + ((size_t)n * H * W) * C + (size_t)cg * CPT;

  float ring[GW][CPT];  // input rows y .. y+10 of this column
  float s_x[CPT], s_var[CPT];
#pragma unroll
  for (int c = 0; c < CPT; ++c) s_x[c] = 0.f, s_var[c] = 0.f;

  auto load_row = [&](int r, float* dst) {
    if (incol && r < H) {
      const T* p = base + ((size_t)r * W + col) * C;
      if constexpr (CPT == 8 && sizeof(T) == 2) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
        for (int c = 0; c < 8; ++c) dst[c] = (float)v[c];
      } else {
#pragma unroll
        for (int c = 0; c < CPT; ++c) dst[c] = (float)p[c];
      }
    } else {
#pragma unroll
      for (int c = 0; c < CPT; ++c) dst[c] = 0.f;
    }
  };
  // rows whose plain sum this band owns: [y0, y1) plus, for the last band, the trailing GW-1 rows
  const int own_hi = (band == n_bands - 1) ? H : y1;

#pragma unroll
  for (int k = 0; k < GW - 1; ++k) {
    load_row(y0 + k, ring[k]);
  }
  for (int y = y0; y < y1; ++y) {
    // slide: newest row enters at slot (y + 10) % 11 -> keep the ring in a fixed order by shifting (registers)
    float fresh[CPT];
    load_row(y + GW - 1, fresh);
    float v1[CPT], v2[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      v1[c] = gw.g[GW - 1] * fresh[c];
      v2[c] = gw.g[GW - 1] * fresh[c] * fresh[c];
    }
#pragma unroll
    for (int k = 0; k < GW - 1; ++k)
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        v1[c] = fmaf(gw.g[k], ring[k][c], v1[c]);
        v2[c] = fmaf(gw.g[k] * ring[k][c], ring[k][c], v2[c]);
      }
    // plain sum of the oldest row (row y) if this band owns it; trailing rows are added after the loop
#pragma unroll
    for (int c = 0; c < CPT; ++c) s_x[c] += ring[0][c];
#pragma unroll
    for (int k = 0; k < GW - 2; ++k)
#pragma unroll
      for (int c = 0; c < CPT; ++c) ring[k][c] = ring[k + 1][c];
#pragma unroll
    for (int c = 0; c < CPT; ++c) ring[GW - 2][c] = fresh[c];

    __syncthreads();  // previous row's readers are done
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      sv[c * 256 + col] = v1[c];
      sv[256 * CPT + c * 256 + col] = v2[c];
    }
    __syncthreads();
    if (col < Wo) {
#pragma unroll
      for (int c = 0; c < CPT; ++c) {
        float mu = 0.f, e2 = 0.f;
#pragma unroll
        for (int t = 0; t < GW; ++t) {
          mu = fmaf(gw.g[t], sv[c * 256 + col + t], mu);
          e2 = fmaf(gw.g[t], sv[256 * CPT + c * 256 + col + t], e2);
        }
        s_var[c] += e2 - mu * mu;
      }
    }
  }
  // the ring now holds rows y1 .. y1+9; the last band owns them for the plain sum
  if (own_hi > y1) {
#pragma unroll
    for (int k = 0; k < GW - 1; ++k)
      if (y1 + k < own_hi) {
#pragma unroll
        for (int c = 0; c < CPT; ++c) s_x[c] += ring[k][c];
      }
  }
  // deterministic workgroup reduction over columns: waves, then 4 partials in LDS
  __syncthreads();
  float* red = sv;  // [4 waves][2][CPT]
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const float a = wave_sum(s_x[c]), b = wave_sum(s_var[c]);
    if ((threadIdx.x & 63) == 0) {
      red[(threadIdx.x >> 6) * 2 * CPT + c] = a;
      red[(threadIdx.x >> 6) * 2 * CPT + CPT + c] = b;
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 * CPT) {
    const float t = (red[threadIdx.x] + red[2 * CPT + threadIdx.x]) + (red[4 * CPT + threadIdx.x] + red[6 * CPT + threadIdx.x]);
    // partial[n][band][2][C]
    const int which = threadIdx.x / CPT, c = threadIdx.x % CPT;
    partial[(((size_t)n * n_bands + band) * 2 + which) * C + cg * CPT + c] = t;
  }
}

// The 16-bit NHWC form (the generator's 32-channel feature map, ten launches per video step at 8 samples each: 65 us per launch
// in the generic kernel above, which waits for each row's load, crosses two barriers per row and convolves x^2 as well).  Here:
//  * mean over the valid windows of G*(x^2) is a POINTWISE sum -- sum_p x_p^2 wy(p.y) wx(p.x) with w = the Gaussian mass of the
//    valid windows that cover the pixel (1 in the interior) -- so only G*x is convolved;
//  * the row loop is unrolled over the eleven ring positions (register ring with compile-time slots, no shifting);
//  * the next row is requested one iteration ahead, and the vertical results go through two LDS buffers: one barrier per row.
// Per column the thread keeps sum x^2 w and sum mu^2 apart (fp32, at most band_rows terms each) and subtracts at the end.
__device__ __forceinline__ float window_mass(const GaussW& gw, int p, int n_out) {
  if (p >= GW - 1 && p < n_out) return 1.f;
  float w = 0.f;
#pragma unroll
  for (int t = 0; t < GW; ++t)
    if (p - t >= 0 && p - t < n_out) w += gw.g[t];
  return w;
}

__global__ __launch_bounds__(256) void gauss_stats_h16_kernel(const bf16_t* __restrict__ x, float* __restrict__ partial, int H,
                                                              int W, int C, int band_rows, int n_bands, GaussW gw) {
  constexpr int CPT = 8;
  extern __shared__ float sv[];  // [2][CPT][256] vertical-pass results of two consecutive output rows
  const int col = threadIdx.x;
  const int band = blockIdx.x, cg = blockIdx.y, n = blockIdx.z;
  const int Ho = H - (GW - 1), Wo = W - (GW - 1);
  const int y0 = band * band_rows;
  const int y1 = min(y0 + band_rows, Ho);  // output rows [y0, y1)
  const bool incol = col < W;
  const bf16_t* base = x + ((size_t)n * H * W) * C + (size_t)cg * CPT;
  const float wx = incol ? window_mass(gw, col, Wo) : 0.f;

  float ring[GW][CPT];
  float s_x[CPT], s_e2[CPT], s_mu2[CPT];
#pragma unroll
  for (int c = 0; c < CPT; ++c) s_x[c] = 0.f, s_e2[c] = 0.f, s_mu2[c] = 0.f;
  auto load_raw = [&](int r) {
    bf16x8 v;
#pragma unroll
    for (int c = 0; c < CPT; ++c) v[c] = (bf16_t)0.f;
    if (incol && r < H) v = *reinterpret_cast<const bf16x8*>(base + ((size_t)r * W + col) * C);
    return v;
  };
  // rows whose plain sums this band owns: [y0, y1) plus, for the last band, the trailing GW-1 rows
  const int own_hi = (band == n_bands - 1) ? H : y1;
#pragma unroll
  for (int k = 0; k < GW - 1; ++k) {
    const bf16x8 v = load_raw(y0 + k);
#pragma unroll
    for (int c = 0; c < CPT; ++c) ring[k][c] = (float)v[c];
  }
  bf16x8 nxt = load_raw(y0 + GW - 1);
  for (int yb = y0; yb < y1; yb += GW) {
#pragma unroll
    for (int j = 0; j < GW; ++j) {
      const int y = yb + j;
      if (y < y1) {
        // slot (j + k) % GW holds input row y + k; the row requested last iteration completes the window
#pragma unroll
        for (int c = 0; c < CPT; ++c) ring[(j + GW - 1) % GW][c] = (float)nxt[c];
        nxt = load_raw(y + GW);
        float v1[CPT];
#pragma unroll
        for (int c = 0; c < CPT; ++c) v1[c] = gw.g[0] * ring[j % GW][c];
#pragma unroll
        for (int k = 1; k < GW; ++k)
#pragma unroll
          for (int c = 0; c < CPT; ++c) v1[c] = fmaf(gw.g[k], ring[(j + k) % GW][c], v1[c]);
        const float w = window_mass(gw, y, Ho) * wx;
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
          const float xv = ring[j % GW][c];
          s_x[c] += xv;
          s_e2[c] = fmaf(xv * xv, w, s_e2[c]);
        }
        float* buf = sv + ((y - y0) & 1) * CPT * 256;
#pragma unroll
        for (int c = 0; c < CPT; ++c) buf[c * 256 + col] = v1[c];
        __syncthreads();
        if (col < Wo) {
#pragma unroll
          for (int c = 0; c < CPT; ++c) {
            float mu = 0.f;
#pragma unroll
            for (int t = 0; t < GW; ++t) mu = fmaf(gw.g[t], buf[c * 256 + col + t], mu);
            s_mu2[c] = fmaf(mu, mu, s_mu2[c]);
          }
        }
      }
    }
  }
  // the last band owns the trailing rows y1 .. H-1 (their ring slots depend on the band's length: read them again)
  for (int r = y1; r < own_hi; ++r) {
    const bf16x8 v = load_raw(r);
    const float w = window_mass(gw, r, Ho) * wx;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const float xv = (float)v[c];
      s_x[c] += xv;
      s_e2[c] = fmaf(xv * xv, w, s_e2[c]);
    }
  }
  __syncthreads();
  float* red = sv;  // [4 waves][2][CPT]
#pragma unroll
  for (int c = 0; c < CPT; ++c) {
    const float a = wave_sum(s_x[c]), b = wave_sum(s_e2[c] - s_mu2[c]);
    if ((threadIdx.x & 63) == 0) {
      red[(threadIdx.x >> 6) * 2 * CPT + c] = a;
      red[(threadIdx.x >> 6) * 2 * CPT + CPT + c] = b;
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 * CPT) {
    const float t = (red[threadIdx.x] + red[2 * CPT + threadIdx.x]) + (red[4 * CPT + threadIdx.x] + red[6 * CPT + threadIdx.x]);
    const int which = threadIdx.x / CPT, c = threadIdx.x % CPT;
    partial[(((size_t)n * n_bands + band) * 2 + which) * C + cg * CPT + c] = t;
  }
}

__global__ void gauss_stats_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int n_bands, int C,
                                         double inv_hw, double inv_howo, int total) {
  // out[n][2][C]
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = i % C, which = (i / C) % 2, n = i / (2 * C);
  double s = 0.0;
  for (int b = 0; b < n_bands; ++b) s += (double)partial[(((size_t)n * n_bands + b) * 2 + which) * C + c];
  out[i] = (float)(s * (which == 0 ? inv_hw : inv_howo));
}

}  // namespace

extern "C" size_t uncl_gauss_stats_workspace_bytes(int N, int H, int C) {
  const int band_rows = 8;        // the thinnest band uncl_gauss_stats may choose
  const int n_bands = (H - 10 + band_rows - 1) / band_rows;
  return (size_t)N * (n_bands > 0 ? n_bands : 1) * 2 * C * sizeof(float);
}

// x: NHWC (N,H,W,C) in dtype (C == 1: a plain (N,H,W) image).  out: fp32 (N,2,C): [mean(x), mean(local variance)].
extern "C" int uncl_gauss_stats(const void* x, int dtype, float* out, int N, int H, int W, int C, void* workspace,
                                void* stream) {
  if (!x || !out || !workspace || N <= 0 || H < GW || W < GW || W > 256) return UNCL_ERR_ARG;
  if (!(C == 1 || C % 8 == 0)) return UNCL_ERR_ARG;
  if (C == 1 && dtype != UNCL_F32) return UNCL_ERR_ARG;
  GaussW gw;
  double s = 0.0, g[GW];
  for (int k = 0; k < GW; ++k) { g[k] = exp(-((k - 5) * (k - 5)) / (2.0 * 1.5 * 1.5)); s += g[k]; }
  for (int k = 0; k < GW; ++k) gw.g[k] = (float)(g[k] / s);
  const int Ho = H - (GW - 1), Wo = W - (GW - 1);
  // bands of output rows per workgroup: 32 rows re-read 31 % of halo rows; small batches (the per-frame calls of the video
  // generator: 8 samples) take thinner bands so that the launch has more workgroups than CUs
  int band_rows = 32;
  // (the 16-bit kernel, measured at 8 / 32 samples of 256 x 256 x 32: 8-row bands 51 / 159 us, 16-row 33 / 99, 32-row 43 / 99)
  static const int wg_env = [] { const char* e = getenv("UNCL_GAUSS_WG_MIN"); return e ? atoi(e) : 0; }();
  const int wg_min = wg_env > 0 ? wg_env : (dtype == UNCL_BF16 && C != 1 ? 300 : 768);
  while (band_rows > 8 && (long long)((Ho + band_rows - 1) / band_rows) * (C >= 8 ? C / 8 : 1) * N < wg_min) band_rows /= 2;
  const int n_bands = (Ho + band_rows - 1) / band_rows;
  float* partial = reinterpret_cast<float*>(workspace);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (C == 1) {
    hipLaunchKernelGGL((gauss_stats_kernel<float, 1>), dim3(n_bands, 1, N), dim3(256), 2 * 256 * 1 * sizeof(float), st,
                       (const float*)x, partial, H, W, C, band_rows, n_bands, gw);
  } else if (dtype == UNCL_BF16) {
    static const int h16_on = [] { const char* e = getenv("UNCL_GAUSS_H16"); return e ? atoi(e) : 1; }();   // 0: the generic kernel (A/B)
    if (h16_on)
      hipLaunchKernelGGL(gauss_stats_h16_kernel, dim3(n_bands, C / 8, N), dim3(256), 2 * 256 * 8 * sizeof(float), st,
                         (const bf16_t*)x, partial, H, W, C, band_rows, n_bands, gw);
    else
      hipLaunchKernelGGL((gauss_stats_kernel<bf16_t, 8>), dim3(n_bands, C / 8, N), dim3(256), 2 * 256 * 8 * sizeof(float), st,
                         (const bf16_t*)x, partial, H, W, C, band_rows, n_bands, gw);
  } else if (dtype == UNCL_F32) {
    hipLaunchKernelGGL((gauss_stats_kernel<float, 8>), dim3(n_bands, C / 8, N), dim3(256), 2 * 256 * 8 * sizeof(float), st,
                       (const float*)x, partial, H, W, C, band_rows, n_bands, gw);
  } else {
    return UNCL_ERR_ARG;
  }
  UNCL_CHECK_LAUNCH();
  const int total = N * 2 * C;
  hipLaunchKernelGGL(gauss_stats_final_kernel, dim3((total + 255) / 256), dim3(256), 0, st, partial, out, n_bands, C,
                     1.0 / ((double)H * W), 1.0 / ((double)Ho * Wo), total);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}
