// GPU-side data path of the trainers: the per-sample arithmetic of `npy_loader` (utils/ProcessedDatasetFolderImg.py:43-206,
// utils/ProcessedDatasetFolder.py:43-236) -- random resize, random 256 x 256 crop, RGB -> Y, LDR normalisation / HDR log
// compression -- on device tensors.  The random choices (mode, size, crop corner) stay on the host, drawn from numpy's
// generator in the reference's order; everything that touches pixels runs here.  HBM-bound element-wise / gather kernels.
//
// cv2 is neither in the reference tree nor in this image: `cv2.resize` (INTER_LINEAR) and `cv2.cvtColor(RGB2YUV)` are restated
// from OpenCV's published definitions (resize.cpp source-coordinate rule, the 0.299 / 0.587 / 0.114 luma row) -- parity with
// cv2 itself is unpinned; the HDR branch (to_gray_tensor, shift, log10, normalisations) is pinned by a golden from the
// reference.
#include "common.h"

namespace {

__device__ __forceinline__ void lin_coord_(int d, double scale, int n_src, int& s0, int& s1, float& w1) {
#pragma clang fp contract(off)
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { s = 0; f = 0.f; }
  if (s >= n_src - 1) { s = n_src - 1; f = 0.f; }
  s0 = s; s1 = min(s + 1, n_src - 1); w1 = f;
}

// src: (H, W, 3) fp32 (the .npy layout).  Virtual image R = cv2.resize(src, (rw, rh)); output = R[yy:yy+P, xx:xx+P] as
// color (3, P, P) planes and y_out (P, P) = y_scale * (0.299 R + 0.587 G + 0.114 B)
__global__ __launch_bounds__(256) void loader_resize_crop_kernel(const float* __restrict__ src, int H, int W, int rh, int rw, int yy,
                                                                 int xx, int P, float y_scale, float* __restrict__ color,
                                                                 float* __restrict__ y_out) {
#pragma clang fp contract(off)
  const double sc_x = 1.0 / ((double)rw / (double)W), sc_y = 1.0 / ((double)rh / (double)H);
  const size_t total = (size_t)P * P;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int oy = (int)(i / P), ox = (int)(i - (size_t)oy * P);
    float c[3];
    if (rh == H && rw == W) {        // cv2.resize to the same size is a copy
      const float* p = src + ((size_t)(yy + oy) * W + xx + ox) * 3;
      c[0] = p[0]; c[1] = p[1]; c[2] = p[2];
    } else {
      int x0, x1, y0, y1;
      float fx, fy;
      lin_coord_(xx + ox, sc_x, W, x0, x1, fx);
      lin_coord_(yy + oy, sc_y, H, y0, y1, fy);
      const float ax = 1.f - fx, ay = 1.f - fy;
      const float* p00 = src + ((size_t)y0 * W + x0) * 3;
      const float* p01 = src + ((size_t)y0 * W + x1) * 3;
      const float* p10 = src + ((size_t)y1 * W + x0) * 3;
      const float* p11 = src + ((size_t)y1 * W + x1) * 3;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float top = p00[k] * ax + p01[k] * fx;
        const float bot = p10[k] * ax + p11[k] * fx;
        c[k] = top * ay + bot * fy;
      }
    }
    color[i] = c[0]; color[total + i] = c[1]; color[2 * total + i] = c[2];
    // y_scale < 0: DIVIDE by -y_scale (the reference's `/ 255`, ProcessedDatasetFolder.py:18-19, is a division: multiplying
    // by 1/255 differs from it in the last bit of some values)
    if (y_out) {
      const float yv = c[0] * 0.299f + c[1] * 0.587f + c[2] * 0.114f;
      y_out[i] = y_scale < 0.f ? __fdiv_rn(yv, -y_scale) : yv * y_scale;
    }
  }
}

// st: [2] luminance min, [3] luminance max of `color` (as written by uncl_hdr_log_gray).  gray_norm = Y / Y.max(),
// gray_shift = Y - Y.min()   (ProcessedDatasetFolderImg.py:137-144), Y = to_gray_tensor(color) (hdr_image_util.py:68-74)
__global__ __launch_bounds__(256) void loader_gray_kernel(const float* __restrict__ color, size_t hw, const float* __restrict__ st,
                                                          float* __restrict__ gray_norm, float* __restrict__ gray_shift) {
#pragma clang fp contract(off)
  const float gmin = st[2], gmax = st[3];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (size_t)gridDim.x * 256) {
    const float y = (0.299f * color[i] + 0.587f * color[hw + i]) + 0.114f * color[2 * hw + i];
    gray_norm[i] = y / gmax;
    gray_shift[i] = y - gmin;
  }
}

// y = x / max(x) ("max_normalization") or clip(((x - min) / max) * a - b, 0, 1) ("stretch"); st = {min, max} on the device
__global__ __launch_bounds__(256) void loader_ldr_norm_kernel(float* __restrict__ x, size_t n, const float* __restrict__ st, int mode,
                                                              float max_stretch, float min_stretch) {
#pragma clang fp contract(off)
  const float mn = st[0], mx = st[1];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float v = x[i];
    if (mode == 0) v = v / mx;
    else v = fminf(fmaxf(((v - mn) / mx) * max_stretch - min_stretch, 0.f), 1.f);
    x[i] = v;
  }
}

__global__ __launch_bounds__(256) void loader_minmax_kernel(const float* __restrict__ x, size_t n, float* __restrict__ partial) {
  __shared__ float smn[4], smx[4];
  float mn = INFINITY, mx = -INFINITY;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { mn = fminf(mn, x[i]); mx = fmaxf(mx, x[i]); }
  for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o, 64)); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); }
  if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
    partial[2 * blockIdx.x + 1] = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
  }
}
__global__ void loader_minmax_final_kernel(const float* __restrict__ partial, int count, float* __restrict__ st) {
  float mn = INFINITY, mx = -INFINITY;
  for (int i = threadIdx.x; i < count; i += 64) { mn = fminf(mn, partial[2 * i]); mx = fmaxf(mx, partial[2 * i + 1]); }
  for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o, 64)); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); }
  if (threadIdx.x == 0) { st[0] = mn; st[1] = mx; }
}

inline int nbl(size_t n, int cap = 1024) {
  const size_t b = (n + 255) / 256;
  return (int)(b < (size_t)cap ? (b ? b : 1) : (size_t)cap);
}

}  // namespace

extern "C" int uncl_loader_resize_crop(const float* src_hwc, int H, int W, int rh, int rw, int yy, int xx, int patch, float y_scale,
                                       float* color_chw, float* y_plane, void* stream) {
  if (!src_hwc || !color_chw || H <= 0 || W <= 0 || rh <= 0 || rw <= 0 || patch <= 0) return UNCL_ERR_ARG;
  if (yy < 0 || xx < 0 || yy + patch > rh || xx + patch > rw) return UNCL_ERR_ARG;
  hipLaunchKernelGGL(loader_resize_crop_kernel, dim3(nbl((size_t)patch * patch)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     src_hwc, H, W, rh, rw, yy, xx, patch, y_scale, color_chw, y_plane);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

extern "C" int uncl_loader_gray_outputs(const float* color_chw, int H, int W, const float* stats, float* gray_norm, float* gray_shift,
                                        void* stream) {
  if (!color_chw || !stats || !gray_norm || !gray_shift) return UNCL_ERR_ARG;
  hipLaunchKernelGGL(loader_gray_kernel, dim3(nbl((size_t)H * W)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), color_chw,
                     (size_t)H * W, stats, gray_norm, gray_shift);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

// mode 0: x /= max(x); mode 1: clip(((x - min) / max) * max_stretch - min_stretch, 0, 1).  workspace: 2 * 1024 + 2 floats
extern "C" int uncl_loader_ldr_normalize(float* x, long long n, int mode, float max_stretch, float min_stretch, void* workspace,
                                         void* stream) {
  if (!x || n <= 0 || !workspace || (mode != 0 && mode != 1)) return UNCL_ERR_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  float* part = reinterpret_cast<float*>(workspace);
  float* st = part + 2048;
  const int blocks = nbl((size_t)n);
  hipLaunchKernelGGL(loader_minmax_kernel, dim3(blocks), dim3(256), 0, s, x, (size_t)n, part);
  hipLaunchKernelGGL(loader_minmax_final_kernel, dim3(1), dim3(64), 0, s, part, blocks, st);
  hipLaunchKernelGGL(loader_ldr_norm_kernel, dim3(blocks), dim3(256), 0, s, x, (size_t)n, st, mode, max_stretch, min_stretch);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}
