// Inference pre- / post-processing either side of the overlap tiler, on device:
//   load_inference / load_inference2 arithmetic (utils/model_save_util.py:203-240): shift, luminance, log compression
//   resize_im / add_frame_to_im (utils/data_loader_util.py:135-185): replicate padding to the tiler's grid
//   np.percentile clamp, min-max stretch, back_to_color_tensor (model_save_util.py:389-401, hdr_image_util.py:120-131)
//   save_gray_tensor_as_numpy_stretch / to_0_1_range_outlier (hdr_image_util.py:93-103, 237-241): 8-bit output
// All of it is HBM-bound byte work: coalesced fp32 passes, two-stage deterministic reductions, and an exact radix select
// for the order statistics behind np.percentile (no sort).  Arithmetic that the reference does in fp32 with separately
// rounded multiplies and adds uses __fmul_rn / __fadd_rn so that no FMA contraction changes the last bit.
#include "common.h"

namespace {

constexpr int RB = 1024;  // reduction workgroups (partials per quantity)

__device__ __forceinline__ float gray_of(float r, float g, float b) {
  // hdr_image_util.to_gray_tensor: (0.299 * r + 0.587 * g + 0.114 * b), every product and sum rounded to fp32
  return __fadd_rn(__fadd_rn(__fmul_rn(0.299f, r), __fmul_rn(0.587f, g)), __fmul_rn(0.114f, b));
}

__device__ __forceinline__ void block_minmax(float& mn, float& mx) {
  __shared__ float smn[4], smx[4];
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_down(mn, o));
    mx = fmaxf(mx, __shfl_down(mx, o));
  }
  if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; ++i) { mn = fminf(mn, smn[i]); mx = fmaxf(mx, smx[i]); }
  }
}

// partial[2*b] = min, partial[2*b+1] = max over this block's slice of x (n values)
__global__ __launch_bounds__(256) void minmax_kernel(const float* __restrict__ x, size_t n, float* __restrict__ partial) {
  float mn = INFINITY, mx = -INFINITY;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = x[i];
    mn = fminf(mn, v); mx = fmaxf(mx, v);
  }
  block_minmax(mn, mx);
  if (threadIdx.x == 0) { partial[2 * blockIdx.x] = mn; partial[2 * blockIdx.x + 1] = mx; }
}

// min / max of the luminance of (rgb - shift), shift = min(rgb_min, 0)
__global__ __launch_bounds__(256) void gray_minmax_kernel(const float* __restrict__ rgb, size_t hw, const float* __restrict__ st,
                                                          float* __restrict__ partial) {
  const float sh = fminf(st[0], 0.f);
  float mn = INFINITY, mx = -INFINITY;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (size_t)gridDim.x * 256) {
    const float v = gray_of(rgb[i] - sh, rgb[hw + i] - sh, rgb[2 * hw + i] - sh);
    mn = fminf(mn, v); mx = fmaxf(mx, v);
  }
  block_minmax(mn, mx);
  if (threadIdx.x == 0) { partial[2 * blockIdx.x] = mn; partial[2 * blockIdx.x + 1] = mx; }
}

// st[dst] = min over partials, st[dst+1] = max over partials
__global__ void minmax_final_kernel(const float* __restrict__ partial, int count, float* __restrict__ st, int dst) {
  float mn = INFINITY, mx = -INFINITY;
  for (int i = threadIdx.x; i < count; i += 64) { mn = fminf(mn, partial[2 * i]); mx = fmaxf(mx, partial[2 * i + 1]); }
  for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_down(mn, o)); mx = fmaxf(mx, __shfl_down(mx, o)); }
  if (threadIdx.x == 0) { st[dst] = mn; st[dst + 1] = mx; }
}

// st: [0] rgb min, [1] rgb max, [2] gray min, [3] gray max.  rgb_out = rgb - min(rgb_min,0);
// gray_log = log10(((gray - gmin) / (gmax - gmin)) * f + 1) / log10(f + 1)      (model_save_util.py:213-217)
__global__ __launch_bounds__(256) void log_gray_kernel(const float* __restrict__ rgb, size_t hw, const float* __restrict__ st,
                                                       float f_factor, float* __restrict__ rgb_out, float* __restrict__ gray_log) {
  const float sh = fminf(st[0], 0.f);
  const float gmin = st[2], gmax = __fsub_rn(st[3], st[2]);
  // the reference divides by the maximum of the compressed image, which is log10(1 * f + 1) evaluated in fp32
  const float top = log10f(__fadd_rn(__fmul_rn(1.0f, f_factor), 1.0f));
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (size_t)gridDim.x * 256) {
    const float r = rgb[i] - sh, g = rgb[hw + i] - sh, b = rgb[2 * hw + i] - sh;
    if (rgb_out) { rgb_out[i] = r; rgb_out[hw + i] = g; rgb_out[2 * hw + i] = b; }
    const float gr = __fsub_rn(gray_of(r, g, b), gmin);
    const float v = log10f(__fadd_rn(__fmul_rn(__fdiv_rn(gr, gmax), f_factor), 1.0f));
    gray_log[i] = __fdiv_rn(v, top);
  }
}

// y[p][i][j] = x[p][clamp(i - top, 0, H-1)][clamp(j - left, 0, W-1)]     (F.pad(..., mode='replicate'))
__global__ __launch_bounds__(256) void replicate_pad_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W,
                                                            int top, int left, int H1, int W1) {
  const int p = blockIdx.y;
  const float* xs = x + (size_t)p * H * W;
  float* yd = y + (size_t)p * H1 * W1;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < H1 * W1; i += gridDim.x * 256) {
    const int r = i / W1, c = i - r * W1;
    const int sr = min(max(r - top, 0), H - 1), sc = min(max(c - left, 0), W - 1);
    yd[i] = xs[(size_t)sr * W + sc];
  }
}

// ---- exact order statistics by radix select (np.percentile needs the two neighbours of a fractional rank) ----------
constexpr int MAXR = 8;
struct SelState {
  unsigned prefix[MAXR];         // key bits decided so far (high bits)
  unsigned long long rank[MAXR]; // rank still to be resolved inside the prefix bucket
  unsigned hist[MAXR][256];
};

__device__ __forceinline__ unsigned key_of(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // order-preserving map of fp32 onto uint32
}
__device__ __forceinline__ float val_of(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ __launch_bounds__(256) void sel_hist_kernel(const float* __restrict__ x, size_t n, SelState* __restrict__ st, int nr,
                                                       int pass) {
  __shared__ unsigned h[MAXR][256];
  for (int i = threadIdx.x; i < nr * 256; i += 256) h[i >> 8][i & 255] = 0;
  __syncthreads();
  const int shift = 24 - 8 * pass;
  const unsigned himask = pass == 0 ? 0u : (0xffffffffu << (32 - 8 * pass));
  unsigned pre[MAXR];
  for (int r = 0; r < nr; ++r) pre[r] = st->prefix[r];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const unsigned k = key_of(x[i]);
    for (int r = 0; r < nr; ++r)
      if ((k & himask) == pre[r]) atomicAdd(&h[r][(k >> shift) & 255u], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nr * 256; i += 256)
    if (h[i >> 8][i & 255]) atomicAdd(&st->hist[i >> 8][i & 255], h[i >> 8][i & 255]);
}

// one wave per rank: find the bin holding the rank, extend the prefix, clear the histogram for the next pass
__global__ void sel_scan_kernel(SelState* __restrict__ st, int nr, int pass, float* __restrict__ out) {
  const int r = blockIdx.x;
  if (threadIdx.x == 0) {
    unsigned long long want = st->rank[r], acc = 0;
    int bin = 255;
    for (int b = 0; b < 256; ++b) {
      const unsigned c = st->hist[r][b];
      if (acc + c > want) { bin = b; break; }
      acc += c;
    }
    st->rank[r] = want - acc;
    st->prefix[r] |= (unsigned)bin << (24 - 8 * pass);
    if (pass == 3) out[r] = val_of(st->prefix[r]);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < 256; b += 64) st->hist[r][b] = 0;
}

__global__ void sel_init_kernel(SelState* __restrict__ st, int nr, const unsigned long long* __restrict__ ranks) {
  const int r = blockIdx.x;
  if (threadIdx.x == 0) { st->prefix[r] = 0; st->rank[r] = ranks[r]; }
  for (int b = threadIdx.x; b < 256; b += 64) st->hist[r][b] = 0;
}

// out[c][i][j] = sqrt(rgb[c][i+top][j+left] / (gray + eps)) * (clamp(fake[i+top][j+left], lo, hi) - lo) / (hi - lo)
// over the cropped window (H x W inside the padded H1 x W1); model_save_util.py:393-401, hdr_image_util.py:120-131
__global__ __launch_bounds__(256) void color_finish_kernel(const float* __restrict__ rgb, const float* __restrict__ fake,
                                                           float* __restrict__ out, int H1, int W1, int top, int left, int H,
                                                           int W, float lo, float hi, float eps, const float* __restrict__ lohi) {
  const size_t hw1 = (size_t)H1 * W1, hw = (size_t)H * W;
  if (lohi) { lo = lohi[0]; hi = lohi[1]; }     // percentiles left on the device (uncl_percentile_lerp)
  const float span = __fsub_rn(hi, lo);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < H * W; i += gridDim.x * 256) {
    const int r = i / W, c = i - r * W;
    const size_t s = (size_t)(r + top) * W1 + c + left;
    const float f = __fdiv_rn(__fsub_rn(fminf(fmaxf(fake[s], lo), hi), lo), span);
    const float cr = rgb[s], cg = rgb[hw1 + s], cb = rgb[2 * hw1 + s];
    const float den = __fadd_rn(gray_of(cr, cg, cb), eps);
    out[i] = __fmul_rn(sqrtf(__fdiv_rn(cr, den)), f);
    out[hw + i] = __fmul_rn(sqrtf(__fdiv_rn(cg, den)), f);
    out[2 * hw + i] = __fmul_rn(sqrtf(__fdiv_rn(cb, den)), f);
  }
}

// 8-bit image: clamp(x,0,1) -> (v - lo) / (hi - lo) -> clip(0,1) -> uint8(v * 255), CHW fp32 -> HWC uint8
// (hdr_image_util.py:237-241 save_gray_tensor_as_numpy_stretch with to_0_1_range_outlier :93-103)
__global__ __launch_bounds__(256) void to_uint8_kernel(const float* __restrict__ x, unsigned char* __restrict__ out, int C,
                                                       size_t hw, float lo, float hi, const float* __restrict__ lohi) {
  if (lohi) {
    lo = lohi[0]; hi = lohi[1];
    if (__fsub_rn(hi, lo) == 0.f) hi = __fadd_rn(hi, 1e-8f);      // hdr_image_util.py:98-99
  }
  const float span = __fsub_rn(hi, lo);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < hw * C; i += (size_t)gridDim.x * 256) {
    const size_t p = i / C;
    const int c = (int)(i - p * C);
    float v = fminf(fmaxf(x[(size_t)c * hw + p], 0.f), 1.f);
    v = __fdiv_rn(__fsub_rn(v, lo), span);
    v = fminf(fmaxf(v, 0.f), 1.f);
    out[i] = (unsigned char)(__fmul_rn(v, 255.f));
  }
}

__global__ __launch_bounds__(256) void clamp01_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = fminf(fmaxf(x[i], 0.f), 1.f);
}

inline int nb(size_t n, int cap) {
  const size_t b = (n + 255) / 256;
  return (int)(b < (size_t)cap ? (b ? b : 1) : cap);
}

}  // namespace

extern "C" size_t uncl_frame_workspace_bytes(void) { return (size_t)RB * 2 * 4 + 64 + sizeof(SelState) + MAXR * 8 + 256; }

// rgb: fp32 (3,H,W) linear radiance.  rgb_out (3,H,W, may be NULL): rgb shifted to be non-negative; gray_log (H,W): the
// generator's input; stats (4 floats, device): rgb min, rgb max, luminance min, luminance max.
extern "C" int uncl_hdr_log_gray(const float* rgb, int H, int W, float f_factor, float* rgb_out, float* gray_log, float* stats,
                                 void* workspace, void* stream) {
  if (!rgb || !gray_log || !stats || !workspace || H <= 0 || W <= 0 || !(f_factor > 0.f)) return UNCL_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  float* partial = reinterpret_cast<float*>(workspace);
  const size_t hw = (size_t)H * W;
  const int b3 = nb(3 * hw, RB), b1 = nb(hw, RB);
  hipLaunchKernelGGL(minmax_kernel, dim3(b3), dim3(256), 0, st, rgb, 3 * hw, partial);
  hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(64), 0, st, partial, b3, stats, 0);
  hipLaunchKernelGGL(gray_minmax_kernel, dim3(b1), dim3(256), 0, st, rgb, hw, stats, partial);
  hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(64), 0, st, partial, b1, stats, 2);
  hipLaunchKernelGGL(log_gray_kernel, dim3(nb(hw, 4096)), dim3(256), 0, st, rgb, hw, stats, f_factor, rgb_out, gray_log);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

extern "C" int uncl_replicate_pad(const float* x, float* y, int planes, int H, int W, int top, int left, int H1, int W1,
                                  void* stream) {
  if (!x || !y || planes <= 0 || H <= 0 || W <= 0 || top < 0 || left < 0 || H1 < H + top || W1 < W + left) return UNCL_ERR_ARG;
  hipLaunchKernelGGL(replicate_pad_kernel, dim3(nb((size_t)H1 * W1, 2048), planes), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), x, y, H, W, top, left, H1, W1);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

// out[r] = the ranks[r]-th smallest value of x (0-based, ranks < n), exact.  ranks: HOST array, nr <= 8.
extern "C" int uncl_order_stats(const float* x, long long n, const unsigned long long* ranks, int nr, float* out, void* workspace,
                                void* stream) {
  if (!x || !ranks || !out || !workspace || n <= 0 || nr <= 0 || nr > MAXR) return UNCL_ERR_ARG;
  for (int r = 0; r < nr; ++r)
    if (ranks[r] >= (unsigned long long)n) return UNCL_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  char* w = reinterpret_cast<char*>(workspace) + (size_t)RB * 2 * 4 + 64;
  SelState* state = reinterpret_cast<SelState*>(w);
  unsigned long long* dranks = reinterpret_cast<unsigned long long*>(w + sizeof(SelState));
  if (hipMemcpyAsync(dranks, ranks, sizeof(unsigned long long) * nr, hipMemcpyHostToDevice, st) != hipSuccess) return UNCL_ERR_LAUNCH;
  hipLaunchKernelGGL(sel_init_kernel, dim3(nr), dim3(64), 0, st, state, nr, dranks);
  for (int pass = 0; pass < 4; ++pass) {
    hipLaunchKernelGGL(sel_hist_kernel, dim3(nb((size_t)n, 1024)), dim3(256), 0, st, x, (size_t)n, state, nr, pass);
    hipLaunchKernelGGL(sel_scan_kernel, dim3(nr), dim3(64), 0, st, state, nr, pass, out);
  }
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

// rgb: (3,H1,W1) padded, non-negative; fake: (H1,W1) generator output; out: (3,H,W) = window at (top,left)
extern "C" int uncl_color_finish(const float* rgb, const float* fake, float* out, int H1, int W1, int top, int left, int H, int W,
                                 float lo, float hi, void* stream) {
  if (!rgb || !fake || !out || H <= 0 || W <= 0 || top < 0 || left < 0 || top + H > H1 || left + W > W1 || !(hi > lo))
    return UNCL_ERR_ARG;
  hipLaunchKernelGGL(color_finish_kernel, dim3(nb((size_t)H * W, 4096)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), rgb,
                     fake, out, H1, W1, top, left, H, W, lo, hi, 1e-8f, (const float*)nullptr);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

// the same with [lo, hi] read from device memory (uncl_percentile_lerp output): no host round trip between the tiled forward
// and the colour step
extern "C" int uncl_color_finish_dev(const float* rgb, const float* fake, float* out, int H1, int W1, int top, int left, int H, int W,
                                     const float* lohi, void* stream) {
  if (!rgb || !fake || !out || !lohi || H <= 0 || W <= 0 || top < 0 || left < 0 || top + H > H1 || left + W > W1) return UNCL_ERR_ARG;
  hipLaunchKernelGGL(color_finish_kernel, dim3(nb((size_t)H * W, 4096)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), rgb,
                     fake, out, H1, W1, top, left, H, W, 0.f, 1.f, 1e-8f, lohi);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

extern "C" int uncl_clamp01(const float* x, float* y, long long n, void* stream) {
  if (!x || !y || n <= 0) return UNCL_ERR_ARG;
  hipLaunchKernelGGL(clamp01_kernel, dim3(nb((size_t)n, 4096)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, y, (size_t)n);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

// x: fp32 (C,H,W); out: uint8 (H,W,C)
extern "C" int uncl_to_uint8(const float* x, unsigned char* out, int C, int H, int W, float lo, float hi, void* stream) {
  if (!x || !out || C <= 0 || H <= 0 || W <= 0 || !(hi > lo)) return UNCL_ERR_ARG;
  hipLaunchKernelGGL(to_uint8_kernel, dim3(nb((size_t)C * H * W, 4096)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, out,
                     C, (size_t)H * W, lo, hi, (const float*)nullptr);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

// [lo, hi] from device memory; a zero span gets the reference's +1e-8 on hi (hdr_image_util.py:98-99)
extern "C" int uncl_to_uint8_dev(const float* x, unsigned char* out, int C, int H, int W, const float* lohi, void* stream) {
  if (!x || !out || !lohi || C <= 0 || H <= 0 || W <= 0) return UNCL_ERR_ARG;
  hipLaunchKernelGGL(to_uint8_kernel, dim3(nb((size_t)C * H * W, 4096)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, out,
                     C, (size_t)H * W, 0.f, 1.f, lohi);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

namespace {
struct LerpArgs {
  double gamma[8];
  int f64[8];
  int n;
};
// numpy's _lerp (lib/_function_base_impl.py): a + (b - a) t, and b - (b - a)(1 - t) where t >= 0.5, every operation rounded
// on its own, in the precision numpy's type promotion gives the expression (float32 values with a float32 or float64 gamma)
__global__ void percentile_lerp_kernel(const float* __restrict__ pairs, const LerpArgs t, float* __restrict__ out) {
#pragma clang fp contract(off)      // numpy rounds the product before the sum: no fused multiply-add here
  const int i = threadIdx.x;
  if (i >= t.n) return;
  const float a = pairs[2 * i], b = pairs[2 * i + 1];
  // plain operators under contract(off): the *_rn helpers are inlined with the translation unit's default (contraction on)
  if (t.f64[i]) {
    const double g = t.gamma[i], ad = (double)a, bd = (double)b;
    const double d = bd - ad;
    const double p1 = d * g;
    double r = ad + p1;
    if (g >= 0.5) {
      const double omg = 1.0 - g;
      const double p2 = d * omg;
      r = bd - p2;
    }
    out[i] = (float)r;
  } else {
    const float g = (float)t.gamma[i];
    const float d = b - a;
    const float p1 = d * g;
    float r = a + p1;
    if (g >= 0.5f) {
      const float omg = 1.f - g;
      const float p2 = d * omg;
      r = b - p2;
    }
    out[i] = r;
  }
}
}  // namespace

// out[i] = numpy-lerp(pairs[2i], pairs[2i+1], gamma[i]) on the device.  pairs: device order statistics (uncl_order_stats),
// gamma / f64: HOST arrays (they depend on the element count and the percentile only), n <= 8.
extern "C" int uncl_percentile_lerp(const float* pairs, const double* gamma, const int* f64, int n, float* out, void* stream) {
  if (!pairs || !gamma || !f64 || !out || n <= 0 || n > 8) return UNCL_ERR_ARG;
  LerpArgs t = {};
  t.n = n;
  for (int i = 0; i < n; ++i) { t.gamma[i] = gamma[i]; t.f64[i] = f64[i]; }
  hipLaunchKernelGGL(percentile_lerp_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), pairs, t, out);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// warp_flow (GanTrainer.py:584-595): res = cv2.remap(img, flow + pixel grid, None, cv2.INTER_LINEAR) on an 8-bit image.
// cv2 is absent from the reference tree and from this image: the kernel restates OpenCV's published fixed-point algorithm for
// CV_8U + INTER_LINEAR + BORDER_CONSTANT(0) (imgproc/src/imgwarp.cpp, remap / remapBilinear): coordinates are rounded to 1/32 of
// a pixel (cvRound(x * 32), round-half-even), the four weights are (32 - fy)(32 - fx) * 32 etc. out of 2^15 (exact integers for
// the bilinear table, so the table's sum correction never fires) and the result is (sum + 2^14) >> 15; a tap outside the image
// contributes 0.  One thread per output pixel, all channels.
// ---------------------------------------------------------------------------------------------------------------------
__global__ void warp_flow_kernel(const unsigned char* __restrict__ img, const float* __restrict__ flow, unsigned char* __restrict__ out,
                                 int H, int W, int C, int Hf, int Wf) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= Hf * Wf) return;
  const int y = idx / Wf, x = idx - y * Wf;
  // flow[:, :, 0] += arange(wf); flow[:, :, 1] += arange(hf)[:, None]  (float32 additions)
  const float mx = __fadd_rn(flow[2 * idx], (float)x), my = __fadd_rn(flow[2 * idx + 1], (float)y);
  // cvRound of the float product (exact: a power-of-two scale); non-finite / huge values saturate like saturate_cast<short>
  const float fx32 = fminf(fmaxf(mx * 32.f, -2147483000.f), 2147483000.f), fy32 = fminf(fmaxf(my * 32.f, -2147483000.f), 2147483000.f);
  const int sxq = (fx32 == fx32) ? __float2int_rn(fx32) : 0, syq = (fy32 == fy32) ? __float2int_rn(fy32) : 0;
  const int ax = sxq & 31, ay = syq & 31;
  const int sx = min(max(sxq >> 5, -32768), 32767), sy = min(max(syq >> 5, -32768), 32767);
  const int w00 = (32 - ay) * (32 - ax) * 32, w01 = (32 - ay) * ax * 32, w10 = ay * (32 - ax) * 32, w11 = ay * ax * 32;
  const bool x0 = sx >= 0 && sx < W, x1 = sx + 1 >= 0 && sx + 1 < W, y0 = sy >= 0 && sy < H, y1 = sy + 1 >= 0 && sy + 1 < H;
  for (int c = 0; c < C; ++c) {
    const int p00 = (x0 && y0) ? img[((size_t)sy * W + sx) * C + c] : 0;
    const int p01 = (x1 && y0) ? img[((size_t)sy * W + sx + 1) * C + c] : 0;
    const int p10 = (x0 && y1) ? img[((size_t)(sy + 1) * W + sx) * C + c] : 0;
    const int p11 = (x1 && y1) ? img[((size_t)(sy + 1) * W + sx + 1) * C + c] : 0;
    out[(size_t)idx * C + c] = (unsigned char)((p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + (1 << 14)) >> 15);
  }
}

extern "C" int uncl_warp_flow(const unsigned char* img, const float* flow, unsigned char* out, int H, int W, int C, int Hf, int Wf,
                              void* stream) {
  if (!img || !flow || !out || H <= 0 || W <= 0 || C <= 0 || C > 4 || Hf <= 0 || Wf <= 0 || H > 32767 || W > 32767) return UNCL_ERR_ARG;
  const int n = Hf * Wf;
  hipLaunchKernelGGL(warp_flow_kernel, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), img, flow, out, H, W,
                     C, Hf, Wf);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}
