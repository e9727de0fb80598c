// Dense inverse optical flow for the video evaluator's warp error (Tester.py:379-389; GanTrainer.py:597-646 compute_flow /
// estimate_invflow; metrics/compute_wrap_error.py:91-125).  The reference calls cv2.optflow DeepFlow, a third-party estimator that
// is absent from its tree and from this image: parity with cv2 is unpinned.  These kernels follow oracle/flow.py operation for
// operation -- coarse-to-fine iterative Lucas-Kanade with 15 x 15 box windows -- which is pinned by synthetic motions with known flow
// (tests/test_flow.py).  Everything is fp32 and HBM-bound: 4 - 20 bytes per pixel per pass, a few passes per iteration; a 1080p pair
// is ~20 ms, once per evaluated scene.  gfx950 only.
#include "common.h"

namespace {

constexpr int FL_MIN_SIDE = 16, FL_ITERS = 4, FL_RADIUS = 7;
constexpr float FL_LAM = 1e-2f;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// 5-tap binomial blur (edge replicate) + every second pixel: dst (Hd, Wd) = blur(src)[::2, ::2]
__global__ __launch_bounds__(256) void flow_down_kernel(const float* __restrict__ src, int H, int W, float* __restrict__ dst, int Hd, int Wd) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Hd * Wd) return;
  const int y = i / Wd, x = i - y * Wd;
  const float k[5] = {1.f / 16.f, 4.f / 16.f, 6.f / 16.f, 4.f / 16.f, 1.f / 16.f};
  float v = 0.f;
#pragma unroll
  for (int a = 0; a < 5; ++a) {
    const float* row = src + (size_t)clampi(2 * y + a - 2, 0, H - 1) * W;
    float h = 0.f;
#pragma unroll
    for (int b = 0; b < 5; ++b) h += k[b] * row[clampi(2 * x + b - 2, 0, W - 1)];
    v += k[a] * h;
  }
  dst[i] = v;
}

// central differences of the source image (edge replicate) and their three products
__global__ __launch_bounds__(256) void flow_grad_kernel(const float* __restrict__ S, int H, int W, float* __restrict__ Ix, float* __restrict__ Iy,
                                                        float* __restrict__ pxx, float* __restrict__ pxy, float* __restrict__ pyy) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= H * W) return;
  const int y = i / W, x = i - y * W;
  const float gx = 0.5f * (S[(size_t)y * W + clampi(x + 1, 0, W - 1)] - S[(size_t)y * W + clampi(x - 1, 0, W - 1)]);
  const float gy = 0.5f * (S[(size_t)clampi(y + 1, 0, H - 1) * W + x] - S[(size_t)clampi(y - 1, 0, H - 1) * W + x]);
  Ix[i] = gx; Iy[i] = gy;
  pxx[i] = gx * gx; pxy[i] = gx * gy; pyy[i] = gy * gy;
}

__device__ __forceinline__ float bilinear(const float* __restrict__ img, int H, int W, float x, float y) {
  x = fminf(fmaxf(x, 0.f), (float)(W - 1));
  y = fminf(fmaxf(y, 0.f), (float)(H - 1));
  const int x0 = W > 1 ? min((int)floorf(x), W - 2) : 0, y0 = H > 1 ? min((int)floorf(y), H - 2) : 0;
  const float fx = x - (float)x0, fy = y - (float)y0;
  const int x1 = W > 1 ? x0 + 1 : 0, y1 = H > 1 ? y0 + 1 : 0;
  const float a = img[(size_t)y0 * W + x0], b = img[(size_t)y0 * W + x1], c = img[(size_t)y1 * W + x0], d = img[(size_t)y1 * W + x1];
  return (1.f - fy) * ((1.f - fx) * a + fx * b) + fy * ((1.f - fx) * c + fx * d);
}

// It = bilinear(A, p + f) - S, and the two products with the source gradients
__global__ __launch_bounds__(256) void flow_it_kernel(const float* __restrict__ A, const float* __restrict__ S, const float* __restrict__ Ix,
                                                      const float* __restrict__ Iy, const float* __restrict__ fx, const float* __restrict__ fy, int H,
                                                      int W, float* __restrict__ pxt, float* __restrict__ pyt) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= H * W) return;
  const int y = i / W, x = i - y * W;
  const float it = bilinear(A, H, W, (float)x + fx[i], (float)y + fy[i]) - S[i];
  pxt[i] = Ix[i] * it;
  pyt[i] = Iy[i] * it;
}

// zero-padded box SUM along one axis (radius r), `n` planes of H x W one behind the other
template <bool HORIZ>
__global__ __launch_bounds__(256) void flow_box_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int r, int n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)n * H * W) return;
  const int p = (int)(i % ((size_t)H * W));
  const float* s = src + (i - p);
  const int y = p / W, x = p - y * W;
  float v = 0.f;
  if (HORIZ) {
    for (int k = max(x - r, 0); k <= min(x + r, W - 1); ++k) v += s[(size_t)y * W + k];
  } else {
    for (int k = max(y - r, 0); k <= min(y + r, H - 1); ++k) v += s[(size_t)k * W + x];
  }
  dst[i] = v;
}

// d = -G^-1 b with G = window sums + LAM I, clamped to one pixel per component; f <- f + d
__global__ __launch_bounds__(256) void flow_solve_kernel(const float* __restrict__ gxx, const float* __restrict__ gxy, const float* __restrict__ gyy,
                                                         const float* __restrict__ bx, const float* __restrict__ by, float* __restrict__ fx,
                                                         float* __restrict__ fy, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float a = gxx[i] + FL_LAM, b = gxy[i], c = gyy[i] + FL_LAM;
  const float det = a * c - b * b;
  const float dx = -(c * bx[i] - b * by[i]) / det, dy = -(a * by[i] - b * bx[i]) / det;
  fx[i] += fminf(fmaxf(dx, -1.f), 1.f);
  fy[i] += fminf(fmaxf(dy, -1.f), 1.f);
}

// 3 x 3 mean over the pixels inside the image (two planes)
__global__ __launch_bounds__(256) void flow_smooth_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)2 * H * W) return;
  const int p = (int)(i % ((size_t)H * W));
  const float* s = src + (i - p);
  const int y = p / W, x = p - y * W;
  float v = 0.f;
  int cnt = 0;
  for (int yy = max(y - 1, 0); yy <= min(y + 1, H - 1); ++yy)
    for (int xx = max(x - 1, 0); xx <= min(x + 1, W - 1); ++xx) { v += s[(size_t)yy * W + xx]; ++cnt; }
  dst[i] = v / (float)cnt;
}

// this level's field from the coarser one: 2 * bilinear(fc, x / 2, y / 2), two planes
__global__ __launch_bounds__(256) void flow_up_kernel(const float* __restrict__ fc, int Hc, int Wc, float* __restrict__ f, int H, int W) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)2 * H * W) return;
  const int pl = (int)(i / ((size_t)H * W)), p = (int)(i - (size_t)pl * H * W);
  const int y = p / W, x = p - y * W;
  f[i] = 2.f * bilinear(fc + (size_t)pl * Hc * Wc, Hc, Wc, 0.5f * (float)x, 0.5f * (float)y);
}

// planes (fx, fy) -> interleaved (H, W, 2)
__global__ __launch_bounds__(256) void flow_interleave_kernel(const float* __restrict__ f, float* __restrict__ out, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  out[2 * i] = f[i];
  out[2 * i + 1] = f[n + i];
}

inline unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }

struct Levels {
  int n, h[16], w[16];
  size_t total;      // pixels of all levels
};
Levels make_levels(int H, int W) {
  Levels L;
  L.n = 0; L.total = 0;
  int h = H, w = W;
  for (;;) {
    L.h[L.n] = h; L.w[L.n] = w; L.total += (size_t)h * w; ++L.n;
    if ((h < w ? h : w) < 2 * FL_MIN_SIDE || L.n == 16) break;
    h = (h + 1) / 2; w = (w + 1) / 2;
  }
  return L;
}

}  // namespace

// workspace: both pyramids + 16 planes of the finest level
extern "C" size_t uncl_optical_flow_workspace_bytes(int H, int W) {
  if (H < 2 || W < 2) return 0;
  const Levels L = make_levels(H, W);
  return (2 * L.total + 16 * (size_t)H * W) * sizeof(float);
}

// img_to_align, img_source: (H, W) fp32 planes in [0, 255] (channel 0 of the two frames, GanTrainer.py:640-641); flow: (H, W, 2) fp32,
// flow[..., 0] along x, with img_to_align(p + flow(p)) ~ img_source(p) -- what uncl_warp_flow / GanTrainer.warp_flow take.
extern "C" int uncl_optical_flow(const float* img_to_align, const float* img_source, int H, int W, float* flow, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  if (!img_to_align || !img_source || !flow || !workspace || H < 2 || W < 2 || (long long)H * W >= (1LL << 30)) return UNCL_ERR_ARG;
  if (workspace_bytes < uncl_optical_flow_workspace_bytes(H, W)) return UNCL_ERR_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const Levels L = make_levels(H, W);
  float* ws = reinterpret_cast<float*>(workspace);
  const float* pa[16];
  const float* ps[16];
  pa[0] = img_to_align; ps[0] = img_source;
  float* p = ws;
  for (int l = 1; l < L.n; ++l) {
    float* da = p; p += (size_t)L.h[l] * L.w[l];
    float* ds = p; p += (size_t)L.h[l] * L.w[l];
    hipLaunchKernelGGL(flow_down_kernel, dim3(nblk((size_t)L.h[l] * L.w[l])), dim3(256), 0, s, pa[l - 1], L.h[l - 1], L.w[l - 1], da, L.h[l], L.w[l]);
    hipLaunchKernelGGL(flow_down_kernel, dim3(nblk((size_t)L.h[l] * L.w[l])), dim3(256), 0, s, ps[l - 1], L.h[l - 1], L.w[l - 1], ds, L.h[l], L.w[l]);
    pa[l] = da; ps[l] = ds;
  }
  float* planes = ws + 2 * L.total;            // 16 planes of H x W (coarser levels use the head of each)
  const size_t HW = (size_t)H * W;
  float* Ix = planes; float* Iy = planes + HW;
  float* prod = planes + 2 * HW;                // pxx, pxy, pyy (3 planes, contiguous)
  float* G = planes + 5 * HW;                   // gxx, gxy, gyy
  float* tmp = planes + 8 * HW;                 // 3 planes of scratch for the separable box
  float* pt = planes + 11 * HW;                 // pxt, pyt -> (after the box) bx, by in place of tmp's first two ... see below
  float* f = planes + 13 * HW;                  // fx, fy
  float* f2 = planes + 15 * HW;                 // one more plane pair is needed for the smoothing / up-sampling: reuse tmp (below)
  (void)f2;
  float* fprev = nullptr;
  int hp = 0, wp = 0;
  for (int l = L.n - 1; l >= 0; --l) {
    const int h = L.h[l], w = L.w[l];
    const size_t n = (size_t)h * w;
    // planes of this level: packed at pitch n inside their groups
    float* lIx = Ix; float* lIy = Iy;
    float* lprod = prod;                         // 3 n
    float* lG = G;                               // 3 n
    float* ltmp = tmp;                           // 3 n
    float* lpt = pt;                             // 2 n
    float* lf = f;                               // 2 n
    if (fprev == nullptr) {
      if (hipMemsetAsync(lf, 0, 2 * n * sizeof(float), s) != hipSuccess) return UNCL_ERR_LAUNCH;
    } else {
      // the coarser field sits in ltmp (copied there below): up-sample into lf
      hipLaunchKernelGGL(flow_up_kernel, dim3(nblk(2 * n)), dim3(256), 0, s, fprev, hp, wp, lf, h, w);
    }
    hipLaunchKernelGGL(flow_grad_kernel, dim3(nblk(n)), dim3(256), 0, s, ps[l], h, w, lIx, lIy, lprod, lprod + n, lprod + 2 * n);
    hipLaunchKernelGGL(flow_box_kernel<true>, dim3(nblk(3 * n)), dim3(256), 0, s, lprod, ltmp, h, w, FL_RADIUS, 3);
    hipLaunchKernelGGL(flow_box_kernel<false>, dim3(nblk(3 * n)), dim3(256), 0, s, ltmp, lG, h, w, FL_RADIUS, 3);
    for (int it = 0; it < FL_ITERS; ++it) {
      hipLaunchKernelGGL(flow_it_kernel, dim3(nblk(n)), dim3(256), 0, s, pa[l], ps[l], lIx, lIy, lf, lf + n, h, w, lpt, lpt + n);
      hipLaunchKernelGGL(flow_box_kernel<true>, dim3(nblk(2 * n)), dim3(256), 0, s, lpt, ltmp, h, w, FL_RADIUS, 2);
      hipLaunchKernelGGL(flow_box_kernel<false>, dim3(nblk(2 * n)), dim3(256), 0, s, ltmp, lpt, h, w, FL_RADIUS, 2);
      hipLaunchKernelGGL(flow_solve_kernel, dim3(nblk(n)), dim3(256), 0, s, lG, lG + n, lG + 2 * n, lpt, lpt + n, lf, lf + n, (int)n);
      hipLaunchKernelGGL(flow_smooth_kernel, dim3(nblk(2 * n)), dim3(256), 0, s, lf, ltmp, h, w);
      if (hipMemcpyAsync(lf, ltmp, 2 * n * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return UNCL_ERR_LAUNCH;
    }
    if (l > 0) {
      // keep this level's field where the next (finer) level will not overwrite it before it has up-sampled it: the tail of tmp's
      // third plane group is free until that level's first box pass, which runs AFTER its up-sampling
      fprev = tmp + HW;                          // planes 9, 10 of the finest-level layout: 2 n <= 2 HW / 4 floats fit
      if (hipMemcpyAsync(fprev, lf, 2 * n * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return UNCL_ERR_LAUNCH;
      hp = h; wp = w;
    }
  }
  hipLaunchKernelGGL(flow_interleave_kernel, dim3(nblk(HW)), dim3(256), 0, s, f, flow, (int)HW);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}
