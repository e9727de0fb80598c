// Radiance .hdr (RGBE) input for the inference entry: `hdr_image_util.read_hdr_image` reads these
// through imageio's FreeImage plugin (utils/hdr_image_util.py:35-39), a dependency that is not part of the reference tree.
// The file format is restated from its published description (Radiance `color.c` / Bruce Walter's rgbe.c, which FreeImage's
// PluginHDR follows): header lines up to an empty one, the resolution line "-Y H +X W", then per scanline either the
// run-length form [2, 2, W>>8, W&255, then R, G, B, E planes each as (count>128: run of count-128 | count: literals)] or
// flat RGBE quadruples.  value = mantissa * 2^(E - 136), zero when E == 0 (no +0.5 bias: rgbe.c, not Radiance's colr_color).
//
// The byte-level run-length decode is sequential host work (uncl_rgbe_decode, no GPU involved); the conversion to fp32
// planes (and the integer down-scale the reference's load_inference2 applies, model_save_util.py:225-226) is a device kernel.
#include "common.h"

#include <cstring>

// data: the bytes after the resolution line.  out: H*W*4 RGBE bytes, row-major.  Returns UNCL_OK or UNCL_ERR_ARG on a
// truncated / malformed stream.
extern "C" int uncl_rgbe_decode(const uint8_t* data, size_t n, int H, int W, uint8_t* out) {
  if (!data || !out || H <= 0 || W <= 0) return UNCL_ERR_ARG;
  size_t p = 0;
  const bool rle_width = W >= 8 && W <= 0x7fff;
  for (int y = 0; y < H; ++y) {
    uint8_t* row = out + (size_t)y * W * 4;
    bool rle = false;
    if (rle_width && p + 4 <= n && data[p] == 2 && data[p + 1] == 2 && !(data[p + 2] & 0x80)) {
      if (((int)data[p + 2] << 8 | data[p + 3]) != W) return UNCL_ERR_ARG;
      rle = true;
      p += 4;
    }
    if (!rle) {
      // flat pixels for the rest of the file (rgbe.c: a file either is run-length encoded or is not)
      const size_t need = ((size_t)(H - y)) * W * 4;
      if (p + need > n) return UNCL_ERR_ARG;
      memcpy(row, data + p, need);
      return UNCL_OK;
    }
    for (int c = 0; c < 4; ++c) {
      int x = 0;
      while (x < W) {
        if (p >= n) return UNCL_ERR_ARG;
        int cnt = data[p++];
        if (cnt > 128) {
          cnt -= 128;
          if (cnt == 0 || x + cnt > W || p >= n) return UNCL_ERR_ARG;
          const uint8_t v = data[p++];
          for (int i = 0; i < cnt; ++i) row[(size_t)(x + i) * 4 + c] = v;
        } else {
          if (cnt == 0 || x + cnt > W || p + cnt > n) return UNCL_ERR_ARG;
          for (int i = 0; i < cnt; ++i) row[(size_t)(x + i) * 4 + c] = data[p + i];
          p += cnt;
        }
        x += cnt;
      }
    }
  }
  return UNCL_OK;
}

namespace {

__device__ __forceinline__ void rgbe_px(const uint8_t* q, float& r, float& g, float& b) {
  const uchar4 v = *reinterpret_cast<const uchar4*>(q);
  if (v.w == 0) { r = g = b = 0.f; return; }
  const float f = ldexpf(1.0f, (int)v.w - 136);
  r = (float)v.x * f; g = (float)v.y * f; b = (float)v.z * f;
}

// cv2's INTER_LINEAR source coordinate for destination index d (resize.cpp: fx = (float)((d + 0.5) * scale - 0.5) with
// scale = 1 / ((double)n_dst / n_src); s = floor(fx), fx -= s; clamped to the first / last source sample)
__device__ __forceinline__ void lin_coord(int d, double scale, int n_src, int& s0, int& s1, float& w1) {
#pragma clang fp contract(off)
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { s = 0; f = 0.f; }
  if (s >= n_src - 1) { s = n_src - 1; f = 0.f; }
  s0 = s; s1 = min(s + 1, n_src - 1); w1 = f;
}

// out: (3, Ho, Wo) fp32 planes.  scale 1: the image itself.  Otherwise cv2.resize(img, (W//scale, H//scale)) with its default
// INTER_LINEAR (model_save_util.py:225-226): the per-axis ratio is W / (W // scale), NOT `scale`, so for sizes that are not
// multiples of `scale` (belgium.hdr is 769 x 1025) the sample point drifts across the image; horizontal pass first
// (a (1 - fx) + b fx per source row), then the same vertically, every product and sum rounded separately like cv2's float path.
__global__ __launch_bounds__(256) void rgbe_to_planes_kernel(const uint8_t* __restrict__ rgbe, float* __restrict__ out, int H, int W,
                                                             int Ho, int Wo, int scale) {
#pragma clang fp contract(off)      // cv2 (and the oracle) round every product before the sum: no fused multiply-add here
  const size_t total = (size_t)Ho * Wo;
  const double sc_x = 1.0 / ((double)Wo / (double)W), sc_y = 1.0 / ((double)Ho / (double)H);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int oy = (int)(i / Wo), ox = (int)(i - (size_t)oy * Wo);
    float r, g, b;
    if (scale == 1) {
      rgbe_px(rgbe + ((size_t)oy * W + ox) * 4, r, g, b);
    } else {
      int x0, x1, y0, y1;
      float fx, fy;
      lin_coord(ox, sc_x, W, x0, x1, fx);
      lin_coord(oy, sc_y, H, y0, y1, fy);
      const float ax = 1.f - fx, ay = 1.f - fy;
      float c[2][3];
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {
        float r0, g0, b0, r1, g1, b1;
        const int yy = dy ? y1 : y0;
        rgbe_px(rgbe + ((size_t)yy * W + x0) * 4, r0, g0, b0);
        rgbe_px(rgbe + ((size_t)yy * W + x1) * 4, r1, g1, b1);
        // plain operators: the contract(off) pragma governs THESE instructions (the __fmul_rn / __fadd_rn wrappers are inlined
        // with their header's own floating-point options and fuse again)
        c[dy][0] = r0 * ax + r1 * fx;
        c[dy][1] = g0 * ax + g1 * fx;
        c[dy][2] = b0 * ax + b1 * fx;
      }
      r = c[0][0] * ay + c[1][0] * fy;
      g = c[0][1] * ay + c[1][1] * fy;
      b = c[0][2] * ay + c[1][2] * fy;
    }
    out[i] = r; out[total + i] = g; out[2 * total + i] = b;
  }
}

}  // namespace

// rgbe: device (H, W, 4) bytes; out: device (3, H/scale, W/scale) fp32; scale >= 1 (H, W need not be multiples of it)
extern "C" int uncl_rgbe_to_planes(const uint8_t* rgbe, float* out, int H, int W, int scale, void* stream) {
  if (!rgbe || !out || H <= 0 || W <= 0 || scale < 1) return UNCL_ERR_ARG;
  const int Ho = H / scale, Wo = W / scale;
  if (Ho <= 0 || Wo <= 0) return UNCL_ERR_ARG;
  const size_t total = (size_t)Ho * Wo;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(rgbe_to_planes_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), rgbe, out, H, W, Ho, Wo,
                     scale);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}
