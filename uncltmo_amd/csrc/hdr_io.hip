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

// out: (3, Ho, Wo) fp32 planes.  scale 1: the image itself.  Even scale s: cv2.resize(img, (W//s, H//s)) with its default
// INTER_LINEAR, whose sample point (d + 0.5) s - 0.5 lies midway between source pixels s d + s/2 - 1 and s d + s/2 on both
// axes: horizontal pass 0.5 a + 0.5 b per row, then the same vertically (model_save_util.py:226).
__global__ __launch_bounds__(256) void rgbe_to_planes_kernel(const uint8_t* __restrict__ rgbe, float* __restrict__ out, int H, int W,
                                                             int Ho, int Wo, int scale) {
  const size_t total = (size_t)Ho * Wo;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int oy = (int)(i / Wo), ox = (int)(i - (size_t)oy * Wo);
    float r, g, b;
    if (scale == 1) {
      rgbe_px(rgbe + ((size_t)oy * W + ox) * 4, r, g, b);
    } else {
      const int y0 = oy * scale + scale / 2 - 1, x0 = ox * scale + scale / 2 - 1;
      float c[2][3];
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {
        float r0, g0, b0, r1, g1, b1;
        const int yy = min(y0 + dy, H - 1);
        rgbe_px(rgbe + ((size_t)yy * W + x0) * 4, r0, g0, b0);
        rgbe_px(rgbe + ((size_t)yy * W + min(x0 + 1, W - 1)) * 4, r1, g1, b1);
        c[dy][0] = r0 * 0.5f + r1 * 0.5f; c[dy][1] = g0 * 0.5f + g1 * 0.5f; c[dy][2] = b0 * 0.5f + b1 * 0.5f;
      }
      r = c[0][0] * 0.5f + c[1][0] * 0.5f; g = c[0][1] * 0.5f + c[1][1] * 0.5f; b = c[0][2] * 0.5f + c[1][2] * 0.5f;
    }
    out[i] = r; out[total + i] = g; out[2 * total + i] = b;
  }
}

}  // namespace

// rgbe: device (H, W, 4) bytes; out: device (3, H/scale, W/scale) fp32; scale 1 or an even factor
extern "C" int uncl_rgbe_to_planes(const uint8_t* rgbe, float* out, int H, int W, int scale, void* stream) {
  if (!rgbe || !out || H <= 0 || W <= 0 || scale < 1 || (scale > 1 && (scale & 1))) return UNCL_ERR_ARG;
  const int Ho = H / scale, Wo = W / scale;
  if (Ho <= 0 || Wo <= 0) return UNCL_ERR_ARG;
  const size_t total = (size_t)Ho * Wo;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(rgbe_to_planes_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), rgbe, out, H, W, Ho, Wo,
                     scale);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}
