// Implicit-GEMM convolution for gfx950 (MI355X): 3x3 (valid / full) and 1x1, NHWC activations, MFMA 32x32.
//
// One workgroup = 256 threads = 4 wavefronts.  The workgroup owns an output tile of MT_Y x (MT_X*32) pixels
// and CT = 32*NT output channels; wave w owns MPW = MT_Y*MT_X/4 M-tiles (32 pixels each) x NT N-tiles.
// The GEMM is oriented D[cout][pixel] = W[cout][k] * X[k][pixel]: the MFMA "A" operand is the weight
// fragment, "B" the activation fragment, so every lane ends up holding 4 *consecutive* output channels of one
// pixel per register quad -> NHWC stores are 8-byte (bf16) / 16-byte (f32) vectors, bias/activation are
// lane-local, and the trailing 1-channel 1x1 (outc) is an in-lane dot product plus one cross-half shuffle.
//
// K is walked in 64-byte chunks (32 bf16 / 16 f32 channels).  Per chunk the workgroup stages
//   sX: the (MT_Y+k-1) x (MT_X*32+k-1) input halo tile,  [pixel][64 B]
//   sW: the weights of all taps for its CT channels,      [tap][cout][64 B]
// into LDS with 16-byte writes.  Both images XOR-swizzle the 16-byte slot with bits 2-3 of the row index so a
// ds_read_b128 lane group (16 lanes = 16 consecutive rows mod 16) covers all 16 slots of the 256-byte bank
// row: conflict-free for every tap offset.
//
// The loader synthesises the reference's input-side ops on the fly instead of materialising them in HBM:
// MaxPool2d(2) (unet_parts.py:212,233), the skip concat [x2, x1, x2^2, sqrt(x2+1e-8)] with replicate padding
// of x1 (unet_parts.py:292-298, 319-322), and the video generator's recurrent channel hand-off (Unet.py:244,270).
#include <cstdlib>

#include "common.h"

namespace {

struct ConvArgs {
  const void* src0;
  const void* src1;
  const void* prev0;
  const void* weight;
  const float* bias;
  const float* scale_n;
  const void* res;
  void* out;
  const float* out1_w;
  const float* out1_b;
  float* out1;
  int N, H, W, Cin, Cout, pad, src_mode;
  int s0H, s0W, s0C, s1H, s1W, s1C, prev_ch;
  int act, res_b0, oH, oW, oC;
  int z_mode, n_ct, Hout, Wout, tiles_x;
  int flat, rH, rW;
  int out1_act, skip_main;
};

template <typename T>
__device__ __forceinline__ typename Elem<T>::vec ldg(const void* base, size_t elem_off) {
  return *reinterpret_cast<const typename Elem<T>::vec*>(reinterpret_cast<const T*>(base) + elem_off);
}

// channels [0, prev_ch) of this vector come from the previous frame's tensor
template <typename T>
__device__ __forceinline__ typename Elem<T>::vec mix_prev(typename Elem<T>::vec v, const void* prev, size_t off, int c,
                                                           int prev_ch) {
  if (prev != nullptr && c < prev_ch) {
    typename Elem<T>::vec p = ldg<T>(prev, off);
#pragma unroll
    for (int i = 0; i < Elem<T>::EPV; ++i)
      if (c + i < prev_ch) v[i] = p[i];
  }
  return v;
}

// One 16-byte vector of logical input channel chunk (kc, ch) at logical pixel (n, iy, ix); caller guarantees
// 0 <= iy < H, 0 <= ix < W.
template <typename T>
__device__ __forceinline__ typename Elem<T>::vec load_src(const ConvArgs& a, int c_log, int n, int iy, int ix) {
  using E = Elem<T>;
  using vec = typename E::vec;
  if (a.src_mode == UNCL_SRC_PLAIN) {
    size_t off = (((size_t)n * a.s0H + iy) * a.s0W + ix) * a.s0C + c_log;
    vec v = ldg<T>(a.src0, off);
    return mix_prev<T>(v, a.prev0, off, c_log, a.prev_ch);
  }
  if (a.src_mode == UNCL_SRC_MAXPOOL2) {
    float m[E::EPV];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      size_t off = (((size_t)n * a.s0H + 2 * iy + (q >> 1)) * a.s0W + 2 * ix + (q & 1)) * a.s0C + c_log;
      vec v = mix_prev<T>(ldg<T>(a.src0, off), a.prev0, off, c_log, a.prev_ch);
      float f[E::EPV];
      E::unpack(v, f);
#pragma unroll
      for (int i = 0; i < E::EPV; ++i) m[i] = (q == 0) ? f[i] : fmaxf(m[i], f[i]);
    }
    return E::pack(m);
  }
  // concat modes: group g of width s0C
  const int g = c_log / a.s0C;
  const int c = c_log - g * a.s0C;
  if (g == 1) {
    int dy = (a.s0H - a.s1H) >> 1, dx = (a.s0W - a.s1W) >> 1;
    int sy = min(max(iy - dy, 0), a.s1H - 1), sx = min(max(ix - dx, 0), a.s1W - 1);
    return ldg<T>(a.src1, (((size_t)n * a.s1H + sy) * a.s1W + sx) * a.s1C + c);
  }
  vec v = ldg<T>(a.src0, (((size_t)n * a.s0H + iy) * a.s0W + ix) * a.s0C + c);
  if (g == 0) return v;
  float f[E::EPV];
  E::unpack(v, f);
  if (g == 2) {
#pragma unroll
    for (int i = 0; i < E::EPV; ++i) f[i] = f[i] * f[i];
  } else {
#pragma unroll
    for (int i = 0; i < E::EPV; ++i) f[i] = sqrtf(f[i] + 1e-8f);
  }
  return E::pack(f);
}

template <typename T>
__device__ __forceinline__ f32x16 mma(const typename Elem<T>::vec& a, const typename Elem<T>::vec& b, f32x16 c);

template <>
__device__ __forceinline__ f32x16 mma<bf16_t>(const bf16x8& a, const bf16x8& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x16 mma<f16_t>(const f16x8& a, const f16x8& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x16 mma<float>(const f32x4& a, const f32x4& b, f32x16 c) {
  // exact fp32: four K=2 steps; lane half h supplies channel 4h+j of the 8-channel group for step j
#pragma unroll
  for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], c, 0, 0, 0);
  return c;
}

template <typename T>
__device__ __forceinline__ void store4(void* base, size_t elem_off, const float* v);
template <>
__device__ __forceinline__ void store4<float>(void* base, size_t off, const float* v) {
  *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + off) = f32x4{v[0], v[1], v[2], v[3]};
}
template <>
__device__ __forceinline__ void store4<bf16_t>(void* base, size_t off, const float* v) {
  bf16x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = (bf16_t)v[i];
  *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(base) + off) = o;
}
template <>
__device__ __forceinline__ void store4<f16_t>(void* base, size_t off, const float* v) {
  f16x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = (f16_t)v[i];
  *reinterpret_cast<f16x4*>(reinterpret_cast<f16_t*>(base) + off) = o;
}
template <typename T>
__device__ __forceinline__ void load4(const void* base, size_t off, float* v);
template <>
__device__ __forceinline__ void load4<f16_t>(const void* base, size_t off, float* v) {
  f16x4 t = *reinterpret_cast<const f16x4*>(reinterpret_cast<const f16_t*>(base) + off);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = (float)t[i];
}
template <>
__device__ __forceinline__ void load4<float>(const void* base, size_t off, float* v) {
  f32x4 t = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + off);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = t[i];
}
template <>
__device__ __forceinline__ void load4<bf16_t>(const void* base, size_t off, float* v) {
  bf16x4 t = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(base) + off);
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = (float)t[i];
}

template <typename T, int KS, int MT_Y, int MT_X, int NT>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs a) {
  using E = Elem<T>;
  using vec = typename E::vec;
  constexpr int KC = E::KC, EPV = E::EPV;
  constexpr int TAPS = KS * KS;
  constexpr int TILE_H = MT_Y, TILE_W = MT_X * 32;
  constexpr int HH = TILE_H + KS - 1, HW = TILE_W + KS - 1;
  constexpr int NPIX = HH * HW;
  constexpr int CT = NT * 32;
  constexpr int MPW = MT_Y * MT_X / 4;
  static_assert(MPW * 4 == MT_Y * MT_X, "M-tiles must split evenly over 4 waves");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sX = smem;
  char* sW = smem + NPIX * 64;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;

  const int tile_y = blockIdx.x / a.tiles_x, tile_x = blockIdx.x - tile_y * a.tiles_x;
  const int n = blockIdx.y;
  const int ct = blockIdx.z % a.n_ct, zg = blockIdx.z / a.n_ct;
  const int y0 = tile_y * TILE_H, x0 = tile_x * TILE_W;
  const int cout0 = ct * CT;
  const int pad = (KS == 1) ? 0 : a.pad;

  int in_ch_off = 0, out_ch_off = 0, ooy = 0, oox = 0, os = 1;
  size_t w_z_off = 0;
  if (a.z_mode == UNCL_Z_GROUPS) {
    in_ch_off = zg * a.Cin;
    out_ch_off = zg * a.Cout;
    w_z_off = (size_t)zg * TAPS * a.Cout * a.Cin;
  } else if (a.z_mode == UNCL_Z_UP2X2) {
    w_z_off = (size_t)zg * a.Cout * a.Cin;
    ooy = zg >> 1;
    oox = zg & 1;
    os = 2;
  }

  f32x16 acc[MPW][NT];
#pragma unroll
  for (int m = 0; m < MPW; ++m)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][nt][i] = 0.f;

  const int nk = a.Cin / KC;
  for (int kc = 0; kc < nk; ++kc) {
    __syncthreads();
    // ---- stage the input halo tile
    for (int idx = tid; idx < NPIX * 4; idx += 256) {
      const int pix = idx >> 2, ch = idx & 3;
      const int hy = pix / HW, hx = pix - hy * HW;
      const int iy = y0 + hy - pad, ix = x0 + hx - pad;
      vec v = E::zero();
      if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) v = load_src<T>(a, in_ch_off + kc * KC + ch * EPV, n, iy, ix);
      *reinterpret_cast<vec*>(sX + pix * 64 + ((ch ^ ((pix >> 2) & 3)) << 4)) = v;
    }
    // ---- stage the weights of every tap for this K-chunk
    for (int idx = tid; idx < TAPS * CT * 4; idx += 256) {
      const int row = idx >> 2, ch = idx & 3;
      const int tap = row / CT, co = row - tap * CT;
      // (3x3 weights of the 16-bit types are K-chunk-major: common.h)
      const size_t off = uncl_w3_chunk_major(sizeof(T), TAPS, a.Cin)
                             ? w_z_off + uncl_w3_index(tap, cout0 + co, kc * KC + ch * EPV, a.Cout)
                             : w_z_off + ((size_t)(tap * a.Cout + cout0 + co)) * a.Cin + kc * KC + ch * EPV;
      *reinterpret_cast<vec*>(sW + row * 64 + ((ch ^ ((row >> 2) & 3)) << 4)) = ldg<T>(a.weight, off);
    }
    __syncthreads();
    // ---- MFMA over taps x 2 k-steps
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int tyy = tap / KS, txx = tap % KS;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int chunk = 2 * ks + lh;
        vec A[NT], B[MPW];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int row = tap * CT + nt * 32 + lr;
          A[nt] = *reinterpret_cast<const vec*>(sW + row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4));
        }
#pragma unroll
        for (int m = 0; m < MPW; ++m) {
          const int mt = wave * MPW + m;
          const int my = mt / MT_X, mx = mt % MT_X;
          const int pix = (my + tyy) * HW + mx * 32 + lr + txx;
          B[m] = *reinterpret_cast<const vec*>(sX + pix * 64 + ((chunk ^ ((pix >> 2) & 3)) << 4));
        }
#pragma unroll
        for (int m = 0; m < MPW; ++m)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[m][nt] = mma<T>(A[nt], B[m], acc[m][nt]);
      }
    }
  }

  // ---- epilogue: bias, activation, DropPath scale, residual, NHWC store, optional fused 1-channel 1x1
#pragma unroll
  for (int m = 0; m < MPW; ++m) {
    const int mt = wave * MPW + m;
    const int my = mt / MT_X, mx = mt % MT_X;
    const int oy = y0 + my, ox = x0 + mx * 32 + lr;
    const bool valid = (oy < a.Hout) && (ox < a.Wout);
    int n_ = n, yy = oy, xx = ox;
    if (a.flat) {
      const int hw = a.rH * a.rW;
      n_ = ox / hw;
      const int rem = ox - n_ * hw;
      yy = rem / a.rW;
      xx = rem - yy * a.rW;
    }
    const size_t opix = ((size_t)n_ * a.oH + (yy * os + ooy)) * a.oW + (xx * os + oox);
    const size_t rpix = a.res_b0 ? ((size_t)(yy * os + ooy) * a.oW + (xx * os + oox)) : opix;
    const float sc = (a.scale_n != nullptr && valid) ? a.scale_n[n_] : 1.f;
    float o1 = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cb = cout0 + nt * 32 + 8 * q + 4 * lh;  // 4 consecutive output channels
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float t = acc[m][nt][4 * q + r];
          if (a.bias != nullptr) t += a.bias[out_ch_off + cb + r];
          v[r] = uncl_act(t, a.act) * sc;
        }
        if (a.res != nullptr && valid) {
          float rr[4];
          load4<T>(a.res, rpix * a.oC + out_ch_off + cb, rr);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += rr[r];
        }
        if (a.out1_w != nullptr) {
#pragma unroll
          for (int r = 0; r < 4; ++r) o1 += v[r] * a.out1_w[cb + r];
        }
        if (valid && !a.skip_main) store4<T>(a.out, opix * a.oC + out_ch_off + cb, v);
      }
    }
    if (a.out1_w != nullptr) {
      o1 += __shfl_xor(o1, 32, 64);
      if (valid && lh == 0)
        a.out1[((size_t)n_ * a.Hout + oy) * a.Wout + ox] = uncl_act(o1 + a.out1_b[0], a.out1_act);
    }
  }
}

template <typename T, int KS, int MT_Y, int MT_X, int NT>
int launch(const ConvArgs& a, dim3 grid, hipStream_t s) {
  constexpr int HH = MT_Y + KS - 1, HW = MT_X * 32 + KS - 1;
  constexpr size_t lds = (size_t)HH * HW * 64 + (size_t)KS * KS * NT * 32 * 64;
  auto kern = conv_igemm_kernel<T, KS, MT_Y, MT_X, NT>;
  static UnclDevOnce attr_done;  // per instantiation and device
  if (attr_done.need()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
      return UNCL_ERR_LAUNCH;
    attr_done.done();
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

template <typename T>
int dispatch(const ConvArgs& a, int ksize, int nt, dim3 grid, hipStream_t s) {
  if (ksize == 3) {
    if (nt == 1) return launch<T, 3, 8, 1, 1>(a, grid, s);
    if (nt == 2) return launch<T, 3, 8, 1, 2>(a, grid, s);
    return launch<T, 3, 8, 1, 4>(a, grid, s);
  }
  if (nt == 1) return launch<T, 1, 1, 8, 1>(a, grid, s);
  if (nt == 2) return launch<T, 1, 1, 8, 2>(a, grid, s);
  return launch<T, 1, 1, 8, 4>(a, grid, s);
}



// ------------------------------------------------------------------------------------------------------------------
// 1x1 convolutions of the graph block in bf16 (Grapher fc1 / fc2, the grouped graph conv, FFN fc1 / fc2: Unet_singleFrame.py:
// 25-41, torch_vertex.py:190-199, torch_nn.py:58) -- plain GEMMs over 144 nodes per sample, a few GFLOP, LATENCY-bound: the
// generic kernel above walks K in 32-channel steps with a load -> LDS -> barrier -> multiply round trip each (8-16 serial
// memory latencies, 19-40 us per launch).  Here a workgroup stages its slice of the weights ([CT][CIN], swizzled) once, and
// per 128-pixel tile every wave requests ALL of its K (one 16-byte B fragment per 16 channels, straight from global memory)
// before the first multiply: one memory latency per tile.  Epilogue = the generic kernel's: act(acc + bias) * scale_n + res.
// ------------------------------------------------------------------------------------------------------------------
struct C1Args {
  const bf16_t* x;      // pixels x ld_in
  const bf16_t* w;      // [groups][Cout][CIN] packed
  const float* bias;    // [groups * Cout] or NULL
  const float* scale_n; // per sample or NULL
  const bf16_t* res;    // same layout as out, or NULL
  bf16_t* out;
  int M, n_tiles, ld_in, ld_out, Cout, act, px_per_sample, res_b0, n_ct;
  // GELU fused into the store (the graph block's training passes; same rounding points as the separate gelu kernels):
  // gmode 1: zbuf <- the value as stored (pre-activation), out <- gelu of that ROUNDED value
  // gmode 2: out <- the value as stored x gelu'(zbuf)           (data gradient through the GELU)
  bf16_t* zbuf;
  int gmode;
};

template <int CIN, int NT>
__global__ __launch_bounds__(256) void conv1x1_direct_kernel(const C1Args a) {
  using vec = bf16x8;
  constexpr int KSTEPS = CIN / 16, S = CIN / 8, CT = NT * 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sW = smem;                                   // [CT][CIN] bf16, 16-byte slots XOR row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int ct = blockIdx.y % a.n_ct, grp = blockIdx.y / a.n_ct;
  const int in_off = grp * CIN, out_off = grp * a.Cout + ct * CT;
  const bf16_t* wg = a.w + ((size_t)grp * a.Cout + ct * CT) * CIN;
  // every load of the launch's latency chain is requested before the first wait: the weight slice (CT*S/256 16-byte loads per
  // thread), the biases (as float4s) and the first tile's K fragments
  constexpr int WIT = (CT * S + 255) / 256;
  vec wtmp[WIT];
#pragma unroll
  for (int i = 0; i < WIT; ++i) {
    const int v = min(tid + i * 256, CT * S - 1);
    const int row = v / S, slot = v - row * S;
    wtmp[i] = *reinterpret_cast<const vec*>(wg + (size_t)row * CIN + slot * 8);
  }
  float bv[NT][16];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b4 = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + out_off + nt * 32 + 8 * q + 4 * lh) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[nt][4 * q + r] = b4[r];
    }
  vec B[KSTEPS];
  {
    const int m0 = blockIdx.x * 128 + wave * 32 + lr;
    const bf16_t* xp0 = a.x + (size_t)min(m0, a.M - 1) * a.ld_in + in_off + lh * 8;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) B[ks] = *reinterpret_cast<const vec*>(xp0 + ks * 16);
  }
#pragma unroll
  for (int i = 0; i < WIT; ++i) {
    const int v = tid + i * 256;
    if (v < CT * S) {
      const int row = v / S, slot = v - row * S;
      *reinterpret_cast<vec*>(sW + row * (CIN * 2) + ((slot ^ (row & (S - 1))) << 4)) = wtmp[i];
    }
  }
  __syncthreads();
  for (int t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
    const int m = t * 128 + wave * 32 + lr;
    const bool valid = m < a.M;
    if (t != (int)blockIdx.x) {
      const bf16_t* xp = a.x + (size_t)min(m, a.M - 1) * a.ld_in + in_off + lh * 8;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) B[ks] = *reinterpret_cast<const vec*>(xp + ks * 16);
    }
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[nt][i] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int row = nt * 32 + lr;
        const vec A = *reinterpret_cast<const vec*>(sW + row * (CIN * 2) + (((2 * ks + lh) ^ (row & (S - 1))) << 4));
        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B[ks], acc[nt], 0, 0, 0);
      }
    if (valid) {
      const int smp = m / a.px_per_sample;
      const float sc = a.scale_n ? a.scale_n[smp] : 1.f;
      const size_t opix = (size_t)m * a.ld_out + out_off;
      const size_t rpix = (size_t)(a.res_b0 ? m - smp * a.px_per_sample : m) * a.ld_out + out_off;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int cb = nt * 32 + 8 * q + 4 * lh;
          float v4[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v4[r] = uncl_act(acc[nt][4 * q + r] + bv[nt][4 * q + r], a.act) * sc;
          if (a.res) {
            float rr[4];
            load4<bf16_t>(a.res, rpix + cb, rr);
#pragma unroll
            for (int r = 0; r < 4; ++r) v4[r] += rr[r];
          }
          if (a.gmode == 1) {
            store4<bf16_t>(a.zbuf, opix + cb, v4);
#pragma unroll
            for (int r = 0; r < 4; ++r) v4[r] = uncl_gelu((float)(bf16_t)v4[r]);
          } else if (a.gmode == 2) {
            float zz[4];
            load4<bf16_t>(a.zbuf, opix + cb, zz);
#pragma unroll
            for (int r = 0; r < 4; ++r) {          // as gelu_bwd_kernel (csrc/backward_kernels.hip) on the rounded gradient
              const float x = zz[r];
              const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
              const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
              v4[r] = (float)(bf16_t)v4[r] * (cdf + x * pdf);
            }
          }
          store4<bf16_t>(a.out, opix + cb, v4);
        }
    }
  }
}

template <int CIN, int NT>
int launch_c1(const C1Args& a, int groups, hipStream_t s) {
  constexpr size_t lds = (size_t)NT * 32 * CIN * 2;
  auto kern = conv1x1_direct_kernel<CIN, NT>;
  static UnclDevOnce attr;
  if (attr.need()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return UNCL_ERR_LAUNCH;
    attr.done();
  }
  // every workgroup stages up to 64 KB of weights: a strided share of the tiles per resident slot, not one workgroup per tile
  const int per_cu = (int)(160 * 1024 / lds) < 1 ? 1 : ((int)(160 * 1024 / lds) > 4 ? 4 : (int)(160 * 1024 / lds));
  int gx = (256 * per_cu) / (a.n_ct * groups);
  if (gx < 1) gx = 1;
  if (gx > a.n_tiles) gx = a.n_tiles;
  hipLaunchKernelGGL(kern, dim3(gx, a.n_ct * groups), dim3(256), lds, s, a);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

// returns C1_NOT_MINE when the descriptor is outside this kernel's scope (the caller then takes the generic path)
constexpr int C1_NOT_MINE = 1;
int conv1x1_direct(const uncl_conv_desc* d, hipStream_t s, void* zbuf = nullptr, int gmode = 0) {
  if (d->dtype != UNCL_BF16 || d->ksize != 1 || d->src_mode != UNCL_SRC_PLAIN || d->prev0 != nullptr || d->out1_w != nullptr ||
      d->skip_main_store || d->src1 != nullptr)
    return C1_NOT_MINE;
  if (d->z_mode != UNCL_Z_NONE && d->z_mode != UNCL_Z_GROUPS) return C1_NOT_MINE;
  const int groups = d->z_mode == UNCL_Z_GROUPS ? d->groups : 1;
  if (d->Cin != 128 && d->Cin != 256 && d->Cin != 512) return C1_NOT_MINE;
  const long long M = (long long)d->N * d->H * d->W;
  if (M <= 0 || M > 0x7fffffffLL / 1024) return C1_NOT_MINE;
  // output channels per workgroup: 128 (64 for K = 512: 64 KB of weights at most) when there are pixel tiles to amortise the
  // weight staging over, 32 for the small batches of the training steps, where the launch is one workgroup's latency chain
  const long long tiles = (M + 127) / 128;
  // the latency regime only (training batches: a few dozen samples x 144 nodes): with hundreds of pixel tiles the generic
  // kernel's small weight chunks win over staging a weight slice per workgroup (measured at 100 / 200 tiles: 25-40 us vs
  // 40-70 us), here one memory latency per launch wins over eight (9 us vs 19 us)
  static const int force = [] { const char* e = getenv("UNCL_C1_FORCE"); return e ? atoi(e) : 0; }();   // A/B: tools/c1_probe.py
  if (tiles > 48 && !force) return C1_NOT_MINE;
  int nt = d->Cin == 512 ? 2 : 4;
  while (nt > 1 && tiles * (d->Cout / (nt * 32)) * groups < 256) nt /= 2;
  if (d->Cout % (nt * 32) != 0) return C1_NOT_MINE;
  if (d->res != nullptr && d->res_batch_stride0 && (d->out_H != d->H || d->out_W != d->W)) return C1_NOT_MINE;
  C1Args a;
  a.x = (const bf16_t*)d->src0; a.w = (const bf16_t*)d->weight; a.bias = d->bias; a.scale_n = d->scale_n;
  a.res = (const bf16_t*)d->res; a.out = (bf16_t*)d->out;
  a.M = (int)M; a.n_tiles = (int)((M + 127) / 128); a.ld_in = d->src0_C; a.ld_out = d->out_C; a.Cout = d->Cout; a.act = d->act;
  a.px_per_sample = d->H * d->W; a.res_b0 = d->res_batch_stride0; a.n_ct = d->Cout / (nt * 32);
  a.zbuf = (bf16_t*)zbuf; a.gmode = zbuf ? gmode : 0;
  if (d->Cin == 128) return nt == 4 ? launch_c1<128, 4>(a, groups, s) : (nt == 2 ? launch_c1<128, 2>(a, groups, s) : launch_c1<128, 1>(a, groups, s));
  if (d->Cin == 256) return nt == 4 ? launch_c1<256, 4>(a, groups, s) : (nt == 2 ? launch_c1<256, 2>(a, groups, s) : launch_c1<256, 1>(a, groups, s));
  return nt == 2 ? launch_c1<512, 2>(a, groups, s) : launch_c1<512, 1>(a, groups, s);
}

}  // namespace

// Internal (generator.hip): the bf16 1x1 convolution of `d` with the GELU of the graph block fused into its store -- gmode 1
// forward (zbuf receives the pre-activation, d->out the activation), gmode 2 backward (d->out = conv x gelu'(zbuf)); zbuf has
// d->out's layout.  Returns UNCL_GELU_NOT_FUSED (> 0) when the direct kernel does not take the descriptor: the caller then runs
// the convolution and the gelu kernel as two launches.
int uncl_conv1x1_gelu(const uncl_conv_desc* d, void* zbuf, int gmode, void* stream) {
  if (d == nullptr || zbuf == nullptr || (gmode != 1 && gmode != 2)) return UNCL_ERR_ARG;
  static const int on = [] { const char* e = getenv("UNCL_C1_GELU"); return e ? atoi(e) : 1; }();     // 0: two launches (A/B)
  if (!on) return C1_NOT_MINE;
  return conv1x1_direct(d, reinterpret_cast<hipStream_t>(stream), zbuf, gmode);
}

extern "C" int uncl_conv_igemm(const uncl_conv_desc* d, void* stream) {
  if (d == nullptr) return UNCL_ERR_ARG;
  if (d->dtype != UNCL_F32 && !uncl_is_h16(d->dtype)) return UNCL_ERR_ARG;
  if (d->ksize != 3 && d->ksize != 1) return UNCL_ERR_ARG;
  const int KC = uncl_is_h16(d->dtype) ? 32 : 16;
  if (d->Cin <= 0 || d->Cin % KC != 0 || d->Cout <= 0 || d->Cout % 32 != 0) return UNCL_ERR_ARG;
  if (d->ksize == 3 && d->pad != 0 && d->pad != 2) return UNCL_ERR_ARG;
  if (d->src_mode < UNCL_SRC_PLAIN || d->src_mode > UNCL_SRC_CONCAT2) return UNCL_ERR_ARG;
  if (d->src0 == nullptr || d->weight == nullptr) return UNCL_ERR_ARG;
  if (d->out == nullptr && !(d->skip_main_store && d->out1 != nullptr)) return UNCL_ERR_ARG;
  if ((d->src_mode == UNCL_SRC_CONCAT_SSR || d->src_mode == UNCL_SRC_CONCAT2)) {
    if (d->src1 == nullptr || d->src0_C != d->src1_C || d->src0_C % KC != 0) return UNCL_ERR_ARG;
    const int groups = d->src_mode == UNCL_SRC_CONCAT_SSR ? 4 : 2;
    if (d->Cin != groups * d->src0_C) return UNCL_ERR_ARG;
    if (d->src1_H > d->src0_H || d->src1_W > d->src0_W) return UNCL_ERR_ARG;
  }
  if (d->src_mode == UNCL_SRC_MAXPOOL2 && (d->H != d->src0_H / 2 || d->W != d->src0_W / 2)) return UNCL_ERR_ARG;
  if (d->out1_w != nullptr && (d->Cout != 32 || d->out1 == nullptr || d->z_mode != UNCL_Z_NONE)) return UNCL_ERR_ARG;
  if (d->z_mode == UNCL_Z_UP2X2 && d->ksize != 1) return UNCL_ERR_ARG;
  if (d->z_mode == UNCL_Z_GROUPS && d->groups <= 0) return UNCL_ERR_ARG;

  {
    const int rc1 = conv1x1_direct(d, reinterpret_cast<hipStream_t>(stream));
    if (rc1 != C1_NOT_MINE) return rc1;
  }
  ConvArgs a;
  a.src0 = d->src0; a.src1 = d->src1; a.prev0 = d->prev0;
  a.weight = d->weight; a.bias = d->bias; a.scale_n = d->scale_n; a.res = d->res; a.out = d->out;
  a.out1_w = d->out1_w; a.out1_b = d->out1_b; a.out1 = d->out1;
  a.Cin = d->Cin; a.Cout = d->Cout; a.pad = d->pad; a.src_mode = d->src_mode;
  a.s0H = d->src0_H; a.s0W = d->src0_W; a.s0C = d->src0_C;
  a.s1H = d->src1_H; a.s1W = d->src1_W; a.s1C = d->src1_C;
  a.prev_ch = d->prev_ch;
  a.act = d->act; a.res_b0 = d->res_batch_stride0;
  a.oH = d->out_H; a.oW = d->out_W; a.oC = d->out_C;
  a.z_mode = d->z_mode; a.out1_act = d->out1_act; a.skip_main = d->skip_main_store;

  int nt = d->Cout >= 128 ? 4 : (d->Cout == 64 ? 2 : (d->Cout % 128 == 0 ? 4 : (d->Cout % 64 == 0 ? 2 : 1)));
  // bf16 1x1 layers (the graph block: a few GFLOP on 144 nodes per sample) are launch-parallelism bound, not tile-efficiency
  // bound: 128 pixels x 64 channels per workgroup gives 4x the workgroups of the 256 x 128 tile
  const bool small_1x1 = d->ksize == 1 && uncl_is_h16(d->dtype) && d->z_mode != UNCL_Z_UP2X2 && nt >= 2;
  if (small_1x1) nt = 2;
  const int CT = nt * 32;
  if (d->Cout % CT != 0) return UNCL_ERR_ARG;
  a.n_ct = d->Cout / CT;
  int zcount = 1;
  if (d->z_mode == UNCL_Z_GROUPS) zcount = d->groups;
  if (d->z_mode == UNCL_Z_UP2X2) zcount = 4;

  dim3 grid;
  if (d->ksize == 3) {
    a.flat = 0; a.rH = d->H; a.rW = d->W;
    a.N = d->N; a.H = d->H; a.W = d->W;
    a.Hout = d->H + 2 * d->pad - 2; a.Wout = d->W + 2 * d->pad - 2;
    if (a.Hout <= 0 || a.Wout <= 0) return UNCL_ERR_ARG;
    a.tiles_x = (a.Wout + 31) / 32;
    const int tiles_y = (a.Hout + 7) / 8;
    grid = dim3(a.tiles_x * tiles_y, d->N, a.n_ct * zcount);
  } else {
    // 1x1: flatten (N,H,W) into one row of M pixels; the epilogue decodes (n,y,x) again
    if (d->src_mode != UNCL_SRC_PLAIN) return UNCL_ERR_ARG;
    const long long M = (long long)d->N * d->H * d->W;
    if (M <= 0 || M > 0x7fffffffLL) return UNCL_ERR_ARG;
    a.flat = 1; a.rH = d->H; a.rW = d->W;
    a.N = 1; a.H = 1; a.W = (int)M;
    a.s0H = 1; a.s0W = (int)M;
    a.Hout = 1; a.Wout = (int)M;
    const int tw = small_1x1 ? 128 : 256;
    a.tiles_x = (int)((M + tw - 1) / tw);
    grid = dim3(a.tiles_x, 1, a.n_ct * zcount);
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (small_1x1) return d->dtype == UNCL_F16 ? launch<f16_t, 1, 1, 4, 2>(a, grid, s) : launch<bf16_t, 1, 1, 4, 2>(a, grid, s);
  if (d->dtype == UNCL_BF16) return dispatch<bf16_t>(a, d->ksize, nt, grid, s);
  if (d->dtype == UNCL_F16) return dispatch<f16_t>(a, d->ksize, nt, grid, s);
  return dispatch<float>(a, d->ksize, nt, grid, s);
}
