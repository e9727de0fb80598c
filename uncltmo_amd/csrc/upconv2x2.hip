// ConvTranspose2d(k=2, s=2) + bias for gfx950 (bf16), the `up` of every decoder stage (unet_parts.py:269,288).
//
// out[n, 2y+dy, 2x+dx, co] = b[co] + sum_ci x[n, y, x, ci] * W[ci, co, dy, dx]
//
// This is a skinny GEMM (K = Cin <= 256, N = 4*Cout) that reads each input pixel once and writes four output
// pixels: 25 FLOP per byte at C = 32, i.e. HBM-bound.  The kernel is therefore built around its memory access
// pattern, not the MFMA rate:
//   * activations go straight from global memory to MFMA B-fragments (each pixel is used by exactly one wave);
//   * the packed weights [tap][cout][cin] are one [4*Cout][Cin] matrix; a workgroup keeps a 128-row slice in LDS
//     (XOR-swizzled 16-byte slots) for all the pixel tiles it walks;
//   * the 128 pixels x 128 virtual channels result tile is transposed through LDS so that every store
//     instruction writes whole 2*Cout-element runs of an output row (a wave writes contiguous KiBs), instead of
//     8-byte pieces scattered at a 2-pixel stride.
#include "common.h"
#include <cstdlib>

#ifndef UNCL_UP_PREFETCH
#define UNCL_UP_PREFETCH 1
#endif
#ifndef UNCL_UP_PREFETCH_MAXC
#define UNCL_UP_PREFETCH_MAXC 256
#endif
// input widths from which a workgroup takes a 64-row slice of the virtual channels instead of 128 (512 = never), and whether
// the narrow form keeps the next tile's fragments in flight (the registers decide between two and three waves per SIMD).
// Same box, min of 3, ms per 200 tiles (FILES=upconv2x2 tools/ab_variants.sh, tools/ab_layers.sh): 256-channel level 0.059 ->
// 0.047 - 0.049 with or without the prefetch; the 128-channel level (two wide workgroups per CU already) 0.067 -> 0.071: stays wide.
#ifndef UNCL_UP_NARROW_MINC
#define UNCL_UP_NARROW_MINC 256
#endif
#ifndef UNCL_UP_PREFETCH_NARROW
#define UNCL_UP_PREFETCH_NARROW 1
#endif

namespace {

struct UpArgs {
  const bf16_t* x;
  const bf16_t* prev;
  const bf16_t* w;      // [4*Cout][Cin]
  const float* bias;    // [Cout]
  bf16_t* out;          // (N, 2H, 2W, Cout)
  int H, W, C, Cout, prev_ch;
  int M;                // N*H*W input pixels
  int n_tiles;          // ceil(M / 128)
  UNCL_CHK_MEMBER       // checked build: the tensors of this launch (common.h)
};

// CT = virtual output channels (tap-major) per workgroup: 128, or 64 for the wide levels -- with CIN = 256 a 128-row weight slice
// plus the 128 x 128 result image is 96 KB, i.e. ONE four-wave workgroup per CU, and nothing covers its load / multiply /
// transpose / store phases; 64 rows are 48 KB (three workgroups per CU)
template <typename T, int CIN, bool PREV, int CT>
__global__ __launch_bounds__(256) void upconv2x2_kernel(const UpArgs a) {
  using E = Elem<T>;
  using vec = typename Elem<T>::vec;
  using vec4 = typename Elem<T>::vec4;
  constexpr int KS = CIN / 16;           // MFMA k-steps
  constexpr int S = CIN / 8;             // 16-byte slots per weight row
  constexpr int NTC = CT / 32;           // 32-row MFMA tiles of the slice
  constexpr int OS = CT / 8;             // 16-byte slots per pixel of the result image
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sW = smem;                       // [CT][CIN] bf16, swizzled
  char* sO = smem + CT * CIN * 2;        // [128 pixels][CT] bf16, swizzled (slot ^ pixel within a pixel's OS slots)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int ct = blockIdx.y;             // which 128-wide slice of the 4*Cout virtual channels

  auto wswz = [](int row, int slot) {
    const int f = S == 4 ? ((row >> 2) & 3) : (S == 8 ? ((row >> 1) & 7) : (row & 15));
    return slot ^ f;
  };
  // stage this slice of the weights once
  for (int v = tid; v < CT * S; v += 256) {
    const int row = v / S, slot = v - row * S;
    UNCL_CHK(a.chk, a.w + ((size_t)(ct * CT + row)) * CIN + slot * 8, 16);
    const vec wv = *reinterpret_cast<const vec*>(a.w + ((size_t)(ct * CT + row)) * CIN + slot * 8);
    *reinterpret_cast<vec*>(sW + row * (CIN * 2) + (wswz(row, slot) << 4)) = wv;
  }
  // bias of the slice's 128 virtual channels (c' = ct*128 + i -> co = c' % Cout) in LDS: sixty-four registers of per-lane
  // copies kept the kernel at two workgroups per CU
  float* sB = reinterpret_cast<float*>(sO + 128 * CT * 2);
  if (tid < CT) { if (a.bias) UNCL_CHK(a.chk, a.bias + (ct * CT + tid) % a.Cout, 4); sB[tid] = a.bias ? a.bias[(ct * CT + tid) % a.Cout] : 0.f; }
  __syncthreads();

  f32x16 zero16;
#pragma unroll
  for (int i = 0; i < 16; ++i) zero16[i] = 0.f;

  // B fragments of a tile: pixel m0 + wave*32 + lr, channels 16*ks + 8*lh .. +7.  The next tile's are requested right after
  // this tile's MFMAs, so that their latency runs under the transpose and the stores instead of in front of the next tile's
  // first MFMA (same box, us per 200 tiles, C = 64 / 128 / 256 levels: 146 / 72 / 59 -> 116 / 68 / 59 together with the bias
  // moved from 64 registers per lane into LDS, which took the kernel from two to three workgroups per CU).
  constexpr bool PRE = CIN <= UNCL_UP_PREFETCH_MAXC && UNCL_UP_PREFETCH && (CT == 128 || UNCL_UP_PREFETCH_NARROW);
  auto loadB = [&](int t, vec* Bv) __attribute__((always_inline)) {
    const int mp = min(t * 128 + wave * 32 + lr, a.M - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const size_t off = (size_t)mp * CIN + ks * 16 + lh * 8;
      UNCL_CHK(a.chk, a.x + off, 16);
      Bv[ks] = *reinterpret_cast<const vec*>(a.x + off);
      if (PREV && ks * 16 + lh * 8 < a.prev_ch) {
        UNCL_CHK(a.chk, a.prev + off, 16);
        const vec p = *reinterpret_cast<const vec*>(a.prev + off);
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (ks * 16 + lh * 8 + i < a.prev_ch) Bv[ks][i] = p[i];
      }
    }
  };
  vec B[KS], Bn[PRE ? KS : 1];
  if (PRE && (int)blockIdx.x < a.n_tiles) loadB(blockIdx.x, B);
  for (int t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
    const int m0 = t * 128;
    if (!PRE) loadB(t, B);
    f32x16 acc[NTC];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int nt = 0; nt < NTC; ++nt) {
        const int row = nt * 32 + lr;
        const vec A = *reinterpret_cast<const vec*>(sW + row * (CIN * 2) + (wswz(row, 2 * ks + lh) << 4));
        acc[nt] = mfma32x16(A, B[ks], ks == 0 ? zero16 : acc[nt]);
      }
    }
    if (PRE) {
      const int tn = t + (int)gridDim.x;
      if (tn < a.n_tiles) loadB(tn, Bn);
    }
    // ---- transpose through LDS: image [pixel (128)][virtual channel (128)], 16-byte slots XOR pixel
    __syncthreads();  // previous tile's readers are done with sO
    const int pl = wave * 32 + lr;
#pragma unroll
    for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        vec4 o;
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(sB + nt * 32 + 8 * q + 4 * lh);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (T)(acc[nt][4 * q + r] + b4[r]);
        *reinterpret_cast<vec4*>(sO + pl * (CT * 2) + (((nt * 4 + q) ^ (pl & (OS - 1))) << 4) + (lh << 3)) = o;
      }
    __syncthreads();
    // ---- coalesced stores.  A run = the channels of one (pixel, tap-row) that are contiguous in the output:
    // RUN virtual channels = min(128, 2*Cout) -> RUN/8 slots; runs of consecutive pixels are adjacent in memory.
    const int run = min(CT, 2 * a.Cout);          // virtual channels per contiguous run
    const int rs = run >> 3;                      // slots per run (a power of two)
    const int rs_sh = __builtin_ctz(rs);
    const int runs_per_px = CT / run;             // 2 (Cout=32), 1 otherwise
    // pixel coordinates without per-element integer division: one division per tile for the sample index, then an exact
    // float reciprocal (rem < H*W <= 2^16) with a one-step correction for the row
    const int hw = a.H * a.W;
    const int n0 = m0 / hw, rem0 = m0 - n0 * hw;
    const float rcpW = 1.0f / (float)a.W;
    for (int v = tid; v < 128 * OS; v += 256) {
      // order: [run index within pixel][pixel][slot in run]
      const int sl = v & (rs - 1);
      const int p = (v >> rs_sh) & 127;
      const int ri = v >> (rs_sh + 7);
      if (ri >= runs_per_px) continue;
      const int m = m0 + p;
      if (m >= a.M) continue;
      const int slot = ri * rs + sl;              // slot inside the 128-channel tile
      const int cv = ct * CT + slot * 8;          // virtual channel
      const int tap = cv >= 2 * a.Cout ? (cv >= 3 * a.Cout ? 3 : 2) : (cv >= a.Cout ? 1 : 0);
      const int co = cv - tap * a.Cout;
      int rem = rem0 + p, n = n0;
      while (rem >= hw) { rem -= hw; ++n; }       // a 128-pixel tile crosses at most one sample boundary when H*W >= 128
      int y = (int)((float)rem * rcpW);
      int x = rem - y * a.W;
      if (x < 0) { --y; x += a.W; } else if (x >= a.W) { ++y; x -= a.W; }
      const size_t opix = ((size_t)n * 2 * a.H + 2 * y + (tap >> 1)) * (2 * a.W) + 2 * x + (tap & 1);
      const vec val = *reinterpret_cast<const vec*>(sO + p * (CT * 2) + ((slot ^ (p & (OS - 1))) << 4));
      UNCL_CHK(a.chk, a.out + opix * a.Cout + co, 16);
      *reinterpret_cast<vec*>(a.out + opix * a.Cout + co) = val;
    }
    if (PRE) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) B[ks] = Bn[ks];
    }
  }
}

template <typename T, int CIN, int CT>
int launch_up_ct(const UpArgs& a, bool prev, hipStream_t s) {
  constexpr size_t lds = (size_t)CT * CIN * 2 + 128 * CT * 2 + CT * 4;
  static UnclDevOnce attr_done[2];
  auto k0 = upconv2x2_kernel<T, CIN, false, CT>;
  auto k1 = upconv2x2_kernel<T, CIN, true, CT>;
  const void* kp = prev ? reinterpret_cast<const void*>(k1) : reinterpret_cast<const void*>(k0);
  if (attr_done[prev].need()) {
    if (hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return UNCL_ERR_LAUNCH;
    attr_done[prev].done();
  }
  const int n_ct = 4 * a.Cout / CT;
  // every workgroup stages its 128-row weight slice (up to 64 KB) before its first tile: a grid of one (CIN >= 128) or two
  // workgroups per resident slot, each walking a strided share of the tiles, instead of one workgroup per tile
  // (measured, 200 tiles: 12^2 x 256 level 105 -> 57 us, 28^2 x 128 106 -> 67 us, 61^2 x 64 158 -> 139 us)
  const int per_cu = (int)(160 * 1024 / lds) < 1 ? 1 : (int)(160 * 1024 / lds);
  int gx = (256 * per_cu * (CIN >= 128 ? 1 : 2)) / n_ct;
  if (gx < 1) gx = 1;
  if (gx > a.n_tiles) gx = a.n_tiles;
  if (prev)
    hipLaunchKernelGGL(k1, dim3(gx, n_ct), dim3(256), lds, s, a);
  else
    hipLaunchKernelGGL(k0, dim3(gx, n_ct), dim3(256), lds, s, a);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

template <typename T, int CIN>
int launch_up(const UpArgs& a, bool prev, hipStream_t s) {
  // narrow slices where the wide one leaves a CU with a single workgroup (and the output rows still split into whole runs)
  if (CIN >= UNCL_UP_NARROW_MINC && a.Cout % 64 == 0 && a.Cout >= 64) return launch_up_ct<T, CIN, 64>(a, prev, s);
  return launch_up_ct<T, CIN, 128>(a, prev, s);
}

}  // namespace

// x: (N,H,W,C) bf16; w: packed [4][Cout][C] bf16 (uncl_pack_conv_weight, transposed=1, flip=0); out: (N,2H,2W,Cout).
// prev / prev_ch: video recurrence (first prev_ch input channels come from `prev`, Unet.py:270).
extern "C" int uncl_upconv2x2(const void* x, const void* prev, int prev_ch, const void* w, const float* bias, void* out,
                              int N, int H, int W, int C, int Cout, void* stream) {
  return uncl_upconv2x2_dt(x, prev, prev_ch, w, bias, out, UNCL_BF16, N, H, W, C, Cout, stream);
}

// the same with the element type stated: UNCL_BF16 or UNCL_F16 (inference)
extern "C" int uncl_upconv2x2_dt(const void* x, const void* prev, int prev_ch, const void* w, const float* bias, void* out,
                                 int dtype, int N, int H, int W, int C, int Cout, void* stream) {
  if (!uncl_is_h16(dtype)) return UNCL_ERR_ARG;
  if (!x || !w || !out || N <= 0 || H <= 0 || W <= 0) return UNCL_ERR_ARG;
  if (Cout % 32 != 0 || (C != 32 && C != 64 && C != 128 && C != 256)) return UNCL_ERR_ARG;
  const long long M = (long long)N * H * W;
  if (M > 0x7fffffffLL / 256) return UNCL_ERR_ARG;
  UpArgs a;
  a.x = (const bf16_t*)x; a.prev = (const bf16_t*)prev; a.w = (const bf16_t*)w; a.bias = bias; a.out = (bf16_t*)out;
  a.H = H; a.W = W; a.C = C; a.Cout = Cout; a.prev_ch = prev_ch; a.M = (int)M; a.n_tiles = (int)((M + 127) / 128);
  const bool pv = prev != nullptr && prev_ch > 0;
#ifdef UNCL_CHECKED
  uncl_chk_reset(a.chk);
  uncl_chk_add(a.chk, x, (unsigned long long)M * C * 2);
  uncl_chk_add(a.chk, prev, (unsigned long long)M * C * 2);
  uncl_chk_add(a.chk, w, 4ull * Cout * C * 2);
  uncl_chk_add(a.chk, bias, (unsigned long long)Cout * 4);
  uncl_chk_add(a.chk, out, (unsigned long long)M * 4 * Cout * 2);
#endif
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == UNCL_F16) {
    switch (C) {
      case 32: return launch_up<f16_t, 32>(a, pv, s);
      case 64: return launch_up<f16_t, 64>(a, pv, s);
      case 128: return launch_up<f16_t, 128>(a, pv, s);
      default: return launch_up<f16_t, 256>(a, pv, s);
    }
  }
  switch (C) {
    case 32: return launch_up<bf16_t, 32>(a, pv, s);
    case 64: return launch_up<bf16_t, 64>(a, pv, s);
    case 128: return launch_up<bf16_t, 128>(a, pv, s);
    default: return launch_up<bf16_t, 256>(a, pv, s);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Data gradient of ConvTranspose2d(k2, s2):  g_x[n,y,x,ci] = sum_{dy,dx,co} g_up[n,2y+dy,2x+dx,co] * W[ci,co,dy,dx]
// Same skinny-GEMM structure as the forward with the roles of the channel dimensions swapped: K = Cout per tap (four
// taps accumulate into the same tile), weights re-packed as [tap][ci][co].  The stored gradient is multiplied by the
// activation derivative of the layer that produced x (mask > 0 ? 1 : slope) when a mask is given.
// ------------------------------------------------------------------------------------------------------------------
namespace {

struct UpBwdArgs {
  const bf16_t* gy;     // (N, 2H, 2W, Cout)
  const bf16_t* wt;     // [4][Cin][Cout]
  const bf16_t* mask;   // (N,H,W,Cin) or NULL
  bf16_t* gx;           // (N,H,W,Cin)
  int H, W, Cin, M, n_tiles, rows;  // rows = the Cin slice a workgroup handles: 32, 64 or 128 (<= Cin)
  float slope;
  // clips: the recurrent hand-off of the first pc channels (head_handoff_kernel, csrc/backward_kernels.hip) applied in the store,
  // before the mask: carries are (M, pc) bf16, either may be NULL
  const bf16_t* carry_in;
  bf16_t* carry_out;
  int pc;
};

template <int COUT>
__global__ __launch_bounds__(256) void upconv2x2_dgrad_kernel(const UpBwdArgs a) {
  using E = Elem<bf16_t>;
  using vec = bf16x8;
  constexpr int KS = COUT / 16, S = COUT / 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sW = smem;                        // [rows][COUT] bf16, swizzled
  char* sO = smem + 128 * COUT * 2;       // [128 pixels][rows] bf16
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int ct = blockIdx.y;              // `rows`-row slice of Cin
  const int NTI = a.rows / 32;            // N-tiles (ci) per workgroup: 1..4
  auto wswz = [](int row, int slot) {
    const int f = S == 4 ? ((row >> 2) & 3) : (S == 8 ? ((row >> 1) & 7) : (row & 15));
    return slot ^ f;
  };
  f32x16 zero16;
#pragma unroll
  for (int i = 0; i < 16; ++i) zero16[i] = 0.f;
  const int slots_o = a.rows / 8;         // 16-byte slots per output pixel

  for (int t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
    const int m0 = t * 128;
    const int mp = min(m0 + wave * 32 + lr, a.M - 1);
    const int n = mp / (a.H * a.W), rem = mp - n * (a.H * a.W);
    const int y = rem / a.W, x = rem - y * a.W;
    f32x16 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = zero16;
    for (int tap = 0; tap < 4; ++tap) {
      __syncthreads();  // previous tap's (or tile's) readers are done with sW / sO
      for (int v = tid; v < a.rows * S; v += 256) {
        const int row = v / S, slot = v - row * S;
        const vec wv = *reinterpret_cast<const vec*>(a.wt + ((size_t)tap * a.Cin + ct * a.rows + row) * COUT + slot * 8);
        *reinterpret_cast<vec*>(sW + row * (COUT * 2) + (wswz(row, slot) << 4)) = wv;
      }
      const size_t gp = (((size_t)n * 2 * a.H + 2 * y + (tap >> 1)) * (2 * a.W) + 2 * x + (tap & 1)) * COUT;
      vec B[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) B[ks] = *reinterpret_cast<const vec*>(a.gy + gp + ks * 16 + lh * 8);
      __syncthreads();
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          if (nt < NTI) {
            const int row = nt * 32 + lr;
            const vec A = *reinterpret_cast<const vec*>(sW + row * (COUT * 2) + (wswz(row, 2 * ks + lh) << 4));
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B[ks], acc[nt], 0, 0, 0);
          }
    }
    // transpose through LDS: [pixel][rows] bf16
    const int pl = wave * 32 + lr;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      if (nt < NTI) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          bf16x4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (bf16_t)acc[nt][4 * q + r];
          *reinterpret_cast<bf16x4*>(sO + pl * (a.rows * 2) + (((nt * 4 + q) ^ (pl & (slots_o - 1))) << 4) + (lh << 3)) = o;
        }
      }
    __syncthreads();
    for (int v = tid; v < 128 * slots_o; v += 256) {
      const int p = v / slots_o, sl = v - p * slots_o;
      const int m = m0 + p;
      if (m >= a.M) continue;
      vec val = *reinterpret_cast<const vec*>(sO + p * (a.rows * 2) + ((sl ^ (p & (slots_o - 1))) << 4));
      const size_t off = (size_t)m * a.Cin + ct * a.rows + sl * 8;
      const bool head = ct == 0 && sl == 0 && (a.carry_in != nullptr || a.carry_out != nullptr);
      if (a.mask != nullptr || head) {
        float f[8];
        E::unpack(val, f);
        if (head) {
#pragma unroll
          for (int i = 0; i < 8; ++i)
            if (i < a.pc) {
              float hx = f[i];
              if (a.carry_out) { a.carry_out[(size_t)m * a.pc + i] = (bf16_t)hx; hx = 0.f; }
              if (a.carry_in) hx += (float)a.carry_in[(size_t)m * a.pc + i];
              f[i] = hx;
            }
        }
        if (a.mask != nullptr) {
          float mk[8];
          E::unpack(*reinterpret_cast<const vec*>(a.mask + off), mk);
#pragma unroll
          for (int i = 0; i < 8; ++i) f[i] = mk[i] > 0.f ? f[i] : a.slope * f[i];
        }
        val = E::pack(f);
      }
      *reinterpret_cast<vec*>(a.gx + off) = val;
    }
  }
}

template <int COUT>
int launch_up_bwd(const UpBwdArgs& a, hipStream_t s) {
  constexpr size_t lds = (size_t)128 * COUT * 2 + 128 * 256;
  auto kern = upconv2x2_dgrad_kernel<COUT>;
  static UnclDevOnce attr_done;
  if (attr_done.need()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return UNCL_ERR_LAUNCH;
    attr_done.done();
  }
  const int n_ct = a.Cin / a.rows;
  const int gx = a.n_tiles < 2048 ? a.n_tiles : 2048;
  hipLaunchKernelGGL(kern, dim3(gx, n_ct), dim3(256), lds, s, a);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

}  // namespace

// gy: (N,2H,2W,Cout) bf16; wt: packed [4][Cin][Cout] bf16 (uncl_pack_conv_weight on the (Cin,Cout,2,2) weight with
// transposed=0, i.e. treating it as a Conv2d weight); gx: (N,H,W,Cin).  mask (optional): the activation x itself.
static int upconv2x2_dgrad_impl(const void* gy, const void* wt, const void* mask, float slope, void* gx, int N, int H, int W,
                                int Cin, int Cout, const void* carry_in, void* carry_out, int pc, void* stream);
extern "C" int uncl_upconv2x2_dgrad(const void* gy, const void* wt, const void* mask, float slope, void* gx, int N, int H, int W,
                                    int Cin, int Cout, void* stream) {
  return upconv2x2_dgrad_impl(gy, wt, mask, slope, gx, N, H, W, Cin, Cout, nullptr, nullptr, 0, stream);
}
// Internal (generator.hip, clips): the same launch with the head hand-off of the first pc (<= 8) channels folded into its store --
// carry_out receives this frame's head gradient, carry_in replaces it (either may be NULL), then the mask is applied, as
// bwd_head_handoff does after the plain launch
int bwd_upconv2x2_dgrad_handoff(const void* gy, const void* wt, const void* mask, float slope, void* gx, int N, int H, int W, int Cin,
                                int Cout, const void* carry_in, void* carry_out, int pc, void* stream) {
  if (pc < 0 || pc > 8) return UNCL_ERR_ARG;
  return upconv2x2_dgrad_impl(gy, wt, mask, slope, gx, N, H, W, Cin, Cout, carry_in, carry_out, pc, stream);
}
static int upconv2x2_dgrad_impl(const void* gy, const void* wt, const void* mask, float slope, void* gx, int N, int H, int W,
                                int Cin, int Cout, const void* carry_in, void* carry_out, int pc, void* stream) {
  if (!gy || !wt || !gx || N <= 0 || Cin % 32 != 0) return UNCL_ERR_ARG;
  if (Cout != 32 && Cout != 64 && Cout != 128 && Cout != 256) return UNCL_ERR_ARG;
  if (Cin > 128 && Cin % 128 != 0) return UNCL_ERR_ARG;
  const long long M = (long long)N * H * W;
  if (M > 0x7fffffffLL / 256) return UNCL_ERR_ARG;
  UpBwdArgs a;
  a.gy = (const bf16_t*)gy; a.wt = (const bf16_t*)wt; a.mask = (const bf16_t*)mask; a.gx = (bf16_t*)gx;
  a.H = H; a.W = W; a.Cin = Cin; a.M = (int)M; a.n_tiles = (int)((M + 127) / 128); a.rows = Cin < 128 ? Cin : 128; a.slope = slope;
  a.carry_in = (const bf16_t*)carry_in; a.carry_out = (bf16_t*)carry_out; a.pc = pc;
  // A workgroup stages its Cin slice of all four taps' weights (rows x Cout x 2 bytes each) one after the other before it multiplies:
  // on the 12 x 12 / 28 x 28 levels there are 18 - 72 workgroups of 128 rows and the launch takes 45 us whatever the batch (8 or 32
  // samples).  Narrower slices give proportionally more workgroups with proportionally shorter staging: down to 32 rows while the
  // launch has fewer than two workgroups per CU.
  static const int narrow_on = [] { const char* e = getenv("UNCL_UPBWD_NARROW"); return e ? atoi(e) : 1; }();
  while (narrow_on && a.rows > 32 && (long long)a.n_tiles * (Cin / a.rows) < 512) a.rows /= 2;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (Cout) {
    case 32: return launch_up_bwd<32>(a, s);
    case 64: return launch_up_bwd<64>(a, s);
    case 128: return launch_up_bwd<128>(a, s);
    default: return launch_up_bwd<256>(a, s);
  }
}
