// Weight-gradient GEMM for the 3x3 (and 1x1) convolutions, bf16 operands, fp32 accumulation, gfx950.
//
//   dW[tap][co][ci] = sum over output pixels p of  gY[p][co] * X[p + tap][ci]
//
// The reduction runs over PIXELS, which is the strided dimension of both NHWC operands.  Both tiles are staged
// into LDS in their natural [pixel][32 channels] order -- the 1x1 / single-row forms as 64-byte rows with XOR-swizzled
// 16-byte slots, the 3x3 form (wgrad3_kernel) as four PLANES of 16-byte slots (tr_frag_plane: one per-lane base, the tap /
// row offsets are immediates, no swizzle arithmetic in the inner loop) -- and read back with
// the transposing LDS read of CDNA4 (ds_read_b64_tr_b16): a 16-lane group fetches a 4-pixel x 16-channel block and
// each lane receives 4 consecutive pixels of ONE channel — exactly the K-contiguous fragment the 32x32x16 MFMA
// wants (A = gY^T: rows = co, k = pixel; B = X: k = pixel, cols = ci).  Lane mapping verified on hardware.
//
// Work split: blockIdx.y = vertical tap ty (three accumulators, tx = 0..2, per wave), blockIdx.z = (ci chunk,
// co chunk) pair, blockIdx.x = group of 16x32 pixel tiles walked persistently with register prefetch.  Waves split
// the tile's rows; partial sums are combined through LDS and added to the packed fp32 gradient with one float
// atomic per element per workgroup.  The input tile supports the same synthesised sources as the forward kernel
// (skip concat [x2, x1, x2^2, sqrt(x2+1e-8)] with replicate padding).
#include "common.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));

struct WgArgs {
  const bf16_t* src0;   // X source (skip x2 in concat mode)
  const bf16_t* src1;   // upsampled x1 in concat mode
  const bf16_t* gy;     // (N, Hout, Wout, Cout) gradient w.r.t. the conv output (already through the activation)
  float* dw;            // packed [taps][Cout][Cin] fp32, accumulated with atomics
  float* gb;            // 3x3 forms: bias gradient [Cout] (+= column sums of gy, float atomics) or NULL
  int H, W, Cin, Cout, pad, ks;  // ks = 3 or 1
  int s0H, s0W, s0C, s1H, s1W, s1C;
  int Hout, Wout;
  int tiles_x, tiles_y, total_tiles, tiles_per_wg, nci;
  long long M;          // ks == 1: number of valid flattened pixels (rows of 32)
  int gy_ld;            // elements between consecutive pixels of gy (>= Cout; grouped convs pass the full width)
  int groups;           // ks == 1, up_tap < 0: grouped 1x1 conv, blockIdx.y = group (channels g*Cin.. of X, g*Cout.. of gy)
  int up_tap, upH, upW; // ks == 1, up_tap >= 0: gy is the (N,2*upH,2*upW,Cout) output gradient of a 2x2 stride-2
                        // transposed conv and pixel m=(n,y,x) pairs with gy pixel (n, 2y+dy, 2x+dx)
  // Deterministic form (uncl_wgrad_set_scratch): instead of adding its sums into dw / gb with float atomics, pixel-range group
  // g = blockIdx.x STORES them at part[g * E + (the element's offset in dw)] / gb_part[g * Cout + co]; every (g, element) is
  // written by exactly one workgroup, and wgrad_reduce_kernel then adds the groups up in the order 0 .. G - 1.
  float* part;          // NULL: atomics
  float* gb_part;
  long long E;          // elements of dw this launch covers
  UNCL_CHK_MEMBER       // checked build: the tensors of this launch (common.h)
};

// partial sums of the G pixel-range groups -> out[i] += sum_g part[g * E + i] in a FIXED order: a block owns 32 elements, its
// eight 32-thread slices each add up one eighth of the groups (g ascending), the eight slice sums are added 0..7.  (One thread per
// element over all groups left 9216-element gradients with 36 workgroups and 512 dependent-latency loads each: 1.3 ms per step.)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int G, long long E, float* __restrict__ out) {
  __shared__ float sl[8][32];
  const int e = threadIdx.x & 31, q = threadIdx.x >> 5;
  const long long i = (long long)blockIdx.x * 32 + e;
  const int g0 = (int)((long long)G * q / 8), g1 = (int)((long long)G * (q + 1) / 8);
  float s = 0.f;
  if (i < E)
    for (int g = g0; g < g1; ++g) s += part[(size_t)g * E + i];
  sl[q][e] = s;
  __syncthreads();
  if (q == 0 && i < E) {
    float t = sl[0][e];
#pragma unroll
    for (int k = 1; k < 8; ++k) t += sl[k][e];
    out[i] += t;
  }
}

// accumulate into the gradient: atomics, or this group's slot of the partial buffer
__device__ __forceinline__ void wg_emit(const WgArgs& a, size_t off, float v) {
  if (a.part != nullptr) { UNCL_CHK(a.chk, a.part + (size_t)blockIdx.x * a.E + off, 4); a.part[(size_t)blockIdx.x * a.E + off] = v; }
  else { UNCL_CHK(a.chk, a.dw + off, 4); atomicAdd(a.dw + off, v); }
}

__device__ __forceinline__ bf16x8 ld16g(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
// wave-uniform base + 32-bit per-lane BYTE offset (global_load_dwordx4 v, v_off, s[base]); the empty asm keeps the zero-extension
// next to the load (conv3x3_args.h: ld16o)
__device__ __forceinline__ bf16x8 wg_ld16o(const bf16_t* base, unsigned byte_off) {
  asm volatile("" : "+v"(byte_off));
  return *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(base) + byte_off);
}

// checked build: the same loads with the address looked up in the launch's tensor table first (`a` = the kernel's WgArgs)
#define WG_LD16O(base, off) (UNCL_CHK(a.chk, reinterpret_cast<const char*>(base) + (off), 16), wg_ld16o((base), (off)))
#define WG_LD16G(p) (UNCL_CHK(a.chk, (p), 16), ld16g(p))
#define WG_LDV(p) (UNCL_CHK(a.chk, (p), 16), *reinterpret_cast<const vec*>(p))

__device__ __forceinline__ bf16x8 tr_frag(const char* lds_row0_base, int pix0, int c0, int lane) {
  // 8 consecutive pixels (pix0 + 8h' ... handled by caller) x one channel per lane: two 4-pixel transposed reads
  const int li = lane & 15, q = li >> 2, pp = li & 3;
  bf16x8 out;
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) {
    const int p = pix0 + 4 * rd + q;
    const int slot = (c0 >> 3) + (pp >> 1);
    const char* addr = lds_row0_base + p * 64 + ((slot ^ ((p >> 2) & 3)) << 4) + ((pp & 1) << 3);
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)addr);
#pragma unroll
    for (int i = 0; i < 4; ++i) out[4 * rd + i] = __builtin_bit_cast(bf16_t, (short)v[i]);
  }
  return out;
}

// The same fragment from a PLANE layout (plane s = channels 8s .. 8s+7 of every pixel, 16 bytes per pixel, planes padded to
// 4 (mod 16) slots so that the four planes a 16-lane group touches sit in distinct bank ranges): the address is a per-lane base
// (computed once per kernel) plus a compile-time pixel offset -- no per-read address arithmetic.  The swizzled layout above
// costs ~8 vector instructions per transposed read (the XOR depends on the pixel), 8 per MFMA in the 3x3 kernel's inner loop,
// on the issue port the MFMAs share.
__device__ __forceinline__ bf16x8 tr_frag_plane(const char* lane_base, int pix_off_bytes) {
  bf16x8 out;
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) {
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lane_base + pix_off_bytes + rd * 64));
#pragma unroll
    for (int i = 0; i < 4; ++i) out[4 * rd + i] = __builtin_bit_cast(bf16_t, (short)v[i]);
  }
  return out;
}
// Plane lengths stay 4 (mod 16) slots = a stride of 16 banks (mod 64): the four planes a 32-lane group of ds_read_b64_tr_b16 touches
// then tile the 64 READ banks exactly.  WRITES are banked mod 32, so with "four lanes = the four planes of a pixel" every staging
// write here is a two-way conflict (SQ_LDS_BANK_CONFLICT 40 - 50 % of SQ_LDS_IDX_ACTIVE).  Measured and left alone: mapping
// the staging lanes so that eight consecutive lanes write one plane removes the conflicts (40 % -> 3 %, 40 % fewer LDS cycles) but
// the same lanes then load eight different pixels' 16-byte slots, and these kernels are bound by the L2 -> L1 path, not by LDS:
// launches +4 %, step +0.04 ms.  No plane stride serves both the reads (16 or 48 mod 64) and the writes (8 or 24 mod 32).
constexpr int wg_plane(int rows) { return (rows + ((4 - rows % 16) + 16) % 16) * 16; }

// MODE: 0 plain, 1 concat-ssr.  KS: 3 (three horizontal taps per workgroup) or 1
template <int MODE, int KS>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgArgs a) {
  using E = Elem<bf16_t>;
  using vec = bf16x8;
  constexpr int TH = 16, TW = 32;
  constexpr int XW = TW + KS - 1;            // staged input columns
  constexpr int NX = TH * XW;                // staged input pixels (one vertical tap: no vertical halo)
  constexpr int NG = TH * TW;
  constexpr int XV = (NX * 4 + 255) / 256, GV = (NG * 4) / 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sX = smem;
  char* sG = smem + NX * 64;
  float* sR = reinterpret_cast<float*>(smem);  // [KS][32][32] cross-wave reduction (after the loop)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // KS == 3: blockIdx.y = vertical tap; KS == 1 on the 2x2 stride-2 transposed conv: blockIdx.y = its tap (dy, dx)
  const int ty = KS == 3 ? (int)blockIdx.y : 0;
  const int up_tap = (KS == 1 && a.up_tap >= 0) ? (int)blockIdx.y : -1;
  const int cgrp = (KS == 1 && a.up_tap < 0) ? (int)blockIdx.y : 0;     // grouped 1x1: this workgroup's group
  const int kc = blockIdx.z % a.nci, cc = blockIdx.z / a.nci;  // input / output channel chunk
  int tile = (int)blockIdx.x * a.tiles_per_wg;
  const int tile_end = min(tile + a.tiles_per_wg, a.total_tiles);
  if (tile >= tile_end) return;

  int g = 0, cbase = kc * 32;
  if (MODE != 0) { g = cbase / a.s0C; cbase -= g * a.s0C; }
  const int p0 = tid >> 2, ch = tid & 3;

  vec xr[XV], gr[GV];
  unsigned xvalid = 0, gvalid = 0;

  auto load_tile = [&](int t) {
    int r = t;
    const int tx_ = r % a.tiles_x; r /= a.tiles_x;
    const int ty_ = r % a.tiles_y; r /= a.tiles_y;
    const int n = r, y0 = ty_ * TH, x0 = tx_ * TW;
    const int iy0 = y0 + ty - a.pad, ix0 = x0 - a.pad;
    unsigned valid = 0;
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int pix = p0 + j * 64;
      const int hy = pix / XW, hx = pix - hy * XW;
      const int iy = iy0 + hy, ix = ix0 + hx;
      bool ok = pix < NX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      if (KS == 1) ok = ok && ((long long)iy * 32 + ix < a.M);
      valid |= (ok ? 1u : 0u) << j;
      if (MODE != 0 && g == 1) {
        const int dy = (a.s0H - a.s1H) >> 1, dx = (a.s0W - a.s1W) >> 1;
        const int sy = min(max(iy - dy, 0), a.s1H - 1), sx = min(max(ix - dx, 0), a.s1W - 1);
        xr[j] = WG_LD16G(a.src1 + ((size_t)n * a.s1H * a.s1W + (size_t)sy * a.s1W + sx) * a.s1C + cbase + ch * 8);
      } else {
        const size_t off = ok ? ((size_t)n * a.s0H * a.s0W + (size_t)iy * a.s0W + ix) * a.s0C : 0;
        xr[j] = WG_LD16G(a.src0 + off + cgrp * a.Cin + cbase + ch * 8);
      }
    }
    xvalid = valid;
    unsigned gval = 0;
#pragma unroll
    for (int j = 0; j < GV; ++j) {
      const int pix = p0 + j * 64;
      const int gy_ = y0 + pix / TW, gx_ = x0 + pix % TW;
      bool ok = gy_ < a.Hout && gx_ < a.Wout;
      if (KS == 1) ok = ok && ((long long)gy_ * 32 + gx_ < a.M);
      size_t off = ok ? ((size_t)n * a.Hout * a.Wout + (size_t)gy_ * a.Wout + gx_) * a.gy_ld : 0;
      if (KS == 1 && up_tap >= 0 && ok) {
        const long long m = (long long)gy_ * 32 + gx_;
        const int hw = a.upH * a.upW;
        const int nn = (int)(m / hw), rem = (int)(m - (long long)nn * hw);
        const int yy = rem / a.upW, xx = rem - yy * a.upW;
        off = (((size_t)nn * 2 * a.upH + 2 * yy + (up_tap >> 1)) * (2 * a.upW) + 2 * xx + (up_tap & 1)) * a.gy_ld;
      }
      // (zeroed at write time: a select on the value just requested makes the request wait for its own data, i.e. the prefetch
      // of the next tile would finish before the first MFMA of this one)
      gval |= (ok ? 1u : 0u) << j;
      gr[j] = WG_LD16G(a.gy + off + cgrp * a.Cout + cc * 32 + ch * 8);
    }
    gvalid = gval;
  };
  auto write_lds = [&]() {
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int pix = p0 + j * 64;
      if (pix >= NX) continue;
      vec v = xr[j];
      if (MODE == 1 && g >= 2) {
        float f[8];
        E::unpack(v, f);
        if (g == 2) {
#pragma unroll
          for (int i = 0; i < 8; ++i) f[i] = f[i] * f[i];
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) f[i] = __builtin_amdgcn_sqrtf(f[i] + 1e-8f);
        }
        v = E::pack(f);
      }
      if (!((xvalid >> j) & 1u)) v = E::zero();
      *reinterpret_cast<vec*>(sX + pix * 64 + ((ch ^ ((pix >> 2) & 3)) << 4)) = v;
    }
#pragma unroll
    for (int j = 0; j < GV; ++j) {
      const int pix = p0 + j * 64;
      *reinterpret_cast<vec*>(sG + pix * 64 + ((ch ^ ((pix >> 2) & 3)) << 4)) = ((gvalid >> j) & 1u) ? gr[j] : E::zero();
    }
  };

  f32x16 acc[KS];
#pragma unroll
  for (int t = 0; t < KS; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  const int grp = lane >> 4;                  // 16-lane group
  const int c0 = 16 * (grp & 1), kb = 8 * (grp >> 1);

  load_tile(tile);
  write_lds();
  __syncthreads();
  while (true) {
    const bool more = tile + 1 < tile_end;
    if (more) load_tile(tile + 1);
    // this wave's rows: 4*wave .. 4*wave+3; k-steps of 16 pixels (half rows)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wave * 4 + r;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const vec A = tr_frag(sG, row * TW + half * 16 + kb, c0, lane);
#pragma unroll
        for (int tx = 0; tx < KS; ++tx) {
          const vec B = tr_frag(sX, row * XW + half * 16 + tx + kb, c0, lane);
          acc[tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[tx], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (!more) break;
    write_lds();
    __syncthreads();
    ++tile;
  }
  // ---- combine the four waves' partial sums through LDS in a FIXED order (one slab per wave, summed 0..3), then one atomic
  // (or one partial-buffer store) per element
  const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int tx = 0; tx < KS; ++tx)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int co = (i & 3) + 8 * (i >> 2) + 4 * lh;  // accumulator row
      sR[wave * KS * 1024 + (tx * 32 + co) * 32 + lr] = acc[tx][i];
    }
  __syncthreads();
  for (int i = tid; i < KS * 1024; i += 256) {
    const int tx = i >> 10, co = (i >> 5) & 31, ci = i & 31;
    const int tap = KS == 3 ? ty * 3 + tx : (up_tap >= 0 ? up_tap : 0);
    const float v = ((sR[i] + sR[KS * 1024 + i]) + sR[2 * KS * 1024 + i]) + sR[3 * KS * 1024 + i];
    wg_emit(a, (size_t)cgrp * a.Cout * a.Cin + ((size_t)tap * a.Cout + cc * 32 + co) * a.Cin + kc * 32 + ci, v);
  }
}

// 3x3 form: ONE workgroup covers all nine taps of its (ci chunk, co chunk) pair.  Six waves = (vertical tap ty) x (upper /
// lower eight rows of the 16 x 32 pixel tile), three accumulators (tx) each; the input tile carries its two halo rows, so a
// tile pass moves 39 KB of X and 32 KB of gY for 288 MFMAs where three single-ty workgroups moved 3 x 67 KB.  The kernel is
// bound by what it pulls through L2 -> L1 (every pair re-streams both tensors), so the bytes per MFMA are its speed.
template <int MODE>
__global__ __launch_bounds__(384, 3) void wgrad3_kernel(const WgArgs a) {
  using E = Elem<bf16_t>;
  using vec = bf16x8;
  constexpr int TH = 16, TW = 32, XH = TH + 2, XW = TW + 2;
  constexpr int NX = XH * XW, NG = TH * TW, NT = 384, PP = NT / 4;   // PP pixels staged per pass
  constexpr int XV = (NX + PP - 1) / PP, GV = (NG + PP - 1) / PP;
  constexpr int XPLB = wg_plane(NX), GPLB = wg_plane(NG);     // bytes per plane
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sX = smem;
  char* sG = smem + 4 * XPLB;
  float* sR = reinterpret_cast<float*>(smem);  // [9][32][32] cross-wave reduction (after the loop)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ty = wave >> 1, rh = wave & 1;
  const int kc = blockIdx.z % a.nci, cc = blockIdx.z / a.nci;
  int tile = (int)blockIdx.x * a.tiles_per_wg;
  const int tile_end = min(tile + a.tiles_per_wg, a.total_tiles);
  if (tile >= tile_end) return;

  int g = 0, cbase = kc * 32;
  if (MODE != 0) { g = cbase / a.s0C; cbase -= g * a.s0C; }
  vec xr[XV], gr[GV];
  unsigned xvalid = 0, gvalid = 0;
  // bias gradient = column sums of gy: the workgroups of the first ci chunk add up the gy vectors they stage anyway (the
  // vector pipe is idle beside the MFMAs of the other waves); summed over the staging threads and added with 32 / 64 atomics
  const bool do_bias = a.gb != nullptr && kc == 0;      // workgroup-uniform
  float bs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) bs[i] = 0.f;

  auto load_tile = [&](int t) {
    int r = t;
    const int tx_ = r % a.tiles_x; r /= a.tiles_x;
    const int ty_ = r % a.tiles_y; r /= a.tiles_y;
    const int n = r, y0 = ty_ * TH, x0 = tx_ * TW;
    const int iy0 = y0 - a.pad, ix0 = x0 - a.pad;
    // the per-thread staging pattern is recomputed per tile (a handful of VALU ops) instead of living in ~30 registers
    int t4 = tid;
    asm volatile("" : "+v"(t4));
    const int p0 = t4 >> 2, ch = t4 & 3;
    unsigned valid = 0;
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int pix = min(p0 + j * PP, NX - 1);
      const int hy = pix / XW, hx = pix - hy * XW;
      const int iy = iy0 + hy, ix = ix0 + hx;
      const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      valid |= (ok ? 1u : 0u) << j;
      // wave-uniform sample base + 32-bit per-lane offset: one address register per load
      if (MODE != 0 && g == 1) {
        const int dy = (a.s0H - a.s1H) >> 1, dx = (a.s0W - a.s1W) >> 1;
        const int sy = min(max(iy - dy, 0), a.s1H - 1), sx = min(max(ix - dx, 0), a.s1W - 1);
        const bf16_t* base = a.src1 + (size_t)n * a.s1H * a.s1W * a.s1C + cbase;
        xr[j] = WG_LDV(base + (unsigned)((sy * a.s1W + sx) * a.s1C + ch * 8));
      } else {
        const bf16_t* base = a.src0 + (size_t)n * a.s0H * a.s0W * a.s0C + cbase;
        const unsigned off = ok ? (unsigned)((iy * a.s0W + ix) * a.s0C + ch * 8) : 0u;
        xr[j] = WG_LDV(base + off);
      }
    }
    xvalid = valid;
    unsigned gval = 0;
#pragma unroll
    for (int j = 0; j < GV; ++j) {
      const int pix = min(p0 + j * PP, NG - 1);
      const int gy_ = y0 + pix / TW, gx_ = x0 + pix % TW;
      const bool ok = gy_ < a.Hout && gx_ < a.Wout;
      const bf16_t* base = a.gy + (size_t)n * a.Hout * a.Wout * a.gy_ld + cc * 32;
      const unsigned off = ok ? (unsigned)((gy_ * a.Wout + gx_) * a.gy_ld + ch * 8) : 0u;
      gval |= (ok ? 1u : 0u) << j;          // (zeroed at write time, see wgrad_kernel)
      gr[j] = WG_LDV(base + off);
    }
    gvalid = gval;
  };
  auto write_lds = [&]() {
    int t4 = tid;
    asm volatile("" : "+v"(t4));
    const int p0 = t4 >> 2, ch = t4 & 3;
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int pix = p0 + j * PP;
      if (pix >= NX) continue;
      vec v = xr[j];
      if (MODE == 1 && g >= 2) {
        float f[8];
        E::unpack(v, f);
        if (g == 2) {
#pragma unroll
          for (int i = 0; i < 8; ++i) f[i] = f[i] * f[i];
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) f[i] = __builtin_amdgcn_sqrtf(f[i] + 1e-8f);
        }
        v = E::pack(f);
      }
      if (!((xvalid >> j) & 1u)) v = E::zero();
      *reinterpret_cast<vec*>(sX + ch * XPLB + pix * 16) = v;
    }
#pragma unroll
    for (int j = 0; j < GV; ++j) {
      const int pix = p0 + j * PP;
      if (pix >= NG) continue;
      const vec gv = ((gvalid >> j) & 1u) ? gr[j] : E::zero();
      *reinterpret_cast<vec*>(sG + ch * GPLB + pix * 16) = gv;
      if (do_bias) {
        float f[8];
        E::unpack(gv, f);
#pragma unroll
        for (int i = 0; i < 8; ++i) bs[i] += f[i];
      }
    }
  };

  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  const int grp = lane >> 4;
  const int c0 = 16 * (grp & 1), kb = 8 * (grp >> 1);
  // per-lane fragment bases (see tr_frag_plane): 16-lane group -> channels c0.., pixels kb..; lane li -> pixel li >> 2 of the
  // 4-pixel block, 16-byte slot (li & 3) >> 1, its lower / upper 8 bytes
  const int li = lane & 15;
  const int frag_lane = (li >> 2) * 16 + ((li & 1) << 3);
  const char* lbG = sG + ((c0 >> 3) + ((li & 3) >> 1)) * GPLB + (rh * 8 * TW + kb) * 16 + frag_lane;
  const char* lbX = sX + ((c0 >> 3) + ((li & 3) >> 1)) * XPLB + ((rh * 8 + ty) * XW + kb) * 16 + frag_lane;

  load_tile(tile);
  write_lds();
  __syncthreads();
  while (true) {
    const bool more = tile + 1 < tile_end;
    if (more) load_tile(tile + 1);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const vec A = tr_frag_plane(lbG, (r * TW + half * 16) * 16);
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
          const vec B = tr_frag_plane(lbX, (r * XW + half * 16 + tx) * 16);
          acc[tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[tx], 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (!more) break;
    write_lds();
    __syncthreads();
    ++tile;
  }
  if (do_bias) {
    // (the tile loop ended with a barrier: the staging area is free) [thread][8] partial sums -> 32 channels
    float* sBs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 8; ++i) sBs[tid * 8 + i] = bs[i];
    __syncthreads();
    if (tid < 32) {
      const int sl = tid >> 3, e = tid & 7;
      float t = 0.f;
      for (int p = 0; p < NT / 4; ++p) t += sBs[(p * 4 + sl) * 8 + e];
      if (a.gb_part != nullptr) { UNCL_CHK(a.chk, a.gb_part + (size_t)blockIdx.x * a.Cout + cc * 32 + tid, 4); a.gb_part[(size_t)blockIdx.x * a.Cout + cc * 32 + tid] = t; }
      else { UNCL_CHK(a.chk, a.gb + cc * 32 + tid, 4); atomicAdd(a.gb + cc * 32 + tid, t); }
    }
    __syncthreads();
  }
  // ---- the two row halves of every tap are summed through LDS (fixed order), then one atomic per element
  const int lr = lane & 31, lh = lane >> 5;
  if (rh == 0) {
#pragma unroll
    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = (i & 3) + 8 * (i >> 2) + 4 * lh;
        sR[((ty * 3 + tx) * 32 + co) * 32 + lr] = acc[tx][i];
      }
  }
  __syncthreads();
  if (rh == 1) {
#pragma unroll
    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = (i & 3) + 8 * (i >> 2) + 4 * lh;
        sR[((ty * 3 + tx) * 32 + co) * 32 + lr] += acc[tx][i];
      }
  }
  __syncthreads();
#if defined(UNCL_WG_ABLATE_ATOMICS)      // timing ablation only (wrong gradients): what the atomic tail costs
  if (a.dw == nullptr)
#endif
  for (int i = tid; i < 9 * 1024; i += NT) {
    const int tap = i >> 10, co = (i >> 5) & 31, ci = i & 31;
    wg_emit(a, ((size_t)tap * a.Cout + cc * 32 + co) * a.Cin + kc * 32 + ci, sR[i]);
  }
}

// 3x3 form for layers with Cin and Cout multiples of 64: ONE workgroup covers a 64 (ci) x 64 (co) block of all nine taps over
// 8 x 32 pixel tiles.  Why: a 32-channel chunk is 64 bytes of a pixel, so the 32 x 32 kernel above pulls HALF of every 128-byte
// line it touches through L2 -> L1 (the other half goes to another workgroup), and moves 71 KB per 288 MFMAs; its launches sit
// at 8 - 20 % of the matrix peak with a 35 - 45 us floor on the small maps.  Here a staging thread group reads whole lines
// (8 x 16 bytes = 64 channels of a pixel) and a tile pass moves 75 KB (X: 10 x 34 halo pixels, gY: 8 x 32) for 576 MFMAs.
// Six waves = (vertical tap ty) x (ci half); each holds the 2 (co halves) x 3 (tx) accumulators of its taps for the WHOLE pixel
// range of the workgroup, so there is no cross-wave reduction: the sums go from the registers to the packed gradient with one
// float atomic per element.  Two workgroups per CU overlap each other's staging.
template <int MODE>
__global__ __launch_bounds__(384, 3) void wgrad3w_kernel(const WgArgs a) {
  using E = Elem<bf16_t>;
  using vec = bf16x8;
  constexpr int TH = 8, TW = 32, XH = TH + 2, XW = TW + 2;
  constexpr int NX = XH * XW, NG = TH * TW, NT = 384, PP = NT / 8;   // PP pixels (x 8 slots of 16 bytes) staged per pass
  constexpr int XV = (NX + PP - 1) / PP, GV = (NG + PP - 1) / PP;
  constexpr int XPLB = wg_plane(NX), GPLB = wg_plane(NG);     // bytes per plane (8 planes per tile: 64 channels)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sX = smem;
  char* sG = smem + 8 * XPLB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ty = wave >> 1, cih = wave & 1;
  const int nci = a.Cin >> 6;
  const int kc = blockIdx.z % nci, cc = blockIdx.z / nci;
  int tile = (int)blockIdx.x * a.tiles_per_wg;
  const int tile_end = min(tile + a.tiles_per_wg, a.total_tiles);
  if (tile >= tile_end) return;

  int g = 0, cbase = kc * 64;
  if (MODE != 0) { g = cbase / a.s0C; cbase -= g * a.s0C; }
  vec xr[XV], gr[GV];
  unsigned xvalid = 0;

  auto load_tile = [&](int t) {
    int r = t;
    const int tx_ = r % a.tiles_x; r /= a.tiles_x;
    const int ty_ = r % a.tiles_y; r /= a.tiles_y;
    const int n = r, y0 = ty_ * TH, x0 = tx_ * TW;
    const int iy0 = y0 - a.pad, ix0 = x0 - a.pad;
    int t8 = tid;
    asm volatile("" : "+v"(t8));
    const int p0 = t8 >> 3, ch = t8 & 7;
    unsigned valid = 0;
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int pix = min(p0 + j * PP, NX - 1);
      const int hy = pix / XW, hx = pix - hy * XW;
      const int iy = iy0 + hy, ix = ix0 + hx;
      const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      valid |= (ok ? 1u : 0u) << j;
      if (MODE != 0 && g == 1) {
        const int dy = (a.s0H - a.s1H) >> 1, dx = (a.s0W - a.s1W) >> 1;
        const int sy = min(max(iy - dy, 0), a.s1H - 1), sx = min(max(ix - dx, 0), a.s1W - 1);
        const bf16_t* base = a.src1 + (size_t)n * a.s1H * a.s1W * a.s1C + cbase;
        xr[j] = WG_LDV(base + (unsigned)((sy * a.s1W + sx) * a.s1C + ch * 8));
      } else {
        const bf16_t* base = a.src0 + (size_t)n * a.s0H * a.s0W * a.s0C + cbase;
        const unsigned off = ok ? (unsigned)((iy * a.s0W + ix) * a.s0C + ch * 8) : 0u;
        xr[j] = WG_LDV(base + off);
      }
    }
    xvalid = valid;
#pragma unroll
    for (int j = 0; j < GV; ++j) {
      const int pix = min(p0 + j * PP, NG - 1);
      const int gy_ = y0 + pix / TW, gx_ = x0 + pix % TW;
      const bool ok = gy_ < a.Hout && gx_ < a.Wout;
      const bf16_t* base = a.gy + (size_t)n * a.Hout * a.Wout * a.gy_ld + cc * 64;
      const unsigned off = ok ? (unsigned)((gy_ * a.Wout + gx_) * a.gy_ld + ch * 8) : 0u;
      // (the other kernels zero at write time so that the request does not wait for its own data; this one is at its 168-register
      // budget -- the concat instantiation already spills 7 -- and off the default paths since wgrad3c_kernel: left as measured)
      vec v = WG_LDV(base + off);
      if (!ok) v = E::zero();
      gr[j] = v;
    }
  };
  auto write_lds = [&]() {
    int t8 = tid;
    asm volatile("" : "+v"(t8));
    const int p0 = t8 >> 3, ch = t8 & 7;
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int pix = p0 + j * PP;
      if (pix >= NX) continue;
      vec v = xr[j];
      if (MODE == 1 && g >= 2) {
        float f[8];
        E::unpack(v, f);
        if (g == 2) {
#pragma unroll
          for (int i = 0; i < 8; ++i) f[i] = f[i] * f[i];
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) f[i] = __builtin_amdgcn_sqrtf(f[i] + 1e-8f);
        }
        v = E::pack(f);
      }
      if (!((xvalid >> j) & 1u)) v = E::zero();
      *reinterpret_cast<vec*>(sX + ch * XPLB + pix * 16) = v;
    }
#pragma unroll
    for (int j = 0; j < GV; ++j) {
      const int pix = p0 + j * PP;
      if (pix >= NG) continue;
      *reinterpret_cast<vec*>(sG + ch * GPLB + pix * 16) = gr[j];
    }
  };

  f32x16 acc[2][3];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[h][t][i] = 0.f;

  // per-lane fragment bases, as in wgrad3_kernel (tr_frag_plane); the X base carries this wave's ci half and vertical tap
  const int grp = lane >> 4;
  const int c0 = 16 * (grp & 1), kb = 8 * (grp >> 1);
  const int li = lane & 15;
  const int frag_lane = (li >> 2) * 16 + ((li & 1) << 3);
  const char* lbG = sG + ((c0 >> 3) + ((li & 3) >> 1)) * GPLB + kb * 16 + frag_lane;
  const char* lbX = sX + (4 * cih + (c0 >> 3) + ((li & 3) >> 1)) * XPLB + (ty * XW + kb) * 16 + frag_lane;

  // No register prefetch of the next tile: 14 more vectors per thread on top of the 96 accumulator registers do not fit three
  // waves per SIMD without spilling; the CU's other workgroup multiplies while this one waits for its loads.
  for (; tile < tile_end; ++tile) {
    load_tile(tile);
    write_lds();
    __syncthreads();
#pragma unroll
    for (int r = 0; r < TH; ++r) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        vec B[3];
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) B[tx] = tr_frag_plane(lbX, (r * XW + half * 16 + tx) * 16);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const vec A = tr_frag_plane(lbG + 4 * h * GPLB, (r * TW + half * 16) * 16);
#pragma unroll
          for (int tx = 0; tx < 3; ++tx) acc[h][tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B[tx], acc[h][tx], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  // accumulator (h, tx): rows = co (cc*64 + 32 h + ...), columns = ci (kc*64 + 32 cih + lane & 31): 128 contiguous bytes per row
  const int lr = lane & 31, lh = lane >> 5;
#if defined(UNCL_WG_ABLATE_ATOMICS)
  if (a.dw == nullptr)
#endif
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = cc * 64 + 32 * h + (i & 3) + 8 * (i >> 2) + 4 * lh;
        wg_emit(a, ((size_t)(ty * 3 + tx) * a.Cout + co) * a.Cin + kc * 64 + 32 * cih + lr, acc[h][tx][i]);
      }
}

// ------------------------------------------------------------------------------------------------------------------
// 3x3 form with split roles (round 4): 512 threads, ONE workgroup per CU, a two-stage LDS ring.
//
//   waves 0-3  multiply: wave w owns output rows 4w .. 4w+3 of the 16 x 32 pixel tile and ALL NINE taps of the (ci chunk, co
//              chunk) pair -- nine 32 x 32 accumulators.  The X fragments of halo row R serve output rows R, R-1, R-2 (vertical
//              taps 0, 1, 2), so a wave walks its six halo rows once per 16-pixel half: 18 X + 4 gY transposed fragment reads for
//              36 MFMAs (0.61 reads per MFMA; wgrad3_kernel above, whose waves own ONE vertical tap each, re-reads every halo row
//              for each tap: 1.33 per MFMA, which is more than the LDS delivers at the matrix rate).
//   waves 4-7  stage: global loads of tile t+2 into registers, registers of tile t+1 (square / square-root of the skip slice,
//              zero padding, the bias sums of gY) into the stage the multiplying waves are not reading.
//
// One barrier per tile.  The four row blocks are added up through LDS once, after the workgroup's last tile (fixed order), then
// one float atomic (or one partial-buffer store) per element as in the other forms.  Same operands, same products; only the
// order of the fp32 additions differs from wgrad3_kernel.
template <int MODE>
__global__ __launch_bounds__(512, 1) void wgrad3r_kernel(const WgArgs a) {
  using E = Elem<bf16_t>;
  using vec = bf16x8;
  constexpr int TH = 16, TW = 32, XH = TH + 2, XW = TW + 2;
  constexpr int NX = XH * XW, NG = TH * TW, NS = 256, PP = NS / 4;   // staging threads; PP pixels staged per pass
  constexpr int XV = (NX + PP - 1) / PP, GV = (NG + PP - 1) / PP;
  constexpr int XPLB = wg_plane(NX), GPLB = wg_plane(NG);     // bytes per plane
  constexpr int STAGE = 4 * XPLB + 4 * GPLB;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kc = blockIdx.z % a.nci, cc = blockIdx.z / a.nci;
  const int tile0 = (int)blockIdx.x * a.tiles_per_wg;
  const int tile_end = min(tile0 + a.tiles_per_wg, a.total_tiles);
  if (tile0 >= tile_end) return;
  const bool do_bias = a.gb != nullptr && kc == 0;      // workgroup-uniform
  float* const sR = reinterpret_cast<float*>(smem);                 // [9][32][32] after the loop
  float* const sBs = reinterpret_cast<float*>(smem + 9 * 1024 * 4); // [256][8] after the loop

  f32x16 acc[3][3];
  if (wave < 4) {
    // ================================================================================================================
    // multiplying waves
    // ================================================================================================================
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
      for (int tx = 0; tx < 3; ++tx)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[ty][tx][i] = 0.f;
    // per-lane fragment bases (tr_frag_plane): 16-lane group -> channels c0.., pixels kb..; lane li -> pixel li >> 2 of the
    // 4-pixel block, 16-byte slot (li & 3) >> 1, its lower / upper 8 bytes
    const int grp = lane >> 4;
    const int c0 = 16 * (grp & 1), kb = 8 * (grp >> 1);
    const int li = lane & 15;
    const int frag_lane = (li >> 2) * 16 + ((li & 1) << 3);
    const int psel = (c0 >> 3) + ((li & 3) >> 1);
    const int offX = psel * XPLB + (4 * wave * XW + kb) * 16 + frag_lane;
    const int offG = 4 * XPLB + psel * GPLB + (4 * wave * TW + kb) * 16 + frag_lane;
    __syncthreads();                 // stage 0 holds the first tile
    for (int t = tile0; t < tile_end; ++t) {
      const char* st = smem + ((t - tile0) & 1) * STAGE;
      const char* lbX = st + offX;
      const char* lbG = st + offG;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        vec A[4], B[2][3];
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) B[0][tx] = tr_frag_plane(lbX, (h * 16 + tx) * 16);
#pragma unroll
        for (int R = 0; R < 6; ++R) {
          if (R < 5) {
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) B[(R + 1) & 1][tx] = tr_frag_plane(lbX, ((R + 1) * XW + h * 16 + tx) * 16);
          }
          if (R < 4) A[R] = tr_frag_plane(lbG, (R * TW + h * 16) * 16);
#pragma unroll
          for (int ty = 0; ty < 3; ++ty) {
            const int r = R - ty;                    // output row (of this wave's four) that halo row R feeds through tap row ty
            if (r < 0 || r > 3) continue;
#pragma unroll
            for (int tx = 0; tx < 3; ++tx)
              acc[ty][tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[r], B[R & 1][tx], acc[ty][tx], 0, 0, 0);
          }
        }
      }
      __syncthreads();               // done with this stage; the other one holds the next tile
    }
  } else {
    // ================================================================================================================
    // staging waves
    // ================================================================================================================
    const int ptid = tid - 256;
    int g = 0, cbase = kc * 32;
    if (MODE != 0) { g = cbase / a.s0C; cbase -= g * a.s0C; }
    vec xr[XV], gr[GV];
    unsigned xvalid = 0, gvalid = 0;
    float bs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) bs[i] = 0.f;
    // per-thread constants of the staging pattern (the same for every tile), loads as "wave-uniform sample base + 32-bit lane
    // offset" (see wgrad3c_kernel)
    const int p0c = ptid >> 2, chc = ptid & 3;
    int hyj[XV], hxj[XV], gyj[GV], gxj[GV];
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int pix = min(p0c + j * PP, NX - 1);
      hyj[j] = pix / XW; hxj[j] = pix - hyj[j] * XW;
    }
#pragma unroll
    for (int j = 0; j < GV; ++j) {
      const int pix = min(p0c + j * PP, NG - 1);
      gyj[j] = pix / TW; gxj[j] = pix % TW;
    }
    const int dy1 = (a.s0H - a.s1H) >> 1, dx1 = (a.s0W - a.s1W) >> 1;
    auto load_tile = [&](int t) __attribute__((always_inline)) {
      int r = t;
      const int tx_ = r % a.tiles_x; r /= a.tiles_x;
      const int ty_ = r % a.tiles_y; r /= a.tiles_y;
      const int n = r, y0 = ty_ * TH, x0 = tx_ * TW;
      const int iy0 = y0 - a.pad, ix0 = x0 - a.pad;
      unsigned valid = 0;
      const bf16_t* base1 = (MODE != 0 && g == 1) ? a.src1 + (size_t)n * a.s1H * a.s1W * a.s1C + cbase : nullptr;
      const bf16_t* base0 = a.src0 + (size_t)n * a.s0H * a.s0W * a.s0C + cbase;
      const bf16_t* baseg = a.gy + (size_t)n * a.Hout * a.Wout * a.gy_ld + cc * 32;
#pragma unroll
      for (int j = 0; j < XV; ++j) {
        const int iy = iy0 + hyj[j], ix = ix0 + hxj[j];
        const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        valid |= (ok ? 1u : 0u) << j;
        if (MODE != 0 && g == 1) {
          const int sy = min(max(iy - dy1, 0), a.s1H - 1), sx = min(max(ix - dx1, 0), a.s1W - 1);
          xr[j] = WG_LD16O(base1, (unsigned)((sy * a.s1W + sx) * a.s1C + chc * 8) * 2u);
        } else {
          xr[j] = WG_LD16O(base0, ok ? (unsigned)((iy * a.s0W + ix) * a.s0C + chc * 8) * 2u : 0u);
        }
      }
      xvalid = valid;
      unsigned gval = 0;
#pragma unroll
      for (int j = 0; j < GV; ++j) {
        const int gy_ = y0 + gyj[j], gx_ = x0 + gxj[j];
        const bool ok = gy_ < a.Hout && gx_ < a.Wout;
        // (no select on the loaded value here: it would make the request wait for its own data -- zeroed at write time)
        gval |= (ok ? 1u : 0u) << j;
        gr[j] = WG_LD16O(baseg, ok ? (unsigned)((gy_ * a.Wout + gx_) * a.gy_ld + chc * 8) * 2u : 0u);
      }
      gvalid = gval;
    };
    auto write_lds = [&](char* st) {
      char* sX = st;
      char* sG = st + 4 * XPLB;
      int t4 = ptid;
      asm volatile("" : "+v"(t4));
      const int p0 = t4 >> 2, ch = t4 & 3;
#pragma unroll
      for (int j = 0; j < XV; ++j) {
        const int pix = p0 + j * PP;
        if (pix >= NX) continue;
        vec v = xr[j];
        if (MODE == 1 && g >= 2) {
          float f[8];
          E::unpack(v, f);
          if (g == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = f[i] * f[i];
          } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) f[i] = __builtin_amdgcn_sqrtf(f[i] + 1e-8f);
          }
          v = E::pack(f);
        }
        if (!((xvalid >> j) & 1u)) v = E::zero();
        *reinterpret_cast<vec*>(sX + ch * XPLB + pix * 16) = v;
      }
#pragma unroll
      for (int j = 0; j < GV; ++j) {
        const int pix = p0 + j * PP;
        if (pix >= NG) continue;
        const vec gv = ((gvalid >> j) & 1u) ? gr[j] : E::zero();
        *reinterpret_cast<vec*>(sG + ch * GPLB + pix * 16) = gv;
        if (do_bias) {
          float f[8];
          E::unpack(gv, f);
#pragma unroll
          for (int i = 0; i < 8; ++i) bs[i] += f[i];
        }
      }
    };
    // (ONE register set, requests one tile ahead.  Two sets with requests two tiles ahead -- straight-line, exact vmcnt(35 .. 18)
    // in the ISA -- measured SLOWER on every layer (same box: 82.5 -> 85.9, 51.9 -> 55.9, 69.5 -> 73.3 us ...): with 36 loads per
    // staging thread in flight the CU's vector-memory queue is what the requests wait in, as on the forward kernels.)
    load_tile(tile0);
    write_lds(smem);
    if (tile0 + 1 < tile_end) load_tile(tile0 + 1);
    __syncthreads();
    for (int t = tile0; t < tile_end; ++t) {
      if (t + 1 < tile_end) {
        write_lds(smem + ((t + 1 - tile0) & 1) * STAGE);
        if (t + 2 < tile_end) load_tile(t + 2);
      }
      __syncthreads();
    }
    if (do_bias) {
#pragma unroll
      for (int i = 0; i < 8; ++i) sBs[ptid * 8 + i] = bs[i];
    }
  }
  // ---- both roles: the four row blocks of every tap are summed through LDS in wave order, then one emission per element
  const int lr = lane & 31, lh = lane >> 5;
#pragma unroll 1
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int ty = 0; ty < 3; ++ty)
#pragma unroll
        for (int tx = 0; tx < 3; ++tx)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int co = (i & 3) + 8 * (i >> 2) + 4 * lh;
            float* p = sR + ((ty * 3 + tx) * 32 + co) * 32 + lr;
            *p = w == 0 ? acc[ty][tx][i] : *p + acc[ty][tx][i];
          }
    }
    __syncthreads();
  }
  if (do_bias && tid < 32) {
    const int sl = tid >> 3, e = tid & 7;
    float t = 0.f;
    for (int p = 0; p < NS / 4; ++p) t += sBs[(p * 4 + sl) * 8 + e];
    if (a.gb_part != nullptr) { UNCL_CHK(a.chk, a.gb_part + (size_t)blockIdx.x * a.Cout + cc * 32 + tid, 4); a.gb_part[(size_t)blockIdx.x * a.Cout + cc * 32 + tid] = t; }
    else { UNCL_CHK(a.chk, a.gb + cc * 32 + tid, 4); atomicAdd(a.gb + cc * 32 + tid, t); }
  }
  for (int i = tid; i < 9 * 1024; i += 512) {
    const int tap = i >> 10, co = (i >> 5) & 31, ci = i & 31;
    wg_emit(a, ((size_t)tap * a.Cout + cc * 32 + co) * a.Cin + kc * 32 + ci, sR[i]);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Skip-concat form with split roles: ONE workgroup covers the FOUR members [x2 | x1 | x2^2 | sqrt(x2 + 1e-8)] of a 32-channel
// slice of the skip against one 32-channel chunk of gY.  These launches are bound by the bytes they pull in (the per-pair kernels
// stage 4 x (X tile + gY tile) for what is two tensors' worth of fresh bytes: x2 three times, gY four times; the 252^2 layer ran at
// 1.8 TB/s of useful traffic): here x1, x2 and gY are read ONCE per tile and the staging waves derive the square and the root in
// registers (under the previous tile's multiplies).
//
//   waves 0-3  multiply: wave m = member m, all nine taps, all 8 rows of the 8 x 32 pixel tile (halo rows walked once: 60 X + 16
//              gY fragment reads per 144 MFMAs); every wave owns its 9 x 32 x 32 sums for the whole launch -- no cross-wave
//              reduction, the sums go from the registers to the gradient.
//   waves 4-7  stage: loads of tile t+1 in flight during tile t's multiplies, x2^2 / sqrt computed from them, then all five
//              images (4 x 21.8 KB + 16.6 KB: one stage, 104 KB) written between two barriers.
template <int UNUSED>
__global__ __launch_bounds__(512, 1) void wgrad3c_kernel(const WgArgs a) {
  using E = Elem<bf16_t>;
  using vec = bf16x8;
  constexpr int TH = 8, TW = 32, XH = TH + 2, XW = TW + 2;
  constexpr int NX = XH * XW, NG = TH * TW, NS = 256, PP = NS / 4;
  constexpr int XV = (NX + PP - 1) / PP, GV = (NG + PP - 1) / PP;
  constexpr int XPLB = wg_plane(NX), GPLB = wg_plane(NG);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const sG = smem + 16 * XPLB;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nsl = a.s0C >> 5;                       // 32-channel slices of the skip
  const int sl = blockIdx.z % nsl, cc = blockIdx.z / nsl;
  const int tile0 = (int)blockIdx.x * a.tiles_per_wg;
  const int tile_end = min(tile0 + a.tiles_per_wg, a.total_tiles);
  if (tile0 >= tile_end) return;
  const bool do_bias = a.gb != nullptr && sl == 0;      // workgroup-uniform

  if (wave < 4) {
    f32x16 acc[3][3];
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
      for (int tx = 0; tx < 3; ++tx)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[ty][tx][i] = 0.f;
    const int grp = lane >> 4;
    const int c0 = 16 * (grp & 1), kb = 8 * (grp >> 1);
    const int li = lane & 15;
    const int frag_lane = (li >> 2) * 16 + ((li & 1) << 3);
    const int psel = (c0 >> 3) + ((li & 3) >> 1);
    const char* lbX = smem + (4 * wave + psel) * XPLB + kb * 16 + frag_lane;
    const char* lbG = sG + psel * GPLB + kb * 16 + frag_lane;
    for (int t = tile0; t < tile_end; ++t) {
      __syncthreads();               // the stage holds tile t
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        vec A[3], B[2][3];
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) B[0][tx] = tr_frag_plane(lbX, (h * 16 + tx) * 16);
#pragma unroll
        for (int R = 0; R < XH; ++R) {
          if (R + 1 < XH) {
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) B[(R + 1) & 1][tx] = tr_frag_plane(lbX, ((R + 1) * XW + h * 16 + tx) * 16);
          }
          if (R < TH) A[R % 3] = tr_frag_plane(lbG, (R * TW + h * 16) * 16);
#pragma unroll
          for (int ty = 0; ty < 3; ++ty) {
            const int r = R - ty;
            if (r < 0 || r >= TH) continue;
#pragma unroll
            for (int tx = 0; tx < 3; ++tx)
              acc[ty][tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[r % 3], B[R & 1][tx], acc[ty][tx], 0, 0, 0);
          }
          // (left alone the scheduler requests several halo rows ahead and spills: one row of look-ahead is the design)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();               // done reading: the staging waves may overwrite the stage
    }
    // member `wave` occupies channels wave * s0C + 32 sl .. of the weight's K layout
    const int lr = lane & 31, lh = lane >> 5;
    const int ci = wave * a.s0C + sl * 32 + lr;
#if defined(UNCL_WG_ABLATE_ATOMICS)
    if (a.dw == nullptr)
#endif
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
      for (int tx = 0; tx < 3; ++tx)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int co = cc * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
          wg_emit(a, ((size_t)(ty * 3 + tx) * a.Cout + co) * a.Cin + ci, acc[ty][tx][i]);
        }
    if (do_bias) { __syncthreads(); __syncthreads(); }     // (the staging waves' bias reduction below)
    return;
  }
  // ==================================================================================================================
  // staging waves
  // ==================================================================================================================
  const int ptid = tid - 256;
  const int cbase = sl * 32;
  // TWO raw register sets: the loads of tile t + 2 are requested at the start of tile t's multiplies, a whole tile period before
  // derive() needs them (with one set the request went out just before it was consumed: load latency + derive + write + multiply
  // ran back to back, 5.7 us per tile for 2.3 us of MFMAs)
  struct Raw { vec x1[XV], x2[XV], g[GV]; unsigned valid, gvalid; };
  Raw ra, rb;
  vec sq[XV], rt[XV];
  float bs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) bs[i] = 0.f;
  // per-thread constants of the staging pattern (the same for every tile): halo pixel (hyj, hxj) of X slot j, tile pixel
  // (gyj, gxj) of gY slot j, this thread's 16-byte channel group `ch`
  const int p0c = ptid >> 2, chc = ptid & 3;
  int hyj[XV], hxj[XV], gyj[GV], gxj[GV];
#pragma unroll
  for (int j = 0; j < XV; ++j) {
    const int pix = min(p0c + j * PP, NX - 1);
    hyj[j] = pix / XW; hxj[j] = pix - hyj[j] * XW;
  }
#pragma unroll
  for (int j = 0; j < GV; ++j) {
    const int pix = min(p0c + j * PP, NG - 1);
    gyj[j] = pix / TW; gxj[j] = pix % TW;
  }
  const int dy1 = (a.s0H - a.s1H) >> 1, dx1 = (a.s0W - a.s1W) >> 1;
  // loads are "wave-uniform sample base + 32-bit per-lane byte offset" (global_load_dwordx4 v, v_off, s[base]): a 64-bit per-lane
  // address is built by the compiler inside the load's destination registers, two 64-bit vector operations per load
  auto load_tile = [&](int t, Raw& q) __attribute__((always_inline)) {
    int r = t;
    const int tx_ = r % a.tiles_x; r /= a.tiles_x;
    const int ty_ = r % a.tiles_y; r /= a.tiles_y;
    const int n = r, y0 = ty_ * TH, x0 = tx_ * TW;
    const int iy0 = y0 - a.pad, ix0 = x0 - a.pad;
    unsigned valid = 0;
    const bf16_t* base1 = a.src1 + (size_t)n * a.s1H * a.s1W * a.s1C + cbase;
    const bf16_t* base2 = a.src0 + (size_t)n * a.s0H * a.s0W * a.s0C + cbase;
    const bf16_t* baseg = a.gy + (size_t)n * a.Hout * a.Wout * a.gy_ld + cc * 32;
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int iy = iy0 + hyj[j], ix = ix0 + hxj[j];
      const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      valid |= (ok ? 1u : 0u) << j;
      // the up-sampled operand, replicate-padded to the skip's extent (unet_parts.py:292-298)
      const int sy = min(max(iy - dy1, 0), a.s1H - 1), sx = min(max(ix - dx1, 0), a.s1W - 1);
      q.x1[j] = WG_LD16O(base1, (unsigned)((sy * a.s1W + sx) * a.s1C + chc * 8) * 2u);
      q.x2[j] = WG_LD16O(base2, ok ? (unsigned)((iy * a.s0W + ix) * a.s0C + chc * 8) * 2u : 0u);
    }
    q.valid = valid;
    unsigned gval = 0;
#pragma unroll
    for (int j = 0; j < GV; ++j) {
      const int gy_ = y0 + gyj[j], gx_ = x0 + gxj[j];
      const bool ok = gy_ < a.Hout && gx_ < a.Wout;
      // (no select on the loaded value here: it would make the request wait for its own data -- the zeroing happens at write time)
      gval |= (ok ? 1u : 0u) << j;
      q.g[j] = WG_LD16O(baseg, ok ? (unsigned)((gy_ * a.Wout + gx_) * a.gy_ld + chc * 8) * 2u : 0u);
    }
    q.gvalid = gval;
  };
  // x2^2 and sqrt(x2 + 1e-8) of the loaded slice, zero padding applied to all four members (registers only)
  auto derive = [&](Raw& q) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      float f[8], s2[8], rr[8];
      E::unpack(q.x2[j], f);
#pragma unroll
      for (int i = 0; i < 8; ++i) { s2[i] = f[i] * f[i]; rr[i] = __builtin_amdgcn_sqrtf(f[i] + 1e-8f); }
      sq[j] = E::pack(s2);
      rt[j] = E::pack(rr);
      if (!((q.valid >> j) & 1u)) { q.x1[j] = E::zero(); q.x2[j] = E::zero(); sq[j] = E::zero(); rt[j] = E::zero(); }
    }
  };
  auto write_lds = [&](const Raw& q, float bias_w) __attribute__((always_inline)) {
    int t4 = ptid;
    asm volatile("" : "+v"(t4));
    const int p0 = t4 >> 2, ch = t4 & 3;
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int pix = p0 + j * PP;
      if (pix >= NX) continue;
      char* d = smem + ch * XPLB + pix * 16;
      *reinterpret_cast<vec*>(d) = q.x2[j];                   // member order of the weight's K layout: [x2 | x1 | x2^2 | sqrt]
      *reinterpret_cast<vec*>(d + 4 * XPLB) = q.x1[j];
      *reinterpret_cast<vec*>(d + 8 * XPLB) = sq[j];
      *reinterpret_cast<vec*>(d + 12 * XPLB) = rt[j];
    }
#pragma unroll
    for (int j = 0; j < GV; ++j) {
      const int pix = p0 + j * PP;
      if (pix >= NG) continue;
      const vec gv = ((q.gvalid >> j) & 1u) ? q.g[j] : E::zero();
      *reinterpret_cast<vec*>(sG + ch * GPLB + pix * 16) = gv;
      if (do_bias) {
        float f[8];
        E::unpack(gv, f);
#pragma unroll
        for (int i = 0; i < 8; ++i) bs[i] = fmaf(f[i], bias_w, bs[i]);     // (weight 0: the write past the end of the range)
      }
    }
  };
  // tile t0 + k lives in set (k & 1): a = even, b = odd.  The steps are STRAIGHT-LINE code: past the end of the range they request
  // the last tile again and write a stage nobody reads.  With the requests inside `if (t + 2 < tile_end)` the compiler's waitcnt
  // insertion merges the two paths pessimistically and derive() waits for 12 of the 16 loads just requested (vmcnt(14) .. (4) in
  // the ISA), i.e. for a full memory latency per tile.
  const int last = tile_end - 1;
  load_tile(tile0, ra);
  load_tile(min(tile0 + 1, last), rb);
  derive(ra);
  float bias_on = 1.f;
  write_lds(ra, bias_on);
  // one step of the pipeline while the multiplying waves work on tile t (whose successor sits in `nxt`, and whose registers `cur`
  // are free again): request tile t + 2, derive tile t + 1, hand the stage over, write tile t + 1
  auto step = [&](int t, Raw& cur, Raw& nxt) __attribute__((always_inline)) {
    __syncthreads();                 // the stage holds tile t
    load_tile(min(t + 2, last), cur);
    derive(nxt);
    __syncthreads();                 // the multiplying waves are done with tile t
    write_lds(nxt, t + 1 < tile_end ? 1.f : 0.f);
  };
  for (int t = tile0; t < tile_end; t += 2) {
    step(t, ra, rb);
    if (t + 1 < tile_end) step(t + 1, rb, ra);
  }
  if (do_bias) {
    // (a region of its own behind the stage: the last step's write of the stage -- other waves' -- is not ordered against these)
    float* sBs = reinterpret_cast<float*>(sG + 4 * GPLB);
#pragma unroll
    for (int i = 0; i < 8; ++i) sBs[ptid * 8 + i] = bs[i];
    __syncthreads();
    if (ptid < 32) {
      const int ch8 = ptid >> 3, e = ptid & 7;
      float t = 0.f;
      for (int p = 0; p < NS / 4; ++p) t += sBs[(p * 4 + ch8) * 8 + e];
      if (a.gb_part != nullptr) { UNCL_CHK(a.chk, a.gb_part + (size_t)blockIdx.x * a.Cout + cc * 32 + ptid, 4); a.gb_part[(size_t)blockIdx.x * a.Cout + cc * 32 + ptid] = t; }
      else { UNCL_CHK(a.chk, a.gb + cc * 32 + ptid, 4); atomicAdd(a.gb + cc * 32 + ptid, t); }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------------
// 64 (ci) x 64 (co) channel blocks with split roles (plain sources): the per-pair kernels read X once per gY chunk and gY once per X
// chunk, and a 32-channel chunk is HALF of every 128-byte line of a tensor with >= 64 channels -- these launches are bound by the
// bytes they pull in.  Here a workgroup stages whole 128-byte pixels of both tensors for an 8 x 32 pixel tile (43.5 + 32 KB per
// stage, two stages) and its four multiplying waves are the four (ci half, co half) quadrants: every wave holds all nine taps of
// its 32 x 32 quadrant for all 8 rows (halo rows walked once: 60 X + 16 gY fragment reads per 144 MFMAs) and owns its sums for
// the whole launch -- no cross-wave reduction.  75.5 KB staged per 576 MFMAs where four pairs staged 4 x 71 KB per 1152.
template <int UNUSED>
__global__ __launch_bounds__(512, 1) void wgrad3q_kernel(const WgArgs a) {
  using E = Elem<bf16_t>;
  using vec = bf16x8;
  constexpr int TH = 8, TW = 32, XH = TH + 2, XW = TW + 2;
  constexpr int NX = XH * XW, NG = TH * TW, NS = 256, PP = NS / 8;   // staging threads; PP pixels (x 8 slots of 16 bytes) per pass
  constexpr int XV = (NX + PP - 1) / PP, GV = (NG + PP - 1) / PP;
  constexpr int XPLB = wg_plane(NX), GPLB = wg_plane(NG);     // bytes per plane; eight planes per tensor
  constexpr int STAGE = 8 * XPLB + 8 * GPLB;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nci = a.Cin >> 6;
  const int kc = blockIdx.z % nci, cc = blockIdx.z / nci;     // 64-channel blocks
  const int tile0 = (int)blockIdx.x * a.tiles_per_wg;
  const int tile_end = min(tile0 + a.tiles_per_wg, a.total_tiles);
  if (tile0 >= tile_end) return;
  const bool do_bias = a.gb != nullptr && kc == 0;      // workgroup-uniform

  if (wave < 4) {
    const int cih = wave & 1, coh = wave >> 1;
    f32x16 acc[3][3];
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
      for (int tx = 0; tx < 3; ++tx)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[ty][tx][i] = 0.f;
    const int grp = lane >> 4;
    const int c0 = 16 * (grp & 1), kb = 8 * (grp >> 1);
    const int li = lane & 15;
    const int frag_lane = (li >> 2) * 16 + ((li & 1) << 3);
    const int psel = (c0 >> 3) + ((li & 3) >> 1);
    const int offX = (4 * cih + psel) * XPLB + kb * 16 + frag_lane;
    const int offG = 8 * XPLB + (4 * coh + psel) * GPLB + kb * 16 + frag_lane;
    __syncthreads();                 // stage 0 holds the first tile
    for (int t = tile0; t < tile_end; ++t) {
      const char* st = smem + ((t - tile0) & 1) * STAGE;
      const char* lbX = st + offX;
      const char* lbG = st + offG;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        vec A[3], B[2][3];
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) B[0][tx] = tr_frag_plane(lbX, (h * 16 + tx) * 16);
#pragma unroll
        for (int R = 0; R < XH; ++R) {
          if (R + 1 < XH) {
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) B[(R + 1) & 1][tx] = tr_frag_plane(lbX, ((R + 1) * XW + h * 16 + tx) * 16);
          }
          if (R < TH) A[R % 3] = tr_frag_plane(lbG, (R * TW + h * 16) * 16);
#pragma unroll
          for (int ty = 0; ty < 3; ++ty) {
            const int r = R - ty;
            if (r < 0 || r >= TH) continue;
#pragma unroll
            for (int tx = 0; tx < 3; ++tx)
              acc[ty][tx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[r % 3], B[R & 1][tx], acc[ty][tx], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);      // one halo row of look-ahead (see wgrad3c_kernel)
        }
      }
      __syncthreads();               // done with this stage; the other one holds the next tile
    }
    const int lr = lane & 31, lh = lane >> 5;
    const int ci = kc * 64 + 32 * cih + lr;
#if defined(UNCL_WG_ABLATE_ATOMICS)
    if (a.dw == nullptr)
#endif
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
      for (int tx = 0; tx < 3; ++tx)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int co = cc * 64 + 32 * coh + (i & 3) + 8 * (i >> 2) + 4 * lh;
          wg_emit(a, ((size_t)(ty * 3 + tx) * a.Cout + co) * a.Cin + ci, acc[ty][tx][i]);
        }
    if (do_bias) { __syncthreads(); __syncthreads(); }     // (the staging waves' bias reduction below)
    return;
  }
  // ==================================================================================================================
  // staging waves: thread = (pixel p0 + j PP, 16-byte channel group ch of the pixel's 128 bytes)
  // ==================================================================================================================
  const int ptid = tid - 256;
  const int p0c = ptid >> 3, chc = ptid & 7;
  int hyj[XV], hxj[XV], gyj[GV], gxj[GV];
#pragma unroll
  for (int j = 0; j < XV; ++j) {
    const int pix = min(p0c + j * PP, NX - 1);
    hyj[j] = pix / XW; hxj[j] = pix - hyj[j] * XW;
  }
#pragma unroll
  for (int j = 0; j < GV; ++j) {
    const int pix = min(p0c + j * PP, NG - 1);
    gyj[j] = pix / TW; gxj[j] = pix % TW;
  }
  vec xr[XV], gr[GV];
  unsigned xvalid = 0, gvalid = 0;
  float bs[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) bs[i] = 0.f;
  auto load_tile = [&](int t) __attribute__((always_inline)) {
    int r = t;
    const int tx_ = r % a.tiles_x; r /= a.tiles_x;
    const int ty_ = r % a.tiles_y; r /= a.tiles_y;
    const int n = r, y0 = ty_ * TH, x0 = tx_ * TW;
    const int iy0 = y0 - a.pad, ix0 = x0 - a.pad;
    unsigned valid = 0, gval = 0;
    const bf16_t* base0 = a.src0 + (size_t)n * a.s0H * a.s0W * a.s0C + kc * 64;
    const bf16_t* baseg = a.gy + (size_t)n * a.Hout * a.Wout * a.gy_ld + cc * 64;
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int iy = iy0 + hyj[j], ix = ix0 + hxj[j];
      const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      valid |= (ok ? 1u : 0u) << j;
      xr[j] = WG_LD16O(base0, ok ? (unsigned)((iy * a.s0W + ix) * a.s0C + chc * 8) * 2u : 0u);
    }
#pragma unroll
    for (int j = 0; j < GV; ++j) {
      const int gy_ = y0 + gyj[j], gx_ = x0 + gxj[j];
      const bool ok = gy_ < a.Hout && gx_ < a.Wout;
      gval |= (ok ? 1u : 0u) << j;
      gr[j] = WG_LD16O(baseg, ok ? (unsigned)((gy_ * a.Wout + gx_) * a.gy_ld + chc * 8) * 2u : 0u);
    }
    xvalid = valid; gvalid = gval;
  };
  auto write_lds = [&](char* st) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int pix = p0c + j * PP;
      if (pix >= NX) continue;
      *reinterpret_cast<vec*>(st + chc * XPLB + pix * 16) = ((xvalid >> j) & 1u) ? xr[j] : E::zero();
    }
#pragma unroll
    for (int j = 0; j < GV; ++j) {
      const int pix = p0c + j * PP;
      if (pix >= NG) continue;
      const vec gv = ((gvalid >> j) & 1u) ? gr[j] : E::zero();
      *reinterpret_cast<vec*>(st + 8 * XPLB + chc * GPLB + pix * 16) = gv;
      if (do_bias) {
        float f[8];
        E::unpack(gv, f);
#pragma unroll
        for (int i = 0; i < 8; ++i) bs[i] += f[i];
      }
    }
  };
  load_tile(tile0);
  write_lds(smem);
  if (tile0 + 1 < tile_end) load_tile(tile0 + 1);
  __syncthreads();
  for (int t = tile0; t < tile_end; ++t) {
    if (t + 1 < tile_end) {
      write_lds(smem + ((t + 1 - tile0) & 1) * STAGE);
      if (t + 2 < tile_end) load_tile(t + 2);
    }
    __syncthreads();
  }
  if (do_bias) {
    // (every wave is past the last stage barrier: the stages are free) [thread][8] partial sums -> 64 channels
    float* sBs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 8; ++i) sBs[ptid * 8 + i] = bs[i];
    __syncthreads();
    if (ptid < 64) {
      const int ch8 = ptid >> 3, e = ptid & 7;
      float t = 0.f;
      for (int p = 0; p < NS / 8; ++p) t += sBs[(p * 8 + ch8) * 8 + e];
      if (a.gb_part != nullptr) { UNCL_CHK(a.chk, a.gb_part + (size_t)blockIdx.x * a.Cout + cc * 64 + ptid, 4); a.gb_part[(size_t)blockIdx.x * a.Cout + cc * 64 + ptid] = t; }
      else { UNCL_CHK(a.chk, a.gb + cc * 64 + ptid, 4); atomicAdd(a.gb + cc * 64 + ptid, t); }
    }
    __syncthreads();
  }
}

// Deterministic mode: a caller-owned scratch buffer for the per-group partial sums (thread-local: uncl_gen_backward sets it
// around its pass; every weight-gradient launch of a pass and its reduction run on ONE stream, in order, so the buffer is reused
// launch after launch).  NULL (default) = float atomics.
thread_local float* t_wg_scratch = nullptr;
thread_local size_t t_wg_scratch_floats = 0;

// how many pixel-range groups the scratch holds for a launch that covers E gradient elements (+ Cout bias sums)
int wg_scratch_groups(const WgArgs& a, int groups) {
  if (t_wg_scratch == nullptr) return groups;
  const size_t per = (size_t)a.E + (a.gb != nullptr ? (size_t)a.Cout : 0);
  const size_t fit = t_wg_scratch_floats / (per ? per : 1);
  return fit < (size_t)groups ? (int)(fit ? fit : 1) : groups;
}
// after `groups` is final: point the launch at the scratch (false: it does not even hold one group -> atomics)
bool wg_use_scratch(WgArgs& a, int groups) {
  a.part = nullptr; a.gb_part = nullptr;
  if (t_wg_scratch == nullptr) return false;
  const size_t need = (size_t)groups * ((size_t)a.E + (a.gb != nullptr ? (size_t)a.Cout : 0));
  if (need > t_wg_scratch_floats) return false;
  a.part = t_wg_scratch;
  if (a.gb != nullptr) a.gb_part = t_wg_scratch + (size_t)groups * a.E;
#ifdef UNCL_CHECKED
  uncl_chk_add(a.chk, t_wg_scratch, (unsigned long long)t_wg_scratch_floats * sizeof(float));
#endif
  return true;
}
int wg_reduce(const WgArgs& a, int groups, hipStream_t s) {
  if (a.part == nullptr) return UNCL_OK;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((a.E + 31) / 32)), dim3(256), 0, s, a.part, groups, a.E, a.dw);
  if (a.gb_part != nullptr)
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((a.Cout + 31) / 32)), dim3(256), 0, s, a.gb_part, groups, (long long)a.Cout, a.gb);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

template <int MODE>
int launch_wg3w(WgArgs& a, hipStream_t s) {
  constexpr size_t lds = 8 * (size_t)wg_plane(10 * 34) + 8 * (size_t)wg_plane(8 * 32);
  auto kern = wgrad3w_kernel<MODE>;
  static UnclDevOnce attr_done;
  if (attr_done.need()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return UNCL_ERR_LAUNCH;
    attr_done.done();
  }
  // 8-row tiles (the caller's tile counts are for 16 rows)
  a.tiles_y = (a.Hout + 7) / 8;
  a.total_tiles = (a.total_tiles / ((a.Hout + 15) / 16)) * a.tiles_y;
  const int pairs = (a.Cin / 64) * (a.Cout / 64);
  // Pixel-range groups per block: every group ends with Cin x Cout x 9 float atomics into the same gradient, so the groups
  // multiply that traffic.  Measured per layer (N = 32 step, us, 32 x 32 kernel -> this one at 512 / 256 / 128 workgroups):
  // up_path.0.conv.conv 158 -> 157 / 125 / 219, up_path.1.conv.conv 153 -> 155 / 127 / 220, down_path.2 conv1 75 -> 84 / 63 /
  // 78, 64 -> 64 at 122^2 94 -> 120 / 89 / 103, every other eligible layer slower at any count (46 -> 52 ... 141).
  static const int slots = [] { const char* e = getenv("UNCL_WG_WIDE_SLOTS"); return e ? atoi(e) : 256; }();
  int groups = slots / pairs;
  if (groups < 1) groups = 1;
  if (groups > a.total_tiles) groups = a.total_tiles;
  a.E = (long long)9 * a.Cout * a.Cin;
  groups = wg_scratch_groups(a, groups);
  a.tiles_per_wg = (a.total_tiles + groups - 1) / groups;
  groups = (a.total_tiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
  wg_use_scratch(a, groups);
  hipLaunchKernelGGL(kern, dim3(groups, 1, pairs), dim3(384), lds, s, a);
  UNCL_CHECK_LAUNCH();
  return wg_reduce(a, groups, s);
}

template <int MODE>
int launch_wg3(WgArgs& a, hipStream_t s) {
  constexpr size_t lds = 4 * (size_t)wg_plane(18 * 34) + 4 * (size_t)wg_plane(16 * 32);
  auto kern = wgrad3_kernel<MODE>;
  static UnclDevOnce attr_done;
  if (attr_done.need()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return UNCL_ERR_LAUNCH;
    attr_done.done();
  }
  const int pairs = a.nci * (a.Cout / 32);
  int groups = 512 / pairs;      // one persistent workgroup per resident slot, see launch_wg
  if (groups < 1) groups = 1;
  if (groups > a.total_tiles) groups = a.total_tiles;
  a.E = (long long)9 * a.Cout * a.Cin;
  groups = wg_scratch_groups(a, groups);
  a.tiles_per_wg = (a.total_tiles + groups - 1) / groups;
  groups = (a.total_tiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
  wg_use_scratch(a, groups);
  hipLaunchKernelGGL(kern, dim3(groups, 1, pairs), dim3(384), lds, s, a);
  UNCL_CHECK_LAUNCH();
  return wg_reduce(a, groups, s);
}

template <int MODE>
int launch_wg3r(WgArgs& a, hipStream_t s) {
  constexpr size_t lds = 2 * (4 * (size_t)wg_plane(18 * 34) + 4 * (size_t)wg_plane(16 * 32));
  auto kern = wgrad3r_kernel<MODE>;
  static UnclDevOnce attr_done;
  if (attr_done.need()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return UNCL_ERR_LAUNCH;
    attr_done.done();
  }
  const int pairs = a.nci * (a.Cout / 32);
  const int cus = uncl_cu_count() > 0 ? uncl_cu_count() : 256;
  int groups = cus / pairs;      // one persistent workgroup per CU
  if (groups < 1) groups = 1;
  if (groups > a.total_tiles) groups = a.total_tiles;
  a.E = (long long)9 * a.Cout * a.Cin;
  groups = wg_scratch_groups(a, groups);
  a.tiles_per_wg = (a.total_tiles + groups - 1) / groups;
  groups = (a.total_tiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
  wg_use_scratch(a, groups);
  hipLaunchKernelGGL(kern, dim3(groups, 1, pairs), dim3(512), lds, s, a);
  UNCL_CHECK_LAUNCH();
  return wg_reduce(a, groups, s);
}

int launch_wg3q(WgArgs& a, hipStream_t s) {
  constexpr size_t lds = 2 * (8 * (size_t)wg_plane(10 * 34) + 8 * (size_t)wg_plane(8 * 32));
  auto kern = wgrad3q_kernel<0>;
  static UnclDevOnce attr_done;
  if (attr_done.need()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return UNCL_ERR_LAUNCH;
    attr_done.done();
  }
  // 8-row tiles (the caller's tile counts are for 16 rows)
  a.tiles_y = (a.Hout + 7) / 8;
  a.total_tiles = (a.total_tiles / ((a.Hout + 15) / 16)) * a.tiles_y;
  const int blocks = (a.Cin / 64) * (a.Cout / 64);
  const int cus = uncl_cu_count() > 0 ? uncl_cu_count() : 256;
  int groups = cus / blocks;
  if (groups < 1) groups = 1;
  if (groups > a.total_tiles) groups = a.total_tiles;
  a.E = (long long)9 * a.Cout * a.Cin;
  groups = wg_scratch_groups(a, groups);
  a.tiles_per_wg = (a.total_tiles + groups - 1) / groups;
  groups = (a.total_tiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
  wg_use_scratch(a, groups);
  hipLaunchKernelGGL(kern, dim3(groups, 1, blocks), dim3(512), lds, s, a);
  UNCL_CHECK_LAUNCH();
  return wg_reduce(a, groups, s);
}

int launch_wg3c(WgArgs& a, hipStream_t s) {
  constexpr size_t lds = 16 * (size_t)wg_plane(10 * 34) + 4 * (size_t)wg_plane(8 * 32) + 256 * 8 * sizeof(float);   // stage + bias sums
  auto kern = wgrad3c_kernel<0>;
  static UnclDevOnce attr_done;
  if (attr_done.need()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return UNCL_ERR_LAUNCH;
    attr_done.done();
  }
  // 8-row tiles (the caller's tile counts are for 16 rows)
  a.tiles_y = (a.Hout + 7) / 8;
  a.total_tiles = (a.total_tiles / ((a.Hout + 15) / 16)) * a.tiles_y;
  const int blocks = (a.s0C / 32) * (a.Cout / 32);      // (skip slice, gY chunk) pairs, four members each
  const int cus = uncl_cu_count() > 0 ? uncl_cu_count() : 256;
  int groups = cus / blocks;
  if (groups < 1) groups = 1;
  if (groups > a.total_tiles) groups = a.total_tiles;
  a.E = (long long)9 * a.Cout * a.Cin;
  groups = wg_scratch_groups(a, groups);
  a.tiles_per_wg = (a.total_tiles + groups - 1) / groups;
  groups = (a.total_tiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
  wg_use_scratch(a, groups);
  hipLaunchKernelGGL(kern, dim3(groups, 1, blocks), dim3(512), lds, s, a);
  UNCL_CHECK_LAUNCH();
  return wg_reduce(a, groups, s);
}

template <int MODE, int KS>
int launch_wg(WgArgs& a, hipStream_t s) {
  constexpr int XW = 32 + KS - 1;
  constexpr size_t lds = (size_t)16 * XW * 64 + 512 * 64;
  auto kern = wgrad_kernel<MODE, KS>;
  static UnclDevOnce attr_done;
  if (attr_done.need()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return UNCL_ERR_LAUNCH;
    attr_done.done();
  }
  const int pairs = a.nci * (a.Cout / 32);
  // one persistent workgroup per resident slot (2 per CU x 256 CUs): every workgroup ends with a cross-wave reduction and
  // 1024 * KS float atomics, so oversubscribing the CUs only multiplies that tail (measured on the training step: 2048
  // workgroups 4.7 ms of weight gradients, 512 workgroups 3.4 ms, 128 workgroups 6.8 ms)
  // grid.y: vertical taps, the four taps of the 2x2 kernel, or the groups of a grouped 1x1
  const int gy_ = (KS == 1 && a.up_tap >= 0) ? 4 : (KS == 1 ? a.groups : KS);
  int groups = 512 / (gy_ * pairs);
  if (groups < 1) groups = 1;
  if (groups > a.total_tiles) groups = a.total_tiles;
  // elements of dw the launch covers: KS taps, or the four taps of the 2x2 kernel, or the groups of a grouped 1x1
  a.E = (long long)(KS == 3 ? 9 : (a.up_tap >= 0 ? 4 : a.groups)) * a.Cout * a.Cin;
  groups = wg_scratch_groups(a, groups);
  a.tiles_per_wg = (a.total_tiles + groups - 1) / groups;
  groups = (a.total_tiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
  wg_use_scratch(a, groups);
  hipLaunchKernelGGL(kern, dim3(groups, gy_, pairs), dim3(256), lds, s, a);
  UNCL_CHECK_LAUNCH();
  return wg_reduce(a, groups, s);
}

// column sums of a [rows][C] bf16 matrix -> fp32 (bias gradients), deterministic two-stage
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ x, float* __restrict__ partial, size_t rows, int C,
                                                     int ld) {
  // thread = (row lane, 8-channel vector); blockDim = 256 = (256 / (C/8)) rows x (C/8) vectors
  const int VC = C / 8;
  const int v = threadIdx.x % VC, rl = threadIdx.x / VC, rpb = 256 / VC;
  float s[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = 0.f;
  for (size_t r = (size_t)blockIdx.x * rpb + rl; r < rows; r += (size_t)gridDim.x * rpb) {
    const bf16x8 t = *reinterpret_cast<const bf16x8*>(x + r * ld + v * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] += (float)t[i];
  }
  __shared__ float red[256 * 8];
#pragma unroll
  for (int i = 0; i < 8; ++i) red[threadIdx.x * 8 + i] = s[i];
  __syncthreads();
  if (threadIdx.x < C) {
    const int vv = threadIdx.x / 8, ii = threadIdx.x % 8;
    float t = 0.f;
    for (int r = 0; r < rpb; ++r) t += red[(r * VC + vv) * 8 + ii];
    partial[(size_t)blockIdx.x * C + threadIdx.x] = t;
  }
}
// 32 columns x 8 partial-row lanes per workgroup, fp64, fixed order
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, int blocks, int C, float* __restrict__ out,
                                                           int accumulate) {
  __shared__ double red[8][33];
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double s = 0.0;
  if (c < C)
    for (int b = rg; b < blocks; b += 8) s += (double)partial[(size_t)b * C + c];
  red[rg][cl] = s;
  __syncthreads();
  if (rg == 0 && c < C) {
    double t = 0.0;
    for (int i = 0; i < 8; ++i) t += red[i][cl];
    out[c] = accumulate ? out[c] + (float)t : (float)t;
  }
}

// the same reduction for up to UNCL_COLSUM_MAX_ITEMS staged column sums in ONE launch (grid.y = item): a backward pass ends
// with one of these instead of a dependent 16 us launch after every layer
struct ColsumBatch {
  uncl_colsum_item it[UNCL_COLSUM_MAX_ITEMS];
};
__global__ __launch_bounds__(256) void colsum_final_batch_kernel(const ColsumBatch t) {
  const uncl_colsum_item& e = t.it[blockIdx.y];
  if ((int)blockIdx.x * 32 >= e.C) return;
  __shared__ double red[8][33];
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double s = 0.0;
  if (c < e.C) {
    int b = rg;
    for (; b + 24 < e.blocks; b += 32) {      // four loads in flight, summed in the order of the one-at-a-time loop
      const float v0 = e.partial[(size_t)b * e.C + c], v1 = e.partial[(size_t)(b + 8) * e.C + c];
      const float v2 = e.partial[(size_t)(b + 16) * e.C + c], v3 = e.partial[(size_t)(b + 24) * e.C + c];
      s += (double)v0; s += (double)v1; s += (double)v2; s += (double)v3;
    }
    for (; b < e.blocks; b += 8) s += (double)e.partial[(size_t)b * e.C + c];
  }
  red[rg][cl] = s;
  __syncthreads();
  if (rg == 0 && c < e.C) {
    double tt = 0.0;
    for (int i = 0; i < 8; ++i) tt += red[i][cl];
    e.out[c] = e.accumulate ? e.out[c] + (float)tt : (float)tt;
  }
}

// packed fp32 [tap][Cout][Cin] gradient -> reference-layout fp32 gradient (inverse of uncl_pack_conv_weight)
__global__ void unpack_wgrad_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cout, int Cin, int kk,
                                    int transposed, int flip, int accumulate) {
  const size_t total = (size_t)kk * Cout * Cin;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin), co = (int)((i / Cin) % Cout), tap = (int)(i / ((size_t)Cin * Cout));
    const int ts = flip ? (kk - 1 - tap) : tap;
    const size_t d = transposed ? (((size_t)ci * Cout + co) * kk + ts) : (((size_t)co * Cin + ci) * kk + ts);
    dst[d] = accumulate ? dst[d] + src[i] : src[i];
  }
}

struct UnpackBatch {
  uncl_unpack_item it[UNCL_PACK_MAX_ITEMS];
};
// 32 x 32 channel blocks through LDS, the mirror image of pack_weight_batch_kernel: packed runs of 32 channels in, runs of
// 32*kk contiguous reference-layout floats out
__global__ __launch_bounds__(256) void unpack_wgrad_batch_kernel(const UnpackBatch t) {
  const uncl_unpack_item& e = t.it[blockIdx.y];
  const int kk = e.k * e.k, Cin = e.Cin, Cout = e.Cout;
  const int A = e.transposed ? Cin : Cout, B = e.transposed ? Cout : Cin;
  const int tb = B >> 5, tiles = (A >> 5) * tb;
  if ((A | B) & 31) {
    const size_t total = (size_t)kk * Cout * Cin;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
      const int ci = (int)(i % Cin), co = (int)((i / Cin) % Cout), tap = (int)(i / ((size_t)Cin * Cout));
      const int ts = e.flip ? (kk - 1 - tap) : tap;
      const size_t d = e.transposed ? (((size_t)ci * Cout + co) * kk + ts) : (((size_t)co * Cin + ci) * kk + ts);
      e.dst[d] = e.accumulate ? e.dst[d] + e.packed[i] : e.packed[i];
    }
    return;
  }
  __shared__ float sm[32 * (32 * 9 + 1)];
  const int pitch = 32 * kk + 1;
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int a0 = (tile / tb) << 5, b0 = (tile % tb) << 5;
    __syncthreads();
    for (int v = threadIdx.x; v < 32 * 32 * kk; v += 256) {
      const int ci_l = v & 31, co_l = (v >> 5) & 31, tap = v >> 10;
      const int ts = e.flip ? (kk - 1 - tap) : tap;
      const int a = e.transposed ? ci_l : co_l, bb = e.transposed ? co_l : ci_l;
      const int co = e.transposed ? b0 + co_l : a0 + co_l, ci = e.transposed ? a0 + ci_l : b0 + ci_l;
      sm[a * pitch + bb * kk + ts] = e.packed[((size_t)tap * Cout + co) * Cin + ci];
    }
    __syncthreads();
    for (int v = threadIdx.x; v < 32 * 32 * kk; v += 256) {
      const int a = v / (32 * kk), r = v - a * (32 * kk);
      float* d = e.dst + ((size_t)(a0 + a) * B + b0) * kk + r;
      *d = e.accumulate ? *d + sm[a * pitch + r] : sm[a * pitch + r];
    }
  }
}

}  // namespace

static std::atomic<int> g_wg_wide{[] { const char* e = getenv("UNCL_WG_WIDE"); return e ? atoi(e) : 1; }()};
// 3x3 layers with Cin and Cout multiples of 64 can use the 64 x 64 channel-block kernel: 1 (default) the skip-concat layers, 2 every
// eligible layer, 0 none (A/B timing, and the parity tests that compare the two kernels).  Returns the previous setting.
extern "C" int uncl_wgrad_set_wide(int on) { return g_wg_wide.exchange(on < 0 ? 0 : (on > 2 ? 2 : on)); }
static std::atomic<int> g_wg_cat{[] { const char* e = getenv("UNCL_WG_CAT"); return e ? atoi(e) : 1; }()};
// skip-concat 3x3 layers: 1 (default) one workgroup per (32-channel skip slice, gY chunk) covering all four members; 0 the per-pair
// kernels.  Returns the previous setting.
extern "C" int uncl_wgrad_set_cat(int on) { return g_wg_cat.exchange(on ? 1 : 0); }
static std::atomic<int> g_wg_quad{[] { const char* e = getenv("UNCL_WG_QUAD"); return e ? atoi(e) : 1; }()};
static const int g_wg_quad_min = [] { const char* e = getenv("UNCL_WG_QUAD_MIN"); return e ? atoi(e) : 6; }();
// plain 3x3 layers with Cin and Cout multiples of 64: 1 (default) the split-role 64 x 64 block kernel where both sides have >= 128
// channels and a workgroup gets >= UNCL_WG_QUAD_MIN (6) tiles, 2 always, 0 never (the per-pair kernels).  Returns the previous setting.
extern "C" int uncl_wgrad_set_quad(int on) { return g_wg_quad.exchange(on < 0 ? 0 : (on > 2 ? 2 : on)); }
static std::atomic<int> g_wg_roll{[] { const char* e = getenv("UNCL_WG_ROLL"); return e ? atoi(e) : 1; }()};
static const int g_wg_roll_min = [] { const char* e = getenv("UNCL_WG_ROLL_MIN"); return e ? atoi(e) : 4; }();
// 3x3 layers on the 32 x 32 channel-pair path: 1 (default) the split-role kernel where a workgroup gets enough tiles, 2 always,
// 0 the six-wave kernel (A/B timing, parity tests between the two).  Returns the previous setting.
extern "C" int uncl_wgrad_set_roll(int on) { return g_wg_roll.exchange(on < 0 ? 0 : (on > 2 ? 2 : on)); }

// Deterministic weight gradients: with a scratch buffer set (device memory, `bytes` >= uncl_wgrad_scratch_bytes() for every
// layer of the generator at full speed; less only reduces the number of pixel-range groups per launch), the weight- and
// bias-gradient kernels called from THIS thread store per-group partial sums there and add them up in a fixed order instead of
// using float atomics: two runs on the same inputs give the same bits.  NULL / 0 restores the atomics.
extern "C" int uncl_wgrad_set_scratch(void* scratch, size_t bytes) {
  t_wg_scratch = reinterpret_cast<float*>(scratch);
  t_wg_scratch_floats = scratch ? bytes / sizeof(float) : 0;
  return UNCL_OK;
}
// largest partial buffer a generator layer asks for: up_path.0.conv.conv, 9 x 1024 x 128 elements x 8 groups (+ slack)
extern "C" size_t uncl_wgrad_scratch_bytes(void) { return (size_t)48 << 20; }
bool uncl_wgrad_deterministic() { return t_wg_scratch != nullptr; }

// dw_packed must be zeroed by the caller (it is accumulated with atomics).  Descriptor fields used: ksize (3 or 1),
// pad, src_mode (PLAIN / CONCAT_SSR), N, H, W, Cin, Cout, src0/src1 (+dims); `gy` is (N, Hout, Wout, Cout) bf16.
extern "C" int uncl_conv_wgrad(const uncl_conv_desc* d, const void* gy, float* dw_packed, void* stream) {
  return uncl_conv_wgrad_bias(d, gy, dw_packed, nullptr, stream);
}

// The same with the bias gradient: gb[co] += sum over pixels of gy[..][co] (3x3 layers only; float atomics like dw_packed: zero
// it first).  The kernel stages every gy tile anyway, so the sums replace a separate column-sum pass over gy.
extern "C" int uncl_conv_wgrad_bias(const uncl_conv_desc* d, const void* gy, float* dw_packed, float* gb, void* stream) {
  if (!d || !gy || !dw_packed || d->dtype != UNCL_BF16) return UNCL_ERR_ARG;
  if (gb != nullptr && d->ksize != 3) return UNCL_ERR_ARG;
  if (d->ksize != 3 && d->ksize != 1) return UNCL_ERR_ARG;
  if (d->src_mode != UNCL_SRC_PLAIN && d->src_mode != UNCL_SRC_CONCAT_SSR) return UNCL_ERR_ARG;
  if (d->Cin % 32 != 0 || d->Cout % 32 != 0 || d->src0 == nullptr) return UNCL_ERR_ARG;
  if (d->src_mode == UNCL_SRC_CONCAT_SSR && (d->src1 == nullptr || d->Cin != 4 * d->src0_C)) return UNCL_ERR_ARG;
  // (the concat kernels walk 32-channel slices of two members of equal width: a member of 8 / 16 / 24 channels passed the Cin % 32
  // test above and divided by zero in launch_wg3c's workgroup count)
  if (d->src_mode == UNCL_SRC_CONCAT_SSR && (d->src0_C % 32 != 0 || d->src1_C != d->src0_C)) return UNCL_ERR_ARG;
  WgArgs a;
  a.src0 = (const bf16_t*)d->src0; a.src1 = (const bf16_t*)d->src1; a.gy = (const bf16_t*)gy; a.dw = dw_packed; a.gb = gb;
  a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.pad = d->ksize == 3 ? d->pad : 0; a.ks = d->ksize;
  a.s0H = d->src0_H; a.s0W = d->src0_W; a.s0C = d->src0_C; a.s1H = d->src1_H; a.s1W = d->src1_W; a.s1C = d->src1_C;
  if (d->ksize == 1) {
    // flatten: one row of M pixels
    const long long M = (long long)d->N * d->H * d->W;
    if (M > 0x7fffffffLL) return UNCL_ERR_ARG;
    const int rows = (int)((M + 31) / 32);  // a (rows x 32) image with a ragged last row
    a.H = rows; a.W = 32; a.s0H = rows; a.s0W = 32;
    a.Hout = rows; a.Wout = 32;
    a.M = M;
    a.tiles_x = 1; a.tiles_y = (rows + 15) / 16;
    a.total_tiles = a.tiles_y;
  } else {
    a.Hout = d->H + 2 * d->pad - 2; a.Wout = d->W + 2 * d->pad - 2;
    a.tiles_x = (a.Wout + 31) / 32; a.tiles_y = (a.Hout + 15) / 16;
    a.total_tiles = d->N * a.tiles_x * a.tiles_y;
  }
  a.nci = d->Cin / 32;
  if (d->ksize == 3) a.M = 0;
  a.up_tap = -1; a.upH = a.upW = 0;
  a.groups = (d->ksize == 1 && d->z_mode == UNCL_Z_GROUPS && d->groups > 1) ? d->groups : 1;
  a.gy_ld = d->out_C > 0 ? d->out_C : d->Cout;
#ifdef UNCL_CHECKED
  {
    const unsigned long long N = (unsigned long long)d->N, px_in = (unsigned long long)d->H * d->W;
    uncl_chk_reset(a.chk);
    uncl_chk_add(a.chk, d->src0, N * (d->ksize == 3 ? (unsigned long long)d->src0_H * d->src0_W : px_in) * d->src0_C * 2);
    uncl_chk_add(a.chk, d->src1, N * (unsigned long long)d->src1_H * d->src1_W * d->src1_C * 2);
    uncl_chk_add(a.chk, gy, N * (d->ksize == 3 ? (unsigned long long)a.Hout * a.Wout : px_in) * a.gy_ld * 2);
    uncl_chk_add(a.chk, dw_packed, (unsigned long long)(d->ksize == 3 ? 9 : a.groups) * d->Cout * d->Cin * 4);
    uncl_chk_add(a.chk, gb, (unsigned long long)d->Cout * 4);
  }
#endif
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (d->ksize == 3) {
    // 64 x 64 channel blocks where the layer has them (concat sources: a 64-channel chunk must lie inside one member)
    // mode 1 (default): the skip-concat layers only -- the layers it was measured faster on (launch_wg3w); 2: every eligible layer
    const int wmode = g_wg_wide.load(std::memory_order_relaxed);
    // (the wide kernel has no registers left for the bias sums: a call that asks for them takes the 32 x 32 kernel)
    bool wide = wmode != 0 && gb == nullptr && d->Cin % 64 == 0 && d->Cout % 64 == 0 && a.gy_ld % 8 == 0 &&
                (d->src_mode == UNCL_SRC_PLAIN ? wmode == 2 : d->src0_C % 64 == 0);
    if (wide && wmode == 1) {
      // ... and only with enough tiles per workgroup to amortise its bigger atomic tail: the N = 32 image step gains 0.075 ms
      // (16 tiles per workgroup on both layers), the video step's N = 8 passes (4 tiles) lose 0.07 ms
      const int pairs64 = (d->Cin / 64) * (d->Cout / 64);
      const int groups = 256 / pairs64 > 0 ? 256 / pairs64 : 1;
      wide = d->N * a.tiles_x * ((a.Hout + 7) / 8) >= 8 * groups;
    }
    // skip-concat layers: all four members of a slice in one workgroup (wgrad3c_kernel); UNCL_WG_CAT=0: the per-pair kernels
    if (d->src_mode == UNCL_SRC_CONCAT_SSR && g_wg_cat.load(std::memory_order_relaxed) != 0 && a.gy_ld % 8 == 0) return launch_wg3c(a, s);
    if (wide) return d->src_mode == UNCL_SRC_PLAIN ? launch_wg3w<0>(a, s) : launch_wg3w<1>(a, s);
    // plain sources with 64-channel blocks on both sides: the split-role 64 x 64 kernel (UNCL_WG_QUAD: 1 = where a workgroup gets
    // at least UNCL_WG_QUAD_MIN 8-row tiles, 2 = always, 0 = never)
    // Measured per layer (tools/wgrad_ab.py, N = 32, us, split-role per pair -> 64 x 64 blocks): 128 -> 128 at 57^2 65.9 -> 60.7,
    // 256 -> 256 at 24^2 63.3 -> 51.6; 64 -> 64 at 122^2 69.5 -> 74.6, 64 -> 128 at 59^2 45.5 -> 50.6, 128 -> 256 at 26^2 43.8 -> 46.0,
    // the 10 / 12 pixel maps 41.8 -> 44.4: it pays where both sides have at least two blocks' worth of re-reading to save AND a
    // workgroup still gets six tiles; mode 1 takes exactly those.
    const int quad = g_wg_quad.load(std::memory_order_relaxed);
    if (quad != 0 && d->src_mode == UNCL_SRC_PLAIN && d->Cin % 64 == 0 && d->Cout % 64 == 0 && a.gy_ld % 8 == 0 && d->src0_C % 8 == 0) {
      const int blocks = (d->Cin / 64) * (d->Cout / 64);
      const int groups = 256 / blocks > 0 ? 256 / blocks : 1;
      if (quad == 2 || (d->Cin >= 128 && d->Cout >= 128 && d->N * a.tiles_x * ((a.Hout + 7) / 8) >= g_wg_quad_min * groups))
        return launch_wg3q(a, s);
    }
    // split-role kernel (wgrad3r_kernel): 1 = where it has at least `UNCL_WG_ROLL_MIN` tiles per workgroup, 2 = always, 0 = never
    const int roll = g_wg_roll.load(std::memory_order_relaxed);
    if (roll != 0) {
      const int pairs = a.nci * (d->Cout / 32);
      const int groups = 256 / pairs > 0 ? 256 / pairs : 1;
      // (N = 8 video frames, tools/wgrad_ab.py 8: with two tiles per workgroup the split-role kernel still wins on the layers with few
      // channel pairs -- 32 -> 64 at 124^2 46 -> 35 us, 64 -> 128 at 59^2 41 -> 31 -- and loses where 32 / 64 pairs leave four groups)
      if (roll == 2 || a.total_tiles >= g_wg_roll_min * groups || (pairs <= 8 && a.total_tiles >= 2 * groups))
        return d->src_mode == UNCL_SRC_PLAIN ? launch_wg3r<0>(a, s) : launch_wg3r<1>(a, s);
    }
    return d->src_mode == UNCL_SRC_PLAIN ? launch_wg3<0>(a, s) : launch_wg3<1>(a, s);
  }
  return launch_wg<0, 1>(a, s);
}

// Weight gradient of ConvTranspose2d(k2, s2): dw_packed[tap][Cout][C] += sum_p gy[(n,2y+dy,2x+dx)][co] * x[(n,y,x)][ci].
// x: (N,H,W,C) bf16, gy: (N,2H,2W,Cout) bf16; dw_packed zeroed by the caller.
extern "C" int uncl_upconv2x2_wgrad(const void* x, const void* gy, float* dw_packed, int N, int H, int W, int C, int Cout,
                                    void* stream) {
  if (!x || !gy || !dw_packed || C % 32 != 0 || Cout % 32 != 0) return UNCL_ERR_ARG;
  const long long M = (long long)N * H * W;
  if (M > 0x7fffffffLL) return UNCL_ERR_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  WgArgs a;     // one launch: grid.y enumerates the four taps, each adding into its own [Cout][C] slice of dw_packed
  a.src0 = (const bf16_t*)x; a.src1 = nullptr; a.gy = (const bf16_t*)gy; a.dw = dw_packed; a.gb = nullptr;
  a.Cin = C; a.Cout = Cout; a.pad = 0; a.ks = 1;
  a.s0C = C; a.s1H = a.s1W = a.s1C = 0;
  const int rows = (int)((M + 31) / 32);
  a.H = rows; a.W = 32; a.s0H = rows; a.s0W = 32; a.Hout = rows; a.Wout = 32; a.M = M;
  a.tiles_x = 1; a.tiles_y = (rows + 15) / 16; a.total_tiles = a.tiles_y;
  a.nci = C / 32;
  a.up_tap = 0; a.upH = H; a.upW = W; a.gy_ld = Cout; a.groups = 1;
#ifdef UNCL_CHECKED
  uncl_chk_reset(a.chk);
  uncl_chk_add(a.chk, x, (unsigned long long)M * C * 2);
  uncl_chk_add(a.chk, gy, (unsigned long long)M * 4 * Cout * 2);
  uncl_chk_add(a.chk, dw_packed, 4ull * Cout * C * 4);
#endif
  return launch_wg<0, 1>(a, s);
}

// every packed gradient of a network -> reference layout, one launch per UNCL_PACK_MAX_ITEMS tensors
extern "C" int uncl_unpack_conv_wgrads(const uncl_unpack_item* items, int n_items, void* stream) {
  if (n_items == 0) return UNCL_OK;
  if (!items || n_items < 0) return UNCL_ERR_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  for (int i0 = 0; i0 < n_items; i0 += UNCL_PACK_MAX_ITEMS) {
    const int n = n_items - i0 < UNCL_PACK_MAX_ITEMS ? n_items - i0 : UNCL_PACK_MAX_ITEMS;
    UnpackBatch t = {};
    for (int i = 0; i < n; ++i) {
      const uncl_unpack_item& e = items[i0 + i];
      if (!e.packed || !e.dst || e.Cout <= 0 || e.Cin <= 0 || e.k <= 0) return UNCL_ERR_ARG;
      t.it[i] = e;
    }
    hipLaunchKernelGGL(unpack_wgrad_batch_kernel, dim3(96, n), dim3(256), 0, s, t);
  }
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

extern "C" size_t uncl_colsum_workspace_bytes(int C) { return (size_t)512 * C * sizeof(float); }

// out[c] (+)= sum over rows of x[row][c]; x bf16 [rows][ld] (any ld >= C), C a multiple of 8 (bias gradients)
extern "C" int uncl_colsum_bf16(const void* x, long long rows, int C, int ld, float* out, int accumulate, void* workspace,
                                void* stream) {
  if (!x || !out || !workspace || rows <= 0 || C < 8 || C % 8 != 0 || ld < C) return UNCL_ERR_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  for (int c0 = 0; c0 < C; c0 += 256) {
    const int cw = C - c0 < 256 ? C - c0 : 256;
    if (256 % (cw / 8) != 0) return UNCL_ERR_ARG;
    const int rpb = 256 / (cw / 8);
    const int blocks = (int)((rows + rpb - 1) / rpb < 512 ? (rows + rpb - 1) / rpb : 512);
    hipLaunchKernelGGL(colsum_kernel, dim3(blocks), dim3(256), 0, s, (const bf16_t*)x + c0, (float*)workspace, (size_t)rows, cw, ld);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((cw + 31) / 32), dim3(256), 0, s, (const float*)workspace, blocks, cw, out + c0,
                       accumulate);
  }
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

// Two-call form for a whole backward pass: stage 1 per layer into its own partial buffer (512*C floats, C <= 256), then one
// uncl_colsum_finish over all staged items.  Same arithmetic and summation order as uncl_colsum_bf16.
extern "C" int uncl_colsum_bf16_stage(const void* x, long long rows, int C, int ld, float* partial, float* out, int accumulate,
                                      uncl_colsum_item* item, void* stream) {
  if (!x || !out || !partial || !item || rows <= 0 || C < 8 || C > 256 || C % 8 != 0 || ld < C || 256 % (C / 8) != 0)
    return UNCL_ERR_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int rpb = 256 / (C / 8);
  const int blocks = (int)((rows + rpb - 1) / rpb < 512 ? (rows + rpb - 1) / rpb : 512);
  hipLaunchKernelGGL(colsum_kernel, dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, partial, (size_t)rows, C, ld);
  UNCL_CHECK_LAUNCH();
  item->partial = partial; item->blocks = blocks; item->C = C; item->out = out; item->accumulate = accumulate;
  return UNCL_OK;
}

extern "C" int uncl_colsum_finish(const uncl_colsum_item* items, int n_items, void* stream) {
  if (n_items == 0) return UNCL_OK;
  if (!items || n_items < 0 || n_items > UNCL_COLSUM_MAX_ITEMS) return UNCL_ERR_ARG;
  ColsumBatch t = {};
  for (int i = 0; i < n_items; ++i) {
    if (!items[i].partial || !items[i].out || items[i].C <= 0 || items[i].C > 256 || items[i].blocks <= 0) return UNCL_ERR_ARG;
    t.it[i] = items[i];
  }
  hipLaunchKernelGGL(colsum_final_batch_kernel, dim3(8, n_items), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), t);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

extern "C" int uncl_unpack_conv_wgrad(const float* packed, float* dst, int Cout, int Cin, int k, int transposed, int flip,
                                      int accumulate, void* stream) {
  if (!packed || !dst || Cout <= 0 || Cin <= 0 || k <= 0) return UNCL_ERR_ARG;
  const size_t total = (size_t)k * k * Cout * Cin;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), packed, dst, Cout,
                     Cin, k * k, transposed, flip, accumulate);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}
