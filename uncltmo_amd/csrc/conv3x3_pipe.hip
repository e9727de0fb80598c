// Pipelined 3x3 implicit-GEMM convolution for gfx950 — the generator's hot kernel (bf16).
//
// Same GEMM orientation as conv_igemm.hip (D[cout][pixel] = W[cout][k] X[k][pixel], 64-byte K-chunks; LDS rows padded
// to 80 bytes instead of swizzled) with three structural changes that matter on MI355X:
//
//  * persistent workgroups + register prefetch: a workgroup walks a contiguous range of tiles, each a list of
//    K-chunk steps; the global loads of step s+1 are issued before the MFMAs of step s and land in VGPRs while
//    the matrix pipe is busy, then are written to the single LDS buffer after a barrier (HBM/L2 latency hides
//    under MFMA; two workgroups per CU cover the LDS-write / barrier bubbles).  All per-slot addressing is
//    precomputed once (32-bit offsets from a wave-uniform sample base), loads are unconditional (clamped, then
//    masked) so the compiler can issue them back to back;
//  * sliding-window fragment reuse: a wave owns MPW consecutive output rows, so the activation fragment of
//    input row r serves output rows r, r-1, r-2 for the three vertical taps: (MPW+2)*3*2 instead of MPW*9*2
//    LDS reads per chunk;
//  * LDS-transposed epilogue: bias/activation in registers, the bf16 tile is written to LDS as [pixel][cout]
//    and read back row-wise, so every global store instruction writes 1 KiB of contiguous NHWC bytes; the same
//    image feeds the fused 2x2 max-pool output (unet_parts.py:212,233) and the fused 1-channel 1x1 + sigmoid
//    (outconv, Unet_singleFrame.py:207-209).
#include <cstdlib>

#include "conv3x3_args.h"

namespace {

// Phase timing (measurement builds only, -DUNCL_PIPE_TIMING; tools/pipe_phase_timing.py): wave 0 of every workgroup
// accumulates s_memtime deltas per loop phase into g_pipe_t[]; the product library compiles all of this away.
#ifdef UNCL_PIPE_TIMING
__device__ unsigned long long g_pipe_t[16];
__device__ __forceinline__ unsigned long long pt_now() {
  unsigned long long t;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define PT_DECL unsigned long long pt_acc[12] = {}; unsigned long long pt_last = pt_now();
#define PT(i) { const unsigned long long pt_t = pt_now(); pt_acc[i] += pt_t - pt_last; pt_last = pt_t; }
#define PT_FLUSH() if (threadIdx.x == 0) { for (int i = 0; i < 12; ++i) atomicAdd(&g_pipe_t[i], pt_acc[i]); atomicAdd(&g_pipe_t[15], 1ull); }
#else
#define PT_DECL
#define PT(i)
#define PT_FLUSH()
#endif

// MODE: 0 plain, 1 concat [x2, x1, x2^2, sqrt(x2+1e-8)], 2 concat [x2, x1], 3 first layer recomputed from the image,
//       4 = 1 with x1 = ConvTranspose2d(k2, s2)(src1) recomputed inside the loader (32 channels, same extent as the skip)
//
// FLAT (small maps: a whole sample's output is <= 128 pixels): the tile is not a rectangle of one sample but flat_S WHOLE
// samples -- the 256 accumulator rows are the output pixels of those samples one after the other (consecutive in memory, so
// the store loop sees a 32-pixel-wide image), the staging area holds their zero-padded input maps back to back, and every
// lane reads its own window: fragment address = lane base (its pixel's top-left input slot) + tap offset.  A 10 x 10 map
// fills 78 % of the tile instead of 20 % of a 16 x 32 rectangle.
template <typename T, int NT, int MPW, int WAVES, int MODE, bool PREV, bool FLAT = false, bool O1D = false>
__global__ __launch_bounds__(WAVES * 64, (NT == 1 && MPW == 2) ? 3 : 2) void conv3x3_pipe_kernel(const PipeArgs a) {
  static_assert(!FLAT || (MODE == 0 && !PREV), "flat tiles: plain source only");
  using E = Elem<T>;
  using vec = typename Elem<T>::vec;
  using vec4 = typename Elem<T>::vec4;
  static_assert(WAVES == 4, "staging pattern below is written for 256 threads");
  constexpr int NTHR = WAVES * 64;
  constexpr int TH = MPW * WAVES, TW = 32;
  constexpr int HH = TH + 2, HW = TW + 2;
  constexpr int NPIX = HH * HW;
  constexpr int CT = NT * 32;
  static_assert(HH % 2 == 0, "two halo rows per staging pass");
  constexpr int RS = HH / 2;   // regular slots: 32 columns x 2 rows x 4 vectors per pass
  constexpr int XV = RS + 1;   // + one slot for the two extra halo columns
  constexpr int WROWS = 9 * CT;
  constexpr int WVN = (WROWS + 63) / 64;
  constexpr bool W_RAGGED = WROWS % 64 != 0;
  constexpr int SLOTS = CT / 8;
  // The epilogue image [TH*TW pixels][CT channels] re-uses the activation staging area; where it is the larger of the two
  // (64-channel tiles: 32 KB vs 27 KB) the weight image starts behind it, so that the weights of a single-chunk layer
  // (down_path.0.mpconv.1.conv, 32 -> 64) stay resident across tiles instead of being re-staged (37 KB) for every tile.
  constexpr int XIMG = NPIX * 80 > TH * TW * CT * 2 ? NPIX * 80 : TH * TW * CT * 2;
  constexpr int ST_IT = (TH * TW * SLOTS) / NTHR;             // main-store passes
  constexpr int ROWS_PER_IT = NTHR / (TW * SLOTS);            // output rows covered per pass
  static_assert(ROWS_PER_IT * TW * SLOTS == NTHR, "store pass must cover whole rows");
  static_assert(ST_IT % 4 == 0, "main-store passes are issued in groups of four");

  // LDS rows are padded from 64 to 80 bytes: 16 consecutive rows then start on 16 distinct 16-byte slots of the
  // 256-byte bank row (20*p mod 64 covers every multiple of 4), so ds_read_b128 is conflict-free with NO swizzle and
  // every fragment address is "lane base + compile-time immediate" (zero address arithmetic in the MFMA loop).
  constexpr int RP = 80;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sX = smem;
  char* sW = smem + XIMG;
  char* sB = sW + WROWS * RP;  // CT fp32 biases of the current cout tile (outside the epilogue image)
  // MODE 3: the fp32 image patch under the halo tile ((HH+2) x (HW+2))
  constexpr int PW3 = HW + 2, PN3 = (HH + 2) * PW3;
  float* sP = reinterpret_cast<float*>(sB + CT * 4);
  char* sO = smem;
  // MODE 4: source patch of the fused 2x2 transposed conv [9 x 17 px][64 B], its weights [128 rows][64 B] and bias; they
  // live in the weight image's space between the barrier that retires the previous step and the staging of the new weights
  constexpr int UPH = HH / 2, UPW = HW / 2, UPN = UPH * UPW;
  char* sU = sW;
  char* sUW = sU + ((UPN * 64 + 255) & ~255);
  float* sUB = reinterpret_cast<float*>(sUW + 128 * 64);
  static_assert(MODE != 4 || ((UPN * 64 + 255) & ~255) + 128 * 64 + 128 <= WROWS * 80, "fused up-conv scratch fits in the weight image");
  static_assert(MODE != 4 || (NT == 1 && MPW == 4 && UPN * 4 <= 3 * 256), "fused up-conv staging is written for the 16 x 32 tile");

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;

  int tile = (int)blockIdx.x * a.tiles_per_wg;
  const int tile_end = min(tile + a.tiles_per_wg, a.total_tiles);
  if (tile >= tile_end) return;
  const bool w_static = a.nk == 1 && a.n_ct == 1;      // one weight chunk for the whole launch: staged once

  // ---- per-thread constants of the staging pattern (identical for every step)
  // regular slot j: halo pixel (hy0 + 2j, hx), vector ch; extra slot: halo pixel (ey, 32 + ec), vector ch
  const int ch = tid & 3, p0 = tid >> 2;
  const int hx = p0 & 31, hy0 = p0 >> 5;
  const int ey = tid >> 3, ec = (tid >> 2) & 1;
  const bool e_on = tid < HH * 8;
  const int row_el = a.s0W * a.s0C;                          // elements per source row
  const int xoff_r = (hy0 * a.s0W + hx) * a.s0C + ch * 8;    // + j * 2 * row_el
  const int xoff_e = (ey * a.s0W + 32 + ec) * a.s0C + ch * 8;
  const int pix_r0 = hy0 * HW + hx;                          // + j * 2 * HW
  const int pix_e = ey * HW + 32 + ec;
  const int lds_w0 = p0 * 80 + (ch << 4);  // + j * 64 * 80
  const int woff0 = ((p0 / CT) * a.Cout + (p0 % CT)) * 32 + ch * 8;  // K-chunk-major weights (common.h); + j * (64 / CT) * Cout * 32

  // FLAT: source sample (or -1: zero padding / unused slot) and element offset inside the group for each staging slot
  int flat_s[FLAT ? XV : 1];
  unsigned flat_off[FLAT ? XV : 1];
  int flat_lane[FLAT ? MPW : 1];     // byte address of the top-left input slot of this lane's output pixel, per M-tile
  if (FLAT) {
    const int Hp = a.H + 2 * a.pad, Wp = a.W + 2 * a.pad, hpwp = Hp * Wp;
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int sl = j < XV - 1 ? pix_r0 + j * 2 * HW : pix_e;
      const int sj = sl / hpwp, r = sl - sj * hpwp;
      const int py = r / Wp, px = r - py * Wp;
      const int iy = py - a.pad, ix = px - a.pad;
      const bool in = sj < a.flat_S && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      flat_s[j] = in ? sj : -1;
      flat_off[j] = in ? (unsigned)(((sj * a.H + iy) * a.W + ix) * a.s0C + ch * 8) : 0u;
    }
    const int Wo = a.W + 2 * a.pad - 2;
#pragma unroll
    for (int m = 0; m < MPW; ++m) {
      const int q = min((wave * MPW + m) * 32 + lr, a.flat_S * a.flat_hw - 1);
      const int sj = q / a.flat_hw, r = q - sj * a.flat_hw;
      const int oy = r / Wo, ox = r - oy * Wo;
      flat_lane[m] = (sj * hpwp + oy * Wp + ox) * RP;
    }
  }

  // ---- tile cursor (contiguous range per workgroup, carried without divisions)
  int t_ct, t_tx, t_ty, t_n;
  {
    int r = tile;
    t_ct = r % a.n_ct; r /= a.n_ct;
    t_tx = r % a.tiles_x; r /= a.tiles_x;
    t_ty = r % a.tiles_y; r /= a.tiles_y;
    t_n = r;
  }
  int c_n = t_n, c_y0 = t_ty * TH, c_x0 = t_tx * TW, c_co = t_ct * CT, c_kc = 0;
  int n_n = 0, n_y0 = 0, n_x0 = 0, n_co = 0, n_kc = 0;

  vec xr[XV];
  vec wr[WVN];
  constexpr int IRN = (PN3 + NTHR - 1) / NTHR;   // MODE 3: image-patch values prefetched per thread
  float ir[IRN];
  vec preA;                                       // MODE 3: first-layer weight fragment (cout = lr, k = 8*lh + j -> tap)
  float preB[16];                                 // MODE 3: first-layer biases of this lane's channels 8q + 4*lh + r
  if (MODE == 3) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 8 * lh + j;
      if (k < 9) UNCL_CHK(a.chk, a.pre_w + lr * 9 + k, 4);
      preA[j] = (T)(k < 9 ? a.pre_w[lr * 9 + k] : 0.f);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) { if (a.pre_b) UNCL_CHK(a.chk, a.pre_b + 8 * (i >> 2) + 4 * lh + (i & 3), 4); preB[i] = a.pre_b ? a.pre_b[8 * (i >> 2) + 4 * lh + (i & 3)] : 0.f; }
#pragma unroll
    for (int k = 0; k < IRN; ++k) ir[k] = 0.f;
  }
  f32x4 br = {0.f, 0.f, 0.f, 0.f};
  unsigned xvalid = 0;
  int g_pending = 0;
  bool b_pending = false;
  int u_iy0 = 0, u_ix0 = 0;   // MODE 4: image coordinates of the halo tile whose up-conv patch is in flight

  // K-chunk order.  MODE 1 walks each 32-channel slice of the skip as [x1, sqrt(x2), x2^2, x2] (ssr_member): the x2 registers
  // loaded for the second step are kept and re-staged for the third and fourth, so the skip is
  // read from memory once instead of three times; `wk` is the chunk's position in the weight's K layout
  // [x2 | x1 | x2^2 | sqrt] (unet_parts.py:319-322).
  auto load_regs = [&](int n, int y0, int x0, int cout0, int kc, bool with_w) {
    int g = 0, cbase = kc * 32, wk = kc;
    bool reuse = false;
    if (MODE == 1 || MODE == 4) {
      const int ph = kc & 3;
      cbase = (kc >> 2) * 32;
      g = ssr_member(ph);
      wk = g * (a.s0C >> 5) + (kc >> 2);
      reuse = ph >= 2;
    } else if (MODE == 2) {
      g = cbase / a.s0C;
      cbase -= g * a.s0C;
    }
    g_pending = g;
    b_pending = kc == 0;
    const int iy0 = y0 - a.pad, ix0 = x0 - a.pad;
    // the upsampled map has the skip's extent on three of the four decoder levels (2*126 = 252, 2*61 = 122, 2*12 = 24):
    // then it needs no replicate padding and takes the same scalar-base / masked paths as the skip itself
    const bool same_ext = a.s1H == a.s0H && a.s1W == a.s0W;
    const bf16_t* xsrc = (MODE != 0 && g == 1) ? a.src1 : a.src0;
    if (MODE == 3) {
      // the image patch under the halo tile: first-layer pixel (y, x) reads image rows y..y+2, columns x..x+2.  Rows and
      // columns past the image only feed outputs that are never stored (valid convolution), so they are clamped.
      if (a.img_off) UNCL_CHK(a.chk, a.img_off + n, 4);
      const float* ib = a.img + (a.img_off ? (size_t)a.img_off[n] : (size_t)n * a.imgH * a.imgW);
#pragma unroll
      for (int k = 0; k < IRN; ++k) {
        const int idx = min(tid + k * NTHR, PN3 - 1);
        const int pr = idx / PW3, pc = idx - pr * PW3;
        UNCL_CHK(a.chk, ib + (size_t)min(iy0 + pr, a.imgH - 1) * a.imgP + min(ix0 + pc, a.imgW - 1), 4);
        ir[k] = ib[(size_t)min(iy0 + pr, a.imgH - 1) * a.imgP + min(ix0 + pc, a.imgW - 1)];
      }
      xvalid = 0xffffffffu;
    } else if (reuse) {
      // xr / xvalid still hold this tile's x2 slice
    } else if (MODE == 4 && g == 1) {
      // the 9 x 17 source pixels of the 2x2 stride-2 transposed conv under the 18 x 34 halo tile (tile origins are even),
      // 4 vectors of 8 channels each, and its 8 KB of weights; out-of-image source pixels are clamped (their outputs are
      // outside the image too and are staged as zeros)
      const int sy0 = iy0 >> 1, sx0 = ix0 >> 1;
      u_iy0 = iy0; u_ix0 = ix0;
      const bf16_t* ub = a.src1 + (size_t)n * a.s1H * a.s1W * 32;
      // once per tile: the thread's addressing is recomputed here rather than carried in registers across the whole loop
      int t4 = tid;
      asm volatile("" : "+v"(t4));
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int v = min(t4 + k * NTHR, UPN * 4 - 1);
        const int sp = v >> 2, sl = v & 3;
        const int spy = sp / UPW, spx = sp - spy * UPW;
        const int yy = min(max(sy0 + spy, 0), a.s1H - 1), xx = min(max(sx0 + spx, 0), a.s1W - 1);
        xr[k] = LD16OV(vec, ub, (unsigned)(((yy * a.s1W + xx) * 32 + sl * 8) * 2));
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) xr[3 + k] = LD16OV(vec, a.up_w, (unsigned)((t4 + k * NTHR) * 16));
      xvalid = 0xffffffffu;
    } else if (MODE != 0 && MODE != 4 && g == 1 && !same_ext) {
      // upsampled map, replicate-padded to the skip's extent (unet_parts.py:292-298)
      const bf16_t* base = a.src1 + (size_t)n * a.s1H * a.s1W * a.s1C + cbase + ch * 8;
      const int dy = (a.s0H - a.s1H) >> 1, dx = (a.s0W - a.s1W) >> 1;
      unsigned valid = 0;
      const int ix = ix0 + hx;
      const bool xok = (unsigned)ix < (unsigned)a.W;
      const int sx = min(max(ix - dx, 0), a.s1W - 1);
#pragma unroll
      for (int j = 0; j < RS; ++j) {
        const int iy = iy0 + hy0 + 2 * j;
        const bool ok = xok && (unsigned)iy < (unsigned)a.H;
        const int sy = min(max(iy - dy, 0), a.s1H - 1);
        valid |= (ok ? 1u : 0u) << j;
        xr[j] = LD16OV(vec, base, (unsigned)((sy * a.s1W + sx) * a.s1C) * 2u);
      }
      {
        const int iy = iy0 + ey, ixe = ix0 + 32 + ec;
        const bool ok = e_on && (unsigned)iy < (unsigned)a.H && (unsigned)ixe < (unsigned)a.W;
        const int sy = min(max(iy - dy, 0), a.s1H - 1), sxe = min(max(ixe - dx, 0), a.s1W - 1);
        valid |= (ok ? 1u : 0u) << RS;
        xr[RS] = LD16OV(vec, base, (unsigned)((sy * a.s1W + sxe) * a.s1C) * 2u);
      }
      xvalid = valid;
    } else if (FLAT) {
      // slot sl of the staging area = sample sl / (Hp Wp) of this tile's group, padded-map pixel (py, px): the thread's slots
      // and their source offsets do not depend on the tile, only the group's base does
      const bf16_t* base = a.src0 + (size_t)n * a.flat_S * a.s0H * a.s0W * a.s0C + cbase;
      const int left = a.flat_N - n * a.flat_S;          // samples of the batch still covered by this group
      unsigned valid = 0;
#pragma unroll
      for (int j = 0; j <= RS; ++j) {
        const int sj = flat_s[j];
        const bool ok = sj >= 0 && sj < left && (j < RS || e_on);
        valid |= (ok ? 1u : 0u) << j;
        xr[j] = LD16OV(vec, base, ok ? flat_off[j] * 2u : 0u);
      }
      xvalid = valid;
    } else {
#ifdef UNCL_FORCE_INTERIOR  // timing experiment only (tools/pipe_phase_timing.py --force-interior): wrong borders
      const bool interior = true;
#else
      const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + HH <= a.H && ix0 + HW <= a.W;  // wave-uniform
#endif
      if (interior && !PREV) {
        // the whole halo tile is inside the image: one scalar base, a constant stride between slots
        const bf16_t* base = xsrc + ((size_t)n * a.s0H * a.s0W + (size_t)iy0 * a.s0W + ix0) * a.s0C + cbase;
        // the stride between slots goes into the scalar base, so one offset VGPR serves all regular slots
#pragma unroll
        for (int j = 0; j < RS; ++j) xr[j] = LD16OV(vec, base + j * 2 * row_el, (unsigned)xoff_r * 2u);
        xr[RS] = LD16OV(vec, base, (unsigned)(e_on ? xoff_e : 0) * 2u);
        xvalid = 0xffffffffu;
      } else {
        const bf16_t* base = xsrc + (size_t)n * a.s0H * a.s0W * a.s0C + cbase;
        const int toff = (iy0 * a.s0W + ix0) * a.s0C;  // may be negative on the border; masked below
        unsigned valid = 0;
        const bool xok = (unsigned)(ix0 + hx) < (unsigned)a.W;
#pragma unroll
        for (int j = 0; j <= RS; ++j) {
          bool ok;
          int eoff;
          if (j < RS) {
            ok = xok && (unsigned)(iy0 + hy0 + 2 * j) < (unsigned)a.H;
            eoff = xoff_r + j * 2 * row_el;
          } else {
            ok = e_on && (unsigned)(iy0 + ey) < (unsigned)a.H && (unsigned)(ix0 + 32 + ec) < (unsigned)a.W;
            eoff = xoff_e;
          }
          valid |= (ok ? 1u : 0u) << j;
          const unsigned off = ok ? (unsigned)(toff + eoff) : 0u;
          xr[j] = LD16OV(vec, base, off * 2u);
          if (PREV) {
            const int c = cbase + ch * 8;
            if (c < a.prev_ch) {
              const vec p = LD16V(vec, a.prev0 + (size_t)n * a.s0H * a.s0W * a.s0C + cbase + off);
#pragma unroll
              for (int i = 0; i < 8; ++i)
                if (c + i < a.prev_ch) xr[j][i] = p[i];
            }
          }
        }
        xvalid = valid;
      }
    }
    if (with_w) {
      const bf16_t* wb = a.weight + ((size_t)wk * 9 * a.Cout + cout0) * 32;
      const int wstride = (64 / CT) * a.Cout * 32;  // CT = 32: two taps per pass, CT = 64: one
#pragma unroll
      for (int j = 0; j < WVN; ++j) {
        unsigned off = (unsigned)woff0;
        if (W_RAGGED && j == WVN - 1) off = (p0 + 64 * j < WROWS) ? off : 0u;
        wr[j] = LD16OV(vec, wb + j * wstride, off * 2u);
      }
    }
    // last on purpose: the alternative x paths above are laid out one after the other, and the hazard check of a later
    // one waits for whatever an earlier one might have issued -- with the bias load in front that wait was real
    if (b_pending && tid < CT / 4 && a.bias != nullptr) br = LD16O_F32(a.bias + cout0, (unsigned)tid * 16u);
  };

  f32x16 acc[MPW][NT];
  f32x16 zero16;
#pragma unroll
  for (int i = 0; i < 16; ++i) zero16[i] = 0.f;

  auto write_lds = [&](bool with_w) {
    const bool all_ok = xvalid == 0xffffffffu;  // wave-uniform in practice (interior tiles)
    if (MODE == 3) {
      // stage the image patch, then build the 32-channel halo tile with the matrix cores: per 32 halo pixels two
      // v_mfma_f32_32x32x16_bf16 over the nine taps of the bf16 head and of the bf16 tail of the fp32 input (x = hi + lo
      // to 2^-17); the D layout (lane = pixel, 4 consecutive channels per register quad) is already the staging layout.
#pragma unroll
      for (int k = 0; k < IRN; ++k)
        if (tid + k * NTHR < PN3) sP[tid + k * NTHR] = ir[k];
      __syncthreads();
      f32x16 preBv;
#pragma unroll
      for (int i = 0; i < 16; ++i) preBv[i] = preB[i];
      constexpr int MT3 = (NPIX + 31) / 32, IT3 = (MT3 + WAVES - 1) / WAVES;
      static_assert(HW == 34 && NPIX < 2048, "the reciprocal multiply below divides by 34");
      // every lane issues its eight tap reads unconditionally (the upper half-wave only owns tap 8: its other slots read
      // a valid address and are zeroed), the M-tile loop is unrolled so that the reads of one tile overlap the MFMAs and
      // LDS writes of the previous one
#pragma unroll
      for (int i = 0; i < IT3; ++i) {
        const int mt = wave + WAVES * i;           // wave-uniform
        if (mt < MT3) {
          const int p = mt * 32 + lr;
          const int pcl = min(p, NPIX - 1);
          const int py = (pcl * 241) >> 13, px = pcl - py * HW;   // / 34 for pcl < 2048 (HW == 34)
          const float* pp = sP + py * PW3 + px;
          vec Bh, Bl;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int o0 = (j / 3) * PW3 + j % 3;  // tap j (lower half-wave)
            float v = pp[lh ? 2 * PW3 + 2 : o0];   // upper half-wave: tap 8 in slot 0
            if (j > 0) v = lh ? 0.f : v;
            const T hi = (T)v;
            Bh[j] = hi;
            Bl[j] = (T)(v - (float)hi);
          }
          // the bias is the first MFMA's C operand, as in conv3x3_pc_kernel (the two structures are bit-identical)
          f32x16 c3 = mfma32x16(preA, Bh, preBv);
          c3 = mfma32x16(preA, Bl, c3);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            vec4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float t = c3[4 * q + e];
              o[e] = (T)(fmaxf(t, 0.f) + a.slope * fminf(t, 0.f));
            }
            if (p < NPIX) *reinterpret_cast<vec4*>(sX + p * RP + (q << 4) + (lh << 3)) = o;
          }
        }
      }
    } else if (MODE == 4 && g_pending == 1) {
      // x1 = ConvTranspose2d(k2, s2)(src1) + bias for the halo tile: a skinny GEMM per 32 source pixels (K = 32 input
      // channels, 4 taps x 32 output channels), every result scattered to output pixel (2 sy + dy, 2 sx + dx) of the
      // staging image in the D layout it already has; pixels outside the image are the conv's zero padding
      int t4 = tid;
      asm volatile("" : "+v"(t4));   // as in load_regs: no loop-carried addressing registers for this once-per-tile block
      const int lr = t4 & 31, lh = (t4 >> 5) & 1;
      static_assert(WAVES == 4, "one wave per tap of the 2x2 kernel");
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int v = t4 + k * NTHR;
        if (v < UPN * 4) *reinterpret_cast<vec*>(sU + (v >> 2) * 64 + (((v & 3) ^ (((v >> 2) >> 2) & 3)) << 4)) = xr[k];
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int v = t4 + k * NTHR, row = v >> 2;
        *reinterpret_cast<vec*>(sUW + row * 64 + (((v & 3) ^ ((row >> 2) & 3)) << 4)) = xr[3 + k];
      }
      if (t4 < 32) { if (a.up_b) UNCL_CHK(a.chk, a.up_b + t4, 4); sUB[t4] = a.up_b ? a.up_b[t4] : 0.f; }
      __syncthreads();
      // Work split: wave = tap (dy, dx), five M-tiles of 32 source pixels each.  The A rows are read in the order
      // cout(r) = 16*bit2(r) + 4*(r >> 3) + (r & 3), which makes the D registers of a lane 16 CONSECUTIVE output channels
      // (16 lh .. 16 lh + 15) of its pixel: two 16-byte LDS writes per pixel instead of four 8-byte ones.  The bias is the
      // initial accumulator.
      constexpr int MTU = (UPN + 31) / 32;
      const int tap = __builtin_amdgcn_readfirstlane(t4 >> 6);
      const int arow = tap * 32 + (((lr >> 2) & 1) << 4) + ((lr >> 3) << 2) + (lr & 3);
      vec Au[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        Au[ks] = *reinterpret_cast<const vec*>(sUW + arow * 64 + (((2 * ks + lh) ^ ((arow >> 2) & 3)) << 4));
      f32x16 cb;
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        const f32x4 ubv = *reinterpret_cast<const f32x4*>(sUB + 16 * lh + 4 * qd);
#pragma unroll
        for (int e = 0; e < 4; ++e) cb[4 * qd + e] = ubv[e];
      }
      const int iy0h = u_iy0 + (tap >> 1), ix0h = u_ix0 + (tap & 1);
#pragma unroll
      for (int mt = 0; mt < MTU; ++mt) {
        const int sp = mt * 32 + lr, spc = min(sp, UPN - 1);
        const int spy = (spc * 241) >> 12, spx = spc - spy * UPW;     // / 17 for spc < 1024
        f32x16 cu = cb;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const vec Bu = *reinterpret_cast<const vec*>(sU + spc * 64 + (((2 * ks + lh) ^ ((spc >> 2) & 3)) << 4));
          cu = mfma32x16(Au[ks], Bu, cu);
        }
        const bool in_img = (unsigned)(iy0h + 2 * spy) < (unsigned)a.H && (unsigned)(ix0h + 2 * spx) < (unsigned)a.W;
        char* dst = sX + ((2 * spy + (tap >> 1)) * HW + 2 * spx + (tap & 1)) * RP + (lh << 5);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float f[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = cu[8 * h + e];
          vec o = E::pack(f);
          if (!in_img) o = E::zero();
          if (MTU * 32 == UPN || mt < MTU - 1 || sp < UPN) *reinterpret_cast<vec*>(dst + (h << 4)) = o;
        }
      }
      __syncthreads();   // the scratch is in the weight image: everybody is done with it before the new weights land
      // the up-conv chunk is the first of its tile (32-channel skip): the accumulators are dead here, and saying so gives
      // their registers to the block above
#pragma unroll
      for (int m = 0; m < MPW; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[m][nt] = zero16;
    } else
#pragma unroll
    for (int j = 0; j <= RS; ++j) {
      if (j == RS && !e_on) continue;
      vec v = xr[j];
      if ((MODE == 1 || MODE == 4) && g_pending >= 2) {
        float f[8];
        E::unpack(v, f);
        if (g_pending == 2) {
#pragma unroll
          for (int i = 0; i < 8; ++i) f[i] = f[i] * f[i];
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) f[i] = __builtin_amdgcn_sqrtf(f[i] + 1e-8f);
        }
        v = E::pack(f);
      }
      if (!all_ok && !((xvalid >> j) & 1u)) v = E::zero();
      const int pix = j < RS ? pix_r0 + j * 2 * HW : pix_e;
      *reinterpret_cast<vec*>(sX + pix * RP + (ch << 4)) = v;
    }
    if (with_w) {
#pragma unroll
      for (int j = 0; j < WVN; ++j) {
        if (W_RAGGED && j == WVN - 1 && p0 + 64 * j >= WROWS) continue;
        *reinterpret_cast<vec*>(sW + lds_w0 + j * 64 * RP) = wr[j];
      }
    }
    if (b_pending && tid < CT / 4) *reinterpret_cast<f32x4*>(sB + tid * 16) = br;
    // Name every prefetch register as consumed on every path.  The waitcnt insertion is path-insensitive: a register
    // that one path leaves unread (weights of a static layer, the extra-column slot of idle lanes) counts as still in
    // flight at the loop head, and the next step's loads would each wait for the ones issued just before them.
#pragma unroll
    for (int j = 0; j < XV; ++j) asm volatile("" ::"v"(xr[j]));
#pragma unroll
    for (int j = 0; j < WVN; ++j) asm volatile("" ::"v"(wr[j]));
    asm volatile("" ::"v"(br));
    if (MODE == 3) {
#pragma unroll
      for (int k = 0; k < IRN; ++k) asm volatile("" ::"v"(ir[k]));
    }
  };

  // advance the (tile, kc) cursor by one step; returns false past the end of this workgroup's range
  auto advance = [&]() {
    n_n = c_n; n_y0 = c_y0; n_x0 = c_x0; n_co = c_co; n_kc = c_kc + 1;
    if (n_kc < a.nk) return true;
    n_kc = 0;
    if (++tile >= tile_end) return false;
    if (++t_ct == a.n_ct) {
      t_ct = 0;
      if (++t_tx == a.tiles_x) {
        t_tx = 0;
        if (++t_ty == a.tiles_y) { t_ty = 0; ++t_n; }
      }
    }
    n_n = t_n; n_y0 = t_ty * TH; n_x0 = t_tx * TW; n_co = t_ct * CT;
    return true;
  };


  // epilogue addressing that does not change between tiles
  const int sw_e = (lr >> 1) & (SLOTS - 1);                            // swizzle of this lane's pixel column
  const int eo_base = (wave * MPW * TW + lr) * (CT * 2) + (lh << 3);   // + m*TW*CT*2 + ((slot ^ sw_e) << 4)
  const int st_pl = tid / SLOTS, st_sl = tid - st_pl * SLOTS;          // main store: pixel / slot of pass 0
  const int st_row = st_pl / TW, st_col = st_pl - st_row * TW;
  const int st_lds = st_pl * (CT * 2) + ((st_sl ^ ((st_pl >> 1) & (SLOTS - 1))) << 4);

  // ---- MFMA phase over the staged chunk
  auto mfma_phase = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int chunk = 2 * ks + lh;
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        vec A[3][NT], B[MPW + 2];
#pragma unroll
        for (int ty = 0; ty < 3; ++ty)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int row = (ty * 3 + tx) * CT + nt * 32 + lr;
            A[ty][nt] = *reinterpret_cast<const vec*>(sW + row * RP + (chunk << 4));
          }
        if (FLAT) {
          // every lane reads its own 3x3 window: no sharing of rows between the M-tiles
          const int Wp = a.W + 2 * a.pad;
#pragma unroll
          for (int m = 0; m < MPW; ++m) {
            vec Bf[3];
#pragma unroll
            for (int ty = 0; ty < 3; ++ty)
              Bf[ty] = *reinterpret_cast<const vec*>(sX + flat_lane[m] + (ty * Wp + tx) * RP + (chunk << 4));
#pragma unroll
            for (int ty = 0; ty < 3; ++ty)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt)
                acc[m][nt] = mfma32x16(A[ty][nt], Bf[ty], (ks == 0 && tx == 0 && ty == 0 && c_kc == 0) ? zero16 : acc[m][nt]);
          }
          continue;
        }
#pragma unroll
        for (int r = 0; r < MPW + 2; ++r) {
          const int pix = (wave * MPW + r) * HW + lr + tx;
          B[r] = *reinterpret_cast<const vec*>(sX + pix * RP + (chunk << 4));
        }
        if (ks == 0 && tx == 0) {
          // first tap of the chunk: on the first chunk of a tile the accumulation starts from zero (inline C = 0)
          if (MODE != 4 && c_kc == 0) {
#pragma unroll
            for (int m = 0; m < MPW; ++m)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt)
                acc[m][nt] = mfma32x16(A[0][nt], B[m], zero16);
          } else {
#pragma unroll
            for (int m = 0; m < MPW; ++m)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt)
                acc[m][nt] = mfma32x16(A[0][nt], B[m], acc[m][nt]);
          }
#pragma unroll
          for (int m = 0; m < MPW; ++m)
#pragma unroll
            for (int ty = 1; ty < 3; ++ty)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt)
                acc[m][nt] = mfma32x16(A[ty][nt], B[m + ty], acc[m][nt]);
        } else {
#pragma unroll
          for (int m = 0; m < MPW; ++m)
#pragma unroll
            for (int ty = 0; ty < 3; ++ty)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt)
                acc[m][nt] = mfma32x16(A[ty][nt], B[m + ty], acc[m][nt]);
        }
      }
    }
  };

  // register-direct outconv (see the epilogue): three 16-bit pieces of the fp32 weights in rows 0..2 of the A operand.  The two
  // fragments of a lane are parked in LDS (2 KB behind the bias / patch area) and read back per tile: held in registers for the
  // whole launch they push the 166 / 252-register instances over their budget (16 - 72 bytes of scratch per lane)
  // (its own instantiation, O1D: beside the LDS epilogue the extra live values spilled 16 - 72 bytes per lane in sixteen others)
  constexpr bool o1_direct = O1D;
  static_assert(!O1D || (!FLAT && NT == 1), "register-direct 1x1 tail: 32-channel tiles");
  char* const sO1F = smem + a.o1_lds_off;
  if (o1_direct && wave == 0) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      vec f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        UNCL_CHK(a.chk, a.out1_w + 16 * ks + 8 * (j >> 2) + 4 * lh + (j & 3), 4);
        float w = a.out1_w[16 * ks + 8 * (j >> 2) + 4 * lh + (j & 3)];
        T piece = (T)0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const T h = (T)w;
          if (lr == t) piece = h;
          w -= (float)h;
        }
        f[j] = piece;
      }
      *reinterpret_cast<vec*>(sO1F + (ks * 64 + lane) * 16) = f;
    }
  }
  load_regs(c_n, c_y0, c_x0, c_co, c_kc, true);
  write_lds(true);
  __syncthreads();

  PT_DECL
  for (;;) {
    const bool more = advance();
    if (more) load_regs(n_n, n_y0, n_x0, n_co, n_kc, !w_static);
    PT(0)  // cursor + prefetch issue
    // waves in their multiply phase take issue priority over the other workgroup's waves that are staging or storing on
    // the same SIMD: the matrix pipe is the resource to keep fed (same-box A/B: -1.1 % per step, three of three pairs)
    __builtin_amdgcn_s_setprio(3);
    mfma_phase();
    __builtin_amdgcn_s_setprio(0);
    PT(1)  // MFMA phase (LDS fragment reads + matrix pipe)
    // ---- tile finished.  Inference's last layer (only the one-channel map is wanted: outconv + last activation of the ROUNDED,
    // activated features, unet_parts.py:338-345) never needs the 32-channel tile outside the registers: the 32 x 32 accumulator of
    // a row (channel = register, pixel = lane) is the B operand of two MFMAs whose A rows 0..2 hold three 16-bit pieces of the
    // fp32 outconv weights (see outc_row in conv3x3_pc.hip: the same arithmetic, so the two kernels agree bit for bit) -- no LDS
    // transposition, no barrier, no 64 B per pixel read back and 32 fma per pixel on the vector pipe.
    if (c_kc == a.nk - 1 && o1_direct) {
      const vec o1w[2] = {*reinterpret_cast<const vec*>(sO1F + lane * 16), *reinterpret_cast<const vec*>(sO1F + (64 + lane) * 16)};
      UNCL_CHK(a.chk, a.out1_b, 4);
      const float o1b = a.out1_b[0];
#pragma unroll
      for (int m = 0; m < MPW; ++m) {
        vec Bf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          typedef float f32x2o __attribute__((ext_vector_type(2)));
          typedef short s16x8o __attribute__((ext_vector_type(8)));
          vec o;
#pragma unroll
          for (int hq = 0; hq < 2; ++hq) {
            const int q = 2 * ks + hq;
            const f32x4 b = *reinterpret_cast<const f32x4*>(sB + (8 * q + 4 * lh) * 4);
            const f32x2o s0 = f32x2o{acc[m][0][4 * q], acc[m][0][4 * q + 1]} + f32x2o{b[0], b[1]};
            const f32x2o s1 = f32x2o{acc[m][0][4 * q + 2], acc[m][0][4 * q + 3]} + f32x2o{b[2], b[3]};
            o[4 * hq] = (T)s0[0]; o[4 * hq + 1] = (T)s0[1]; o[4 * hq + 2] = (T)s1[0]; o[4 * hq + 3] = (T)s1[1];
          }
          s16x8o si = __builtin_bit_cast(s16x8o, o);
          si = __builtin_elementwise_max(si, s16x8o{0, 0, 0, 0, 0, 0, 0, 0});
          Bf[ks] = __builtin_bit_cast(vec, si);
        }
        f32x16 d = mfma32x16(o1w[0], Bf[0], zero16);
        d = mfma32x16(o1w[1], Bf[1], d);
        const float tot = (d[2] + d[1]) + d[0] + o1b;
        const int oy = c_y0 + wave * MPW + m, ox = c_x0 + lr;
        if (lh == 0 && oy < a.Hout && ox < a.Wout) { UNCL_CHK(a.chk, a.out1 + ((size_t)c_n * a.Hout + oy) * a.Wout + ox, 4); a.out1[((size_t)c_n * a.Hout + oy) * a.Wout + ox] = uncl_act(tot, a.out1_act); }
      }
    } else if (!o1_direct && c_kc == a.nk - 1) {
      __syncthreads();  // every wave is done reading sX / sW
      PT(2)  // barrier: slowest wave's MFMA phase
      // ACT 0: ReLU, applied to the rounded bf16 pair as a signed 16-bit max against zero (one packed op per two
      // values; rounding is sign-symmetric, so relu(round(t)) == round(relu(t))); ACT 1: identity (gradient mode);
      // ACT 2: general max(t,0) + slope*min(t,0).
      auto transpose_out = [&](auto act_tag) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(sB + (nt * 32 + 8 * q + 4 * lh) * 4);
#pragma unroll
            for (int m = 0; m < MPW; ++m) {
              vec4 o;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float t = acc[m][nt][4 * q + r] + b[r];
                o[r] = (T)(ACT == 2 ? fmaxf(t, 0.f) + a.slope * fminf(t, 0.f) : t);
              }
              if (ACT == 0) {
                s16x4 si = __builtin_bit_cast(s16x4, o);
                si = __builtin_elementwise_max(si, s16x4{0, 0, 0, 0});
                o = __builtin_bit_cast(vec4, si);
              }
              *reinterpret_cast<vec4*>(sO + eo_base + m * (TW * CT * 2) + (((nt * 4 + q) ^ sw_e) << 4)) = o;
            }
          }
        }
      };
      if (a.slope == 0.f) transpose_out(IntTag<0>{});
      else if (a.slope == 1.f) transpose_out(IntTag<1>{});
      else transpose_out(IntTag<2>{});
      __syncthreads();
      PT(3)  // bias/activation, transposing LDS writes, barrier
      if (!a.skip_main) {
        const int ox = c_x0 + st_col;
        // FLAT: the tile's pixels are flat_S whole samples, consecutive in memory: a 32-wide image of Hout = 8 rows whose
        // last valid pixel is given by the samples this group really covers (a.Hout / a.Wout are that virtual image)
        const size_t n_px = FLAT ? (size_t)c_n * a.flat_S * a.flat_hw : (size_t)c_n * a.Hout * a.Wout;
        const int flat_valid = FLAT ? min(a.flat_S, a.flat_N - c_n * a.flat_S) * a.flat_hw : 0;
        if (ox < a.Wout && c_y0 + st_row < a.Hout) {
          const size_t pix0 = (size_t)(c_y0 + st_row) * a.Wout + ox;
          bf16_t* ob = a.out + (n_px + pix0) * a.oC + c_co + st_sl * 8;
          // passes with it*ROWS_PER_IT < rows_left are in range
          const int rows_left = FLAT ? ((flat_valid - (int)pix0 + TW * ROWS_PER_IT - 1) / (TW * ROWS_PER_IT)) * ROWS_PER_IT
                                     : a.Hout - (c_y0 + st_row);
          const unsigned row_stride = (unsigned)(ROWS_PER_IT * a.Wout * a.oC);
          // LDS reads of the image in groups of four ahead of their (conditional) stores: one LDS latency per group
          auto VAL = [&](int it) { return *reinterpret_cast<const vec*>(sO + st_lds + it * (NTHR / SLOTS) * (CT * 2)); };
          if (a.res == nullptr && a.mask == nullptr && !a.accumulate) {
#pragma unroll
            for (int it = 0; it < ST_IT; it += 4) {
              const vec v0 = VAL(it), v1 = VAL(it + 1), v2 = VAL(it + 2), v3 = VAL(it + 3);
              if (it * ROWS_PER_IT < rows_left) { UNCL_CHK(a.chk, ob + it * row_stride, 16); *reinterpret_cast<vec*>(ob + it * row_stride) = v0; }
              if ((it + 1) * ROWS_PER_IT < rows_left) { UNCL_CHK(a.chk, ob + (it + 1) * row_stride, 16); *reinterpret_cast<vec*>(ob + (it + 1) * row_stride) = v1; }
              if ((it + 2) * ROWS_PER_IT < rows_left) { UNCL_CHK(a.chk, ob + (it + 2) * row_stride, 16); *reinterpret_cast<vec*>(ob + (it + 2) * row_stride) = v2; }
              if ((it + 3) * ROWS_PER_IT < rows_left) { UNCL_CHK(a.chk, ob + (it + 3) * row_stride, 16); *reinterpret_cast<vec*>(ob + (it + 3) * row_stride) = v3; }
            }
          } else if (a.res == nullptr) {
            // gradient store: ReLU mask of the producing layer and / or accumulation into an existing gradient
            const bf16_t* mb = a.mask ? a.mask + (n_px + pix0) * a.oC + c_co + st_sl * 8 : nullptr;
            // rows past the end are clamped to the last valid one for the loads, so they issue back to back
#pragma unroll
            for (int it = 0; it < ST_IT; ++it) {
              if (it * ROWS_PER_IT < rows_left) {
                float f[8];
                E::unpack(VAL(it), f);
                if (mb) {
                  float m[8];
                  E::unpack(LD16V(vec, mb + it * row_stride), m);
#pragma unroll
                  for (int i = 0; i < 8; ++i) f[i] = m[i] > 0.f ? f[i] : a.mask_slope * f[i];
                }
                if (a.accumulate) {
                  float o[8];
                  E::unpack(LD16V(vec, ob + it * row_stride), o);
#pragma unroll
                  for (int i = 0; i < 8; ++i) f[i] += o[i];
                }
                UNCL_CHK(a.chk, ob + it * row_stride, 16);
                *reinterpret_cast<vec*>(ob + it * row_stride) = E::pack(f);
              }
            }
          } else {
            // residual (pos_embed, Unet_singleFrame.py:94) is added to the stored bf16 features in fp32
            const bf16_t* rb = a.res + ((a.res_b0 ? 0 : n_px) + pix0) * a.oC + c_co + st_sl * 8;
#pragma unroll
            for (int it = 0; it < ST_IT; ++it) {
              if (it * ROWS_PER_IT < rows_left) {
                float f[8], g[8];
                E::unpack(VAL(it), f);
                E::unpack(LD16V(vec, rb + it * row_stride), g);
#pragma unroll
                for (int i = 0; i < 8; ++i) f[i] += g[i];
                UNCL_CHK(a.chk, ob + it * row_stride, 16);
                *reinterpret_cast<vec*>(ob + it * row_stride) = E::pack(f);
              }
            }
          }
        }
      }
      if (!FLAT && a.pool_out != nullptr) {
        bf16_t* pb = a.pool_out + (size_t)c_n * a.pH * a.pW * a.oC + c_co;
        for (int v = tid; v < (TH / 2) * (TW / 2) * SLOTS; v += NTHR) {
          const int pp = v / SLOTS, sl = v - pp * SLOTS;
          const int py = pp / (TW / 2), px = pp - py * (TW / 2);
          const int gy = (c_y0 >> 1) + py, gx = (c_x0 >> 1) + px;
          if (gy < a.pH && gx < a.pW) {
            float m[8];
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
              const int pl = (2 * py + (qd >> 1)) * TW + 2 * px + (qd & 1);
              const vec val = *reinterpret_cast<const vec*>(sO + pl * (CT * 2) + ((sl ^ ((pl >> 1) & (SLOTS - 1))) << 4));
              float f[8];
              E::unpack(val, f);
#pragma unroll
              for (int i = 0; i < 8; ++i) m[i] = qd == 0 ? f[i] : fmaxf(m[i], f[i]);
            }
            UNCL_CHK(a.chk, pb + (unsigned)((gy * a.pW + gx) * a.oC + sl * 8), 16);
            *reinterpret_cast<vec*>(pb + (unsigned)((gy * a.pW + gx) * a.oC + sl * 8)) = E::pack(m);
          }
        }
      }
      if (!FLAT && a.out1_w != nullptr) {
        for (int pl = tid; pl < TH * TW; pl += NTHR) {
          const int prow = pl / TW, pcol = pl - prow * TW;
          const int oy = c_y0 + prow, ox = c_x0 + pcol;
          if (oy < a.Hout && ox < a.Wout) {
            UNCL_CHK(a.chk, a.out1_b, 4);
            float sum = a.out1_b[0];
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
              const vec val = *reinterpret_cast<const vec*>(sO + pl * (CT * 2) + ((sl ^ ((pl >> 1) & (SLOTS - 1))) << 4));
              float f[8];
              E::unpack(val, f);
#pragma unroll
              for (int i = 0; i < 8; ++i) { UNCL_CHK(a.chk, a.out1_w + sl * 8 + i, 4); sum = fmaf(f[i], a.out1_w[sl * 8 + i], sum); }
            }
            UNCL_CHK(a.chk, a.out1 + ((size_t)c_n * a.Hout + oy) * a.Wout + ox, 4);
            a.out1[((size_t)c_n * a.Hout + oy) * a.Wout + ox] = uncl_act(sum, a.out1_act);
          }
        }
      }
    }
    PT(4)  // global stores issued (main, pooled copy, fused 1x1 tail)
    __syncthreads();
    PT(5)  // end-of-step barrier
    // All prefetched registers are needed from here on.  The explicit vmcnt(0) sits BEFORE the exit test on purpose: the
    // compiler routes the exit through the loop latch, and its (path-insensitive) waitcnt insertion would otherwise see a
    // path "loads issued -> staging skipped -> loop head" and make every load of the next step wait for its predecessors.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), lgkmcnt/expcnt untouched
    if (!more) break;
    PT(6)  // wait for the prefetched registers (and everything older in the vmcnt queue)
    write_lds(!w_static);
    PT(7)  // LDS staging writes
    __syncthreads();
    PT(8)  // barrier after staging
    c_n = n_n; c_y0 = n_y0; c_x0 = n_x0; c_co = n_co; c_kc = n_kc;
  }
  PT_FLUSH()
}

template <typename T, int NT, int MPW, int WAVES, int MODE, bool PREV, bool FLAT = false, bool O1D = false>
int launch_pipe(PipeArgs& a, hipStream_t s) {
  constexpr int TH = MPW * WAVES;
  constexpr size_t ximg = (size_t)(TH + 2) * 34 * 80 > (size_t)TH * 32 * NT * 32 * 2 ? (size_t)(TH + 2) * 34 * 80 : (size_t)TH * 32 * NT * 32 * 2;
  constexpr size_t lds0 = ximg + (size_t)9 * NT * 32 * 80 + (size_t)NT * 32 * 4 +
                          (MODE == 3 ? (size_t)((TH + 4) * 36 + 32) * 4 : 0);
  constexpr size_t lds = lds0 + (O1D ? 2048 : 0);      // + the parked outconv fragments (register-direct 1x1 tail)
  a.o1_lds_off = (int)lds0;
  auto kern = conv3x3_pipe_kernel<T, NT, MPW, WAVES, MODE, PREV, FLAT, O1D>;
  static UnclDevOnce attr_done;
  static std::atomic<int> per_cu_s{0};
  if (attr_done.need()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess)
      return UNCL_ERR_LAUNCH;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), WAVES * 64, lds) !=
            hipSuccess || per_cu <= 0)
      per_cu = 1;
    if (per_cu > ((NT == 1 && MPW == 2) ? 3 : 2)) per_cu = (NT == 1 && MPW == 2) ? 3 : 2;
    per_cu_s.store(per_cu, std::memory_order_relaxed);
    attr_done.done();
  }
  const int max_blocks = per_cu_s.load(std::memory_order_relaxed) * uncl_cu_count();
  if (max_blocks <= 0) return UNCL_ERR_LAUNCH;
  int grid = a.total_tiles < max_blocks ? a.total_tiles : max_blocks;
  a.tiles_per_wg = (a.total_tiles + grid - 1) / grid;
  grid = (a.total_tiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, s, a);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

template <typename T, int NT, int MPW>
int dispatch_mode(PipeArgs& a, int mode, bool prev, hipStream_t s) {
  if (mode == UNCL_SRC_IMAGE1) {
    if constexpr (NT == 1 && MPW == 4) return launch_pipe<T, 1, 4, 4, 3, false>(a, s);
    return UNCL_ERR_ARG;
  }
  if (mode == UNCL_SRC_PLAIN) return prev ? launch_pipe<T, NT, MPW, 4, 0, true>(a, s) : launch_pipe<T, NT, MPW, 4, 0, false>(a, s);
  if (mode == UNCL_SRC_CONCAT_SSR) return launch_pipe<T, NT, MPW, 4, 1, false>(a, s);
  if (mode == UNCL_SRC_CONCAT_SSR_UP) {
    if constexpr (NT == 1 && MPW == 4) return launch_pipe<T, 1, 4, 4, 4, false>(a, s);
    return UNCL_ERR_ARG;
  }
  return launch_pipe<T, NT, MPW, 4, 2, false>(a, s);
}

// element type of the launch: bf16 or (inference) fp16
template <int NT, int MPW>
int dispatch_type(PipeArgs& a, int dtype, int mode, bool prev, hipStream_t s) {
  if (dtype == UNCL_F16) return dispatch_mode<f16_t, NT, MPW>(a, mode, prev, s);
  return dispatch_mode<bf16_t, NT, MPW>(a, mode, prev, s);
}

}  // namespace

// 2 (default): every layer the producer / consumer kernel builds faster; 3: also the fused 1x1 tails (slower there); 1: the
// concat-source layers only; 0: the four-wave kernel
#ifndef UNCL_PC_TWO_DEFAULT
#define UNCL_PC_TWO_DEFAULT 0
#endif
static int g_use_pc = [] { const char* e = getenv("UNCL_PC"); return e ? atoi(e) : 2; }();
// flat M-tiles (conv3x3_flat.hip): 0 off, 1 (default) where the cost model prefers them, 2 / 3 / 4: wherever the kernel applies,
// with that many M-tiles per multiplying wave (tests, A/B)
static int g_use_flat = [] { const char* e = getenv("UNCL_FLAT"); return e ? atoi(e) : 1; }();
extern "C" int uncl_conv3x3_set_flat(int on) {
  const int old = g_use_flat;
  g_use_flat = on < 0 ? 0 : (on > 4 ? 4 : on);
  return old;
}
extern "C" int uncl_conv3x3_set_pc(int on) {
  const int old = g_use_pc;
  if (on == -1) return old;          // query: the host packs the fused skip backward's weights only where this structure runs them
  g_use_pc = on < 0 ? 0 : (on > 3 ? 3 : on);
  return old;
}

// Same descriptor as uncl_conv_igemm; handles bf16 3x3 with src_mode PLAIN / CONCAT_SSR / CONCAT2.
// `pool_out` (optional) receives maxpool2x2(out) as NHWC (N, Hout/2, Wout/2, Cout).
struct SsrBwd { const void* x2; void* g_x2; void* g_x1; int acc; };
static int conv3x3_pipe_impl(const uncl_conv_desc* d, void* pool_out, const void* mask, float mask_slope, int accumulate,
                             void* stream, const SsrBwd* ssr = nullptr) {
  if (d == nullptr || !uncl_is_h16(d->dtype) || d->ksize != 3) return UNCL_ERR_ARG;
  if (d->dtype == UNCL_F16 && (mask != nullptr || accumulate)) return UNCL_ERR_ARG;    // fp16: forward only
  if (d->pad != 0 && d->pad != 2) return UNCL_ERR_ARG;
  if (d->src_mode == UNCL_SRC_MAXPOOL2 || d->src_mode < 0 || d->src_mode > UNCL_SRC_CONCAT_SSR_UP) return UNCL_ERR_ARG;
  if (d->src_mode == UNCL_SRC_IMAGE1) {
    // x = act(conv3x3_valid(image)): 32 channels over (H, W) = image extent - 2; this layer itself must be 32 -> 32, valid
    if (d->Cin != 32 || d->Cout != 32 || d->pad != 0 || d->pre_w == nullptr || d->src0_C != 1 || d->src0_H != d->H + 2 ||
        d->src0_W != d->W + 2 || d->prev0 != nullptr || d->res != nullptr || mask != nullptr)
      return UNCL_ERR_ARG;
  }
  if (d->Cin <= 0 || d->Cin % 32 != 0 || d->Cout <= 0 || d->Cout % 32 != 0) return UNCL_ERR_ARG;
  if (d->z_mode != UNCL_Z_NONE || d->scale_n != nullptr) return UNCL_ERR_ARG;
  if (d->act != UNCL_ACT_NONE && d->act != UNCL_ACT_RELU && d->act != UNCL_ACT_LRELU) return UNCL_ERR_ARG;
  if (d->src0 == nullptr || d->weight == nullptr) return UNCL_ERR_ARG;
  if (d->out == nullptr && !(d->skip_main_store && d->out1 != nullptr) && ssr == nullptr) return UNCL_ERR_ARG;
  if (d->src_mode == UNCL_SRC_CONCAT_SSR || d->src_mode == UNCL_SRC_CONCAT2) {
    if (d->src1 == nullptr || d->src0_C != d->src1_C || d->src0_C % 32 != 0 || d->prev0 != nullptr) return UNCL_ERR_ARG;
    const int groups = d->src_mode == UNCL_SRC_CONCAT_SSR ? 4 : 2;
    if (d->Cin != groups * d->src0_C) return UNCL_ERR_ARG;
    if (d->src1_H > d->src0_H || d->src1_W > d->src0_W) return UNCL_ERR_ARG;
  }
  if (d->src_mode == UNCL_SRC_CONCAT_SSR_UP) {
    // src1 is the INPUT of the 2x2 stride-2 transposed conv: 32 (or, producer / consumer kernel only, 64) channels in and out, its
    // output must have exactly the skip's extent
    if (d->src1 == nullptr || d->up_w == nullptr || (d->src0_C != 32 && d->src0_C != 64) || d->src1_C != d->src0_C ||
        d->Cin != 4 * d->src0_C || d->Cout != 32 || 2 * d->src1_H != d->src0_H || 2 * d->src1_W != d->src0_W ||
        d->prev0 != nullptr || mask != nullptr)
      return UNCL_ERR_ARG;
    if (d->src0_C == 64 && d->tail_w != nullptr) return UNCL_ERR_ARG;
  }
  if (d->out1_w != nullptr && (d->Cout != 32 || d->out1 == nullptr)) return UNCL_ERR_ARG;
  // 32-bit element offsets inside one sample
  if ((long long)d->src0_H * d->src0_W * d->src0_C >= (1LL << 31)) return UNCL_ERR_ARG;
  PipeArgs a;
  a.src0 = (const bf16_t*)d->src0; a.src1 = (const bf16_t*)d->src1; a.prev0 = (const bf16_t*)d->prev0;
  a.weight = (const bf16_t*)d->weight; a.bias = d->bias; a.res = (const bf16_t*)d->res;
  a.mask = (const bf16_t*)mask; a.accumulate = accumulate; a.mask_slope = mask_slope;
  a.out = (bf16_t*)d->out; a.pool_out = (bf16_t*)pool_out; a.out1_w = d->out1_w; a.out1_b = d->out1_b; a.out1 = d->out1;
  a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.pad = d->pad;
  a.s0H = d->src0_H; a.s0W = d->src0_W; a.s0C = d->src0_C;
  a.s1H = d->src1_H; a.s1W = d->src1_W; a.s1C = d->src1_C; a.prev_ch = d->prev_ch;
  a.slope = d->act == UNCL_ACT_RELU ? 0.f : (d->act == UNCL_ACT_LRELU ? 0.2f : 1.f);
  if (mask != nullptr) {
    if (d->act != UNCL_ACT_NONE || d->res != nullptr) return UNCL_ERR_ARG;  // gradient mode: identity epilogue
  }
  a.res_b0 = d->res_batch_stride0;
  a.Hout = d->H + 2 * d->pad - 2; a.Wout = d->W + 2 * d->pad - 2; a.oC = d->out_C;
  if (a.Hout <= 0 || a.Wout <= 0) return UNCL_ERR_ARG;
  if (d->out != nullptr && (d->out_H != a.Hout || d->out_W != a.Wout)) return UNCL_ERR_ARG;
  a.pH = a.Hout / 2; a.pW = a.Wout / 2;
  a.out1_act = d->out1_act; a.skip_main = d->skip_main_store;
  a.img = nullptr; a.pre_w = nullptr; a.pre_b = nullptr; a.imgH = a.imgW = 0; a.img_off = nullptr; a.imgP = 0;
  if (d->src_mode == UNCL_SRC_IMAGE1) {
    a.img = (const float*)d->src0; a.pre_w = d->pre_w; a.pre_b = d->pre_b; a.imgH = d->src0_H; a.imgW = d->src0_W;
    // tiles cut out of larger frames (the tiler's gather folded into this loader): src1 = DEVICE int32[N] element offsets of the
    // tiles' top-left pixels inside src0, src1_W = the frames' row pitch
    a.img_off = (const int*)d->src1; a.imgP = d->src1 != nullptr ? d->src1_W : d->src0_W;
    if (d->src1 != nullptr && d->src1_W < d->src0_W) return UNCL_ERR_ARG;
    a.s0H = d->H; a.s0W = d->W; a.s0C = 32;
  }
  a.up_w = (const bf16_t*)d->up_w; a.up_b = d->up_b;
  a.flat_S = 0; a.flat_hw = 0; a.flat_N = d->N;
  a.nk = d->Cin / 32;
  a.tail_w = (const bf16_t*)d->tail_w; a.tail_b = d->tail_b; a.oH = a.oW = 0;
  a.epo2 = 0;
  a.ssr_x2 = nullptr; a.ssr_gx2 = nullptr; a.ssr_gx1 = nullptr; a.ssr_C = 0; a.ssr_acc = 0;
  if (ssr != nullptr) {
    // data gradient of a skip-concat layer, skip operator's backward in the epilogue: plain source, 4 C output channels in the
    // interleaved order, bf16, identity activation; only the producer / consumer kernel with 64-channel tiles builds it
    if (d->dtype != UNCL_BF16 || d->src_mode != UNCL_SRC_PLAIN || d->Cout % 64 != 0 || d->act != UNCL_ACT_NONE || d->res != nullptr ||
        d->prev0 != nullptr || pool_out != nullptr || mask != nullptr || accumulate || d->out1_w != nullptr || !ssr->x2 || !ssr->g_x2 ||
        !ssr->g_x1 || !g_use_pc)
      return UNCL_ERR_ARG;
    a.ssr_x2 = (const bf16_t*)ssr->x2; a.ssr_gx2 = (bf16_t*)ssr->g_x2; a.ssr_gx1 = (bf16_t*)ssr->g_x1;
    a.ssr_C = d->Cout / 4; a.ssr_acc = ssr->acc;
    a.mask_slope = mask_slope;
    if ((size_t)a.Hout * a.Wout * a.ssr_C * 2 >= (1u << 30)) return UNCL_ERR_ARG;
  }
#ifdef UNCL_CHECKED
  {
    // the tensors this launch may touch, from the descriptor's own dimensions (16-bit elements unless stated)
    const unsigned long long es = 2, N = (unsigned long long)d->N;
    uncl_chk_reset(a.chk);
    if (d->src_mode == UNCL_SRC_IMAGE1 && d->src1 != nullptr) {          // tiles inside frames: src1_H rows of src1_W fp32 pixels, the offset table
      uncl_chk_add(a.chk, d->src0, (unsigned long long)d->src1_H * d->src1_W * 4);
      uncl_chk_add(a.chk, d->src1, N * 4);
    } else if (d->src_mode == UNCL_SRC_IMAGE1) {
      uncl_chk_add(a.chk, d->src0, N * d->src0_H * d->src0_W * 4);          // fp32 image
    } else {
      uncl_chk_add(a.chk, d->src0, N * d->src0_H * d->src0_W * d->src0_C * es);
      uncl_chk_add(a.chk, d->src1, N * d->src1_H * d->src1_W * d->src1_C * es);
    }
    uncl_chk_add(a.chk, d->prev0, N * d->src0_H * d->src0_W * d->src0_C * es);
    uncl_chk_add(a.chk, d->weight, 9ull * d->Cout * d->Cin * es);
    uncl_chk_add(a.chk, d->bias, (unsigned long long)d->Cout * 4);
    const unsigned long long out_px = (unsigned long long)a.Hout * a.Wout;
    uncl_chk_add(a.chk, d->res, (d->res_batch_stride0 ? 1ull : N) * out_px * d->out_C * es);
    uncl_chk_add(a.chk, mask, N * out_px * d->out_C * es);
    uncl_chk_add(a.chk, d->out, N * out_px * d->out_C * es);
    uncl_chk_add(a.chk, pool_out, N * (unsigned long long)a.pH * a.pW * d->out_C * es);
    uncl_chk_add(a.chk, d->out1_w, 32 * 4);
    uncl_chk_add(a.chk, d->out1_b, 4);
    uncl_chk_add(a.chk, d->out1, N * (d->tail_w != nullptr ? (unsigned long long)(a.Hout + 2) * (a.Wout + 2) : out_px) * 4);
    uncl_chk_add(a.chk, d->pre_w, 288 * 4);
    uncl_chk_add(a.chk, d->pre_b, 32 * 4);
    uncl_chk_add(a.chk, d->up_w, 4ull * (d->src1_C > 0 ? d->src1_C : 32) * (d->src1_C > 0 ? d->src1_C : 32) * es);
    uncl_chk_add(a.chk, d->up_b, (unsigned long long)(d->src_mode == UNCL_SRC_CONCAT_SSR_UP ? d->src1_C : 32) * 4);
    uncl_chk_add(a.chk, d->tail_w, 9ull * 32 * 32 * es);
    uncl_chk_add(a.chk, d->tail_b, 32 * 4);
    if (ssr != nullptr) {
      const unsigned long long sb = N * out_px * (unsigned long long)(d->Cout / 4) * es;
      uncl_chk_add(a.chk, ssr->x2, sb); uncl_chk_add(a.chk, ssr->g_x2, sb); uncl_chk_add(a.chk, ssr->g_x1, sb);
    }
  }
#endif
  if (d->tail_w != nullptr) {
    // fused last decoder stage: this layer's output and the next layer's never leave the CU; only out1 is written
    if (d->src_mode != UNCL_SRC_CONCAT_SSR_UP || d->act != UNCL_ACT_RELU || !d->skip_main_store || d->out1_w == nullptr ||
        d->out1_b == nullptr || d->out1 == nullptr || pool_out != nullptr || mask != nullptr || accumulate || d->res != nullptr)
      return UNCL_ERR_ARG;
    return uncl_conv3x3_tail_launch(a, d->dtype, reinterpret_cast<hipStream_t>(stream));
  }
  const bool prev = d->prev0 != nullptr && d->prev_ch > 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // multi-chunk layers: producer / consumer workgroups (conv3x3_pc.hip); everything it does not build falls through
  const int pc_mode = d->src_mode == UNCL_SRC_PLAIN ? (prev ? -1 : 0)
                      : d->src_mode == UNCL_SRC_CONCAT_SSR ? 1 : d->src_mode == UNCL_SRC_CONCAT_SSR_UP ? (d->src0_C == 64 ? 5 : 4)
                      : d->src_mode == UNCL_SRC_IMAGE1 ? 3 : -1;
  // measured per layer at bench size (tools/pc_phase_timing.py --product): with one staging register set, unpadded LDS planes
  // and resident weights the producer / consumer structure is faster on every layer it builds (3 - 30 %), single-chunk ones
  // included; UNCL_PC=1 restricts it to the concat layers, 0 turns it off
  // A fused 1x1 tail / skipped main store stays on the four-wave kernel: the producer / consumer epilogue builds it too
  // (bit-identical, set_pc(3): A/B and tests) but its channel-order chain crosses the half-waves three times per row with
  // nobody to hide the exchanges behind -- 0.43 against 0.32 ms on up_path.3.conv.conv1.
  const bool pc_ok = g_use_pc && pc_mode >= (g_use_pc >= 2 ? 0 : 1) && a.nk >= 1 && d->res == nullptr &&
                     ((d->out1_w == nullptr && !d->skip_main_store) || (g_use_pc == 3 && d->Cout == 32 && d->out1_b != nullptr));
  if (d->Cout == 32) {
    // Round 6: inference's last layer (plain 32 -> 32 source, only the one-channel map wanted) on the producer / consumer structure
    // with the 1x1 tail straight from the accumulators (conv3x3_pc.hip, O1C): 16-row tiles, resident weights, eight staging waves
    // that do nothing but load -- the four-wave form below reads its 10 x 34 halo per 8 x 32 outputs through waves that also
    // multiply.  Same arithmetic per output (outc_row): the two forms agree to fp32 reassociation of the three partial dot products.
    static const int o1c_on = [] { const char* e = getenv("UNCL_PC_O1C"); return e ? atoi(e) : 1; }();      // 0: the four-wave form (A/B)
    if (o1c_on && g_use_pc >= 2 && pc_mode == 0 && a.nk == 1 && d->out1_w != nullptr && d->skip_main_store && d->out1_b != nullptr &&
        d->out1 != nullptr && a.slope == 0.f && d->res == nullptr && pool_out == nullptr && mask == nullptr && !accumulate) {
      PipeArgs b = a;
      b.n_ct = 1;
      b.tiles_x = (a.Wout + 31) / 32; b.tiles_y = (a.Hout + 15) / 16;
      b.total_tiles = d->N * b.tiles_x * b.tiles_y;
      const int rc = uncl_conv3x3_pc_launch(b, d->dtype, 1, 4, 0, s);
      if (rc != UNCL_ERR_ARG) return rc;
    }
    // 8-row tiles at three workgroups per CU (50 KB LDS, 168 VGPRs) overlap the serial load / stage / store phases of the
    // single-chunk 32 -> 32 transposed layers better than 16-row tiles at two (measured: up_path.{2,3}.conv.conv1 -9 %);
    // the valid 32 -> 32 layer and the multi-chunk concat layers are faster (or read less) with the larger tile
    if (d->pad == 2 && d->src_mode == UNCL_SRC_PLAIN && d->Cin == 32 && !prev && !(pc_ok && pool_out == nullptr)) {
      a.n_ct = 1;
      a.tiles_x = (a.Wout + 31) / 32; a.tiles_y = (a.Hout + 7) / 8;
      a.total_tiles = d->N * a.tiles_x * a.tiles_y;
      // inference's last layer (only the one-channel map is wanted): the 1x1 tail straight from the accumulators
      if (d->out1_w != nullptr && d->skip_main_store && d->out1_b != nullptr && a.slope == 0.f && d->res == nullptr &&
          pool_out == nullptr && mask == nullptr && !accumulate)
        return d->dtype == UNCL_F16 ? launch_pipe<f16_t, 1, 2, 4, 0, false, false, true>(a, s)
                                    : launch_pipe<bf16_t, 1, 2, 4, 0, false, false, true>(a, s);
      return dispatch_type<1, 2>(a, d->dtype, d->src_mode, prev, s);
    }
    constexpr int TH = 16;
    a.n_ct = 1;
    a.tiles_x = (a.Wout + 31) / 32; a.tiles_y = (a.Hout + TH - 1) / TH;
    a.total_tiles = d->N * a.tiles_x * a.tiles_y;
    if (pc_ok) {
      // single-chunk layers (Cin = 32, plain or rebuilt-from-the-image source): 8-row tiles at two workgroups per CU
      static const int two = [] { const char* e = getenv("UNCL_PC_TWO"); return e ? atoi(e) : UNCL_PC_TWO_DEFAULT; }();
      if (two && a.nk == 1 && (pc_mode == 0 || pc_mode == 3) && d->out1_w == nullptr && !d->skip_main_store) {
        PipeArgs b = a;
        b.tiles_y = (a.Hout + 7) / 8;
        b.total_tiles = d->N * b.tiles_x * b.tiles_y;
        const int rc = uncl_conv3x3_pc_launch(b, d->dtype, 1, 2, pc_mode, s);
        if (rc != UNCL_ERR_ARG) return rc;
      }
      // the fused first layer: 12-row tiles with the epilogue parked for the staging waves (conv3x3_pc.hip, EPO)
      static const int epo12 = [] { const char* e = getenv("UNCL_PC_EPO12"); return e ? atoi(e) : 1; }();
      if (epo12 && pc_mode == 3 && a.nk == 1 && d->out1_w == nullptr && !d->skip_main_store) {
        PipeArgs b = a;
        b.tiles_y = (a.Hout + 11) / 12;
        b.total_tiles = d->N * b.tiles_x * b.tiles_y;
        const int rc = uncl_conv3x3_pc_launch(b, d->dtype, 1, 3, pc_mode, s);
        if (rc != UNCL_ERR_ARG) return rc;
      }
      // Round 6, built and measured, OFF (UNCL_PC_EPO4=1 turns it on): the concat layer with the fused 32-channel up-conv (inference's
      // dominant launch) on 12-row tiles with the epilogue parked for the staging waves (conv3x3_pc.hip, EPO with one buffer).  Its
      // multiplying waves spend 19 % of their time storing, but its staging waves are 64 % busy already: with the parked tile to
      // store and 14 halo rows per 12 outputs they become what the launch waits for -- same-box A/B 0.772 -> 0.831 ms per 200 tiles
      static const int epo4 = [] { const char* e = getenv("UNCL_PC_EPO4"); return e ? atoi(e) : 0; }();
      if (epo4 && pc_mode == 4 && a.nk == 4 && d->out1_w == nullptr && !d->skip_main_store && pool_out == nullptr) {
        PipeArgs b = a;
        b.tiles_y = (a.Hout + 11) / 12;
        b.total_tiles = d->N * b.tiles_x * b.tiles_y;
        const int rc = uncl_conv3x3_pc_launch(b, d->dtype, 1, 3, pc_mode, s);
        if (rc != UNCL_ERR_ARG) return rc;
      }
      const int rc = uncl_conv3x3_pc_launch(a, d->dtype, 1, 4, pc_mode, s);
      if (rc != UNCL_ERR_ARG) return rc;
    }
    if (pc_mode == 5) return UNCL_ERR_ARG;        // the 64-channel fused up-conv exists in the producer / consumer kernel only
    return dispatch_type<1, 4>(a, d->dtype, d->src_mode, prev, s);
  }
  if (d->Cout % 64 != 0) return UNCL_ERR_ARG;
  constexpr int TH = 8;
  a.n_ct = d->Cout / 64;
  a.flat_S = 0; a.flat_hw = 0; a.flat_N = d->N;
  {
    // small maps (the 10 x 10 / 12 x 12 bottleneck levels): whole samples per tile instead of a mostly empty rectangle
    const int howo = a.Hout * a.Wout, hpwp = (d->H + 2 * d->pad) * (d->W + 2 * d->pad);
    int S = howo > 0 ? 256 / howo : 0;
    while (S > 1 && S * hpwp > 10 * 34) --S;
    const bool plain_store = pool_out == nullptr && d->out1_w == nullptr && !d->skip_main_store;   // gradient stores are fine
    if (S >= 1 && S * hpwp <= 10 * 34 && d->src_mode == UNCL_SRC_PLAIN && !prev && plain_store && ssr == nullptr &&
        (d->res == nullptr || !d->res_batch_stride0 || S == 1)) {
      // Round 5: the flat M-tiles of conv3x3_flat.hip cut these maps finer (a 12 x 12 output fills 144 of its 192 slots against
      // 144 of 256 here) and run them on the producer / consumer structure; same cost figure, the four-wave kernel's step
      // priced UNCL_FLAT_SMALL per cent dearer (it stages and multiplies in turn); 0 = keep whole-sample tiles
      static const int small_pct = [] { const char* e = getenv("UNCL_FLAT_SMALL"); return e ? atoi(e) : 125; }();
      if (g_use_flat && small_pct > 0 && pc_ok && pc_mode == 0) {
        const int n_cu = uncl_cu_count();
        const long long steps = (long long)((d->N + S - 1) / S) * a.n_ct;
        const double cost = n_cu > 0 ? (double)((steps + n_cu - 1) / n_cu) * 2.35 * small_pct / 100. : 0.;
        PipeArgs b = a;
        const int rc = uncl_conv3x3_flat_launch(b, d->dtype, pc_mode, g_use_flat >= 2 ? g_use_flat : 0, g_use_flat >= 2 ? 0. : cost, s);
        if (rc != UNCL_ERR_ARG) return rc;
      }
      a.flat_S = S; a.flat_hw = howo;
      a.Hout = 8; a.Wout = 32;      // the store loop's view of the tile: 256 consecutive pixels
      a.tiles_x = a.tiles_y = 1;
      a.total_tiles = ((d->N + S - 1) / S) * a.n_ct;
      return d->dtype == UNCL_F16 ? launch_pipe<f16_t, 2, 2, 4, 0, false, true>(a, s) : launch_pipe<bf16_t, 2, 2, 4, 0, false, true>(a, s);
    }
  }
  a.tiles_x = (a.Wout + 31) / 32; a.tiles_y = (a.Hout + TH - 1) / TH;
  a.total_tiles = d->N * a.tiles_x * a.tiles_y * a.n_ct;
  if (pc_ok) {
    // 16-row tiles (16 x 32 pixels x 64 channels, 128 accumulator registers) halve the weight re-streaming per output of the
    // 8-row tile: same-box A/B -4 ... -14 % per layer wherever they cover the map without extra rows (24 x 24 maps: +7 %)
    static const int tall = [] { const char* e = getenv("UNCL_PC_TALL"); return e ? atoi(e) : 1; }();
    // ... and give every CU a tile (the video step's 8-sample launches: the halved tile count would idle CUs, +4 % there;
    // thresholds of 128 ... 384 tiles measured alike, 768 cost the two-part inference forward 2 %)
    const int tall_tiles = d->N * a.tiles_x * ((a.Hout + 15) / 16) * a.n_ct;
    static const int tall_min = [] { const char* e = getenv("UNCL_PC_TALL_MIN"); return e ? atoi(e) : 256; }();
    const bool go_tall = tall && ((a.Hout + 15) / 16) * 16 <= ((a.Hout + 7) / 8) * 8 && tall_tiles >= tall_min;
    // Round 6, built and measured, OFF (UNCL_PC_EPO2=1 turns it on): one cout tile, at most two K-chunks, forward store (the two layers
    // of down_path.0) on 8-row tiles with the epilogue PARKED for the staging waves.  On 16-row tiles their multiplying waves spend 41 -
    // 52 % of a tile's time issuing its 64 KB of stores beside staging waves that idle 70 - 80 % (tools/pc_phase_timing.py --layers
    // d0a,d0b) -- but what they wait for is HBM, not the issue slot: the two launches move 0.62 / 0.92 GB at 4.4 / 4.0 TB/s as they
    // are, and the parked form (bit-identical, every conv test green with it on) took 0.154 / 0.250 ms against 0.142 / 0.228
    static const int epo2 = [] { const char* e = getenv("UNCL_PC_EPO2"); return e ? atoi(e) : 0; }();
    if (epo2 && pc_mode == 0 && a.n_ct == 1 && a.nk <= 2 && a.Hout >= 100 && ssr == nullptr && mask == nullptr && !accumulate &&
        a.slope == 0.f && d->out1_w == nullptr && !d->skip_main_store) {
      PipeArgs b = a;
      b.epo2 = 1;
      const int rc = uncl_conv3x3_pc_launch(b, d->dtype, 2, 2, pc_mode, s);
      if (rc != UNCL_ERR_ARG) return rc;
    }
    if (g_use_flat && (pc_mode == 0 || pc_mode == 1) && pool_out == nullptr && ssr == nullptr) {
      // flat M-tiles (conv3x3_flat.hip) where the rectangles above leave a large part of their pixels outside the map: the 24 ..
      // 61-pixel levels.  Chosen by the same cost figure for both tilings: rounds of the persistent grid x (M-tiles per wave +
      // 0.35); g_use_flat 2 / 3 / 4 (tests) forces that many M-tiles per wave wherever the kernel applies
      const int n_cu = uncl_cu_count();
      const long long rect_steps = go_tall ? tall_tiles : a.total_tiles;
      const double rect_cost = n_cu > 0 ? (double)((rect_steps + n_cu - 1) / n_cu) * ((go_tall ? 4 : 2) + 0.35) : 0.;
      PipeArgs b = a;
      const int rc = uncl_conv3x3_flat_launch(b, d->dtype, pc_mode, g_use_flat >= 2 ? g_use_flat : 0,
                                              g_use_flat >= 2 ? 0. : 0.97 * rect_cost, s);
      if (rc != UNCL_ERR_ARG) return rc;
    }
    if (go_tall) {
      PipeArgs b = a;
      b.tiles_y = (a.Hout + 15) / 16;
      b.total_tiles = tall_tiles;
      const int rc = uncl_conv3x3_pc_launch(b, d->dtype, 2, 4, pc_mode, s);
      if (rc != UNCL_ERR_ARG) return rc;
    }
    const int rc = uncl_conv3x3_pc_launch(a, d->dtype, 2, 2, pc_mode, s);
    if (rc != UNCL_ERR_ARG) return rc;
  }
  if (ssr != nullptr) return UNCL_ERR_ARG;       // (the four-wave kernel has no such epilogue)
  return dispatch_type<2, 2>(a, d->dtype, d->src_mode, prev, s);
}

#ifdef UNCL_PIPE_TIMING
// measurement builds only: copy (and optionally clear) the per-phase cycle counters
extern "C" int uncl_pipe_timing_read(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_pipe_t), sizeof(unsigned long long) * 16) != hipSuccess) return UNCL_ERR_LAUNCH;
  if (reset) {
    unsigned long long z[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_pipe_t), z, sizeof(z)) != hipSuccess) return UNCL_ERR_LAUNCH;
  }
  return UNCL_OK;
}
#endif

extern "C" int uncl_conv3x3_pipe(const uncl_conv_desc* d, void* pool_out, void* stream) {
  return conv3x3_pipe_impl(d, pool_out, nullptr, 0.f, 0, stream);
}

// Data gradient of a 3x3 layer = the same kernel over re-packed weights (see uncl_pack_conv_weight), identity
// activation; the stored gradient is multiplied by the activation derivative of the layer that produced the tensor it
// flows into (mask > 0 ? 1 : mask_slope) and optionally added to the gradient already there (skip connections).
extern "C" int uncl_conv3x3_dgrad(const uncl_conv_desc* d, const void* mask, float mask_slope, int accumulate, void* stream) {
  return conv3x3_pipe_impl(d, nullptr, mask, mask_slope, accumulate, stream);
}

extern "C" int uncl_conv3x3_dgrad_ssr(const uncl_conv_desc* d, const void* x2, void* g_x2, void* g_x1, float slope, int accumulate_x2,
                                      void* stream) {
  if (d == nullptr || d->Cout % 64 != 0) return UNCL_ERR_ARG;
  const SsrBwd ssr = {x2, g_x2, g_x1, accumulate_x2};
  return conv3x3_pipe_impl(d, nullptr, nullptr, slope, 0, stream, &ssr);
}
