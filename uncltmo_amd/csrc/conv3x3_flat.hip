// Producer / consumer 3x3 implicit-GEMM convolution over FLAT M-tiles (gfx950, bf16 / fp16): the 24 .. 61-pixel levels of the
// generator (unet_parts.py:56-87 `double_conv`, :98-112 / :149-162 the transposed pairs, :311-332 the skip operator).
//
// conv3x3_pc.hip tiles a sample into rectangles of 8 / 16 rows x 32 pixels; a 26 x 26 map fills 66 % of its rectangles, 24 x 24
// 75 %, 57 x 57 79 %, 59 x 59 85 % -- a fifth of the forward's matrix instructions multiplied padding (round 4: 1.20 x executed /
// algorithmic).  Here the output pixels of a sample are LINEARISED on the pitch of the padded input, P = Wout + 2:
//
//     q = y * P + x          (x < P: the two columns x >= Wout of every row are computed and dropped)
//     tap (ty, tx) of output q reads padded-input pixel q + ty * P + tx      -- a CONSTANT offset, whatever the lane's (y, x)
//
// and cut into M-tiles of 32 consecutive q.  A sample is MTS = ceil(Hout * P / 32) M-tiles, the batch N * MTS of them in one
// list; a workgroup tile is 4 * MPW consecutive M-tiles of that list (4 multiplying waves x MPW), wherever sample borders fall.
// Fill: 26 x 26 92 %, 24 x 24 90 %, 57 x 57 95.8 %, 59 x 59 96.3 %.
//
// The staged image of a tile is the range of the batch's padded-input pixels its M-tiles read, in "virtual" coordinates
// V = n * VS + p (p = padded-linear pixel of sample n, VS = 32 MTS + HALO, HALO = 2 P + 2): an M-tile's 32 + HALO pixels are
// contiguous in V, M-tile j of the tile starts at slot 32 j + c_j HALO (c_j = sample borders before it), so a fragment address is
// "per-(M-tile, tap row) base register + immediate" and the 3 x 3 window never needs a per-lane division.  What it costs: no
// row sharing between M-tiles -- nine B-fragment reads per M-tile and 16 K instead of 4.5 (16 rows per wave) -- paid from an LDS
// array that was 44 % busy; columns are therefore single taps (NT + MPW reads under NT * MPW MFMAs) instead of tap columns.
// Same accumulation order per output element as conv3x3_pc / conv3x3_pipe (chunk, K-half, tx, ty): bit-identical results
// (tests/test_gpu_conv.py::test_flat_*_bitwise).
//
// Roles, barriers, LDS planes, resident weights, one staging register set with two-chunk look-ahead, the concat order
// [x1, sqrt, x2^2, x2]: as in conv3x3_pc.hip.  What is new on the staging side: a slot's source address
// is not "tile origin + constant" but (sample, row, column) of its V coordinate, computed once per tile (one reciprocal multiply
// per slot) and reused for every K-chunk and both channel tiles of that tile.
#include <cstdlib>

#include "conv3x3_args.h"

// Ablations (measurement builds only: tools/ab_variants.sh <name> "-DUNCL_FL_ABL_MASK=<bits>" with FILES=conv3x3_flat; compile-time,
// so that the product's code is otherwise unchanged; WRONG results): 16 no activation loads, 32 no weight loads, 64 no output stores,
// 128 no MFMAs (fragments still read), 256 no activation staging writes, 512 no square / square-root transforms, 1024 no weight
// staging writes, 2048 no slot-offset arithmetic
#ifndef UNCL_FL_ABL_MASK
#define UNCL_FL_ABL_MASK 0
#endif
#define FL_ABL(bit) (((UNCL_FL_ABL_MASK) & (bit)) != 0)
// streamed weights: 1 = two register sets, a chunk's weights are requested TWO iterations before they are staged (one set: one)
#ifndef UNCL_FL_PRIO_DEFAULT
#define UNCL_FL_PRIO_DEFAULT 1
#endif
#ifndef UNCL_FL_W2
#define UNCL_FL_W2 0
#endif

namespace {

// Phase timing (measurement builds only, -DUNCL_FL_TIMING; tools/fl_phase_timing.py): wave 0 (multiplying) and wave 4 (staging) of
// every workgroup accumulate s_memtime deltas per loop phase; the product library compiles all of this away.
#ifdef UNCL_FL_TIMING
__device__ unsigned long long g_fl_t[48];
__device__ __forceinline__ unsigned long long flt_now() {
  unsigned long long t;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define FLT_DECL unsigned long long flt_acc[16] = {}; unsigned long long flt_last = flt_now();
#define FLT(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long flt_t = flt_now(); __builtin_amdgcn_sched_barrier(0); \
                 flt_acc[i] += flt_t - flt_last; flt_last = flt_t; }
#define FLT_FLUSH(base, cnt) if (lane == 0) { for (int i = 0; i < 16; ++i) atomicAdd(&g_fl_t[base + i], flt_acc[i]); atomicAdd(&g_fl_t[cnt], 1ull); }
#else
#define FLT_DECL
#define FLT(i)
#define FLT_FLUSH(base, cnt)
#endif

__device__ __forceinline__ void fl_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// plane lengths: 6 (mod 8) slots, see conv3x3_pc.hip (ds_write_b128 is banked (address / 4) mod 32)
constexpr int fl_plane(int slots) { return (slots + ((6 - slots % 8) + 8) % 8) * 16; }

// slots of one staged activation plane: what the 160 KB leave next to the weights (two streamed stages of one chunk, or up to
// 4 / NT resident chunks -- the same 74.5 KB for 64-channel tiles) and the bias ring
constexpr int fl_wplane(int nt) { return fl_plane(9 * nt * 32); }
constexpr int fl_lmax(int nt, bool resw) {
  const int wbytes = (resw ? 4 / nt : 2) * 4 * fl_wplane(nt);
  const int left = 163840 - wbytes - 4 * nt * 32 * 4 - 256;
  int slots = left / (2 * 4 * 16);
  slots -= (slots - 6 + 8) % 8;       // largest count = 6 (mod 8) that fits
  return slots;
}

// sample of a global M-tile index: floor(g / MTS) by reciprocal multiply (exact below 2^24, checked by the launcher); MTS = 1 has no
// 32-bit reciprocal (2^32 / 1 + 1 wraps) and needs none
__device__ __forceinline__ int fl_sample_of(const PipeArgs& a, int g) {
  return a.fl_mts == 1 ? g : (int)__umulhi((unsigned)g, a.fl_div_mts);
}

template <typename T, int NT, int MPW, int MODE, int PW, bool RESW, bool GRAD>
__global__ __launch_bounds__((4 + PW) * 64, PW == 8 ? 3 : 2) void conv3x3_fl_kernel(const PipeArgs a) {
  constexpr int NCW = 4;
  static_assert(PW == 4 || PW == 8, "four or eight staging waves");
  static_assert(MODE == 0 || MODE == 1, "plain or concat-ssr source");
  using E = Elem<T>;
  using vec = typename Elem<T>::vec;
  using vec4 = typename Elem<T>::vec4;
  constexpr int TM = NCW * MPW;                   // M-tiles per workgroup tile
  constexpr int CT = NT * 32;
  constexpr int WROWS = 9 * CT;
  constexpr int LMAX = fl_lmax(NT, RESW);
  constexpr int XPL = fl_plane(LMAX), WPL = fl_plane(WROWS);
  static_assert(XPL == LMAX * 16, "LMAX is already a legal plane length");
  constexpr int XBYTES = 4 * XPL, WBYTES = 4 * WPL;
  constexpr int STAGE = RESW ? XBYTES : XBYTES + WBYTES;
  constexpr bool CAT = MODE == 1;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wres = smem + 2 * XBYTES;
  float* const sBias = reinterpret_cast<float*>(smem + (RESW ? 2 * XBYTES + a.nk * WBYTES : 2 * STAGE));   // [4 tiles][CT]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;

  const int step0 = (int)blockIdx.x * a.tiles_per_wg;                      // tile-steps: (tile, cout tile), cout tile fastest
  const int step_end = min(step0 + a.tiles_per_wg, a.total_tiles);
  if (step0 >= step_end) return;
  const int P = a.fl_pitch, MTS = a.fl_mts, HALO = a.fl_halo;
  const unsigned VS = (unsigned)(32 * MTS + HALO);

  if (wave < NCW) {
    // =================================================================================================================
    // multiplying waves
    // =================================================================================================================
    const int cw = wave;
    if ((a.pc_prio & 3) == 1) __builtin_amdgcn_s_setprio(2);
    f32x16 acc[MPW][NT];
    f32x16 zero16;
#pragma unroll
    for (int i = 0; i < 16; ++i) zero16[i] = 0.f;
    // fragment bases (LDS byte addresses): weights row lr of K-slot plane lh; per (M-tile, tap row) the lane's first pixel
    unsigned abase = (unsigned)(lh * WPL + lr * 16) + (RESW ? 2u * XBYTES : (unsigned)XBYTES);
    // (eight staging waves: 168 registers for everybody -- one base per M-tile, the tap row's offset added per read)
    constexpr int NB = PW == 8 ? 1 : 3;
    unsigned bb[MPW][NB];
    const unsigned ty_step = (unsigned)P * 16u;
    const unsigned blane = (unsigned)(lh * XPL + (cw * MPW * 32 + lr) * 16);
    // the tile's geometry: crossings before each of this wave's M-tiles (and whether the M-tile exists)
    int mt_n[MPW], mt_q0[MPW];
    auto tile_bases = [&](int step, unsigned stage_off) __attribute__((always_inline)) {
      const int tt = step >> a.fl_ct_shift;
      const int g0 = tt * TM;
      const int n_first = fl_sample_of(a, g0);
#pragma unroll
      for (int m = 0; m < MPW; ++m) {
        const int g = g0 + cw * MPW + m;
        const int gc = min(g, a.fl_total_mt - 1);          // M-tiles past the end of the batch read the last one's pixels
        const int n = fl_sample_of(a, gc);
        mt_n[m] = g < a.fl_total_mt ? n : -1;
        mt_q0[m] = (gc - n * MTS) * 32;
        const int cj = n - n_first;
        // (an M-tile past the end keeps the slot its index has: inside the image, contents irrelevant)
        const unsigned cro = (unsigned)((g < a.fl_total_mt ? cj : 0) * HALO) * 16u;
#pragma unroll
        for (int ty = 0; ty < NB; ++ty) bb[m][ty] = stage_off + blane + cro + (unsigned)(ty * P) * 16u;
      }
    };

    vec A[2][NT], B[2][MPW];
    constexpr int NRD = NT + MPW, NMM = NT * MPW;
    auto rd = [&](int set, int col) __attribute__((always_inline)) {
      const int ks = col / 9, r9 = col - 9 * ks, tx = r9 / 3, ty = r9 - 3 * tx;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        A[set][nt] = *reinterpret_cast<const vec*>(smem + abase + 2 * ks * WPL + ((ty * 3 + tx) * CT + nt * 32) * 16);
#pragma unroll
      for (int m = 0; m < MPW; ++m)
        B[set][m] = *reinterpret_cast<const vec*>(smem + (NB == 3 ? bb[m][ty] : bb[m][0] + (unsigned)ty * ty_step) + 2 * ks * XPL + (m * 32 + tx) * 16);
    };
    auto mm = [&](int col) __attribute__((always_inline)) {
      const int set = col & 1;
#if UNCL_FL_ABL_MASK
      if (FL_ABL(128)) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) asm volatile("" ::"v"(A[set][nt]));
#pragma unroll
        for (int m = 0; m < MPW; ++m) asm volatile("" ::"v"(B[set][m]));
        return;
      }
#endif
#pragma unroll
      for (int m = 0; m < MPW; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[m][nt] = mfma32x16(A[set][nt], B[set][m], acc[m][nt]);
    };
    auto sched = [&]() __attribute__((always_inline)) {
      constexpr int K = NRD < NMM ? NRD : NMM;
      // (more reads than multiplies: the surplus goes into the first gap)
      __builtin_amdgcn_sched_group_barrier(0x100, 1 + (NRD - K), 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
      for (int k = 1; k < K; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      if (NMM > K) __builtin_amdgcn_sched_group_barrier(0x008, NMM - K, 0);
    };
    auto cols_0_16 = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int col = 0; col < 17; ++col) {
        rd((col & 1) ^ 1, col + 1);
        mm(col);
        sched();
      }
    };
    auto flip_stage = [&](unsigned to_stage, int next_kc) __attribute__((always_inline)) {
      // X bases: stage (s & 1); weights: the other streamed stage, or resident chunk next_kc
      const unsigned d = to_stage ? (unsigned)STAGE : (unsigned)(0 - STAGE);
#pragma unroll
      for (int m = 0; m < MPW; ++m)
#pragma unroll
        for (int ty = 0; ty < NB; ++ty) bb[m][ty] += d;
      if (RESW) abase = (unsigned)(lh * WPL + lr * 16) + 2u * XBYTES + (unsigned)next_kc * WBYTES;
      else abase += d;
    };
    auto zero_acc = [&]() __attribute__((always_inline)) {
      vec zv = E::zero();
      asm volatile("" : "+v"(zv));
#pragma unroll
      for (int m = 0; m < MPW; ++m)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[m][nt] = mfma32x16(zv, zv, zero16);
    };

    // Epilogue: bias + activation + rounding on packed registers, v_permlane32_swap widens a lane's two 4-channel quads to
    // eight consecutive channels, one 16-byte buffer store per lane; the pixel (y, x) of a lane comes from ONE reciprocal
    // multiply per M-tile, and a dropped column / row / M-tile adds 2^30 to the offset (beyond num_records: the hardware drops
    // the store).  GRAD: identity activation, then the producing layer's ReLU mask and / or the gradient already in memory.
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    auto epilogue = [&](int step, int tpar) __attribute__((always_inline)) {
      constexpr unsigned BAD = 0x40000000u;
      const float* sBt = sBias + tpar * CT;
      const int co = (step & (a.n_ct - 1)) * CT;
      const unsigned sample = (unsigned)(a.Hout * a.Wout * a.oC) * 2u;
#pragma unroll
      for (int m = 0; m < MPW; ++m) {
        const int n = mt_n[m];                              // wave-uniform; -1: no such M-tile
        const unsigned q = (unsigned)(mt_q0[m] + lr);
        const unsigned y = __umulhi(q, a.fl_div_pitch), x = q - y * (unsigned)P;
        const bool ok = n >= 0 && x < (unsigned)a.Wout && y < (unsigned)a.Hout;
        const unsigned voff = (((y * (unsigned)a.Wout + x) * (unsigned)a.oC + (unsigned)(co + 8 * lh)) * 2u) | (ok ? 0u : BAD);
        const size_t sbase = (size_t)(n < 0 ? 0 : n) * sample;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.out) + sbase, (short)0, (int)sample, 0x00020000);
        __amdgpu_buffer_rsrc_t rm = rs;
        if (GRAD && a.mask != nullptr)
          rm = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.mask)) + sbase, (short)0, (int)sample, 0x00020000);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int qp = 0; qp < 2; ++qp) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(sBt + nt * 32 + 16 * qp + 4 * lh);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(sBt + nt * 32 + 16 * qp + 8 + 4 * lh);
            const f32x16& v = acc[m][nt];
            auto pack4 = [&](int qd, const f32x4& b) __attribute__((always_inline)) {
              const f32x2 s0 = f32x2{v[4 * qd], v[4 * qd + 1]} + f32x2{b[0], b[1]};
              const f32x2 s1 = f32x2{v[4 * qd + 2], v[4 * qd + 3]} + f32x2{b[2], b[3]};
              vec4 o;
              o[0] = (T)s0[0]; o[1] = (T)s0[1]; o[2] = (T)s1[0]; o[3] = (T)s1[1];
              return o;
            };
            const u32x2 d0 = __builtin_bit_cast(u32x2, pack4(2 * qp, b0)), d1 = __builtin_bit_cast(u32x2, pack4(2 * qp + 1, b1));
            const auto sx = __builtin_amdgcn_permlane32_swap(d0[0], d1[0], false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(d0[1], d1[1], false, false);
            const u32x4 w = {sx[0], sy[0], sx[1], sy[1]};
            s16x8 si = __builtin_bit_cast(s16x8, w);
            const unsigned off = voff + (unsigned)(nt * 32 + 16 * qp) * 2u;
            if (!GRAD) {
              si = __builtin_elementwise_max(si, s16x8{0, 0, 0, 0, 0, 0, 0, 0});      // ReLU on the rounded values
            } else {
              if (a.mask != nullptr) {
                // zero where the producing layer's ReLU output is not positive (signed 16-bit compare orders 16-bit floats of
                // one sign like their values: mask > 0 <=> int16(mask) > 0), as conv3x3_pc's gradient epilogue
#ifdef UNCL_CHECKED
                if (off < BAD) UNCL_CHK(a.chk, reinterpret_cast<const char*>(a.mask) + sbase + off, 16);
#endif
                const s16x8 mk = __builtin_bit_cast(s16x8, __builtin_amdgcn_raw_buffer_load_b128(rm, off, 0, 0));
                const s16x8 one = {1, 1, 1, 1, 1, 1, 1, 1}, zero = {0, 0, 0, 0, 0, 0, 0, 0};
                const s16x8 keep = __builtin_elementwise_max(__builtin_elementwise_min(mk, one), zero);
                si = (s16x8)(si * keep);
              }
              if (a.accumulate) {
                float f[8], o[8];
                E::unpack(__builtin_bit_cast(vec, si), f);
#ifdef UNCL_CHECKED
                if (off < BAD) UNCL_CHK(a.chk, reinterpret_cast<const char*>(a.out) + sbase + off, 16);
#endif
                E::unpack(__builtin_bit_cast(vec, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0)), o);
#pragma unroll
                for (int i = 0; i < 8; ++i) f[i] += o[i];
                si = __builtin_bit_cast(s16x8, E::pack(f));
              }
            }
#ifdef UNCL_CHECKED
            if (off < BAD) UNCL_CHK(a.chk, reinterpret_cast<const char*>(a.out) + sbase + off, 16);
#endif
#if UNCL_FL_ABL_MASK
            if (FL_ABL(64)) { asm volatile("" ::"v"(si)); continue; }
#endif
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, si), rs, off, 0, 0);
          }
      }
    };

#pragma unroll
    for (int m = 0; m < MPW; ++m)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[m][nt] = zero16;
    int tpar = 0;
    tile_bases(step0, 0u);
    FLT_DECL
    fl_barrier();                         // stage 0 is staged
    FLT(10)
    rd(0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
    int s = 0;                            // chunks multiplied so far: chunk s sits in stage s & 1
    for (int t = step0; t < step_end; ++t) {
      for (int kc = 0; kc < a.nk - 1; ++kc) {
        cols_0_16();
        FLT(kc & 3)
        __builtin_amdgcn_sched_barrier(0);
        fl_barrier();                     // column 17's fragments have landed: done with stage s & 1; stage (s + 1) & 1 is staged
        __builtin_amdgcn_sched_barrier(0);
        FLT(4 + (kc & 3))
        ++s;
        flip_stage(s & 1, kc + 1);
        rd(0, 0);
        mm(17);
        sched();
      }
      cols_0_16();
      mm(17);
      __builtin_amdgcn_sched_group_barrier(0x008, NMM, 0);
      FLT((a.nk - 1) & 3)
      epilogue(t, tpar);
      zero_acc();
      FLT(8)
      fl_barrier();
      FLT(4 + ((a.nk - 1) & 3))
      ++s;
      if (t + 1 < step_end) {
        tile_bases(t + 1, (s & 1) ? (unsigned)STAGE : 0u);
        if (RESW) abase = (unsigned)(lh * WPL + lr * 16) + 2u * XBYTES;
        else abase = (unsigned)(lh * WPL + lr * 16) + (unsigned)XBYTES + ((s & 1) ? (unsigned)STAGE : 0u);
        rd(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
      }
      tpar = (tpar + 1) & 3;
      FLT(9)
    }
    FLT_FLUSH(0, 32)
    return;
  }

  // ===================================================================================================================
  // staging waves
  // ===================================================================================================================
  if ((a.pc_prio & 3) == 2) __builtin_amdgcn_s_setprio(2);
  const int ptid = tid - NCW * 64;
  constexpr int NPROD = PW * 64;
  constexpr int PPP = NPROD / 4;                     // pixels (slots) per pass
  constexpr int XV = (LMAX + PPP - 1) / PPP;
  constexpr int WPP = NPROD / 4;                     // weight rows per pass
  static_assert(WPP % CT == 0, "a weight pass covers whole taps");
  constexpr int WVN = (WROWS + WPP - 1) / WPP;
  constexpr bool W_RAGGED = WROWS % WPP != 0;
  const int ch = ptid & 3, p0 = ptid >> 2;
  const int lds_w0 = ch * WPL + p0 * 16;
  // weight row p0 + WPP j = tap (p0 / CT + (WPP / CT) j), cout p0 % CT of the tile: one per-thread offset, the pass stride goes
  // into the scalar base
  const int woff0 = ((p0 / CT) * a.Cout + (p0 % CT)) * 32 + ch * 8;      // K-chunk-major weights, [Cin / 32][9][Cout][32] (common.h)

  vec xa[XV], xb[XV];       // xb is dead for plain sources
  constexpr bool W2 = UNCL_FL_W2 && !RESW;
  vec wv[W2 ? 2 : 1][WVN];
  f32x4 br = {0.f, 0.f, 0.f, 0.f};
  unsigned validA = 0, validB = 0;
  int npA = XV, npB = XV;          // passes the registers in xa / xb (plain: xa) hold data for
  int bpar = 0, ppar = 0;
  bool bp = false;

  auto weight_chunk = [&](int kc) __attribute__((always_inline)) {
    if (!CAT) return kc;
    return ssr_member(kc & 3) * (a.s0C >> 5) + (kc >> 2);
  };
  auto load_weights = [&](int cout0, int kc, int set) __attribute__((always_inline)) {
    const bf16_t* wb_ = a.weight + ((size_t)weight_chunk(kc) * 9 * a.Cout + cout0) * 32;
    const int wstride = (WPP / CT) * a.Cout * 32;     // taps per pass x one tap
#pragma unroll
    for (int j = 0; j < WVN; ++j) {
      unsigned off = (unsigned)woff0;
      if (W_RAGGED && j == WVN - 1) off = (p0 + WPP * j < WROWS) ? off : 0u;
      if (FL_ABL(32)) continue;
      wv[set][j] = LD16OV(vec, wb_ + j * wstride, off * 2u);
    }
  };
  auto write_weights = [&](char* wst, int set) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < WVN; ++j) {
      if (W_RAGGED && j == WVN - 1 && p0 + WPP * j >= WROWS) continue;
      if (FL_ABL(1024)) { asm volatile("" ::"v"(wv[set][j])); continue; }
      *reinterpret_cast<vec*>(wst + lds_w0 + j * WPP * 16) = wv[set][j];
    }
  };

  // The slots of a tile.  Slot i of tile tt is V = V0 + i -> (sample, padded row, padded column); computed ONCE per tile and kept
  // as a code per slot -- bit 31 valid (inside the tile's image, the batch and the un-padded input), bits 24..28 sample - first
  // sample of the tile, bits 12..23 input row, bits 0..11 input column -- from which a request builds its byte offsets with three
  // 24-bit multiply-adds (full rate; the 32-bit multiplies of a from-scratch address are quarter rate, and a concat layer
  // requests registers for every 32-channel slice of both sources: the staging waves are what these launches wait for).
  struct TileCodes { unsigned code[XV]; int tt, n0, np; };
  // passes 0 .. JMIN - 1 cover the tile's own 32 TM pixels and are always needed; the rest depends on the tile's halo count
  constexpr int JMIN = (32 * TM) / PPP;
  auto tile_codes = [&](TileCodes& tc, int tt) __attribute__((always_inline)) {
    const int g0 = tt * TM;
    const int n_first = fl_sample_of(a, g0);
    const int g_last = min(g0 + TM, a.fl_total_mt) - 1;
    const int n_last = fl_sample_of(a, g_last);
    const unsigned p_first = (unsigned)(g0 - n_first * MTS) * 32u;
    const unsigned L = (unsigned)(32 * TM + (n_last - n_first + 1) * HALO);
    tc.tt = tt;
    tc.n0 = n_first;
    tc.np = __builtin_amdgcn_readfirstlane((int)((L + PPP - 1) / PPP));
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      tc.code[j] = 0u;
      if (j >= JMIN && j >= tc.np) continue;            // wave-uniform
      if (FL_ABL(2048)) { tc.code[j] = 0x80000000u | (unsigned)(p0 & 15); continue; }
      const unsigned slot = (unsigned)(p0 + j * PPP);
      unsigned p = p_first + slot;
      unsigned dn = 0;
      // (a tile spans at most fl_cmax sample borders: launcher)
      for (int c = 0; c < a.fl_cmax; ++c)
        if (p >= VS) { p -= VS; ++dn; }
      const unsigned yy = __umulhi(p, a.fl_div_pitch), xx = p - __umul24(yy, (unsigned)P);
      const unsigned iy = yy - (unsigned)a.pad, ix = xx - (unsigned)a.pad;
      const bool ok = slot < L && (int)dn + n_first < a.flat_N && iy < (unsigned)a.H && ix < (unsigned)a.W;
      tc.code[j] = ok ? (0x80000000u | (dn << 24) | (iy << 12) | ix) : 0u;
    }
  };
  // SRC1: offsets into the up-sampled map (replicate-padded to the skip's extent, unet_parts.py:292-298)
  const int dy1 = (a.s0H - a.s1H) >> 1, dx1 = (a.s0W - a.s1W) >> 1;
  auto slot_offset = [&](unsigned code, int n0, bool src1) __attribute__((always_inline)) {
    const unsigned n = (unsigned)n0 + ((code >> 24) & 31u);
    const int iy = (int)((code >> 12) & 0xfffu), ix = (int)(code & 0xfffu);
    unsigned o;
    if (src1) {
      const int sy = min(max(iy - dy1, 0), a.s1H - 1), sx = min(max(ix - dx1, 0), a.s1W - 1);
      o = __umul24(__umul24(__umul24(n, (unsigned)a.s1H) + (unsigned)sy, (unsigned)a.s1W) + (unsigned)sx, (unsigned)a.s1C * 2u);
    } else {
      o = __umul24(__umul24(__umul24(n, (unsigned)a.s0H) + (unsigned)iy, (unsigned)a.s0W) + (unsigned)ix, (unsigned)a.s0C * 2u);
    }
    o += (unsigned)ch * 16u;
    return (int)code < 0 ? o : 0u;
  };
  auto codes_valid = [&](const TileCodes& tc) __attribute__((always_inline)) {
    unsigned valid = 0;
#pragma unroll
    for (int j = 0; j < XV; ++j) valid |= (tc.code[j] >> 31) << j;
    return valid;
  };

  // cursor of the chunk stream: tile-step, K-chunk
  struct Cur { int step, kc; };
  auto cur_next = [&](Cur& c) __attribute__((always_inline)) {
    if (++c.kc < a.nk) return true;
    c.kc = 0;
    return ++c.step < step_end;
  };
  Cur pc{step0, 0};
  TileCodes tcB, tcA;                // x2 / plain source: the tile of the main cursor; x1 (concat): the tile of the look-ahead
  tcB.tt = -1; tcA.tt = -1; tcB.n0 = tcA.n0 = 0; tcB.np = tcA.np = XV;
  const int total = (step_end - step0) * a.nk;
  int loaded = 0;

  auto load_xa = [&](const Cur& c) __attribute__((always_inline)) {
    // x1 slice (concat) of the slice that starts at chunk c
    const int tt = c.step >> a.fl_ct_shift;
    if (tt != tcA.tt) tile_codes(tcA, tt);
    validA = codes_valid(tcA);
    npA = tcA.np;
    const bf16_t* base = a.src1 + (c.kc >> 2) * 32;
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      if (j >= JMIN && j >= npA) continue;
      if (FL_ABL(16)) continue;
      xa[j] = LD16OV(vec, base, slot_offset(tcA.code[j], tcA.n0, true));
    }
  };
  auto load_step = [&](const Cur& c, auto p_tag) __attribute__((always_inline)) {
    constexpr int PH = decltype(p_tag)::value;
    const int tt = c.step >> a.fl_ct_shift, cout0 = (c.step & (a.n_ct - 1)) * CT;
    bp = c.kc == 0;
    bpar = ppar;
    if (!CAT || PH == 0) {
      // plain: this chunk's channels; concat: the x2 slice of this group -- loaded once, staged three times (sqrt, square, as is)
      if (tt != tcB.tt) tile_codes(tcB, tt);
      validB = codes_valid(tcB);
      npB = tcB.np;
      const bf16_t* base = a.src0 + (CAT ? (c.kc >> 2) : c.kc) * 32;
      vec (&xr)[XV] = CAT ? xb : xa;
#pragma unroll
      for (int j = 0; j < XV; ++j) {
        if (j >= JMIN && j >= npB) continue;
        if (FL_ABL(16)) continue;
        xr[j] = LD16OV(vec, base, slot_offset(tcB.code[j], tcB.n0, false));
      }
    }
    if (!RESW) {
      if (!W2) load_weights(cout0, c.kc, 0);
      if (bp && ptid < CT / 4 && a.bias != nullptr) br = LD16O_F32(a.bias + cout0, (unsigned)ptid * 16u);
    }
  };
  auto write_step = [&](char* st, auto p_tag, int wset) __attribute__((always_inline)) {
    constexpr int PH = decltype(p_tag)::value;
    constexpr bool SET_A = !CAT || PH == 0;
    vec (&xr)[XV] = SET_A ? xa : xb;
    const unsigned xvalid = (!CAT || !SET_A) ? validB : validA;
    const int np = (!CAT || !SET_A) ? npB : npA;
    auto transform = [&](vec v) __attribute__((always_inline)) {
      if (CAT && ssr_member(PH) >= 2 && !FL_ABL(512)) {
        float f[8];
        E::unpack(v, f);
        if (ssr_member(PH) == 2) {
#pragma unroll
          for (int i = 0; i < 8; ++i) f[i] = f[i] * f[i];
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) f[i] = __builtin_amdgcn_sqrtf(f[i] + 1e-8f);
        }
        v = E::pack(f);
      }
      return v;
    };
    const bool wave_ok = __builtin_amdgcn_ballot_w64(xvalid != 0xffffffffu) == 0;
    if (wave_ok) {
#pragma unroll
      for (int j = 0; j < XV; ++j) {
        if (j >= JMIN && j >= np) continue;
        if ((j + 1) * PPP > LMAX && p0 + j * PPP >= LMAX) continue;
        if (FL_ABL(256)) { asm volatile("" ::"v"(xr[j])); continue; }
        *reinterpret_cast<vec*>(st + ch * XPL + (p0 + j * PPP) * 16) = transform(xr[j]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < XV; ++j) {
        if (j >= JMIN && j >= np) continue;
        if ((j + 1) * PPP > LMAX && p0 + j * PPP >= LMAX) continue;
        vec v = transform(xr[j]);
        if (!((xvalid >> j) & 1u)) v = E::zero();
        *reinterpret_cast<vec*>(st + ch * XPL + (p0 + j * PPP) * 16) = v;
      }
    }
    if (!RESW) {
      write_weights(st + XBYTES, wset);
      if (bp && ptid < CT / 4) *reinterpret_cast<f32x4*>(sBias + bpar * CT + ptid * 4) = br;
    }
  };

  auto load_next = [&](auto p_tag) __attribute__((always_inline)) {
    constexpr int PH = decltype(p_tag)::value;
    if (loaded < total) {
      load_step(pc, p_tag);
      // x1 of the NEXT slice (chunk loaded + 3): xa has been free since this slice's x1 chunk was staged
      if (CAT && PH == 1 && loaded + 3 < total) {
        Cur la = pc;
#pragma unroll
        for (int k = 0; k < 3; ++k) cur_next(la);
        load_xa(la);
      }
      ++loaded;
      const int s_old = pc.step;
      if (cur_next(pc) && pc.step != s_old) ppar = (ppar + 1) & 3;
    }
  };
  if (RESW) {
    // the layer's whole weight tensor (one cout tile, nk chunks) becomes resident, and so does its bias (all four slots)
    for (int kc = 0; kc < a.nk; ++kc) {
      load_weights(0, kc, 0);
      write_weights(wres + kc * WBYTES, 0);
    }
    if (ptid < CT / 4) {
      if (a.bias != nullptr) UNCL_CHK(a.chk, a.bias + ptid * 4, 16);
      const f32x4 b4 = a.bias != nullptr ? *reinterpret_cast<const f32x4*>(a.bias + ptid * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) *reinterpret_cast<f32x4*>(sBias + sl * CT + ptid * 4) = b4;
    }
  }
  // W2: the weights' own cursor runs one chunk ahead of the activations'
  Cur wc{step0, 0};
  int wloaded = 0;
  auto load_w_next = [&](int set) __attribute__((always_inline)) {
    if (W2 && wloaded < total) {
      load_weights((wc.step & (a.n_ct - 1)) * CT, wc.kc, set);
      ++wloaded;
      cur_next(wc);
    }
  };
  load_w_next(0);                       // chunk 0
  if (CAT) load_xa(pc);                 // the first slice's x1
  load_next(IntTag<0>{});
  load_w_next(1);                       // chunk 1
  write_step(smem, IntTag<0>{}, 0);
  load_next(IntTag<1>{});
  load_w_next(0);                       // chunk 2
  FLT_DECL
  fl_barrier();                         // stage 0 is staged
  FLT(12)
  // iteration s (the multiplying waves work on chunk s): chunk s + 1 goes from registers to the stage they left at the last
  // barrier, then the loads of chunk s + 2 (W2: the weights of chunk s + 3) are issued into the same registers
  int s = 0;
  auto iter = [&](auto q_tag) __attribute__((always_inline)) {
    constexpr int Q = decltype(q_tag)::value;
    if (s + 1 >= total) {
      fl_barrier();                     // the multiplying waves' last chunk
      return true;
    }
    write_step(smem + ((Q + 1) & 1) * STAGE, IntTag<(Q + 1) & 3>{}, W2 ? (Q + 1) & 1 : 0);
    FLT(Q)
    load_next(IntTag<(Q + 2) & 3>{});
    load_w_next((Q + 1) & 1);
    FLT(4 + Q)
    fl_barrier();
    FLT(8 + Q)
    ++s;
    return false;
  };
  for (;;) {
    if (iter(IntTag<0>{})) break;
    if (iter(IntTag<1>{})) break;
    if (iter(IntTag<2>{})) break;
    if (iter(IntTag<3>{})) break;
  }
  FLT_FLUSH(16, 33)
}

template <int NT>
constexpr size_t fl_lds_bytes(bool resw, int nk) {
  const size_t xb = 4 * (size_t)fl_lmax(NT, resw) * 16, wb = 4 * (size_t)fl_wplane(NT);
  return (resw ? 2 * xb + nk * wb : 2 * (xb + wb)) + 4 * NT * 32 * 4;
}

template <typename T, int NT, int MPW, int MODE, int PW, bool RESW, bool GRAD>
int launch_fl(PipeArgs& a, hipStream_t s) {
  const size_t lds = fl_lds_bytes<NT>(RESW, a.nk);
  if (lds > 163840) return UNCL_ERR_ARG;
  auto kern = conv3x3_fl_kernel<T, NT, MPW, MODE, PW, RESW, GRAD>;
  static UnclDevOnce attr_done;
  if (attr_done.need()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)fl_lds_bytes<NT>(RESW, 4 / NT)) != hipSuccess)
      return UNCL_ERR_LAUNCH;
    attr_done.done();
  }
  const int n_cu = uncl_cu_count();
  if (n_cu <= 0) return UNCL_ERR_LAUNCH;
  int grid = a.total_tiles < n_cu ? a.total_tiles : n_cu;
  a.tiles_per_wg = (a.total_tiles + grid - 1) / grid;
  grid = (a.total_tiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
  hipLaunchKernelGGL(kern, dim3(grid), dim3((4 + PW) * 64), lds, s, a);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

// Staging waves: 4 or 8.  Eight need the multiplying waves in 168 registers (three waves per SIMD): MPW <= 3, one fragment base per
// M-tile instead of one per (M-tile, tap row).  UNCL_FL_PW8: 0 = four everywhere, 1 = eight for the concat form, 2 (default) = for
// the plain forward form too.  Same-box A/B, ms per 200 tiles, four -> eight: up_path.1.conv.conv 0.336 -> 0.305, up_path.0.conv.conv
// 0.238 -> 0.233, down_path.1 first conv 0.122 -> 0.117, up_path.1.conv.conv1 0.063 -> 0.060; whole forward 3.80 -> 3.76 ms.
#ifndef UNCL_FL_PW8_DEFAULT
#define UNCL_FL_PW8_DEFAULT 2
#endif
template <typename T, int MPW, bool GRAD>
int fl_dispatch(PipeArgs& a, int mode, bool resw, hipStream_t s) {
  static const int pw8 = [] { const char* e = getenv("UNCL_FL_PW8"); return e ? atoi(e) : UNCL_FL_PW8_DEFAULT; }();
  if constexpr (MPW <= 3 && !GRAD) {
    if (pw8 && mode == 1 && !resw) return launch_fl<T, 2, MPW, 1, 8, false, GRAD>(a, s);
    if (pw8 >= 2 && mode == 0) return resw ? launch_fl<T, 2, MPW, 0, 8, true, GRAD>(a, s) : launch_fl<T, 2, MPW, 0, 8, false, GRAD>(a, s);
  }
  if (mode == 0) return resw ? launch_fl<T, 2, MPW, 0, 4, true, GRAD>(a, s) : launch_fl<T, 2, MPW, 0, 4, false, GRAD>(a, s);
  if (mode == 1) return resw ? UNCL_ERR_ARG : launch_fl<T, 2, MPW, 1, 4, false, GRAD>(a, s);    // (nk >= 4: never resident)
  return UNCL_ERR_ARG;
}

}  // namespace

#ifdef UNCL_FL_TIMING
// measurement builds only: copy (and optionally clear) the per-phase cycle counters
extern "C" int uncl_fl_timing_read(unsigned long long* out48, int reset) {
  if (hipMemcpyFromSymbol(out48, HIP_SYMBOL(g_fl_t), sizeof(unsigned long long) * 48) != hipSuccess) return UNCL_ERR_LAUNCH;
  if (reset) {
    unsigned long long z[48] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_fl_t), z, sizeof(z)) != hipSuccess) return UNCL_ERR_LAUNCH;
  }
  return UNCL_OK;
}
#endif

static std::atomic<long long> g_flat_launches{0};
// launches that took the flat tiles so far (tests: the cost figure must choose them where they were built for)
extern "C" long long uncl_conv3x3_flat_count() { return g_flat_launches.load(std::memory_order_relaxed); }

// Flat M-tiles for a layer of 64-channel cout tiles: geometry, LDS fit (sample borders inside a tile included), and which MPW.
// Returns UNCL_ERR_ARG where this kernel does not apply (the caller falls through to the rectangular tiles).
//   mpw_pref 0: choose; 2 / 3 / 4: that many M-tiles per multiplying wave or nothing (tests)
//   max_cost > 0: only if the cost model (rounds of the persistent grid x (M-tiles per wave + 0.35)) stays below it -- the caller
//   passes the same figure for its rectangular tiles
int uncl_conv3x3_flat_launch(PipeArgs& a, int dtype, int mode, int mpw_pref, double max_cost, hipStream_t s) {
  if (a.Cout % 64 != 0 || a.nk < 1 || a.res != nullptr || a.flat_S != 0 || a.pool_out != nullptr || a.out1_w != nullptr ||
      a.skip_main || a.prev0 != nullptr)
    return UNCL_ERR_ARG;
  if (mode != 0 && mode != 1) return UNCL_ERR_ARG;
  if (mode == 1 && (a.nk & 3) != 0) return UNCL_ERR_ARG;
  const int n_ct = a.Cout / 64;
  if (n_ct & (n_ct - 1)) return UNCL_ERR_ARG;
  // forward: bias + ReLU; gradient: identity with ReLU mask (slope 0) and / or accumulation
  const bool fwd = a.slope == 0.f && a.mask == nullptr && !a.accumulate;
  const bool grad = a.slope == 1.f && a.mask_slope == 0.f && dtype != UNCL_F16;
  if (!fwd && !grad) return UNCL_ERR_ARG;
  const int P = a.Wout + 2;
  if (a.W + 2 * a.pad != P || a.H + 2 * a.pad != a.Hout + 2) return UNCL_ERR_ARG;
  const long long px = (long long)a.Hout * P;
  if (px >= (1 << 20) || P > 1024) return UNCL_ERR_ARG;                       // exact reciprocal multiplies
  const int mts = (int)((px + 31) / 32), halo = 2 * P + 2;
  const long long total_mt = (long long)a.flat_N * mts;
  if (total_mt >= (1 << 24)) return UNCL_ERR_ARG;
  // 32-bit byte offsets over the whole batch, one buffer descriptor per sample
  if ((unsigned long long)a.flat_N * a.s0H * a.s0W * a.s0C * 2 >= (1ull << 32)) return UNCL_ERR_ARG;
  if (mode == 1 && (unsigned long long)a.flat_N * a.s1H * a.s1W * a.s1C * 2 >= (1ull << 32)) return UNCL_ERR_ARG;
  if ((unsigned long long)a.Hout * a.Wout * a.oC * 2 >= (1ull << 30)) return UNCL_ERR_ARG;
  // slot codes: 12-bit rows / columns, 24-bit multiply operands (pixel index in the batch)
  if (a.H >= 4096 || a.W >= 4096 || (long long)a.flat_N * a.s0H * a.s0W >= (1 << 24) ||
      (mode == 1 && (long long)a.flat_N * a.s1H * a.s1W >= (1 << 24)))
    return UNCL_ERR_ARG;
  const bool resw = n_ct == 1 && a.nk * 2 <= 4;
  const int lmax = fl_lmax(2, resw);
  const int n_cu = uncl_cu_count();
  if (n_cu <= 0) return UNCL_ERR_LAUNCH;
  int best = 0, best_cmax = 0;
  double best_cost = 0.;
  for (int mpw = 4; mpw >= 2; --mpw) {
    if (mpw_pref != 0 && mpw != mpw_pref) continue;
    const int tm = 4 * mpw;
    const int cmax = mts >= tm ? 1 : (tm - 2) / mts + 1;                       // sample borders one tile can span
    if (32 * tm + (cmax + 1) * halo > lmax || cmax > 31) continue;
    // cost model: rounds of the persistent grid x M-tiles per wave (+ one M-tile's worth per tile-step for the epilogue / hand-over)
    const long long steps = ((total_mt + tm - 1) / tm) * n_ct;
    const long long rounds = (steps + n_cu - 1) / n_cu;
    const double cost = (double)rounds * (mpw + 0.35);
    if (best == 0 || cost < best_cost) { best = mpw; best_cost = cost; best_cmax = cmax; }
  }
  if (best == 0 || (max_cost > 0. && best_cost >= max_cost)) return UNCL_ERR_ARG;
  a.fl_pitch = P; a.fl_mts = mts; a.fl_halo = halo; a.fl_total_mt = (int)total_mt; a.fl_cmax = best_cmax;
  a.fl_div_pitch = (unsigned)((1ull << 32) / (unsigned)P) + 1u;
  a.fl_div_mts = (unsigned)((1ull << 32) / (unsigned)mts) + 1u;
  a.n_ct = n_ct;
  a.fl_ct_shift = n_ct == 1 ? 0 : (n_ct == 2 ? 1 : (n_ct == 4 ? 2 : (n_ct == 8 ? 3 : 4)));
  if ((1 << a.fl_ct_shift) != n_ct) return UNCL_ERR_ARG;
  a.total_tiles = (int)(((total_mt + 4 * best - 1) / (4 * best)) * n_ct);
  // wave priorities: 1 = multiplying waves raised (the rectangular kernel's default), 2 = staging waves raised, 0 = none
  static const int prio = [] { const char* e = getenv("UNCL_FL_PRIO"); return e ? atoi(e) : UNCL_FL_PRIO_DEFAULT; }();
  a.pc_prio = prio;
  g_flat_launches.fetch_add(1, std::memory_order_relaxed);
#define UNCL_FL_GO(T, G)                                     \
  (best == 4 ? fl_dispatch<T, 4, G>(a, mode, resw, s)        \
   : best == 3 ? fl_dispatch<T, 3, G>(a, mode, resw, s)      \
               : fl_dispatch<T, 2, G>(a, mode, resw, s))
  if (dtype == UNCL_F16) return UNCL_FL_GO(f16_t, false);
  return fwd ? UNCL_FL_GO(bf16_t, false) : UNCL_FL_GO(bf16_t, true);
#undef UNCL_FL_GO
}
