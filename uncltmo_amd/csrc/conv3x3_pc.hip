// Producer / consumer 3x3 implicit-GEMM convolution for gfx950 (bf16): the multi-chunk (Cin >= 64) layers of the generator.
//
// conv3x3_pipe.hip runs four waves that take turns loading, staging and multiplying, two workgroups per CU; its waves spend
// two thirds of their time outside the MFMA phase and the matrix pipe of a SIMD idles whenever both of its waves are there.
// Here the roles are split instead, ONE 512-thread workgroup per CU:
//
//   waves 0-3  (one per SIMD, raised priority): nothing but LDS fragment reads + v_mfma_f32_32x32x16_bf16 over the staged
//              K-chunk, and the tile's epilogue (bias, activation, wave-private LDS transposition, 1-KiB row stores, fused
//              2x2 max-pool copy).  They never touch global loads or staging writes.
//   waves 4-7  (their SIMD partners): global loads of chunk s+2 into registers, LDS staging of chunk s+1 (square / square-root
//              of the skip slice, replicate padding, zero borders, the 2x2 transposed conv of MODE 4 straight from global
//              fragments), one chunk ahead of the multiplying waves in a two-stage LDS ring.
//
// One s_barrier per K-chunk hands a stage from the producers to the consumers and the previous one back; global loads stay
// in flight across it (raw s_barrier + lgkmcnt only).  Same tiles and the same MFMA order per accumulator as conv3x3_pipe
// -> bit-identical results; tests/test_gpu_conv.py compares the two kernels element for element.
//
// LDS images are stored as four PLANES, one per 16-byte K-slot: plane s holds channels 8s..8s+7 of every halo pixel (or
// weight row) back to back, so the 16 lanes of a ds_read_b128 group read 256 contiguous bytes (conflict-free without the
// 80-byte row padding of conv3x3_pipe, which cost 25 % of the LDS) and every fragment address is "lane base + immediate".
// Plane lengths are = 4 (mod 16) slots, which spreads the four planes a staging quad writes over distinct banks.
// With the padding gone, the whole weight tensor of a layer with Cout = one tile and <= 4 K-chunks of 32 output channels
// (or 2 of 64) fits next to two activation stages and stays RESIDENT for the launch (RESW): the staging waves then load
// activations only -- per-CU vector-memory throughput (L2 -> L1, tens of GB/s per CU), not HBM, is what these kernels queue on.
// The epilogue needs no LDS (lane-widening v_permlane32_swap instead of a transposition).
#include <cstdlib>
#include <type_traits>

#include "conv3x3_args.h"

// concat sources: the iteration (its load slot P = loaded chunk index & 3) from which the next slice's x1 registers are requested
// (0: together with the x2 registers, at the slice's own first chunk), and whether in two halves (slots P and P + 1).
// Same-box A/B (tools/ab_variants.sh, tools/ab_layers.sh; ms per 200 tiles, slot 0 = round 2): slot 1 -- the iteration that has
// just staged x1 and otherwise only streams weights -- is best where few slices end in the multiplying waves' stores (up_path.0 /
// 1 / 2.conv.conv 0.410 -> 0.370, 0.399 -> 0.358, 0.474 -> 0.434); the fused up-conv layer stores after EVERY slice in exactly
// that iteration (0.817 -> 0.800 with slot 1) and takes slot 3 (0.758).  Halves from two slots pay the address generation twice.
#ifndef UNCL_PC_XA_SLOT
// (TAIL consumes the up-conv's source fragments one iteration earlier -- up_compute -- and requests the next ones from the
// iteration after that, its lightest: slot 1 = the load step of iteration 3)
#define UNCL_PC_XA_SLOT ((MODE == 4 || MODE == 5) ? (TAIL ? 1 : 3) : 1)
#endif
// cache-policy bits of the straight-line epilogue's buffer stores (0 default, 2 = nt: streaming)
#ifndef UNCL_PC_STORE_AUX
#define UNCL_PC_STORE_AUX 0
#endif
#ifndef UNCL_PC_LEAN_DEFAULT
#define UNCL_PC_LEAN_DEFAULT 1
#endif
// fused up-conv: 1 = one source M-tile per staging wave, all four taps, weights resident in LDS; 0 = one tap per wave (round 1 - 4)
// (MODE 4, the 32-channel up-conv of the dominant launch: a tie, 0.702 against 0.710 ms per 200 tiles in same-box A/B, so it keeps
// the per-tap form; MODE 5, 64 channels, always takes the M-tile form: 12 -> 4 source requests per wave and slice, 0.526 -> 0.451 ms)
#ifndef UNCL_PC_UP_TILE
#define UNCL_PC_UP_TILE 0
#endif
// compile-time ablations (measurement builds: tools/ab_variants.sh <name> "-DUNCL_PC_ABL_MASK=<bits>"; WRONG results):
// 1 no stores of the parked tile, 2 no pooled stores, 4 no rebuild of the first layer's halo tile (MODE 3), 8 no MFMAs in the
// multiplying waves (fragments still read), 16 no parking (the multiplying waves skip their epilogue altogether)
#ifndef UNCL_PC_ABL_MASK
#define UNCL_PC_ABL_MASK 0
#endif
#define PC_ABL(bit) (((UNCL_PC_ABL_MASK) & (bit)) != 0)
#ifndef UNCL_PC_XA_SPLIT
#define UNCL_PC_XA_SPLIT 0
#endif
// parked epilogue of the four-chunk concat layer: the staging iteration (chunk of the next tile) in which the parked tile is stored
#ifndef UNCL_PC_EPO_Q
#define UNCL_PC_EPO_Q 2
#endif

namespace {

__device__ __forceinline__ void pc_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Phase timing (measurement builds only, -DUNCL_PC_TIMING; tools/pc_phase_timing.py): wave 0 (consumer) and wave 4 (producer) of
// every workgroup accumulate s_memtime deltas per loop phase into g_pc_t[]; the product library compiles all of this away.
#ifdef UNCL_PC_TIMING
__device__ unsigned long long g_pc_t[32];    // [16..31]: per chunk kind (index & 3): producer busy / wait, consumer busy / wait
__device__ __forceinline__ unsigned long long pct_now() {
  unsigned long long t;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define PCT_DECL unsigned long long pct_acc[4] = {}; unsigned long long pct_k[8] = {}; unsigned long long pct_last = pct_now();
// sched_barrier: register-only instructions (MFMAs) must not drift across the stamp (an asm memory clobber does not hold them)
#define PCT(i) { __builtin_amdgcn_sched_barrier(0); const unsigned long long pct_t = pct_now(); __builtin_amdgcn_sched_barrier(0); \
                 pct_acc[i] += pct_t - pct_last; pct_last = pct_t; }
#define PCT_FLUSH(base) if (threadIdx.x == 0 || threadIdx.x == NCW * 64) { for (int i = 0; i < 4; ++i) atomicAdd(&g_pc_t[base + i], pct_acc[i]); atomicAdd(&g_pc_t[8 + base / 4], 1ull); \
                          for (int i = 0; i < 8; ++i) atomicAdd(&g_pc_t[(base ? 16 : 24) + i], pct_k[i]); }
// per chunk kind: K(i, slot) adds the time since the last stamp to slot (busy: 0..3, wait: 4..7) + the chunk's index & 3
#define PCT_K(i, slot) { __builtin_amdgcn_sched_barrier(0); const unsigned long long pct_t = pct_now(); __builtin_amdgcn_sched_barrier(0); \
                         pct_acc[i] += pct_t - pct_last; pct_k[slot] += pct_t - pct_last; pct_last = pct_t; }
#else
#define PCT_DECL
#define PCT(i)
#define PCT_K(i, slot)
#define PCT_FLUSH(base)
#endif

// bytes of one LDS plane of `rows` 16-byte slots: the slot count is padded to 4 (mod 16)
// Plane lengths.  A staging quad (four lanes = the four K-slot planes of one pixel; eight consecutive lanes = two pixels, one LDS
// cycle of a ds_write_b128) must land on 32 distinct banks.  WRITES are banked (address / 4) mod 32 (MI355X_MICROARCH.md, LDS
// table; reads mod 64), so the plane stride has to be 8 or 24 banks (mod 32), i.e. the slot count 2 or 6 (mod 8).  The rule of
// rounds 2 - 3, 4 (mod 16) slots, is 16 banks (mod 32): planes 0 / 2 and 1 / 3 collided, every staging write was a two-way
// conflict (SQ_LDS_BANK_CONFLICT = 16 - 25 % of SQ_LDS_IDX_ACTIVE on every launch of this kernel, profiles/r4c_pmc_layers.txt).
#ifndef UNCL_PLANE_RULE
#define UNCL_PLANE_RULE 1
#endif
constexpr int pc_plane16(int rows) { return (rows + ((4 - rows % 16) + 16) % 16) * 16; }
constexpr int pc_plane(int rows) { return UNCL_PLANE_RULE ? (rows + ((6 - rows % 8) + 8) % 8) * 16 : pc_plane16(rows); }

struct TileCur { int tile, ct, tx, ty, n, kc; };

// STRIP (fused last decoder stage): a workgroup walks DOWN a strip of tiles (ty fastest, one cout tile), because the second
// layer needs the last two rows of the previous tile's intermediate map
template <bool STRIP = false>
__device__ __forceinline__ void cur_init(TileCur& c, int tile, const PipeArgs& a) {
  int r = tile;
  c.tile = tile;
  if (STRIP) {
    c.ct = 0;
    c.ty = r % a.tiles_y; r /= a.tiles_y;
    c.tx = r % a.tiles_x; r /= a.tiles_x;
  } else {
    c.ct = r % a.n_ct; r /= a.n_ct;
    c.tx = r % a.tiles_x; r /= a.tiles_x;
    c.ty = r % a.tiles_y; r /= a.tiles_y;
  }
  c.n = r;
  c.kc = 0;
}
// one K-chunk further; false past the end of this workgroup's tile range
template <bool STRIP = false>
__device__ __forceinline__ bool cur_next(TileCur& c, const PipeArgs& a, int tile_end) {
  if (++c.kc < a.nk) return true;
  c.kc = 0;
  if (++c.tile >= tile_end) return false;
  if (STRIP) {
    if (++c.ty == a.tiles_y) {
      c.ty = 0;
      if (++c.tx == a.tiles_x) { c.tx = 0; ++c.n; }
    }
    return true;
  }
  if (++c.ct == a.n_ct) {
    c.ct = 0;
    if (++c.tx == a.tiles_x) {
      c.tx = 0;
      if (++c.ty == a.tiles_y) { c.ty = 0; ++c.n; }
    }
  }
  return true;
}

// MODE: 0 plain, 1 concat [x2, x1, x2^2, sqrt(x2+1e-8)], 4 = 1 with x1 = ConvTranspose2d(k2, s2)(src1) computed by the
//       producers (32 channels, same extent as the skip), 3 = the 32-channel source is act(conv3x3_valid(fp32 image)) rebuilt
//       by the producers' matrix cores from the image patch under the halo tile (inc.conv.conv fused into inc.conv.conv1),
//       5 = 4 for a 64-channel skip: x1 = ConvTranspose2d(64, 64, k2, s2)(src1), one 32-channel slice of its output per x1 chunk
//       (K = 64: four MFMAs per 32 source pixels; the slice's weight fragments and bias are requested with its source pixels)
#if defined(UNCL_PC_MFMA16_PROXY)
// TIMING PROXY ONLY (tools/ab_variants.sh, never the product build): the same operand registers through two
// v_mfma_f32_16x16x32 per 32x32x16 -- equal FLOP, equal LDS fragment reads, WRONG results -- to see what clock the chip holds on
// the other shape before the fragment layouts are rebuilt for it (MI355X_MICROARCH.md, DVFS give-back item 7)
__device__ __forceinline__ f32x4 mfma16x32_proxy(const bf16x8& a, const bf16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16x32_proxy(const f16x8& a, const f16x8& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
struct PcAcc {
  f32x4 q[4];
  __device__ __forceinline__ float operator[](int i) const { return q[i >> 2][i & 3]; }
  __device__ __forceinline__ PcAcc& operator=(const f32x16& v) {
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = f32x4{v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
    return *this;
  }
};
template <typename V>
__device__ __forceinline__ PcAcc pc_mm(const V& a, const V& b, PcAcc c, int ty) {
  const int j = 2 * (ty & 1);
  c.q[j] = mfma16x32_proxy(a, b, c.q[j]);
  c.q[j + 1] = mfma16x32_proxy(a, b, c.q[j + 1]);
  return c;
}
template <typename V>
__device__ __forceinline__ PcAcc pc_mm_zero(const V& z) {
  PcAcc c;
#pragma unroll
  for (int i = 0; i < 4; ++i) c.q[i] = mfma16x32_proxy(z, z, f32x4{0.f, 0.f, 0.f, 0.f});
  return c;
}
#else
typedef f32x16 PcAcc;
template <typename V>
__device__ __forceinline__ PcAcc pc_mm(const V& a, const V& b, const PcAcc& c, int) { return mfma32x16(a, b, c); }
#endif
// EPO ("parked epilogue", round 5; single-chunk 32-channel layers): the multiplying waves do not store their tile.  They round it
// (bias, ReLU, the same packed arithmetic as epilogue_lean: bit-identical values) into one of two LDS buffers and go on to the next
// tile; the staging waves -- 70 - 80 % idle on these layers, while the multiplying waves spent more than half of their time in the
// epilogue (tools/pc_phase_timing.py: MFMA 31 - 43 %, epilogue 52 - 66 %) -- read the parked tile after the tile's barrier, build the
// pooled copy from it and issue the stores beside the next tile's MFMAs.
// SSRB (round 5, training): the launch is the data gradient of a skip-concat layer and its epilogue the backward of the skip
// operator -- see PipeArgs.ssr_x2.  64-channel tiles = [g0 | g1 | g2 | g3] of 16 skip channels each (the weights' cout order is
// interleaved by the pack kernel), so a lane holds all four members of its four channels in its own accumulators.
// O1C (round 6, inference's last layer): the tile's only result is the fused 1x1 tail (outconv + last activation, unet_parts.py:338-345;
// Unet_singleFrame.py:207-209) computed by the multiplying waves straight from their accumulators -- outc_row: bias, rounding and
// packed ReLU on the registers, two MFMAs against the three 16-bit pieces of the fp32 outconv weights, two adds -- with the weight
// fragments built ONCE per workgroup by the staging waves (LDS, then two registers per multiplying wave for the launch).  No 32-channel
// store, no parked tile: the 16 x 32 tile leaves as 2 KB of fp32.
template <typename T, int NT, int MPW, int MODE, int PW, bool RESW, bool TAIL = false, bool EPO = false, bool SSRB = false, int O1C = 0>
// (NT == 1 && MPW == 2 with four staging waves: 8-row tiles, 64 KB of LDS and <= 128 registers, TWO workgroups per CU, so that
// one workgroup's epilogue runs under the other's MFMAs -- the single-chunk 32-channel layers spend 52 - 61 % of a multiplying
// wave's time in the epilogue)
__global__ __launch_bounds__((4 + PW) * 64, PW == 8 ? 3 : (NT == 1 && MPW == 2 ? 4 : 2)) void conv3x3_pc_kernel(const PipeArgs a) {
  constexpr int NCW = 4;                      // multiplying waves
  static_assert(PW == 4 || PW == 8, "four or eight staging waves");
  using E = Elem<T>;
  using vec = typename Elem<T>::vec;
  using vec4 = typename Elem<T>::vec4;
  constexpr int TH = MPW * 4, TW = 32;
  constexpr int HH = TH + 2, HW = TW + 2;
  constexpr int NPIX = HH * HW;
  constexpr int CT = NT * 32;
  constexpr int WROWS = 9 * CT;
  // (TAIL: 14 x 34 = 476 slots = 12 (mod 16) spreads a staging quad's four planes over distinct banks just as well, and the
  // 8 slots of padding per plane are what lets the second layer's weights fit into the 160 KB)
  // (TAIL keeps the old weight / carry plane lengths: its 160 KB are full to the last 128 bytes)
  constexpr int XPL = TAIL ? NPIX * 16 : pc_plane(NPIX), WPL = TAIL ? pc_plane16(WROWS) : pc_plane(WROWS);      // bytes per plane
  static_assert(!TAIL || NPIX % 16 == 12 || NPIX % 16 == 4, "unpadded planes: quad writes must land on distinct banks");
  constexpr int W1PL = 9 * 32 * 16;                                // TAIL: one (unpadded) plane of the second layer's weights
  constexpr int XBYTES = 4 * XPL, WBYTES = 4 * WPL;
  // streamed weights: two stages of [activations | weights]; resident weights: [X stage 0 | X stage 1 | nk weight chunks]
  constexpr int STAGE = RESW ? XBYTES : XBYTES + WBYTES;
  static_assert(MODE != 4 || (NT == 1 && (MPW == 4 || TAIL || (MPW == 3 && EPO))), "fused up-conv: 16 x 32 (12 x 32: fused last stage, parked epilogue) tiles of 32 channels");
  static_assert(MODE != 5 || (NT == 1 && MPW == 4 && !TAIL && !RESW), "fused 64-channel up-conv: 16 x 32 tiles of 32 channels, streamed weights");
  static_assert(MODE != 3 || (NT == 1 && RESW), "fused first layer: 32 -> 32 channels, one chunk, resident weights");
  static_assert(MPW != 3 || TAIL || EPO, "12-row tiles: the fused last stage, or a parked epilogue (whose pooled copy does not need row pairs per wave)");
  static_assert(!TAIL || (NT == 1 && MPW == 3 && RESW && PW == 8 && MODE == 4), "fused last stage: 12 x 32 x 32 tiles, resident weights");
  constexpr int PW3 = HW + 2, PN3 = (HH + 2) * PW3;            // MODE 3: fp32 image patch under the halo tile
  // TAIL: a tile of the intermediate map is TW columns wide but only XSTEP = TW - 2 columns further than its left neighbour
  // (the second layer's output columns [XSTEP tx, XSTEP tx + XSTEP) need intermediate columns XSTEP tx - 2 ..); its first
  // column is XSTEP tx + XOFF
  constexpr int XSTEP = TAIL ? TW - 2 : TW, XOFF = TAIL ? -2 : 0;
  constexpr int CPL = pc_plane16(2 * HW);                      // TAIL: one plane of the two carried rows
  // fused up-conv, UPT form: one source M-tile per staging wave with all four taps, weights and bias resident in LDS (below)
  constexpr bool UPT = MODE == 5 || (MODE == 4 && !TAIL && PW == 8 && UNCL_PC_UP_TILE);
  constexpr int UPC = MODE == 5 ? 64 : 32, UKS = UPC / 16;        // up-conv channels (in = out), K-steps per source pixel

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wres = smem + 2 * XBYTES;                               // RESW: chunk kc at wres + kc * WBYTES
  // bias ring of four tiles: the multiplying waves read a tile's slot in its epilogue, AFTER the barrier of its last chunk (the
  // barrier sits before the last tap column), while the staging waves may already write the slot of the tile two further on
  float* const sBias = reinterpret_cast<float*>(smem + (RESW ? 2 * XBYTES + a.nk * WBYTES : 2 * STAGE));   // [4 tiles][CT]
  float* const sO1 = sBias + 4 * CT;                                  // fused 1x1 tail: 32 weights + its bias (+ pad)
  float* const sP = sO1 + 64;                                         // MODE 3: [2][PN3] image patches
  float* const sB1 = sO1 + 64;                                        // TAIL: bias of the second layer (32)
  char* const sW1 = reinterpret_cast<char*>(sB1 + 32);                // TAIL: the second layer's weights, [4 planes][9 taps x 32 rows]
  char* const sCarry = sW1 + 4 * W1PL;                                // TAIL: [2][4 planes][2 rows x HW] last two intermediate rows
  static_assert(!EPO || (!TAIL && RESW && ((NT == 1 && (MODE == 0 || MODE == 3 || MODE == 4)) || (NT == 2 && MODE == 0 && MPW == 2))),
                "parked epilogue: 32-channel tiles, or 8-row 64-channel tiles of a plain source; resident weights");
  // EPO: one parked tile, [row][K-slot][pixel] 16-byte vectors.  64-channel tiles (round 6) pad a K-slot's 32 pixels by one vector: the
  // staging waves read eight slots of one pixel with neighbouring lanes, and at a 512-byte pitch all of them start on one bank
  constexpr int PSL = NT == 2 ? 33 * 16 : 32 * 16;                    // bytes per (row, K-slot)
  constexpr int PARKB = TH * (NT * 4) * PSL;
  // Single-chunk layers park into two buffers (the tile's own barrier is the only hand-over).  The four-chunk concat layer (MODE 4,
  // round 6: 12-row tiles, whose stages leave 24 KB of the 160) has ONE: a tile is parked at the end of its last chunk and stored by
  // the staging waves during chunk UNCL_PC_EPO_Q of the NEXT tile, i.e. at least one barrier before the next tile is parked.
  constexpr int NPARK = MODE == 4 ? 1 : 2;
  // (64-channel tiles: two buffers for a single-chunk layer, ONE -- stored during the next tile's first chunk -- for two chunks, whose
  // second weight chunk takes the room: a run-time mask on the buffer index)
  const int park_mask = NPARK == 2 && (NT == 1 || a.nk == 1) ? 1 : 0;
  char* const sPark = MODE == 3 ? reinterpret_cast<char*>(sP + 2 * PN3) : reinterpret_cast<char*>(sO1 + 64);     // EPO: [2][PARKB]
  static_assert(!O1C || (NT == 1 && MODE == 0 && RESW && !TAIL && !EPO && !SSRB), "accumulator-direct 1x1 tail: 32-channel tiles, plain source, resident weights");
  char* const sO1F = reinterpret_cast<char*>(sO1 + 64);               // O1C: the outconv's two A fragments, [2][64 lanes] 16-byte vectors
  float* const sUpB = sO1 + 64;                                       // MODE 5: the up-conv's bias (64) ...
  char* const sUpW = reinterpret_cast<char*>(sUpB + 64);              // ... and its packed weights [4 taps][64 cout][64 cin] (32 KB)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;

  int tile0 = (int)blockIdx.x * a.tiles_per_wg;
  int tile_end = min(tile0 + a.tiles_per_wg, a.total_tiles);
  bool warm = false;
  if (TAIL) {
    // an even share of the (strip, row tile) steps; a share that starts inside a strip first runs the tile above it for the
    // two intermediate rows it hands down (that tile's results are not stored: its owner stores them)
    const int own0 = (int)((long long)blockIdx.x * a.total_tiles / gridDim.x);
    tile_end = (int)((long long)(blockIdx.x + 1) * a.total_tiles / gridDim.x);
    warm = own0 % a.tiles_y != 0;
    tile0 = own0 - (warm ? 1 : 0);
  }
  if (tile0 >= tile_end) return;

  if (wave < NCW) {
    // =================================================================================================================
    // consumers
    // =================================================================================================================
    const int cw = wave;                      // row block of the tile this wave owns
    if ((a.pc_prio & 3) == 1) __builtin_amdgcn_s_setprio(2);
    PcAcc acc[MPW][NT];
    f32x16 zero16;
#pragma unroll
    for (int i = 0; i < 16; ++i) zero16[i] = 0.f;
    // fragment bases of this lane: A = weights row lr, B = halo pixel (wave*MPW) * HW + lr, K-slot plane lh (+ 2 ks)
    const int aoff = lh * WPL + lr * 16;
    const int boff = lh * XPL + (cw * MPW * HW + lr) * 16;

    // One staged K-chunk = six (ks, tx) tap columns of 12 MFMAs each.  The fragments of column i + 1 are read while the MFMAs
    // of column i issue (two register sets, one DS read per MFMA gap pinned with sched_group_barrier): a single wave per SIMD
    // has nobody to cover its LDS latency, and left to itself the compiler emits "9 reads, wait, 12 MFMAs" per column.
    // The accumulators are zeroed after each tile's epilogue, so the phase has no branch and is one scheduling region.
    constexpr int NRD = 3 * NT + MPW + 2;      // DS reads per column
    constexpr int NMM = 3 * NT * MPW;          // MFMAs per column
    // Fragment registers live across the chunk loop: column 0 of chunk s + 1 is requested while column 5 of chunk s multiplies.
    vec A[2][3][NT], B[2][MPW + 2];
    auto rd = [&](const char* st, const char* wst, int set, int col) __attribute__((always_inline)) {
      const char* pa = wst + aoff;
      const char* pb = st + boff;
      const int ks = col / 3, tx = col - 3 * ks;
#pragma unroll
      for (int ty = 0; ty < 3; ++ty)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          A[set][ty][nt] = *reinterpret_cast<const vec*>(pa + 2 * ks * WPL + ((ty * 3 + tx) * CT + nt * 32) * 16);
#pragma unroll
      for (int r = 0; r < MPW + 2; ++r) B[set][r] = *reinterpret_cast<const vec*>(pb + 2 * ks * XPL + (r * HW + tx) * 16);
    };
    auto mfma_col = [&](int col) __attribute__((always_inline)) {
      const int set = col & 1;
#if UNCL_PC_ABL_MASK
      if (PC_ABL(8)) {
#pragma unroll
        for (int ty = 0; ty < 3; ++ty)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) asm volatile("" ::"v"(A[set][ty][nt]));
#pragma unroll
        for (int r = 0; r < MPW + 2; ++r) asm volatile("" ::"v"(B[set][r]));
        return;
      }
#endif
      if (col == 0) {
        // same order as conv3x3_pipe's first column: tap row 0 of every output row, then rows 1 and 2
#pragma unroll
        for (int m = 0; m < MPW; ++m)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[m][nt] = pc_mm(A[set][0][nt], B[set][m], acc[m][nt], 0);
#pragma unroll
        for (int m = 0; m < MPW; ++m)
#pragma unroll
          for (int ty = 1; ty < 3; ++ty)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[m][nt] = pc_mm(A[set][ty][nt], B[set][m + ty], acc[m][nt], ty);
      } else {
#pragma unroll
        for (int m = 0; m < MPW; ++m)
#pragma unroll
          for (int ty = 0; ty < 3; ++ty)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[m][nt] = pc_mm(A[set][ty][nt], B[set][m + ty], acc[m][nt], ty);
      }
    };
    auto sched_reads_under_mfmas = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < NRD; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one DS read of the next column ...
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // ... per MFMA of this one
      }
      __builtin_amdgcn_sched_group_barrier(0x008, NMM - NRD, 0);   // (two reads per gap, all in the first half: no gain)
    };
    // columns 0 .. 4 of the staged chunk (column 0's fragments are already in flight or landed); column 5's are requested
    auto mfma_cols_0_4 = [&](const char* st, const char* wst) __attribute__((always_inline)) {
#pragma unroll
      for (int col = 0; col < 5; ++col) {
        rd(st, wst, (col & 1) ^ 1, col + 1);
        mfma_col(col);
        sched_reads_under_mfmas();
      }
    };

    // Epilogue without an LDS round trip.  A lane holds four consecutive channels (8q + 4 lh ..) of its pixel per register
    // quad; v_permlane32_swap between the two half-waves turns the bf16 pairs of quads (2 qp, 2 qp + 1) into EIGHT consecutive
    // channels per lane (lower half-wave: quad 2 qp, upper: quad 2 qp + 1), i.e. one 16-byte store per lane and a wave
    // instruction that writes 32 contiguous bytes of each of its 32 pixels.
    // ACT 0: ReLU on the rounded bf16 pair (signed 16-bit max against zero; rounding is sign-symmetric); 1: identity
    // (gradient mode); 2: max(t,0) + slope*min(t,0)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    auto epilogue = [&](const TileCur& c, int tpar, auto act_tag) __attribute__((always_inline)) {
      constexpr int ACT = decltype(act_tag)::value;
      const float* sBt = sBias + tpar * CT;
      const int y0 = c.ty * TH + cw * MPW, x0 = c.tx * TW, co = c.ct * CT;
      f32x4 bq[NT][4];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[nt][q] = *reinterpret_cast<const f32x4*>(sBt + nt * 32 + 8 * q + 4 * lh);
      auto act_pack = [&](const PcAcc& v, int q, const f32x4& b) __attribute__((always_inline)) {
        vec4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float t = v[4 * q + r] + b[r];
          o[r] = (T)(ACT == 2 ? fmaxf(t, 0.f) + a.slope * fminf(t, 0.f) : t);
        }
        if (ACT == 0) {
          s16x4 si = __builtin_bit_cast(s16x4, o);
          si = __builtin_elementwise_max(si, s16x4{0, 0, 0, 0});
          o = __builtin_bit_cast(vec4, si);
        }
        return o;
      };
      // two quads of four channels -> this lane's eight consecutive channels (8 (2 qp + lh) ..)
      auto widen = [&](const vec4& o0, const vec4& o1) __attribute__((always_inline)) {
        const u32x2 d0 = __builtin_bit_cast(u32x2, o0), d1 = __builtin_bit_cast(u32x2, o1);
        const auto sx = __builtin_amdgcn_permlane32_swap(d0[0], d1[0], false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(d0[1], d1[1], false, false);
        const u32x4 w = {sx[0], sy[0], sx[1], sy[1]};
        return __builtin_bit_cast(vec, w);
      };
      const int ox = x0 + lr;
      const bool plain = a.mask == nullptr && !a.accumulate;     // wave-uniform
      if (!a.skip_main) {
#pragma unroll
        for (int m = 0; m < MPW; ++m) {
          const int oy = y0 + m;
          const bool in = oy < a.Hout && ox < a.Wout;
          const size_t e0 = (((size_t)c.n * a.Hout + oy) * a.Wout + ox) * a.oC + co + 8 * lh;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int qp = 0; qp < 2; ++qp) {
              vec v = widen(act_pack(acc[m][nt], 2 * qp, bq[nt][2 * qp]), act_pack(acc[m][nt], 2 * qp + 1, bq[nt][2 * qp + 1]));
#ifdef UNCL_PC_TIMING
              if (a.pc_prio & 128) v = E::zero();       // experiment: no epilogue arithmetic (wrong results)
#endif
              const size_t e = e0 + nt * 32 + 16 * qp;
              if (in) {
                if (!plain) {
                  // gradient store: ReLU mask of the producing layer and / or accumulation into an existing gradient
                  float f[8];
                  E::unpack(v, f);
                  if (a.mask) {
                    float mk[8];
                    E::unpack(LD16V(vec, a.mask + e), mk);
#pragma unroll
                    for (int i = 0; i < 8; ++i) f[i] = mk[i] > 0.f ? f[i] : a.mask_slope * f[i];
                  }
                  if (a.accumulate) {
                    float o[8];
                    E::unpack(LD16V(vec, a.out + e), o);
#pragma unroll
                    for (int i = 0; i < 8; ++i) f[i] += o[i];
                  }
                  v = E::pack(f);
                }
#ifdef UNCL_PC_TIMING
                if (a.pc_prio & 64) { asm volatile("" ::"v"(v)); continue; }     // experiment: no output stores (wrong results)
#endif
                UNCL_CHK(a.chk, a.out + e, 16);
                *reinterpret_cast<vec*>(a.out + e) = v;
              }
            }
        }
      }
      if (NT == 2 && MPW == 2 && a.pool_out != nullptr) {
        // MaxPool2d(2) of the wave's two rows (unet_parts.py:212,233): vertical max in registers, horizontal max with the
        // neighbouring pixel's lane through DPP quad_perm [1,0,3,2]; the even lanes store the 16 pooled pixels
        // (16-row tiles of 64 channels: the launcher only takes a pooled copy with the forward epilogue below)
        const int gy = (c.ty * TH >> 1) + cw, gx = (x0 >> 1) + (lr >> 1);
        const bool in = (lr & 1) == 0 && gy < a.pH && gx < a.pW;
        const size_t e0 = (((size_t)c.n * a.pH + gy) * a.pW + gx) * a.oC + co + 8 * lh;
        auto pooled = [&](int nt, int q) __attribute__((always_inline)) {
          const vec4 r0 = act_pack(acc[0][nt], q, bq[nt][q]), r1 = act_pack(acc[MPW - 1][nt], q, bq[nt][q]);
          vec4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float mv = fmaxf((float)r0[r], (float)r1[r]);
            const float other = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, mv), 0xB1, 0xF, 0xF, true));
            o[r] = (T)fmaxf(mv, other);
          }
          return o;
        };
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int qp = 0; qp < 2; ++qp) {
            const vec v = widen(pooled(nt, 2 * qp), pooled(nt, 2 * qp + 1));
            if (in) { UNCL_CHK(a.chk, a.pool_out + e0 + nt * 32 + 16 * qp, 16); *reinterpret_cast<vec*>(a.pool_out + e0 + nt * 32 + 16 * qp) = v; }
          }
      }
    };

    // The forward epilogue (bias + ReLU, plain store, optional pooled copy) with nothing in it that is not arithmetic on
    // packed registers: 32 v_pk_add_f32 + 32 v_cvt_pk + 16 v_permlane32_swap + 32 v_pk_max_i16 per 64 accumulators, stores
    // as "scalar row base + 32-bit lane offset + immediate".  The generic form above costs 3-7x the instructions (64-bit
    // per-lane addresses, per-element selects, float round trips in the pooled copy) and was half of a 64-channel tile's time.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    // Units (pr, nt, qp): rows 2 pr and 2 pr + 1 of the wave, channels 32 nt + 16 qp.. of them (two row stores) and the same
    // channels of pooled row pr.
    // GRAD = 0: forward (ReLU, optional pooled copy); GRAD = 1: gradient store (identity, ReLU mask of the producing layer and /
    // or accumulation into the gradient already there), on the same packed registers
    auto epilogue_fast = [&](const TileCur& c, int tpar, auto grad_tag) __attribute__((always_inline)) {
      constexpr bool GRAD = decltype(grad_tag)::value != 0;
      const float* sBt = sBias + tpar * CT;
      const int y0 = c.ty * TH + cw * MPW, x0 = c.tx * TW, co = c.ct * CT;
      const int ox = x0 + lr;
      auto pack4 = [&](const PcAcc& v, int q, const f32x4& b) __attribute__((always_inline)) {
        const f32x2 s0 = f32x2{v[4 * q], v[4 * q + 1]} + f32x2{b[0], b[1]};
        const f32x2 s1 = f32x2{v[4 * q + 2], v[4 * q + 3]} + f32x2{b[2], b[3]};
        vec4 o;
        o[0] = (T)s0[0]; o[1] = (T)s0[1]; o[2] = (T)s1[0]; o[3] = (T)s1[1];
        return o;
      };
      auto widen_relu = [&](const vec4& o0, const vec4& o1) __attribute__((always_inline)) {
        const u32x2 d0 = __builtin_bit_cast(u32x2, o0), d1 = __builtin_bit_cast(u32x2, o1);
        const auto sx = __builtin_amdgcn_permlane32_swap(d0[0], d1[0], false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(d0[1], d1[1], false, false);
        const u32x4 w = {sx[0], sy[0], sx[1], sy[1]};
        s16x8 si = __builtin_bit_cast(s16x8, w);
        if (!GRAD) si = __builtin_elementwise_max(si, s16x8{0, 0, 0, 0, 0, 0, 0, 0});    // ReLU on the rounded values (sign-symmetric rounding)
        return __builtin_bit_cast(vec, si);
      };
      // gradient store: zero where the producing layer's ReLU output is not positive (a signed 16-bit compare orders bf16 / f16
      // values of one sign like their floats: mask > 0 <=> int16(mask) > 0; 0 / 1 factor, 16-bit multiply), then add the gradient
      // already in memory in fp32 and round again -- the operation order of the generic epilogue
      auto grad_ops = [&](vec w, const char* mrow, const char* orow, unsigned off) __attribute__((always_inline)) {
        if (a.mask != nullptr) {
          UNCL_CHK(a.chk, mrow + off, 16);
          const s16x8 mk = __builtin_bit_cast(s16x8, *reinterpret_cast<const vec*>(mrow + off));
          const s16x8 one = {1, 1, 1, 1, 1, 1, 1, 1}, zero = {0, 0, 0, 0, 0, 0, 0, 0};
          const s16x8 keep = __builtin_elementwise_max(__builtin_elementwise_min(mk, one), zero);
          w = __builtin_bit_cast(vec, (s16x8)(__builtin_bit_cast(s16x8, w) * keep));
        }
        if (a.accumulate) {
          float f[8], o[8];
          E::unpack(w, f);
          UNCL_CHK(a.chk, orow + off, 16);
          E::unpack(*reinterpret_cast<const vec*>(orow + off), o);
#pragma unroll
          for (int i = 0; i < 8; ++i) f[i] += o[i];
          w = E::pack(f);
        }
        return w;
      };
      const bool xin = ox < a.Wout;
      unsigned loff = (unsigned)(ox * a.oC + 8 * lh) * 2u;
      asm volatile("" : "+v"(loff));        // keep the 32-bit lane offset (global_store ... v_off, s[base] offset:imm)
      const int gx = (x0 >> 1) + (lr >> 1);
      unsigned poff = (unsigned)(gx * a.oC + 8 * lh) * 2u;
      asm volatile("" : "+v"(poff));
      static_assert(MPW % 2 == 0, "whole pooled rows per wave");
#pragma unroll
      for (int pr = 0; pr < MPW / 2; ++pr) {
        float o1sum[2] = {0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int qp = 0; qp < 2; ++qp) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(sBt + nt * 32 + 16 * qp + 4 * lh);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(sBt + nt * 32 + 16 * qp + 8 + 4 * lh);
            vec wv[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
              const int m = 2 * pr + r, oy = y0 + m;                              // wave-uniform
              char* rowp = reinterpret_cast<char*>(a.out + ((size_t)c.n * a.Hout + oy) * a.Wout * a.oC + co);
              wv[r] = widen_relu(pack4(acc[m][nt], 2 * qp, b0), pack4(acc[m][nt], 2 * qp + 1, b1));
              if (!GRAD && NT == 1 && a.out1_w != nullptr) {
                // fused 1x1 tail (outconv + last activation, unet_parts.py:338-345): one sequential fmaf chain over the 32
                // ROUNDED channels in channel order, as the four-wave kernel computes it from its LDS image.  After the
                // widening the lower half-wave holds channels 16 qp + 0..7 of its pixel and the upper half 16 qp + 8..15: the
                // running sum goes lower -> upper -> (next qp) lower -> upper, handed over by a half-wave exchange.
                float f[8];
                E::unpack(wv[r], f);
                const float* w8 = sO1 + 16 * qp + 8 * lh;
                float t = qp == 0 ? sO1[32] : o1sum[r];                          // lower half: the chain so far
#pragma unroll
                for (int i = 0; i < 8; ++i) t = fmaf(f[i], w8[i], t);
                float u = __shfl_xor(t, 32, 64);                                 // upper half: the lower half's sum so far
#pragma unroll
                for (int i = 0; i < 8; ++i) u = fmaf(f[i], w8[i], u);
                o1sum[r] = __shfl_xor(u, 32, 64);                                // lower half: the upper half's sum
                // (half-wave exchanges: only the receiving half of each hand-over holds the chain; the final sum is read in
                // the LOWER half)
                if (qp == 1 && lh == 0 && oy < a.Hout && xin)
                  { UNCL_CHK(a.chk, a.out1 + ((size_t)c.n * a.Hout + oy) * a.Wout + ox, 4); a.out1[((size_t)c.n * a.Hout + oy) * a.Wout + ox] = uncl_act(o1sum[r], a.out1_act); }
              }
              if (a.skip_main) continue;
#ifdef UNCL_PC_TIMING
              if (a.pc_prio & 64) { asm volatile("" ::"v"(wv[r])); continue; }     // experiment: no output stores (wrong results)
#endif
              if (oy < a.Hout && xin) {
                if (GRAD) {
                  const char* mrow = reinterpret_cast<const char*>(a.mask + ((size_t)c.n * a.Hout + oy) * a.Wout * a.oC + co);
                  wv[r] = grad_ops(wv[r], mrow, rowp, loff + (nt * 32 + 16 * qp) * 2);
                }
                UNCL_CHK(a.chk, rowp + loff + (nt * 32 + 16 * qp) * 2, 16);
                *reinterpret_cast<vec*>(rowp + loff + (nt * 32 + 16 * qp) * 2) = wv[r];
              }
            }
            if (!GRAD && a.pool_out != nullptr) {
              // MaxPool2d(2) of the two rows (unet_parts.py:212,233) on the packed, non-negative values: a signed 16-bit
              // maximum orders them like their floats; vertical maximum between the rows, horizontal with the neighbouring
              // pixel's lane (DPP quad_perm [1,0,3,2]); the even lanes store the 16 pooled pixels
              const int gy = (c.ty * TH >> 1) + cw * (MPW / 2) + pr;
              const bool in = (lr & 1) == 0 && gy < a.pH && gx < a.pW;
              char* rowp = reinterpret_cast<char*>(a.pool_out + ((size_t)c.n * a.pH + gy) * a.pW * a.oC + co);
              s16x8 v = __builtin_elementwise_max(__builtin_bit_cast(s16x8, wv[0]), __builtin_bit_cast(s16x8, wv[1]));
              u32x4 u = __builtin_bit_cast(u32x4, v), h;
#pragma unroll
              for (int i = 0; i < 4; ++i) h[i] = (unsigned)__builtin_amdgcn_mov_dpp((int)u[i], 0xB1, 0xF, 0xF, true);
              v = __builtin_elementwise_max(v, __builtin_bit_cast(s16x8, h));
#ifdef UNCL_PC_TIMING
              if (a.pc_prio & 64) { asm volatile("" ::"v"(v)); continue; }
#endif
              if (in) { UNCL_CHK(a.chk, rowp + poff + (nt * 32 + 16 * qp) * 2, 16); *reinterpret_cast<vec*>(rowp + poff + (nt * 32 + 16 * qp) * 2) = __builtin_bit_cast(vec, v); }
            }
          }
      }
    };

    TileCur cc;
    cur_init<TAIL>(cc, tile0, a);
    int tpar = 0;
    const bool fast_relu = a.slope == 0.f && a.mask == nullptr && !a.accumulate && (!a.skip_main || a.out1_w != nullptr) &&
                           !(a.pc_prio & 256);   // wave-uniform
    // gradient stores of ReLU networks (mask slope 0), and plain identity stores
    const bool fast_grad = a.slope == 1.f && a.mask_slope == 0.f && !a.skip_main && a.pool_out == nullptr && !(a.pc_prio & (256 | 1024));
#pragma unroll
    for (int m = 0; m < MPW; ++m)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[m][nt] = zero16;
    // Accumulators are cleared by the matrix pipe itself: D = 0 x 0 + 0 (inline-zero C) is one instruction per accumulator,
    // where sixteen v_mov each -- 64 / 128 per tile -- were 5 - 8 % of a short-K tile's time on a wave that is alone on its
    // SIMD; the pipe idles during the epilogue anyway.  (The operand is made opaque so that the product is not folded back into
    // moves; starting a tile's first MFMAs from an inline-zero C instead needs a second copy of every phase, and every
    // formulation of that sent the register allocator into spills.)
    auto zero_acc = [&]() __attribute__((always_inline)) {
      vec zv = E::zero();
      asm volatile("" : "+v"(zv));
#pragma unroll
      for (int m = 0; m < MPW; ++m)
#pragma unroll
#if defined(UNCL_PC_MFMA16_PROXY)
        for (int nt = 0; nt < NT; ++nt) acc[m][nt] = pc_mm_zero(zv);
#else
        for (int nt = 0; nt < NT; ++nt) acc[m][nt] = mfma32x16(zv, zv, zero16);
#endif
    };
    // The plain forward store (bias + ReLU, optional pooled copy; no fused 1x1 tail) once more, as straight-line code.  A
    // multiplying wave is alone on its SIMD and the matrix pipe idles while it runs its epilogue, so the epilogue costs its
    // INSTRUCTION COUNT (4 - 5 cycles each; ablation in a timing build: without its stores it is still 44 % of the first
    // layer's step): epilogue_fast spends ~650 instructions per 32-channel tile, most of them per-row 64-bit address arithmetic
    // on the scalar unit and an exec-mask branch around every store.  Here one buffer descriptor per tile (base = the sample,
    // num_records = its bytes) replaces both: a store's offset is "lane offset + row offset", where an out-of-image column or
    // row contributes 2^30, i.e. lands beyond num_records and is dropped by the hardware (every sample is < 1 GiB: launcher).
    typedef unsigned u32x4l __attribute__((ext_vector_type(4)));
    // OUT1 (32-channel tiles): the fused 1x1 tail (outconv + last activation, unet_parts.py:338-345) on the ROUNDED, activated
    // channels, as four MFMAs per row (outc_row above).  1: also store the 32-channel map; 2: only the 1-channel map (inference
    // does not need up_x).
    // The fused 1x1 outconv on the matrix cores.  The 32x32 accumulator of a row (channel = register index, pixel = lane) IS a B
    // operand once its registers 8 s .. 8 s + 7 are rounded pairwise to 16 bits -- exactly the rounding the stored map would
    // get: element j of lane half h is channel 16 s + 8 (j >> 2) + 4 h + (j & 3).  The fp32 outconv weights enter as THREE 16-bit
    // pieces (w = p0 + p1 + p2 to 2^-25 |w|; the products are exact in the fp32 accumulator) in ROWS 0, 1, 2 of the A operand
    // (the other rows are zero): D rows 0..2 -- registers 0..2 of the lower half-wave -- are the three partial dot products of
    // the lane's pixel.  Two MFMAs and two adds per row replace 16 unpacks + 16 fma + a half-wave exchange.
    auto outc_frags = [&](vec (&wp)[2]) __attribute__((always_inline)) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float w = sO1[16 * ks + 8 * (j >> 2) + 4 * lh + (j & 3)];
          T piece = (T)0.f;
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const T h = (T)w;
            if (lr == t) piece = h;
            w -= (float)h;
          }
          wp[ks][j] = piece;
        }
    };
    // bias + ReLU + rounding of one row's accumulator, then the dot product with the outconv weights (without its bias); the
    // result is valid in the LOWER half-wave
    auto outc_row = [&](const PcAcc& v, const float* bsrc, const vec (&wp)[2]) __attribute__((always_inline)) {
      vec Bf[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        typedef float f32x2o __attribute__((ext_vector_type(2)));
        typedef short s16x8o __attribute__((ext_vector_type(8)));
        vec o;
#pragma unroll
        for (int hq = 0; hq < 2; ++hq) {
          const int q = 2 * ks + hq;
          const f32x4 b = *reinterpret_cast<const f32x4*>(bsrc + 8 * q + 4 * lh);
          const f32x2o s0 = f32x2o{v[4 * q], v[4 * q + 1]} + f32x2o{b[0], b[1]};
          const f32x2o s1 = f32x2o{v[4 * q + 2], v[4 * q + 3]} + f32x2o{b[2], b[3]};
          o[4 * hq] = (T)s0[0]; o[4 * hq + 1] = (T)s0[1]; o[4 * hq + 2] = (T)s1[0]; o[4 * hq + 3] = (T)s1[1];
        }
        s16x8o si = __builtin_bit_cast(s16x8o, o);
        si = __builtin_elementwise_max(si, s16x8o{0, 0, 0, 0, 0, 0, 0, 0});        // ReLU on the rounded values
        Bf[ks] = __builtin_bit_cast(vec, si);
      }
      f32x16 d = mfma32x16(wp[0], Bf[0], zero16);
      d = mfma32x16(wp[1], Bf[1], d);
      return (d[2] + d[1]) + d[0];             // smallest piece first
    };
    auto epilogue_lean = [&](const TileCur& c, int tp, auto pool_tag, auto out1_tag) __attribute__((always_inline)) {
      constexpr bool POOL = decltype(pool_tag)::value != 0;
      constexpr int OUT1 = decltype(out1_tag)::value;
      static_assert(OUT1 == 0 || !POOL, "the fused 1x1 tail has no pooled copy");
      constexpr unsigned BAD = 0x40000000u;
      const float* sBt = sBias + tp * CT;
      const int y0 = c.ty * TH + cw * MPW, x0 = c.tx * TW, co = c.ct * CT;
      const int ox = x0 + lr;
      const unsigned sample = (unsigned)(a.Hout * a.Wout * a.oC) * 2u;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.out) + (size_t)c.n * sample, (short)0,
                                                                          (int)sample, 0x00020000);
      // (both arms of the selects are constants: an arm with arithmetic in it is compiled as a branch)
      const unsigned loff = ((unsigned)(ox * a.oC + co + 8 * lh) * 2u) | (ox < a.Wout ? 0u : BAD);
      const unsigned rowb = (unsigned)(a.Wout * a.oC) * 2u;
      __amdgpu_buffer_rsrc_t prs = rs;
      unsigned ploff = BAD, prowb = 0;
      if (POOL) {
        const unsigned psample = (unsigned)(a.pH * a.pW * a.oC) * 2u;
        const int gx = (x0 >> 1) + (lr >> 1);
        prs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.pool_out) + (size_t)c.n * psample, (short)0, (int)psample, 0x00020000);
        ploff = ((unsigned)(gx * a.oC + co + 8 * lh) * 2u) | (gx < a.pW ? 0u : BAD);
        prowb = (unsigned)(a.pW * a.oC) * 2u;
      }
      auto pack4 = [&](const PcAcc& v, int q, const f32x4& b) __attribute__((always_inline)) {
        const f32x2 s0 = f32x2{v[4 * q], v[4 * q + 1]} + f32x2{b[0], b[1]};
        const f32x2 s1 = f32x2{v[4 * q + 2], v[4 * q + 3]} + f32x2{b[2], b[3]};
        vec4 o;
        o[0] = (T)s0[0]; o[1] = (T)s0[1]; o[2] = (T)s1[0]; o[3] = (T)s1[1];
        return o;
      };
      auto widen_relu = [&](const vec4& o0, const vec4& o1) __attribute__((always_inline)) {
        const u32x2 d0 = __builtin_bit_cast(u32x2, o0), d1 = __builtin_bit_cast(u32x2, o1);
        const auto sx = __builtin_amdgcn_permlane32_swap(d0[0], d1[0], false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(d0[1], d1[1], false, false);
        const u32x4l w = {sx[0], sy[0], sx[1], sy[1]};
        s16x8 si = __builtin_bit_cast(s16x8, w);
        si = __builtin_elementwise_max(si, s16x8{0, 0, 0, 0, 0, 0, 0, 0});    // ReLU on the rounded values (sign-symmetric rounding)
        return si;
      };
#pragma unroll
      for (int pr = 0; pr < MPW / 2; ++pr) {
        unsigned voff[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int oy = y0 + 2 * pr + r;                                    // wave-uniform
          voff[r] = loff + (((unsigned)oy * rowb) | (oy < a.Hout ? 0u : BAD));
        }
        unsigned pvoff = BAD;
        if (POOL) {
          const int gy = (c.ty * TH >> 1) + cw * (MPW / 2) + pr;
          pvoff = ploff + (((unsigned)gy * prowb) | (gy < a.pH ? 0u : BAD));
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int qp = 0; qp < 2; ++qp) {
            if (OUT1 == 2) continue;                 // only the one-channel map is wanted: no widened 16-byte vectors
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(sBt + nt * 32 + 16 * qp + 4 * lh);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(sBt + nt * 32 + 16 * qp + 8 + 4 * lh);
            s16x8 wv[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
              const int m = 2 * pr + r;
              wv[r] = widen_relu(pack4(acc[m][nt], 2 * qp, b0), pack4(acc[m][nt], 2 * qp + 1, b1));
#if defined(UNCL_PC_TIMING) || defined(UNCL_PC_ABLATE)
              // ablations (wrong results): 128 = no conversion arithmetic (the raw accumulator bits are stored), 64 = no stores
              if (a.pc_prio & 128) wv[r] = __builtin_bit_cast(s16x8, f32x4{acc[m][nt][4 * qp], acc[m][nt][4 * qp + 1], acc[m][nt][4 * qp + 2], acc[m][nt][4 * qp + 3]});
              if (a.pc_prio & 64) { asm volatile("" ::"v"(wv[r])); continue; }
#endif
              if (OUT1 != 2)
              {
#ifdef UNCL_CHECKED
                const unsigned bo = voff[r] + (unsigned)(nt * 32 + 16 * qp) * 2u;
                if (bo < BAD) UNCL_CHK(a.chk, reinterpret_cast<const char*>(a.out) + (size_t)c.n * sample + bo, 16);
#endif
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4l, wv[r]), rs, voff[r] + (unsigned)(nt * 32 + 16 * qp) * 2u, 0, UNCL_PC_STORE_AUX);
              }
            }
            if (POOL) {
              // MaxPool2d(2) of the two rows (unet_parts.py:212,233) on the packed, non-negative values: a signed 16-bit
              // maximum orders them like their floats; vertical, then horizontal with the neighbouring pixel's lane
              s16x8 v = __builtin_elementwise_max(wv[0], wv[1]);
              u32x4l u = __builtin_bit_cast(u32x4l, v), h;
#pragma unroll
              for (int i = 0; i < 4; ++i) h[i] = (unsigned)__builtin_amdgcn_mov_dpp((int)u[i], 0xB1, 0xF, 0xF, true);
              v = __builtin_elementwise_max(v, __builtin_bit_cast(s16x8, h));
              // (the odd lanes hold the same maxima: masked off by EXEC rather than by offset, so that the memory pipeline
              // sees a 32-lane store)
              if ((lr & 1) == 0)
              {
#ifdef UNCL_CHECKED
                const unsigned bo = pvoff + (unsigned)(nt * 32 + 16 * qp) * 2u;
                if (bo < BAD) UNCL_CHK(a.chk, reinterpret_cast<const char*>(a.pool_out) + (size_t)c.n * ((unsigned)(a.pH * a.pW * a.oC) * 2u) + bo, 16);
#endif
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4l, v), prs, pvoff + (unsigned)(nt * 32 + 16 * qp) * 2u, 0, UNCL_PC_STORE_AUX);
              }
            }
          }
        if (OUT1 != 0) {
          // (this path is a test / A-B form: the weight fragments are rebuilt per tile rather than kept in registers)
          vec o1w[2];
          outc_frags(o1w);
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const int oy = y0 + 2 * pr + r;
            const float tot = outc_row(acc[2 * pr + r][0], sBt, o1w) + sO1[32];
            if (lh == 0 && oy < a.Hout && ox < a.Wout) { UNCL_CHK(a.chk, a.out1 + ((size_t)c.n * a.Hout + oy) * a.Wout + ox, 4); a.out1[((size_t)c.n * a.Hout + oy) * a.Wout + ox] = uncl_act(tot, a.out1_act); }
          }
        }
      }
    };
    const bool lean = fast_relu && a.lean && (NT == 1 || a.out1_w == nullptr);      // wave-uniform
    // EPO: the tile as epilogue_lean would have stored it (bias + rounding + widening + ReLU), parked in LDS for the staging waves
    int pk = 0;                           // tiles parked so far: tile k goes to buffer k & 1
    auto park = [&](int tp) __attribute__((always_inline)) {
      const float* sBt = sBias + tp * CT;
      char* const pb = sPark + (pk & park_mask) * PARKB + (cw * MPW * (NT * 4) + lh) * PSL + lr * 16;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(sBt + nt * 32 + 16 * qp + 4 * lh);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(sBt + nt * 32 + 16 * qp + 8 + 4 * lh);
#pragma unroll
        for (int m = 0; m < MPW; ++m) {
          const PcAcc& v = acc[m][nt];
          auto p4 = [&](int q, const f32x4& b) __attribute__((always_inline)) {
            const f32x2 s0 = f32x2{v[4 * q], v[4 * q + 1]} + f32x2{b[0], b[1]};
            const f32x2 s1 = f32x2{v[4 * q + 2], v[4 * q + 3]} + f32x2{b[2], b[3]};
            vec4 o;
            o[0] = (T)s0[0]; o[1] = (T)s0[1]; o[2] = (T)s1[0]; o[3] = (T)s1[1];
            return o;
          };
          const u32x2 d0 = __builtin_bit_cast(u32x2, p4(2 * qp, b0)), d1 = __builtin_bit_cast(u32x2, p4(2 * qp + 1, b1));
          const auto sx = __builtin_amdgcn_permlane32_swap(d0[0], d1[0], false, false);
          const auto sy = __builtin_amdgcn_permlane32_swap(d0[1], d1[1], false, false);
          const u32x4 w = {sx[0], sy[0], sx[1], sy[1]};
          s16x8 si = __builtin_bit_cast(s16x8, w);
          si = __builtin_elementwise_max(si, s16x8{0, 0, 0, 0, 0, 0, 0, 0});    // ReLU on the rounded values
          *reinterpret_cast<s16x8*>(pb + (m * (NT * 4) + nt * 4 + 2 * qp) * PSL) = si;
        }
      }
      ++pk;
    };
    // SSRB: g_x2 / g_x1 of the tile's 16 skip channels from the accumulators (no bias in a data gradient).  Per row and quad h (the
    // lane's channels 16 ct + 8 h + 4 lh + r): g0 = acc[m][0][4 h + r], g1 = acc[m][0][8 + 4 h + r], g2 / g3 the same of acc[m][1].
    // The result quads of the two half-waves are widened to eight consecutive channels per lane (v_permlane32_swap) and leave
    // as one 16-byte buffer store per lane and tensor; x2 comes in as two 8-byte buffer loads per row (out-of-image lanes read 0).
    auto epilogue_ssr = [&](const TileCur& c) __attribute__((always_inline)) {
      static_assert(!SSRB || NT == 2, "the skip operator's backward needs the four members of a channel in one tile");
      constexpr unsigned BAD = 0x40000000u;
      const int y0 = c.ty * TH + cw * MPW, ox = c.tx * TW + lr;
      const int C = a.ssr_C;
      const unsigned sample = (unsigned)(a.Hout * a.Wout * C) * 2u;
      const size_t sb = (size_t)c.n * sample;
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(a.ssr_x2)) + sb, (short)0, (int)sample, 0x00020000);
      const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.ssr_gx2) + sb, (short)0, (int)sample, 0x00020000);
      const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.ssr_gx1) + sb, (short)0, (int)sample, 0x00020000);
      const unsigned colb = ((unsigned)(ox * C + c.ct * 16) * 2u) | (ox < a.Wout ? 0u : BAD);
      const unsigned rowb = (unsigned)(a.Wout * C) * 2u;
      typedef unsigned u32x2b __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int m = 0; m < MPW; ++m) {
        const int oy = y0 + m;
        const unsigned po = colb + (((unsigned)oy * rowb) | (oy < a.Hout ? 0u : BAD));
        u32x2 d2[2], d1[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          // this lane's four x2 channels of the quad: 8 bytes at channel 8 h + 4 lh
          const unsigned xo = po + (unsigned)(8 * h + 4 * lh) * 2u;
#ifdef UNCL_CHECKED
          if (xo < BAD) UNCL_CHK(a.chk, reinterpret_cast<const char*>(a.ssr_x2) + sb + xo, 8);
#endif
          const u32x2b xr = __builtin_bit_cast(u32x2b, __builtin_amdgcn_raw_buffer_load_b64(rx, xo, 0, 0));
          const vec4 xv = __builtin_bit_cast(vec4, xr);
          vec4 o2, o1;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float x = (float)xv[r];
            const float g0 = acc[m][0][4 * h + r], g1 = acc[m][0][8 + 4 * h + r], g2 = acc[m][1][4 * h + r], g3 = acc[m][1][8 + 4 * h + r];
            const float t = g0 + 2.f * x * g2 + g3 * 0.5f / __builtin_amdgcn_sqrtf(x + 1e-8f);
            o2[r] = (T)(x > 0.f ? t : a.mask_slope * t);
            o1[r] = (T)g1;
          }
          d2[h] = __builtin_bit_cast(u32x2, o2);
          d1[h] = __builtin_bit_cast(u32x2, o1);
        }
        // quads (h = 0, h = 1) of the two half-waves -> eight consecutive channels per lane (lower half-wave: channels 0..7 of the
        // tile's 16, upper: 8..15)
        auto widen2 = [&](const u32x2& q0, const u32x2& q1) __attribute__((always_inline)) {
          const auto sx = __builtin_amdgcn_permlane32_swap(q0[0], q1[0], false, false);
          const auto sy = __builtin_amdgcn_permlane32_swap(q0[1], q1[1], false, false);
          return u32x4{sx[0], sy[0], sx[1], sy[1]};
        };
        u32x4 w2 = widen2(d2[0], d2[1]);
        const u32x4 w1 = widen2(d1[0], d1[1]);
        const unsigned so = po + (unsigned)(8 * lh) * 2u;
        if (a.ssr_acc) {
          float f[8], o[8];
          E::unpack(__builtin_bit_cast(vec, w2), f);
          E::unpack(__builtin_bit_cast(vec, __builtin_amdgcn_raw_buffer_load_b128(r2, so, 0, 0)), o);
#pragma unroll
          for (int i = 0; i < 8; ++i) f[i] += o[i];
          w2 = __builtin_bit_cast(u32x4, E::pack(f));
        }
#ifdef UNCL_CHECKED
        if (so < BAD) { UNCL_CHK(a.chk, reinterpret_cast<const char*>(a.ssr_gx2) + sb + so, 16); UNCL_CHK(a.chk, reinterpret_cast<const char*>(a.ssr_gx1) + sb + so, 16); }
#endif
        __builtin_amdgcn_raw_buffer_store_b128(w2, r2, so, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(w1, r1, so, 0, 0);
      }
    };
    // O1C: rows of the tile -> outc_row -> last activation -> fp32 (the lower half-wave holds the sums)
    vec o1c_w[2];
    auto epilogue_o1c = [&](const TileCur& c, int tp) __attribute__((always_inline)) {
      // Straight-line code: the four rows' chains (rounding -> two MFMAs -> sum -> activation) interleave only if nothing between
      // them branches.  The activation selector is wave-uniform and tested once per tile, not once per row (uncl_act's switch was a
      // dozen scalar branches per row); rows and columns outside the map add 2^30 to the store's offset, which the buffer
      // descriptor (one per sample, num_records = its bytes) drops, as in epilogue_lean; the upper half-wave (whose registers do
      // not hold the sums) is dropped the same way.
      constexpr unsigned BAD = 0x40000000u;
      const float* sBt = sBias + tp * CT;
      const int y0 = c.ty * TH + cw * MPW, ox = c.tx * TW + lr;
      const unsigned sample = (unsigned)(a.Hout * a.Wout) * 4u;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.out1) + (size_t)c.n * sample, (short)0,
                                                                          (int)sample, 0x00020000);
      const unsigned loff = ((unsigned)ox * 4u) | ((lh == 0 && ox < a.Wout) ? 0u : BAD);
      const unsigned rowb = (unsigned)a.Wout * 4u;
      // outc_row with the tile's bias quads read once (not once per row): the same operations on the same values
      f32x4 bq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const f32x4*>(sBt + 8 * q + 4 * lh);
      const float ob = sO1[32];
      typedef float f32x2o __attribute__((ext_vector_type(2)));
      typedef short s16x8o __attribute__((ext_vector_type(8)));
      vec Bf[MPW][2];
#pragma unroll
      for (int m = 0; m < MPW; ++m)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const PcAcc& v = acc[m][0];
          vec o;
#pragma unroll
          for (int hq = 0; hq < 2; ++hq) {
            const int q = 2 * ks + hq;
            const f32x2o s0 = f32x2o{v[4 * q], v[4 * q + 1]} + f32x2o{bq[q][0], bq[q][1]};
            const f32x2o s1 = f32x2o{v[4 * q + 2], v[4 * q + 3]} + f32x2o{bq[q][2], bq[q][3]};
            o[4 * hq] = (T)s0[0]; o[4 * hq + 1] = (T)s0[1]; o[4 * hq + 2] = (T)s1[0]; o[4 * hq + 3] = (T)s1[1];
          }
          s16x8o si = __builtin_bit_cast(s16x8o, o);
          si = __builtin_elementwise_max(si, s16x8o{0, 0, 0, 0, 0, 0, 0, 0});        // ReLU on the rounded values
          Bf[m][ks] = __builtin_bit_cast(vec, si);
        }
      float tot[MPW];
      f32x16 dd[MPW];
#pragma unroll
      for (int m = 0; m < MPW; ++m) dd[m] = mfma32x16(o1c_w[0], Bf[m][0], zero16);
#pragma unroll
      for (int m = 0; m < MPW; ++m) dd[m] = mfma32x16(o1c_w[1], Bf[m][1], dd[m]);
#pragma unroll
      for (int m = 0; m < MPW; ++m) tot[m] = ((dd[m][2] + dd[m][1]) + dd[m][0]) + ob;     // smallest piece first, as outc_row
      if (a.out1_act == UNCL_ACT_SIGMOID) {
#pragma unroll
        for (int m = 0; m < MPW; ++m) tot[m] = uncl_sigmoid(tot[m]);
      } else {
#pragma unroll
        for (int m = 0; m < MPW; ++m) tot[m] = uncl_act(tot[m], a.out1_act);
      }
#pragma unroll
      for (int m = 0; m < MPW; ++m) {
        const int oy = y0 + m;                                     // wave-uniform
        const unsigned off = loff + (((unsigned)oy * rowb) | (oy < a.Hout ? 0u : BAD));
#ifdef UNCL_CHECKED
        if (off < BAD) UNCL_CHK(a.chk, reinterpret_cast<const char*>(a.out1) + (size_t)c.n * sample + off, 4);
#endif
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, tot[m]), rs, off, 0, 0);
      }
    };
    auto run_epilogue = [&](const TileCur& c, auto tp) __attribute__((always_inline)) {      // (generic: instantiated only where called)
      if constexpr (O1C != 0) {
        epilogue_o1c(c, tp);
      } else if constexpr (SSRB) {
        epilogue_ssr(c);
      } else if constexpr (EPO) {
        if (!PC_ABL(16)) park(tp);
      } else {
        if (lean) {
          if (NT == 1 && a.out1_w != nullptr) {
            if (a.skip_main) epilogue_lean(c, tp, IntTag<0>{}, IntTag<NT == 1 ? 2 : 0>{});
            else epilogue_lean(c, tp, IntTag<0>{}, IntTag<NT == 1 ? 1 : 0>{});
          } else if (a.pool_out != nullptr) epilogue_lean(c, tp, IntTag<1>{}, IntTag<0>{});
          else epilogue_lean(c, tp, IntTag<0>{}, IntTag<0>{});
        } else if (fast_relu) epilogue_fast(c, tp, IntTag<0>{});
        else if (fast_grad) epilogue_fast(c, tp, IntTag<1>{});
        else if (a.slope == 0.f) epilogue(c, tp, IntTag<0>{});
        else if (a.slope == 1.f) epilogue(c, tp, IntTag<1>{});
        else epilogue(c, tp, IntTag<2>{});
      }
      zero_acc();
    };
    if constexpr (TAIL) {
      // ===============================================================================================================
      // Fused last decoder stage (inference; unet_parts.py:149-162, 338-345; Unet_singleFrame.py:207-209):
      //   tile t of a strip:  chunks 0..3 (concat-ssr K = 128) -> 12 x 32 x 32-channel tile of the intermediate map
      //                       -> bias + ReLU + rounding, written to an LDS image (rows 2..13 of stage 1; image rows 0 / 1 are
      //                          the last two rows of the previous tile of the strip, kept in a small double buffer)
      //                       -> second transposed 3x3 (K = 32 x 9, weights resident in LDS) from that image: 12 rows x 30 columns
      //                       -> bias + ReLU + rounding + 1x1 outconv + last activation -> one fp32 channel to memory.
      // The 32-channel maps of both layers never reach HBM (826 MB written and 1.05 GB read back per 200 tiles otherwise).
      // Barriers per tile: four chunk barriers (before each chunk's last tap column, as in the loop below), one after the
      // image is written, one before the second layer's last tap column; the staging waves take the two extra ones idle.
      // ===============================================================================================================
      char* const img = smem + STAGE;                   // stage 1: every tile's last chunk (3) sits there
      if (cw == 3)                                      // a strip's first tile sees zero rows above it
        for (int i = lane; i < 2 * 4 * CPL / 16; i += 64) *reinterpret_cast<vec*>(sCarry + i * 16) = E::zero();
      auto t_pack4 = [&](const PcAcc& v, int q, const f32x4& b) __attribute__((always_inline)) {
        const f32x2 s0 = f32x2{v[4 * q], v[4 * q + 1]} + f32x2{b[0], b[1]};
        const f32x2 s1 = f32x2{v[4 * q + 2], v[4 * q + 3]} + f32x2{b[2], b[3]};
        vec4 o;
        o[0] = (T)s0[0]; o[1] = (T)s0[1]; o[2] = (T)s1[0]; o[3] = (T)s1[1];
        return o;
      };
      auto t_widen_relu = [&](const vec4& o0, const vec4& o1) __attribute__((always_inline)) {
        const u32x2 d0 = __builtin_bit_cast(u32x2, o0), d1 = __builtin_bit_cast(u32x2, o1);
        const auto sx = __builtin_amdgcn_permlane32_swap(d0[0], d1[0], false, false);
        const auto sy = __builtin_amdgcn_permlane32_swap(d0[1], d1[1], false, false);
        const u32x4l w = {sx[0], sy[0], sx[1], sy[1]};
        s16x8 si = __builtin_bit_cast(s16x8, w);
        si = __builtin_elementwise_max(si, s16x8{0, 0, 0, 0, 0, 0, 0, 0});    // ReLU on the rounded values
        return si;
      };
      // the tile of the intermediate map -> registers (the same bits epilogue_lean would have stored); pixels outside the map are
      // the second layer's zero padding.  Arithmetic first, LDS writes after the barrier that ends chunk 3.
      s16x8 mid_v[MPW][2];
      auto mid_compute = [&](const TileCur& c) __attribute__((always_inline)) {
        const int my0 = c.ty * TH + cw * MPW, mx = c.tx * XSTEP + XOFF + lr;
        const bool colok = (unsigned)mx < (unsigned)a.Wout;
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(sBias + 16 * qp + 4 * lh);
          const f32x4 b1 = *reinterpret_cast<const f32x4*>(sBias + 16 * qp + 8 + 4 * lh);
#pragma unroll
          for (int m = 0; m < MPW; ++m) {
            s16x8 wv = t_widen_relu(t_pack4(acc[m][0], 2 * qp, b0), t_pack4(acc[m][0], 2 * qp + 1, b1));
            if (!(colok && my0 + m < a.Hout)) wv = s16x8{0, 0, 0, 0, 0, 0, 0, 0};
            mid_v[m][qp] = wv;
          }
        }
      };
      auto mid_write = [&](const TileCur& c, int par) __attribute__((always_inline)) {
        char* const cnext = sCarry + (par ^ 1) * 4 * CPL;
        const bool last_of_strip = c.ty == a.tiles_y - 1;
#pragma unroll
        for (int m = 0; m < MPW; ++m)
#pragma unroll
          for (int qp = 0; qp < 2; ++qp) {
            *reinterpret_cast<s16x8*>(img + (2 * qp + lh) * XPL + ((2 + cw * MPW + m) * HW + lr) * 16) = mid_v[m][qp];
            if (m >= MPW - 2 && cw == NCW - 1)
              *reinterpret_cast<s16x8*>(cnext + (2 * qp + lh) * CPL + ((m - (MPW - 2)) * HW + lr) * 16) =
                  last_of_strip ? s16x8{0, 0, 0, 0, 0, 0, 0, 0} : mid_v[m][qp];
          }
      };
      // second layer's accumulators -> bias + ReLU + rounding -> 1x1 outconv on the matrix cores (outc_row) -> last activation
      // -> fp32 store
      vec o1w[2];
      auto out_epilogue = [&](const TileCur& c, bool store) __attribute__((always_inline)) {
        const int oy0 = c.ty * TH + cw * MPW, ox = c.tx * XSTEP + lr;
        const bool xin = store && lr < XSTEP && ox < a.oW && lh == 0;
        float* const orow = a.out1 + ((size_t)c.n * a.oH + oy0) * a.oW + ox;
        float tot[MPW];
#pragma unroll
        for (int m = 0; m < MPW; ++m) tot[m] = outc_row(acc[m][0], sB1, o1w) + sO1[32];
#pragma unroll
        for (int m = 0; m < MPW; ++m)
          if (xin && oy0 + m < a.oH) { UNCL_CHK(a.chk, orow + (size_t)m * a.oW, 4); orow[(size_t)m * a.oW] = uncl_act(tot[m], a.out1_act); }
      };
      // fragments of the second layer for tap column (ks, tx): A = its weights (LDS, resident), B = the image -- rows 0 / 1 of the
      // wave's MPW + 2 come from `p01` (wave 0: the carried rows; the others: the rows the wave above wrote), the rest are its own
      const char* const a1off = sW1 + lh * W1PL + lr * 16;
      auto rd1 = [&](const char* p01k0, const char* p01k1, int set, int col) __attribute__((always_inline)) {
        const int ks = col / 3, tx = col - 3 * ks;
        const char* p01 = ks ? p01k1 : p01k0;
        const char* pb = img + boff + 2 * ks * XPL;
#pragma unroll
        for (int ty = 0; ty < 3; ++ty)
          A[set][ty][0] = *reinterpret_cast<const vec*>(a1off + 2 * ks * W1PL + ((ty * 3 + tx) * 32) * 16);
#pragma unroll
        for (int r = 0; r < MPW + 2; ++r)
          B[set][r] = *reinterpret_cast<const vec*>((r < 2 ? p01 : pb) + (r * HW + tx) * 16);
      };
      TileCur tc;
      cur_init<TAIL>(tc, tile0, a);
      int par = 0;
      bool store = !warm;
      // (timing builds: phase 0 = the four chunks' MFMAs, 1 = the two epilogues, 2 = barrier waits, 3 = the second layer's MFMAs;
      // per-kind slots 0 / 1 = wait at the image / second-layer barrier, 2 / 3 = image / result epilogue, 4..7 = chunk barriers)
      PCT_DECL
      pc_barrier();                         // stage 0 is staged (and the carry buffers are cleared, the outconv weights in LDS)
      PCT(2)
      outc_frags(o1w);
      rd(smem, wres, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
      int s = 0;
      for (int t = tile0; t < tile_end; ++t) {
        // chunks 0..2: as in the loop below
        for (int kc = 0; kc < 3; ++kc) {
          mfma_cols_0_4(smem + (s & 1) * STAGE, wres + kc * WBYTES);
          PCT(0)
          __builtin_amdgcn_sched_barrier(0);
          pc_barrier();
          __builtin_amdgcn_sched_barrier(0);
          PCT_K(2, 4 + kc)
          rd(smem + ((s + 1) & 1) * STAGE, wres + (kc + 1) * WBYTES, 0, 0);
          mfma_col(5);
          sched_reads_under_mfmas();
          PCT(0)
          ++s;
        }
        // chunk 3 (stage 1).  Its barrier comes AFTER the tile's conversion arithmetic: the staging waves' heaviest iteration
        // (the next tile's chunk 0 with the up-conv's MFMAs) runs beside this chunk and gets that much longer; once every wave
        // is past it nobody reads stage 1 any more and the image goes there
        mfma_cols_0_4(img, wres + 3 * WBYTES);
        mfma_col(5);
        __builtin_amdgcn_sched_group_barrier(0x008, NMM, 0);
        PCT(0)
        mid_compute(tc);
        zero_acc();
        PCT_K(1, 2)
        __builtin_amdgcn_sched_barrier(0);
        pc_barrier();
        __builtin_amdgcn_sched_barrier(0);
        PCT_K(2, 7)
        ++s;
        mid_write(tc, par);
        __builtin_amdgcn_sched_barrier(0);
        pc_barrier();                       // the image (and the carried rows of the NEXT tile) are written
        __builtin_amdgcn_sched_barrier(0);
        PCT_K(2, 0)
        const char* const ccur = sCarry + par * 4 * CPL;
        const char* const p01k0 = cw == 0 ? ccur + lh * CPL + lr * 16 : img + boff;
        const char* const p01k1 = cw == 0 ? ccur + (2 + lh) * CPL + lr * 16 : img + boff + 2 * XPL;
        rd1(p01k0, p01k1, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
#pragma unroll
        for (int col = 0; col < 5; ++col) {
          rd1(p01k0, p01k1, (col & 1) ^ 1, col + 1);
          mfma_col(col);
          sched_reads_under_mfmas();
        }
        PCT(3)
        __builtin_amdgcn_sched_barrier(0);
        pc_barrier();                       // done with the image: the staging waves may overwrite stage 1
        __builtin_amdgcn_sched_barrier(0);
        PCT_K(2, 1)
        if (t + 1 < tile_end) rd(smem, wres, 0, 0);
        mfma_col(5);
        if (t + 1 < tile_end) sched_reads_under_mfmas();
        PCT(3)
        out_epilogue(tc, store);
        zero_acc();
        PCT_K(1, 3)
        store = true;
        par ^= 1;
        tc.kc = a.nk - 1;
        cur_next<TAIL>(tc, a, tile_end);
      }
      PCT_FLUSH(0)
      return;
    } else {
    PCT_DECL
    if (MODE == 3 || UPT) pc_barrier();     // the staging waves' first image patch / the up-conv's resident weights
    pc_barrier();     // stage 0 is staged
    PCT(2)
    if constexpr (O1C == 1) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) o1c_w[ks] = *reinterpret_cast<const vec*>(sO1F + (ks * 64 + lane) * 16);
    }
    if constexpr (O1C == 2) {
      // =================================================================================================================
      // Two accumulator sets (round 6): tile t multiplies into one set while tile t - 1 leaves from the other -- its rounding, the
      // outconv's eight MFMAs, the activation and the stores are issued in the MFMA stream's shadow (the matrix pipe holds the
      // SIMD's issue port for 8 of an MFMA's 32 cycles; this wave is alone with one staging wave on its SIMD).  With one set the
      // multiplying waves of this layer spent 44 % of their time in that epilogue beside staging waves that idled 73 %
      // (tools/pc_phase_timing.py --layers o1c).  200+ registers: four staging waves (two waves per SIMD).  A tile's first MFMA
      // per accumulator takes an inline-zero C: nothing to clear.  Same MFMA order per accumulator -> the same bits as O1C == 1.
      // =================================================================================================================
      static_assert(NT == 1 && PW == 4, "two accumulator sets: 32-channel tiles, two waves per SIMD");
      PcAcc acc2[MPW][NT];
      auto col_into = [&](auto set_tag, int col) __attribute__((always_inline)) {
        constexpr int SET = decltype(set_tag)::value;
        PcAcc (&ac)[MPW][NT] = *(SET ? &acc2 : &acc);
        const int fs = col & 1;
        if (col == 0) {
#pragma unroll
          for (int m = 0; m < MPW; ++m) ac[m][0] = pc_mm(A[fs][0][0], B[fs][m], zero16, 0);
#pragma unroll
          for (int m = 0; m < MPW; ++m)
#pragma unroll
            for (int ty = 1; ty < 3; ++ty) ac[m][0] = pc_mm(A[fs][ty][0], B[fs][m + ty], ac[m][0], ty);
        } else {
#pragma unroll
          for (int m = 0; m < MPW; ++m)
#pragma unroll
            for (int ty = 0; ty < 3; ++ty) ac[m][0] = pc_mm(A[fs][ty][0], B[fs][m + ty], ac[m][0], ty);
        }
      };
      typedef float f32x2o __attribute__((ext_vector_type(2)));
      typedef short s16x8o __attribute__((ext_vector_type(8)));
      // the leaving tile, two rows at a time, in three pieces: rounding (bias, ReLU on the rounded pair: outc_row's operations), the
      // outconv's MFMAs, sums + activation + stores (epilogue_o1c's).  Row pairs and sched_barriers between the tap columns bound
      // the live ranges: everything at once (the scheduler's choice when left alone) is 266 registers, and a spilled register's
      // reload waits on vmcnt, i.e. on the wave's output stores (0.364 against 0.250 ms with four spill reloads per tile)
      constexpr int HP = MPW / 2;
      static_assert(MPW % 2 == 0, "row pairs");
      auto leave_bf = [&](auto set_tag, auto h_tag, vec (&Bf)[2][2]) __attribute__((always_inline)) {
        constexpr int SET = decltype(set_tag)::value, H = decltype(h_tag)::value;
        PcAcc (&ac)[MPW][NT] = *(SET ? &acc2 : &acc);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const PcAcc& v = ac[2 * H + r][0];
            vec o;
#pragma unroll
            for (int hq = 0; hq < 2; ++hq) {
              const int q = 2 * ks + hq;
              const f32x4 bq = *reinterpret_cast<const f32x4*>(sBias + 8 * q + 4 * lh);
              const f32x2o s0 = f32x2o{v[4 * q], v[4 * q + 1]} + f32x2o{bq[0], bq[1]};
              const f32x2o s1 = f32x2o{v[4 * q + 2], v[4 * q + 3]} + f32x2o{bq[2], bq[3]};
              o[4 * hq] = (T)s0[0]; o[4 * hq + 1] = (T)s0[1]; o[4 * hq + 2] = (T)s1[0]; o[4 * hq + 3] = (T)s1[1];
            }
            s16x8o si = __builtin_bit_cast(s16x8o, o);
            si = __builtin_elementwise_max(si, s16x8o{0, 0, 0, 0, 0, 0, 0, 0});
            Bf[r][ks] = __builtin_bit_cast(vec, si);
          }
      };
      // (the outconv's fragments are read from LDS where they are used: eight registers held for the launch were the ones spilled)
      auto leave_mm = [&](const vec (&Bf)[2][2], f32x16 (&dd)[2]) __attribute__((always_inline)) {
        const vec w0 = *reinterpret_cast<const vec*>(sO1F + lane * 16), w1 = *reinterpret_cast<const vec*>(sO1F + (64 + lane) * 16);
#pragma unroll
        for (int r = 0; r < 2; ++r) dd[r] = mfma32x16(w0, Bf[r][0], zero16);
#pragma unroll
        for (int r = 0; r < 2; ++r) dd[r] = mfma32x16(w1, Bf[r][1], dd[r]);
      };
      // (this instantiation IS the sigmoid's -- the launcher takes it only then: a branch inside the tile's body ends the scheduling
      // region the leaving tile is interleaved in, and the matrix pipe then idles for the length of the activation)
      auto leave_fin = [&](const TileCur& c, auto h_tag, const f32x16 (&dd)[2]) __attribute__((always_inline)) {
        constexpr int H = decltype(h_tag)::value;
        constexpr unsigned BAD = 0x40000000u;
        const int y0 = c.ty * TH + cw * MPW + 2 * H, ox = c.tx * TW + lr;
        const unsigned sample = (unsigned)(a.Hout * a.Wout) * 4u;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.out1) + (size_t)c.n * sample, (short)0,
                                                                            (int)sample, 0x00020000);
        const unsigned loff = ((unsigned)ox * 4u) | ((lh == 0 && ox < a.Wout) ? 0u : BAD);
        const unsigned rowb = (unsigned)a.Wout * 4u;
        const float ob = sO1[32];
        float tot[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) tot[r] = uncl_sigmoid(((dd[r][2] + dd[r][1]) + dd[r][0]) + ob);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int oy = y0 + r;
          const unsigned off = loff + (((unsigned)oy * rowb) | (oy < a.Hout ? 0u : BAD));
#ifdef UNCL_CHECKED
          if (off < BAD) UNCL_CHK(a.chk, reinterpret_cast<const char*>(a.out1) + (size_t)c.n * sample + off, 4);
#endif
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, tot[r]), rs, off, 0, 0);
        }
      };
      // per tap column: its MFMAs, the next column's fragment reads (one per MFMA gap), and `valu` vector instructions of the leaving
      // tile per gap
      auto sched_col = [&](auto mm_tag, auto reads_tag, auto valu_tag) __attribute__((always_inline)) {
        constexpr int MM = decltype(mm_tag)::value, READS = decltype(reads_tag)::value, VALU = decltype(valu_tag)::value;
#pragma unroll
        for (int k = 0; k < MM; ++k) {
          if (k < READS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (VALU > 0) __builtin_amdgcn_sched_group_barrier(0x002, VALU, 0);
        }
      };
      int s = 0;
      rd(smem, wres, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
      // the tile entering set SET; `leaving`: the tile in the other set (cursor lc) leaves beside it
      auto tile_step = [&](auto set_tag, auto leaving_tag, const TileCur& lc, bool more) __attribute__((always_inline)) {
        constexpr int SET = decltype(set_tag)::value;
        constexpr bool LEAVING = decltype(leaving_tag)::value != 0;
        static_assert(MPW == 4, "two row pairs");
        const char* const st = smem + (s & 1) * STAGE;
        vec Bf0[2][2], Bf1[2][2];
        f32x16 dd0[2], dd1[2];
        // column 0: rows 0 / 1 of the leaving tile are rounded
        rd(st, wres, 1, 1); col_into(set_tag, 0);
        if (LEAVING) leave_bf(IntTag<SET ^ 1>{}, IntTag<0>{}, Bf0);
        sched_col(IntTag<NMM>{}, IntTag<NRD>{}, IntTag<(LEAVING ? 4 : 0)>{});
        __builtin_amdgcn_sched_barrier(0);
        // column 1: their outconv MFMAs; rows 2 / 3 are rounded (the leaving set is dead from here on)
        rd(st, wres, 0, 2); col_into(set_tag, 1);
        if (LEAVING) { leave_mm(Bf0, dd0); leave_bf(IntTag<SET ^ 1>{}, IntTag<1>{}, Bf1); }
        sched_col(IntTag<NMM + (LEAVING ? 4 : 0)>{}, IntTag<NRD>{}, IntTag<(LEAVING ? 4 : 0)>{});
        __builtin_amdgcn_sched_barrier(0);
        // column 2: the outconv MFMAs of rows 2 / 3
        rd(st, wres, 1, 3); col_into(set_tag, 2);
        if (LEAVING) leave_mm(Bf1, dd1);
        sched_col(IntTag<NMM + (LEAVING ? 4 : 0)>{}, IntTag<NRD>{}, IntTag<0>{});
        __builtin_amdgcn_sched_barrier(0);
        // columns 3, 4: sums, activation, stores
        rd(st, wres, 0, 4); col_into(set_tag, 3);
        if (LEAVING) leave_fin(lc, IntTag<0>{}, dd0);
        sched_col(IntTag<NMM>{}, IntTag<NRD>{}, IntTag<(LEAVING ? 4 : 0)>{});
        __builtin_amdgcn_sched_barrier(0);
        rd(st, wres, 1, 5); col_into(set_tag, 4);
        if (LEAVING) leave_fin(lc, IntTag<1>{}, dd1);
        sched_col(IntTag<NMM>{}, IntTag<NRD>{}, IntTag<(LEAVING ? 4 : 0)>{});
        __builtin_amdgcn_sched_barrier(0);
        col_into(set_tag, 5);
        __builtin_amdgcn_sched_group_barrier(0x008, NMM, 0);
        __builtin_amdgcn_sched_barrier(0);
        pc_barrier();
        __builtin_amdgcn_sched_barrier(0);
        ++s;
        if (more) {
          rd(smem + (s & 1) * STAGE, wres, 0, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
        }
      };
      TileCur lc = cc;                     // the leaving tile's coordinates
      int t = tile0;
      tile_step(IntTag<0>{}, IntTag<0>{}, lc, t + 1 < tile_end);
      ++t;
      int last = 0;                        // the set of the last tile multiplied
      for (;;) {
        if (t >= tile_end) { last = 0; break; }
        lc = cc; cc.kc = a.nk - 1; cur_next<TAIL>(cc, a, tile_end);
        tile_step(IntTag<1>{}, IntTag<1>{}, lc, t + 1 < tile_end);
        ++t;
        if (t >= tile_end) { last = 1; break; }
        lc = cc; cc.kc = a.nk - 1; cur_next<TAIL>(cc, a, tile_end);
        tile_step(IntTag<0>{}, IntTag<1>{}, lc, t + 1 < tile_end);
        ++t;
      }
      {
        vec Bf0[2][2], Bf1[2][2];
        f32x16 dd0[2], dd1[2];
        if (last == 0) { leave_bf(IntTag<0>{}, IntTag<0>{}, Bf0); leave_bf(IntTag<0>{}, IntTag<1>{}, Bf1); }
        else { leave_bf(IntTag<1>{}, IntTag<0>{}, Bf0); leave_bf(IntTag<1>{}, IntTag<1>{}, Bf1); }
        leave_mm(Bf0, dd0);
        leave_mm(Bf1, dd1);
        leave_fin(cc, IntTag<0>{}, dd0);
        leave_fin(cc, IntTag<1>{}, dd1);
      }
      return;
    }
    // The barrier of a chunk sits between its LAST fragment read and the MFMAs of its last tap column: "done with the stage" is
    // true as soon as column 5's fragments have landed, and the twelve-plus MFMAs still to go then cover the LDS latency of the
    // next chunk's first fragments (a wave alone on its SIMD has nobody else to hide it: the MFMA phase ran at 80 - 84 % of its
    // issue rate with the bubble at every chunk start).  The staging waves get the stage back that much earlier, too.
    rd(smem, RESW ? wres : smem + XBYTES, 0, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
    // Chunk kinds are positions in straight-line code (inner loop: every chunk but a tile's last; then the last one), not
    // branches inside one loop body: with branches between the barrier and the last column the register allocator spills the
    // in-flight fragment set across them.
    int s = 0;                            // chunks multiplied so far: chunk s sits in stage s & 1
    auto stage_x = [&](int k) __attribute__((always_inline)) { return smem + (k & 1) * STAGE; };
    auto stage_w = [&](int k, int kc) __attribute__((always_inline)) { return RESW ? wres + kc * WBYTES : smem + (k & 1) * STAGE + XBYTES; };
    for (int t = tile0; t < tile_end; ++t) {
      for (int kc = 0; kc < a.nk - 1; ++kc) {
        mfma_cols_0_4(stage_x(s), stage_w(s, kc));
        PCT_K(0, kc & 3)
        __builtin_amdgcn_sched_barrier(0);
        pc_barrier();   // (waits for lgkmcnt(0): column 5's fragments) done with stage s & 1; stage (s + 1) & 1 is staged
        __builtin_amdgcn_sched_barrier(0);
        PCT_K(2, 4 + (kc & 3))
        rd(stage_x(s + 1), stage_w(s + 1, kc + 1), 0, 0);
        mfma_col(5);
        sched_reads_under_mfmas();
        PCT_K(0, kc & 3)
        ++s;
      }
      // a tile's last chunk keeps its barrier BEHIND the epilogue: released earlier, the staging waves' next loads would queue
      // beside the epilogue's stores (same-box A/B: the one- and two-chunk layers 10 - 38 % slower), and the next tile's first
      // fragments held across the epilogue would push every epilogue form over the register budget
      mfma_cols_0_4(stage_x(s), stage_w(s, a.nk - 1));
      mfma_col(5);
      __builtin_amdgcn_sched_group_barrier(0x008, NMM, 0);
      PCT_K(0, (a.nk - 1) & 3)
      run_epilogue(cc, tpar);
      PCT_K(1, (a.nk - 1) & 3)
      pc_barrier();
      PCT_K(2, 4 + ((a.nk - 1) & 3))
      ++s;
      if (t + 1 < tile_end) {
        rd(stage_x(s), stage_w(s, 0), 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
      }
      tpar = (tpar + 1) & 3;
      cc.kc = a.nk - 1;
      cur_next<TAIL>(cc, a, tile_end);          // the next tile's coordinates (unused past the end of the range)
    }
    PCT_FLUSH(0)
    return;
    }
  }

  // ===================================================================================================================
  // producers
  // ===================================================================================================================
  if ((a.pc_prio & 3) == 2) __builtin_amdgcn_s_setprio(2);
  PCT_DECL          // (timing builds: declared before the staging lambdas, which stamp inside the fused first layer's rebuild)
  const int ptid = tid - NCW * 64;
  const int pwave = wave - NCW;
  // per-thread constants of the staging pattern (identical for every step): regular slot j = halo pixel (hy0 + RSTEP*j, hx),
  // vector ch; extra slot = halo pixel (ey, 32 + ec), vector ch
  constexpr int NPROD = PW * 64;
  constexpr int RSTEP = NPROD / 128;                 // halo rows covered per regular pass (2 or 4)
  constexpr int RSN = (HH + RSTEP - 1) / RSTEP;      // regular passes
  constexpr bool R_RAGGED = HH % RSTEP != 0;
  constexpr int XV = RSN + 1;
  constexpr int WPP = NPROD / 4;                     // weight rows per pass
  static_assert(WPP % CT == 0, "a weight pass covers whole taps");
  constexpr int WVN = (WROWS + WPP - 1) / WPP;
  constexpr bool W_RAGGED = WROWS % WPP != 0;
  static_assert(NPROD >= HH * 8, "extra-column slot");
  const int ch = ptid & 3, p0 = ptid >> 2;
  const int hx = p0 & 31, hy0 = p0 >> 5;
  const int ey = ptid >> 3, ec = (ptid >> 2) & 1;
  const bool e_on = ptid < HH * 8;
  const bool r_last_on = !R_RAGGED || hy0 + RSTEP * (RSN - 1) < HH;   // the ragged last regular pass
  const int row_el = a.s0W * a.s0C;
  const int xoff_r = (hy0 * a.s0W + hx) * a.s0C + ch * 8;
  const int xoff_e = (ey * a.s0W + 32 + ec) * a.s0C + ch * 8;
  const int pix_r0 = hy0 * HW + hx;
  const int pix_e = ey * HW + 32 + ec;
  const int lds_w0 = ch * WPL + p0 * 16;             // + XBYTES (streamed) / kc * WBYTES (resident)
  const int woff0 = ((p0 / CT) * a.Cout + (p0 % CT)) * 32 + ch * 8;      // K-chunk-major weights, [Cin / 32][9][Cout][32] (common.h)

  // Staging registers: ONE set.  At step s (the consumers multiply chunk s) chunk s + 1 goes from registers to LDS and the
  // loads of chunk s + 2 are issued into the registers that just became free, so every load has a whole step of the workgroup
  // to land and a staging step waits only for loads issued a step earlier.  (A second set with a third chunk in flight was
  // slower: the vector-memory queue of the CU fills, and both the staging waves' next loads and the multiplying waves' output
  // stores then stall at issue.)
  //   concat source:  steps of a 32-channel slice are [x1, sqrt(x2), x2^2, x2] (phase = step & 3 -> ssr_member, nk is a multiple of 4):
  //                   xa holds the x1 chunk (MODE 4: the up-conv's source fragments), xb the x2 slice, which is staged three
  //                   times (as is, squared, square-rooted) and so read from memory once
  constexpr bool CAT = MODE == 1 || MODE == 4 || MODE == 5;
  vec xa[XV], xb[XV];       // xb is dead (and costs no registers) for plain sources
  vec wv[WVN];
  f32x4 br = {0.f, 0.f, 0.f, 0.f};
  unsigned xvalid_a = 0, xvalid_b = 0;
  int bpar = 0;
  bool bp = false;
  int u_iy0 = 0, u_ix0 = 0;
  int ppar = 0;                 // bias slot (tile count & 3) of the tile the producers are loading
  int p3par = 0;                // MODE 3: the patch buffer the next build reads

  // MODE 4: this wave is tap (dy, dx) of the 2x2 stride-2 transposed conv (and, with eight staging waves, one half of the
  // M-tiles).  Its weight fragment (A rows in the order cout(r) = 16*bit2(r) + 4*(r >> 3) + (r & 3), which makes a lane's D
  // registers 16 consecutive output channels of its pixel) and the bias it starts the accumulation from stay in registers
  // for the whole launch.
  constexpr int UPH = HH / 2, UPW = HW / 2, UPN = UPH * UPW, MTU = (UPN + 31) / 32;
  constexpr int MT_STEP = PW / 4;                       // M-tiles are dealt round-robin over the waves of a tap
  constexpr int MT_PER = (MTU + MT_STEP - 1) / MT_STEP;
  static_assert(MODE != 4 || 2 * MT_PER <= XV, "the up-conv's source fragments fit in the staging registers");
  const int tap = pwave & 3, mt0 = pwave >> 2;
  vec ua[2];
  f32x16 cb;
  // MODE 5: the source fragments of the wave's M-tiles (K = 64: four per M-tile); the weight fragments and the bias of a slice
  // are read from LDS when the slice is built (resident there for the launch: as registers they pushed the eight-staging-wave
  // form, 168 registers, into 49 spilled ones)
  // (MODE 5 deals whole M-TILES to the waves, all four taps each: the source fragments of an M-tile are then requested once
  // instead of by each of the four tap waves -- 32 lines per request, the most expensive loads of the kernel)
  // UPT: that scheme.  MODE 5 always; MODE 4 (32 channels: K = 32, 8 KB of weights -- what the resident-weight form of the
  // dominant launch has left of its 160 KB) with eight staging waves outside the fused last stage
  static_assert(!UPT || MTU <= PW, "one source M-tile per staging wave");
  vec ubf[4];
  int u_sl = 0;
  if (MODE == 4 && !UPT) {
    const int arow = tap * 32 + (((lr >> 2) & 1) << 4) + ((lr >> 3) << 2) + (lr & 3);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) ua[ks] = LD16V(vec, a.up_w + arow * 32 + (2 * ks + lh) * 8);
#pragma unroll
    for (int i = 0; i < 16; ++i) { if (a.up_b) UNCL_CHK(a.chk, a.up_b + 16 * lh + i, 4); cb[i] = a.up_b ? a.up_b[16 * lh + i] : 0.f; }
  }

  // MODE 3: image-patch values in flight, the first layer's weight fragment (cout = lr, k = 8 lh + j -> tap) and the biases
  // of this lane's channels 8q + 4 lh + r
  constexpr int IRN = (PN3 + NPROD - 1) / NPROD;
  float ir[IRN];
  vec preA;
  float preB[16];
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

  // MODE 3: an fp32 image value as its 16-bit head and tail (x = head + tail to 2^-17 |x|), packed head << 16 | tail
  auto pack_head_tail = [&](float v) __attribute__((always_inline)) {
    const T hi = (T)v;
    const T lo = (T)(v - (float)hi);
    return ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16) | (unsigned)__builtin_bit_cast(unsigned short, lo);
  };
  // position of K-chunk kc in the weight's K layout: concat layers store [x2 | x1 | x2^2 | sqrt] and are walked slice by slice
  auto weight_chunk = [&](int kc) __attribute__((always_inline)) {
    if (!CAT) return kc;
    return ssr_member(kc & 3) * (a.s0C >> 5) + (kc >> 2);
  };
  auto load_weights = [&](int cout0, int kc) __attribute__((always_inline)) {
    const bf16_t* wb_ = a.weight + ((size_t)weight_chunk(kc) * 9 * a.Cout + cout0) * 32;
    const int wstride = (WPP / CT) * a.Cout * 32;     // taps per pass x one tap
#pragma unroll
    for (int j = 0; j < WVN; ++j) {
      unsigned off = (unsigned)woff0;
      if (W_RAGGED && j == WVN - 1) off = (p0 + WPP * j < WROWS) ? off : 0u;
      wv[j] = LD16OV(vec, wb_ + j * wstride, off * 2u);
    }
  };
  auto write_weights = [&](char* wst) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < WVN; ++j) {
      if (W_RAGGED && j == WVN - 1 && p0 + WPP * j >= WROWS) continue;
      *reinterpret_cast<vec*>(wst + lds_w0 + j * WPP * 16) = wv[j];
    }
  };

  // one 32-channel chunk of a same-extent source: halo tile at (iy0, ix0) of sample n, channels cbase..; returns the slot
  // validity mask.  Offsets first (branchy, VALU only), then the loads in straight-line code: loads inside the arms of a
  // uniform branch make the compiler's waitcnt insertion assume the other arm's loads may be in flight into these registers.
  // (registers J0 <= j < J1 only: concat sources request their x1 registers in two halves; bits outside the range come back 1)
  auto load_x = [&](vec (&xr)[XV], const bf16_t* xsrc, int n, int iy0, int ix0, int cbase, auto j0_tag, auto j1_tag) __attribute__((always_inline)) {
    constexpr int J0 = decltype(j0_tag)::value, J1 = decltype(j1_tag)::value;
    unsigned valid = 0xffffffffu;
    const bf16_t* base = xsrc + (size_t)n * a.s0H * a.s0W * a.s0C + cbase;
    unsigned off[XV];
    const int toff = (iy0 * a.s0W + ix0) * a.s0C;  // may be negative on the border; masked below
    const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + HH <= a.H && ix0 + HW <= a.W;  // wave-uniform
    if (interior) {
#pragma unroll
      for (int j = 0; j < RSN; ++j)
        off[j] = (unsigned)((j < RSN - 1 || r_last_on) ? toff + xoff_r + j * RSTEP * row_el : 0) * 2u;
      off[RSN] = (unsigned)(e_on ? toff + xoff_e : 0) * 2u;
    } else {
      valid = ~(((1u << J1) - 1u) & ~((1u << J0) - 1u));
      const bool xok = (unsigned)(ix0 + hx) < (unsigned)a.W;
#pragma unroll
      for (int j = J0; j < J1; ++j) {
        bool ok;
        int eoff;
        if (j < RSN) {
          ok = xok && (unsigned)(iy0 + hy0 + RSTEP * j) < (unsigned)a.H && (j < RSN - 1 || r_last_on);
          eoff = xoff_r + j * RSTEP * row_el;
        } else {
          ok = e_on && (unsigned)(iy0 + ey) < (unsigned)a.H && (unsigned)(ix0 + 32 + ec) < (unsigned)a.W;
          eoff = xoff_e;
        }
        valid |= (ok ? 1u : 0u) << j;
        off[j] = (ok ? (unsigned)(toff + eoff) : 0u) * 2u;
      }
    }
#pragma unroll
    for (int j = J0; j < J1; ++j) xr[j] = LD16OV(vec, base, off[j]);
    return valid;
  };

  // loads of the chunk the cursor points at; P = chunk index & 3 (compile time): for the concat sources which member of
  // [x1, sqrt, x2^2, x2] (ssr_member) the chunk is.
  // Concat sources spread the requests of a slice over the iterations: the x2 registers (xb) are requested
  // with the slice's first chunk (P = 0: the iteration that has just staged the last of the previous slice's three chunks from them), the x1
  // registers (xa) earlier, one half each from the two iterations before that one (`xa_tag` = 1 / 2, cursor two chunks / one
  // chunk ahead).  Both in one iteration were one 12-load burst per thread every fourth iteration -- 98 KB per CU at once: the
  // waves stall at issue until the CU's vector-memory queue drains, and that iteration (which also has the slice's only
  // transcendental transform) took twice a multiplying step; the other three issued no loads at all.
  auto load_step = [&](const TileCur& c, auto p_tag, auto xa_tag) __attribute__((always_inline)) {
    constexpr int P = decltype(p_tag)::value;
    constexpr int XA_PART = decltype(xa_tag)::value;                 // concat: only the x1 registers of the slice starting at c:
    constexpr bool XA_ONLY = XA_PART != 0;                           // 1 = their first half, 2 = the second, 3 = all of them
    constexpr int XH = (XV + 1) / 2;
    constexpr int XJ0 = XA_PART == 2 ? XH : 0, XJ1 = XA_PART == 1 ? XH : XV;   // register range of this call
    static_assert(!XA_ONLY || (CAT && P == 0), "x1 requests are slice starts of concat sources");
    constexpr bool SET_A = !CAT || P == 0;     // X registers this chunk loads into (if it loads any)
    constexpr bool X_LOAD = !CAT || XA_ONLY || (P == 0 && UNCL_PC_XA_SLOT == 0);   // this call requests the registers `xr`
    vec (&xr)[XV] = SET_A ? xa : xb;
    const int n = c.n, y0 = c.ty * TH, x0 = c.tx * XSTEP + XOFF, cout0 = c.ct * CT, kc = c.kc;
    int g = 0, cbase = kc * 32;
    if (CAT) {
      cbase = (kc >> 2) * 32;
      g = ssr_member(P);
    }
    if (!XA_ONLY) {
      bp = kc == 0;
      bpar = ppar;
    }
    const int iy0 = y0 - a.pad, ix0 = x0 - a.pad;
    const bool same_ext = a.s1H == a.s0H && a.s1W == a.s0W;
    const bf16_t* xsrc = (MODE != 0 && g == 1) ? a.src1 : a.src0;
    unsigned valid = 0xffffffffu;
    if (MODE == 3) {
      // the image patch under the halo tile: first-layer pixel (y, x) reads image rows y..y+2, columns x..x+2.  Rows and
      // columns past the image only feed outputs that are never stored (valid convolution), so they are clamped.
      if (a.img_off) UNCL_CHK(a.chk, a.img_off + n, 4);
      const float* ib = a.img + (a.img_off ? (size_t)a.img_off[n] : (size_t)n * a.imgH * a.imgW);
#pragma unroll
      for (int k = 0; k < IRN; ++k) {
        const int idx = min(ptid + k * NPROD, PN3 - 1);
        const int pr = idx / PW3, pcl = idx - pr * PW3;
        UNCL_CHK(a.chk, ib + (size_t)min(iy0 + pr, a.imgH - 1) * a.imgP + min(ix0 + pcl, a.imgW - 1), 4);
        ir[k] = ib[(size_t)min(iy0 + pr, a.imgH - 1) * a.imgP + min(ix0 + pcl, a.imgW - 1)];
      }
    } else if (!X_LOAD) {
      // x2^2 / sqrt(x2): staged from the x2 registers
    } else if (MODE == 4 && !UPT && P == 0) {
      // B fragments of the up-conv straight from global memory: source pixel sp = 32 mt + lr of the 9 x 17 patch under the
      // 18 x 34 halo tile, channels 8 (2 ks + lh) .. +7; out-of-image source pixels are clamped (their outputs are outside the
      // image too and are staged as zeros)
      const int sy0 = iy0 >> 1, sx0 = ix0 >> 1;
      u_iy0 = iy0; u_ix0 = ix0;
      const bf16_t* ub = a.src1 + (size_t)n * a.s1H * a.s1W * 32;
      constexpr int IH = (MT_PER + 1) / 2;
#pragma unroll
      for (int i = (XA_PART == 2 ? IH : 0); i < (XA_PART == 1 ? IH : MT_PER); ++i) {
        const int mt = mt0 + MT_STEP * i;
        const int spc = min(mt * 32 + lr, UPN - 1);
        const int spy = (spc * 241) >> 12, spx = spc - spy * UPW;     // / 17 for spc < 1024
        const int yy = min(max(sy0 + spy, 0), a.s1H - 1), xx = min(max(sx0 + spx, 0), a.s1W - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          xr[2 * i + ks] = LD16OV(vec, ub, (unsigned)(((yy * a.s1W + xx) * 32 + (2 * ks + lh) * 8) * 2));
      }
    } else if (UPT && P == 0) {
      // the source fragments of this wave's M-tile (UKS K-steps per source pixel); weights and bias are resident in LDS
      const int sy0 = iy0 >> 1, sx0 = ix0 >> 1;
      u_iy0 = iy0; u_ix0 = ix0;
      u_sl = kc >> 2;
      const bf16_t* ub = a.src1 + (size_t)n * a.s1H * a.s1W * UPC;
      {
        const int spc = min(pwave * 32 + lr, UPN - 1);          // (waves past the last M-tile request its last pixels: unused)
        const int spy = (spc * 241) >> 12, spx = spc - spy * UPW;     // / 17 for spc < 1024
        const int yy = min(max(sy0 + spy, 0), a.s1H - 1), xx = min(max(sx0 + spx, 0), a.s1W - 1);
#pragma unroll
        for (int ks = 0; ks < UKS; ++ks)
          ubf[ks] = LD16OV(vec, ub, (unsigned)(((yy * a.s1W + xx) * UPC + (2 * ks + lh) * 8) * 2));
      }
    } else if (MODE == 1 && P == 0 && !same_ext) {
      // upsampled map, replicate-padded to the skip's extent (unet_parts.py:292-298)
      const bf16_t* base = a.src1 + (size_t)n * a.s1H * a.s1W * a.s1C + cbase + ch * 8;
      unsigned off[XV];
      const int dy = (a.s0H - a.s1H) >> 1, dx = (a.s0W - a.s1W) >> 1;
      valid = ~(((1u << XJ1) - 1u) & ~((1u << XJ0) - 1u));
      const int ix = ix0 + hx;
      const bool xok = (unsigned)ix < (unsigned)a.W;
      const int sx = min(max(ix - dx, 0), a.s1W - 1);
#pragma unroll
      for (int j = XJ0; j < (XJ1 < RSN ? XJ1 : RSN); ++j) {
        const int iy = iy0 + hy0 + RSTEP * j;
        const bool ok = xok && (unsigned)iy < (unsigned)a.H && (j < RSN - 1 || r_last_on);
        const int sy = min(max(iy - dy, 0), a.s1H - 1);
        valid |= (ok ? 1u : 0u) << j;
        off[j] = (unsigned)((sy * a.s1W + sx) * a.s1C) * 2u;
      }
      if (XJ1 == XV) {
        const int iy = iy0 + ey, ixe = ix0 + 32 + ec;
        const bool ok = e_on && (unsigned)iy < (unsigned)a.H && (unsigned)ixe < (unsigned)a.W;
        const int sy = min(max(iy - dy, 0), a.s1H - 1), sxe = min(max(ixe - dx, 0), a.s1W - 1);
        valid |= (ok ? 1u : 0u) << RSN;
        off[RSN] = (unsigned)((sy * a.s1W + sxe) * a.s1C) * 2u;
      }
#pragma unroll
      for (int j = XJ0; j < XJ1; ++j) xr[j] = LD16OV(vec, base, off[j]);
    } else {
      valid = load_x(xr, xsrc, n, iy0, ix0, cbase, IntTag<XJ0>{}, IntTag<XJ1>{});
    }
    if (CAT && P == 0 && !XA_ONLY) {
      // the x2 slice of this group (its own step would be the one in which the multiplying waves store the previous tile: with
      // no loads queued at that time their stores do not wait behind ours)
      xvalid_b = load_x(xb, a.src0, n, iy0, ix0, cbase, IntTag<0>{}, IntTag<XV>{});
    }
    if (X_LOAD) xvalid_a = XA_PART == 2 ? (xvalid_a & valid) : valid;     // (a half's mask has ones outside its range)
    if (XA_ONLY) return;
    if (!RESW) {
      load_weights(cout0, kc);
      if (bp && ptid < CT / 4 && a.bias != nullptr) br = LD16O_F32(a.bias + cout0, (unsigned)ptid * 16u);
    }
  };

  // TAIL: the up-conv of the NEXT tile's x1 chunk, arithmetic only (MFMAs, rounding, zero padding) -- run at the end of the
  // iteration BEFORE the one that stages it: that iteration (the tile's last chunk) is the one the multiplying waves waited
  // for, and the stage it writes to is not free yet, but registers are
  vec upv[MT_PER][2];
  auto up_compute = [&](auto i0_tag, auto i1_tag) __attribute__((always_inline)) {
    constexpr int I0 = decltype(i0_tag)::value, I1 = decltype(i1_tag)::value;
    const int iy0h = u_iy0 + (tap >> 1), ix0h = u_ix0 + (tap & 1);
#pragma unroll
    for (int i = I0; i < I1; ++i) {
      const int mt = mt0 + MT_STEP * i;
      if (mt < MTU) {       // wave-uniform
        const int sp = mt * 32 + lr, spc = min(sp, UPN - 1);
        const int spy = (spc * 241) >> 12, spx = spc - spy * UPW;
        f32x16 cu = cb;
        cu = mfma32x16(ua[0], xa[2 * i], cu);
        cu = mfma32x16(ua[1], xa[2 * i + 1], cu);
        const bool in_img = (unsigned)(iy0h + 2 * spy) < (unsigned)a.H && (unsigned)(ix0h + 2 * spx) < (unsigned)a.W;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float f[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = cu[8 * h + e];
          vec o = E::pack(f);
          if (!in_img) o = E::zero();
          upv[i][h] = o;
        }
      }
    }
  };
  auto write_step = [&](char* st, auto p_tag) __attribute__((always_inline)) {
    constexpr int P = decltype(p_tag)::value;
    constexpr bool SET_A = !CAT || P == 0;
    vec (&xr)[XV] = SET_A ? xa : xb;
    const unsigned xvalid = SET_A ? xvalid_a : xvalid_b;
    const bool all_ok = xvalid == 0xffffffffu;
    if (MODE == 3) {
      // Build the 32-channel halo tile of THIS step's tile from the image patch staged one step ago (sP[par], published by
      // the last barrier) with the matrix cores: per 32 halo pixels two MFMAs over the nine taps of the bf16 head and of the
      // bf16 tail of the fp32 input (x = hi + lo to 2^-17); a lane's D quad q is channels 8q + 4 lh.. of its pixel, i.e. one
      // half of K-slot plane q.  Then park the patch of the NEXT tile (in registers since the last step) in the other buffer.
      // (the patch is parked as packed (head << 16 | tail) 16-bit pairs: the split of an fp32 value costs ~5 vector instructions
      // and was done per tap and per halo pixel -- 9 x 1.3 times per image pixel -- by the waves this layer is bound by)
      const unsigned* sPt = reinterpret_cast<const unsigned*>(sP) + p3par * PN3;
      f32x16 preBv;
#pragma unroll
      for (int i = 0; i < 16; ++i) preBv[i] = preB[i];
      constexpr int MT3 = (NPIX + 31) / 32, IT3 = (MT3 + PW - 1) / PW;
      static_assert(HW == 34 && NPIX < 2048, "the reciprocal multiply below divides by 34");
      // Round 6: the wave's M-tiles in PHASES -- every tile's taps gathered, then every tile's MFMAs, then every tile's rounding and
      // LDS writes -- as straight-line code.  One tile after the other behind a wave-uniform `if (mt < MT3)` each was two dependent
      // chains (LDS gather -> two MFMAs -> rounding -> LDS write: 985 cycles per tile in the stamped build, on the waves the whole
      // launch waits for) that the compiler could not interleave across the branch.  A wave past the last M-tile (one of the eight,
      // at 15 M-tiles) computes a clamped copy of the last one and stores nothing.
      typedef unsigned u32x4p __attribute__((ext_vector_type(4)));
      vec Bh[IT3], Bl[IT3];
#pragma unroll
      for (int i = 0; i < IT3; ++i) {
        const int pcl = min((pwave + PW * i) * 32 + lr, NPIX - 1);
        const int py = (pcl * 241) >> 13, px = pcl - py * HW;   // / 34 for pcl < 2048 (HW == 34)
        const unsigned* pp = sPt + py * PW3 + px;
        unsigned tp[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int o0 = (j / 3) * PW3 + j % 3;  // tap j (lower half-wave)
          unsigned u = pp[lh ? 2 * PW3 + 2 : o0];   // upper half-wave: tap 8 in slot 0
          if (j > 0) u = lh ? 0u : u;
          tp[j] = u;
        }
        // heads of taps (2k, 2k + 1) -> register k of the head fragment, tails -> the tail fragment: one v_perm_b32 each
        u32x4p bh, bl;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          bh[k] = __builtin_amdgcn_perm(tp[2 * k + 1], tp[2 * k], 0x07060302u);
          bl[k] = __builtin_amdgcn_perm(tp[2 * k + 1], tp[2 * k], 0x05040100u);
        }
        Bh[i] = __builtin_bit_cast(vec, bh);
        Bl[i] = __builtin_bit_cast(vec, bl);
      }
      // the bias is the first MFMA's C operand (sixteen adds per lane and 32 pixels otherwise: the staging waves of this layer issue
      // ~8 vector instructions per MFMA of the whole kernel)
      f32x16 c3[IT3];
#pragma unroll
      for (int i = 0; i < IT3; ++i) c3[i] = mfma32x16(preA, Bh[i], preBv);
#pragma unroll
      for (int i = 0; i < IT3; ++i) c3[i] = mfma32x16(preA, Bl[i], c3[i]);
      if (PC_ABL(4)) {
#pragma unroll
        for (int i = 0; i < IT3; ++i) asm volatile("" ::"v"(c3[i][0]), "v"(c3[i][15]));     // (a 64-byte operand breaks the HOST pass: every stub of the kernel vanishes)
      } else if (a.slope == 0.f) {
        // ReLU on the rounded pair (rounding is sign-symmetric), like the multiplying waves' epilogue.  The two half-waves
        // hold the two halves of a pixel's 16-byte K-slot: v_permlane32_swap turns quads (2 qp, 2 qp + 1) into one whole slot
        // per lane (lower half-wave: plane 2 qp, upper: 2 qp + 1), i.e. one ds_write_b128 whose eight-lane groups cover the 32
        // banks exactly -- the 8-byte half-slot stores were two-way bank conflicts (lanes lr and lr + 8)
        typedef unsigned u32x2w __attribute__((ext_vector_type(2)));
        typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
        typedef short s16x8w __attribute__((ext_vector_type(8)));
#pragma unroll
        for (int i = 0; i < IT3; ++i) {
          const int mt = pwave + PW * i, pix = mt * 32 + lr;
#pragma unroll
          for (int qp = 0; qp < 2; ++qp) {
            vec4 o0, o1;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o0[e] = (T)c3[i][8 * qp + e]; o1[e] = (T)c3[i][8 * qp + 4 + e]; }
            const u32x2w d0 = __builtin_bit_cast(u32x2w, o0), d1 = __builtin_bit_cast(u32x2w, o1);
            const auto sx = __builtin_amdgcn_permlane32_swap(d0[0], d1[0], false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(d0[1], d1[1], false, false);
            const u32x4w w = {sx[0], sy[0], sx[1], sy[1]};
            s16x8w si = __builtin_bit_cast(s16x8w, w);
            si = __builtin_elementwise_max(si, s16x8w{0, 0, 0, 0, 0, 0, 0, 0});
            if (mt < MT3 && pix < NPIX) *reinterpret_cast<s16x8w*>(st + (2 * qp + lh) * XPL + pix * 16) = si;
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < IT3; ++i) {
          const int mt = pwave + PW * i, pix = mt * 32 + lr;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            vec4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float t = c3[i][4 * q + e];
              o[e] = (T)(fmaxf(t, 0.f) + a.slope * fminf(t, 0.f));
            }
            if (mt < MT3 && pix < NPIX) *reinterpret_cast<vec4*>(st + q * XPL + pix * 16 + (lh << 3)) = o;
          }
        }
      }
      PCT_K(0, 1)       // (fused first layer, timing builds: slot 1 = the rebuild, slot 2 = parking the next patch)
#pragma unroll
      for (int k = 0; k < IRN; ++k)
        if (ptid + k * NPROD < PN3) reinterpret_cast<unsigned*>(sP)[(p3par ^ 1) * PN3 + ptid + k * NPROD] = pack_head_tail(ir[k]);
      p3par ^= 1;
    } else if (MODE == 4 && P == 0 && TAIL) {
      // (computed by up_compute() one iteration earlier, while the multiplying waves were still reading this stage)
#pragma unroll
      for (int i = 0; i < MT_PER; ++i) {
        const int mt = mt0 + MT_STEP * i;
        if (mt < MTU) {
          const int sp = mt * 32 + lr, spc = min(sp, UPN - 1);
          const int spy = (spc * 241) >> 12, spx = spc - spy * UPW;
          char* dst = st + 2 * lh * XPL + ((2 * spy + (tap >> 1)) * HW + 2 * spx + (tap & 1)) * 16;
#pragma unroll
          for (int h = 0; h < 2; ++h)
            if (sp < UPN) *reinterpret_cast<vec*>(dst + h * XPL) = upv[i][h];
        }
      }
    } else if (UPT && P == 0) {
      // x1 slice = 32 output channels of ConvTranspose2d(UPC, UPC, k2, s2)(src1) + bias for the halo tile: UKS MFMAs per 32 source
      // pixels and tap, results scattered to output pixel (2 sy + dy, 2 sx + dx) of the staging image: a lane's D registers are 16
      // consecutive output channels, i.e. K-slot planes 2 lh and 2 lh + 1 of its pixel
      if (pwave < MTU) {       // wave-uniform: this wave's source M-tile, all four taps
        const int sp = pwave * 32 + lr, spc = min(sp, UPN - 1);
        const int spy = (spc * 241) >> 12, spx = spc - spy * UPW;
        f32x16 cb5;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(sUpB + 32 * u_sl + 16 * lh + 4 * q4);
#pragma unroll
          for (int e = 0; e < 4; ++e) cb5[4 * q4 + e] = b4[e];
        }
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
          // A rows = the 32 output channels of this slice in the order that makes a lane's D registers 16 consecutive channels
          const int arow = tp * UPC + 32 * u_sl + (((lr >> 2) & 1) << 4) + ((lr >> 3) << 2) + (lr & 3);
          f32x16 cu = cb5;
#pragma unroll
          for (int ks = 0; ks < UKS; ++ks)
            cu = mfma32x16(*reinterpret_cast<const vec*>(sUpW + arow * (UPC * 2) + (2 * ks + lh) * 16), ubf[ks], cu);
          const int oy = u_iy0 + (tp >> 1) + 2 * spy, ox = u_ix0 + (tp & 1) + 2 * spx;
          const bool in_img = (unsigned)oy < (unsigned)a.H && (unsigned)ox < (unsigned)a.W;
          char* dst = st + 2 * lh * XPL + ((2 * spy + (tp >> 1)) * HW + 2 * spx + (tp & 1)) * 16;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = cu[8 * h + e];
            vec o = E::pack(f);
            if (!in_img) o = E::zero();
            if (sp < UPN) *reinterpret_cast<vec*>(dst + h * XPL) = o;
          }
        }
      }
    } else if (MODE == 4 && !UPT && P == 0) {
      // x1 = ConvTranspose2d(k2, s2)(src1) + bias for the halo tile: per 32 source pixels two MFMAs (K = 32 input channels),
      // results scattered to output pixel (2 sy + dy, 2 sx + dx) of the staging image: a lane's D registers are 16
      // consecutive output channels, i.e. K-slot planes 2 lh and 2 lh + 1 of its pixel
      const int iy0h = u_iy0 + (tap >> 1), ix0h = u_ix0 + (tap & 1);
#pragma unroll
      for (int i = 0; i < MT_PER; ++i) {
        const int mt = mt0 + MT_STEP * i;
        if (mt < MTU) {       // wave-uniform
          const int sp = mt * 32 + lr, spc = min(sp, UPN - 1);
          const int spy = (spc * 241) >> 12, spx = spc - spy * UPW;
          f32x16 cu = cb;
          cu = mfma32x16(ua[0], xr[2 * i], cu);
          cu = mfma32x16(ua[1], xr[2 * i + 1], cu);
          const bool in_img = (unsigned)(iy0h + 2 * spy) < (unsigned)a.H && (unsigned)(ix0h + 2 * spx) < (unsigned)a.W;
          const bool any_out = __builtin_amdgcn_ballot_w64(!in_img) != 0;      // wave-uniform: interior tiles skip the selects
          char* dst = st + 2 * lh * XPL + ((2 * spy + (tap >> 1)) * HW + 2 * spx + (tap & 1)) * 16;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = cu[8 * h + e];
            vec o = E::pack(f);
            if (any_out && !in_img) o = E::zero();
            if (sp < UPN) *reinterpret_cast<vec*>(dst + h * XPL) = o;
          }
        }
      }
    } else {
      // x2^2 / sqrt(x2 + 1e-8) as scalar fp32 (the packed v_pk_mul_f32 / v_pk_add_f32 forms made the concat layers 10 % slower);
      // the validity select only exists on the path of a wave that has an out-of-image slot at all (wave-uniform test):
      // interior tiles stage their registers as loaded
      auto transform = [&](vec v) __attribute__((always_inline)) {
        if (CAT && ssr_member(P) >= 2) {
          float f[8];
          E::unpack(v, f);
          if (ssr_member(P) == 2) {
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
      const bool wave_ok = __builtin_amdgcn_ballot_w64(!all_ok) == 0;
      if (wave_ok) {
#pragma unroll
        for (int j = 0; j <= RSN; ++j) {
          if (j == RSN && !e_on) continue;
          if (R_RAGGED && j == RSN - 1 && !r_last_on) continue;
          const int pix = j < RSN ? pix_r0 + j * RSTEP * HW : pix_e;
          *reinterpret_cast<vec*>(st + ch * XPL + pix * 16) = transform(xr[j]);
        }
      } else {
#pragma unroll
        for (int j = 0; j <= RSN; ++j) {
          if (j == RSN && !e_on) continue;
          if (R_RAGGED && j == RSN - 1 && !r_last_on) continue;
          vec v = transform(xr[j]);
          if (!((xvalid >> j) & 1u)) v = E::zero();
          const int pix = j < RSN ? pix_r0 + j * RSTEP * HW : pix_e;
          *reinterpret_cast<vec*>(st + ch * XPL + pix * 16) = v;
        }
      }
    }
    if (!RESW) {
      write_weights(st + XBYTES);
      if (bp && ptid < CT / 4) *reinterpret_cast<f32x4*>(sBias + bpar * CT + ptid * 4) = br;
    }
  };

  // EPO: the tile the multiplying waves parked before the last barrier goes to memory from here.  A staging thread owns K-slot
  // (ptid & 3) of pixels (ptid >> 2) & 15 and + 16 of a ROW PAIR (one pair per wave and pass): four 16-byte vectors, each wave
  // store = 16 whole pixels = 1 KiB of contiguous NHWC bytes; the 2x2 max-pool of the pair (unet_parts.py:212,233) is the signed
  // 16-bit maximum of the two rows and of the neighbouring pixel's lane (lane ^ 4), stored by the even pixels.  Offsets as in
  // epilogue_lean: one buffer descriptor per sample, an out-of-image row or column adds 2^30 and the hardware drops the store.
  TileCur sc;
  cur_init<TAIL>(sc, tile0, a);
  int spk = 0;                        // tiles stored so far
  auto store_parked = [&]() __attribute__((always_inline)) {
    typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
    typedef short s16x8s __attribute__((ext_vector_type(8)));
    constexpr unsigned BAD = 0x40000000u;
    constexpr int NSL = NT * 4;                      // K-slots (16-byte vectors) per pixel of the tile
    constexpr int PXW = 64 / NSL;                    // pixels per wave store: 16 x 64 B (32-channel tiles), 8 x 128 B (64-channel tiles)
    constexpr int NH = 32 / PXW;                     // wave stores per tile row
    const char* const pb = sPark + (spk & park_mask) * PARKB;
    const int slot = ptid & (NSL - 1), px = (ptid / NSL) & (PXW - 1);
    const int y0 = sc.ty * TH, x0 = sc.tx * TW, co = sc.ct * CT;
    const unsigned sample = (unsigned)(a.Hout * a.Wout * a.oC) * 2u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(a.out) + (size_t)sc.n * sample, (short)0,
                                                                        (int)sample, 0x00020000);
    const unsigned rowb = (unsigned)(a.Wout * a.oC) * 2u;
    const bool pool = a.pool_out != nullptr;          // wave-uniform
    const unsigned psample = (unsigned)(a.pH * a.pW * a.oC) * 2u;
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<char*>(pool ? a.pool_out : a.out) + (size_t)sc.n * (pool ? psample : sample), (short)0, (int)(pool ? psample : sample), 0x00020000);
    const unsigned prowb = (unsigned)(a.pW * a.oC) * 2u;
#pragma unroll
    for (int rp = pwave; rp < TH / 2; rp += PW) {
      s16x8s v[2][NH];
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int h = 0; h < NH; ++h)
          v[dy][h] = *reinterpret_cast<const s16x8s*>(pb + ((2 * rp + dy) * NSL + slot) * PSL + (PXW * h + px) * 16);
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {
        const int oy = y0 + 2 * rp + dy;
        const unsigned ro = ((unsigned)oy * rowb) | (oy < a.Hout ? 0u : BAD);
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          const int ox = x0 + PXW * h + px;
          const unsigned off = (((unsigned)(ox * a.oC + co + 8 * slot) * 2u) | (ox < a.Wout ? 0u : BAD)) + ro;
#ifdef UNCL_CHECKED
          if (off < BAD) UNCL_CHK(a.chk, reinterpret_cast<const char*>(a.out) + (size_t)sc.n * sample + off, 16);
#endif
          if (PC_ABL(1)) { asm volatile("" ::"v"(v[dy][h])); continue; }
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, v[dy][h]), rs, off, 0, UNCL_PC_STORE_AUX);
        }
      }
      if (pool) {
        const int gy = (y0 >> 1) + rp;
        const unsigned ro = ((unsigned)gy * prowb) | (gy < a.pH ? 0u : BAD);
#pragma unroll
        for (int h = 0; h < NH; ++h) {
          s16x8s m = __builtin_elementwise_max(v[0][h], v[1][h]);
          u32x4s u = __builtin_bit_cast(u32x4s, m), o;
          // the neighbouring pixel's lane: lane ^ 4 (four K-slots per pixel) / lane ^ 8 (eight)
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = (unsigned)__builtin_amdgcn_ds_swizzle((int)u[i], NSL == 4 ? 0x101F : 0x201F);
          m = __builtin_elementwise_max(m, __builtin_bit_cast(s16x8s, o));
          const int gx = (x0 >> 1) + (PXW / 2) * h + (px >> 1);
          const unsigned off = (((unsigned)(gx * a.oC + co + 8 * slot) * 2u) | (gx < a.pW ? 0u : BAD)) + ro;
          if ((px & 1) == 0) {
#ifdef UNCL_CHECKED
            if (off < BAD) UNCL_CHK(a.chk, reinterpret_cast<const char*>(a.pool_out) + (size_t)sc.n * psample + off, 16);
#endif
            if (PC_ABL(2)) { asm volatile("" ::"v"(m)); continue; }
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, m), prs, off, 0, UNCL_PC_STORE_AUX);
          }
        }
      }
    }
    ++spk;
    sc.kc = a.nk - 1;
    cur_next<TAIL>(sc, a, tile_end);
  };

  TileCur pc;
  cur_init<TAIL>(pc, tile0, a);
  const int total = (tile_end - tile0) * a.nk;     // chunks this workgroup walks
  int loaded = 0;                                  // chunks handed to load_step so far; the cursor points at chunk `loaded`
  auto load_next = [&](auto p_tag) __attribute__((always_inline)) {
    constexpr int P = decltype(p_tag)::value;
    if (loaded < total) {
      load_step(pc, p_tag, IntTag<0>{});
      // x1 of the NEXT slice (chunk loaded + 4 - P): xa has been free since this slice's x1 chunk was staged
      constexpr int SLOT = UNCL_PC_XA_SLOT, SPLIT = UNCL_PC_XA_SPLIT;
      constexpr int PART = !CAT || SLOT == 0 ? 0 : (!SPLIT ? (P == SLOT ? 3 : 0) : (P == SLOT ? 1 : (P == ((SLOT + 1) & 3) ? 2 : 0)));
      static_assert(!SPLIT || SLOT == 2 || SLOT == 1, "the second half's slot must not be the slice's own first step");
      if (PART != 0 && loaded + (4 - P) < total) {
        TileCur la = pc;
#pragma unroll
        for (int k = 0; k < 4 - P; ++k) cur_next<TAIL>(la, a, tile_end);
        load_step(la, IntTag<0>{}, IntTag<PART>{});
      }
      ++loaded;
      const int t_old = pc.tile;
      if (cur_next<TAIL>(pc, a, tile_end) && pc.tile != t_old) ppar = (ppar + 1) & 3;
    }
  };
  if (a.out1_w != nullptr && ptid < 33) { UNCL_CHK(a.chk, ptid < 32 ? a.out1_w + ptid : a.out1_b, 4); sO1[ptid] = ptid < 32 ? a.out1_w[ptid] : a.out1_b[0]; }     // fused 1x1 tail (CT == 32)
  if (O1C && ptid < 128) {
    // the outconv's A fragments as outc_frags builds them: lane (row fr, K-slot fh) of K-step ks holds piece fr (of three 16-bit pieces,
    // rows 0..2; zero elsewhere) of weights 16 ks + 8 (j >> 2) + 4 fh + (j & 3)
    const int ks = ptid >> 6, fr = ptid & 31, fh = (ptid >> 5) & 1;
    vec fv;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      UNCL_CHK(a.chk, a.out1_w + 16 * ks + 8 * (j >> 2) + 4 * fh + (j & 3), 4);
      float w = a.out1_w[16 * ks + 8 * (j >> 2) + 4 * fh + (j & 3)];
      T piece = (T)0.f;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const T h = (T)w;
        if (fr == t) piece = h;
        w -= (float)h;
      }
      fv[j] = piece;
    }
    *reinterpret_cast<vec*>(sO1F + ptid * 16) = fv;
  }
  if (TAIL && ptid < 32) { if (a.tail_b != nullptr) UNCL_CHK(a.chk, a.tail_b + ptid, 4); sB1[ptid] = a.tail_b != nullptr ? a.tail_b[ptid] : 0.f; }
  if (UPT) {
    // the up-conv's weights and bias become resident (read again per slice and tile by every staging wave)
    for (int i = ptid; i < 4 * UPC * UPC / 8; i += NPROD) *reinterpret_cast<vec*>(sUpW + i * 16) = LD16V(vec, a.up_w + i * 8);
    if (ptid < UPC) { if (a.up_b) UNCL_CHK(a.chk, a.up_b + ptid, 4); sUpB[ptid] = a.up_b ? a.up_b[ptid] : 0.f; }
    pc_barrier();       // every staging wave reads every row of them (the multiplying waves take this barrier idle)
  }
  if (TAIL) {
    // the second layer's weights become resident too: global [tap][cout][cin] -> plane (cin slot) x row (tap * 32 + cout)
    for (int i = ptid; i < 9 * 32 * 4; i += NPROD)
      *reinterpret_cast<vec*>(sW1 + (i & 3) * W1PL + (i >> 2) * 16) = LD16V(vec, a.tail_w + (i >> 2) * 32 + (i & 3) * 8);
  }
  if (RESW) {
    // the layer's whole weight tensor (one cout tile, nk chunks) becomes resident, in the consumers' chunk order, and so
    // does its bias (all four slots)
    for (int kc = 0; kc < a.nk; ++kc) {
      load_weights(0, kc);
      write_weights(wres + kc * WBYTES);
    }
    if (ptid < CT / 4) {
      if (a.bias != nullptr) UNCL_CHK(a.chk, a.bias + ptid * 4, 16);
      const f32x4 b4 = a.bias != nullptr ? *reinterpret_cast<const f32x4*>(a.bias + ptid * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int sl = 0; sl < 4; ++sl) *reinterpret_cast<f32x4*>(sBias + sl * CT + ptid * 4) = b4;
    }
  }
  if (CAT && UNCL_PC_XA_SLOT != 0) load_step(pc, IntTag<0>{}, IntTag<CAT ? 3 : 0>{});       // the first slice's x1
  load_next(IntTag<0>{});
  if (MODE == 3) {
    // one tile further ahead: patch 0 is parked in LDS, patch 1 requested, and a barrier publishes the former to every
    // staging wave before the first build
#pragma unroll
    for (int k = 0; k < IRN; ++k)
      if (ptid + k * NPROD < PN3) reinterpret_cast<unsigned*>(sP)[ptid + k * NPROD] = pack_head_tail(ir[k]);
    load_next(IntTag<1>{});
    pc_barrier();
  }
  if constexpr (TAIL) {
    up_compute(IntTag<0>{}, IntTag<MT_PER>{});
    // the second tile's x1 sources (from then on every tile's iteration 3 requests those of the tile after the next)
    if (total > a.nk) {
      TileCur la = pc;          // the cursor is at chunk 1
#pragma unroll
      for (int k = 0; k < 3; ++k) cur_next<TAIL>(la, a, tile_end);
      load_step(la, IntTag<0>{}, IntTag<3>{});
    }
  }
  write_step(smem, IntTag<0>{});
  load_next(IntTag<1>{});
  pc_barrier();                         // stage 0 is staged
  PCT(3)
  // iteration s (the consumers multiply chunk s): chunk s + 1 goes from registers to the stage the consumers left at the
  // last barrier, then the loads of chunk s + 2 are issued into the same registers
  int s = 0;
  auto iter = [&](auto q_tag) __attribute__((always_inline)) {
    constexpr int Q = decltype(q_tag)::value;
    if (EPO && NPARK == 2 && s > 0 && s % a.nk == 0) { store_parked(); if (MODE == 3) { PCT_K(0, 0) } }     // the tile that ended with chunk s - 1
    // (one buffer, four chunks per tile: the previous tile leaves during this tile's chunk UNCL_PC_EPO_Q -- 2: the lightest staging
    // iteration, the plain x2 slice; 3 would race with the multiplying waves' next park)
    if (EPO && NPARK == 1 && Q == UNCL_PC_EPO_Q && s >= a.nk) store_parked();
    if (s + 1 >= total) {
      pc_barrier();                     // the consumers' last chunk
      if (EPO) store_parked();          // ... and its tile
      return true;
    }
    write_step(smem + ((Q + 1) & 1) * STAGE, IntTag<(Q + 1) & 3>{});
    PCT_K(1, MODE == 3 ? 2 : Q)
    load_next(IntTag<(Q + 2) & 3>{});
    // the next tile's x1 chunk (chunk s + 3 / s + 2): one half of its source tiles per iteration, so that neither iteration
    // outlasts the multiplying waves' chunk by much
    if (TAIL && Q == 1 && s + 3 < total) up_compute(IntTag<0>{}, IntTag<MT_PER / 2>{});
    if (TAIL && Q == 2 && s + 2 < total) up_compute(IntTag<MT_PER / 2>{}, IntTag<MT_PER>{});
    PCT_K(2, MODE == 3 ? 3 : Q)
    pc_barrier();
    PCT_K(3, 4 + Q)
    ++s;
    return false;
  };
  for (;;) {
    if (iter(IntTag<0>{})) break;
    if (iter(IntTag<1>{})) break;
    if (iter(IntTag<2>{})) break;
    const bool done = iter(IntTag<3>{});
    if (TAIL) {
      // the multiplying waves write the intermediate image into stage 1, publish it, and run the second layer from it: nothing
      // may be staged there before the second of these barriers (the next tile's chunk 0 already sits in stage 0)
      pc_barrier();
      pc_barrier();
      PCT(0)
    }
    if (done) break;
  }
  PCT_FLUSH(4)
}

template <int NT, int MPW>
constexpr size_t pc_lds_bytes(bool resw, int nk, bool patch, int upc = 0, int epo = 0, bool o1c = false) {
  constexpr size_t xb = 4 * (size_t)pc_plane((MPW * 4 + 2) * 34), wb = 4 * (size_t)pc_plane(9 * NT * 32);
  return (resw ? 2 * xb + nk * wb : 2 * (xb + wb)) + 4 * NT * 32 * 4 + 64 * 4 + (patch ? 2 * (size_t)(MPW * 4 + 4) * 36 * 4 : 0) +
         (upc ? 64 * 4 + 4 * upc * upc * 2 : 0) +      // (the up-conv's bias and weights, where they are resident)
         (size_t)epo * (MPW * 4) * (NT * 4) * (NT == 2 ? 33 * 16 : 32 * 16) +      // (`epo` parked tiles)
         (o1c ? 2048 : 0);                                  // (the outconv's A fragments)
}

// fused last decoder stage: one workgroup per CU walks an even share of the (strip, row tile) steps
template <typename T, int MODE>
int launch_tail(PipeArgs& a, hipStream_t s) {
  // two activation stages of 14 x 34 halo pixels (unpadded planes), the first layer's four weight chunks, bias ring + outconv
  // weights + the second layer's bias, the second layer's weights, two carry buffers of two rows
  constexpr size_t lds = 2 * 4 * (size_t)(14 * 34 * 16) + 4 * 4 * (size_t)pc_plane16(9 * 32) + (4 * 32 + 64 + 32) * 4 +
                         4 * (size_t)(9 * 32 * 16) + 2 * 4 * (size_t)pc_plane16(2 * 34);
  static_assert(lds <= 163840, "one workgroup's LDS");
  auto kern = conv3x3_pc_kernel<T, 1, 3, MODE, 8, true, true>;
  static UnclDevOnce attr_done;
  if (attr_done.need()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return UNCL_ERR_LAUNCH;
    attr_done.done();
  }
  const int n_cu = uncl_cu_count();
  if (n_cu <= 0) return UNCL_ERR_LAUNCH;
  const int grid = a.total_tiles < n_cu ? a.total_tiles : n_cu;
  a.tiles_per_wg = (a.total_tiles + grid - 1) / grid;     // (unused by this form: shares are computed from the grid size)
  hipLaunchKernelGGL(kern, dim3(grid), dim3(768), lds, s, a);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

template <typename T, int NT, int MPW, int MODE, int PW, bool RESW, bool EPO = false, bool SSRB = false, int O1C = 0>
int launch_pc(PipeArgs& a, hipStream_t s) {
  constexpr int UPC_LDS = MODE == 5 ? 64 : (MODE == 4 && PW == 8 && UNCL_PC_UP_TILE ? 32 : 0);
  const size_t lds = pc_lds_bytes<NT, MPW>(RESW, a.nk, MODE == 3, UPC_LDS, EPO ? ((MODE == 4 || (NT == 2 && a.nk > 1)) ? 1 : 2) : 0, O1C != 0);
  static_assert(pc_lds_bytes<NT, MPW>(false, 0, false, UPC_LDS) <= 163840, "one workgroup's LDS");
  if (lds > 163840) return UNCL_ERR_ARG;
  auto kern = conv3x3_pc_kernel<T, NT, MPW, MODE, PW, RESW, false, EPO, SSRB, O1C>;
  static UnclDevOnce attr_done;
  if (attr_done.need()) {
    // the largest footprint this instance can be launched with (resident weights: up to four chunks of 32 / two of 64 channels)
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(RESW && !EPO && !O1C ? pc_lds_bytes<NT, MPW>(true, 4 / NT, MODE == 3, UPC_LDS)
                                                       : (EPO ? (size_t)163840 : lds))) != hipSuccess)     // (parked tiles: chunk count decides)
      return UNCL_ERR_LAUNCH;
    attr_done.done();
  }
  const int n_cu = uncl_cu_count();
  if (n_cu <= 0) return UNCL_ERR_LAUNCH;
  const int slots = (NT == 1 && MPW == 2) ? 2 * n_cu : n_cu;      // resident workgroups
  int grid = a.total_tiles < slots ? a.total_tiles : slots;
  a.tiles_per_wg = (a.total_tiles + grid - 1) / grid;
  grid = (a.total_tiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
  hipLaunchKernelGGL(kern, dim3(grid), dim3((4 + PW) * 64), lds, s, a);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

}  // namespace

#ifdef UNCL_PC_TIMING
// measurement builds only: copy (and optionally clear) the per-phase cycle counters
extern "C" int uncl_pc_timing_read(unsigned long long* out32, int reset) {
  if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_pc_t), sizeof(unsigned long long) * 32) != hipSuccess) return UNCL_ERR_LAUNCH;
  if (reset) {
    unsigned long long z[32] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_pc_t), z, sizeof(z)) != hipSuccess) return UNCL_ERR_LAUNCH;
  }
  return UNCL_OK;
}
#endif

template <typename T>
static int pc_launch_t(PipeArgs& a, int nt, int mpw, int mode, hipStream_t s) {
  if (a.nk < 1 || a.res != nullptr || a.flat_S != 0) return UNCL_ERR_ARG;
  static const int prio = [] { const char* e = getenv("UNCL_PC_PRIO"); return e ? atoi(e) : 1; }();
  a.pc_prio = prio;
  static const int resw_on = [] { const char* e = getenv("UNCL_PC_RESW"); return e ? atoi(e) : 1; }();
  // resident weights: one cout tile for the whole layer and its K-chunks fit next to two activation stages
  const bool resw = resw_on && a.n_ct == 1 && a.nk * nt <= 4;
  // a pooled copy beside 32-channel tiles: only the forward epilogue (ReLU, plain store) builds it
  const bool fwd_relu = a.slope == 0.f && a.mask == nullptr && !a.accumulate && (!a.skip_main || a.out1_w != nullptr);
  // the fused 1x1 tail: 32-channel tiles, forward epilogue, no pooled copy
  if (a.out1_w != nullptr && !(nt == 1 && fwd_relu && a.pool_out == nullptr && a.out1 != nullptr && a.out1_b != nullptr)) return UNCL_ERR_ARG;
  if (a.skip_main && a.out1_w == nullptr) return UNCL_ERR_ARG;
  // the straight-line forward epilogue addresses a sample through one 32-bit buffer descriptor
  static const int lean_on = [] { const char* e = getenv("UNCL_PC_LEAN"); return e ? atoi(e) : UNCL_PC_LEAN_DEFAULT; }();
  a.lean = lean_on && (size_t)a.Hout * a.Wout * a.oC * 2 < (1u << 30) && (size_t)a.pH * a.pW * a.oC * 2 < (1u << 30);
  if (nt == 1 && mpw == 4) {
    if (a.pool_out != nullptr && !fwd_relu) return UNCL_ERR_ARG;
    static const int pw = [] { const char* e = getenv("UNCL_PC_PW"); return e ? atoi(e) : 8; }();      // experiment: staging waves
    if (pw == 4) {
      if (mode == 3) return resw && a.nk == 1 ? launch_pc<T, 1, 4, 3, 4, true>(a, s) : UNCL_ERR_ARG;
      if (resw) {
        if (mode == 0) return launch_pc<T, 1, 4, 0, 4, true>(a, s);
        if (mode == 1) return launch_pc<T, 1, 4, 1, 4, true>(a, s);
        if (mode == 4) return launch_pc<T, 1, 4, 4, 4, true>(a, s);
      } else {
        if (mode == 0) return launch_pc<T, 1, 4, 0, 4, false>(a, s);
        if (mode == 1) return launch_pc<T, 1, 4, 1, 4, false>(a, s);
        if (mode == 4) return launch_pc<T, 1, 4, 4, 4, false>(a, s);
      }
      return UNCL_ERR_ARG;
    }
    // single-chunk layers with the plain forward store (optionally the pooled copy): the epilogue moves to the staging waves
    static const int epo_on = [] { const char* e = getenv("UNCL_PC_EPO"); return e ? atoi(e) : 1; }();
    const bool epo = epo_on && resw && a.nk == 1 && fwd_relu && a.lean && a.out1_w == nullptr && !a.skip_main;
    if (epo && mode == 0) return launch_pc<T, 1, 4, 0, 8, true, true>(a, s);
    // inference's last layer: only the one-channel map is wanted -> the 1x1 tail straight from the accumulators (O1C)
    static const int o1c_on = [] { const char* e = getenv("UNCL_PC_O1C"); return e ? atoi(e) : 2; }();
    if (o1c_on && mode == 0 && resw && a.nk == 1 && fwd_relu && a.lean && a.out1_w != nullptr && a.skip_main && a.pool_out == nullptr) {
      static const int o1c_prio = [] { const char* e = getenv("UNCL_PC_O1C_PRIO"); return e ? atoi(e) : -1; }();
      if (o1c_prio >= 0) a.pc_prio = o1c_prio;
      // two accumulator sets, four staging waves (the sigmoid's instantiation: the activation is compiled in, see tile_step)
      if (o1c_on >= 2 && a.out1_act == UNCL_ACT_SIGMOID) return launch_pc<T, 1, 4, 0, 4, true, false, false, 2>(a, s);
      return launch_pc<T, 1, 4, 0, 8, true, false, false, 1>(a, s);
    }
    if (mode == 3) return resw && a.nk == 1 ? launch_pc<T, 1, 4, 3, 8, true>(a, s) : UNCL_ERR_ARG;
    if (resw) {
      if (mode == 0) return launch_pc<T, 1, 4, 0, 8, true>(a, s);
      if (mode == 1) return launch_pc<T, 1, 4, 1, 8, true>(a, s);
      if (mode == 4) return launch_pc<T, 1, 4, 4, 8, true>(a, s);
    } else {
      if (mode == 0) return launch_pc<T, 1, 4, 0, 8, false>(a, s);
      if (mode == 1) return launch_pc<T, 1, 4, 1, 8, false>(a, s);
      if (mode == 4) return launch_pc<T, 1, 4, 4, 8, false>(a, s);
      if (mode == 5) return launch_pc<T, 1, 4, 5, 8, false>(a, s);
    }
    return UNCL_ERR_ARG;
  }
  if (nt == 1 && mpw == 3) {
    // 12-row tiles with the parked epilogue: the fused first layer (two parked tiles + its image patches do not fit beside 16-row
    // stages; 252 rows are 21 tiles exactly)
    static const int epo_on = [] { const char* e = getenv("UNCL_PC_EPO"); return e ? atoi(e) : 1; }();
    if (a.pool_out != nullptr && !fwd_relu) return UNCL_ERR_ARG;
    if (mode == 4) {
      // the four-chunk concat layer with the fused up-conv (inference's dominant launch): parked epilogue, one buffer
      if (!(epo_on && resw && a.nk == 4 && fwd_relu && a.lean && a.out1_w == nullptr && !a.skip_main && a.pool_out == nullptr)) return UNCL_ERR_ARG;
      return launch_pc<T, 1, 3, 4, 8, true, true>(a, s);
    }
    if (!(epo_on && resw && a.nk == 1 && fwd_relu && a.lean && a.out1_w == nullptr && !a.skip_main)) return UNCL_ERR_ARG;
    if (mode == 3) {
      // the fused first layer waits for its STAGING waves (round 6, tools/pc_phase_timing.py: per tile 1840 cycles storing the parked
      // tile + 1970 rebuilding the next halo tile, the multiplying waves 40 % at the barrier): they get the raised priority here, not
      // the multiplying waves -- same-box A/B 0.346 -> 0.327 ms per 200 tiles (the multi-chunk layers lose 6 % that way)
      static const int prio3 = [] { const char* e = getenv("UNCL_PC_PRIO3"); return e ? atoi(e) : 2; }();
      static const bool prio_env = getenv("UNCL_PC_PRIO") != nullptr;
      if (!prio_env) a.pc_prio = prio3;
      return launch_pc<T, 1, 3, 3, 8, true, true>(a, s);
    }
    if (mode == 0) return launch_pc<T, 1, 3, 0, 8, true, true>(a, s);
    return UNCL_ERR_ARG;
  }
  if (nt == 1 && mpw == 2) {
    // two workgroups per CU: single-chunk 32-channel layers with resident weights only
    if (a.pool_out != nullptr && !fwd_relu) return UNCL_ERR_ARG;
    if (!resw || a.nk != 1) return UNCL_ERR_ARG;
    if (mode == 0) return launch_pc<T, 1, 2, 0, 4, true>(a, s);
    if (mode == 3) return launch_pc<T, 1, 2, 3, 4, true>(a, s);
    return UNCL_ERR_ARG;
  }
  if (a.ssr_x2 != nullptr) {
    // data gradient of a skip-concat layer with the skip operator's backward as its epilogue: 64-channel tiles, plain source,
    // streamed weights (cout' = 4 C >= 128), bf16 only
    if constexpr (std::is_same<T, bf16_t>::value) {
      if (nt != 2 || mode != 0 || resw || a.pool_out != nullptr || a.out1_w != nullptr) return UNCL_ERR_ARG;
      if (mpw == 4) return launch_pc<T, 2, 4, 0, 4, false, false, true>(a, s);
      if (mpw == 2) return launch_pc<T, 2, 2, 0, 4, false, false, true>(a, s);
    }
    return UNCL_ERR_ARG;
  }
  if (nt == 2 && mpw == 4) {
    // 16 x 32 pixels x 64 channels: half the weight re-streaming per output of the 8-row tile; 128 accumulator registers
    if (a.pool_out != nullptr && !fwd_relu) return UNCL_ERR_ARG;
    if (resw) {
      if (mode == 0) return launch_pc<T, 2, 4, 0, 4, true>(a, s);
      if (mode == 1) return launch_pc<T, 2, 4, 1, 4, true>(a, s);
    } else {
      if (mode == 0) return launch_pc<T, 2, 4, 0, 4, false>(a, s);
      if (mode == 1) return launch_pc<T, 2, 4, 1, 4, false>(a, s);
    }
    return UNCL_ERR_ARG;
  }
  if (nt == 2 && mpw == 2) {
    // 64-channel tiles: the multiplying waves need 234 registers (two fragment sets of ten vectors), which leaves room for two
    // waves per SIMD, i.e. four staging waves
    // Round 5: ... the 8-row form itself fits 168 registers, i.e. three waves per SIMD and EIGHT staging waves (UNCL_PC_NT2_PW8: 1 =
    // plain sources, 2 = concat sources too).  Same-box A/B at 200 tiles: down_path.2 second conv (24 x 24: the layer that keeps
    // 8-row tiles) 0.166 -> 0.155 ms; forcing 8-row tiles with eight staging waves where 16-row tiles with four are chosen today
    // loses (down_path.1 second conv 0.202 -> 0.219), so the choice between the two tile heights stays as it was.
    static const int pw8 = [] { const char* e = getenv("UNCL_PC_NT2_PW8"); return e ? atoi(e) : 1; }();
    // parked epilogue (the launcher asks for it through a.epo2): one cout tile, one or two resident chunks, forward store (+ pooled copy)
    if (a.epo2 && mode == 0 && resw && a.nk <= 2 && fwd_relu && a.lean && a.out1_w == nullptr && !a.skip_main)
      return launch_pc<T, 2, 2, 0, 8, true, true>(a, s);
    if (a.epo2) return UNCL_ERR_ARG;
    if (pw8 && mode == 0) return resw ? launch_pc<T, 2, 2, 0, 8, true>(a, s) : launch_pc<T, 2, 2, 0, 8, false>(a, s);
    if (pw8 >= 2 && mode == 1 && !resw) return launch_pc<T, 2, 2, 1, 8, false>(a, s);
    if (resw) {
      if (mode == 0) return launch_pc<T, 2, 2, 0, 4, true>(a, s);
      if (mode == 1) return launch_pc<T, 2, 2, 1, 4, true>(a, s);
    } else {
      if (mode == 0) return launch_pc<T, 2, 2, 0, 4, false>(a, s);
      if (mode == 1) return launch_pc<T, 2, 2, 1, 4, false>(a, s);
    }
    return UNCL_ERR_ARG;
  }
  return UNCL_ERR_ARG;
}

int uncl_conv3x3_tail_launch(PipeArgs& a, int dtype, hipStream_t s) {
  // concat-ssr source with the 2x2 up-conv recomputed in the loader, 128 -> 32 channels, ReLU, then 32 -> 32, ReLU, 1x1
  if (a.nk != 4 || a.Cout != 32 || a.res != nullptr || a.flat_S != 0 || a.slope != 0.f || a.mask != nullptr || a.accumulate ||
      a.tail_w == nullptr || a.out1_w == nullptr || a.out1_b == nullptr || a.out1 == nullptr || a.up_w == nullptr || a.pad != 2)
    return UNCL_ERR_ARG;
  // wave priorities: none.  Here the STAGING waves are what the multiplying waves wait for (four of them, with the up-conv's
  // MFMAs in their heaviest iteration): raising the multiplying waves (the default of the other forms) measured 1.113 ms per 200
  // tiles against 1.085 without priorities, raising the staging waves 1.162
  static const int prio = [] { const char* e = getenv("UNCL_TAIL_PRIO"); return e ? atoi(e) : 0; }();
  a.pc_prio = prio;
  a.lean = 1;
  a.n_ct = 1;
  a.oH = a.Hout + 2; a.oW = a.Wout + 2;
  a.tiles_x = (a.oW + 29) / 30;               // strips of 30 result columns (32 intermediate columns each)
  a.tiles_y = (a.oH + 11) / 12;               // 12 result rows per step
  a.total_tiles = a.flat_N * a.tiles_x * a.tiles_y;
  if (dtype == UNCL_F16) return launch_tail<f16_t, 4>(a, s);
  return launch_tail<bf16_t, 4>(a, s);
}

int uncl_conv3x3_pc_launch(PipeArgs& a, int dtype, int nt, int mpw, int mode, hipStream_t s) {
  if (dtype == UNCL_F16) return pc_launch_t<f16_t>(a, nt, mpw, mode, s);
  return pc_launch_t<bf16_t>(a, nt, mpw, mode, s);
}
