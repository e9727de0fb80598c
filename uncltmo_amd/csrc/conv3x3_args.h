// Launch descriptor and load helpers shared by the two bf16 3x3 implicit-GEMM kernels (conv3x3_pipe.hip: four waves that
// stage and multiply in turn, two workgroups per CU; conv3x3_pc.hip: four multiplying + four staging waves, one per CU).
#pragma once
#include "common.h"

struct PipeArgs {
  const bf16_t* src0;
  const bf16_t* src1;
  const bf16_t* prev0;
  const bf16_t* weight;
  const float* bias;
  const bf16_t* res;
  const bf16_t* mask;   // backward: stored value is zeroed where mask <= 0 (ReLU of the layer that produced `mask`)
  int accumulate;       // backward: add to what `out` already holds
  float mask_slope;     // backward: derivative on the non-positive side (0 ReLU, 0.2 LeakyReLU)
  bf16_t* out;
  bf16_t* pool_out;
  const float* out1_w;
  const float* out1_b;
  float* out1;
  int H, W, Cin, Cout, pad;
  int s0H, s0W, s0C, s1H, s1W, s1C, prev_ch;
  float slope;  // activation as max(t,0) + slope*min(t,0): 0 relu, 0.2 leaky, 1 identity
  int res_b0;
  int Hout, Wout, oC;
  int pH, pW;
  int tiles_x, tiles_y, n_ct, total_tiles, tiles_per_wg, nk;
  int out1_act, skip_main;
  const float* img;     // MODE 3: fp32 one-channel image (N, imgH, imgW); x = act(conv3x3_valid(img; pre_w, pre_b))
  const float* pre_w;   // (32,1,3,3)
  const float* pre_b;   // (32) or NULL
  int imgH, imgW;
  const int* img_off;   // MODE 3, optional: element offset of sample n's (imgH x imgW) window inside `img` (tiles cut out of larger frames);
  int imgP;             //          row pitch of `img` in elements (imgW when img_off is NULL: samples back to back)
  int flat_S, flat_hw, flat_N;  // FLAT: whole samples per tile, output pixels per sample, samples in the batch
  const bf16_t* up_w;   // MODE 4: packed [4 taps][32 cout][32 cin] weights of the 2x2 stride-2 transposed conv
  const float* up_b;    // MODE 4: its bias (32) or NULL
  int pc_prio;          // conv3x3_pc: 0 no priorities, 1 multiplying waves raised, 2 staging waves raised
  int lean;             // conv3x3_pc: 1 = plain forward stores take the straight-line epilogue (buffer stores, no branches)
  // conv3x3_pc TAIL (fused last decoder stage, inference): this layer's 32-channel output (Hout x Wout) never leaves the CU --
  // a second transposed 3x3 (32 -> 32, pad-2 correlation over tail_w) and the 1x1 outconv run from an LDS image of it
  const bf16_t* tail_w; // packed [9 taps][32 cout][32 cin] of the second layer
  const float* tail_b;  // its bias (32) or NULL
  int oH, oW;           // extent of the final one-channel map: Hout + 2, Wout + 2
  int o1_lds_off;       // conv3x3_pipe: byte offset of the parked outconv fragments in LDS (register-direct 1x1 tail)
  int epo2;             // conv3x3_pc, 8-row 64-channel tiles: 1 = the launcher asks for the parked epilogue (round 6)
  // conv3x3_pc, gradient mode, SSRB epilogue: this launch is the data gradient of a skip-concat layer (cout' = 4 C in the interleaved
  // order of uncl_pack_item.cout_order = 1) and its epilogue is the backward of the skip operator (unet_parts.py:319-322):
  // g_x2 = (g0 + 2 x2 g2 + g3 / (2 sqrt(x2 + 1e-8))) relu'(x2) [written or accumulated], g_x1 = g1; the 4 C-channel gradient of
  // the concatenation never reaches memory
  const bf16_t* ssr_x2;
  bf16_t* ssr_gx2;
  bf16_t* ssr_gx1;
  int ssr_C, ssr_acc;
  // conv3x3_flat: output pixels linearised on the padded input's pitch, M-tiles of 32 of them (conv3x3_flat.hip)
  int fl_pitch, fl_mts, fl_halo, fl_total_mt;   // Wout + 2, M-tiles per sample, 2 pitch + 2, M-tiles in the batch
  int fl_cmax, fl_ct_shift;                     // sample borders one tile can span, log2(n_ct)
  unsigned fl_div_pitch, fl_div_mts;            // floor(2^32 / d) + 1: x / d = umulhi(x, .) for x d < 2^32
  UNCL_CHK_MEMBER       // checked build: the tensors of this launch (common.h)
};

// conv3x3_pc.hip: the producer / consumer kernel for the multi-chunk layers; returns UNCL_ERR_ARG when (nt, mpw, mode) is not built
int uncl_conv3x3_pc_launch(PipeArgs& a, int dtype, int nt, int mpw, int mode, hipStream_t s);
// conv3x3_pc.hip, TAIL form: concat-ssr + fused up-conv source (mode 4) -> transposed 3x3 -> transposed 3x3 -> 1x1 + last
// activation as ONE launch; a.Hout / a.Wout = extent of the intermediate map, a.oH / a.oW = extent of the result a.out1
int uncl_conv3x3_tail_launch(PipeArgs& a, int dtype, hipStream_t s);
// conv3x3_flat.hip: flat M-tiles for the 64-channel-tile layers of the small / mid-size maps (mode 0 plain, 1 concat-ssr);
// UNCL_ERR_ARG where it does not apply.  mpw_pref 0: choose the M-tiles per multiplying wave; 2 / 3 / 4: that or nothing;
// max_cost > 0: only below that cost (rounds of the persistent grid x (M-tiles per wave + 0.35))
int uncl_conv3x3_flat_launch(PipeArgs& a, int dtype, int mode, int mpw_pref, double max_cost, hipStream_t s);

namespace {

// the descriptor's 16-bit tensors are typed bf16_t for addressing only; the kernels read them as their own element type T
__device__ __forceinline__ bf16x8 ld16(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
template <typename V>
__device__ __forceinline__ V ld16v(const bf16_t* p) { return *reinterpret_cast<const V*>(p); }
template <typename V>
__device__ __forceinline__ V ld16ov(const bf16_t* base, unsigned byte_off) {
  asm volatile("" : "+v"(byte_off));     // see ld16o
#if defined(UNCL_LOAD_NT) && UNCL_LOAD_NT
  // (A/B build, tools/ab_variants.sh: every 16-byte vector load of the 3x3 kernels non-temporal)
  return __builtin_nontemporal_load(reinterpret_cast<const V*>(reinterpret_cast<const char*>(base) + byte_off));
#else
  return *reinterpret_cast<const V*>(reinterpret_cast<const char*>(base) + byte_off);
#endif
}
// wave-uniform base + 32-bit per-lane BYTE offset: lowers to global_load_dwordx4 v, v_off, s[base] (no address VGPR pair)
__device__ __forceinline__ bf16x8 ld16o(const bf16_t* base, unsigned byte_off) {
  // The empty asm keeps the zero-extension of the offset next to the load: hoisted out of the loop it turns the access
  // into a 64-bit VGPR address that the compiler builds inside the destination registers, and the write-after-write
  // hazard check then waits (vmcnt) for every load issued so far before the next one can go out.
  asm volatile("" : "+v"(byte_off));
  return *reinterpret_cast<const bf16x8*>(reinterpret_cast<const char*>(base) + byte_off);
}


template <int V>
struct IntTag { static constexpr int value = V; };

// Concat-ssr sources (unet_parts.py:319-322, weight K layout [x2 | x1 | x2^2 | sqrt(x2 + 1e-8)]): every 32-channel slice of the
// skip is walked as four K-chunks, phase ph = chunk index & 3 -> member g of that layout.  Order [x1, sqrt, x2^2, x2]: the x2
// registers are loaded once (for ph 1) and staged three times; the staging with the transcendental comes first and the plain copy
// last, because the iteration that stages the LAST of the three also has to request the next slice's x2 registers
// (producer / consumer kernel: balances the staging iterations against the multiplying waves' steps).  Both 3x3 kernels use this
// one mapping, so their accumulation order -- and every output bit -- is the same.
#ifndef UNCL_SSR_ORDER
#define UNCL_SSR_ORDER 1
#endif
__host__ __device__ constexpr int ssr_member(int ph) {
  return UNCL_SSR_ORDER ? (ph == 0 ? 1 : (ph == 1 ? 3 : (ph == 2 ? 2 : 0))) : (ph == 0 ? 1 : (ph == 1 ? 0 : ph));
}
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 ld16o_f32(const float* base, unsigned byte_off) {
  asm volatile("" : "+v"(byte_off));
  return *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(base) + byte_off);
}

}  // namespace

// checked build: the same loads with the address looked up in the launch's tensor table first (`a` = the kernel's PipeArgs)
#define LD16V(V, p) (UNCL_CHK(a.chk, (p), 16), ld16v<V>(p))
#define LD16OV(V, base, off) (UNCL_CHK(a.chk, reinterpret_cast<const char*>(base) + (off), 16), ld16ov<V>((base), (off)))
#define LD16O(base, off) (UNCL_CHK(a.chk, reinterpret_cast<const char*>(base) + (off), 16), ld16o((base), (off)))
#define LD16O_F32(base, off) (UNCL_CHK(a.chk, reinterpret_cast<const char*>(base) + (off), 16), ld16o_f32((base), (off)))
