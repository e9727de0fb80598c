// 3x3 implicit-GEMM convolution with LDS-DMA staging (gfx950 `global_load_lds_dwordx4`), bf16, plain source.
//
// Same GEMM orientation, tile shapes and sliding-window fragment reuse as conv3x3_pipe.hip, but the operands never pass
// through VGPRs on their way to LDS:
//   * K is walked in 16-channel steps; a stage = halo tile [HH*34 px][32 B] + weights [9*CT rows][32 B] (29 KiB), two
//     stages per workgroup, two workgroups per CU;
//   * the DMA of step s+1 is issued right after the barrier that opens step s and lands while the 36 MFMAs of step s run
//     (and, at a tile boundary, while the epilogue runs: the epilogue transposes through a WAVE-PRIVATE 2 KiB scratch, so
//     it needs neither a barrier nor the staging buffers);
//   * one barrier per step, preceded by `s_waitcnt vmcnt(0)` for this wave's pieces of the stage about to be read;
//   * a DMA wave-instruction writes 1 KiB of LDS linearly (lane i -> byte 16 i), so the 32-byte rows cannot be padded.
//     Bank conflicts are avoided by swapping the two 16-byte halves of a row where bit 3 of the pixel column (activations)
//     or of the output channel (weights) is set -- applied to the per-lane SOURCE address of the DMA and to the fragment
//     read address, which stays "lane base + immediate" (the swap bit is a per-lane constant for each horizontal tap).
// Out-of-image halo pixels are fetched from a zero page.  The DMA is issued from inline asm (hipcc would otherwise drain
// it with vmcnt(0) before every LDS read that might alias it); all waits on it are explicit.
//
// Scope: forward 3x3 layers with a plain bf16 NHWC source, bias, ReLU / LeakyReLU / identity, optional fused 2x2 max-pool
// copy.  Everything else (concat loaders, residuals, gradient epilogues, video hand-off, fused 1x1 tail) stays on
// conv3x3_pipe.hip.
#include "common.h"

namespace {

__device__ __attribute__((aligned(1024))) unsigned char g_zero_page[1024];

struct DmaArgs {
  const bf16_t* src;
  const bf16_t* weight;   // packed [tap][Cout][Cin]
  const float* bias;
  bf16_t* out;
  bf16_t* pool_out;
  int H, W, C, Cin, Cout, pad;
  int Hout, Wout, pH, pW;
  float slope;
  int tiles_x, tiles_y, n_ct, total_tiles, tiles_per_wg, nk16;
};

typedef short s16x4 __attribute__((ext_vector_type(4)));

// one LDS-DMA piece: 64 lanes x 16 bytes -> 1 KiB at LDS byte address lds_dst; src = sbase + voff (bytes)
__device__ __forceinline__ void dma16_s(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst));
}
// per-lane 64-bit source
__device__ __forceinline__ void dma16_v(const void* src, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(src), "s"(lds_dst));
}

template <int NT>
__global__ __launch_bounds__(256, 2) void conv3x3_dma_kernel(const DmaArgs a) {
  using vec = bf16x8;
  constexpr int MPW = 4 / NT;                 // NT=1: 16x32x32 tiles, NT=2: 8x32x64
  constexpr int TH = MPW * 4, TW = 32, HH = TH + 2, HW = TW + 2, NPIX = HH * HW;
  constexpr int CT = NT * 32, WROWS = 9 * CT;
  constexpr int NPX = (NPIX * 32 + 1023) / 1024, NPWP = (WROWS * 32) / 1024;   // 1 KiB pieces per stage
  static_assert((WROWS * 32) % 1024 == 0, "weight image is a whole number of DMA pieces");
  constexpr int XS = NPX * 1024, STAGE = XS + NPWP * 1024, NP = NPX + NPWP;
  constexpr int KP = (NP + 3) / 4;            // pieces per wave
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  // [stage 0][stage 1][4 x 2 KiB wave scratch]
  char* sScr = smem + 2 * STAGE;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: it selects LDS destinations (M0) and DMA pieces
  const int lr = lane & 31, lh = lane >> 5;

  int tile = (int)blockIdx.x * a.tiles_per_wg;
  const int tile_end = min(tile + a.tiles_per_wg, a.total_tiles);
  if (tile >= tile_end) return;

  // ---- per-lane constants of this wave's DMA pieces (piece i = wave + 4k)
  unsigned poff[KP];      // byte offset of the lane's 16 bytes relative to the step's scalar base
  unsigned pyx[KP];       // activations: (py << 16) | px of the lane's halo pixel, 0xffffffff past the end of the image
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    const int i = wave + 4 * k;
    const int S = (i < NPX ? i : i - NPX) * 64 + lane, R = S >> 1, h = S & 1;
    if (i < NPX) {
      const int py = R / HW, px = R - py * HW;
      const int lhalf = h ^ ((px >> 3) & 1);
      poff[k] = (unsigned)(((py * a.W + px) * a.C) * 2 + lhalf * 16);
      pyx[k] = R < NPIX ? (unsigned)((py << 16) | px) : 0xffffffffu;
    } else {
      const int tap = R / CT, co = R - tap * CT;
      const int lhalf = h ^ ((co >> 3) & 1);
      poff[k] = (unsigned)(((tap * a.Cout + co) * a.Cin) * 2 + lhalf * 16);
      pyx[k] = 0;
    }
  }
  // LDS byte address of the dynamic segment (the 32-bit address-space-3 pointer value)
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  // ---- fragment read bases (bytes inside a stage)
  // B: pixel (r, lr + tx): row index (r*HW + lr + tx), half swapped on bit 3 of (lr + tx)
  int bB[3];
#pragma unroll
  for (int tx = 0; tx < 3; ++tx) {
    const int px = lr + tx;
    bB[tx] = (wave * MPW * HW + px) * 32 + ((lh ^ ((px >> 3) & 1)) << 4);
  }
  const int bA = XS + lr * 32 + ((lh ^ ((lr >> 3) & 1)) << 4);

  // ---- tile cursor
  int t_ct, t_tx, t_ty, t_n;
  {
    int r = tile;
    t_ct = r % a.n_ct; r /= a.n_ct;
    t_tx = r % a.tiles_x; r /= a.tiles_x;
    t_ty = r % a.tiles_y; r /= a.tiles_y;
    t_n = r;
  }
  // issue the DMA of K-step `ks` of tile (n, y0, x0, co0) into stage `st`
  auto issue = [&](int n, int y0, int x0, int co0, int ks, int st, int k_begin, int k_end) __attribute__((always_inline)) {
    const int iy0 = y0 - a.pad, ix0 = x0 - a.pad;
    const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + HH <= a.H && ix0 + HW <= a.W;
    const char* xb = reinterpret_cast<const char*>(a.src + ((size_t)n * a.H * a.W + (size_t)iy0 * a.W + ix0) * a.C + ks * 16);
    const char* wb = reinterpret_cast<const char*>(a.weight + (size_t)co0 * a.Cin + ks * 16);
    const unsigned sbase = lds0 + (unsigned)st * STAGE;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      if (k < k_begin || k >= k_end) continue;
      const int i = wave + 4 * k;
      if (i >= NP) continue;
      const unsigned dst = sbase + (unsigned)i * 1024;
      if (i >= NPX) {
        dma16_s(wb, poff[k], dst);
      } else if (interior) {
        if (pyx[k] != 0xffffffffu) dma16_s(xb, poff[k], dst);   // lanes past the halo tile's end do not write
      } else {
        const int py = (int)(pyx[k] >> 16), px = (int)(pyx[k] & 0xffffu);
        const bool in_img = (unsigned)(iy0 + py) < (unsigned)a.H && (unsigned)(ix0 + px) < (unsigned)a.W;
        const char* src = in_img ? xb + poff[k] : reinterpret_cast<const char*>(g_zero_page) + lane * 16;
        if (pyx[k] != 0xffffffffu) dma16_v(src, dst);
      }
    }
  };

  f32x16 acc[MPW][NT];
  f32x16 zero16;
#pragma unroll
  for (int i = 0; i < 16; ++i) zero16[i] = 0.f;

  int c_n = t_n, c_y0 = t_ty * TH, c_x0 = t_tx * TW, c_co = t_ct * CT, c_ks = 0;
  int stage = 0;
  issue(c_n, c_y0, c_x0, c_co, 0, 0, 0, KP);

  for (;;) {
    // ---- next step's coordinates
    int n_n = c_n, n_y0 = c_y0, n_x0 = c_x0, n_co = c_co, n_ks = c_ks + 1;
    bool more = true;
    if (n_ks == a.nk16) {
      n_ks = 0;
      if (++tile >= tile_end) {
        more = false;
      } else {
        if (++t_ct == a.n_ct) {
          t_ct = 0;
          if (++t_tx == a.tiles_x) {
            t_tx = 0;
            if (++t_ty == a.tiles_y) { t_ty = 0; ++t_n; }
          }
        }
        n_n = t_n; n_y0 = t_ty * TH; n_x0 = t_tx * TW; n_co = t_ct * CT;
      }
    }
    // ---- this wave's pieces of the current stage have landed; after the barrier everybody's have, and everybody is
    // done reading the other stage (it was the previous step's)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");   // no LDS read of the new stage may be hoisted above the barrier
    // the next step's pieces go out in three instalments, one ahead of each horizontal-tap group of MFMAs, instead of as
    // one burst of 7-8 per wave into the CU's address pipe
    constexpr int K1 = (KP + 2) / 3, K2 = 2 * K1 < KP ? 2 * K1 : KP;

    // ---- MFMA phase over the 16-channel step
    const char* sS = smem + stage * STAGE;
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
      if (more) issue(n_n, n_y0, n_x0, n_co, n_ks, stage ^ 1, tx == 0 ? 0 : (tx == 1 ? K1 : K2), tx == 0 ? K1 : (tx == 1 ? K2 : KP));
      vec A[3][NT], B[MPW + 2];
#pragma unroll
      for (int ty = 0; ty < 3; ++ty)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          A[ty][nt] = *reinterpret_cast<const vec*>(sS + bA + ((ty * 3 + tx) * CT + nt * 32) * 32);
#pragma unroll
      for (int r = 0; r < MPW + 2; ++r) B[r] = *reinterpret_cast<const vec*>(sS + bB[tx] + r * HW * 32);
      if (tx == 0) {
        // first tap of the step: on the first step of a tile the accumulation starts from zero (inline C = 0)
        if (c_ks == 0) {
#pragma unroll
          for (int m = 0; m < MPW; ++m)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0][nt], B[m], zero16, 0, 0, 0);
        } else {
#pragma unroll
          for (int m = 0; m < MPW; ++m)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0][nt], B[m], acc[m][nt], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < MPW; ++m)
#pragma unroll
          for (int ty = 1; ty < 3; ++ty)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ty][nt], B[m + ty], acc[m][nt], 0, 0, 0);
      } else {
#pragma unroll
        for (int m = 0; m < MPW; ++m)
#pragma unroll
          for (int ty = 0; ty < 3; ++ty)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[m][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[ty][nt], B[m + ty], acc[m][nt], 0, 0, 0);
      }
    }

    // ---- tile finished: epilogue from the accumulators through this wave's private scratch
    if (c_ks == a.nk16 - 1) {
      char* st = sScr + wave * 2048;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        float bv[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 b = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + c_co + nt * 32 + 8 * q + 4 * lh) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 4; ++e) bv[4 * q + e] = b[e];
        }
        bf16x4 keep[MPW][4];
#pragma unroll
        for (int m = 0; m < MPW; ++m) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float t = acc[m][nt][4 * q + e] + bv[4 * q + e];
              o[e] = (bf16_t)(fmaxf(t, 0.f) + a.slope * fminf(t, 0.f));
            }
            keep[m][q] = o;
            *reinterpret_cast<bf16x4*>(st + lr * 64 + ((q ^ ((lr >> 1) & 3)) << 4) + (lh << 3)) = o;
          }
          const int oy = c_y0 + wave * MPW + m;
#pragma unroll
          for (int ps = 0; ps < 2; ++ps) {
            const int p = ps * 16 + (lane >> 2), sl = lane & 3;
            const vec v = *reinterpret_cast<const vec*>(st + p * 64 + ((sl ^ ((p >> 1) & 3)) << 4));
            const int ox = c_x0 + p;
            if (oy < a.Hout && ox < a.Wout)
              *reinterpret_cast<vec*>(a.out + (((size_t)c_n * a.Hout + oy) * a.Wout + ox) * a.Cout + c_co + nt * 32 + sl * 8) = v;
          }
        }
        if (a.pool_out != nullptr) {
          // 2x2 max: rows (m, m+1) are in this lane, the neighbouring column is lane ^ 1 (bf16 max on the stored values)
#pragma unroll
          for (int m = 0; m < MPW; m += 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              float f0[4], f1[4];
              bf16x4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                f0[e] = fmaxf((float)keep[m][q][e], (float)keep[m + 1][q][e]);
                f1[e] = __shfl_xor(f0[e], 1, 64);
                o[e] = (bf16_t)fmaxf(f0[e], f1[e]);
              }
              // even lanes own pooled pixel lr/2
              if ((lr & 1) == 0) {
                const int pp = lr >> 1;
                *reinterpret_cast<bf16x4*>(st + pp * 64 + ((q ^ ((pp >> 1) & 3)) << 4) + (lh << 3)) = o;
              }
            }
            const int gy = ((c_y0 + wave * MPW + m) >> 1);
            {
              const int p = lane >> 2, sl = lane & 3;        // 16 pooled pixels x 4 slots = 64 lanes
              const vec v = *reinterpret_cast<const vec*>(st + p * 64 + ((sl ^ ((p >> 1) & 3)) << 4));
              const int gx = (c_x0 >> 1) + p;
              if (gy < a.pH && gx < a.pW)
                *reinterpret_cast<vec*>(a.pool_out + (((size_t)c_n * a.pH + gy) * a.pW + gx) * a.Cout + c_co + nt * 32 + sl * 8) = v;
            }
          }
        }
      }
    }
    if (!more) break;
    c_n = n_n; c_y0 = n_y0; c_x0 = n_x0; c_co = n_co; c_ks = n_ks;
    stage ^= 1;
  }
}

template <int NT>
int launch_dma(DmaArgs& a, hipStream_t s) {
  constexpr int MPW = 4 / NT, TH = MPW * 4, NPIX = (TH + 2) * 34, CT = NT * 32;
  constexpr int XS = ((NPIX * 32 + 1023) / 1024) * 1024, STAGE = XS + 9 * CT * 32;
  constexpr size_t lds = 2 * (size_t)STAGE + 4 * 2048;
  auto kern = conv3x3_dma_kernel<NT>;
  static bool attr_done = false;
  static int max_blocks = 0;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return UNCL_ERR_LAUNCH;
    hipDeviceProp_t p;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return UNCL_ERR_LAUNCH;
    max_blocks = 2 * p.multiProcessorCount;
    attr_done = true;
  }
  int grid = a.total_tiles < max_blocks ? a.total_tiles : max_blocks;
  a.tiles_per_wg = (a.total_tiles + grid - 1) / grid;
  grid = (a.total_tiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

}  // namespace

// Returns UNCL_OK if the layer was launched here, UNCL_ERR_ARG if it is outside this kernel's scope (the caller then uses
// conv3x3_pipe), another error code on a launch failure.
int conv3x3_dma_try(const uncl_conv_desc* d, void* pool_out, hipStream_t s) {
  if (d->dtype != UNCL_BF16 || d->ksize != 3 || d->src_mode != UNCL_SRC_PLAIN) return UNCL_ERR_ARG;
  if (d->prev0 != nullptr || d->res != nullptr || d->out1_w != nullptr || d->skip_main_store || d->out == nullptr) return UNCL_ERR_ARG;
  if (d->Cin % 16 != 0 || d->Cout % 64 != 0 || d->src0_C != d->Cin) return UNCL_ERR_ARG;
  if ((long long)d->src0_H * d->src0_W * d->src0_C * 2 >= (1LL << 31)) return UNCL_ERR_ARG;
  DmaArgs a;
  a.src = (const bf16_t*)d->src0; a.weight = (const bf16_t*)d->weight; a.bias = d->bias;
  a.out = (bf16_t*)d->out; a.pool_out = (bf16_t*)pool_out;
  a.H = d->H; a.W = d->W; a.C = d->src0_C; a.Cin = d->Cin; a.Cout = d->Cout; a.pad = d->pad;
  if (d->src0_H != d->H || d->src0_W != d->W) return UNCL_ERR_ARG;
  a.Hout = d->H + 2 * d->pad - 2; a.Wout = d->W + 2 * d->pad - 2;
  if (a.Hout <= 0 || a.Wout <= 0 || d->out_C != d->Cout) return UNCL_ERR_ARG;
  a.pH = a.Hout / 2; a.pW = a.Wout / 2;
  a.slope = d->act == UNCL_ACT_RELU ? 0.f : (d->act == UNCL_ACT_LRELU ? 0.2f : 1.f);
  a.nk16 = d->Cin / 16;
  constexpr int TH = 8;
  a.n_ct = d->Cout / 64;
  a.tiles_x = (a.Wout + 31) / 32; a.tiles_y = (a.Hout + TH - 1) / TH;
  a.total_tiles = d->N * a.tiles_x * a.tiles_y * a.n_ct;
  return launch_dma<2>(a, s);
}
