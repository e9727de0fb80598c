// Pipelined 3x3 implicit-GEMM convolution for gfx950 — the generator's hot kernel (bf16).
//
// Same GEMM orientation and LDS images as conv_igemm.hip (D[cout][pixel] = W[cout][k] X[k][pixel], 64-byte
// K-chunks, XOR-swizzled 16-byte slots) with three structural changes that matter on MI355X:
//
//  * persistent workgroups + register prefetch: a workgroup walks a list of (tile, K-chunk) steps; the global
//    loads of step s+1 are issued before the MFMAs of step s and land in VGPRs while the matrix pipe is busy,
//    then are written to the single LDS buffer after a barrier (HBM/L2 latency hides under MFMA, two
//    workgroups per CU cover the LDS write / barrier bubbles);
//  * sliding-window fragment reuse: a wave owns MPW consecutive output rows, so the activation fragment of
//    input row r serves output rows r, r-1, r-2 for the three vertical taps: (MPW+2)*3*2 instead of MPW*9*2
//    LDS reads per chunk;
//  * LDS-transposed epilogue: bias/activation in registers, the bf16 tile is written to LDS as [pixel][cout]
//    and read back row-wise, so every global store instruction writes 1 KiB of contiguous NHWC bytes; the same
//    image feeds the fused 2x2 max-pool output (unet_parts.py:212,233) and the fused 1-channel 1x1 + sigmoid
//    (outconv, Unet_singleFrame.py:207-209).
#include "common.h"

namespace {

struct PipeArgs {
  const void* src0;
  const void* src1;
  const void* prev0;
  const void* weight;
  const float* bias;
  const void* res;
  void* out;
  void* pool_out;
  const float* out1_w;
  const float* out1_b;
  float* out1;
  int N, H, W, Cin, Cout, pad, src_mode;
  int s0H, s0W, s0C, s1H, s1W, s1C, prev_ch;
  int act, res_b0;
  int Hout, Wout, oC;
  int pH, pW;  // pooled output extent (floor(Hout/2), floor(Wout/2))
  int tiles_x, tiles_y, n_ct, total_tiles, nk;
  int out1_act, skip_main;
};

struct Step {
  int n, y0, x0, cout0, kc;
};

template <typename T>
__device__ __forceinline__ typename Elem<T>::vec ldv(const void* base, size_t elem_off) {
  return *reinterpret_cast<const typename Elem<T>::vec*>(reinterpret_cast<const T*>(base) + elem_off);
}

__device__ __forceinline__ f32x16 mma_bf16(const bf16x8& a, const bf16x8& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

template <int NT, int MPW, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2) void conv3x3_pipe_kernel(const PipeArgs a) {
  using T = bf16_t;
  using E = Elem<T>;
  using vec = E::vec;
  constexpr int KC = E::KC, EPV = E::EPV;
  constexpr int NTHR = WAVES * 64;
  constexpr int TH = MPW * WAVES, TW = 32;
  constexpr int HH = TH + 2, HW = TW + 2;
  constexpr int NPIX = HH * HW;
  constexpr int CT = NT * 32;
  constexpr int XV = (NPIX * 4 + NTHR - 1) / NTHR;     // 16-byte vectors of the input tile per thread
  constexpr int WVN = (9 * CT * 4 + NTHR - 1) / NTHR;  // ... of the weight chunk per thread
  constexpr int SLOTS = CT / 8;                        // 16-byte slots per output pixel
  static_assert(TH * TW * CT * 2 <= NPIX * 64 + 9 * CT * 64, "epilogue image must fit in the staging LDS");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sX = smem;
  char* sW = smem + NPIX * 64;
  char* sO = smem;  // epilogue image [TH*TW][CT] bf16, reuses the staging area

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;

  const int my_tiles = (a.total_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  if (my_tiles <= 0) return;
  const int nsteps = my_tiles * a.nk;
  // weights stay in LDS across tiles when there is one K-chunk, one Cout tile, and the epilogue image does not
  // reach into sW
  const bool w_static = (a.nk == 1 && a.n_ct == 1) && (TH * TW * CT * 2 <= NPIX * 64);

  auto decode = [&](int s) {
    Step st;
    const int ti = (int)blockIdx.x + (s / a.nk) * (int)gridDim.x;
    st.kc = s - (s / a.nk) * a.nk;
    int r = ti;
    const int ct = r % a.n_ct; r /= a.n_ct;
    const int tx = r % a.tiles_x; r /= a.tiles_x;
    const int ty = r % a.tiles_y; r /= a.tiles_y;
    st.n = r;
    st.y0 = ty * TH;
    st.x0 = tx * TW;
    st.cout0 = ct * CT;
    return st;
  };

  vec xr[XV];
  vec wr[WVN];
  unsigned xvalid = 0;
  int g_pending = 0;

  // issue the global loads of one step into registers
  auto load_regs = [&](const Step& st, bool with_w) {
    const int c_log0 = st.kc * KC;
    int g = 0, cbase = c_log0;
    if (a.src_mode != UNCL_SRC_PLAIN) {
      g = c_log0 / a.s0C;
      cbase = c_log0 - g * a.s0C;
    }
    g_pending = g;
    xvalid = 0;
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int idx = tid + j * NTHR;
      xr[j] = E::zero();
      if (idx < NPIX * 4) {
        const int pix = idx >> 2, ch = idx & 3;
        const int hy = pix / HW, hx = pix - hy * HW;
        const int iy = st.y0 + hy - a.pad, ix = st.x0 + hx - a.pad;
        if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
          xvalid |= 1u << j;
          const int c = cbase + ch * EPV;
          if (g == 1) {
            const int dy = (a.s0H - a.s1H) >> 1, dx = (a.s0W - a.s1W) >> 1;
            const int sy = min(max(iy - dy, 0), a.s1H - 1), sx = min(max(ix - dx, 0), a.s1W - 1);
            xr[j] = ldv<T>(a.src1, (((size_t)st.n * a.s1H + sy) * a.s1W + sx) * a.s1C + c);
          } else {
            const size_t off = (((size_t)st.n * a.s0H + iy) * a.s0W + ix) * a.s0C + c;
            vec v = ldv<T>(a.src0, off);
            if (a.prev0 != nullptr && c < a.prev_ch) {
              const vec p = ldv<T>(a.prev0, off);
#pragma unroll
              for (int i = 0; i < EPV; ++i)
                if (c + i < a.prev_ch) v[i] = p[i];
            }
            xr[j] = v;
          }
        }
      }
    }
    if (with_w) {
#pragma unroll
      for (int j = 0; j < WVN; ++j) {
        const int idx = tid + j * NTHR;
        if (idx < 9 * CT * 4) {
          const int row = idx >> 2, ch = idx & 3;
          const int tap = row / CT, co = row - tap * CT;
          wr[j] = ldv<T>(a.weight, ((size_t)(tap * a.Cout + st.cout0 + co)) * a.Cin + st.kc * KC + ch * EPV);
        }
      }
    }
  };

  // registers -> LDS (applying the concat operator's x^2 / sqrt(x + 1e-8) on the way)
  auto write_lds = [&](bool with_w) {
#pragma unroll
    for (int j = 0; j < XV; ++j) {
      const int idx = tid + j * NTHR;
      if (idx < NPIX * 4) {
        const int pix = idx >> 2, ch = idx & 3;
        vec v = xr[j];
        if (a.src_mode == UNCL_SRC_CONCAT_SSR && g_pending >= 2 && ((xvalid >> j) & 1u)) {
          float f[EPV];
          E::unpack(v, f);
          if (g_pending == 2) {
#pragma unroll
            for (int i = 0; i < EPV; ++i) f[i] = f[i] * f[i];
          } else {
#pragma unroll
            for (int i = 0; i < EPV; ++i) f[i] = sqrtf(f[i] + 1e-8f);
          }
          v = E::pack(f);
        }
        *reinterpret_cast<vec*>(sX + pix * 64 + ((ch ^ ((pix >> 2) & 3)) << 4)) = v;
      }
    }
    if (with_w) {
#pragma unroll
      for (int j = 0; j < WVN; ++j) {
        const int idx = tid + j * NTHR;
        if (idx < 9 * CT * 4) {
          const int row = idx >> 2, ch = idx & 3;
          *reinterpret_cast<vec*>(sW + row * 64 + ((ch ^ ((row >> 2) & 3)) << 4)) = wr[j];
        }
      }
    }
  };

  f32x16 acc[MPW][NT];
#pragma unroll
  for (int m = 0; m < MPW; ++m)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][nt][i] = 0.f;

  Step cur = decode(0);
  load_regs(cur, true);
  write_lds(true);
  __syncthreads();

  for (int s = 0; s < nsteps; ++s) {
    const bool has_next = (s + 1 < nsteps);
    Step nxt = cur;
    if (has_next) {
      nxt = decode(s + 1);
      load_regs(nxt, !w_static);
    }
    // ---- MFMA phase over the staged chunk
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
            A[ty][nt] = *reinterpret_cast<const vec*>(sW + row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4));
          }
#pragma unroll
        for (int r = 0; r < MPW + 2; ++r) {
          const int pix = (wave * MPW + r) * HW + lr + tx;
          B[r] = *reinterpret_cast<const vec*>(sX + pix * 64 + ((chunk ^ ((pix >> 2) & 3)) << 4));
        }
#pragma unroll
        for (int m = 0; m < MPW; ++m)
#pragma unroll
          for (int ty = 0; ty < 3; ++ty)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[m][nt] = mma_bf16(A[ty][nt], B[m + ty], acc[m][nt]);
      }
    }
    // ---- tile finished: epilogue through LDS
    if (cur.kc == a.nk - 1) {
      __syncthreads();  // every wave is done reading sX / sW
#pragma unroll
      for (int m = 0; m < MPW; ++m) {
        const int prow = wave * MPW + m;
        const int pl = prow * TW + lr;  // pixel index inside the tile
        const int oy = cur.y0 + prow, ox = cur.x0 + lr;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int cl = nt * 32 + 8 * q + 4 * lh;  // channel inside the CT tile
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float t = acc[m][nt][4 * q + r];
              if (a.bias != nullptr) t += a.bias[cur.cout0 + cl + r];
              v[r] = uncl_act(t, a.act);
              acc[m][nt][4 * q + r] = 0.f;
            }
            if (a.res != nullptr && oy < a.Hout && ox < a.Wout) {
              const size_t rp = (a.res_b0 ? 0 : (size_t)cur.n * a.Hout * a.Wout) + (size_t)oy * a.Wout + ox;
              const bf16x4 rr = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const T*>(a.res) + rp * a.oC +
                                                                 cur.cout0 + cl);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += (float)rr[r];
            }
            bf16x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = (bf16_t)v[r];
            const int slot = (cl >> 3) ^ ((pl >> 1) & (SLOTS - 1));
            *reinterpret_cast<bf16x4*>(sO + pl * (CT * 2) + slot * 16 + (lh << 3)) = o;
          }
        }
      }
      __syncthreads();
      // main NHWC store: consecutive threads -> consecutive 16-byte slots -> 1 KiB per wave instruction
      if (!a.skip_main) {
        for (int v = tid; v < TH * TW * SLOTS; v += NTHR) {
          const int pl = v / SLOTS, sl = v - pl * SLOTS;
          const int prow = pl / TW, pcol = pl - prow * TW;
          const int oy = cur.y0 + prow, ox = cur.x0 + pcol;
          if (oy < a.Hout && ox < a.Wout) {
            const vec val = *reinterpret_cast<const vec*>(sO + pl * (CT * 2) + ((sl ^ ((pl >> 1) & (SLOTS - 1))) << 4));
            *reinterpret_cast<vec*>(reinterpret_cast<T*>(a.out) +
                                    (((size_t)cur.n * a.Hout + oy) * a.Wout + ox) * a.oC + cur.cout0 + sl * 8) = val;
          }
        }
      }
      // fused 2x2 max-pool copy for the next encoder stage
      if (a.pool_out != nullptr) {
        for (int v = tid; v < (TH / 2) * (TW / 2) * SLOTS; v += NTHR) {
          const int pp = v / SLOTS, sl = v - pp * SLOTS;
          const int py = pp / (TW / 2), px = pp - py * (TW / 2);
          const int gy = (cur.y0 >> 1) + py, gx = (cur.x0 >> 1) + px;
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
            *reinterpret_cast<vec*>(reinterpret_cast<T*>(a.pool_out) +
                                    (((size_t)cur.n * a.pH + gy) * a.pW + gx) * a.oC + cur.cout0 + sl * 8) = E::pack(m);
          }
        }
      }
      // fused trailing 1x1 -> one channel (+ sigmoid)
      if (a.out1_w != nullptr) {
        for (int pl = tid; pl < TH * TW; pl += NTHR) {
          const int prow = pl / TW, pcol = pl - prow * TW;
          const int oy = cur.y0 + prow, ox = cur.x0 + pcol;
          if (oy < a.Hout && ox < a.Wout) {
            float sum = a.out1_b[0];
#pragma unroll
            for (int sl = 0; sl < SLOTS; ++sl) {
              const vec val = *reinterpret_cast<const vec*>(sO + pl * (CT * 2) + ((sl ^ ((pl >> 1) & (SLOTS - 1))) << 4));
              float f[8];
              E::unpack(val, f);
#pragma unroll
              for (int i = 0; i < 8; ++i) sum = fmaf(f[i], a.out1_w[sl * 8 + i], sum);
            }
            a.out1[((size_t)cur.n * a.Hout + oy) * a.Wout + ox] = uncl_act(sum, a.out1_act);
          }
        }
      }
    }
    __syncthreads();
    if (has_next) {
      write_lds(!w_static);
      __syncthreads();
      cur = nxt;
    }
  }
}

template <int NT, int MPW, int WAVES>
int launch_pipe(const PipeArgs& a, hipStream_t s) {
  constexpr int TH = MPW * WAVES;
  constexpr size_t lds = (size_t)(TH + 2) * 34 * 64 + (size_t)9 * NT * 32 * 64;
  auto kern = conv3x3_pipe_kernel<NT, MPW, WAVES>;
  static bool attr_done = false;
  static int max_blocks = 0;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess)
      return UNCL_ERR_LAUNCH;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), WAVES * 64, lds) !=
            hipSuccess || per_cu <= 0)
      per_cu = 1;
    hipDeviceProp_t p;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return UNCL_ERR_LAUNCH;
    max_blocks = per_cu * p.multiProcessorCount;
    attr_done = true;
  }
  const int grid = a.total_tiles < max_blocks ? a.total_tiles : max_blocks;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, s, a);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

}  // namespace

// Same descriptor as uncl_conv_igemm; handles bf16 3x3 with src_mode PLAIN / CONCAT_SSR / CONCAT2.
// `pool_out` (optional) receives maxpool2x2(out) as NHWC (N, Hout/2, Wout/2, Cout).
extern "C" int uncl_conv3x3_pipe(const uncl_conv_desc* d, void* pool_out, void* stream) {
  if (d == nullptr || d->dtype != UNCL_BF16 || d->ksize != 3) return UNCL_ERR_ARG;
  if (d->pad != 0 && d->pad != 2) return UNCL_ERR_ARG;
  if (d->src_mode == UNCL_SRC_MAXPOOL2 || d->src_mode < 0 || d->src_mode > UNCL_SRC_CONCAT2) return UNCL_ERR_ARG;
  if (d->Cin <= 0 || d->Cin % 32 != 0 || d->Cout <= 0 || d->Cout % 32 != 0) return UNCL_ERR_ARG;
  if (d->z_mode != UNCL_Z_NONE || d->scale_n != nullptr) return UNCL_ERR_ARG;
  if (d->src0 == nullptr || d->weight == nullptr) return UNCL_ERR_ARG;
  if (d->out == nullptr && !(d->skip_main_store && d->out1 != nullptr)) return UNCL_ERR_ARG;
  if (d->src_mode != UNCL_SRC_PLAIN) {
    if (d->src1 == nullptr || d->src0_C != d->src1_C || d->src0_C % 32 != 0) return UNCL_ERR_ARG;
    const int groups = d->src_mode == UNCL_SRC_CONCAT_SSR ? 4 : 2;
    if (d->Cin != groups * d->src0_C) return UNCL_ERR_ARG;
    if (d->src1_H > d->src0_H || d->src1_W > d->src0_W) return UNCL_ERR_ARG;
  }
  if (d->out1_w != nullptr && (d->Cout != 32 || d->out1 == nullptr)) return UNCL_ERR_ARG;
  PipeArgs a;
  a.src0 = d->src0; a.src1 = d->src1; a.prev0 = d->prev0; a.weight = d->weight; a.bias = d->bias; a.res = d->res;
  a.out = d->out; a.pool_out = pool_out; a.out1_w = d->out1_w; a.out1_b = d->out1_b; a.out1 = d->out1;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.pad = d->pad; a.src_mode = d->src_mode;
  a.s0H = d->src0_H; a.s0W = d->src0_W; a.s0C = d->src0_C;
  a.s1H = d->src1_H; a.s1W = d->src1_W; a.s1C = d->src1_C; a.prev_ch = d->prev_ch;
  a.act = d->act; a.res_b0 = d->res_batch_stride0;
  a.Hout = d->H + 2 * d->pad - 2; a.Wout = d->W + 2 * d->pad - 2; a.oC = d->out_C;
  if (a.Hout <= 0 || a.Wout <= 0) return UNCL_ERR_ARG;
  if (d->out != nullptr && (d->out_H != a.Hout || d->out_W != a.Wout)) return UNCL_ERR_ARG;
  a.pH = a.Hout / 2; a.pW = a.Wout / 2;
  a.out1_act = d->out1_act; a.skip_main = d->skip_main_store;
  a.nk = d->Cin / 32;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (d->Cout == 32) {
    constexpr int TH = 16;
    a.n_ct = 1;
    a.tiles_x = (a.Wout + 31) / 32; a.tiles_y = (a.Hout + TH - 1) / TH;
    a.total_tiles = d->N * a.tiles_x * a.tiles_y;
    return launch_pipe<1, 4, 4>(a, s);
  }
  if (d->Cout % 64 != 0) return UNCL_ERR_ARG;
  constexpr int TH = 8;
  a.n_ct = d->Cout / 64;
  a.tiles_x = (a.Wout + 31) / 32; a.tiles_y = (a.Hout + TH - 1) / TH;
  a.total_tiles = d->N * a.tiles_x * a.tiles_y * a.n_ct;
  return launch_pipe<2, 2, 4>(a, s);
}
