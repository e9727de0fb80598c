// Bandwidth-bound helpers of the generator forward: weight re-layout, the 1-channel first layer, the graph
// block's kNN / max-relative gather, and the overlap-tile gather / cross-fade.  gfx950 only.
#include <cstdlib>

#include "common.h"

// ------------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------------
// Packed layouts.  fp32, 1x1, 2x2 and odd channel counts: [tap][Cout][Cin] (K contiguous).  3x3 in a 16-bit type with Cin a
// multiple of 32 (every layer of the MFMA 3x3 kernels): K-CHUNK-MAJOR, [Cin / 32][9][Cout][32] (uncl_w3_chunk_major, common.h) --
// the kernels stage one 32-channel K-chunk of all nine taps at a time, and in the tap-major layout that is 9 * Cout pieces of 64
// bytes, each half of a 128-byte line (the other half belongs to the next chunk: a CU's 32 KB L1 has long dropped it by then);
// chunk-major it is one contiguous run per tap (the whole chunk for a layer of one cout tile), whole lines, half the requests
// between L2 and L1 -- which is what the staging waves of the streamed-weight layers queue on (round 5).
// cout_order 1 (uncl_pack_item): output channel g C + c of a skip-concat layer's data-gradient weight -> 64 (c / 16) + 16 g + c % 16
__host__ __device__ __forceinline__ int uncl_ssr_cout_pos(int co, int Cout) {
  const int C = Cout >> 2, g = co / C, c = co - g * C;
  return 64 * (c >> 4) + 16 * g + (c & 15);
}
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ src, T* __restrict__ dst, int Cout, int Cin, int kk,
                                   int transposed, int flip) {
  const size_t total = (size_t)kk * Cout * Cin;
  const bool cm = uncl_w3_chunk_major(sizeof(T), kk, Cin);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin);
    const int co = (int)((i / Cin) % Cout);
    const int tap = (int)(i / ((size_t)Cin * Cout));
    const int ts = flip ? (kk - 1 - tap) : tap;
    const size_t s = transposed ? (((size_t)ci * Cout + co) * kk + ts) : (((size_t)co * Cin + ci) * kk + ts);
    dst[cm ? uncl_w3_index(tap, co, ci, Cout) : i] = (T)src[s];
  }
}

extern "C" int uncl_pack_conv_weight(const float* src, void* dst, int dtype, int Cout, int Cin, int k,
                                     int transposed, int flip, void* stream) {
  if (src == nullptr || dst == nullptr || Cout <= 0 || Cin <= 0 || k <= 0) return UNCL_ERR_ARG;
  const size_t total = (size_t)k * k * Cout * Cin;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == UNCL_BF16)
    hipLaunchKernelGGL(pack_weight_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, src, (bf16_t*)dst, Cout, Cin, k * k,
                       transposed, flip);
  else if (dtype == UNCL_F16)
    hipLaunchKernelGGL(pack_weight_kernel<f16_t>, dim3(blocks), dim3(256), 0, s, src, (f16_t*)dst, Cout, Cin, k * k,
                       transposed, flip);
  else if (dtype == UNCL_F32)
    hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(blocks), dim3(256), 0, s, src, (float*)dst, Cout, Cin, k * k,
                       transposed, flip);
  else
    return UNCL_ERR_ARG;
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

// every weight of a network in one launch (grid.y = item): a training step re-packs 50+ tensors after each optimiser step
struct PackBatch {
  uncl_pack_item it[UNCL_PACK_MAX_ITEMS];
};
// One workgroup = a 32 x 32 block of the two channel dimensions with all kk taps.  The source (reference layout
// [A][B][kk]: A = Cout, B = Cin for Conv2d, A = Cin, B = Cout for ConvTranspose2d) is read as 32 runs of 32*kk contiguous
// floats, the packed [tap][Cout][Cin] destination is written as runs of 32 contiguous channels; the transposition happens in
// LDS (rows padded to an odd pitch).  The element-per-thread form read the source with a stride of kk floats.
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_batch_kernel(const PackBatch t) {
  const uncl_pack_item& e = t.it[blockIdx.y];
  const int kk = e.k * e.k, Cin = e.Cin, Cout = e.Cout;
  const int A = e.transposed ? Cin : Cout, B = e.transposed ? Cout : Cin;
  const int tb = B >> 5, tiles = (A >> 5) * tb;
  if ((A | B) & 31) {       // odd channel counts: element-wise
    const size_t total = (size_t)kk * Cout * Cin;
    T* dst = reinterpret_cast<T*>(e.dst);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
      const int ci = (int)(i % Cin), co = (int)((i / Cin) % Cout), tap = (int)(i / ((size_t)Cin * Cout));
      const int ts = e.flip ? (kk - 1 - tap) : tap;
      const size_t s = e.transposed ? (((size_t)ci * Cout + co) * kk + ts) : (((size_t)co * Cin + ci) * kk + ts);
      dst[i] = (T)e.src[s];
    }
    return;
  }
  __shared__ float sm[32 * (32 * 9 + 1)];
  const int pitch = 32 * kk + 1;
  T* dst = reinterpret_cast<T*>(e.dst);
  const bool cm = uncl_w3_chunk_major(sizeof(T), kk, Cin);
  const bool perm = e.cout_order == 1;
  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int a0 = (tile / tb) << 5, b0 = (tile % tb) << 5;
    __syncthreads();
    for (int v = threadIdx.x; v < 32 * 32 * kk; v += 256) {
      const int a = v / (32 * kk), r = v - a * (32 * kk);
      sm[a * pitch + r] = e.src[((size_t)(a0 + a) * B + b0) * kk + r];
    }
    __syncthreads();
    // destination runs: (tap, co, 32 x ci).  lane = ci within the run
    for (int v = threadIdx.x; v < 32 * 32 * kk; v += 256) {
      const int ci_l = v & 31, co_l = (v >> 5) & 31, tap = v >> 10;
      const int ts = e.flip ? (kk - 1 - tap) : tap;
      const int a = e.transposed ? ci_l : co_l, bb = e.transposed ? co_l : ci_l;
      const int co = e.transposed ? b0 + co_l : a0 + co_l, ci = e.transposed ? a0 + ci_l : b0 + ci_l;
      const int cp = perm ? uncl_ssr_cout_pos(co, Cout) : co;
      dst[cm ? uncl_w3_index(tap, cp, ci, Cout) : ((size_t)tap * Cout + cp) * Cin + ci] = (T)sm[a * pitch + bb * kk + ts];
    }
  }
}

extern "C" int uncl_pack_conv_weights(const uncl_pack_item* items, int n_items, int dtype, void* stream) {
  if (n_items == 0) return UNCL_OK;
  if (!items || n_items < 0 || (!uncl_is_h16(dtype) && dtype != UNCL_F32)) return UNCL_ERR_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  for (int i0 = 0; i0 < n_items; i0 += UNCL_PACK_MAX_ITEMS) {
    const int n = n_items - i0 < UNCL_PACK_MAX_ITEMS ? n_items - i0 : UNCL_PACK_MAX_ITEMS;
    PackBatch t = {};
    for (int i = 0; i < n; ++i) {
      const uncl_pack_item& e = items[i0 + i];
      if (!e.src || !e.dst || e.Cout <= 0 || e.Cin <= 0 || e.k <= 0) return UNCL_ERR_ARG;
      // the interleaved cout order exists for the 3x3 16-bit data-gradient weights of the skip-concat layers only
      if (e.cout_order != 0 && (e.cout_order != 1 || !uncl_is_h16(dtype) || e.k != 3 || e.Cout % 64 != 0 || e.Cin % 32 != 0)) return UNCL_ERR_ARG;
      t.it[i] = e;
    }
    if (dtype == UNCL_BF16) hipLaunchKernelGGL(pack_weight_batch_kernel<bf16_t>, dim3(96, n), dim3(256), 0, s, t);
    else if (dtype == UNCL_F16) hipLaunchKernelGGL(pack_weight_batch_kernel<f16_t>, dim3(96, n), dim3(256), 0, s, t);
    else hipLaunchKernelGGL(pack_weight_batch_kernel<float>, dim3(96, n), dim3(256), 0, s, t);
  }
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

// ------------------------------------------------------------------------------------------------------
// first layer: Conv2d(1 -> Cout, 3x3 valid) + bias + act.  One thread = two vertically adjacent output pixels x 8 output
// channels: the 8 weights of a tap are two 16-byte LDS reads shared by both pixels (8 FMAs per LDS read instead of 1),
// 12 input loads serve 144 FMAs, and a wave's stores stay 1 KiB-contiguous (4 lanes = the 64 bytes of one NHWC pixel).
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv_in_c1_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ b, T* __restrict__ out, int N, int H,
                                                         int W, int Cout, int act) {
  extern __shared__ __attribute__((aligned(16))) float sw[];  // [9 taps][Cout] weights + Cout biases
  for (int i = threadIdx.x; i < Cout * 9; i += blockDim.x) sw[(i % 9) * Cout + i / 9] = w[i];
  for (int i = threadIdx.x; i < Cout; i += blockDim.x) sw[Cout * 9 + i] = b ? b[i] : 0.f;
  __syncthreads();
  const int Ho = H - 2, Wo = W - 2, G = Cout / 8, Hp = (Ho + 1) / 2;
  const size_t total = (size_t)N * Hp * Wo * G;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % G);
    const size_t p = i / G;
    const int ox = (int)(p % Wo);
    const int oy = 2 * (int)((p / Wo) % Hp);
    const int n = (int)(p / ((size_t)Wo * Hp));
    const bool two = oy + 1 < Ho;
    float in[4][3];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int yy = min(oy + r, H - 1);  // the fourth row only feeds the second pixel
#pragma unroll
      for (int c = 0; c < 3; ++c) in[r][c] = x[((size_t)n * H + yy) * W + ox + c];
    }
    float v0[8], v1[8];
    {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(sw + Cout * 9 + g * 8);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(sw + Cout * 9 + g * 8 + 4);
#pragma unroll
      for (int c = 0; c < 4; ++c) { v0[c] = v1[c] = b0[c]; v0[c + 4] = v1[c + 4] = b1[c]; }
    }
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(sw + t * Cout + g * 8);
      const f32x4 w1 = *reinterpret_cast<const f32x4*>(sw + t * Cout + g * 8 + 4);
      const float a0 = in[t / 3][t % 3], a1 = in[t / 3 + 1][t % 3];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        v0[c] = fmaf(a0, w0[c], v0[c]); v0[c + 4] = fmaf(a0, w1[c], v0[c + 4]);
        v1[c] = fmaf(a1, w0[c], v1[c]); v1[c + 4] = fmaf(a1, w1[c], v1[c + 4]);
      }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) { v0[c] = uncl_act(v0[c], act); v1[c] = uncl_act(v1[c], act); }
    T* o = out + (((size_t)n * Ho + oy) * Wo + ox) * Cout + g * 8;
    if constexpr (sizeof(T) == 2) {
      *reinterpret_cast<typename Elem<T>::vec*>(o) = Elem<T>::pack(v0);
      if (two) *reinterpret_cast<typename Elem<T>::vec*>(o + (size_t)Wo * Cout) = Elem<T>::pack(v1);
    } else {
      *reinterpret_cast<f32x4*>(o) = f32x4{v0[0], v0[1], v0[2], v0[3]};
      *reinterpret_cast<f32x4*>(o + 4) = f32x4{v0[4], v0[5], v0[6], v0[7]};
      if (two) {
        *reinterpret_cast<f32x4*>(o + (size_t)Wo * Cout) = f32x4{v1[0], v1[1], v1[2], v1[3]};
        *reinterpret_cast<f32x4*>(o + (size_t)Wo * Cout + 4) = f32x4{v1[4], v1[5], v1[6], v1[7]};
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// bf16 path of the first layer on the matrix cores.  K = 9 taps is far too short for an MFMA GEMM to matter as arithmetic,
// but the VALU form needs 288 multiply-adds per pixel (the kernel above is instruction-bound at ~2.6 TB/s of stores).  Here
// a 32-pixel x 32-channel block is two v_mfma_f32_32x32x16_bf16: the fp32 input taps are split into a bf16 head and a bf16
// tail (x = hi + lo to 2^-17), so the input keeps fp32-class precision while the weights are bf16 like every other layer's.
//   A (weights)  lane (cout = lane & 31, half = lane >> 5): k = 8*half + j  ->  tap k (zero for k >= 9)
//   B (patch)    lane (pixel = lane & 31, half):            k = 8*half + j  ->  input tap k of that pixel
// The result tile is transposed through a wave-private LDS scratch so that stores are 16 bytes per lane, 1 KiB contiguous.
// ------------------------------------------------------------------------------------------------------
constexpr int CM_TY = 8, CM_TX = 64;

__global__ __launch_bounds__(256) void conv_in_c1_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ b, bf16_t* __restrict__ out, int H,
                                                              int W, float slope, int tiles_x) {
  constexpr int PW = CM_TX + 2;
  __shared__ float sP[(CM_TY + 2) * PW];
  __shared__ __attribute__((aligned(16))) char sT[4][2048];   // per wave: [32 px][64 B], 16-byte slots XOR (px >> 1) & 3
  const int n = blockIdx.y;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int y0 = ty * CM_TY, x0 = tx * CM_TX;
  const int Ho = H - 2, Wo = W - 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 31, lh = lane >> 5;
  const float* xn = x + (size_t)n * H * W;
  for (int i = tid; i < (CM_TY + 2) * PW; i += 256) {
    const int r = i / PW, c = i - r * PW;
    sP[i] = xn[(size_t)min(y0 + r, H - 1) * W + min(x0 + c, W - 1)];
  }
  // weight fragment and this lane's biases (channels 8q + 4*lh + r)
  bf16x8 A;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * lh + j;
    A[j] = (bf16_t)(k < 9 ? w[lr * 9 + k] : 0.f);
  }
  float bv[16];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[4 * q + r] = b ? b[8 * q + 4 * lh + r] : 0.f;
  f32x16 zero16;
#pragma unroll
  for (int i = 0; i < 16; ++i) zero16[i] = 0.f;
  __syncthreads();
  char* st = sT[wave];
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const int r = wave * 2 + rr, px = cc * 32 + lr;
      bf16x8 Bh, Bl;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int t = 8 * lh + j;                       // tap of this k slot (compile-time per half after unrolling)
        float v = 0.f;
        if (lh == 0) v = sP[(r + j / 3) * PW + px + j % 3];
        else if (j == 0) v = sP[(r + 2) * PW + px + 2];
        (void)t;
        const bf16_t hi = (bf16_t)v;
        Bh[j] = hi;
        Bl[j] = (bf16_t)(v - (float)hi);
      }
      f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bh, zero16, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, Bl, acc, 0, 0, 0);
      // bias + activation, [pixel][channel] image in the wave's scratch
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = acc[4 * q + e] + bv[4 * q + e];
          o[e] = (bf16_t)(fmaxf(t, 0.f) + slope * fminf(t, 0.f));
        }
        *reinterpret_cast<bf16x4*>(st + lr * 64 + ((q ^ ((lr >> 1) & 3)) << 4) + (lh << 3)) = o;
      }
      // read back 16 bytes per lane in memory order and store (two passes of 16 pixels x 4 slots)
      const int oy = y0 + r;
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        const int p = ps * 16 + (lane >> 2), sl = lane & 3;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(st + p * 64 + ((sl ^ ((p >> 1) & 3)) << 4));
        const int ox = x0 + cc * 32 + p;
        if (oy < Ho && ox < Wo) *reinterpret_cast<bf16x8*>(out + (((size_t)n * Ho + oy) * Wo + ox) * 32 + sl * 8) = v;
      }
    }
  }
}

extern "C" int uncl_conv_in_c1(const float* x, const float* w, const float* b, void* out, int dtype, int N, int H,
                               int W, int Cout, int act, void* stream) {
  if (!x || !w || !out || N <= 0 || H < 3 || W < 3 || Cout % 8 != 0) return UNCL_ERR_ARG;
  const size_t total = (size_t)N * ((H - 1) / 2) * (W - 2) * (Cout / 8);
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  const size_t lds = (size_t)Cout * 10 * sizeof(float);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == UNCL_BF16 && Cout == 32 && (act == UNCL_ACT_RELU || act == UNCL_ACT_LRELU || act == UNCL_ACT_NONE)) {
    const int tx = (W - 2 + CM_TX - 1) / CM_TX, ty = (H - 2 + CM_TY - 1) / CM_TY;
    const float slope = act == UNCL_ACT_RELU ? 0.f : (act == UNCL_ACT_LRELU ? 0.2f : 1.f);
    hipLaunchKernelGGL(conv_in_c1_mfma_kernel, dim3(tx * ty, N), dim3(256), 0, s, x, w, b, (bf16_t*)out, H, W, slope, tx);
    UNCL_CHECK_LAUNCH();
    return UNCL_OK;
  }
  if (dtype == UNCL_BF16)
    hipLaunchKernelGGL(conv_in_c1_kernel<bf16_t>, dim3(blocks), dim3(256), lds, s, x, w, b, (bf16_t*)out, N, H, W, Cout,
                       act);
  else if (dtype == UNCL_F16)
    hipLaunchKernelGGL(conv_in_c1_kernel<f16_t>, dim3(blocks), dim3(256), lds, s, x, w, b, (f16_t*)out, N, H, W, Cout,
                       act);
  else if (dtype == UNCL_F32)
    hipLaunchKernelGGL(conv_in_c1_kernel<float>, dim3(blocks), dim3(256), lds, s, x, w, b, (float*)out, N, H, W, Cout,
                       act);
  else
    return UNCL_ERR_ARG;
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

// ------------------------------------------------------------------------------------------------------
// kNN graph on the bottleneck: one workgroup (16 waves) per sample, the whole normalised sample in LDS (fp32)
// ------------------------------------------------------------------------------------------------------
#define KNN_MAX_NODES 144
#define KNN_C 256
#define KNN_SPLIT 3        // workgroups per sample when there are fewer samples than CUs/3 (144 rows -> 48 each, 3 per wave)
#define KNN_LD (KNN_C + 4)  // padded row: consecutive rows shift one 16-byte slot -> conflict-free b128 reads

template <typename T>
__global__ __launch_bounds__(1024) void gcn_knn_kernel(const T* __restrict__ x, const float* __restrict__ rel,
                                                       int32_t* __restrict__ idx, float* __restrict__ dist_out, int n,
                                                       int k) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sx = reinterpret_cast<float*>(smem);  // [n][KNN_LD]
  float* ssq = sx + (size_t)n * KNN_LD;        // [n]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const T* xs = x + (size_t)blockIdx.x * n * KNN_C;
  // L2-normalise each node over channels (F.normalize: x / max(|x|, 1e-12)), keep |xn|^2; a wave's rows are all requested
  // before the first one is reduced (one memory latency per wave instead of one per row)
  constexpr int NR = (KNN_MAX_NODES + 15) / 16;
  float v[NR][4];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int i = min(wave + r * nw, n - 1);
#pragma unroll
    for (int t = 0; t < 4; ++t) v[r][t] = (float)xs[(size_t)i * KNN_C + lane * 4 + t];
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int i = wave + r * nw;
    if (i < n) {
      float ss = wave_sum(v[r][0] * v[r][0] + v[r][1] * v[r][1] + v[r][2] * v[r][2] + v[r][3] * v[r][3]);
      const float inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
#pragma unroll
      for (int t = 0; t < 4; ++t) v[r][t] *= inv;
      *reinterpret_cast<f32x4*>(sx + (size_t)i * KNN_LD + lane * 4) = f32x4{v[r][0], v[r][1], v[r][2], v[r][3]};
      float s2 = wave_sum(v[r][0] * v[r][0] + v[r][1] * v[r][1] + v[r][2] * v[r][2] + v[r][3] * v[r][3]);
      if (lane == 0) ssq[i] = s2;
    }
  }
  __syncthreads();
  // Rows are dealt round-robin to the gridDim.y workgroups of this sample (each re-normalises the sample: cheap) and, inside
  // a workgroup, to its waves; a wave walks its rows THREE at a time: every 16-byte read of a key row x_j then feeds twelve
  // FMAs (three query rows, read as same-address broadcasts) instead of four, which moves the loop from the LDS pipe to the
  // VALU.  Each (i, j) dot product is still one sequential fmaf chain over the channels: bit-identical to the one-row form.
  constexpr int RB = 3;
  const int split = gridDim.y, stride = split * nw;
  for (int ib = blockIdx.y + split * wave; ib < n; ib += RB * stride) {
    const float* xi[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) xi[rb] = sx + (size_t)min(ib + rb * stride, n - 1) * KNN_LD;
    const float* xj[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) xj[t] = sx + (size_t)min(lane + 64 * t, n - 1) * KNN_LD;
    float dot[RB][3];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int t = 0; t < 3; ++t) dot[rb][t] = 0.f;
#pragma unroll 2
    for (int c = 0; c < KNN_C; c += 4) {
      f32x4 a[RB], b[3];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) a[rb] = *reinterpret_cast<const f32x4*>(xi[rb] + c);
#pragma unroll
      for (int t = 0; t < 3; ++t) b[t] = *reinterpret_cast<const f32x4*>(xj[t] + c);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          float dd = dot[rb][t];
          dd = fmaf(a[rb][0], b[t][0], dd);
          dd = fmaf(a[rb][1], b[t][1], dd);
          dd = fmaf(a[rb][2], b[t][2], dd);
          dd = fmaf(a[rb][3], b[t][3], dd);
          dot[rb][t] = dd;
        }
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int i = ib + rb * stride;
      if (i >= n) break;     // wave-uniform
      float d[3];
      int jj[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int j = lane + 64 * t;
        jj[t] = j;
        d[t] = INFINITY;
        if (j < n) {
          // same association as the reference: (|xi|^2 + (-2 xi.xj)) + |xj|^2, then + relative_pos
          float dd = (ssq[i] + (-2.f * dot[rb][t])) + ssq[j];
          if (rel != nullptr) dd += rel[(size_t)i * n + j];
          d[t] = dd;
          if (dist_out != nullptr) dist_out[((size_t)blockIdx.x * n + i) * n + j] = dd;
        }
      }
      // k rounds of wave-wide arg-min; ties go to the lower node index
      for (int r = 0; r < k; ++r) {
        float bv = d[0];
        int bj = jj[0];
#pragma unroll
        for (int t = 1; t < 3; ++t)
          if (d[t] < bv) { bv = d[t]; bj = jj[t]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const float ov = __shfl_xor(bv, o, 64);
          const int oj = __shfl_xor(bj, o, 64);
          if (ov < bv || (ov == bv && oj < bj)) { bv = ov; bj = oj; }
        }
#pragma unroll
        for (int t = 0; t < 3; ++t)
          if (jj[t] == bj) d[t] = INFINITY;
        if (lane == 0) idx[((size_t)blockIdx.x * n + i) * k + r] = bj;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// The same graph for 16-bit features at inference size, on the matrix cores: the 144 x 144 Gram matrix of the RAW bf16 / f16
// rows is five 32 x 32 row tiles x five column tiles of v_mfma_f32_32x32x16 (exact products, fp32 accumulation), scaled to
// the normalised inner products by 1/|x_i| 1/|x_j| afterwards; the top-9 of a row is selected straight from the accumulator
// registers (a row's 160 candidates are ONE register slot of the 32 lanes of a half-wave x 5 column tiles).  The VALU kernel
// above spends 150 us per 200 samples on 7 M sequential fmaf per sample.  Distances agree with it to fp32
// rounding (different summation order), so an index can differ only where two candidates are within ~1e-6: the fp32 parity
// path keeps the VALU kernel, whose indices are pinned bit for bit.
// ------------------------------------------------------------------------------------------------------
#define KNM_ROW 528            // bytes per staged row: 512 + 16 (consecutive rows shift one 16-byte slot)

template <typename T>
__global__ __launch_bounds__(512, 2) void gcn_knn_mfma_kernel(const T* __restrict__ x, const float* __restrict__ rel,
                                                             int32_t* __restrict__ idx, int n) {
  using vec = typename Elem<T>::vec;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sF = smem;                                                    // [n][KNM_ROW]
  float* sInv = reinterpret_cast<float*>(smem + (size_t)KNN_MAX_NODES * KNM_ROW);   // [160] 1/|x_i|
  float* sXX = sInv + 160;                                            // [160] |xn_i|^2
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const T* xs = x + (size_t)blockIdx.x * n * KNN_C;
  // stage the sample: every load of a thread requested before its first LDS write (nine 16-byte vectors)
  {
    constexpr int VPT = (KNN_MAX_NODES * (KNN_C / 8) + 511) / 512;
    vec tmp[VPT];
#pragma unroll
    for (int q = 0; q < VPT; ++q) {
      const int v = min(tid + q * 512, n * (KNN_C / 8) - 1);
      tmp[q] = *reinterpret_cast<const vec*>(xs + (size_t)(v >> 5) * KNN_C + (v & 31) * 8);
    }
#pragma unroll
    for (int q = 0; q < VPT; ++q) {
      const int v = tid + q * 512;
      if (v < n * (KNN_C / 8)) *reinterpret_cast<vec*>(sF + (v >> 5) * KNM_ROW + (v & 31) * 16) = tmp[q];
    }
  }
  __syncthreads();
  // norms, one thread per row from the staged copy (rows are one 16-byte slot apart mod 256 bytes: conflict-free):
  // F.normalize = x / max(|x|, 1e-12); |xn|^2 is the sum over the normalised values, as in the reference
  if (tid < 160) {
    float inv = 0.f, s2 = 0.f;
    if (tid < n) {
      const char* rp = sF + tid * KNM_ROW;
      float ss = 0.f;
      for (int sl = 0; sl < KNN_C / 8; ++sl) {
        float f[8];
        Elem<T>::unpack(*reinterpret_cast<const vec*>(rp + sl * 16), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) ss = fmaf(f[e], f[e], ss);
      }
      inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
      for (int sl = 0; sl < KNN_C / 8; ++sl) {
        float f[8];
        Elem<T>::unpack(*reinterpret_cast<const vec*>(rp + sl * 16), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float t = f[e] * inv; s2 = fmaf(t, t, s2); }
      }
    }
    sInv[tid] = inv;
    sXX[tid] = s2;
  }
  __syncthreads();
  // The Gram matrix is symmetric, so the registers of a LANE are already a row's candidates: with the row tile as the A operand
  // lane lr of column tile ct holds D[i][j] for its node j = 32 ct + lr and the 16 nodes i = 32 rt + 8q + 4 lh + r of every
  // row tile -- half of row j's 160 candidates per half-wave.  Each lane keeps a sorted top-9 of its 80 candidates (they come
  // in ascending i, so a strict compare keeps the lower index on ties), the two half-waves merge their lists.  One wave per
  // column tile; nine rounds of wave-wide arg-min per row were 46 of this kernel's 75 us.
  if (wave < 5) {
    const int ct = wave, j = ct * 32 + lr;
    const bool j_ok = j < n;
    const int jc = min(j, n - 1);
    vec Bf[KNN_C / 16];
    {
      const char* pb = sF + jc * KNM_ROW + lh * 16;
#pragma unroll
      for (int ks = 0; ks < KNN_C / 16; ++ks) Bf[ks] = *reinterpret_cast<const vec*>(pb + ks * 32);
    }
    const float inv_j = sInv[min(j, 159)], xx_j = sXX[min(j, 159)];
    float tv[9];
    int ti[9];
#pragma unroll
    for (int p = 0; p < 9; ++p) { tv[p] = INFINITY; ti[p] = 0x7fffffff; }
#pragma unroll 1
    for (int rt = 0; rt < 5; ++rt) {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      const char* pa = sF + min(rt * 32 + lr, n - 1) * KNM_ROW + lh * 16;
#pragma unroll
      for (int ks = 0; ks < KNN_C / 16; ++ks) acc = mfma32x16(*reinterpret_cast<const vec*>(pa + ks * 32), Bf[ks], acc);
      float rl[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = min(rt * 32 + 8 * (e >> 2) + 4 * lh + (e & 3), n - 1);
        rl[e] = rel != nullptr ? rel[(size_t)jc * n + i] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int i = rt * 32 + 8 * (e >> 2) + 4 * lh + (e & 3);
        // same association as the reference: (|x_row|^2 + (-2 x_row.x_col)) + |x_col|^2, then + relative_pos[row][col]; row = j
        float dd = (xx_j + (-2.f * (acc[e] * inv_j * sInv[min(i, 159)]))) + sXX[min(i, 159)];
        dd += rl[e];
        if (i >= n) dd = INFINITY;
        // insert (dd, i) into the sorted list; strict compare: an equal earlier (lower-index) entry stays in front
        bool c[9];
#pragma unroll
        for (int p = 0; p < 9; ++p) c[p] = dd < tv[p];
#pragma unroll
        for (int p = 8; p > 0; --p) {
          tv[p] = c[p] ? (c[p - 1] ? tv[p - 1] : dd) : tv[p];
          ti[p] = c[p] ? (c[p - 1] ? ti[p - 1] : i) : ti[p];
        }
        tv[0] = c[0] ? dd : tv[0];
        ti[0] = c[0] ? i : ti[0];
      }
    }
    // merge the other half-wave's list (same node, the other half of the candidates): its entries are inserted with the full
    // (value, index) order
    float ov[9];
    int oi[9];
#pragma unroll
    for (int p = 0; p < 9; ++p) { ov[p] = __shfl_xor(tv[p], 32, 64); oi[p] = __shfl_xor(ti[p], 32, 64); }
#pragma unroll
    for (int s2 = 0; s2 < 9; ++s2) {
      const float dd = ov[s2];
      const int i = oi[s2];
      bool c[9];
#pragma unroll
      for (int p = 0; p < 9; ++p) c[p] = dd < tv[p] || (dd == tv[p] && i < ti[p]);
#pragma unroll
      for (int p = 8; p > 0; --p) {
        tv[p] = c[p] ? (c[p - 1] ? tv[p - 1] : dd) : tv[p];
        ti[p] = c[p] ? (c[p - 1] ? ti[p - 1] : i) : ti[p];
      }
      tv[0] = c[0] ? dd : tv[0];
      ti[0] = c[0] ? i : ti[0];
    }
    // A row with NaN / Inf features has fewer than nine candidates that compare below +inf, and its unfilled slots still hold
    // the sentinel: the consumers gather rows by these indices, so a slot that is not a node becomes node p (any valid node;
    // torch.topk gives no meaningful neighbours for such a row either, but it never hands out an index outside the graph)
    if (lh == 0 && j_ok) {
#pragma unroll
      for (int p = 0; p < 9; ++p) idx[((size_t)blockIdx.x * n + j) * 9 + p] = (unsigned)ti[p] < (unsigned)n ? ti[p] : min(p, n - 1);
    }
  }
}

// 16-bit features: 1 (default) the matrix-core kernel, 0 the VALU kernel (A/B runs, and the parity tests that compare the two)
static std::atomic<int> g_knn_mfma{[] { const char* e = getenv("UNCL_KNN_MFMA"); return e ? atoi(e) : 1; }()};
extern "C" int uncl_gcn_set_knn_mfma(int on) { return g_knn_mfma.exchange(on ? 1 : 0); }

extern "C" size_t uncl_gcn_knn_workspace_bytes(int N, int n, int C) {
  (void)N; (void)n; (void)C;
  return 0;  // the sample lives in LDS
}

extern "C" int uncl_gcn_knn(const void* x, int dtype, const float* relative_pos, int32_t* idx, float* dist_out, int N,
                            int n, int C, int k, void* workspace, void* stream) {
  (void)workspace;
  if (!x || !idx || N <= 0 || n <= 0 || n > KNN_MAX_NODES || C != KNN_C || k <= 0 || k > n) return UNCL_ERR_ARG;
  const size_t lds = ((size_t)n * KNN_LD + n) * sizeof(float);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // 16-bit features: the matrix-core kernel, one workgroup per sample whatever the batch (the choice must not depend on how a
  // batch is cut into launches: the two kernels may order near-ties differently)
  if (g_knn_mfma.load(std::memory_order_relaxed) && uncl_is_h16(dtype) && dist_out == nullptr && k == 9 && n > 32) {
    const size_t l2 = (size_t)KNN_MAX_NODES * KNM_ROW + 2 * 160 * sizeof(float);
    static UnclDevOnce attr2[2];
    if (dtype == UNCL_F16) {
      if (attr2[0].need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gcn_knn_mfma_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
        attr2[0].done();
      }
      hipLaunchKernelGGL(gcn_knn_mfma_kernel<f16_t>, dim3(N), dim3(512), l2, s, (const f16_t*)x, relative_pos, idx, n);
    } else {
      if (attr2[1].need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gcn_knn_mfma_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
        attr2[1].done();
      }
      hipLaunchKernelGGL(gcn_knn_mfma_kernel<bf16_t>, dim3(N), dim3(512), l2, s, (const bf16_t*)x, relative_pos, idx, n);
    }
    UNCL_CHECK_LAUNCH();
    return UNCL_OK;
  }
  // one workgroup per CU (the sample fills the LDS): with a CU per sample to spare the rows are not split
  const int split = N * KNN_SPLIT <= 256 ? KNN_SPLIT : 1;
  static UnclDevOnce attr[3];
  if (dtype == UNCL_F16) {
    if (attr[2].need()) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gcn_knn_kernel<f16_t>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)(((size_t)KNN_MAX_NODES * KNN_LD + KNN_MAX_NODES) * 4));
      attr[2].done();
    }
    hipLaunchKernelGGL(gcn_knn_kernel<f16_t>, dim3(N, split), dim3(1024), lds, s, (const f16_t*)x, relative_pos, idx, dist_out,
                       n, k);
  } else if (dtype == UNCL_BF16) {
    if (attr[1].need()) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gcn_knn_kernel<bf16_t>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)(((size_t)KNN_MAX_NODES * KNN_LD + KNN_MAX_NODES) * 4));
      attr[1].done();
    }
    hipLaunchKernelGGL(gcn_knn_kernel<bf16_t>, dim3(N, split), dim3(1024), lds, s, (const bf16_t*)x, relative_pos, idx, dist_out,
                       n, k);
  } else if (dtype == UNCL_F32) {
    if (attr[0].need()) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gcn_knn_kernel<float>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)(((size_t)KNN_MAX_NODES * KNN_LD + KNN_MAX_NODES) * 4));
      attr[0].done();
    }
    hipLaunchKernelGGL(gcn_knn_kernel<float>, dim3(N, split), dim3(1024), lds, s, (const float*)x, relative_pos, idx, dist_out, n,
                       k);
  } else {
    return UNCL_ERR_ARG;
  }
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

// max-relative features, channels interleaved [x_c, max_k(x_c[nbr] - x_c)]
template <typename T>
__global__ __launch_bounds__(256) void gcn_maxrel_kernel(const T* __restrict__ x, const int32_t* __restrict__ idx,
                                                         T* __restrict__ out, int N, int n, int C, int k) {
  using E = Elem<T>;
  using vec = typename E::vec;
  constexpr int EPV = E::EPV;
  const int VC = C / EPV;
  const size_t total = (size_t)N * n * VC;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const int cv = (int)(t % VC);
    const size_t node = t / VC;  // b*n + i
    const size_t b = node / n;
    float xi[EPV], m[EPV];
    E::unpack(*reinterpret_cast<const vec*>(x + node * C + cv * EPV), xi);
#pragma unroll
    for (int e = 0; e < EPV; ++e) m[e] = -INFINITY;
    for (int r = 0; r < k; ++r) {
      const int j = idx[node * k + r];
      float xj[EPV];
      E::unpack(*reinterpret_cast<const vec*>(x + (b * n + j) * C + cv * EPV), xj);
#pragma unroll
      for (int e = 0; e < EPV; ++e) m[e] = fmaxf(m[e], xj[e] - xi[e]);
    }
    float o[2 * EPV];
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      o[2 * e] = xi[e];
      o[2 * e + 1] = m[e];
    }
    T* op = out + node * 2 * C + (size_t)cv * 2 * EPV;
    *reinterpret_cast<vec*>(op) = E::pack(o);
    *reinterpret_cast<vec*>(op + EPV) = E::pack(o + EPV);
  }
}

extern "C" int uncl_gcn_maxrel(const void* x, const int32_t* idx, void* out, int dtype, int N, int n, int C, int k,
                               void* stream) {
  if (!x || !idx || !out || N <= 0 || n <= 0 || C % 8 != 0 || k <= 0) return UNCL_ERR_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == UNCL_F16) {
    const size_t total = (size_t)N * n * (C / 8);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(gcn_maxrel_kernel<f16_t>, dim3(blocks), dim3(256), 0, s, (const f16_t*)x, idx, (f16_t*)out, N, n, C, k);
  } else if (dtype == UNCL_BF16) {
    const size_t total = (size_t)N * n * (C / 8);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(gcn_maxrel_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, idx, (bf16_t*)out, N, n,
                       C, k);
  } else if (dtype == UNCL_F32) {
    const size_t total = (size_t)N * n * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(gcn_maxrel_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)x, idx, (float*)out, N, n, C,
                       k);
  } else {
    return UNCL_ERR_ARG;
  }
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

// ------------------------------------------------------------------------------------------------------
// overlap tiler: 256^2 tiles, stride 192, sequential linear cross-fade along x then along y
// ------------------------------------------------------------------------------------------------------
#define TILE 256
#define TILE_OVERLAP 64
#define TILE_MAX_AXIS 48

struct AxisPlan {
  int count;
  int start[TILE_MAX_AXIS];
  int blend[TILE_MAX_AXIS];
};

static int make_axis_plan(int L, AxisPlan* p) {
  if (L <= TILE) return UNCL_ERR_ARG;  // the reference's loop is undefined there (model_save_util.py:417-441)
  int c = 0;
  for (int idx = 1; TILE * idx - TILE_OVERLAP * (idx - 1) < L; ++idx) {
    if (c >= TILE_MAX_AXIS - 1) return UNCL_ERR_ARG;
    p->start[c] = (TILE - TILE_OVERLAP) * (idx - 1);
    p->blend[c] = idx == 1 ? 0 : TILE_OVERLAP;
    ++c;
  }
  const int end_last = p->start[c - 1] + TILE;
  p->start[c] = L - TILE;
  p->blend[c] = end_last - (L - TILE);
  p->count = c + 1;
  return UNCL_OK;
}

extern "C" int uncl_tile_count(int H, int W) {
  AxisPlan py, px;
  if (make_axis_plan(H, &py) != UNCL_OK || make_axis_plan(W, &px) != UNCL_OK) return UNCL_ERR_ARG;
  return py.count * px.count;
}

__global__ __launch_bounds__(256) void tile_gather_kernel(const float* __restrict__ frames, float* __restrict__ tiles,
                                                          int H, int W, AxisPlan py, AxisPlan px) {
  // grid: (tile rows of 4 px-quads, tiles per frame, frames); four pixels per thread, as one 16-byte access where the tile's
  // column origin and the row pitch allow it
  const int t = blockIdx.y, f = blockIdx.z;
  const int ty = t / px.count, tx = t - ty * px.count;
  const int y0 = py.start[ty], x0 = px.start[tx];
  const float* src = frames + (size_t)f * H * W;
  float* dst = tiles + ((size_t)f * gridDim.y + t) * TILE * TILE;
  const bool al = ((x0 | W) & 3) == 0;
  for (int i = (blockIdx.x * blockDim.x + threadIdx.x) * 4; i < TILE * TILE; i += gridDim.x * blockDim.x * 4) {
    const int y = i >> 8, x = i & 255;
    const float* sp = src + (size_t)(y0 + y) * W + x0 + x;
    f32x4 v;
    if (al) {
      v = *reinterpret_cast<const f32x4*>(sp);
    } else {
      v = f32x4{sp[0], sp[1], sp[2], sp[3]};
    }
    *reinterpret_cast<f32x4*>(dst + i) = v;
  }
}

__device__ __forceinline__ float fold1(float acc, float piece, int i, int blend, bool first) {
  if (first || i >= blend) return piece;
  // same fp32 association as the reference: acc*(b-1-i)/(b-1) + piece*i/(b-1)
  const float bm1 = (float)(blend - 1);
  return (acc * (float)(blend - 1 - i)) / bm1 + (piece * (float)i) / bm1;
}

__global__ __launch_bounds__(256) void tile_blend_kernel(const float* __restrict__ tiles, float* __restrict__ frames,
                                                         int H, int W, AxisPlan py, AxisPlan px) {
  // four consecutive pixels of a row per thread (W % 4 == 0: the host falls back to QUAD = 1 otherwise); a tile's row is read
  // as one 16-byte vector where the quad lies inside the tile and the tile's column origin is a multiple of four
  const int f = blockIdx.y;
  const int T = py.count * px.count;
  const float* tf = tiles + (size_t)f * T * TILE * TILE;
  float* dst = frames + (size_t)f * H * W;
  const bool quad = (W & 3) == 0;
  const int step = quad ? 4 : 1;
  for (int i = (blockIdx.x * blockDim.x + threadIdx.x) * step; i < H * W; i += gridDim.x * blockDim.x * step) {
    const int y = i / W, x = i - y * W;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int sy = 0; sy < py.count; ++sy) {
      const int ly = y - py.start[sy];
      if (ly < 0 || ly >= TILE) continue;
      float strip[4] = {0.f, 0.f, 0.f, 0.f};
      for (int sx = 0; sx < px.count; ++sx) {
        const int lx = x - px.start[sx];
        if (lx + step - 1 < 0 || lx >= TILE) continue;
        const float* tp = tf + ((size_t)(sy * px.count + sx) * TILE + ly) * TILE;
        if (quad && lx >= 0 && lx + 3 < TILE && (px.start[sx] & 3) == 0) {
          const f32x4 p4 = *reinterpret_cast<const f32x4*>(tp + lx);
#pragma unroll
          for (int e = 0; e < 4; ++e) strip[e] = fold1(strip[e], p4[e], lx + e, px.blend[sx], sx == 0);
        } else {
          for (int e = 0; e < step; ++e) {
            const int le = lx + e;
            if (le < 0 || le >= TILE) continue;
            strip[e] = fold1(strip[e], tp[le], le, px.blend[sx], sx == 0);
          }
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = fold1(acc[e], strip[e], ly, py.blend[sy], sy == 0);
    }
    if (quad) {
      *reinterpret_cast<f32x4*>(dst + i) = f32x4{acc[0], acc[1], acc[2], acc[3]};
    } else {
      dst[i] = acc[0];
    }
  }
}

extern "C" int uncl_tile_gather(const float* frames, float* tiles, int F, int H, int W, void* stream) {
  AxisPlan py, px;
  if (!frames || !tiles || F <= 0) return UNCL_ERR_ARG;
  if (make_axis_plan(H, &py) != UNCL_OK || make_axis_plan(W, &px) != UNCL_OK) return UNCL_ERR_ARG;
  hipLaunchKernelGGL(tile_gather_kernel, dim3(32, py.count * px.count, F), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), frames, tiles, H, W, py, px);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

// offsets_host[f * T + t] = element offset of tile t of frame f (uncl_tile_gather's order: rows of tiles, frames outermost) inside
// the stack of F frames of H x W pixels: what uncl_gen_run.x_tile_off / uncl_conv_desc.src1 (UNCL_SRC_IMAGE1) take once it is on
// the device.  HOST array of F * uncl_tile_count(H, W) ints; no GPU work.
extern "C" int uncl_tile_offsets(int F, int H, int W, int32_t* offsets_host) {
  AxisPlan py, px;
  if (!offsets_host || F <= 0) return UNCL_ERR_ARG;
  if (make_axis_plan(H, &py) != UNCL_OK || make_axis_plan(W, &px) != UNCL_OK) return UNCL_ERR_ARG;
  if ((long long)F * H * W >= (1ll << 31)) return UNCL_ERR_ARG;
  const int T = py.count * px.count;
  for (int f = 0; f < F; ++f)
    for (int t = 0; t < T; ++t) {
      const int ty = t / px.count, tx = t - ty * px.count;
      offsets_host[f * T + t] = (int32_t)(((long long)f * H + py.start[ty]) * W + px.start[tx]);
    }
  return UNCL_OK;
}

extern "C" int uncl_tile_blend(const float* tiles, float* frames, int F, int H, int W, void* stream) {
  AxisPlan py, px;
  if (!frames || !tiles || F <= 0) return UNCL_ERR_ARG;
  if (make_axis_plan(H, &py) != UNCL_OK || make_axis_plan(W, &px) != UNCL_OK) return UNCL_ERR_ARG;
  const int work = (W & 3) == 0 ? H * W / 4 : H * W;
  const int blocks = (work + 255) / 256 < 4096 ? (work + 255) / 256 : 4096;
  hipLaunchKernelGGL(tile_blend_kernel, dim3(blocks, F), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), tiles,
                     frames, H, W, py, px);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

extern "C" int uncl_version(void) { return 1; }

extern "C" int uncl_device_ok(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 0;
  const char* a = p.gcnArchName;
  return (a[0] == 'g' && a[1] == 'f' && a[2] == 'x' && a[3] == '9' && a[4] == '5' && a[5] == '0') ? 1 : 0;
}


// ---- checked build (common.h: UNCL_CHECKED): the violation record and its C-ABI reader -------------------------------------------
unsigned long long* uncl_chk_record() {
#ifdef UNCL_CHECKED
  static std::atomic<unsigned long long*> rec[32];
  const int d = uncl_device();
  unsigned long long* p = rec[d].load(std::memory_order_acquire);
  if (p == nullptr) {
    unsigned long long* q = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&q), 4 * sizeof(unsigned long long)) != hipSuccess) return nullptr;
    (void)hipMemset(q, 0, 4 * sizeof(unsigned long long));
    unsigned long long* expect = nullptr;
    if (rec[d].compare_exchange_strong(expect, q)) p = q;
    else { (void)hipFree(q); p = expect; }
  }
  return p;
#else
  return nullptr;
#endif
}

// out4 = {violations, first faulting address, source line of the access, bytes}; reset != 0 clears the record afterwards.
// UNCL_ERR_ARG in the product build (nothing is checked there).  Synchronises the device.
extern "C" int uncl_checked_report(unsigned long long* out4, int reset) {
#ifdef UNCL_CHECKED
  unsigned long long* p = uncl_chk_record();
  if (p == nullptr || out4 == nullptr) return UNCL_ERR_ARG;
  if (hipDeviceSynchronize() != hipSuccess) return UNCL_ERR_LAUNCH;
  if (hipMemcpy(out4, p, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return UNCL_ERR_LAUNCH;
  if (reset && hipMemset(p, 0, 4 * sizeof(unsigned long long)) != hipSuccess) return UNCL_ERR_LAUNCH;
  return UNCL_OK;
#else
  (void)out4; (void)reset;
  return UNCL_ERR_ARG;
#endif
}
