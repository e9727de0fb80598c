// The tail of the graph block as ONE launch (inference, 16-bit): max-relative gather, the grouped 1x1 conv + GELU, fc2 +
// residual (Grapher, gcn_lib/torch_vertex.py:181-227) and the FFN (fc1 + GELU, fc2 + residual; Unet_singleFrame.py:20-41) of
// one sample per workgroup, with every intermediate (max-relative features, the grouped conv's output, the Grapher's output,
// the FFN's hidden layer) in LDS.  As separate launches these were a gather kernel and four 1x1 convolutions of 3.8 - 7.5
// GFLOP each: ~40 us apiece for a microsecond of arithmetic, because each is one workgroup's load -> multiply -> store
// latency chain plus a launch, with the 15 MB intermediates bouncing through L2 / HBM in between.
//
// Layout: lane = node (B operand of v_mfma_f32_32x32x16), weights are the A operand read straight from global memory
// (each wave owns one 32-channel output tile, so a weight row is read by exactly one wave of the workgroup), activations are
// B fragments from LDS rows padded by 16 bytes (consecutive rows shift one 16-byte slot: conflict-free b128 reads).  The
// rounding points are those of the separate kernels (every intermediate is rounded to the 16-bit type where they stored it).
#include "common.h"

namespace {

constexpr int GB_NODES = 144, GB_C = 256, GB_K = 9;
constexpr int GB_ROW = 2 * GB_C + 16;      // bytes per 256-channel row in LDS
constexpr int GB_ROWG = 2 * 128 + 16;      // bytes per 128-channel row (one group of the grouped conv)

struct GbArgs {
  const void* F;         // (N,144,256) fc1 output (the kNN kernel read the same tensor)
  const int32_t* idx;    // (N,144,9)
  const void* X4;        // (N,144,256) input of the Grapher (its residual)
  const void *wg, *w2, *w3, *w4;     // packed [4][128][128], [256][512], [256][256], [256][256]
  const float *bg, *b2, *b3, *b4;    // [512], [256], [256], [256] or NULL
  void* out;             // (N,144,256)
  int N;
  // FULL (the whole block in one launch): fc1 and the kNN graph are computed here too
  const void* w1;        // packed [256][256]
  const float* b1;       // [256] or NULL
  const float* rel;      // relative_pos (144,144) or NULL
  int32_t* idx_out;      // (N,144,9): the graph, also written to global memory (uncl_gen_forward's knn_idx output)
};

// GELU with erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the 16-bit rounding of the value it feeds):
// one reciprocal, one exponential and a five-term Horner chain instead of erff's ~40 instructions -- the block's 110 k GELUs
// per sample were the largest VALU item of this kernel
__device__ __forceinline__ float gelu_as(float x) {
  const float z = x * 0.70710678118654752440f, az = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.f));
  float p = fmaf(t, 1.061405429f, -1.453152027f);
  p = fmaf(t, p, 1.421413741f);
  p = fmaf(t, p, -0.284496736f);
  p = fmaf(t, p, 0.254829592f);
  p *= t;
  const float e = __builtin_amdgcn_exp2f(-az * az * 1.44269504088896340736f);
  const float er = copysignf(1.f - p * e, z);
  return 0.5f * x * (1.f + er);
}

template <typename T, bool FULL>
__global__ __launch_bounds__(512, 2) void graph_tail_kernel(const GbArgs a) {
  using E = Elem<T>;
  using vec = typename E::vec;
  using vec4 = typename E::vec4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* R1 = smem;                                   // F, later the Grapher's output X1   [144][GB_ROW]
  char* R0 = smem + GB_NODES * GB_ROW;               // max-relative / grouped-conv group, later the FFN's hidden layer
  int* sIdx = reinterpret_cast<int*>(smem + 2 * GB_NODES * GB_ROW);   // [144][9]
  float* sInv = reinterpret_cast<float*>(sIdx + GB_NODES * GB_K);     // FULL: [160] 1/|x_i|, [160] |xn_i|^2
  float* sXX = sInv + 160;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int n = blockIdx.x;
  const T* Fg = reinterpret_cast<const T*>(a.F) + (size_t)n * GB_NODES * GB_C;
  const T* Xg = reinterpret_cast<const T*>(a.X4) + (size_t)n * GB_NODES * GB_C;
  T* Og = reinterpret_cast<T*>(a.out) + (size_t)n * GB_NODES * GB_C;

  // B-fragment row pointers of this lane for the five node tiles (rows past the last node are clamped: their columns of the
  // accumulators are never stored)
  int rowB[5];
#pragma unroll
  for (int nt = 0; nt < 5; ++nt) rowB[nt] = min(nt * 32 + lr, GB_NODES - 1);
  auto bias4 = [&](const float* b, int c0) __attribute__((always_inline)) {
    return b ? *reinterpret_cast<const f32x4*>(b + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto gemm256 = [&](const vec (&A)[16], const char* src, f32x16 (&acc)[5]) __attribute__((always_inline)) {
#pragma unroll
    for (int nt = 0; nt < 5; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
      for (int nt = 0; nt < 5; ++nt)
        acc[nt] = mfma32x16(A[ks], *reinterpret_cast<const vec*>(src + rowB[nt] * GB_ROW + (2 * ks + lh) * 16), acc[nt]);
  };
  // ---- stage a (144, 256) tensor of this sample (all loads of a thread before its first LDS write)
  auto stage = [&](const T* src, char* dst) __attribute__((always_inline)) {
    constexpr int VPT = (GB_NODES * (GB_C / 8) + 511) / 512;     // 9
    vec tmp[VPT];
#pragma unroll
    for (int q = 0; q < VPT; ++q) {
      const int v = min(tid + q * 512, GB_NODES * (GB_C / 8) - 1);
      tmp[q] = *reinterpret_cast<const vec*>(src + (size_t)(v >> 5) * GB_C + (v & 31) * 8);
    }
#pragma unroll
    for (int q = 0; q < VPT; ++q) {
      const int v = tid + q * 512;
      if (v < GB_NODES * (GB_C / 8)) *reinterpret_cast<vec*>(dst + (v >> 5) * GB_ROW + (v & 31) * 16) = tmp[q];
    }
  };
  if (!FULL) {
    int ti[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) ti[q] = a.idx[(size_t)n * GB_NODES * GB_K + min(tid + q * 512, GB_NODES * GB_K - 1)];
    stage(Fg, R1);
#pragma unroll
    for (int q = 0; q < 3; ++q)
      if (tid + q * 512 < GB_NODES * GB_K) sIdx[tid + q * 512] = ti[q];
    __syncthreads();
  } else {
    // ---- fc1: F = X4 W1^T + b1 (Grapher_noBN.fc1, torch_vertex.py:190), rounded to the 16-bit type like the stored tensor
    vec A1[16];
    {
      const T* w1r = reinterpret_cast<const T*>(a.w1) + ((size_t)(wave * 32 + lr)) * GB_C + 8 * lh;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) A1[ks] = *reinterpret_cast<const vec*>(w1r + 16 * ks);
    }
    stage(Xg, R0);
    __syncthreads();
    {
      f32x16 acc1[5];
      gemm256(A1, R0, acc1);
#pragma unroll
      for (int nt = 0; nt < 5; ++nt) {
        const int node = nt * 32 + lr;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int co = wave * 32 + 8 * q + 4 * lh;
          const f32x4 b = bias4(a.b1, co);
          vec4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = (T)(acc1[nt][4 * q + r] + b[r]);
          if (node < GB_NODES) *reinterpret_cast<vec4*>(R1 + node * GB_ROW + co * 2) = o;
        }
      }
    }
    __syncthreads();
    // ---- the kNN graph of F (DenseDilatedKnnGraph, torch_edge.py:150-158): as gcn_knn_mfma_kernel (csrc/misc_kernels.hip),
    //      on the copy of F that is already here
    if (tid < 160) {
      float inv = 0.f, s2 = 0.f;
      if (tid < GB_NODES) {
        const char* rp = R1 + tid * GB_ROW;
        float ss = 0.f;
        for (int sl = 0; sl < GB_C / 8; ++sl) {
          float f[8];
          E::unpack(*reinterpret_cast<const vec*>(rp + sl * 16), f);
#pragma unroll
          for (int e = 0; e < 8; ++e) ss = fmaf(f[e], f[e], ss);
        }
        inv = 1.f / fmaxf(sqrtf(ss), 1e-12f);
        for (int sl = 0; sl < GB_C / 8; ++sl) {
          float f[8];
          E::unpack(*reinterpret_cast<const vec*>(rp + sl * 16), f);
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float t = f[e] * inv; s2 = fmaf(t, t, s2); }
        }
      }
      sInv[tid] = inv;
      sXX[tid] = s2;
    }
    __syncthreads();
    constexpr int TS = 10;                                  // 80 (row tile, accumulator slot) units over 8 waves
    const int u0 = wave * TS, u1 = u0 + TS;
    for (int rt = u0 >> 4; rt <= (u1 - 1) >> 4; ++rt) {
      const int s_lo = max(u0 - rt * 16, 0), cnt = min(u1 - rt * 16, 16) - s_lo;
      f32x16 gr[5];
#pragma unroll
      for (int ct = 0; ct < 5; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) gr[ct][e] = 0.f;
      const char* pa = R1 + min(rt * 32 + lr, GB_NODES - 1) * GB_ROW + lh * 16;
#pragma unroll 4
      for (int ks = 0; ks < GB_C / 16; ++ks) {
        const vec A = *reinterpret_cast<const vec*>(pa + ks * 32);
#pragma unroll
        for (int ct = 0; ct < 5; ++ct)
          gr[ct] = mfma32x16(A, *reinterpret_cast<const vec*>(R1 + rowB[ct] * GB_ROW + lh * 16 + ks * 32), gr[ct]);
      }
      float relv[TS][5];                       // relative_pos of this wave's (slot, column tile) pairs, requested together
#pragma unroll
      for (int t = 0; t < TS; ++t) {
        const int sl = min(s_lo + t, 15);
        const int i = min(rt * 32 + 8 * (sl >> 2) + 4 * lh + (sl & 3), GB_NODES - 1);
#pragma unroll
        for (int ct = 0; ct < 5; ++ct)
          relv[t][ct] = a.rel != nullptr ? a.rel[(size_t)i * GB_NODES + min(ct * 32 + lr, GB_NODES - 1)] : 0.f;
      }
      float d[TS][5];
      int row_i[TS];
#pragma unroll
      for (int t = 0; t < TS; ++t) {
        const int sl = min(s_lo + t, 15);                                   // wave-uniform
        float raw[5];
#pragma unroll
        for (int ct = 0; ct < 5; ++ct) {
          float val = gr[ct][0];
#pragma unroll
          for (int e = 1; e < 16; ++e) val = sl == e ? gr[ct][e] : val;
          raw[ct] = val;
        }
        const int i = rt * 32 + 8 * (sl >> 2) + 4 * lh + (sl & 3);
        const bool row_ok = i < GB_NODES && t < cnt;
        row_i[t] = row_ok ? i : -1;
        const float inv_i = sInv[min(i, 159)], xx_i = sXX[min(i, 159)];
#pragma unroll
        for (int ct = 0; ct < 5; ++ct) {
          const int j = ct * 32 + lr;
          d[t][ct] = INFINITY;
          if (row_ok && j < GB_NODES) {
            float dd = (xx_i + (-2.f * (raw[ct] * inv_i * sInv[j]))) + sXX[j];
            if (a.rel != nullptr) dd += relv[t][ct];
            d[t][ct] = dd;
          }
        }
      }
      for (int r = 0; r < GB_K; ++r) {
        float bv[TS];
        int bj[TS];
#pragma unroll
        for (int t = 0; t < TS; ++t) {
          bv[t] = d[t][0];
          bj[t] = lr;
#pragma unroll
          for (int ct = 1; ct < 5; ++ct)
            if (d[t][ct] < bv[t]) { bv[t] = d[t][ct]; bj[t] = ct * 32 + lr; }
        }
#define GB_STEP(CTRL)                                                                                                       \
        _Pragma("unroll") for (int t = 0; t < TS; ++t) {                                                                    \
          const float ov = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, bv[t]), CTRL, 0xF, 0xF, true)); \
          const int oj = __builtin_amdgcn_mov_dpp(bj[t], CTRL, 0xF, 0xF, true);                                             \
          if (ov < bv[t] || (ov == bv[t] && oj < bj[t])) { bv[t] = ov; bj[t] = oj; }                                        \
        }
        GB_STEP(0xB1)
        GB_STEP(0x4E)
        GB_STEP(0x141)
        GB_STEP(0x140)
#undef GB_STEP
#pragma unroll
        for (int t = 0; t < TS; ++t) {
          const float ov = __shfl_xor(bv[t], 16, 64);
          const int oj = __shfl_xor(bj[t], 16, 64);
          if (ov < bv[t] || (ov == bv[t] && oj < bj[t])) { bv[t] = ov; bj[t] = oj; }
        }
#pragma unroll
        for (int t = 0; t < TS; ++t) {
#pragma unroll
          for (int ct = 0; ct < 5; ++ct)
            if (ct * 32 + lr == bj[t]) d[t][ct] = INFINITY;
          if (lr == 0 && row_i[t] >= 0) {
            sIdx[row_i[t] * GB_K + r] = bj[t];
            if (a.idx_out != nullptr) a.idx_out[((size_t)n * GB_NODES + row_i[t]) * GB_K + r] = bj[t];
          }
        }
      }
    }
    __syncthreads();
  }


  // ---- Grapher: per group of the grouped conv (64 source channels -> 128 interleaved max-relative channels -> 128 outputs),
  //      fc2 accumulated over the groups in registers
  f32x16 acc2[5];
#pragma unroll
  for (int nt = 0; nt < 5; ++nt)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc2[nt][e] = 0.f;
  const int ctg = wave & 3;                                   // grouped conv: this wave's 32-channel tile of the group
  const int ntg0 = wave < 4 ? 0 : 3, ntgn = wave < 4 ? 3 : 2;  //               ... and its node tiles
  // Every weight fragment is requested a phase before its multiplies (one workgroup per CU and eight waves: nothing else
  // hides a load's latency): the grouped conv's for group g + 1 and fc2's for group g during group g, the FFN's during the
  // epilogue before them.
  vec Ag[8];
  {
    const T* wr = reinterpret_cast<const T*>(a.wg) + ((size_t)(ctg * 32 + lr)) * 128 + 8 * lh;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) Ag[ks] = *reinterpret_cast<const vec*>(wr + 16 * ks);
  }
  for (int g = 0; g < 4; ++g) {
    // (a) max-relative features of the group, channels interleaved [x_c, max_k(x_c[nbr] - x_c)] (torch_vertex.py:22-29)
    for (int t = tid; t < GB_NODES * 8; t += 512) {
      const int node = t >> 3, cv = t & 7;
      const int cb = (64 * g + 8 * cv) * 2;                  // byte offset of the eight source channels in a row of F
      float xi[8], m[8];
      E::unpack(*reinterpret_cast<const vec*>(R1 + node * GB_ROW + cb), xi);
#pragma unroll
      for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
#pragma unroll
      for (int k = 0; k < GB_K; ++k) {
        const int j = sIdx[node * GB_K + k];
        float xj[8];
        E::unpack(*reinterpret_cast<const vec*>(R1 + j * GB_ROW + cb), xj);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], xj[e] - xi[e]);
      }
      float o[16];
#pragma unroll
      for (int e = 0; e < 8; ++e) { o[2 * e] = xi[e]; o[2 * e + 1] = m[e]; }
      *reinterpret_cast<vec*>(R0 + node * GB_ROWG + 32 * cv) = E::pack(o);
      *reinterpret_cast<vec*>(R0 + node * GB_ROWG + 32 * cv + 16) = E::pack(o + 8);
    }
    __syncthreads();
    // (b) grouped 1x1 conv of the group: 128 -> 128, this wave's 32 output channels x its node tiles
    f32x16 accg[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) accg[i][e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if (i < ntgn)
          accg[i] = mfma32x16(Ag[ks], *reinterpret_cast<const vec*>(R0 + rowB[ntg0 + i] * GB_ROWG + (2 * ks + lh) * 16), accg[i]);
    if (g < 3) {
      const T* wr = reinterpret_cast<const T*>(a.wg) + ((size_t)((g + 1) * 128 + ctg * 32 + lr)) * 128 + 8 * lh;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) Ag[ks] = *reinterpret_cast<const vec*>(wr + 16 * ks);
    }
    // fc2's weight fragments of this group's K slice, requested before the barrier
    vec A2[8];
    {
      const T* wr = reinterpret_cast<const T*>(a.w2) + ((size_t)(wave * 32 + lr)) * 512 + 128 * g + 8 * lh;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) A2[ks] = *reinterpret_cast<const vec*>(wr + 16 * ks);
    }
    __syncthreads();                 // every wave is done reading the max-relative features
    // (c) GELU(conv + bias) back into the same rows
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i >= ntgn) continue;
      const int node = (ntg0 + i) * 32 + lr;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = ctg * 32 + 8 * q + 4 * lh;
        const f32x4 b = bias4(a.bg, g * 128 + co);
        vec4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (T)gelu_as(accg[i][4 * q + r] + b[r]);
        if (node < GB_NODES) *reinterpret_cast<vec4*>(R0 + node * GB_ROWG + co * 2) = o;
      }
    }
    __syncthreads();
    // (d) fc2 over this group's 128 input channels: this wave's 32 output channels x all five node tiles
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
      for (int nt = 0; nt < 5; ++nt)
        acc2[nt] = mfma32x16(A2[ks], *reinterpret_cast<const vec*>(R0 + rowB[nt] * GB_ROWG + (2 * ks + lh) * 16), acc2[nt]);
    __syncthreads();                 // the next group's features overwrite these rows
  }

  // ---- (e) X1 = fc2 + bias + X4 (Grapher residual) -> LDS (F is dead)
  // the Grapher's residual (this wave's 32 channels of every node) and the FFN's first weight fragments
  vec4 xres[5][4];
#pragma unroll
  for (int nt = 0; nt < 5; ++nt)
#pragma unroll
    for (int q = 0; q < 4; ++q)
      xres[nt][q] = *reinterpret_cast<const vec4*>(Xg + (size_t)min(nt * 32 + lr, GB_NODES - 1) * GB_C + wave * 32 + 8 * q + 4 * lh);
  vec A3[16];
  const T* w3r = reinterpret_cast<const T*>(a.w3) + ((size_t)(wave * 32 + lr)) * GB_C + 8 * lh;
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) A3[ks] = *reinterpret_cast<const vec*>(w3r + 16 * ks);
#pragma unroll
  for (int nt = 0; nt < 5; ++nt) {
    const int node = nt * 32 + lr;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co = wave * 32 + 8 * q + 4 * lh;
      const f32x4 b = bias4(a.b2, co);
      vec4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (T)((acc2[nt][4 * q + r] + b[r]) + (float)xres[nt][q][r]);
      if (node < GB_NODES) *reinterpret_cast<vec4*>(R1 + node * GB_ROW + co * 2) = o;
    }
  }
  __syncthreads();

  // ---- (f) FFN fc1 + GELU -> LDS, (g) FFN fc2 + bias + X1 -> global
  f32x16 acc[5];
  gemm256(A3, R1, acc);
  vec A4[16];
  const T* w4r = reinterpret_cast<const T*>(a.w4) + ((size_t)(wave * 32 + lr)) * GB_C + 8 * lh;
#pragma unroll
  for (int ks = 0; ks < 16; ++ks) A4[ks] = *reinterpret_cast<const vec*>(w4r + 16 * ks);
#pragma unroll
  for (int nt = 0; nt < 5; ++nt) {
    const int node = nt * 32 + lr;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co = wave * 32 + 8 * q + 4 * lh;
      const f32x4 b = bias4(a.b3, co);
      vec4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (T)gelu_as(acc[nt][4 * q + r] + b[r]);
      if (node < GB_NODES) *reinterpret_cast<vec4*>(R0 + node * GB_ROW + co * 2) = o;
    }
  }
  __syncthreads();
  gemm256(A4, R0, acc);
#pragma unroll
  for (int nt = 0; nt < 5; ++nt) {
    const int node = nt * 32 + lr;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co = wave * 32 + 8 * q + 4 * lh;
      const f32x4 b = bias4(a.b4, co);
      const vec4 rv = *reinterpret_cast<const vec4*>(R1 + min(node, GB_NODES - 1) * GB_ROW + co * 2);
      vec4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (T)((acc[nt][4 * q + r] + b[r]) + (float)rv[r]);
      if (node < GB_NODES) *reinterpret_cast<vec4*>(Og + (size_t)node * GB_C + co) = o;
    }
  }
}

}  // namespace

static int launch_graph(GbArgs& a, int dtype, bool full, hipStream_t s) {
  const size_t lds = 2 * (size_t)GB_NODES * GB_ROW + (size_t)GB_NODES * GB_K * 4 + 2 * 160 * sizeof(float);
  static UnclDevOnce attr[4];
  const int which = (dtype == UNCL_F16 ? 0 : 1) + (full ? 2 : 0);
  const void* kern = which == 0 ? reinterpret_cast<const void*>(graph_tail_kernel<f16_t, false>)
                     : which == 1 ? reinterpret_cast<const void*>(graph_tail_kernel<bf16_t, false>)
                     : which == 2 ? reinterpret_cast<const void*>(graph_tail_kernel<f16_t, true>)
                                  : reinterpret_cast<const void*>(graph_tail_kernel<bf16_t, true>);
  if (attr[which].need()) {
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return UNCL_ERR_LAUNCH;
    attr[which].done();
  }
  switch (which) {
    case 0: hipLaunchKernelGGL((graph_tail_kernel<f16_t, false>), dim3(a.N), dim3(512), lds, s, a); break;
    case 1: hipLaunchKernelGGL((graph_tail_kernel<bf16_t, false>), dim3(a.N), dim3(512), lds, s, a); break;
    case 2: hipLaunchKernelGGL((graph_tail_kernel<f16_t, true>), dim3(a.N), dim3(512), lds, s, a); break;
    default: hipLaunchKernelGGL((graph_tail_kernel<bf16_t, true>), dim3(a.N), dim3(512), lds, s, a); break;
  }
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

// F: fc1 output, idx: its kNN graph, X4: the block's input.  Weights in the packed 1x1 layout of uncl_pack_conv_weight
// ([group][Cout][Cin], K contiguous).  16-bit types only; no DropPath scaling (inference).
extern "C" int uncl_gcn_tail(const void* F, const int32_t* idx, const void* X4, const void* wg, const float* bg, const void* w2,
                             const float* b2, const void* w3, const float* b3, const void* w4, const float* b4, void* out,
                             int dtype, int N, void* stream) {
  if (!F || !idx || !X4 || !wg || !w2 || !w3 || !w4 || !out || N <= 0 || !uncl_is_h16(dtype)) return UNCL_ERR_ARG;
  GbArgs a = {};
  a.F = F; a.idx = idx; a.X4 = X4; a.wg = wg; a.w2 = w2; a.w3 = w3; a.w4 = w4;
  a.bg = bg; a.b2 = b2; a.b3 = b3; a.b4 = b4; a.out = out; a.N = N;
  return launch_graph(a, dtype, false, reinterpret_cast<hipStream_t>(stream));
}

// The whole block in one launch: fc1 and the kNN graph (relative_pos: (144,144) or NULL) are computed in the kernel as well;
// idx_out (N,144,9), if not NULL, receives the graph.
extern "C" int uncl_gcn_block(const void* X4, const void* w1, const float* b1, const float* relative_pos, const void* wg,
                              const float* bg, const void* w2, const float* b2, const void* w3, const float* b3, const void* w4,
                              const float* b4, int32_t* idx_out, void* out, int dtype, int N, void* stream) {
  if (!X4 || !w1 || !wg || !w2 || !w3 || !w4 || !out || N <= 0 || !uncl_is_h16(dtype)) return UNCL_ERR_ARG;
  GbArgs a = {};
  a.X4 = X4; a.w1 = w1; a.b1 = b1; a.rel = relative_pos; a.idx_out = idx_out;
  a.wg = wg; a.w2 = w2; a.w3 = w3; a.w4 = w4; a.bg = bg; a.b2 = b2; a.b3 = b3; a.b4 = b4; a.out = out; a.N = N;
  return launch_graph(a, dtype, true, reinterpret_cast<hipStream_t>(stream));
}
