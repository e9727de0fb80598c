// Loss heads of the GanTrainer step (GanTrainerImg.py:219-229, 341-439; GanTrainer.py:669-682), the TMQI
// naturalness score used for sample selection (TMQI.py:210-242), the Gaussian-variance backward, and fused Adam.
// All are tiny or HBM-bound; each "loss" kernel returns the weighted loss AND its input gradients in one pass
// (the weights are constants of the step), fp32 with fp64 where the reference is fp64.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------------
// contrastive GAN loss: half(t1,t2) = CE([t1_i, t2_0..t2_{N-1}], 0);  L = w * (half(r, f) + half(-f, -r))
// ------------------------------------------------------------------------------------------------------
__global__ void cgan_kernel(const float* __restrict__ r, const float* __restrict__ f, int N, float w, float* loss,
                            float* g_r, float* g_f, int accumulate_loss) {
  // single workgroup; N <= 4096.  Deterministic: every output element is produced by one thread in a fixed order.
  extern __shared__ float sh[];  // lse0[N] (rows [r_i, f_*]), lse1[N] (rows [-f_i, -r_*])
  float* lse0 = sh;
  float* lse1 = sh + N;
  __shared__ double part[256];
  double lsum = 0.0;
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    float mx0 = r[i], mx1 = -f[i];
    for (int j = 0; j < N; ++j) { mx0 = fmaxf(mx0, f[j]); mx1 = fmaxf(mx1, -r[j]); }
    float s0 = expf(r[i] - mx0), s1 = expf(-f[i] - mx1);
    for (int j = 0; j < N; ++j) { s0 += expf(f[j] - mx0); s1 += expf(-r[j] - mx1); }
    lse0[i] = mx0 + logf(s0);
    lse1[i] = mx1 + logf(s1);
    lsum += (double)(lse0[i] - r[i]) + (double)(lse1[i] + f[i]);
  }
  part[threadIdx.x] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < (int)blockDim.x; ++i) t += part[i];
    loss[0] = (accumulate_loss ? loss[0] : 0.f) + (float)(t * (double)w / (double)N);
  }
  const float s = w / (float)N;
  for (int j = threadIdx.x; j < N; j += blockDim.x) {
    // d/dr_j: own row of half 0 (p0 - 1) and column j of every row of half 1 (logit -r_j)
    float gr = s * (expf(r[j] - lse0[j]) - 1.f);
    float gf = -s * (expf(-f[j] - lse1[j]) - 1.f);
    for (int i = 0; i < N; ++i) {
      gr -= s * expf(-r[j] - lse1[i]);
      gf += s * expf(f[j] - lse0[i]);
    }
    if (g_r) g_r[j] = gr;
    if (g_f) g_f[j] = gf;
  }
}

// ------------------------------------------------------------------------------------------------------
// NCE similarity  s(a,b) = (1/HW) sum_e a*b / (c + k|a-b|)   and the 2-way InfoNCE built on it
// ------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ float ldf(const T* p, size_t i) { return (float)p[i]; }

// pass 1: per-sample partial sums of the two similarities.  grid (blocks, N)
template <typename T>
__global__ __launch_bounds__(256) void nce_sim_kernel(const T* __restrict__ a, const T* __restrict__ p, const T* __restrict__ q,
                                                      size_t E, size_t p_stride, size_t q_stride, float k, float c,
                                                      float* __restrict__ partial, const int* __restrict__ rows) {
  __shared__ float red[4][2];
  const int n = blockIdx.y;
  const T* an = a + (size_t)n * E;
  // rows (device): the shared positive / negative are rows rows[0] / rows[1] of the anchor tensor (selected on the device)
  const T* pn = rows ? a + (size_t)rows[0] * E : p + (size_t)n * p_stride;
  const T* qn = rows ? a + (size_t)rows[1] * E : q + (size_t)n * q_stride;
  float sp = 0.f, sq = 0.f;
  constexpr int V = Elem<T>::EPV;
  using vec = typename Elem<T>::vec;
  const bool vec_ok = E % V == 0 && ((reinterpret_cast<uintptr_t>(an) | reinterpret_cast<uintptr_t>(pn) | reinterpret_cast<uintptr_t>(qn)) & 15) == 0;
  if (vec_ok) {
    // 16-byte loads: the feature maps of the generator are 2 M elements per sample
    for (size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * V; e < E; e += (size_t)gridDim.x * 256 * V) {
      float av[V], pv[V], qv[V];
      Elem<T>::unpack(*reinterpret_cast<const vec*>(an + e), av);
      Elem<T>::unpack(*reinterpret_cast<const vec*>(pn + e), pv);
      Elem<T>::unpack(*reinterpret_cast<const vec*>(qn + e), qv);
#pragma unroll
      for (int i = 0; i < V; ++i) {      // hardware reciprocal (1 ulp) as in the gradient pass
        sp += av[i] * pv[i] * __builtin_amdgcn_rcpf(c + k * fabsf(av[i] - pv[i]));
        sq += av[i] * qv[i] * __builtin_amdgcn_rcpf(c + k * fabsf(av[i] - qv[i]));
      }
    }
  } else {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < E; e += (size_t)gridDim.x * 256) {
      const float av = ldf(an, e), pv = ldf(pn, e), qv = ldf(qn, e);
      sp += av * pv * (1.f / (c + k * fabsf(av - pv)));
      sq += av * qv * (1.f / (c + k * fabsf(av - qv)));
    }
  }
  sp = wave_sum(sp); sq = wave_sum(sq);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = sp; red[threadIdx.x >> 6][1] = sq; }
  __syncthreads();
  if (threadIdx.x < 2)
    partial[((size_t)n * gridDim.x + blockIdx.x) * 2 + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// pass 2: finish the sums, the 2-way cross-entropy and d loss / d s_pos, d loss / d s_neg.  single block
__global__ void nce_ce_kernel(const float* __restrict__ partial, int blocks, int N, double inv_hw, float w, float* loss,
                              float* __restrict__ gs /* [N][2] */, int accumulate_loss /* bit 0: add to loss, bit 1: LMCL */) {
  __shared__ double sl;
  if (threadIdx.x == 0) sl = 0.0;
  __syncthreads();
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    double sp = 0.0, sq = 0.0;
    for (int b = 0; b < blocks; ++b) {
      sp += (double)partial[((size_t)n * blocks + b) * 2];
      sq += (double)partial[((size_t)n * blocks + b) * 2 + 1];
    }
    const float fp = (float)(sp * inv_hw), fq = (float)(sq * inv_hw);
    if (accumulate_loss & 2) {
      // LMCL with one negative (GanTrainerImg.py:441-450): -log(exp(s_pos) / exp(s_neg)) = s_neg - s_pos
      atomicAdd(&sl, (double)(fq - fp));
      gs[2 * n] = -w / (float)N;
      gs[2 * n + 1] = w / (float)N;
    } else {
      const float mx = fmaxf(fp, fq);
      const float lse = mx + logf(expf(fp - mx) + expf(fq - mx));
      atomicAdd(&sl, (double)(lse - fp));
      gs[2 * n] = w / (float)N * (expf(fp - lse) - 1.f);
      gs[2 * n + 1] = w / (float)N * expf(fq - lse);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = ((accumulate_loss & 1) ? loss[0] : 0.f) + (float)(sl * (double)w / (double)N);
}

// the similarities themselves (uncl_nce_similarity): the same finishing sums as nce_ce_kernel, no cross-entropy
__global__ void nce_sims_final_kernel(const float* __restrict__ partial, int blocks, int N, double inv_hw, float* __restrict__ sims) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * N) return;
  const int n = i >> 1, j = i & 1;
  double sacc = 0.0;
  for (int b = 0; b < blocks; ++b) sacc += (double)partial[((size_t)n * blocks + b) * 2 + j];
  sims[i] = (float)(sacc * inv_hw);
}

// pass 3: gradients.  One thread per element e, looping over the samples, so that a positive / negative that is
// ONE row shared by every sample (stride 0, GanTrainerImg.py:401-402) gets its summed gradient without atomics.
template <typename T>
__global__ __launch_bounds__(256) void nce_grad_kernel(const T* __restrict__ a, const T* __restrict__ p, const T* __restrict__ q,
                                                       size_t E, size_t p_stride, size_t q_stride, int N, float k, float c,
                                                       float inv_hw, const float* __restrict__ gs, float* __restrict__ g_a,
                                                       float* __restrict__ g_p, float* __restrict__ g_q, int accumulate) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < E; e += (size_t)gridDim.x * 256) {
    float accp = 0.f, accq = 0.f;
    for (int n = 0; n < N; ++n) {
      const float av = ldf(a, (size_t)n * E + e), pv = ldf(p, (size_t)n * p_stride + e), qv = ldf(q, (size_t)n * q_stride + e);
      const float dp = av - pv, dq = av - qv;
      const float ip = 1.f / (c + k * fabsf(dp)), iq = 1.f / (c + k * fabsf(dq));
      const float sgp = dp > 0.f ? 1.f : (dp < 0.f ? -1.f : 0.f), sgq = dq > 0.f ? 1.f : (dq < 0.f ? -1.f : 0.f);
      const float gp = gs[2 * n] * inv_hw, gq = gs[2 * n + 1] * inv_hw;
      // d/da [a b / (c + k|a-b|)] = b i - a b k sgn i^2 ;  d/db = a i + a b k sgn i^2
      const float tp = av * pv * k * sgp * ip * ip, tq = av * qv * k * sgq * iq * iq;
      const float ga = gp * (pv * ip - tp) + gq * (qv * iq - tq);
      if (g_a) { float* d = g_a + (size_t)n * E + e; *d = accumulate ? *d + ga : ga; }
      const float gpp = gp * (av * ip + tp), gqq = gq * (av * iq + tq);
      if (p_stride == 0) accp += gpp; else if (g_p) { float* d = g_p + (size_t)n * E + e; *d = accumulate ? *d + gpp : gpp; }
      if (q_stride == 0) accq += gqq; else if (g_q) { float* d = g_q + (size_t)n * E + e; *d = accumulate ? *d + gqq : gqq; }
    }
    if (p_stride == 0 && g_p) { float* d = g_p + e; *d = accumulate ? *d + accp : accp; }
    if (q_stride == 0 && g_q) { float* d = g_q + e; *d = accumulate ? *d + accq : accq; }
  }
}

// Backward-time form of pass 3 (uncl_nce_backward): 16-byte vectors, gradients in the feature dtype, the upstream scalar
// read on the device.  p_row / q_row >= 0: the shared positive / negative IS that row of the anchor tensor (infoNCE2,
// GanTrainerImg.py:398-402); its summed gradient is then folded into g_a's row in registers, so the feature tensor gets one
// complete gradient in one pass (no slice-backward zero fill, no adds, no fp32 copy of a 134 MB tensor).
template <typename T, typename GT>
__global__ __launch_bounds__(256) void nce_bwd_kernel(const T* __restrict__ a, const T* __restrict__ p, const T* __restrict__ q,
                                                      size_t E, size_t p_stride, size_t q_stride, int N, float k, float c,
                                                      float inv_hw, const float* __restrict__ gs, const float* __restrict__ upstream,
                                                      GT* __restrict__ g_a, GT* __restrict__ g_p, GT* __restrict__ g_q, int p_row,
                                                      int q_row, const int* __restrict__ rows) {
  constexpr int V = Elem<T>::EPV;
  using vec = typename Elem<T>::vec;
  if (rows) {           // row indices chosen on the device (arg-max / arg-min of the naturalness scores)
    p_row = rows[0]; q_row = rows[1];
    p = a + (size_t)p_row * E; q = a + (size_t)q_row * E;
  }
  const float up = (upstream ? upstream[0] : 1.f) * inv_hw;
  auto store = [&](GT* dst, const float* f) {
    if constexpr (sizeof(GT) == sizeof(T)) {
      *reinterpret_cast<vec*>(dst) = Elem<T>::pack(f);
    } else {
#pragma unroll
      for (int i = 0; i < V; ++i) dst[i] = (GT)f[i];
    }
  };
  for (size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * V; e < E; e += (size_t)gridDim.x * 256 * V) {
    float accp[V], accq[V], hold_p[V], hold_q[V];
#pragma unroll
    for (int i = 0; i < V; ++i) accp[i] = accq[i] = hold_p[i] = hold_q[i] = 0.f;
    float pv[V], qv[V];
    if (p_stride == 0) Elem<T>::unpack(*reinterpret_cast<const vec*>(p + e), pv);
    if (q_stride == 0) Elem<T>::unpack(*reinterpret_cast<const vec*>(q + e), qv);
#pragma unroll 2
    for (int n = 0; n < N; ++n) {
      float av[V], ga[V], gpp[V], gqq[V];
      Elem<T>::unpack(*reinterpret_cast<const vec*>(a + (size_t)n * E + e), av);
      if (p_stride != 0) Elem<T>::unpack(*reinterpret_cast<const vec*>(p + (size_t)n * p_stride + e), pv);
      if (q_stride != 0) Elem<T>::unpack(*reinterpret_cast<const vec*>(q + (size_t)n * q_stride + e), qv);
      const float gp = gs[2 * n] * up, gq = gs[2 * n + 1] * up;
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const float dp = av[i] - pv[i], dq = av[i] - qv[i];
        // hardware reciprocal (1 ulp): the IEEE division sequence made this HBM-sized pass VALU-bound
        const float ip = __builtin_amdgcn_rcpf(c + k * fabsf(dp)), iq = __builtin_amdgcn_rcpf(c + k * fabsf(dq));
        const float sgp = dp > 0.f ? 1.f : (dp < 0.f ? -1.f : 0.f), sgq = dq > 0.f ? 1.f : (dq < 0.f ? -1.f : 0.f);
        const float tp = av[i] * pv[i] * k * sgp * ip * ip, tq = av[i] * qv[i] * k * sgq * iq * iq;
        ga[i] = gp * (pv[i] * ip - tp) + gq * (qv[i] * iq - tq);
        gpp[i] = gp * (av[i] * ip + tp);
        gqq[i] = gq * (av[i] * iq + tq);
        accp[i] += gpp[i];
        accq[i] += gqq[i];
      }
      if (n == p_row || n == q_row) {       // finished after the loop, when the shared rows' sums are known
#pragma unroll
        for (int i = 0; i < V; ++i) {
          if (n == p_row) hold_p[i] = ga[i];
          if (n == q_row) hold_q[i] = ga[i];
        }
      } else if (g_a) {
        store(g_a + (size_t)n * E + e, ga);
      }
      if (p_stride != 0 && g_p) store(g_p + (size_t)n * E + e, gpp);
      if (q_stride != 0 && g_q) store(g_q + (size_t)n * E + e, gqq);
    }
    if (p_row >= 0 && p_row == q_row) {
#pragma unroll
      for (int i = 0; i < V; ++i) hold_p[i] += accp[i] + accq[i];
      if (g_a) store(g_a + (size_t)p_row * E + e, hold_p);
    } else {
      if (p_row >= 0 && g_a) {
#pragma unroll
        for (int i = 0; i < V; ++i) hold_p[i] += accp[i];
        store(g_a + (size_t)p_row * E + e, hold_p);
      }
      if (q_row >= 0 && g_a) {
#pragma unroll
        for (int i = 0; i < V; ++i) hold_q[i] += accq[i];
        store(g_a + (size_t)q_row * E + e, hold_q);
      }
    }
    if (p_stride == 0 && p_row < 0 && g_p) store(g_p + e, accp);
    if (q_stride == 0 && q_row < 0 && g_q) store(g_q + e, accq);
  }
}

// ------------------------------------------------------------------------------------------------------
// Weighted sum of device scalars (the trainers' `err = w1*l1 + w2*l2 + ...`, GanTrainerImg.py:285-313) and its backward:
// one launch each instead of two tiny element-wise launches (and two autograd nodes) per term.
// ------------------------------------------------------------------------------------------------------
struct WSumArgs {
  const float* term[UNCL_WSUM_MAX];
  float w[UNCL_WSUM_MAX];
  int n;
};
__global__ void weighted_sum_kernel(const WSumArgs t, float* __restrict__ out) {
  if (threadIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < t.n; ++i) s = fmaf(t.w[i], t.term[i][0], s);
    out[0] = s;
  }
}
__global__ void weighted_sum_bwd_kernel(const WSumArgs t, const float* __restrict__ g, float* __restrict__ out) {
  if ((int)threadIdx.x < t.n) out[threadIdx.x] = g[0] * t.w[threadIdx.x];
}

// ------------------------------------------------------------------------------------------------------
// L1 between two per-sample scalars: L = w * mean_n |a_n - b_n| ; g_a = w sign / N
// ------------------------------------------------------------------------------------------------------
__global__ void l1_pairs_kernel(const float* a, int a_stride, const float* b, int b_stride, int N, float w, float* loss,
                                float* g_a, float* g_b, int accumulate_loss) {
  __shared__ double sl;
  if (threadIdx.x == 0) sl = 0.0;
  __syncthreads();
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    const float d = a[(size_t)n * a_stride] - b[(size_t)n * b_stride];
    atomicAdd(&sl, (double)fabsf(d));
    const float s = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    if (g_a) g_a[n] = w * s / (float)N;
    if (g_b) g_b[n] = -w * s / (float)N;
  }
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = (accumulate_loss ? loss[0] : 0.f) + (float)(sl * (double)w / (double)N);
}

// The two L1 terms of pseudo_label_loss (GanTrainerImg.py:360-367) in one launch: per-patch statistics s[i] = {mean, mean local
// variance}, the pseudo label is row r = row[0] (chosen on the device), L_j = mean_i |s[i][j] - s[r][j]|.  grad[i][j] = dL_j / ds[i][j]:
// sign / N for the patches, minus the sum of the signs / N for the label row itself (what autograd's expand + index_select backward
// add up to; its own |0| term contributes nothing).
__global__ void l1_to_row_kernel(const float* __restrict__ s, int N, const int* __restrict__ row, float* __restrict__ loss,
                                 float* __restrict__ grad) {
  __shared__ double sl[2];
  __shared__ int ssum[2];
  if (threadIdx.x < 2) { sl[threadIdx.x] = 0.0; ssum[threadIdx.x] = 0; }
  __syncthreads();
  const int r = min(max(row[0], 0), N - 1);
  for (int i = threadIdx.x; i < N; i += blockDim.x)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float d = s[2 * i + j] - s[2 * r + j];
      const int sg = d > 0.f ? 1 : (d < 0.f ? -1 : 0);
      atomicAdd(&sl[j], (double)fabsf(d));
      atomicAdd(&ssum[j], sg);
      grad[2 * i + j] = (float)sg / (float)N;
    }
  __syncthreads();
  if (threadIdx.x < 2) {
    loss[threadIdx.x] = (float)(sl[threadIdx.x] / (double)N);
    grad[2 * r + threadIdx.x] = -(float)ssum[threadIdx.x] / (float)N;
  }
}
// out[0][i] = grad[i][0] * g0, out[1][i] = grad[i][1] * g1: the per-patch gradients of the two statistics, ready for
// uncl_add_per_sample_const / uncl_gauss_var_backward
__global__ void l1_to_row_bwd_kernel(const float* __restrict__ grad, int N, const float* __restrict__ g0, const float* __restrict__ g1,
                                     float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) { out[i] = grad[2 * i] * g0[0]; out[N + i] = grad[2 * i + 1] * g1[0]; }
}

// ------------------------------------------------------------------------------------------------------
// TMQI statistical naturalness, fp64 (TMQI.py:210-242): u = mean(L), sig = mean over 11x11 blocks (zero padded to the
// next multiple of 11, always at least one extra) of the population std; N = beta_pdf(sig/64.29)/C0 * gauss(u)
// ------------------------------------------------------------------------------------------------------
__device__ double beta_pdf(double x, double a, double b) {
  if (x <= 0.0 || x >= 1.0) return 0.0;
  return exp(lgamma(a + b) - lgamma(a) - lgamma(b) + (a - 1.0) * log(x) + (b - 1.0) * log1p(-x));
}

// one workgroup per image region; the image is x*scale with x fp32 in a (rows x row_stride) frame
__global__ __launch_bounds__(256) void tmqi_n_kernel(const float* __restrict__ x, int n_per_frame, int h, int w, int frame_h,
                                                     int frame_w, float scale, double* __restrict__ scores) {
  // image id = blockIdx.x: frame = id / n_per_frame, patch = id % n_per_frame laid out row-major on a
  // (frame_h/h) x (frame_w/w) grid of patches
  __shared__ double red_s[4], red_sig[4];
  const int id = blockIdx.x;
  const int frame = id / n_per_frame, patch = id % n_per_frame;
  const int ppr = frame_w / w;
  const float* base = x + (size_t)frame * frame_h * frame_w + (size_t)(patch / ppr) * h * frame_w + (size_t)(patch % ppr) * w;
  const int bh = (h + (11 - h % 11)) / 11, bw = (w + (11 - w % 11)) / 11;
  double s_all = 0.0, s_sig = 0.0;
  for (int b = threadIdx.x; b < bh * bw; b += 256) {
    const int by = b / bw, bx = b % bw;
    double s1 = 0.0, s2 = 0.0;
    for (int dy = 0; dy < 11; ++dy) {
      const int yy = by * 11 + dy;
      if (yy >= h) continue;
      for (int dx = 0; dx < 11; ++dx) {
        const int xx = bx * 11 + dx;
        if (xx >= w) continue;
        const double v = (double)(base[(size_t)yy * frame_w + xx] * scale);
        s1 += v;
        s2 += v * v;
      }
    }
    s_all += s1;
    const double m = s1 / 121.0;
    const double var = s2 / 121.0 - m * m;
    s_sig += sqrt(var > 0.0 ? var : 0.0);
  }
  s_all = wave_sum_d(s_all); s_sig = wave_sum_d(s_sig);
  if ((threadIdx.x & 63) == 0) { red_s[threadIdx.x >> 6] = s_all; red_sig[threadIdx.x >> 6] = s_sig; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double u = ((red_s[0] + red_s[1]) + (red_s[2] + red_s[3])) / ((double)h * w);
    const double sig = ((red_sig[0] + red_sig[1]) + (red_sig[2] + red_sig[3])) / ((double)bh * bw);
    const double a = 4.4, bb = 10.1, mu = 115.94, sd = 27.99;
    const double mode = (a - 1.0) / (a + bb - 2.0);
    const double pc = beta_pdf(sig / 64.29, a, bb) / beta_pdf(mode, a, bb);
    const double z = (u - mu) / sd;
    scores[id] = exp(-0.5 * z * z) * pc;
  }
}

// first-occurrence arg-max / arg-min (sorted()[-1] / [0] then list.index, GanTrainerImg.py:398-402)
__global__ void argminmax_kernel(const double* __restrict__ s, int n, int32_t* __restrict__ out /* [2]: best, worst */) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int best = 0, worst = 0;
  for (int i = 1; i < n; ++i) {
    if (s[i] > s[best]) best = i;
    if (s[i] < s[worst]) worst = i;
  }
  out[0] = best;
  out[1] = worst;
}

// ------------------------------------------------------------------------------------------------------
// backward of mean(Gaussian local variance): d/dx_i = (2/P) (x_i Wc_i - (G^T mu)_i), times a per-image scale
// ------------------------------------------------------------------------------------------------------
constexpr int GW = 11, GT = 32, GM = GT + GW - 1 /*42 mu's*/, GI = GM + GW - 1 /*52 inputs*/;
struct GaussW { float g[GW]; };

__global__ __launch_bounds__(256) void gauss_var_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gscale,
                                                            float* __restrict__ gx, int H, int W, int tiles_x, GaussW gw,
                                                            int accumulate) {
  __shared__ float sx[GI * GI];
  __shared__ float sh[GI * GM];   // horizontal pass: rows GI, cols GM
  __shared__ float smu[GM * GM];  // mu, zero outside the valid window range
  __shared__ float sv[GM * GT];   // vertical back-projection: rows GT, cols GM
  const int n = blockIdx.y;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int y0 = ty * GT, x0 = tx * GT;
  const int Ho = H - (GW - 1), Wo = W - (GW - 1);
  const float* xn = x + (size_t)n * H * W;
  // inputs [y0-10, y0+GT+10), windows o in [y0-10, y0+GT)
  for (int i = threadIdx.x; i < GI * GI; i += 256) {
    const int ly = i / GI, lx = i - ly * GI;
    const int gy = y0 - 10 + ly, gxx = x0 - 10 + lx;
    sx[i] = (gy >= 0 && gy < H && gxx >= 0 && gxx < W) ? xn[(size_t)gy * W + gxx] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < GI * GM; i += 256) {
    const int ly = i / GM, lx = i - ly * GM;
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < GW; ++t) s = fmaf(gw.g[t], sx[ly * GI + lx + t], s);
    sh[i] = s;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < GM * GM; i += 256) {
    const int ly = i / GM, lx = i - ly * GM;
    const int oy = y0 - 10 + ly, ox = x0 - 10 + lx;
    float s = 0.f;
    if (oy >= 0 && oy < Ho && ox >= 0 && ox < Wo) {
#pragma unroll
      for (int t = 0; t < GW; ++t) s = fmaf(gw.g[t], sh[(ly + t) * GM + lx], s);
    }
    smu[i] = s;
  }
  __syncthreads();
  // (G^T mu)_i = sum_{dy,dx} w_dy w_dx mu[i - (dy,dx)]  (mu is zero outside the valid range)
  for (int i = threadIdx.x; i < GT * GM; i += 256) {
    const int ly = i / GM, lx = i - ly * GM;
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < GW; ++t) s = fmaf(gw.g[t], smu[(ly + 10 - t) * GM + lx], s);
    sv[i] = s;
  }
  __syncthreads();
  const float sc = gscale[n] * 2.f / ((float)Ho * (float)Wo);
  for (int i = threadIdx.x; i < GT * GT; i += 256) {
    const int ly = i / GT, lx = i - ly * GT;
    const int gy = y0 + ly, gxx = x0 + lx;
    if (gy < H && gxx < W) {
      float s = 0.f, wy = 0.f, wx = 0.f;
#pragma unroll
      for (int t = 0; t < GW; ++t) {
        s = fmaf(gw.g[t], sv[ly * GM + lx + 10 - t], s);
        if (gy - t >= 0 && gy - t < Ho) wy += gw.g[t];
        if (gxx - t >= 0 && gxx - t < Wo) wx += gw.g[t];
      }
      const float g = sc * (sx[(ly + 10) * GI + lx + 10] * wy * wx - s);
      float* d = gx + (size_t)n * H * W + (size_t)gy * W + gxx;
      *d = accumulate ? *d + g : g;
    }
  }
}

// Backward of uncl_gauss_stats for NHWC tensors: gx[n,y,x,c] (+)= g[n][0][c] / (H*W) + g[n][1][c] * d mean(local var) / dx.
// One (32x32 tile, channel) per workgroup; the channel's pixels are strided by C in memory, neighbouring channels'
// workgroups read the same lines out of L2.
template <typename T>
__global__ __launch_bounds__(256) void gauss_stats_bwd_kernel(const T* __restrict__ x, const float* __restrict__ gst,
                                                              T* __restrict__ gx, int H, int W, int C, int tiles_x, GaussW gw,
                                                              int accumulate) {
  __shared__ float sx[GI * GI];
  __shared__ float sh[GI * GM];
  __shared__ float smu[GM * GM];
  __shared__ float sv[GM * GT];
  const int n = blockIdx.y, ch = blockIdx.z;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int y0 = ty * GT, x0 = tx * GT;
  const int Ho = H - (GW - 1), Wo = W - (GW - 1);
  const T* xn = x + (size_t)n * H * W * C + ch;
  for (int i = threadIdx.x; i < GI * GI; i += 256) {
    const int ly = i / GI, lx = i - ly * GI;
    const int gy = y0 - 10 + ly, gxx = x0 - 10 + lx;
    sx[i] = (gy >= 0 && gy < H && gxx >= 0 && gxx < W) ? (float)xn[((size_t)gy * W + gxx) * C] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < GI * GM; i += 256) {
    const int ly = i / GM, lx = i - ly * GM;
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < GW; ++t) s = fmaf(gw.g[t], sx[ly * GI + lx + t], s);
    sh[i] = s;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < GM * GM; i += 256) {
    const int ly = i / GM, lx = i - ly * GM;
    const int oy = y0 - 10 + ly, ox = x0 - 10 + lx;
    float s = 0.f;
    if (oy >= 0 && oy < Ho && ox >= 0 && ox < Wo) {
#pragma unroll
      for (int t = 0; t < GW; ++t) s = fmaf(gw.g[t], sh[(ly + t) * GM + lx], s);
    }
    smu[i] = s;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < GT * GM; i += 256) {
    const int ly = i / GM, lx = i - ly * GM;
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < GW; ++t) s = fmaf(gw.g[t], smu[(ly + 10 - t) * GM + lx], s);
    sv[i] = s;
  }
  __syncthreads();
  const float gm = gst[((size_t)n * 2 + 0) * C + ch] / ((float)H * (float)W);
  const float sc = gst[((size_t)n * 2 + 1) * C + ch] * 2.f / ((float)Ho * (float)Wo);
  for (int i = threadIdx.x; i < GT * GT; i += 256) {
    const int ly = i / GT, lx = i - ly * GT;
    const int gy = y0 + ly, gxx = x0 + lx;
    if (gy < H && gxx < W) {
      float s = 0.f, wy = 0.f, wx = 0.f;
#pragma unroll
      for (int t = 0; t < GW; ++t) {
        s = fmaf(gw.g[t], sv[ly * GM + lx + 10 - t], s);
        if (gy - t >= 0 && gy - t < Ho) wy += gw.g[t];
        if (gxx - t >= 0 && gxx - t < Wo) wx += gw.g[t];
      }
      const float g = gm + sc * (sx[(ly + 10) * GI + lx + 10] * wy * wx - s);
      T* d = gx + ((size_t)n * H * W + (size_t)gy * W + gxx) * C + ch;
      *d = (T)(accumulate ? (float)*d + g : g);
    }
  }
}

// g[n][i] (+)= scale[n]  (gradient of a per-sample mean times a per-sample factor already divided by H*W)
__global__ void add_const_kernel(float* __restrict__ g, const float* __restrict__ scale, size_t per, int N, float mul,
                                 int accumulate) {
  const size_t total = per * N;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const float v = scale[i / per] * mul;
    g[i] = accumulate ? g[i] + v : v;
  }
}

// total variation (GanTrainer.py:669-682): L = w * 2 (sum dh^2 / ((H-1) W) + sum dw^2 / (H (W-1))) / N
__global__ __launch_bounds__(256) void tv_kernel(const float* __restrict__ x, float* __restrict__ gx, float* __restrict__ partial,
                                                 int N, int H, int W, float ch, float cw, int accumulate) {
  __shared__ float red[4];
  const size_t total = (size_t)N * H * W;
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int xx = (int)(i % W), yy = (int)((i / W) % H);
    const float v = x[i];
    float g = 0.f;
    if (yy + 1 < H) { const float d = x[i + W] - v; acc = fmaf(ch * d, d, acc); g -= 2.f * ch * d; }
    if (yy > 0) g += 2.f * ch * (v - x[i - W]);
    if (xx + 1 < W) { const float d = x[i + 1] - v; acc = fmaf(cw * d, d, acc); g -= 2.f * cw * d; }
    if (xx > 0) g += 2.f * cw * (v - x[i - 1]);
    if (gx) gx[i] = accumulate ? gx[i] + g : g;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void sum_f_kernel(const float* __restrict__ partial, int count, float* out, int accumulate) {
  double s = 0.0;
  for (int i = threadIdx.x; i < count; i += 64) s += (double)partial[i];
  s = wave_sum_d(s);
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + (float)s;
}

// ------------------------------------------------------------------------------------------------------
// fused multi-tensor Adam (torch.optim.Adam, no weight decay, no amsgrad; main_train_image.py:29-32)
// ------------------------------------------------------------------------------------------------------
constexpr int ADAM_MAX = 64;
struct AdamTable {
  float* p[ADAM_MAX];
  const float* g[ADAM_MAX];
  float* m[ADAM_MAX];
  float* v[ADAM_MAX];
  int n[ADAM_MAX];
  int count;
};

__global__ __launch_bounds__(256) void adam_kernel(AdamTable t, float lr, float b1, float b2, float eps, float bc1, float bc2s) {
  const int ti = blockIdx.y;
  if (ti >= t.count) return;
  float* p = t.p[ti];
  const float* g = t.g[ti];
  float* m = t.m[ti];
  float* v = t.v[ti];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < t.n[ti]; i += gridDim.x * 256) {
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;       // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2s + eps;          // (exp_avg_sq.sqrt() / sqrt(bias_correction2)).add_(eps)
    p[i] = p[i] - (lr / bc1) * (mi / denom);             // param.addcdiv_(exp_avg, denom, value=-lr/bias_correction1)
  }
}

// The same update with lr and the step count read from DEVICE memory (hyper = [lr, step]: the caller advances `step` with a
// device-side add before the launch): nothing of the launch depends on host state that changes from step to step, so a
// captured optimisation step (hipGraph) replays with the right bias corrections.  bc1 / bc2s per thread: two powf, noise next
// to the 28 bytes of memory traffic per element.
__global__ __launch_bounds__(256) void adam_dev_kernel(AdamTable t, const float* __restrict__ hyper, float b1, float b2, float eps) {
  const int ti = blockIdx.y;
  if (ti >= t.count) return;
  const float lr = hyper[0], step = hyper[1];
  const float bc1 = 1.f - powf(b1, step), bc2s = sqrtf(1.f - powf(b2, step));
  float* p = t.p[ti];
  const float* g = t.g[ti];
  float* m = t.m[ti];
  float* v = t.v[ti];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < t.n[ti]; i += gridDim.x * 256) {
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2s + eps;
    p[i] = p[i] - (lr / bc1) * (mi / denom);
  }
}

}  // namespace

extern "C" int uncl_cgan_loss(const float* real, const float* fake, int N, float w, float* loss, float* g_real, float* g_fake,
                              int accumulate_loss, void* stream) {
  if (!real || !fake || !loss || N <= 0 || N > 4096) return UNCL_ERR_ARG;
  hipLaunchKernelGGL(cgan_kernel, dim3(1), dim3(256), 2 * N * sizeof(float), reinterpret_cast<hipStream_t>(stream), real, fake,
                     N, w, loss, g_real, g_fake, accumulate_loss);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

extern "C" size_t uncl_nce_workspace_bytes(int N) { return ((size_t)N * 256 * 2 + (size_t)N * 2) * sizeof(float); }

// anchor (N,E) ; pos / neg (N,E) or a single row shared by all samples (stride_n = 0).  hw = spatial size the mean runs
// over (E = C*hw).  loss (+)= w * mean_n CE([s(a,p), s(a,n)], 0).  Gradients (optional) are written or accumulated.
extern "C" int uncl_nce_loss(const void* anchor, const void* pos, const void* neg, int dtype, int N, long long E, int hw,
                             int pos_shared, int neg_shared, float k, float c, float w, float* loss, float* g_anchor,
                             float* g_pos, float* g_neg, int accumulate_loss, int accumulate_grad, void* workspace,
                             const int* shared_rows, void* stream) {
  if (!anchor || !loss || !workspace || N <= 0 || E <= 0 || hw <= 0) return UNCL_ERR_ARG;
  if (shared_rows) {        // positive / negative = rows of the anchor picked on the device: loss only (see uncl_nce_backward)
    if (g_anchor || g_pos || g_neg) return UNCL_ERR_ARG;
    pos = neg = anchor; pos_shared = neg_shared = 1;
  }
  if (!pos || !neg) return UNCL_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  float* partial = reinterpret_cast<float*>(workspace);
  const int blocks = (int)((E + 255) / 256 < 256 ? (E + 255) / 256 : 256);
  float* gs = partial + (size_t)N * 256 * 2;
  const size_t ps = pos_shared ? 0 : (size_t)E, qs = neg_shared ? 0 : (size_t)E;
  const bool need_grad = g_anchor || g_pos || g_neg;
  const int gblocks = (int)((E + 255) / 256 < 4096 ? (E + 255) / 256 : 4096);
  if (dtype == UNCL_F32) {
    hipLaunchKernelGGL(nce_sim_kernel<float>, dim3(blocks, N), dim3(256), 0, st, (const float*)anchor, (const float*)pos,
                       (const float*)neg, (size_t)E, ps, qs, k, c, partial, shared_rows);
    hipLaunchKernelGGL(nce_ce_kernel, dim3(1), dim3(256), 0, st, partial, blocks, N, 1.0 / (double)hw, w, loss, gs, accumulate_loss);
    if (need_grad)
      hipLaunchKernelGGL(nce_grad_kernel<float>, dim3(gblocks), dim3(256), 0, st, (const float*)anchor, (const float*)pos,
                         (const float*)neg, (size_t)E, ps, qs, N, k, c, 1.f / (float)hw, gs, g_anchor, g_pos, g_neg, accumulate_grad);
  } else if (dtype == UNCL_BF16) {
    hipLaunchKernelGGL(nce_sim_kernel<bf16_t>, dim3(blocks, N), dim3(256), 0, st, (const bf16_t*)anchor, (const bf16_t*)pos,
                       (const bf16_t*)neg, (size_t)E, ps, qs, k, c, partial, shared_rows);
    hipLaunchKernelGGL(nce_ce_kernel, dim3(1), dim3(256), 0, st, partial, blocks, N, 1.0 / (double)hw, w, loss, gs, accumulate_loss);
    if (need_grad)
      hipLaunchKernelGGL(nce_grad_kernel<bf16_t>, dim3(gblocks), dim3(256), 0, st, (const bf16_t*)anchor, (const bf16_t*)pos,
                         (const bf16_t*)neg, (size_t)E, ps, qs, N, k, c, 1.f / (float)hw, gs, g_anchor, g_pos, g_neg, accumulate_grad);
  } else {
    return UNCL_ERR_ARG;
  }
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

// Gradients of uncl_nce_loss at backward time: `workspace` is the one the forward call filled (it holds d loss / d s per
// sample for w = the forward's w), `upstream` an optional device scalar multiplied in.  grad_dtype = dtype (gradients in the
// feature type) or UNCL_F32.  pos_row / neg_row >= 0 (with pos_shared / neg_shared): the shared row is that row of `anchor`
// and its gradient is folded into g_anchor (g_pos / g_neg must then be NULL); -1 otherwise.
extern "C" int uncl_nce_backward(const void* anchor, const void* pos, const void* neg, int dtype, int N, long long E, int hw,
                                 int pos_shared, int neg_shared, float k, float c, const void* workspace, const float* upstream,
                                 void* g_anchor, void* g_pos, void* g_neg, int grad_dtype, int pos_row, int neg_row,
                                 const int* shared_rows, void* stream) {
  if (!anchor || !workspace || N <= 0 || E <= 0 || hw <= 0) return UNCL_ERR_ARG;
  if (shared_rows) {        // device-chosen rows of the anchor: equivalent to pos_row / neg_row read from device memory
    if (g_pos || g_neg) return UNCL_ERR_ARG;
    pos = neg = anchor; pos_shared = neg_shared = 1; pos_row = neg_row = 0;
  }
  if (!pos || !neg) return UNCL_ERR_ARG;
  if (dtype != UNCL_F32 && dtype != UNCL_BF16) return UNCL_ERR_ARG;
  if (grad_dtype != dtype && grad_dtype != UNCL_F32) return UNCL_ERR_ARG;
  const int V = dtype == UNCL_BF16 ? 8 : 4;
  if (E % V != 0) return UNCL_ERR_ARG;
  for (const void* ptr : {anchor, pos, neg, (const void*)g_anchor, (const void*)g_pos, (const void*)g_neg})
    if (ptr && (reinterpret_cast<uintptr_t>(ptr) & 15)) return UNCL_ERR_ARG;
  if (pos_row >= 0 && (!pos_shared || g_pos || pos_row >= N)) return UNCL_ERR_ARG;
  if (neg_row >= 0 && (!neg_shared || g_neg || neg_row >= N)) return UNCL_ERR_ARG;
  if (pos_row < -1 || neg_row < -1) return UNCL_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const float* gs = reinterpret_cast<const float*>(workspace) + (size_t)N * 256 * 2;
  const size_t ps = pos_shared ? 0 : (size_t)E, qs = neg_shared ? 0 : (size_t)E;
  const size_t nv = (size_t)E / V;
  const int blocks = (int)((nv + 255) / 256 < 8192 ? (nv + 255) / 256 : 8192);
  const float inv_hw = 1.f / (float)hw;
#define UNCL_NCE_BWD(T, GT)                                                                                                   \
  hipLaunchKernelGGL((nce_bwd_kernel<T, GT>), dim3(blocks), dim3(256), 0, st, (const T*)anchor, (const T*)pos, (const T*)neg, \
                     (size_t)E, ps, qs, N, k, c, inv_hw, gs, upstream, (GT*)g_anchor, (GT*)g_pos, (GT*)g_neg, pos_row, neg_row, shared_rows)
  if (dtype == UNCL_F32) UNCL_NCE_BWD(float, float);
  else if (grad_dtype == UNCL_BF16) UNCL_NCE_BWD(bf16_t, bf16_t);
  else UNCL_NCE_BWD(bf16_t, float);
#undef UNCL_NCE_BWD
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

// The similarity by itself, for NCE forms the fused pair above does not cover (several positives / negatives per anchor,
// GanTrainerImg.py:410-439 with longer lists): sims[n] = {s(a_n, pos_n), s(a_n, neg_n)}; the caller builds its logits from them.
extern "C" int uncl_nce_similarity(const void* anchor, const void* pos, const void* neg, int dtype, int N, long long E, int hw,
                                   int pos_shared, int neg_shared, float k, float c, float* sims, void* workspace, void* stream) {
  if (!anchor || !pos || !neg || !sims || !workspace || N <= 0 || E <= 0 || hw <= 0) return UNCL_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  float* partial = reinterpret_cast<float*>(workspace);
  const int blocks = (int)((E + 255) / 256 < 256 ? (E + 255) / 256 : 256);
  const size_t ps = pos_shared ? 0 : (size_t)E, qs = neg_shared ? 0 : (size_t)E;
  if (dtype == UNCL_F32)
    hipLaunchKernelGGL(nce_sim_kernel<float>, dim3(blocks, N), dim3(256), 0, st, (const float*)anchor, (const float*)pos,
                       (const float*)neg, (size_t)E, ps, qs, k, c, partial, (const int*)nullptr);
  else if (dtype == UNCL_BF16)
    hipLaunchKernelGGL(nce_sim_kernel<bf16_t>, dim3(blocks, N), dim3(256), 0, st, (const bf16_t*)anchor, (const bf16_t*)pos,
                       (const bf16_t*)neg, (size_t)E, ps, qs, k, c, partial, (const int*)nullptr);
  else
    return UNCL_ERR_ARG;
  UNCL_CHECK_LAUNCH();
  hipLaunchKernelGGL(nce_sims_final_kernel, dim3((2 * N + 255) / 256), dim3(256), 0, st, partial, blocks, N, 1.0 / (double)hw, sims);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

// ... and its backward: g_sims[n] = {dL/ds(a_n, pos_n), dL/ds(a_n, neg_n)} (device, fp32) -> fp32 gradients of the three tensors
// (each optional; a shared row receives the sum over the samples), written or accumulated.
extern "C" int uncl_nce_similarity_backward(const void* anchor, const void* pos, const void* neg, int dtype, int N, long long E,
                                            int hw, int pos_shared, int neg_shared, float k, float c, const float* g_sims,
                                            float* g_anchor, float* g_pos, float* g_neg, int accumulate, void* stream) {
  if (!anchor || !pos || !neg || !g_sims || N <= 0 || E <= 0 || hw <= 0) return UNCL_ERR_ARG;
  if (!g_anchor && !g_pos && !g_neg) return UNCL_OK;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t ps = pos_shared ? 0 : (size_t)E, qs = neg_shared ? 0 : (size_t)E;
  const int gblocks = (int)((E + 255) / 256 < 4096 ? (E + 255) / 256 : 4096);
  if (dtype == UNCL_F32)
    hipLaunchKernelGGL(nce_grad_kernel<float>, dim3(gblocks), dim3(256), 0, st, (const float*)anchor, (const float*)pos,
                       (const float*)neg, (size_t)E, ps, qs, N, k, c, 1.f / (float)hw, g_sims, g_anchor, g_pos, g_neg, accumulate);
  else if (dtype == UNCL_BF16)
    hipLaunchKernelGGL(nce_grad_kernel<bf16_t>, dim3(gblocks), dim3(256), 0, st, (const bf16_t*)anchor, (const bf16_t*)pos,
                       (const bf16_t*)neg, (size_t)E, ps, qs, N, k, c, 1.f / (float)hw, g_sims, g_anchor, g_pos, g_neg, accumulate);
  else
    return UNCL_ERR_ARG;
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

// out[0] = sum_i weights[i] * terms[i][0]   (terms: host array of n device pointers, n <= UNCL_WSUM_MAX)
extern "C" int uncl_weighted_sum(const float* const* terms, const float* weights, int n, float* out, void* stream) {
  if (!terms || !weights || !out || n <= 0 || n > UNCL_WSUM_MAX) return UNCL_ERR_ARG;
  WSumArgs t = {};
  t.n = n;
  for (int i = 0; i < n; ++i) {
    if (!terms[i]) return UNCL_ERR_ARG;
    t.term[i] = terms[i]; t.w[i] = weights[i];
  }
  hipLaunchKernelGGL(weighted_sum_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), t, out);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

// out[i] = g[0] * weights[i]: the n input gradients of uncl_weighted_sum for the upstream gradient g (a device scalar)
extern "C" int uncl_weighted_sum_backward(const float* g, const float* weights, int n, float* out, void* stream) {
  if (!g || !weights || !out || n <= 0 || n > UNCL_WSUM_MAX) return UNCL_ERR_ARG;
  WSumArgs t = {};
  t.n = n;
  for (int i = 0; i < n; ++i) t.w[i] = weights[i];
  hipLaunchKernelGGL(weighted_sum_bwd_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), t, g, out);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

extern "C" int uncl_l1_pairs(const float* a, int a_stride, const float* b, int b_stride, int N, float w, float* loss, float* g_a,
                             float* g_b, int accumulate_loss, void* stream) {
  if (!a || !b || !loss || N <= 0) return UNCL_ERR_ARG;
  hipLaunchKernelGGL(l1_pairs_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a, a_stride, b, b_stride, N,
                     w, loss, g_a, g_b, accumulate_loss);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

extern "C" int uncl_l1_to_row(const float* stats, int N, const int* row, float* loss2, float* grad, void* stream) {
  if (!stats || !row || !loss2 || !grad || N <= 0) return UNCL_ERR_ARG;
  hipLaunchKernelGGL(l1_to_row_kernel, dim3(1), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), stats, N, row, loss2, grad);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

extern "C" int uncl_l1_to_row_backward(const float* grad, int N, const float* g_mean_term, const float* g_var_term, float* out,
                                       void* stream) {
  if (!grad || !g_mean_term || !g_var_term || !out || N <= 0) return UNCL_ERR_ARG;
  hipLaunchKernelGGL(l1_to_row_bwd_kernel, dim3((N + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), grad, N,
                     g_mean_term, g_var_term, out);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

// scores[f * n_per_frame + p] for the (frame_h/h) x (frame_w/w) patches of every frame; x fp32 (F, frame_h, frame_w)
extern "C" int uncl_tmqi_naturalness(const float* x, int F, int frame_h, int frame_w, int h, int w, float scale, double* scores,
                                     int32_t* best_worst, void* stream) {
  if (!x || !scores || F <= 0 || h <= 0 || w <= 0 || frame_h % h != 0 || frame_w % w != 0) return UNCL_ERR_ARG;
  const int npf = (frame_h / h) * (frame_w / w);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(tmqi_n_kernel, dim3(F * npf), dim3(256), 0, st, x, npf, h, w, frame_h, frame_w, scale, scores);
  if (best_worst) hipLaunchKernelGGL(argminmax_kernel, dim3(1), dim3(64), 0, st, scores, F * npf, best_worst);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

// gx[n] (+)= gscale[n] * d mean(local variance of x[n]) / dx      x, gx: fp32 (N,H,W)
extern "C" int uncl_gauss_var_backward(const float* x, const float* gscale, float* gx, int N, int H, int W, int accumulate,
                                       void* stream) {
  if (!x || !gscale || !gx || N <= 0 || H < GW || W < GW) return UNCL_ERR_ARG;
  GaussW gw;
  double s = 0.0, g[GW];
  for (int k = 0; k < GW; ++k) { g[k] = exp(-((k - 5) * (k - 5)) / (2.0 * 1.5 * 1.5)); s += g[k]; }
  for (int k = 0; k < GW; ++k) gw.g[k] = (float)(g[k] / s);
  const int tx = (W + GT - 1) / GT, ty = (H + GT - 1) / GT;
  hipLaunchKernelGGL(gauss_var_bwd_kernel, dim3(tx * ty, N), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, gscale, gx,
                     H, W, tx, gw, accumulate);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

// The same backward for the case that matters (the generator's 32-channel bf16 feature map, five times per video step):
// one workgroup per 16 x 16 pixel tile and ALL 32 channels.  The tile with its 10-pixel halo is read as whole 64-byte
// pixels (the one-channel form above touches 2 of every 64 bytes it pulls in, and writes the same way: 440 us for 8 frames),
// transposed into per-channel bf16 planes in LDS, each wave runs the four separable passes for eight channels in its own
// scratch, and the result goes back through LDS as whole pixels.
// CH = channels per workgroup (blockIdx.z = which group of CH), NWAVE = its waves (CH / NWAVE channels each).  <32, 8>: 109 KB of
// LDS, one workgroup per CU, its load phase and its arithmetic never overlap; <16, 4>: 55 KB, two workgroups per CU whose phases do
// (each reads 32 of a pixel's 64 bytes; the other half's reader finds the line in L2).
constexpr int HT = 16, HM = HT + GW - 1 /*26*/, HI = HM + GW - 1 /*36*/;
// row strides of the input planes (bf16: 19 dwords) and of the first pass's output (fp32: 29 dwords), both odd in dwords: the first
// pass walks DOWN the rows with its lanes (consecutive lanes = consecutive rows of one run of four columns), so a half-wave's 32
// rows land on 32 banks (row-major units on 18- / 26-dword rows: 414 LDS cycles per channel and tile in that pass, now ~250)
constexpr int HIP = HI + 2, SHS = HM + 3;
constexpr int GB_PS = HI * HIP + 2, GB_TMPW = HI * SHS + HM * HM;
template <int CH, int NWAVE>
__global__ __launch_bounds__(NWAVE * 64) void gauss_stats_bwd32_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gst,
                                                                bf16_t* __restrict__ gx, int H, int W, int tiles_x, GaussW gw,
                                                                int accumulate) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PS = GB_PS;           // plane stride in elements: 685 dwords (odd), so planes 8 apart sit 8 banks apart
  bf16_t* sx = reinterpret_cast<bf16_t*>(smem);                               // [CH][PS] input planes
  float* tmp = reinterpret_cast<float*>(smem + CH * PS * 2);                  // per wave: sh[HI*HM] (later sv[HT*HM]), smu[HM*HM]
  constexpr int TMPW = GB_TMPW, NTHR = NWAVE * 64, OCT = CH / 8, SOP = HT * HT + 2;   // SOP: result plane stride
  bf16_t* so = reinterpret_cast<bf16_t*>(smem + CH * PS * 2 + NWAVE * TMPW * 4);  // [CH][SOP] result planes
  const int c0 = blockIdx.z * CH;                                             // first channel of this workgroup
  const int n = blockIdx.y;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int y0 = ty * HT, x0 = tx * HT;
  const int Ho = H - (GW - 1), Wo = W - (GW - 1);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bf16_t* xn = x + (size_t)n * H * W * 32 + c0;
  // two adjacent plane positions per thread: every LDS write is a whole dword (channel i of both pixels)
  for (int v = tid; v < (HI * HI / 2) * OCT; v += NTHR) {
    const int pair = v / OCT, c8 = v - pair * OCT;
    bf16x8 val[2];
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int pix = 2 * pair + h2, ly = pix / HI, lx = pix - ly * HI;
      const int gy = y0 - 10 + ly, gxx = x0 - 10 + lx;
#pragma unroll
      for (int i = 0; i < 8; ++i) val[h2][i] = (bf16_t)0.f;
      if (gy >= 0 && gy < H && gxx >= 0 && gxx < W)
        val[h2] = *reinterpret_cast<const bf16x8*>(xn + ((size_t)gy * W + gxx) * 32 + c8 * 8);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      typedef bf16_t bf16x2_t __attribute__((ext_vector_type(2)));
      bf16x2_t pk;
      pk[0] = val[0][i]; pk[1] = val[1][i];
      *reinterpret_cast<bf16x2_t*>(sx + (c8 * 8 + i) * PS + ((2 * pair) / HI) * HIP + (2 * pair) % HI) = pk;
    }
  }
  __syncthreads();
  // the passes of a channel run inside ONE wave on its private scratch: ordering between them needs the wave's own LDS
  // operations to have completed, not a workgroup barrier
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                         __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
  float* sh = tmp + wave * TMPW;
  float* smu = sh + HI * SHS;
  float* sv = sh;                    // the first pass's buffer is dead when the third pass writes
  // Every pass computes runs of four outputs along the filter direction from 14 inputs held in registers (44 LDS reads
  // become 14).  Unit -> (line, run) maps and the edge weights of the last pass are the same for every channel.
  constexpr int R = 4, NRM = (HM + R - 1) / R /*7 runs across 26*/, NRT = HT / R /*4 runs across 16*/;
  // last pass: this lane's run (row py, columns 4*pr .. +3) and the window-count weights wy, wx of its four pixels
  const int py = lane / NRT, pr = lane - py * NRT;
  float wy4 = 0.f, wx4[R] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < GW; ++t) {
    const int gyv = y0 + py;
    if (gyv - t >= 0 && gyv - t < Ho) wy4 += gw.g[t];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int gxv = x0 + pr * R + r;
      if (gxv - t >= 0 && gxv - t < Wo) wx4[r] += gw.g[t];
    }
  }
  for (int ci = 0; ci < CH / NWAVE; ++ci) {
    const int ch = wave + NWAVE * ci;
    const bf16_t* sc_ = sx + ch * PS;
    // 1: sh[ly][lx] = sum_t g[t] x[ly][lx + t]                     (HI rows x HM columns)
    for (int u = lane; u < HI * NRM; u += 64) {      // unit = (run of four columns, row), rows fastest
      const int run = u / HI, ly = u - run * HI, lx0 = run * R;
      float in[R + GW - 1];
#pragma unroll
      for (int j = 0; j < R + GW - 1; ++j) in[j] = (float)sc_[ly * HIP + min(lx0 + j, HI - 1)];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < GW; ++t) acc = fmaf(gw.g[t], in[r + t], acc);
        if (lx0 + r < HM) sh[ly * SHS + lx0 + r] = acc;
      }
    }
    WAVE_SYNC();
    // 2: mu[ly][lx] = sum_t g[t] sh[ly + t][lx] at valid window positions, else 0        (HM x HM)
    for (int u = lane; u < NRM * HM; u += 64) {
      const int lr_ = u / HM, lx = u - lr_ * HM, ly0 = lr_ * R;
      float in[R + GW - 1];
#pragma unroll
      for (int j = 0; j < R + GW - 1; ++j) in[j] = sh[min(ly0 + j, HI - 1) * SHS + lx];
      const int ox = x0 - 10 + lx;
      const bool xok = ox >= 0 && ox < Wo;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < GW; ++t) acc = fmaf(gw.g[t], in[r + t], acc);
        const int oy = y0 - 10 + ly0 + r;
        if (ly0 + r < HM) smu[(ly0 + r) * HM + lx] = (xok && oy >= 0 && oy < Ho) ? acc : 0.f;
      }
    }
    WAVE_SYNC();
    // 3: sv[ly][lx] = sum_t g[t] mu[ly + 10 - t][lx]                (HT rows x HM columns)
    for (int u = lane; u < NRT * HM; u += 64) {
      const int lr_ = u / HM, lx = u - lr_ * HM, ly0 = lr_ * R;
      float in[R + GW - 1];          // mu rows ly0 .. ly0 + 13
#pragma unroll
      for (int j = 0; j < R + GW - 1; ++j) in[j] = smu[(ly0 + j) * HM + lx];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < GW; ++t) acc = fmaf(gw.g[t], in[r + 10 - t], acc);
        sv[(ly0 + r) * HM + lx] = acc;
      }
    }
    WAVE_SYNC();
    // 4: s[ly][lx] = sum_t g[t] sv[ly][lx + 10 - t]; gradient = g_mean + sc (x wy wx - s)
    const float gm = gst[((size_t)n * 2 + 0) * 32 + c0 + ch] / ((float)H * (float)W);
    const float scv = gst[((size_t)n * 2 + 1) * 32 + c0 + ch] * 2.f / ((float)Ho * (float)Wo);
    {
      float in[R + GW - 1];          // sv columns 4 pr .. 4 pr + 13 of row py
#pragma unroll
      for (int j = 0; j < R + GW - 1; ++j) in[j] = sv[py * HM + pr * R + j];
      bf16_t res[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < GW; ++t) acc = fmaf(gw.g[t], in[r + 10 - t], acc);
        const int lx = pr * R + r;
        res[r] = (bf16_t)(gm + scv * ((float)sc_[(py + 10) * HIP + lx + 10] * wy4 * wx4[r] - acc));
      }
      // results as per-channel PLANES, two pixels per dword write (pixel-major [pixel][channel] rows put all 64 lanes of a write on
      // ONE bank -- pixels four apart are 64 dwords apart: SQ_LDS_BANK_CONFLICT was 64 % of the kernel's LDS cycles)
      typedef bf16_t bf16x2_t __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int r = 0; r < R; r += 2) {
        bf16x2_t pk;
        pk[0] = res[r]; pk[1] = res[r + 1];
        *reinterpret_cast<bf16x2_t*>(so + ch * SOP + py * HT + pr * R + r) = pk;
      }
    }
    WAVE_SYNC();
  }
  __syncthreads();
#undef WAVE_SYNC
  for (int v = tid; v < HT * HT * OCT; v += NTHR) {
    const int pix = v / OCT, c8 = v - pix * OCT, ly = pix / HT, lx = pix - ly * HT;
    const int gy = y0 + ly, gxx = x0 + lx;
    if (gy < H && gxx < W) {
      bf16_t* d = gx + ((size_t)n * H * W + (size_t)gy * W + gxx) * 32 + c0 + c8 * 8;
      bf16x8 val;
#pragma unroll
      for (int i = 0; i < 8; ++i) val[i] = so[(c8 * 8 + i) * SOP + pix];
      if (accumulate) {
        const bf16x8 old = *reinterpret_cast<const bf16x8*>(d);
#pragma unroll
        for (int i = 0; i < 8; ++i) val[i] = (bf16_t)((float)old[i] + (float)val[i]);
      }
      *reinterpret_cast<bf16x8*>(d) = val;
    }
  }
}

// The same gradient in TWO passes (round 6).  s = G_y^T G_x^T M G_x G_y x with M the mask of valid window positions, and M is a
// product of a row mask and a column mask, so s = A_y A_x x with the (H x H) band matrix A = G^T G of one axis:
//   A[i][i + d] = sum_u g[u] g[u + d],  u from max(0, -d, i - Ho + 1) to min(10, 10 - d, i)                       (|d| <= 10)
// -- the 21-tap autocorrelation of the window wherever i is ten or more pixels from both ends, ten truncated rows at each end
// (GaussC::bt; the high end is the low end mirrored, the window is symmetric).  Two passes of 21 taps instead of four of 11:
// per channel and 16 x 16 tile 17.5 k multiply-adds instead of 25.1 k, one fp32 intermediate (36 x 16) through LDS instead of
// three, and the inputs of a run of four outputs read as 12 dwords (two 16-bit pixels each) instead of 14 single elements.
// Interior tiles take the weights from scalar registers; a tile that touches an end of the axis reads a per-position row from
// a small LDS table built at the start.  A wave works on TWO of its channels at a time (288 + 128 units over 64 lanes).
// Measured (8 samples): 148 -> 144 us with scalar multiply-adds, 137 with the packed ones of gc_taps4 -- of which 42 us are the
// load and store phases (-DUNCL_GC_ABL=1); the rest is instruction issue at two waves per SIMD (the planes' 44 KB per workgroup
// allow no third workgroup per CU).
struct GaussC { float c[2 * GW - 1]; float bt[GW - 1][2 * GW - 1]; };
constexpr int CT = 2 * GW - 1 /*21 taps*/, CTP = 20 /*row pitch of the intermediate in floats: 4 CTP = 16 (mod 64) banks per run of rows*/;
constexpr int GC_T = 2 * HI * CTP;             // floats of intermediate per wave (two channels)
// four outputs of a 21-tap filter from 24 inputs: out[r] = sum_t w[t] in[r + t].  Packed fp32 multiply-adds on the ALIGNED input
// pairs (in[2k], in[2k+1]) only: an even tap feeds output pairs (0,1), (2,3); an odd tap the pair (1,2) plus two single
// multiply-adds for outputs 0 and 3 -- 52 instructions instead of 84.  Even and odd taps accumulate separately (fixed order).
typedef float gc_f2 __attribute__((ext_vector_type(2)));
template <typename WF>
__device__ __forceinline__ void gc_taps4(const float (&in)[24], WF w, float (&out)[4]) {
  gc_f2 e01 = {0.f, 0.f}, e23 = {0.f, 0.f}, o12 = {0.f, 0.f};
  float o0 = 0.f, o3 = 0.f;
#pragma unroll
  for (int t = 0; t < 21; t += 2) {
    const float wt = w(t);
    const gc_f2 wv = {wt, wt};
    e01 = __builtin_elementwise_fma(wv, gc_f2{in[t], in[t + 1]}, e01);
    e23 = __builtin_elementwise_fma(wv, gc_f2{in[t + 2], in[t + 3]}, e23);
  }
#pragma unroll
  for (int t = 1; t < 21; t += 2) {
    const float wt = w(t);
    const gc_f2 wv = {wt, wt};
    o12 = __builtin_elementwise_fma(wv, gc_f2{in[t + 1], in[t + 2]}, o12);
    o0 = fmaf(wt, in[t], o0);
    o3 = fmaf(wt, in[t + 3], o3);
  }
  out[0] = e01[0] + o0;
  out[1] = e01[1] + o12[0];
  out[2] = e23[0] + o12[1];
  out[3] = e23[1] + o3;
}
template <int CH, int NWAVE>
__global__ __launch_bounds__(NWAVE * 64) void gauss_stats_bwd32c_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gst,
                                                                 bf16_t* __restrict__ gx, int H, int W, int tiles_x, GaussW gw, GaussC gc,
                                                                 int accumulate) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PS = GB_PS, NTHR = NWAVE * 64, OCT = CH / 8, SOP = HT * HT + 2;
  bf16_t* sx = reinterpret_cast<bf16_t*>(smem);                                            // [CH][PS] input planes
  float* tmp = reinterpret_cast<float*>(smem + CH * PS * 2);                               // per wave: T[2][HI][CTP]
  bf16_t* so = reinterpret_cast<bf16_t*>(smem + CH * PS * 2 + NWAVE * GC_T * 4);           // [CH][SOP] result planes
  float* wtab = reinterpret_cast<float*>(smem + CH * PS * 2 + NWAVE * GC_T * 4 + CH * SOP * 2);   // [2][HT][CT]: x rows, then y rows
  const int c0 = blockIdx.z * CH;
  const int n = blockIdx.y;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int y0 = ty * HT, x0 = tx * HT;
  const int Ho = H - (GW - 1), Wo = W - (GW - 1);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bf16_t* xn = x + (size_t)n * H * W * 32 + c0;
  // interior along an axis: every output position of the tile is >= 10 from the low end and <= L - 11
  const bool xin = x0 >= GW - 1 && x0 + HT - 1 <= W - GW, yin = y0 >= GW - 1 && y0 + HT - 1 <= H - GW;
  for (int v = tid; v < (HI * HI / 2) * OCT; v += NTHR) {
    const int pair = v / OCT, c8 = v - pair * OCT;
    bf16x8 val[2];
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int pix = 2 * pair + h2, ly = pix / HI, lx = pix - ly * HI;
      const int gy = y0 - 10 + ly, gxx = x0 - 10 + lx;
#pragma unroll
      for (int i = 0; i < 8; ++i) val[h2][i] = (bf16_t)0.f;
      if (gy >= 0 && gy < H && gxx >= 0 && gxx < W)
        val[h2] = *reinterpret_cast<const bf16x8*>(xn + ((size_t)gy * W + gxx) * 32 + c8 * 8);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      typedef bf16_t bf16x2_t __attribute__((ext_vector_type(2)));
      bf16x2_t pk;
      pk[0] = val[0][i]; pk[1] = val[1][i];
      *reinterpret_cast<bf16x2_t*>(sx + (c8 * 8 + i) * PS + ((2 * pair) / HI) * HIP + (2 * pair) % HI) = pk;
    }
  }
  if (!xin || !yin) {
    // per-position weight rows of this tile: position i of an axis of length L takes row i of the truncated table at the low end,
    // row L - 1 - i mirrored at the high end, the autocorrelation in between
    for (int v = tid; v < 2 * HT * CT; v += NTHR) {
      const int ax = v / (HT * CT), r = v - ax * (HT * CT), pos = r / CT, d = r - pos * CT;
      const int L = ax == 0 ? W : H, i = (ax == 0 ? x0 : y0) + pos;
      float w = gc.c[d];
      if (i < GW - 1) w = gc.bt[i][d];
      else if (i > L - GW && i < L) w = gc.bt[L - 1 - i][CT - 1 - d];
      wtab[v] = w;
    }
  }
  __syncthreads();
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                         __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
  float* T = tmp + wave * GC_T;
  constexpr int R = 4, NRT = HT / R /*4 runs across 16*/, UA = HI * NRT /*144 units of the first pass per channel*/;
  // second pass: this lane's column pc and rows 4 pq .. 4 pq + 3; the window-count weights of its pixels
  const int pc = lane & (HT - 1), pq = lane >> 4;
  float wx1 = 0.f, wy4[R] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < GW; ++t) {
    const int gxv = x0 + pc;
    if (gxv - t >= 0 && gxv - t < Wo) wx1 += gw.g[t];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int gyv = y0 + pq * R + r;
      if (gyv - t >= 0 && gyv - t < Ho) wy4[r] += gw.g[t];
    }
  }
#ifndef UNCL_GC_ABL
#define UNCL_GC_ABL 0        // timing only: 1 no passes, 2 no first pass, 4 no second pass
#endif
  for (int cp = 0; cp < ((UNCL_GC_ABL & 1) ? 0 : CH / NWAVE); cp += 2) {
    // 1: T[p][ly][c] = sum_d A_x[x0 + c][x0 + c + d] x[ly][c + 10 + d]          (HI rows x HT columns, two channels)
    for (int u = lane; u < ((UNCL_GC_ABL & 2) ? 0 : 2 * UA); u += 64) {
      const int p = u >= UA ? 1 : 0, v = u - p * UA;
      const int run = v / HI, ly = v - run * HI, lx0 = run * R;
      const bf16_t* sc_ = sx + (wave + NWAVE * (cp + p)) * PS + ly * HIP + lx0;
      float in[R + CT - 1];
#pragma unroll
      for (int j = 0; j < (R + CT - 1) / 2; ++j) {
        const unsigned w2 = *reinterpret_cast<const unsigned*>(sc_ + 2 * j);
        in[2 * j] = __builtin_bit_cast(float, w2 << 16);
        in[2 * j + 1] = __builtin_bit_cast(float, w2 & 0xFFFF0000u);
      }
      float* tp = T + (p * HI + ly) * CTP + lx0;
      if (xin) {
        float o[R];
        gc_taps4(in, [&](int t) __attribute__((always_inline)) { return gc.c[t]; }, o);
#pragma unroll
        for (int r = 0; r < R; ++r) tp[r] = o[r];
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float* wr = wtab + (lx0 + r) * CT;
          float acc = 0.f;
#pragma unroll
          for (int t = 0; t < CT; ++t) acc = fmaf(wr[t], in[r + t], acc);
          tp[r] = acc;
        }
      }
    }
    WAVE_SYNC();
    // 2: s[py][pc] = sum_d A_y[y0 + py][y0 + py + d] T[py + 10 + d][pc]; gradient = g_mean + sc (x wy wx - s)
#pragma unroll
    for (int p = 0; p < ((UNCL_GC_ABL & 4) ? 0 : 2); ++p) {
      const int ch = wave + NWAVE * (cp + p);
      const float gm = gst[((size_t)n * 2 + 0) * 32 + c0 + ch] / ((float)H * (float)W);
      const float scv = gst[((size_t)n * 2 + 1) * 32 + c0 + ch] * 2.f / ((float)Ho * (float)Wo);
      float in[R + CT - 1];
      const float* tp = T + (p * HI + pq * R) * CTP + pc;
#pragma unroll
      for (int j = 0; j < R + CT - 1; ++j) in[j] = tp[j * CTP];
      const bf16_t* sc_ = sx + ch * PS;
      float s4[R];
      if (yin) {
        gc_taps4(in, [&](int t) __attribute__((always_inline)) { return gc.c[t]; }, s4);
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float* wr = wtab + (HT + pq * R + r) * CT;
          float acc = 0.f;
#pragma unroll
          for (int t = 0; t < CT; ++t) acc = fmaf(wr[t], in[r + t], acc);
          s4[r] = acc;
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int py = pq * R + r;
        so[ch * SOP + py * HT + pc] = (bf16_t)(gm + scv * ((float)sc_[(py + 10) * HIP + pc + 10] * wy4[r] * wx1 - s4[r]));
      }
    }
    WAVE_SYNC();
  }
  __syncthreads();
#undef WAVE_SYNC
  for (int v = tid; v < HT * HT * OCT; v += NTHR) {
    const int pix = v / OCT, c8 = v - pix * OCT, ly = pix / HT, lx = pix - ly * HT;
    const int gy = y0 + ly, gxx = x0 + lx;
    if (gy < H && gxx < W) {
      bf16_t* d = gx + ((size_t)n * H * W + (size_t)gy * W + gxx) * 32 + c0 + c8 * 8;
      bf16x8 val;
#pragma unroll
      for (int i = 0; i < 8; ++i) val[i] = so[(c8 * 8 + i) * SOP + pix];
      if (accumulate) {
        const bf16x8 old = *reinterpret_cast<const bf16x8*>(d);
#pragma unroll
        for (int i = 0; i < 8; ++i) val[i] = (bf16_t)((float)old[i] + (float)val[i]);
      }
      *reinterpret_cast<bf16x8*>(d) = val;
    }
  }
}

// x, gx: NHWC (N,H,W,C) in dtype; g_stats: fp32 (N,2,C) = [d/d mean, d/d mean-local-variance] per (sample, channel)
extern "C" int uncl_gauss_stats_backward(const void* x, int dtype, const float* g_stats, void* gx, int N, int H, int W, int C,
                                         int accumulate, void* stream) {
  if (!x || !g_stats || !gx || N <= 0 || C <= 0 || H < GW || W < GW) return UNCL_ERR_ARG;
  if (dtype != UNCL_BF16 && dtype != UNCL_F32) return UNCL_ERR_ARG;
  GaussW gw;
  double s = 0.0, g[GW];
  for (int k = 0; k < GW; ++k) { g[k] = exp(-((k - 5) * (k - 5)) / (2.0 * 1.5 * 1.5)); s += g[k]; }
  for (int k = 0; k < GW; ++k) gw.g[k] = (float)(g[k] / s);
  const int tx = (W + GT - 1) / GT, ty = (H + GT - 1) / GT;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (dtype == UNCL_BF16 && C == 32) {
    constexpr size_t lds32 = (size_t)32 * GB_PS * 2 + (size_t)8 * GB_TMPW * 4 + (size_t)(HT * HT + 2) * 32 * 2;
    constexpr size_t lds16 = (size_t)16 * GB_PS * 2 + (size_t)4 * GB_TMPW * 4 + (size_t)(HT * HT + 2) * 16 * 2;
    static_assert(lds32 <= 160 * 1024, "the 32-channel form must fit one CU's LDS");
    static UnclDevOnce attr;
    if (attr.need()) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(gauss_stats_bwd32_kernel<32, 8>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds32) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(gauss_stats_bwd32_kernel<16, 4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds16) != hipSuccess)
        return UNCL_ERR_LAUNCH;
      attr.done();
    }
    const int tx16 = (W + HT - 1) / HT, ty16 = (H + HT - 1) / HT;
    // two 21-tap passes (default; needs both ends of an axis ten pixels apart: H, W >= 21) or the four 11-tap passes
    // (UNCL_GAUSS_BWD_FORM=0; 8 / 32 samples at 256 x 256: 148 / 532 us against 137 / 476; load + store alone: 42 us at 8 samples)
    static const int form = [] { const char* e = getenv("UNCL_GAUSS_BWD_FORM"); return e ? atoi(e) : 1; }();
    if (form && H >= 2 * GW - 1 && W >= 2 * GW - 1) {
      constexpr size_t ldsc = (size_t)16 * GB_PS * 2 + (size_t)4 * GC_T * 4 + (size_t)(HT * HT + 2) * 16 * 2 + (size_t)2 * HT * CT * 4;
      static_assert(2 * ldsc <= 160 * 1024, "two workgroups per CU");
      static UnclDevOnce attrc;
      if (attrc.need()) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gauss_stats_bwd32c_kernel<16, 4>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsc) != hipSuccess)
          return UNCL_ERR_LAUNCH;
        attrc.done();
      }
      GaussC gc;
      double gf[GW];
      for (int k = 0; k < GW; ++k) gf[k] = (double)gw.g[k];             // the fp32 window the other kernels multiply with
      for (int d = -(GW - 1); d <= GW - 1; ++d) {
        for (int i = 0; i < GW; ++i) {                                   // i = GW - 1: the untruncated row = the autocorrelation
          double a = 0.0;
          for (int u = 0; u < GW; ++u)
            if (u + d >= 0 && u + d < GW && u <= i) a += gf[u] * gf[u + d];
          if (i < GW - 1) gc.bt[i][d + GW - 1] = (float)a;
          else gc.c[d + GW - 1] = (float)a;
        }
      }
      hipLaunchKernelGGL((gauss_stats_bwd32c_kernel<16, 4>), dim3(tx16 * ty16, N, 2), dim3(256), ldsc, st, (const bf16_t*)x, g_stats,
                         (bf16_t*)gx, H, W, tx16, gw, gc, accumulate);
      UNCL_CHECK_LAUNCH();
      return UNCL_OK;
    }
    static const int half_on = [] { const char* e = getenv("UNCL_GAUSS_BWD_HALF"); return e ? atoi(e) : 1; }();   // 0: one 32-channel workgroup per tile (8 / 32 samples: 151 / 575 us against 148 / 541)
    if (half_on)
      hipLaunchKernelGGL((gauss_stats_bwd32_kernel<16, 4>), dim3(tx16 * ty16, N, 2), dim3(256), lds16, st, (const bf16_t*)x, g_stats,
                         (bf16_t*)gx, H, W, tx16, gw, accumulate);
    else
      hipLaunchKernelGGL((gauss_stats_bwd32_kernel<32, 8>), dim3(tx16 * ty16, N), dim3(512), lds32, st, (const bf16_t*)x, g_stats,
                         (bf16_t*)gx, H, W, tx16, gw, accumulate);
  } else if (dtype == UNCL_BF16)
    hipLaunchKernelGGL(gauss_stats_bwd_kernel<bf16_t>, dim3(tx * ty, N, C), dim3(256), 0, st, (const bf16_t*)x, g_stats, (bf16_t*)gx,
                       H, W, C, tx, gw, accumulate);
  else
    hipLaunchKernelGGL(gauss_stats_bwd_kernel<float>, dim3(tx * ty, N, C), dim3(256), 0, st, (const float*)x, g_stats, (float*)gx, H,
                       W, C, tx, gw, accumulate);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

extern "C" int uncl_add_per_sample_const(float* g, const float* scale, long long per, int N, float mul, int accumulate,
                                         void* stream) {
  if (!g || !scale || per <= 0 || N <= 0) return UNCL_ERR_ARG;
  const size_t total = (size_t)per * N;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(add_const_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g, scale, (size_t)per, N,
                     mul, accumulate);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

// loss (+)= w * TV(x); gx (+)= w * dTV/dx.  workspace: 1024 floats
extern "C" int uncl_tv_loss(const float* x, int N, int H, int W, float w, float* loss, float* gx, int accumulate_loss,
                            int accumulate_grad, void* workspace, void* stream) {
  if (!x || !loss || !workspace || N <= 0 || H < 2 || W < 2) return UNCL_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t total = (size_t)N * H * W;
  const int blocks = (int)((total + 255) / 256 < 1024 ? (total + 255) / 256 : 1024);
  const float ch = w * 2.f / ((float)(H - 1) * W) / N, cw = w * 2.f / ((float)H * (W - 1)) / N;
  hipLaunchKernelGGL(tv_kernel, dim3(blocks), dim3(256), 0, st, x, gx, (float*)workspace, N, H, W, ch, cw, accumulate_grad);
  hipLaunchKernelGGL(sum_f_kernel, dim3(1), dim3(64), 0, st, (const float*)workspace, blocks, loss, accumulate_loss);
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

// ptrs: HOST arrays of `count` device pointers (params, grads, exp_avg, exp_avg_sq) and element counts
extern "C" int uncl_adam_step(void* const* params, void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                              const int* numel, int count, float lr, float beta1, float beta2, float eps, int step,
                              void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !numel || count <= 0 || step <= 0) return UNCL_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const float bc1 = 1.f - powf(beta1, (float)step), bc2s = sqrtf(1.f - powf(beta2, (float)step));
  for (int base = 0; base < count; base += ADAM_MAX) {
    AdamTable t;
    t.count = count - base < ADAM_MAX ? count - base : ADAM_MAX;
    int mx = 0;
    for (int i = 0; i < t.count; ++i) {
      t.p[i] = (float*)params[base + i]; t.g[i] = (const float*)grads[base + i];
      t.m[i] = (float*)exp_avg[base + i]; t.v[i] = (float*)exp_avg_sq[base + i];
      t.n[i] = numel[base + i];
      if (t.n[i] > mx) mx = t.n[i];
    }
    const int bx = (mx + 255) / 256 < 1024 ? (mx + 255) / 256 : 1024;
    hipLaunchKernelGGL(adam_kernel, dim3(bx, t.count), dim3(256), 0, st, t, lr, beta1, beta2, eps, bc1, bc2s);
  }
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}

// hyper: DEVICE float[2] = {lr, step}; step >= 1 when the kernel runs (the caller adds 1 on the stream before this call)
extern "C" int uncl_adam_step_dev(void* const* params, void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                                  const int* numel, int count, const float* hyper, float beta1, float beta2, float eps, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !numel || !hyper || count <= 0) return UNCL_ERR_ARG;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  for (int base = 0; base < count; base += ADAM_MAX) {
    AdamTable t;
    t.count = count - base < ADAM_MAX ? count - base : ADAM_MAX;
    int mx = 0;
    for (int i = 0; i < t.count; ++i) {
      t.p[i] = (float*)params[base + i]; t.g[i] = (const float*)grads[base + i];
      t.m[i] = (float*)exp_avg[base + i]; t.v[i] = (float*)exp_avg_sq[base + i];
      t.n[i] = numel[base + i];
      if (t.n[i] > mx) mx = t.n[i];
    }
    const int bx = (mx + 255) / 256 < 1024 ? (mx + 255) / 256 : 1024;
    hipLaunchKernelGGL(adam_dev_kernel, dim3(bx, t.count), dim3(256), 0, st, t, hyper, beta1, beta2, eps);
  }
  UNCL_CHECK_LAUNCH();
  return UNCL_OK;
}
