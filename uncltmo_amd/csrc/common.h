// Shared device/host helpers for the gfx950 kernels of the UnCLTMO hot path.
// MI355X only: 64-wide wavefronts, MFMA 32x32 tiles, 160 KiB LDS per CU.  No CUDA-compat paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <cstdlib>

#include "../../include/uncltmo_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define UNCL_CHECK_LAUNCH()                                   \
  do {                                                        \
    hipError_t e__ = hipGetLastError();                       \
    if (e__ != hipSuccess) return UNCL_ERR_LAUNCH;            \
  } while (0)

// Launch preparation that is a property of (kernel, DEVICE) -- hipFuncSetAttribute(MaxDynamicSharedMemorySize), the CU count --
// is cached per device ordinal, not per process: one process may drive several GPUs (SideStreams in generator.hip does), and a
// flag set while device 0 was current must not skip the attribute on device 1.  Two threads racing on a first call both set
// the (idempotent) attribute.
static inline int uncl_device() {
  int d = 0;
  (void)hipGetDevice(&d);
  return d & 31;
}
struct UnclDevOnce {
  std::atomic<unsigned> mask{0};
  bool need() const { return !((mask.load(std::memory_order_acquire) >> uncl_device()) & 1u); }
  void done() { mask.fetch_or(1u << uncl_device(), std::memory_order_release); }
};
static inline int uncl_cu_count() {
  static std::atomic<int> cus[32];
  const int d = uncl_device();
  int c = cus[d].load(std::memory_order_relaxed);
  if (c == 0) {
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, d) != hipSuccess) return 0;
    c = p.multiProcessorCount;
    cus[d].store(c, std::memory_order_relaxed);
  }
  return c;
}

// ---------------------------------------------------------------------------------------------------------------------
// Checked build (-DUNCL_CHECKED; python -c "import __graft_entry__ as g; g.build_checked()" -> uncltmo_amd/libuncltmo_hip_checked.so):
// every global load / store / atomic of the 3x3 convolution kernels (conv3x3_pc.hip, conv3x3_pipe.hip), the weight-gradient
// kernels (wgrad.hip) and the 2x2 up-conv (upconv2x2.hip) first looks its address up in the table of tensors the launch was
// given (base, bytes -- filled by the launcher from the descriptor's own dimensions); an access outside every tensor bumps a
// device-side counter and records the first address, the source line and the size (uncl_checked_report).  The product build
// compiles all of it away.
// ---------------------------------------------------------------------------------------------------------------------
struct UnclChk {
  const char* lo[20];
  const char* hi[20];
  int n;
  unsigned long long* bad;      // [0] violations, [1] first address, [2] source line, [3] bytes
};
unsigned long long* uncl_chk_record();       // device memory, one record per device (misc_kernels.hip)
static inline void uncl_chk_reset(UnclChk& c) { c.n = 0; c.bad = uncl_chk_record(); }
static inline void uncl_chk_add(UnclChk& c, const void* p, unsigned long long bytes) {
  if (p == nullptr || bytes == 0 || c.n >= 20) return;
  // positive control (tools/checked_soak.py --control): UNCL_CHECKED_SHRINK=k registers every tensor k bytes SHORT, so that the
  // accesses to its last bytes must be reported -- a checked run that reports nothing has then really looked
  static const unsigned long long shrink = [] { const char* e = getenv("UNCL_CHECKED_SHRINK"); return e ? strtoull(e, nullptr, 10) : 0ull; }();
  c.lo[c.n] = reinterpret_cast<const char*>(p);
  c.hi[c.n] = c.lo[c.n] + (bytes > shrink ? bytes - shrink : 0);
  ++c.n;
}
#ifdef UNCL_CHECKED
__device__ __forceinline__ void uncl_chk_access(const UnclChk& c, const void* p, unsigned bytes, int line) {
  const char* q = reinterpret_cast<const char*>(p);
  bool ok = false;
  for (int i = 0; i < c.n; ++i) ok = ok || (q >= c.lo[i] && q + bytes <= c.hi[i]);
  if (!ok && c.bad != nullptr) {
    if (atomicAdd(c.bad, 1ull) == 0ull) {
      c.bad[1] = (unsigned long long)(uintptr_t)q;
      c.bad[2] = (unsigned long long)line;
      c.bad[3] = bytes;
    }
  }
}
#define UNCL_CHK(c, p, bytes) uncl_chk_access((c), (p), (bytes), __LINE__)
#define UNCL_CHK_MEMBER UnclChk chk;
#else
#define UNCL_CHK(c, p, bytes) ((void)0)
#define UNCL_CHK_MEMBER
#endif

// Packed 3x3 weights of the 16-bit kernels are K-chunk-major, [Cin / 32][9][Cout][32] (misc_kernels.hip, pack_weight_kernel):
// element (tap, co, ci) of a layer with Cout output channels
__host__ __device__ __forceinline__ bool uncl_w3_chunk_major(size_t elem_bytes, int kk, int Cin) {
  return elem_bytes == 2 && kk == 9 && (Cin & 31) == 0;
}
__host__ __device__ __forceinline__ size_t uncl_w3_index(int tap, int co, int ci, int Cout) {
  return (((size_t)(ci >> 5) * 9 + tap) * Cout + co) * 32 + (ci & 31);
}

// Element traits: a "vec" is always 16 bytes, the unit every loader / LDS access moves.
template <typename T>
struct Elem;

template <>
struct Elem<float> {
  typedef f32x4 vec;
  static constexpr int EPV = 4;   // elements per 16-byte vector
  static constexpr int KC = 16;   // channels per 64-byte K-chunk
  static __device__ __forceinline__ void unpack(const vec& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = v[i];
  }
  static __device__ __forceinline__ vec pack(const float* f) {
    vec v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = f[i];
    return v;
  }
  static __device__ __forceinline__ vec zero() { return vec{0.f, 0.f, 0.f, 0.f}; }
};

template <>
struct Elem<bf16_t> {
  typedef bf16x8 vec;
  typedef bf16x4 vec4;
  static constexpr int EPV = 8;
  static constexpr int KC = 32;
  static __device__ __forceinline__ void unpack(const vec& v, float* f) {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
  }
  static __device__ __forceinline__ vec pack(const float* f) {
    vec v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16_t)f[i];
    return v;
  }
  static __device__ __forceinline__ vec zero() {
    vec v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (bf16_t)0.f;
    return v;
  }
};

// fp16 (BASELINE configs[4]: 4K inference): same 16-byte vectors, same kernels, v_mfma_f32_32x32x16_f16
template <>
struct Elem<f16_t> {
  typedef f16x8 vec;
  typedef f16x4 vec4;
  static constexpr int EPV = 8;
  static constexpr int KC = 32;
  static __device__ __forceinline__ void unpack(const vec& v, float* f) {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
  }
  static __device__ __forceinline__ vec pack(const float* f) {
    vec v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (f16_t)f[i];
    return v;
  }
  static __device__ __forceinline__ vec zero() {
    vec v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (f16_t)0.f;
    return v;
  }
};

// one 32x32x16 matrix instruction on eight 16-bit elements per lane and operand, fp32 accumulators
__device__ __forceinline__ f32x16 mfma32x16(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32x16(const f16x8& a, const f16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
// 16-bit dtype code -> is it one of the two MFMA element types
static inline bool uncl_is_h16(int dtype) { return dtype == UNCL_BF16 || dtype == UNCL_F16; }

__device__ __forceinline__ float uncl_gelu(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float uncl_sigmoid(float x) { return 1.f / (1.f + __expf(-x)); }

__device__ __forceinline__ float uncl_act(float v, int act) {
  switch (act) {
    case UNCL_ACT_RELU: return v > 0.f ? v : 0.f;
    case UNCL_ACT_LRELU: return v > 0.f ? v : 0.2f * v;
    case UNCL_ACT_GELU: return uncl_gelu(v);
    case UNCL_ACT_SIGMOID: return uncl_sigmoid(v);
    case UNCL_ACT_TANH: return tanhf(v);
    case UNCL_ACT_MSIG: return 1.f / (1.f + __expf(-3.f * v));
    default: return v;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

