// Shared device/host helpers for the gfx950 kernels of the UnCLTMO hot path.
// MI355X only: 64-wide wavefronts, MFMA 32x32 tiles, 160 KiB LDS per CU.  No CUDA-compat paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

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

