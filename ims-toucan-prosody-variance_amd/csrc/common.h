// Internal helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/toucan_tts.h"

namespace tts {

void set_error(const char* fmt, ...);

#define TTS_CHECK_ARG(cond, ...)   \
  do {                             \
    if (!(cond)) {                 \
      tts::set_error(__VA_ARGS__); \
      return TTS_E_ARG;            \
    }                              \
  } while (0)

inline int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return TTS_E_LAUNCH;
  }
  return TTS_OK;
}

// Raise a kernel's dynamic-LDS limit to the CU's 160 KiB once per (instantiation, device): `mask` is the instantiation's
// static bit set of devices already done.  The first (eager) launch does it, so it never lands inside a stream capture.
inline hipError_t raise_lds_limit(const void* kernel, unsigned long long& mask) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned long long bit = 1ull << (dev & 63);
  if (mask & bit) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) mask |= bit;
  return e;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  const __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: round to nearest even, NaN preserved
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short u) { return __builtin_bit_cast(float, (unsigned int)u << 16); }

// 16-bit element formats of the MFMA paths: bf16 (compute 1) or IEEE fp16 (compute 2, v_mfma_f32_32x32x16_f16).  Both convert
// round-to-nearest-even, two values per instruction (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32).  fp16 does not saturate: the
// callers keep everything that can leave its range (flow state, norm statistics, accumulators) in fp32.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned short f32_to_f16(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
__device__ __forceinline__ float f16_to_f32(unsigned short u) { return (float)__builtin_bit_cast(_Float16, u); }
template <bool F16>
__device__ __forceinline__ unsigned short to16(float f) {
  if constexpr (F16) return f32_to_f16(f);
  else return f32_to_bf16(f);
}
template <bool F16>
__device__ __forceinline__ float from16(unsigned short u) {
  if constexpr (F16) return f16_to_f32(u);
  else return bf16_to_f32(u);
}
// (lo, hi) -> one dword of two 16-bit elements
template <bool F16>
__device__ __forceinline__ unsigned int pack16(float lo, float hi) {
  if constexpr (F16) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, f16x2));
  } else {
    return (unsigned int)f32_to_bf16(lo) | ((unsigned int)f32_to_bf16(hi) << 16);
  }
}
// run-time format (tensors in HBM whose format is a flag of the call, TTS_IO_F16)
__device__ __forceinline__ float load16(unsigned short u, bool f16) { return f16 ? f16_to_f32(u) : bf16_to_f32(u); }
__device__ __forceinline__ unsigned short store16(float v, bool f16) { return f16 ? f32_to_f16(v) : f32_to_bf16(v); }

// v_mfma_f32_32x32x16_{bf16,f16}: same shape, same operand layout, same cycles
template <bool F16>
__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + __expf(-v)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

}  // namespace tts
