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

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  const __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: round to nearest even, NaN preserved
  return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short u) { return __builtin_bit_cast(float, (unsigned int)u << 16); }

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
