// bf16.h -- bf16 helpers and MFMA fragment types for gfx950.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace se {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// plain casts: hipcc emits v_cvt_pk_bf16_f32 (round to nearest even, NaN preserved)
__device__ __forceinline__ uint16_t f2bf(float x) {
  __bf16 h = (__bf16)x;
  return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  bf16x2 v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

__device__ __forceinline__ float gelu_erf(float v) { return v * 0.5f * (1.0f + erff(v * 0.70710678118654752f)); }

}  // namespace se
