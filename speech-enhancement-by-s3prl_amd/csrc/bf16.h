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

// erf by Abramowitz-Stegun 7.1.26 (|abs error| <= 1.5e-7, i.e. below fp32 resolution of the GELU output that is
// then rounded to bf16): 1 rcp + 1 exp2 + 6 fma, no branches -- libm erff costs ~5x as much in the GEMM epilogue.
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-1.44269504088896f * ax * ax);
  const float r = fmaf(-p * t, e, 1.0f);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float v) { return v * 0.5f * (1.0f + erf_as(v * 0.70710678118654752f)); }

// two values at once: the polynomial, the squares and the final scaling go through the packed fp32 pipe (v_pk_fma_f32 /
// v_pk_mul_f32: one instruction for both lanes of the pair); only rcp / exp2 stay per element.  Same arithmetic as gelu_erf.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 v) {
  const f32x2 x = v * 0.70710678118654752f;
  const f32x2 ax = {fabsf(x.x), fabsf(x.y)};
  const f32x2 one = {1.0f, 1.0f};
  const f32x2 d = __builtin_elementwise_fma(ax, (f32x2){0.3275911f, 0.3275911f}, one);
  const f32x2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  f32x2 p = __builtin_elementwise_fma(t, (f32x2){1.061405429f, 1.061405429f}, (f32x2){-1.453152027f, -1.453152027f});
  p = __builtin_elementwise_fma(p, t, (f32x2){1.421413741f, 1.421413741f});
  p = __builtin_elementwise_fma(p, t, (f32x2){-0.284496736f, -0.284496736f});
  p = __builtin_elementwise_fma(p, t, (f32x2){0.254829592f, 0.254829592f});
  const f32x2 q = ax * ax * -1.44269504088896f;
  const f32x2 e = {__builtin_amdgcn_exp2f(q.x), __builtin_amdgcn_exp2f(q.y)};
  const f32x2 r = __builtin_elementwise_fma(p * t * -1.0f, e, one);            // erf(|x|)
  const f32x2 er = {copysignf(r.x, x.x), copysignf(r.y, x.y)};
  return v * 0.5f * (er + one);
}

// GELU for epilogues whose output is ROUNDED TO bf16 (the FFN hidden activation): x * Phi(x) with Phi(x) = 1/2 + xc P(xc^2), xc = x clamped to
// +-3.75, P an even degree-12 polynomial fitted (weighted minimax) to the exact erf form: |Phi error| <= 8.5e-5 everywhere, |gelu error|
// <= 2.7e-4 absolute (3.3e-5 of |x| for large |x|) -- a tenth of the bf16 rounding of the result -- with NO transcendental: 2 v_med3 +
// 9 packed fp32 instructions per PAIR against ~20 incl. two v_rcp + two v_exp for gelu_erf2.  The FFN1 epilogue was 40 % of that GEMM's
// vector issue time (profiles/r02b_pmc_sq_gemm.json).  Exact-erf GELU stays wherever the value is kept in fp32 (spec head, fp32 mode).
__device__ __forceinline__ f32x2 gelu_poly2(f32x2 v) {
  const f32x2 xc = {__builtin_amdgcn_fmed3f(v.x, -3.75f, 3.75f), __builtin_amdgcn_fmed3f(v.y, -3.75f, 3.75f)};
  const f32x2 t = xc * xc;
  f32x2 p = __builtin_elementwise_fma(t, (f32x2){3.912436597e-08f, 3.912436597e-08f}, (f32x2){-2.376253633e-06f, -2.376253633e-06f});
  p = __builtin_elementwise_fma(p, t, (f32x2){6.234780449e-05f, 6.234780449e-05f});
  p = __builtin_elementwise_fma(p, t, (f32x2){-9.441798320e-04f, -9.441798320e-04f});
  p = __builtin_elementwise_fma(p, t, (f32x2){9.362553246e-03f, 9.362553246e-03f});
  p = __builtin_elementwise_fma(p, t, (f32x2){-6.578987092e-02f, -6.578987092e-02f});
  p = __builtin_elementwise_fma(p, t, (f32x2){3.987064660e-01f, 3.987064660e-01f});
  const f32x2 phi = __builtin_elementwise_fma(xc, p, (f32x2){0.5f, 0.5f});
  return v * phi;
}

}  // namespace se
