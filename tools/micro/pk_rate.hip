// pk_rate.hip -- does packed fp32 VALU (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) issue at the rate of the plain forms in a PURE vector
// loop (no MFMA beside it)?  Decides whether the STFT / iSTFT complex arithmetic should be written on packed (re, im) pairs.
// Cycles are real shader cycles (s_memtime around the loop), W one-wave workgroups per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -o tools/micro/pk_rate tools/micro/pk_rate.hip && tools/micro/pk_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(2))) float f2;
constexpr int ITER = 2048;

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* cyc, float seed) {
  f2 v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = f2{seed + i * 0.001f + threadIdx.x * 1e-6f, seed - i * 0.002f};
  const f2 c = {seed, 1.0f - seed * 1e-3f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) asm volatile("v_add_f32 %0, %1, %2" : "=v"(v[i].x) : "v"(v[i].x), "v"(c.x));
      if (MODE == 1) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v[i].x) : "v"(v[i].x), "v"(c.y), "v"(c.x));
      if (MODE == 2) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(c));
      if (MODE == 3) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(c));
      if (MODE == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(v[i]) : "v"(v[i]), "v"(c), "v"(c));
      if (MODE == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(v[i]) : "v"(v[i]), "v"(c), "v"(c));
      if (MODE == 6) asm volatile("v_sqrt_f32 %0, %1" : "=v"(v[i].x) : "v"(v[i].x));
      if (MODE == 7) asm volatile("v_rsq_f32 %0, %1" : "=v"(v[i].x) : "v"(v[i].x));
      if (MODE == 8) asm volatile("v_sin_f32 %0, %1" : "=v"(v[i].x) : "v"(v[i].x));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i].x + v[i].y;
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int w, float* out, unsigned long long* cyc) {
  const int grid = 256 * 4 * w;
  hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(64), 0, 0, out, cyc, 0.5f);
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(64), 0, 0, out, cyc, 0.5f);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long* h = (unsigned long long*)malloc(grid * 8);
  CHECK(hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost));
  double tot = 0;
  for (int i = 0; i < grid; ++i) tot += (double)h[i];
  free(h);
  const double per_wave_instr = tot / grid / ITER / 16.0;        // shader cycles of one wave per instruction (with w waves sharing the SIMD)
  printf("%-38s waves/SIMD %d: %6.2f cycles per instruction per wave -> %5.2f cycles per instruction per SIMD; wall %7.3f ms\n", name, w, per_wave_instr,
         per_wave_instr / w, ms);
}

int main() {
  float* out;
  unsigned long long* cyc;
  CHECK(hipMalloc(&out, 256 * 4 * 8 * 64 * sizeof(float)));
  CHECK(hipMalloc(&cyc, 256 * 4 * 8 * 8));
  for (int w = 1; w <= 4; w *= 2) {
    run<0>("v_add_f32", w, out, cyc);
    run<1>("v_fma_f32", w, out, cyc);
    run<2>("v_pk_add_f32", w, out, cyc);
    run<3>("v_pk_mul_f32", w, out, cyc);
    run<4>("v_pk_fma_f32", w, out, cyc);
    run<5>("v_pk_fma_f32 op_sel (complex form)", w, out, cyc);
    run<6>("v_sqrt_f32", w, out, cyc);
    run<7>("v_rsq_f32", w, out, cyc);
    run<8>("v_sin_f32", w, out, cyc);
  }
  return 0;
}
