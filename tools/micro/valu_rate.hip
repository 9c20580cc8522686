// valu_rate.hip -- issue-rate microbenchmarks behind DESIGN.md's MHSA analysis (gfx950): how many cycles a SIMD spends per v_exp_f32 /
// v_add_f32 / v_mfma_f32_32x32x16_bf16, and whether the vector instructions of ONE wave issue under its own in-flight MFMAs.
//   hipcc -O3 --offload-arch=gfx950 -o tools/micro/valu_rate tools/micro/valu_rate.hip && tools/micro/valu_rate
// Every test: grid = 256 CUs x waves-per-SIMD x 4 SIMDs worth of 64-thread workgroups... (one wave per workgroup, W workgroups per SIMD),
// ITER iterations of an unrolled body; reported: ns per body iteration per wave and, at an assumed 2.4 GHz, cycles per instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int ITER = 4096;

// MODE 0: 16 independent v_exp_f32 per iteration; 1: 16 v_add_f32; 2: 1 MFMA; 3: 1 MFMA + NV adds; 4: 1 MFMA + NV exps; 5: NV adds only; 6: NV exps only
template <int MODE, int NV>
__global__ __launch_bounds__(64) void k(float* out, float seed) {
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = seed + i * 0.001f + threadIdx.x * 1e-6f;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
  for (int it = 0; it < ITER; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]);
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = v[i] + seed;
    } else {
      if (MODE == 2 || MODE == 3 || MODE == 4) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      if (MODE == 3 || MODE == 5) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i & 15] = v[i & 15] + seed;
      }
      if (MODE == 4 || MODE == 6) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i & 15] = __builtin_amdgcn_exp2f(v[i & 15]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 7 || MODE == 8) {
      // one half-step body of mhsa_pipe.hip: 16 x (sub, exp, add), 7 max3, 8 cvt_pk; MODE 8 adds the 8 MFMAs, one per 2 elements
      float mx = fmaxf(fmaxf(v[0], v[1]), v[2]);
#pragma unroll
      for (int i = 3; i < 15; i += 2) mx = fmaxf(fmaxf(mx, v[i]), v[i + 1]);
      float rs = 0.f;
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        if (MODE == 8) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        const float e0 = __builtin_amdgcn_exp2f(v[i] - mx), e1 = __builtin_amdgcn_exp2f(v[i + 1] - mx);
        rs += e0;
        rs += e1;
        v[i] = e0 + seed;
        v[i + 1] = e1 + seed;
        __builtin_amdgcn_sched_barrier(0);
      }
      typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
      float cv = 0.f;
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        bf2 p = {(__bf16)v[i], (__bf16)v[i + 1]};
        cv += __uint_as_float(__builtin_bit_cast(unsigned, p) & 0x3f800000u);
      }
      v[0] += rs * 1e-30f + cv * 1e-30f;
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i] + acc[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int MODE, int NV>
void run(const char* name, int waves_per_simd, int n_instr, float* out) {
  const int grid = 256 * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MODE, NV>), dim3(grid), dim3(64), 0, 0, out, 0.5f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MODE, NV>), dim3(grid), dim3(64), 0, 0, out, 0.5f);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double ns_iter = ms * 1e6 / ITER;           // per iteration of the body, for the waves_per_simd waves of a SIMD together
  printf("%-44s waves/SIMD %d: %8.2f ns per iteration (all waves of a SIMD) = %7.1f cycles @2.4 GHz; per wave-iteration %7.1f cycles; per instruction %6.2f\n", name,
         waves_per_simd, ns_iter, ns_iter * 2.4, ns_iter * 2.4 / waves_per_simd, ns_iter * 2.4 / waves_per_simd / n_instr);
}

int main() {
  float* out;
  CHECK(hipMalloc(&out, 256 * 4 * 8 * 64 * sizeof(float)));
  for (int w = 1; w <= 4; w *= 2) {
    run<0, 0>("16 x v_exp_f32", w, 16, out);
    run<1, 0>("16 x v_add_f32", w, 16, out);
    run<2, 0>("1 x mfma_32x32x16_bf16", w, 1, out);
  }
  for (int w = 1; w <= 4; ++w) {
    run<7, 0>("softmax half-step body (82 valu, 16 trans)", w, 82, out);
    run<8, 0>("same + 8 mfma", w, 90, out);
  }
  for (int w = 1; w <= 3; ++w) {
    run<5, 4>("4 adds", w, 4, out);
    run<3, 4>("mfma + 4 adds", w, 5, out);
    run<5, 8>("8 adds", w, 8, out);
    run<3, 8>("mfma + 8 adds", w, 9, out);
    run<5, 16>("16 adds", w, 16, out);
    run<3, 16>("mfma + 16 adds", w, 17, out);
    run<6, 4>("4 exps", w, 4, out);
    run<4, 4>("mfma + 4 exps", w, 5, out);
    run<6, 8>("8 exps", w, 8, out);
    run<4, 8>("mfma + 8 exps", w, 9, out);
  }
  return 0;
}
