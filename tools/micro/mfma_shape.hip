// mfma_shape.hip -- does the chip grant the same matrix rate to v_mfma_f32_16x16x32_bf16 and v_mfma_f32_32x32x16_bf16 when all 256 CUs
// multiply back to back under the package power limit?  Both have the nominal rate of 1 024 flop / clk / SIMD, but the 16 x 16 form reads twice
// the operand registers per flop (A 16 x 32 + B 32 x 16 elements per 16 384 flop against 32 x 16 + 16 x 32 per 32 768).
// The body is the register shape of the row-complete GEMM's wave tile (64 x 192 outputs = 192 accumulator registers, gemm4.hip):
//   form 0: 4 x 12 tiles of 16 x 16, 48 MFMAs per 32-deep k-step        form 1: 2 x 6 tiles of 32 x 32, 12 MFMAs per 16-deep k-step (24 per 32)
// grid = 256 workgroups x 8 waves (two per SIMD, 256 registers each), fragments loop-invariant (no memory instruction in the loop).
//   hipcc -O3 --offload-arch=gfx950 -o tools/micro/mfma_shape tools/micro/mfma_shape.hip && tools/micro/mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// uniform in [-1, 1) from an integer hash: every operand bit toggles between lanes and fragments, as with real activations / weights (the matrix
// pipe's power, and so the clock it is given, depends on the data)
__device__ __forceinline__ float rnd(unsigned i, float seed) {
  unsigned h = i * 2654435761u + __float_as_uint(seed);
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
  return (float)(int)(h & 0xffffu) * (1.0f / 32768.0f) - 1.0f;
}

template <int FORM>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k(float* out, float seed, int iters) {
  bf16x8 a[4], b[6];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)rnd(threadIdx.x * 64 + i * 8 + e, seed);
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) b[i][e] = (__bf16)rnd(40000 + threadIdx.x * 64 + i * 8 + e, seed);
  float s = 0.f;
  if (FORM == 0) {
    f32x4 acc[4][12];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int t = 0; t < 12; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) acc[i][4 * p + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[jj], a[i], acc[i][4 * p + jj], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(a[i]), "+v"(b[i]));
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int t = 0; t < 12; ++t) s += acc[i][t][0] + acc[i][t][1] + acc[i][t][2] + acc[i][t][3];
  } else {
    f32x16 acc[2][6];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int t = 0; t < 6; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[t], a[2 * ks + i], acc[i][t], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(a[i]), "+v"(b[i]));
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][t][r];
  }
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int FORM>
static void run(const char* name, float* out, int iters, int reps) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<FORM>, dim3(256), dim3(512), 0, 0, out, 0.01f, iters);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<FORM>, dim3(256), dim3(512), 0, 0, out, 0.01f, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  // flop per wave-iteration: 64 x 192 outputs x 32 deep x 2
  const double flop = 256.0 * 8 * (double)iters * 64 * 192 * 32 * 2 * reps;
  printf("%-28s %8.1f us per launch  %8.1f TFLOP/s  (%.3f of 2 500)\n", name, 1e3 * ms / reps, flop / (ms * 1e-3) / 1e12, flop / (ms * 1e-3) / 2.5e15);
}

int main() {
  float* out;
  CHECK(hipMalloc(&out, 4096));
  for (int round = 0; round < 3; ++round) {
    run<0>("16x16x32, 48 per k-step", out, 600, 3000);
    run<1>("32x32x16, 24 per k-step", out, 600, 3000);
  }
  return 0;
}
