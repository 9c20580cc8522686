// coissue.hip -- do the matrix pipe and the vector pipe of ONE SIMD run concurrently when the work comes from two DIFFERENT waves?
// 512-thread workgroups (waves w and w + 4 share a SIMD); role A (waves 0-3): a dependent chain of v_mfma_f32_32x32x16_bf16; role B (waves 4-7):
// independent v_fma_f32 (or v_exp_f32).  Reported: shader cycles of each role alone and together, one workgroup per CU on every CU.
//   hipcc -O3 --offload-arch=gfx950 -o tools/micro/coissue tools/micro/coissue.hip && tools/micro/coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int ITER = 4096;

// mode bit 0: role A active; bit 1: role B active; VK = 0: v_fma_f32, 1: v_exp_f32; SAME = 1: both roles in EVERY wave (one interleaved stream)
// IND = 1: role A's MFMAs rotate over four independent accumulators (no back-to-back dependency); PRIO = 1: role B raises its priority (s_setprio 3)
template <int VK, int SAME, int IND = 0, int PRIO = 0>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, float seed, int mode, int nv) {
  const int wave = threadIdx.x >> 6;
  const bool roleA = SAME ? true : wave < 4, roleB = SAME ? true : wave >= 4;
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = seed + i * 0.001f + threadIdx.x * 1e-6f;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
  const bool doA = roleA && (mode & 1), doB = roleB && (mode & 2);
  f32x16 acc1 = acc, acc2 = acc, acc3 = acc;
  if (PRIO == 1 && roleB && !SAME) __builtin_amdgcn_s_setprio(3);
  if (PRIO == 2 && roleA && !SAME) __builtin_amdgcn_s_setprio(3);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (doA && !doB) {
    if (IND) {
      for (int it = 0; it < ITER; it += 4) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc3, 0, 0, 0);
      }
    } else
    for (int it = 0; it < ITER; ++it) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  } else if (doB && !doA) {
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (VK == 0) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v[i]) : "v"(v[i]), "v"(seed), "v"(seed));
        else asm volatile("v_exp_f32 %0, %1" : "=v"(v[i]) : "v"(v[i]));
      }
    }
  } else if (doA && doB) {      // SAME: one wave carries both streams, interleaved 1 MFMA : 8 vector instructions
    for (int it = 0; it < ITER; ++it) {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (VK == 0) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v[i]) : "v"(v[i]), "v"(seed), "v"(seed));
        else asm volatile("v_exp_f32 %0, %1" : "=v"(v[i]) : "v"(v[i]));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i] + acc[i] + acc1[i] + acc2[i] + acc3[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int VK, int SAME, int IND = 0, int PRIO = 0>
void run(const char* name, int mode, float* out, unsigned long long* cyc) {
  hipLaunchKernelGGL((k<VK, SAME, IND, PRIO>), dim3(256), dim3(512), 0, 0, out, cyc, 0.5f, mode, 8);
  CHECK(hipDeviceSynchronize());
  hipLaunchKernelGGL((k<VK, SAME, IND, PRIO>), dim3(256), dim3(512), 0, 0, out, cyc, 0.5f, mode, 8);
  CHECK(hipDeviceSynchronize());
  unsigned long long h[256 * 8];
  CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
  double a = 0, b = 0;
  for (int i = 0; i < 256; ++i) for (int w = 0; w < 8; ++w) (w < 4 ? a : b) += (double)h[i * 8 + w];
  printf("%-64s waves 0-3: %7.1f cycles / iteration   waves 4-7: %7.1f\n", name, a / 1024 / ITER, b / 1024 / ITER);
}

int main() {
  float* out;
  unsigned long long* cyc;
  CHECK(hipMalloc(&out, 256 * 512 * 4));
  CHECK(hipMalloc(&cyc, 256 * 8 * 8));
  printf("per iteration: role A = 1 MFMA 32x32x16 (32 cycles of matrix pipe), role B = 8 vector instructions\n");
  run<0, 0>("MFMA waves alone (partner idle)", 1, out, cyc);
  run<0, 0>("v_fma waves alone (partner idle)", 2, out, cyc);
  run<0, 0>("MFMA waves + v_fma waves on the same SIMDs", 3, out, cyc);
  run<1, 0>("v_exp waves alone (partner idle)", 2, out, cyc);
  run<1, 0>("MFMA waves + v_exp waves on the same SIMDs", 3, out, cyc);
  run<0, 1>("every wave: 1 MFMA + 8 v_fma interleaved (2 waves / SIMD)", 3, out, cyc);
  run<1, 1>("every wave: 1 MFMA + 8 v_exp interleaved (2 waves / SIMD)", 3, out, cyc);
  run<0, 0, 1, 0>("independent MFMAs (4 accumulators) alone", 1, out, cyc);
  run<0, 0, 1, 0>("independent MFMAs + v_fma waves", 3, out, cyc);
  run<1, 0, 1, 0>("independent MFMAs + v_exp waves", 3, out, cyc);
  run<0, 0, 0, 1>("dependent MFMAs + v_fma waves at s_setprio 3", 3, out, cyc);
  run<0, 0, 1, 1>("independent MFMAs + v_fma waves at s_setprio 3", 3, out, cyc);
  run<1, 0, 1, 1>("independent MFMAs + v_exp waves at s_setprio 3", 3, out, cyc);
  run<0, 0, 0, 2>("dependent MFMAs at s_setprio 3 + v_fma waves", 3, out, cyc);
  return 0;
}
