// attn_skel.hip -- the matrix + vector SKELETON of the head-dim-64 flash-attention tile with NO memory traffic at all: per (32 query x 64 key)
// unit a wave issues 8 MFMAs (S^T = K Q^T; operands in registers), 32 v_exp_f32 + 32 row-sum adds + 16 bf16 packs on their results, then 8 MFMAs
// (O^T += V^T P^T) on the packed probabilities -- the true dependencies, nothing else.  Run at 1 .. 4 waves per SIMD (one workgroup per CU, all
// CUs): what the SIMD needs per unit when ONLY the matrix pipe and the vector issue port are in play.  If this already costs ~1 300 cycles per
// unit, no staging / LDS / barrier restructuring of mhsa.hip can get under it; if it costs ~600, those are what to fix.
//   MODE 0: phases in program order (QK | softmax | PV), as the kernels are written
//   MODE 1: QK^T of unit u + 1 is issued BEFORE the softmax of unit u (two score buffers): the matrix work next to a wave's own vector work
//   MODE 2: MFMAs only      MODE 3: vector work only (scores = constants)
//   hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -o tools/micro/attn_skel tools/micro/attn_skel.hip && tools/micro/attn_skel
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int UNITS = 256;

__device__ __forceinline__ void qk(const bf16x8 (&kf)[2], const bf16x8 (&qf)[4], f32x16& s0, f32x16& s1) {
  const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[s], s == 0 ? z : s0, 0, 0, 0);
    s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1], qf[s], s == 0 ? z : s1, 0, 0, 0);
  }
}
__device__ __forceinline__ void softmax(const f32x16& s0, const f32x16& s1, bf16x8 (&pf)[2][2], float& l_run) {
  float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float a0 = __builtin_amdgcn_exp2f(s0[8 * s + j]);
      const float a1 = __builtin_amdgcn_exp2f(s1[8 * s + j]);
      rs0 += a0;
      rs1 += a1;
      pf[0][s][j] = (__bf16)a0;
      pf[1][s][j] = (__bf16)a1;
    }
  l_run += rs0 + rs1;
}
__device__ __forceinline__ void pv(const bf16x8 (&vf)[2], const bf16x8 (&pf)[2][2], f32x16& o0, f32x16& o1) {
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0], pf[kb][s], o0, 0, 0, 0);
      o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1], pf[kb][s], o1, 0, 0, 0);
    }
}

template <int MODE, int W>
__global__ __launch_bounds__(256 * W) void k(float* out, unsigned long long* cyc, float seed) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bf16x8 kf[2], vf[2], qf[4];      // ONE pair of K / V^T fragments reused by every MFMA (timing does not depend on the values; registers do matter)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { kf[i][j] = (__bf16)(seed * (float)((lane * 7 + i * 3 + j) % 13 - 6)); vf[i][j] = (__bf16)(seed * (float)((lane * 5 + i + j * 3) % 11 - 5)); }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) qf[i][j] = (__bf16)(seed * (float)((lane + i * 5 + j) % 9 - 4));
  f32x16 o0, o1, s0, s1, t0_, t1_;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; s0[r] = seed * r; s1[r] = -seed * r; t0_[r] = 0.f; t1_[r] = 0.f; }
  float l_run = 0.f;
  bf16x8 pf[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[a][c][j] = (__bf16)(seed * (j + a + c));
  __syncthreads();
  const unsigned long long ta = __builtin_amdgcn_s_memtime();
  if (MODE == 0) {
    for (int u = 0; u < UNITS; ++u) {
      qk(kf, qf, s0, s1);
      softmax(s0, s1, pf, l_run);
      pv(vf, pf, o0, o1);
      qf[0][0] = (__bf16)l_run;            // the next unit's scores depend on this unit (no cross-iteration hoisting)
    }
  } else if (MODE == 1) {
    qk(kf, qf, s0, s1);
    for (int u = 0; u < UNITS; u += 2) {
      qk(kf, qf, t0_, t1_);                // unit u + 1's scores while unit u's are exponentiated
      softmax(s0, s1, pf, l_run);
      pv(vf, pf, o0, o1);
      qk(kf, qf, s0, s1);
      softmax(t0_, t1_, pf, l_run);
      pv(vf, pf, o0, o1);
      qf[0][0] = (__bf16)l_run;
    }
  } else if (MODE == 2) {
    for (int u = 0; u < UNITS; ++u) {
      qk(kf, qf, s0, s1);
      asm volatile("" : "+v"(s0), "+v"(s1));      // all 8 QK^T MFMAs stay (the first version of this line kept 9 of the unit's 16: only s0[0..3] was used)
      pf[0][0] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(s0, s0, 0, 1, 2, 3));
      pv(vf, pf, o0, o1);
    }
  } else {
    for (int u = 0; u < UNITS; ++u) {
      softmax(s0, s1, pf, l_run);
      s0[0] = l_run * 1e-9f;
      asm volatile("" : "+v"(pf[0][0]), "+v"(pf[0][1]), "+v"(pf[1][0]), "+v"(pf[1][1]));
    }
  }
  const unsigned long long tb = __builtin_amdgcn_s_memtime();
  float acc = l_run;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc += o0[r] + o1[r] + s0[r] + s1[r] + t0_[r] + t1_[r];
  acc += (float)pf[0][0][0];
  out[blockIdx.x * 1024 + threadIdx.x] = acc;
  if (lane == 0) { cyc[blockIdx.x * 16 + wave] = ta; cyc[4096 + blockIdx.x * 16 + wave] = tb; }
}

template <int MODE, int W>
void launch(float* out, unsigned long long* cyc) { hipLaunchKernelGGL((k<MODE, W>), dim3(256), dim3(256 * W), 0, 0, out, cyc, 0.01f); CHECK(hipDeviceSynchronize()); }

template <int MODE>
void run(const char* name, float* out, unsigned long long* cyc) {
  static unsigned long long h[2 * 4096];
  printf("%-72s", name);
  for (int w = 1; w <= 4; ++w) {
    if (MODE == 1 && w == 4) { printf("  4/SIMD:    n/a (144 registers)"); continue; }
    for (int rep = 0; rep < 2; ++rep) {
      if (w == 1) launch<MODE, 1>(out, cyc); else if (w == 2) launch<MODE, 2>(out, cyc); else if (w == 3) launch<MODE, 3>(out, cyc); else launch<MODE, 4>(out, cyc);
    }
    CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double a = 0;
    for (int i = 0; i < 256; ++i) {
      unsigned long long t0 = ~0ull, t1 = 0;
      for (int x = 0; x < 4 * w; ++x) { t0 = h[i * 16 + x] < t0 ? h[i * 16 + x] : t0; t1 = h[4096 + i * 16 + x] > t1 ? h[4096 + i * 16 + x] : t1; }
      a += (double)(t1 - t0);
    }
    printf("  %d/SIMD: %6.0f", w, a / 256.0 / UNITS / w);
  }
  printf("   cycles per unit at the SIMD\n");
}

int main() {
  float* out; unsigned long long* cyc;
  CHECK(hipMalloc(&out, 256 * 1024 * 4));
  CHECK(hipMalloc(&cyc, 2 * 4096 * 8));
  // (a MODE 2 "MFMAs only" line used to be printed here: the compiler removes 6-7 of the unit's 16 MFMAs whose results the loop never consumes, so it read 290-390
  //  cycles; the floor is 16 x 32 = 512 by the instruction's issue rate: tools/micro/valu_rate.hip measures 36-40 cycles per back-to-back MFMA at the nominal clock)
  // (likewise no "vector work only" line: with constant scores the compiler hoists 31 of the 32 exponentials out of the loop)
  run<0>("QK^T | softmax | PV in program order", out, cyc);
  run<1>("QK^T of the next unit issued before the softmax of this one", out, cyc);
  return 0;
}
