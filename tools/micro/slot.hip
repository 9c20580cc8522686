// slot.hip -- what does ONE slot of mhsa3.hip cost?  slot = 1 MFMA 32x32x16 + 4 v_exp_f32 + row-sum adds + 2 v_cvt_pk_bf16_f32 (+ optionally an LDS
// fragment read two slots ahead).  512-thread workgroups (2 waves per SIMD) or 256 (1 wave per SIMD), every CU busy.
//   MODE 0: MFMA only; 1: vector share only (dependent as in the kernel: adds consume the exps of the same slot); 2: both interleaved;
//   3: both, vector share software-pipelined (adds / packs consume the PREVIOUS slot's exps); 4: as 2 plus the LDS read + wait
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int ITER = 2048;
#define SB() __builtin_amdgcn_sched_barrier(0)

template <int MODE>
__global__ void k(float* out, unsigned long long* cyc, float seed) {
  __shared__ __attribute__((aligned(16))) char lds[16384];
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = seed * (i + 1) * 0.01f - 3.f + threadIdx.x * 1e-4f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i); b[i] = (__bf16)(seed - i); }
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = seed + i;
  __syncthreads();
  float rs = 0.f;
  typedef float f2t __attribute__((ext_vector_type(2)));
  f2t rs2 = {0.f, 0.f};
  unsigned pk0 = 0, pk1 = 0;
  float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
  const char* lp = lds + (threadIdx.x & 63) * 16;
  bf16x8 f0 = *reinterpret_cast<const bf16x8*>(lp), f1 = *reinterpret_cast<const bf16x8*>(lp + 1024);
  f32x16 accB = acc;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER / 2; ++it) {
    // two steps of 4 slots, as in the kernel: the MFMAs of a step write one accumulator block while the vector share reads the OTHER one
#define STEP(WR, RD)                                                                                                   \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                    \
      bf16x8 fn = f1;                                                                                                  \
      if (MODE == 4) fn = *reinterpret_cast<const bf16x8*>(lp + ((it * 4 + j) & 7) * 1024);                            \
      if (MODE != 1) WR = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MODE == 4 ? f0 : a, b, WR, 0, 0, 0);                  \
      SB();                                                                                                            \
      if (MODE == 1 || MODE == 2 || MODE == 4) {                                                                       \
        const float a0 = __builtin_amdgcn_exp2f(RD[4 * j] * 1e-3f), a1 = __builtin_amdgcn_exp2f(RD[4 * j + 1] * 1e-3f);  \
        const float a2 = __builtin_amdgcn_exp2f(RD[4 * j + 2] * 1e-3f), a3 = __builtin_amdgcn_exp2f(RD[4 * j + 3] * 1e-3f); \
        rs += (a0 + a1) + (a2 + a3);                                                                                   \
        const bf16x2 q0 = {(__bf16)a0, (__bf16)a1}, q1 = {(__bf16)a2, (__bf16)a3};                                     \
        pk0 ^= __builtin_bit_cast(unsigned, q0);                                                                       \
        pk1 ^= __builtin_bit_cast(unsigned, q1);                                                                       \
      } else if (MODE == 5 || MODE == 6) {                                                                             \
        const float a0 = __builtin_amdgcn_exp2f(MODE == 5 ? RD[4 * j] * 1e-3f : RD[4 * j]), a1 = __builtin_amdgcn_exp2f(MODE == 5 ? RD[4 * j + 1] * 1e-3f : RD[4 * j + 1]); \
        typedef float f2_ __attribute__((ext_vector_type(2)));                                                          \
        f2_ pr = {a0, a1};                                                                                             \
        rs2 += pr;                                                                                                     \
        const bf16x2 q0 = {(__bf16)a0, (__bf16)a1};                                                                    \
        pk0 ^= __builtin_bit_cast(unsigned, q0);                                                                       \
        asm volatile("" : "+v"(rs2));                                                                                  \
      } else if (MODE == 3) {                                                                                          \
        rs += (p0 + p1) + (p2 + p3);                                                                                   \
        const bf16x2 q0 = {(__bf16)p0, (__bf16)p1}, q1 = {(__bf16)p2, (__bf16)p3};                                     \
        pk0 ^= __builtin_bit_cast(unsigned, q0);                                                                       \
        pk1 ^= __builtin_bit_cast(unsigned, q1);                                                                       \
        p0 = __builtin_amdgcn_exp2f(RD[4 * j] * 1e-3f); p1 = __builtin_amdgcn_exp2f(RD[4 * j + 1] * 1e-3f);            \
        p2 = __builtin_amdgcn_exp2f(RD[4 * j + 2] * 1e-3f); p3 = __builtin_amdgcn_exp2f(RD[4 * j + 3] * 1e-3f);        \
      }                                                                                                                \
      asm volatile("" : "+v"(rs), "+v"(pk0), "+v"(pk1));                                                               \
      f0 = f1;                                                                                                         \
      f1 = fn;                                                                                                         \
      SB();                                                                                                            \
    }
    STEP(accB, acc)
    if (MODE == 1) { for (int i = 0; i < 16; ++i) accB[i] += rs * 1e-9f; }
    STEP(acc, accB)
    if (MODE == 1) { for (int i = 0; i < 16; ++i) acc[i] += rs * 1e-9f; }
#undef STEP
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = rs + rs2.x + rs2.y + p0 + p1 + p2 + p3 + __uint_as_float(pk0) + __uint_as_float(pk1);
  for (int i = 0; i < 16; ++i) s += acc[i] + accB[i];
  for (int i = 0; i < 8; ++i) s += (float)f0[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads, float* out, unsigned long long* cyc) {
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, out, cyc, 0.5f);
  CHECK(hipDeviceSynchronize());
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(threads), 0, 0, out, cyc, 0.5f);
  CHECK(hipDeviceSynchronize());
  unsigned long long h[256 * 8];
  CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
  const int nw = threads / 64;
  double a = 0, b = 0;
  for (int i = 0; i < 256; ++i) for (int w = 0; w < nw; ++w) (w < 4 ? a : b) += (double)h[i * 8 + w];
  printf("%-72s %d wave(s)/SIMD: %6.1f cycles per slot (waves 0-3)", name, nw / 4, a / 1024 / ITER / 4);
  if (nw > 4) printf("  %6.1f (waves 4-7)  -> %5.1f cycles per MFMA on the SIMD", b / 1024 / ITER / 4, 1.0 / (1.0 / (a / 1024 / ITER / 4) + 1.0 / (b / 1024 / ITER / 4)));
  printf("\n");
}

int main() {
  float* out;
  unsigned long long* cyc;
  CHECK(hipMalloc(&out, 256 * 512 * 4));
  CHECK(hipMalloc(&cyc, 256 * 8 * 8));
  for (int threads = 256; threads <= 512; threads += 256) {
    run<0>("MFMA only", threads, out, cyc);
    run<1>("vector share only (4 exp2 + 4 adds + 2 packs + 2 xor)", threads, out, cyc);
    run<2>("MFMA + vector share", threads, out, cyc);
    run<3>("MFMA + vector share, adds / packs one slot behind their exps", threads, out, cyc);
    run<4>("MFMA (fragment from LDS, read two slots ahead) + vector share", threads, out, cyc);
    run<5>("MFMA + HALF share: 2 mul + 2 exp2 + 1 packed add + 1 pack", threads, out, cyc);
    run<6>("MFMA + half share, pre-scaled: 2 exp2 + 1 packed add + 1 pack", threads, out, cyc);
  }
  return 0;
}
