// lds_rate.hip -- what the LDS delivers per clock and CU for the three instructions the attention / weight-gradient kernels feed their MFMAs
// with: ds_read_b128, ds_read_b64_tr_b16, ds_write_b128 (conflict-free lane-linear addresses), at 1 and 2 waves per SIMD (256- / 512-thread
// workgroups, one per CU, all 256 CUs), drained (s_waitcnt lgkmcnt(0) after every 16) and continuous (no wait inside the loop); then the
// attention forward's own mix -- per 16 v_mfma_f32_32x32x16_bf16: 8 ds_read_b128 (K fragments) + 16 ds_read_b64_tr_b16 (V^T fragments) --
// alone and interleaved with the MFMAs.  Written to settle roofline.json: mhsa_fwd_floor (r03 assumed 128 B/clk for b128 and 64 for tr_b16;
// MI355X_MICROARCH.md's LDS table says 256 B/clk/CU for both).
//   hipcc -O3 --offload-arch=gfx950 -o tools/micro/lds_rate tools/micro/lds_rate.hip && tools/micro/lds_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int ITER = 512;

#define RD128(i) asm volatile("ds_read_b128 %0, %1 offset:" #i : "=v"(r4[(i / 16) & 7]) : "v"(a16))
#define RDTR(i) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #i : "=v"(r2[(i / 8) & 15]) : "v"(a8))
#define WR128(i) asm volatile("ds_write_b128 %0, %1 offset:" #i :: "v"(a16), "v"(w4))

// KIND 0 b128, 1 tr_b16, 2 write_b128; DRAIN 1: lgkmcnt(0) after every 16 instructions
template <int KIND, int DRAIN>
__global__ __launch_bounds__(512) void k_rate(float* out, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) char smem[65536];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) ((float*)smem)[i] = (float)i;
  __syncthreads();
  const unsigned a16 = (unsigned)(size_t)smem + wave * 4096 + lane * 16, a8 = (unsigned)(size_t)smem + wave * 4096 + lane * 8;
  u32x4 r4[8];
  u32x2 r2[16];
  u32x4 w4 = {1u, 2u, 3u, (unsigned)lane};
#pragma unroll
  for (int i = 0; i < 8; ++i) r4[i] = w4;
#pragma unroll
  for (int i = 0; i < 16; ++i) r2[i] = u32x2{1u, 2u};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; ++it) {
    if (KIND == 0) { RD128(0); RD128(1024); RD128(2048); RD128(3072); RD128(16); RD128(1040); RD128(2064); RD128(3088);
                     RD128(32); RD128(1056); RD128(2080); RD128(3104); RD128(48); RD128(1072); RD128(2096); RD128(3120); }
    if (KIND == 1) { RDTR(0); RDTR(512); RDTR(1024); RDTR(1536); RDTR(2048); RDTR(2560); RDTR(3072); RDTR(3584);
                     RDTR(8); RDTR(520); RDTR(1032); RDTR(1544); RDTR(2056); RDTR(2568); RDTR(3080); RDTR(3592); }
    if (KIND == 2) { WR128(0); WR128(1024); WR128(2048); WR128(3072); WR128(0); WR128(1024); WR128(2048); WR128(3072);
                     WR128(0); WR128(1024); WR128(2048); WR128(3072); WR128(0); WR128(1024); WR128(2048); WR128(3072); }
    if (DRAIN) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += r4[i].x + r4[i].w;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += r2[i].x + r2[i].y;
  out[blockIdx.x * 512 + threadIdx.x] = (float)s;
  if (lane == 0) { cyc[blockIdx.x * 8 + wave] = t0; cyc[2048 + blockIdx.x * 8 + wave] = t1; }
}

// the attention forward's unit: 16 MFMAs fed by 8 b128 + 16 tr_b16 reads.  MODE bit 0: the reads, bit 1: the MFMAs; interleaved as the kernel
// would (per MFMA pair: one b128 + two tr_b16), counted waits only (one lgkmcnt(0) per unit, in front of the first MFMA of the NEXT unit)
template <int MODE>
__global__ __launch_bounds__(512) void k_unit(float* out, unsigned long long* cyc, float seed) {
  __shared__ __attribute__((aligned(16))) char smem[65536];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) ((float*)smem)[i] = (float)i * 1e-9f;
  __syncthreads();
  const unsigned a16 = (unsigned)(size_t)smem + wave * 4096 + lane * 16, a8 = (unsigned)(size_t)smem + wave * 4096 + lane * 8;
  u32x4 r4[8];
  u32x2 r2[16];
#pragma unroll
  for (int i = 0; i < 8; ++i) r4[i] = u32x4{1u, 2u, 3u, 4u};
#pragma unroll
  for (int i = 0; i < 16; ++i) r2[i] = u32x2{1u, 2u};
  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  bf16x8 b;
#pragma unroll
  for (int i = 0; i < 8; ++i) b[i] = (__bf16)(seed + i);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; ++it) {
    if (MODE & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define PAIR(j, o128, otr0, otr1)                                                                                         \
    if (MODE & 2) {                                                                                                       \
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, r4[j]), b, acc0, 0, 0, 0);                 \
    }                                                                                                                     \
    if (MODE & 1) { RD128(o128); RDTR(otr0); }                                                                            \
    if (MODE & 2) {                                                                                                       \
      const u32x4 v_ = {r2[2 * j].x, r2[2 * j].y, r2[2 * j + 1].x, r2[2 * j + 1].y};                                      \
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, v_), b, acc1, 0, 0, 0);                    \
    }                                                                                                                     \
    if (MODE & 1) { RDTR(otr1); }
    PAIR(0, 0, 0, 8) PAIR(1, 16, 16, 24) PAIR(2, 32, 32, 40) PAIR(3, 48, 48, 56) PAIR(4, 64, 64, 72) PAIR(5, 80, 80, 88) PAIR(6, 96, 96, 104)
    PAIR(7, 112, 112, 120)
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += (float)r4[i].x;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += (float)r2[i].x;
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (lane == 0) { cyc[blockIdx.x * 8 + wave] = t0; cyc[2048 + blockIdx.x * 8 + wave] = t1; }
}

// cycles from the first wave's start to the LAST wave's end, mean over the 256 workgroups.  (The mean of the per-wave times is misleading at two
// waves per SIMD: issue is arbitrated by age, the older wave runs at nearly its solo speed and the younger one waits, so the mean reads 0.75 of
// the pair's time -- the first version of this tool reported 24 cycles per MFMA that way.)
static double mean_cycles(unsigned long long* cyc, int waves) {
  static unsigned long long h[2 * 256 * 8];
  CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
  double a = 0;
  for (int i = 0; i < 256; ++i) {
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int w = 0; w < waves; ++w) { t0 = h[i * 8 + w] < t0 ? h[i * 8 + w] : t0; t1 = h[2048 + i * 8 + w] > t1 ? h[2048 + i * 8 + w] : t1; }
    a += (double)(t1 - t0);
  }
  return a / 256.0;
}

template <int KIND, int DRAIN>
void run_rate(const char* name, int bytes_per_lane, float* out, unsigned long long* cyc) {
  for (int threads = 256; threads <= 512; threads += 256) {
    const int waves = threads / 64;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k_rate<KIND, DRAIN>), dim3(256), dim3(threads), 0, 0, out, cyc); CHECK(hipDeviceSynchronize()); }
    const double c = mean_cycles(cyc, waves);
    const double per_cu = c / (ITER * 16.0) / waves;                 // the workgroup issued waves x ITER x 16 instructions in c cycles
    printf("%-22s %s  %d waves/SIMD: %6.2f cycles per wave-instruction at the CU  = %6.1f B/clk/CU\n", name, DRAIN ? "drained   " : "continuous", waves / 4,
           per_cu, 64.0 * bytes_per_lane / per_cu);
  }
}

template <int MODE>
void run_unit(const char* name, float* out, unsigned long long* cyc) {
  for (int threads = 256; threads <= 512; threads += 256) {
    const int waves = threads / 64;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k_unit<MODE>), dim3(256), dim3(threads), 0, 0, out, cyc, 0.5f); CHECK(hipDeviceSynchronize()); }
    const double c = mean_cycles(cyc, waves) / ITER;
    printf("%-46s %d waves/SIMD: %7.1f cycles per unit at the SIMD (matrix floor 512; LDS-array floor 64 x 4 waves = 256)\n", name, waves / 4, c / (waves / 4));
  }
}

int main() {
  float* out; unsigned long long* cyc;
  CHECK(hipMalloc(&out, 256 * 512 * 4));
  CHECK(hipMalloc(&cyc, 2 * 256 * 8 * 8));
  run_rate<0, 1>("ds_read_b128", 16, out, cyc);
  run_rate<0, 0>("ds_read_b128", 16, out, cyc);
  run_rate<1, 1>("ds_read_b64_tr_b16", 8, out, cyc);
  run_rate<1, 0>("ds_read_b64_tr_b16", 8, out, cyc);
  run_rate<2, 1>("ds_write_b128", 16, out, cyc);
  run_rate<2, 0>("ds_write_b128", 16, out, cyc);
  run_unit<1>("unit: 8 b128 + 16 tr_b16 reads only", out, cyc);
  run_unit<2>("unit: 16 MFMA 32x32x16 only", out, cyc);
  run_unit<3>("unit: reads interleaved with the MFMAs", out, cyc);
  return 0;
}
