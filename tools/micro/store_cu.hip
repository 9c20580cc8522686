// store_cu.hip -- developer microbenchmark (round 5): what a 16-B-per-lane global store costs the CU that issues it, next to the LDS-DMA loads of a GEMM
// main loop.  One 512-thread workgroup per CU (8 waves, as the persistent GEMMs), every wave issues ITER rounds of
//   L loads  : global_load_lds_dwordx4, 8 rows x 128 B per wave instruction (the GEMM's staging shape), source L2-resident (a 256 KB window per workgroup)
//   S stores : global_store_dwordx4 of one of four shapes into a per-workgroup window (L2-resident: 128 KB per workgroup, re-written every round)
// with a counted vmcnt that keeps ~24 operations in flight.  Prints microseconds and bytes / clk / CU for loads only, stores only, and the GEMM's 6 : 1 mix.
// shapes: 0 = 16 rows x 64 B (today's epilogue), 1 = 8 rows x 128 B (whole lines), 2 = 4 rows x 256 B, 3 = 1 KB contiguous
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/store_cu.hip -o tools/micro/store_cu ; run on the GPU box: tools/micro/store_cu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int NL, int NS>
__global__ __launch_bounds__(512) void mix_kernel(const char* __restrict__ src, char* __restrict__ dst, int iters, int ld_bytes, long long dst_stride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* s = src + (size_t)blockIdx.x * 262144;
  char* d = dst + (size_t)blockIdx.x * dst_stride;
  // load: lane -> row (lane >> 3) of 8, 16-B chunk lane & 7; rows ld_bytes apart
  const uint32_t l_off = (uint32_t)((wave * 8 + (lane >> 3)) * 1536 + (lane & 7) * 16);
  uint32_t s_off;
  if (SHAPE == 0) s_off = (uint32_t)((lane & 15) * ld_bytes + (lane >> 4) * 16);
  else if (SHAPE == 1) s_off = (uint32_t)((lane >> 3) * ld_bytes + (lane & 7) * 16);
  else if (SHAPE == 2) s_off = (uint32_t)((lane >> 4) * ld_bytes + (lane & 15) * 16);
  else s_off = (uint32_t)(lane * 16);
  s_off += (uint32_t)(wave * 16 * ld_bytes);            // every wave its own 16 rows
  const uint32_t lds_wave = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)smem + wave * 1024;
  const u32x4 v = {(unsigned)lane, (unsigned)wave, blockIdx.x, 7u};
  for (int it = 0; it < iters; ++it) {
    const char* sb = s + (size_t)(it & 15) * 12288;       // 16 x 12 KB = 192 KB window, re-read
    char* db = d + (size_t)(it & 1) * 128;                 // two positions along the row, re-written: the window (32 KB per workgroup) stays in L2
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(l_off), "s"(sb + k * 128), "s"(lds_wave + (uint32_t)(k & 7) * 8192)
                   : "memory");
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(s_off), "v"(v), "s"(db + k * 64) : "memory");
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int SHAPE, int NL, int NS>
static float run(const char* src, char* dst, int iters, int ld_bytes, long long dst_stride, int grid) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipFuncSetAttribute(reinterpret_cast<const void*>(mix_kernel<SHAPE, NL, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL((mix_kernel<SHAPE, NL, NS>), dim3(grid), dim3(512), 65536, 0, src, dst, iters, ld_bytes, dst_stride);
  hipEventRecord(a);
  hipLaunchKernelGGL((mix_kernel<SHAPE, NL, NS>), dim3(grid), dim3(512), 65536, 0, src, dst, iters, ld_bytes, dst_stride);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}

template <int SHAPE>
static void shape(const char* name, const char* src, char* dst, int grid, double ghz) {
  const int iters = 2000, ld = 4608;
  const long long stride = 128LL * ld;                   // 128 rows per workgroup
  const float tl = run<SHAPE, 6, 0>(src, dst, iters, ld, stride, grid);
  const float ts = run<SHAPE, 0, 1>(src, dst, iters * 6, ld, stride, grid);
  const float tm = run<SHAPE, 6, 1>(src, dst, iters, ld, stride, grid);
  const double lb = 6.0 * iters * 8 * 1024, sb = 6.0 * iters * 8 * 1024, mb_l = lb, mb_s = 1.0 * iters * 8 * 1024;
  printf("%-22s loads only %7.1f us (%5.1f B/clk/CU) | stores only %7.1f us (%5.1f B/clk/CU) | 6 loads : 1 store %7.1f us (loads alone would be %7.1f, stores alone %7.1f)\n", name,
         tl * 1e3, lb / (tl * 1e-3 * ghz * 1e9), ts * 1e3, sb / (ts * 1e-3 * ghz * 1e9), tm * 1e3, tl * 1e3, ts * 1e3 / 6.0);
  (void)mb_l; (void)mb_s;
}

int main(int argc, char** argv) {
  const int grid = argc > 1 ? atoi(argv[1]) : 256;
  const double ghz = 2.4;
  char *src = nullptr, *dst = nullptr;
  if (hipMalloc(&src, (size_t)grid * 262144 + 65536) != hipSuccess || hipMalloc(&dst, (size_t)grid * 128 * 4608 + 65536) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
  hipMemset(src, 1, (size_t)grid * 262144 + 65536);
  printf("grid %d workgroups x 8 waves; B/clk/CU at a nominal %.1f GHz\n", grid, ghz);
  shape<0>("16 rows x 64 B", src, dst, grid, ghz);
  shape<1>("8 rows x 128 B", src, dst, grid, ghz);
  shape<2>("4 rows x 256 B", src, dst, grid, ghz);
  shape<3>("1 KB contiguous", src, dst, grid, ghz);
  return 0;
}
