// feed.hip -- developer microbenchmark: per-CU feed rate of (a) LDS-DMA (global_load_lds_dwordx4), (b) global_load_dwordx4
// into registers, (c) global_load_dwordx4 + ds_write_b128, for the access shape the GEMM kernels use (8 rows x 128 B per
// wave instruction, rows `ld` bytes apart), on an L2-resident or HBM-sized buffer.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int MODE>
__global__ __launch_bounds__(512) void feed_kernel(const char* __restrict__ buf, size_t span, int ld, int iters, unsigned* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // workgroup w walks its own region; each wave instruction = 8 rows x 128 B
  const size_t base = ((size_t)blockIdx.x * 0x9E3779B1u) % (span / 2);
  const char* p = buf + (base & ~(size_t)4095) + (size_t)(wave * 8 + (lane >> 3)) * ld + (lane & 7) * 16;
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (int it = 0; it < iters; ++it) {
    const char* q = p + (size_t)(it & 63) * 128;       // walk along the rows (next K tile)
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(q + (size_t)i * 64 * ld), (lds_ptr_t)(smem + ((it & 3) * 32 + i * 8 + wave) * 1024), 16, 0, 0);
      if ((it & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      uint4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const uint4*>(q + (size_t)i * 64 * ld);
      if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(smem + ((it & 3) * 32 + i * 8 + wave) * 1024 + lane * 16) = v[i];
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc.x ^= v[i].x; acc.y ^= v[i].y; acc.z ^= v[i].z; acc.w ^= v[i].w; }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (MODE != 1) acc.x ^= *reinterpret_cast<unsigned*>(smem + threadIdx.x * 4);
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

int main(int argc, char** argv) {
  const size_t span = (argc > 1 ? atol(argv[1]) : 64) << 20;       // MiB
  const int ld = argc > 2 ? atoi(argv[2]) : 1536;
  const int iters = 2000;
  char* buf;
  unsigned* sink;
  hipMalloc(&buf, span + (64 << 20));
  hipMemset(buf, 1, span + (64 << 20));
  hipMalloc(&sink, 4);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const char* names[3] = {"lds-dma b128", "global_load b128 -> vgpr", "global_load b128 + ds_write_b128"};
  for (int mode = 0; mode < 3; ++mode) {
    for (int wgs : {256, 512}) {
      auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(feed_kernel<0>, dim3(wgs), dim3(512), 131072, 0, buf, span, ld, iters, sink);
        if (mode == 1) hipLaunchKernelGGL(feed_kernel<1>, dim3(wgs), dim3(512), 131072, 0, buf, span, ld, iters, sink);
        if (mode == 2) hipLaunchKernelGGL(feed_kernel<2>, dim3(wgs), dim3(512), 131072, 0, buf, span, ld, iters, sink);
      };
      hipFuncSetAttribute(reinterpret_cast<const void*>(feed_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      hipFuncSetAttribute(reinterpret_cast<const void*>(feed_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      hipFuncSetAttribute(reinterpret_cast<const void*>(feed_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
      launch();
      hipDeviceSynchronize();
      hipEventRecord(a);
      launch();
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      const double bytes = (double)wgs * 8 * 4 * 1024.0 * iters;
      printf("%-34s span %4zu MiB ld %5d wgs %3d: %7.3f ms  %7.1f GB/s  = %5.1f B/clk/CU @2.4GHz (256 CUs)\n", names[mode], span >> 20, ld, wgs, ms,
             bytes / ms / 1e6, bytes / (ms * 1e-3) / 256 / 2.4e9);
    }
  }
  return 0;
}
