// store.hip -- developer microbenchmark behind the store-shape numbers of DESIGN.md section 5: the same bytes written by wave instructions of
// three shapes -- 16 rows x 32 B (an 8-B lane store of a 16-row MFMA tile), 16 rows x 64 B (16-B lane stores after the lane-pair exchange of
// the GEMM epilogues), one contiguous 1 KiB row segment -- into a (rows, 2304) bf16 matrix (the QKV projection's 147 MB output).
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/store.hip -o tools/micro/store ; run on the GPU box: tools/micro/store
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__global__ __launch_bounds__(512) void store_kernel(unsigned short* __restrict__ out, int rows, int ld, int tiles_n) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  // a 256 x 256 tile per workgroup, waves as 2 x 4 (wave tile 128 x 64) like the GEMMs
  const int r0 = tm * 256 + (wave >> 2) * 128, c0 = tn * 256 + (wave & 3) * 64;
  const uint4 v = make_uint4(lane, wave, blockIdx.x, 7);
  if (MODE == 0) {          // 16 rows x 32 B per instruction: lane -> row lane & 15, 4 bf16 at column 4 (lane >> 4) of a 16-column tile
    for (int i = 0; i < 8; ++i)
      for (int j = 0; j < 4; ++j) {
        const int r = min(r0 + i * 16 + (lane & 15), rows - 1);
        *reinterpret_cast<uint2*>(out + (size_t)r * ld + c0 + j * 16 + 4 * (lane >> 4)) = make_uint2(v.x, v.y);
      }
  } else if (MODE == 1) {   // 16 rows x 64 B: lane -> row lane & 15, 8 bf16 at column 8 (lane >> 4) of a 32-column tile pair
    for (int i = 0; i < 8; ++i)
      for (int j = 0; j < 2; ++j) {
        const int r = min(r0 + i * 16 + (lane & 15), rows - 1);
        *reinterpret_cast<uint4*>(out + (size_t)r * ld + c0 + j * 32 + 8 * (lane >> 4)) = v;
      }
  } else {                  // row-contiguous: one instruction = 8 rows x 128 B (lane -> row lane >> 3, 16 B at lane & 7)
    for (int i = 0; i < 16; ++i) {
      const int r = min(r0 + i * 8 + (lane >> 3), rows - 1);
      *reinterpret_cast<uint4*>(out + (size_t)r * ld + c0 + 8 * (lane & 7)) = v;
    }
  }
}

template <int MODE>
static float run(unsigned short* d, int rows, int ld) {
  const int tiles_n = ld / 256, tiles_m = (rows + 255) / 256;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(store_kernel<MODE>, dim3(tiles_m * tiles_n), dim3(512), 0, 0, d, rows, ld, tiles_n);
  hipEventRecord(a);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(store_kernel<MODE>, dim3(tiles_m * tiles_n), dim3(512), 0, 0, d, rows, ld, tiles_n);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  return ms / 20.f;
}

int main() {
  const int rows = 32032, ld = 2304;
  unsigned short* d = nullptr;
  if (hipMalloc(&d, (size_t)rows * ld * 2) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
  const double bytes = (double)rows * ld * 2;
  const float t0 = run<0>(d, rows, ld), t1 = run<1>(d, rows, ld), t2 = run<2>(d, rows, ld);
  printf("16 rows x 32 B : %7.1f us  %5.2f TB/s\n16 rows x 64 B : %7.1f us  %5.2f TB/s\n 8 rows x 128 B: %7.1f us  %5.2f TB/s\n", t0 * 1e3, bytes / t0 / 1e9,
         t1 * 1e3, bytes / t1 / 1e9, t2 * 1e3, bytes / t2 / 1e9);
  hipFree(d);
  return 0;
}
