// zero.hip -- se::zero_async: clears a small device buffer with a KERNEL on the caller's stream.  The library's accumulators
// (atomic sums, counters, bias gradients) used hipMemsetAsync; inside a hipGraph capture (torch.cuda.CUDAGraph) those small
// memset nodes were not replayed on this ROCm build -- the second replay accumulated onto the first -- while kernel nodes
// always are.  Same cost as the memset it replaces (one tiny launch).
#include <algorithm>
#include "common.h"

namespace se {

__global__ __launch_bounds__(256) void zero_kernel(uint32_t* __restrict__ p, size_t n4, unsigned char* __restrict__ tail, int ntail) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = 0u;
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}

int zero_async(void* ptr, size_t bytes, hipStream_t st) {
  if (!ptr || bytes == 0) return SE_OK;
  // 4-byte words from the first aligned address; every buffer the library clears is at least 4-byte aligned
  if ((reinterpret_cast<uintptr_t>(ptr) & 3) != 0) {
    set_error("zero_async: buffer must be 4-byte aligned");
    return SE_ERR_INVALID;
  }
  const size_t n4 = bytes / 4;
  const int ntail = (int)(bytes & 3);
  const unsigned grid = (unsigned)std::min<size_t>((n4 + 255) / 256 + 1, 4096);
  hipLaunchKernelGGL(zero_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<uint32_t*>(ptr), n4,
                     reinterpret_cast<unsigned char*>(ptr) + n4 * 4, ntail);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

// up to three 4-byte-aligned buffers (bytes % 4 == 0) in ONE launch: blockIdx.y selects the buffer (the LayerNorm backward clears its three
// parameter-gradient rows before every launch: 39 -> 13 zeroing launches per fine-tune step)
struct Zero3 {
  uint32_t* p[3];
  size_t n4[3];
};

__global__ __launch_bounds__(256) void zero3_kernel(Zero3 z) {
  uint32_t* p = z.p[blockIdx.y];
  const size_t n4 = z.n4[blockIdx.y];
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) p[i] = 0u;
}

int zero_async3(void* p0, size_t b0, void* p1, size_t b1, void* p2, size_t b2, hipStream_t st) {
  Zero3 z;
  void* ps[3] = {p0, p1, p2};
  const size_t bs[3] = {b0, b1, b2};
  int n = 0;
  size_t mx = 0;
  for (int i = 0; i < 3; ++i) {
    if (!ps[i] || bs[i] == 0) continue;
    if ((reinterpret_cast<uintptr_t>(ps[i]) & 3) != 0 || (bs[i] & 3) != 0) {
      set_error("zero_async3: buffers must be 4-byte aligned and a multiple of 4 bytes long");
      return SE_ERR_INVALID;
    }
    z.p[n] = reinterpret_cast<uint32_t*>(ps[i]);
    z.n4[n] = bs[i] / 4;
    mx = std::max(mx, z.n4[n]);
    ++n;
  }
  if (n == 0) return SE_OK;
  for (int i = n; i < 3; ++i) { z.p[i] = z.p[0]; z.n4[i] = 0; }
  const unsigned grid = (unsigned)std::min<size_t>((mx + 255) / 256, 4096);
  hipLaunchKernelGGL(zero3_kernel, dim3(grid, n), dim3(256), 0, st, z);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

}  // namespace se
