// zero.hip -- se::zero_async: clears a small device buffer with a KERNEL on the caller's stream.  The library's accumulators
// (atomic sums, counters, bias gradients) used hipMemsetAsync; inside a hipGraph capture (torch.cuda.CUDAGraph) those small
// memset nodes were not replayed on this ROCm build -- the second replay accumulated onto the first -- while kernel nodes
// always are.  Same cost as the memset it replaces (one tiny launch).
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

}  // namespace se
