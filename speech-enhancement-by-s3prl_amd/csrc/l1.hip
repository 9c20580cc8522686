// l1.hip -- row E1: L1.forward (objective.py:103-117) as un-normalised sums + the sign gradient.
//   sums[0] = sum_{b, f < frame_lengths[b], k} | log_pred - log(linear_tar + eps) |,  sums[1] = element count
// sums is double[2] (exact element count for any batch; fp64 atomics across workgroups).
// One pass, HBM-bound (8 B read per element, +4 B if the gradient is stored); per-workgroup partial sums,
// one float atomic per workgroup.
#include "common.h"

namespace se {

__global__ __launch_bounds__(256) void l1_kernel(const float* __restrict__ log_pred, const float* __restrict__ linear_tar,
                                                 const int64_t* __restrict__ frame_lengths, int F, int K, float eps,
                                                 double* __restrict__ sums, float* __restrict__ grad) {
  __shared__ float red[4];
  const int b = blockIdx.y;
  const int64_t len = min((int64_t)F, frame_lengths[b]);
  const size_t base = (size_t)b * F * K;
  const int n_valid = (int)len * K;
  const int n_all = F * K;
  float s = 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n_all; i += gridDim.x * 256) {
    if (i < n_valid) {
      const float d = log_pred[base + i] - logf(linear_tar[base + i] + eps);
      s += fabsf(d);
      if (grad) grad[base + i] = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
    } else if (grad) {
      grad[base + i] = 0.f;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&sums[0], (double)red[0] + (double)red[1] + (double)red[2] + (double)red[3]);
    if (blockIdx.x == 0) atomicAdd(&sums[1], (double)n_valid);
  }
}

}  // namespace se

extern "C" int se_l1_masked_f32(const float* log_pred, const float* linear_tar, const int64_t* frame_lengths,
                                int B, int F, int K, float eps, double* sums, float* grad, void* stream) {
  SE_REQUIRE(log_pred && linear_tar && frame_lengths && sums, "se_l1_masked_f32: null argument");
  SE_REQUIRE(B > 0 && B <= 65535 && F > 0 && K > 0, "se_l1_masked_f32: bad shape");
  hipStream_t st = se::as_stream(stream);
  { const int zrc_ = se::zero_async(sums, 2 * sizeof(double), st); if (zrc_) return zrc_; }
  const int n = F * K;
  dim3 grid(std::min(64, (n + 255) / 256), B);
  hipLaunchKernelGGL(se::l1_kernel, grid, dim3(256), 0, st, log_pred, linear_tar, frame_lengths, F, K, eps, sums, grad);
  SE_LAUNCH_CHECK();
  return SE_OK;
}
