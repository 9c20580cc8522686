// l1.hip -- row E1: L1.forward (objective.py:103-117) as un-normalised sums + the sign gradient.
//   sums[0] = sum_{b, f < frame_lengths[b], k} | log_pred - log(linear_tar + eps) |,  sums[1] = element count
// sums is double[2] (exact element count for any batch; fp64 atomics across workgroups).
// One pass, HBM-bound (8 B read per element, +4 B if the gradient is stored); per-workgroup partial sums,
// one float atomic per workgroup.
#include "common.h"

namespace se {

// len_div > 0: `frame_lengths` holds WAVEFORM lengths and the frame count is lengths / len_div + 1 (runner.py:455: `lengths // hop + 1`), so the
// caller needs no element-wise torch kernels in front of this one.
// FUSED (se_l1_masked_loss_f32): `sums` is a PERSISTENT scratch {sum, count, arrival ticket} that is zero on entry; the last workgroup to arrive
// writes {sum, count} and the loss = sum / count to `out` / `loss` and clears the scratch again -- no zeroing launch in front, no division kernels
// behind (the reference's criterion is one call: objective.py:103-117).
template <int FUSED>
__global__ __launch_bounds__(256) void l1_kernel(const float* __restrict__ log_pred, const float* __restrict__ linear_tar,
                                                 const int64_t* __restrict__ frame_lengths, int len_div, int F, int K, float eps,
                                                 double* __restrict__ sums, float* __restrict__ grad, double* __restrict__ out, float* __restrict__ loss) {
  __shared__ float red[4];
  const int b = blockIdx.y;
  const int64_t fl = len_div > 0 ? frame_lengths[b] / len_div + 1 : frame_lengths[b];
  const int64_t len = min((int64_t)F, fl);
  const size_t base = (size_t)b * F * K;
  const int n_valid = (int)len * K;
  const int n_all = F * K;
  float s = 0.f;
  // four independent elements per thread and trip (all loads of a trip issued before the first logf: the one-element loop was
  // latency-bound)
  const int stride = gridDim.x * 256;
  for (int i0 = blockIdx.x * 256 + threadIdx.x; i0 < n_all; i0 += 4 * stride) {
    float lp[4], lt[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = i0 + j * stride;
      const bool ok = i < n_valid;
      lp[j] = ok ? log_pred[base + i] : 0.f;
      lt[j] = ok ? linear_tar[base + i] : 1.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i = i0 + j * stride;
      if (i < n_valid) {
        const float d = lp[j] - logf(lt[j] + eps);
        s += fabsf(d);
        if (grad) grad[base + i] = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
      } else if (grad && i < n_all) {
        grad[base + i] = 0.f;
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&sums[0], (double)red[0] + (double)red[1] + (double)red[2] + (double)red[3]);
    if (blockIdx.x == 0) atomicAdd(&sums[1], (double)n_valid);
    if (FUSED) {
      // arrival ticket (device-scope atomics all the way: the sums are only ever touched by atomics, so the last arriver's atomic reads see them)
      __threadfence();
      unsigned long long* ticket = reinterpret_cast<unsigned long long*>(sums + 2);
      const unsigned long long total = (unsigned long long)gridDim.x * gridDim.y;
      if (atomicAdd(ticket, 1ull) == total - 1) {
        __threadfence();
        const double sum = atomicAdd(&sums[0], 0.0), cnt = atomicAdd(&sums[1], 0.0);
        out[0] = sum;
        out[1] = cnt;
        *loss = (float)(sum / cnt);
        atomicExch(reinterpret_cast<unsigned long long*>(&sums[0]), 0ull);       // leave the scratch zero for the next call
        atomicExch(reinterpret_cast<unsigned long long*>(&sums[1]), 0ull);
        atomicExch(ticket, 0ull);
      }
    }
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
  // ~512 workgroups in total: every workgroup ends in one fp64 atomic on the same address, and 2 048 of them (64 per utterance at
  // B = 32) serialised for ~20 us of a 34 us launch
  const int per_utt = std::max(1, std::min(64, 512 / B));
  dim3 grid(std::min(per_utt, (n + 255) / 256), B);
  hipLaunchKernelGGL(se::l1_kernel<0>, grid, dim3(256), 0, st, log_pred, linear_tar, frame_lengths, 0, F, K, eps, sums, grad, nullptr, nullptr);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_l1_masked_loss_f32(const float* log_pred, const float* linear_tar, const int64_t* lengths, int len_div, int B, int F, int K, float eps,
                                     double* scratch3, double* sums_out, float* loss_out, float* grad, void* stream) {
  SE_REQUIRE(log_pred && linear_tar && lengths && scratch3 && sums_out && loss_out, "se_l1_masked_loss_f32: null argument");
  SE_REQUIRE(B > 0 && B <= 65535 && F > 0 && K > 0 && len_div >= 0, "se_l1_masked_loss_f32: bad shape");
  const int n = F * K;
  const int per_utt = std::max(1, std::min(64, 512 / B));
  dim3 grid(std::min(per_utt, (n + 255) / 256), B);
  hipLaunchKernelGGL(se::l1_kernel<1>, grid, dim3(256), 0, se::as_stream(stream), log_pred, linear_tar, lengths, len_div, F, K, eps, scratch3, grad,
                     sums_out, loss_out);
  SE_LAUNCH_CHECK();
  return SE_OK;
}
