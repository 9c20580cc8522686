// l1.hip -- row E1: L1.forward (objective.py:103-117) as un-normalised sums + the sign gradient.
//   sums[0] = sum_{b, f < frame_lengths[b], k} | log_pred - log(linear_tar + eps) |,  sums[1] = element count
// sums is double[2] (exact element count for any batch; fp64 atomics across workgroups).
// One pass, HBM-bound (8 B read per element, +4 B if the gradient is stored); per-workgroup partial sums,
// one float atomic per workgroup.
#include "common.h"

namespace se {

// len_div > 0: `frame_lengths` holds WAVEFORM lengths and the frame count is lengths / len_div + 1 (runner.py:455: `lengths // hop + 1`), so the
// caller needs no element-wise torch kernels in front of this one.
// FUSED (se_l1_masked_loss_f32): `sums` is a PERSISTENT scratch {arrival ticket, -, slab of per-workgroup partial sums} whose ticket is zero on entry;
// the last workgroup to arrive folds the slab, writes {sum, count} and the loss = sum / count to `out` / `loss` and clears the ticket again -- no zeroing launch in front, no division kernels
// behind (the reference's criterion is one call: objective.py:103-117).
template <int FUSED>
__global__ __launch_bounds__(256) void l1_kernel(const float* __restrict__ log_pred, const float* __restrict__ linear_tar,
                                                 const int64_t* __restrict__ frame_lengths, int len_div, int F, int K, float eps,
                                                 double* __restrict__ sums, float* __restrict__ grad, double* __restrict__ out, float* __restrict__ loss) {
  __shared__ float red[4];
  const int b = blockIdx.y;
  // 32-bit arithmetic on purpose: a 64-bit division per thread is ~100 instructions (it made this launch 8 us slower than the form it replaces)
  const int wl = (int)min(frame_lengths[b], (int64_t)0x7fffffff);
  const int fl = len_div > 0 ? wl / len_div + 1 : wl;
  const int len = min(F, fl);
  const size_t base = (size_t)b * F * K;
  const int n_valid = len * K;
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
        // libm logf (v_log_f32 x ln 2 measured 1.7 us faster per launch: not worth leaving the reference's arithmetic)
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
  bool last = false;
  if (threadIdx.x == 0) {
    const double part = (double)red[0] + (double)red[1] + (double)red[2] + (double)red[3];
    if (!FUSED) {
      atomicAdd(&sums[0], part);
      if (blockIdx.x == 0) atomicAdd(&sums[1], (double)n_valid);
    } else {
      // partial sum -> its own slot of the slab (write-through store, no contention), drained, then ONE arrival ticket: the same number of
      // same-address atomics as the un-fused form (whose two accumulator adds they replace), and the sum order is fixed (reproducible loss)
      double* slab = sums + 2;
      __hip_atomic_store(&slab[blockIdx.y * gridDim.x + blockIdx.x], part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      unsigned long long* ticket = reinterpret_cast<unsigned long long*>(sums);
      const unsigned long long total = (unsigned long long)gridDim.x * gridDim.y;
      last = atomicAdd(ticket, 1ull) == total - 1;
    }
  }
  if (FUSED) {
    // the last workgroup to arrive folds the slab (all 256 threads) and publishes {sum, count, loss}
    __shared__ int s_last;
    __shared__ double dred[2][4];
    if (threadIdx.x == 0) s_last = last;
    __syncthreads();
    if (!s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const int total = gridDim.x * gridDim.y;
    const double* slab = sums + 2;
    double acc = 0.0, cnt = 0.0;
    for (int i = threadIdx.x; i < total; i += 256) acc += __hip_atomic_load(&slab[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int bb = threadIdx.x; bb < (int)gridDim.y; bb += 256) {
      const int wl2 = (int)min(frame_lengths[bb], (int64_t)0x7fffffff);
      cnt += (double)(min(F, len_div > 0 ? wl2 / len_div + 1 : wl2) * K);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      acc += __shfl_down(acc, off);
      cnt += __shfl_down(cnt, off);
    }
    if ((threadIdx.x & 63) == 0) { dred[0][threadIdx.x >> 6] = acc; dred[1][threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
      const double sum = (dred[0][0] + dred[0][1]) + (dred[0][2] + dred[0][3]);
      const double n = (dred[1][0] + dred[1][1]) + (dred[1][2] + dred[1][3]);
      out[0] = sum;
      out[1] = n;
      *loss = (float)(sum / n);
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(sums), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ticket back to zero
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

extern "C" size_t se_l1_scratch_doubles(int B) { return 2 + (size_t)std::max(1, std::min(64, 512 / std::max(B, 1))) * (size_t)std::max(B, 1); }

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
