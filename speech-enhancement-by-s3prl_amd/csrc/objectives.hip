// objectives.hip -- the spectrogram-domain training criteria besides L1 (objective.py:81-100 `SISDR`, objective.py:119-153 `WSD`;
// SURVEY.md section 8f rank 5): the losses the mask heads (LinearResidual / Residual: `predicted`, `offset`) are trained with.
// HBM-bound two-pass reductions: per-utterance fp64 sums, then one elementwise pass that writes the gradient.
#include <math.h>
#include <algorithm>
#include "common.h"

namespace se {

__device__ __forceinline__ void block_add3(double a0, double a1, double a2, double* dst) {
  __shared__ double red[3][4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a0 += __shfl_xor(a0, off);
    a1 += __shfl_xor(a1, off);
    a2 += __shfl_xor(a2, off);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = a0;
    red[1][threadIdx.x >> 6] = a1;
    red[2][threadIdx.x >> 6] = a2;
  }
  __syncthreads();
  if (threadIdx.x < 3) atomicAdd(&dst[threadIdx.x], (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]));
}

// SISDR: src = sqrt(relu(predicted)) mask, tar = sqrt(relu(linear_tar)) mask, flattened per utterance.
// sums[b] = { <src, tar>, <tar, tar>, <src, src> } over frames < lengths[b]
__global__ __launch_bounds__(256) void sisdr_spec_sums_kernel(const float* __restrict__ pred, const float* __restrict__ tar,
                                                              const int64_t* __restrict__ lengths, int F, int N, double* __restrict__ sums) {
  const int b = blockIdx.y;
  const int64_t L = min((int64_t)F, max((int64_t)0, lengths[b])) * N;
  const float* p = pred + (size_t)b * F * N;
  const float* t = tar + (size_t)b * F * N;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  // four independent element pairs per trip: eight loads in flight per thread instead of two (the launch was latency-bound at 3.6 TB/s); same fp64 sums
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
  for (; i + 3 * stride < L; i += 4 * stride) {
    float pv[4], tv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { pv[u] = p[i + u * stride]; tv[u] = t[i + u * stride]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float s = sqrtf(fmaxf(pv[u], 0.f)), y = sqrtf(fmaxf(tv[u], 0.f));
      a0 += (double)s * y;
      a1 += (double)y * y;
      a2 += (double)s * s;
    }
  }
  for (; i < L; i += stride) {
    const float s = sqrtf(fmaxf(p[i], 0.f)), y = sqrtf(fmaxf(t[i], 0.f));
    a0 += (double)s * y;
    a1 += (double)y * y;
    a2 += (double)s * s;
  }
  block_add3(a0, a1, a2, sums + 3 * b);
}

// loss_b = -10 log10(|a y|^2 / (|a y - s|^2 + eps) + eps), a = <s,y> / (<y,y> + eps); coef[b] = { c_t, c_s }: dL_b/ds_i = c_t y_i + c_s s_i
__global__ void sisdr_spec_final_kernel(const double* __restrict__ sums, int B, float eps, float* __restrict__ loss_b, double* __restrict__ coef) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double S = sums[3 * b], T2 = sums[3 * b + 1], Q = sums[3 * b + 2], e = (double)eps;
  const double kappa = 1.0 / (T2 + e), a = S * kappa;
  const double ay2 = a * a * T2;
  const double norm = a * a * T2 - 2.0 * a * S + Q + e;
  const double R = ay2 / norm + e;
  loss_b[b] = (float)(-10.0 * log10(R));
  const double K0 = -(10.0 / log(10.0)) / (R * norm * norm);
  coef[2 * b] = K0 * (2.0 * a * T2 * kappa * norm - ay2 * ((2.0 * a * T2 - 2.0 * S) * kappa - 2.0 * a));
  coef[2 * b + 1] = K0 * (-2.0 * ay2);
}

// grad wrt predicted: (c_t y + c_s s) * d sqrt(relu(p)) / dp = (c_t y + c_s s) / (2 s) for p > 0 inside the mask, else 0
__global__ __launch_bounds__(256) void sisdr_spec_grad_kernel(const float* __restrict__ pred, const float* __restrict__ tar,
                                                              const int64_t* __restrict__ lengths, int F, int N, const double* __restrict__ coef,
                                                              float scale, float* __restrict__ grad) {
  const int b = blockIdx.y;
  const int64_t L = min((int64_t)F, max((int64_t)0, lengths[b])) * N, tot = (int64_t)F * N;
  const float ct = (float)coef[2 * b] * scale, cs = (float)coef[2 * b + 1] * scale;
  const float* p = pred + (size_t)b * F * N;
  const float* t = tar + (size_t)b * F * N;
  float* g = grad + (size_t)b * F * N;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    float v = 0.f;
    if (i < L && p[i] > 0.f) {
      const float s = sqrtf(p[i]), y = sqrtf(fmaxf(t[i], 0.f));
      v = (ct * y + cs * s) / (2.f * s);
    }
    g[i] = v;
  }
}

// WSD pass 1: energy[b][f] = sum_n S[b][f][n] (every frame, padded ones too, as the reference) and its global maximum
__global__ __launch_bounds__(256) void wsd_energy_kernel(const float* __restrict__ tar, int rows, int N, float* __restrict__ energy,
                                                         unsigned int* __restrict__ max_bits) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float mx = 0.f;
  for (int r = blockIdx.x * 4 + wv; r < rows; r += gridDim.x * 4) {
    float s = 0.f;
    for (int n = lane; n < N; n += 64) s += tar[(size_t)r * N + n];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) energy[r] = s;
    mx = fmaxf(mx, s);
  }
  // energies of power spectra are >= 0: their float bit patterns order like unsigned integers
  if (lane == 0) atomicMax(max_bits, __float_as_uint(fmaxf(mx, 0.f)));
}

// WSD pass 2: sums = { sum_b speech_b, sum_b noise_b }; grad = d(alpha speech_sum + (1 - alpha) noise_sum) / d offset * scale
__global__ __launch_bounds__(256) void wsd_kernel(const float* __restrict__ inp, const float* __restrict__ off, const float* __restrict__ tar,
                                                  const int64_t* __restrict__ lengths, const float* __restrict__ energy,
                                                  const unsigned int* __restrict__ max_bits, int F, int N, float alpha, float db_interval, float eps,
                                                  float scale, double* __restrict__ sums, float* __restrict__ grad) {
  const int b = blockIdx.y;
  const int64_t L = min((int64_t)F, max((int64_t)0, lengths[b])) * N, tot = (int64_t)F * N;
  const float db_thres = 10.f * log10f(__uint_as_float(*max_bits) + eps) - db_interval;
  const size_t base = (size_t)b * F * N;
  double a0 = 0.0, a1 = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    float g = 0.f;
    if (i < L) {
      const int f = (int)(i / N);
      const float S = tar[base + i], G = off[base + i];
      const float Nz = fmaxf(inp[base + i] - S, 0.f);
      const float vm = (10.f * log10f(energy[(size_t)b * F + f] + eps) > db_thres) ? 1.f : 0.f;
      const float sd = (S - G * S) * vm, nd = G * Nz;
      a0 += (double)sd * sd;
      a1 += (double)nd * nd;
      g = scale * (alpha * 2.f * sd * (-S) * vm + (1.f - alpha) * 2.f * nd * Nz);
    }
    if (grad) grad[base + i] = g;
  }
  block_add3(a0, a1, 0.0, sums);
}

// ---- round 5: the criterion of the evaluate()-style pass as TWO launches and nothing else (objective.py:81-100 is one call): no clearing launch
// (per-(utterance, chunk) partial sums go to their own slots of a slab instead of three atomics), frame counts derived inside from the waveform
// lengths (len_div > 0: frames = lengths / len_div + 1, runner.py:455), the mean over the utterances formed by the final launch (the torch tail of
// _SISDRFn.forward was six element-wise launches of ~5 us each).  The sum order is fixed: the loss is reproducible run to run.
__device__ __forceinline__ int frames_of(const int64_t* __restrict__ lengths, int b, int len_div, int F) {
  const int wl = (int)min(max(lengths[b], (int64_t)0), (int64_t)0x7fffffff);
  return min(F, len_div > 0 ? wl / len_div + 1 : wl);
}

__global__ __launch_bounds__(256) void sisdr_spec_slab_kernel(const float* __restrict__ pred, const float* __restrict__ tar,
                                                              const int64_t* __restrict__ lengths, int len_div, int F, int N, double* __restrict__ slab) {
  __shared__ double red[3][4];
  const int b = blockIdx.y;
  const int64_t L = (int64_t)frames_of(lengths, b, len_div, F) * N;
  const float* p = pred + (size_t)b * F * N;
  const float* t = tar + (size_t)b * F * N;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
  for (; i + 3 * stride < L; i += 4 * stride) {
    float pv[4], tv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { pv[u] = p[i + u * stride]; tv[u] = t[i + u * stride]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float s = sqrtf(fmaxf(pv[u], 0.f)), y = sqrtf(fmaxf(tv[u], 0.f));
      a0 += (double)s * y;
      a1 += (double)y * y;
      a2 += (double)s * s;
    }
  }
  for (; i < L; i += stride) {
    const float s = sqrtf(fmaxf(p[i], 0.f)), y = sqrtf(fmaxf(t[i], 0.f));
    a0 += (double)s * y;
    a1 += (double)y * y;
    a2 += (double)s * s;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a0 += __shfl_xor(a0, off);
    a1 += __shfl_xor(a1, off);
    a2 += __shfl_xor(a2, off);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = a0;
    red[1][threadIdx.x >> 6] = a1;
    red[2][threadIdx.x >> 6] = a2;
  }
  __syncthreads();
  if (threadIdx.x < 3)
    slab[((size_t)b * gridDim.x + blockIdx.x) * 3 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// one workgroup: loss_b[b] from the slab, then out = {sum_b loss_b, B} and loss = their ratio.  per_b > 0: utterance b owns per_b slots of 3 doubles
// (sisdr_spec_slab_kernel); per_b == 0: the mask head's fused form (head.hip: head5_kernel) -- every `tile`-row workgroup of the (B F, N) plane left
// two slots, {first utterance it touches, second}, and utterance b collects its own from the workgroups its rows fall into
__global__ __launch_bounds__(256) void sisdr_spec_mean_kernel(const double* __restrict__ slab, int per_b, int B, float eps, float* __restrict__ loss_b,
                                                              double* __restrict__ out, float* __restrict__ loss, int F = 0, int tile = 1) {
  __shared__ double red[4];
  double acc = 0.0;
  for (int b = threadIdx.x; b < B; b += 256) {
    double S = 0.0, T2 = 0.0, Q = 0.0;
    if (per_b > 0) {
      const double* r = slab + (size_t)b * per_b * 3;
      for (int c = 0; c < per_b; ++c) { S += r[3 * c]; T2 += r[3 * c + 1]; Q += r[3 * c + 2]; }
    } else {
      const long long lo = (long long)b * F, hi = lo + F - 1;
      for (long long wg = lo / tile; wg <= hi / tile; ++wg) {
        const int u = b - (int)((wg * tile) / F);                  // 0: b is the first utterance of workgroup wg, 1: the second
        const double* r = slab + ((size_t)wg * 2 + u) * 3;
        S += r[0]; T2 += r[1]; Q += r[2];
      }
    }
    const double e = (double)eps;
    const double a = S / (T2 + e);
    const double ay2 = a * a * T2;
    const double norm = a * a * T2 - 2.0 * a * S + Q + e;
    const float lb = (float)(-10.0 * log10(ay2 / norm + e));
    loss_b[b] = lb;
    acc += (double)lb;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double sum = (red[0] + red[1]) + (red[2] + red[3]);
    out[0] = sum;
    out[1] = (double)B;
    *loss = (float)(sum / (double)B);
  }
}

}  // namespace se

// partial-sum slots per utterance: enough workgroups to fill the chip (~2 048 in all), no more -- the one-workgroup mean launch walks them
static int sisdr_loss_chunks(int B, int F, int N) {
  const int64_t by_size = ((int64_t)F * N + 8191) / 8192;
  return (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(64, by_size), std::max(1, 2048 / B)));
}

extern "C" size_t se_sisdr_spec_loss_scratch_doubles(int B, int F, int N) { return (size_t)B * sisdr_loss_chunks(B, F, N) * 3; }

extern "C" int se_sisdr_spec_loss_f32(const float* predicted, const float* linear_tar, const int64_t* lengths, int len_div, int B, int F, int N, float eps,
                                      double* scratch, float* loss_b, double* sums_out, float* loss_out, void* stream) {
  SE_REQUIRE(predicted && linear_tar && lengths && scratch && loss_b && sums_out && loss_out, "se_sisdr_spec_loss_f32: null argument");
  SE_REQUIRE(B > 0 && B <= 65535 && F > 0 && N > 0 && len_div >= 0, "se_sisdr_spec_loss_f32: bad shape");
  hipStream_t st = se::as_stream(stream);
  const int chunks = sisdr_loss_chunks(B, F, N);
  hipLaunchKernelGGL(se::sisdr_spec_slab_kernel, dim3(chunks, B), dim3(256), 0, st, predicted, linear_tar, lengths, len_div, F, N, scratch);
  SE_LAUNCH_CHECK();
  hipLaunchKernelGGL(se::sisdr_spec_mean_kernel, dim3(1), dim3(256), 0, st, scratch, chunks, B, eps, loss_b, sums_out, loss_out);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_sisdr_head_mean_f32(const double* slab, int B, int F, int tile_rows, float eps, float* loss_b, double* sums_out, float* loss_out, void* stream) {
  SE_REQUIRE(slab && loss_b && sums_out && loss_out && B > 0 && F >= tile_rows && tile_rows > 0, "se_sisdr_head_mean_f32: bad argument");
  hipLaunchKernelGGL(se::sisdr_spec_mean_kernel, dim3(1), dim3(256), 0, se::as_stream(stream), slab, 0, B, eps, loss_b, sums_out, loss_out, F, tile_rows);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_sisdr_spec_f32(const float* predicted, const float* linear_tar, const int64_t* frame_lengths, int B, int F, int N, float eps,
                                 float grad_scale, double* scratch, float* loss_b, float* grad, void* stream) {
  SE_REQUIRE(predicted && linear_tar && frame_lengths && scratch && loss_b, "se_sisdr_spec_f32: null argument");
  SE_REQUIRE(B > 0 && B <= 65535 && F > 0 && N > 0, "se_sisdr_spec_f32: bad shape");
  hipStream_t st = se::as_stream(stream);
  { const int zrc_ = se::zero_async(scratch, sizeof(double) * 5 * B, st); if (zrc_) return zrc_; }
  const int chunks = (int)std::max<int64_t>(1, std::min<int64_t>(64, ((int64_t)F * N + 8191) / 8192));
  hipLaunchKernelGGL(se::sisdr_spec_sums_kernel, dim3(chunks, B), dim3(256), 0, st, predicted, linear_tar, frame_lengths, F, N, scratch);
  SE_LAUNCH_CHECK();
  double* coef = scratch + 3 * (size_t)B;
  hipLaunchKernelGGL(se::sisdr_spec_final_kernel, dim3((B + 255) / 256), dim3(256), 0, st, scratch, B, eps, loss_b, coef);
  SE_LAUNCH_CHECK();
  if (grad) {
    hipLaunchKernelGGL(se::sisdr_spec_grad_kernel, dim3(chunks, B), dim3(256), 0, st, predicted, linear_tar, frame_lengths, F, N, coef, grad_scale, grad);
    SE_LAUNCH_CHECK();
  }
  return SE_OK;
}

extern "C" int se_wsd_energy_f32(const float* linear_tar, int B, int F, int N, float* energy, float* energy_max, void* stream) {
  SE_REQUIRE(linear_tar && energy && energy_max && B > 0 && F > 0 && N > 0, "se_wsd_energy_f32: bad argument");
  hipStream_t st = se::as_stream(stream);
  { const int zrc_ = se::zero_async(energy_max, sizeof(float), st); if (zrc_) return zrc_; }
  const int rows = B * F;
  hipLaunchKernelGGL(se::wsd_energy_kernel, dim3(std::min(1024, (rows + 3) / 4)), dim3(256), 0, st, linear_tar, rows, N, energy,
                     reinterpret_cast<unsigned int*>(energy_max));
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_wsd_f32(const float* linear_inp, const float* offset, const float* linear_tar, const int64_t* frame_lengths, const float* energy,
                          const float* energy_max, int B, int F, int N, float alpha, float db_interval, float eps, float grad_scale, double* sums,
                          float* grad, void* stream) {
  SE_REQUIRE(linear_inp && offset && linear_tar && frame_lengths && energy && energy_max && sums, "se_wsd_f32: null argument");
  SE_REQUIRE(B > 0 && B <= 65535 && F > 0 && N > 0, "se_wsd_f32: bad shape");
  hipStream_t st = se::as_stream(stream);
  { const int zrc_ = se::zero_async(sums, sizeof(double) * 3, st); if (zrc_) return zrc_; }
  const int chunks = (int)std::max<int64_t>(1, std::min<int64_t>(64, ((int64_t)F * N + 8191) / 8192));
  hipLaunchKernelGGL(se::wsd_kernel, dim3(chunks, B), dim3(256), 0, st, linear_inp, offset, linear_tar, frame_lengths, energy,
                     reinterpret_cast<const unsigned int*>(energy_max), F, N, alpha, db_interval, eps, grad_scale, sums, grad);
  SE_LAUNCH_CHECK();
  return SE_OK;
}
