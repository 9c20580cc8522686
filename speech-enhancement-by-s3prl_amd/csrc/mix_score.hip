// mix_score.hip -- the two stages either side of the hot path, batched on the device:
//   se_mix_f32    OnlineDataset.__getitem__'s arithmetic (dataset.py:141-161): normalize_wav_decibel of speech and noise
//                 (dataset.py:106-111), noise tiled / cut to the speech length and mixed at the SNR by add_noise
//                 (dataset.py:54-74), stacked as (noisy, clean, scaled noise) and zero padded as collate_fn does
//                 (dataset.py:169-179) -- one reduction pass + one write pass instead of 12 CPU DataLoader workers
//   se_sisdr_f32  evaluation.sisdr_eval (evaluation.py:5-10) for every utterance of a batch over its own length
//                 (runner.py:597-603 trims each wav to lengths[b]) -- one pass, fp64 sums, no D2H of the waveforms
// Both are HBM-bound: 8 B read per sample and pass, 12 B written per sample.
#include <math.h>
#include <algorithm>
#include "common.h"

namespace se {

// sums[b] = { sum_{i<Ls} s^2, sum_{i<Ln} n^2, sum_{i<Ls} n[i mod Ln]^2 }
__global__ __launch_bounds__(256) void mix_sums_kernel(const float* __restrict__ speech, int ld_s, const int64_t* __restrict__ len_s,
                                                       const float* __restrict__ noise, int ld_n, const int64_t* __restrict__ len_n,
                                                       const int64_t* __restrict__ off_n, double* __restrict__ sums) {
  __shared__ double red[3][4];
  const int b = blockIdx.y;
  const int64_t Ls = len_s[b], Ln = len_n[b];
  const float* s = speech + (size_t)b * ld_s;
  const float* n = noise + (size_t)b * ld_n + (off_n ? off_n[b] : 0);
  const int64_t top = Ls > Ln ? Ls : Ln;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < top; i += (int64_t)gridDim.x * 256) {
    if (i < Ls) {
      const float v = s[i];
      a0 += (double)v * v;
      const float w = n[Ln > 0 ? i % Ln : 0];
      a2 += (double)w * w;
    }
    if (i < Ln) {
      const float w = n[i];
      a1 += (double)w * w;
    }
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
  if (threadIdx.x < 3) atomicAdd(&sums[b * 3 + threadIdx.x], (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]));
}

__global__ __launch_bounds__(256) void mix_write_kernel(const float* __restrict__ speech, int ld_s, const int64_t* __restrict__ len_s,
                                                        const float* __restrict__ noise, int ld_n, const int64_t* __restrict__ len_n,
                                                        const int64_t* __restrict__ off_n, const float* __restrict__ snr_db,
                                                        const double* __restrict__ sums, int T_out, int normalize, float target_level_db,
                                                        float eps, float* __restrict__ wavs) {
  const int b = blockIdx.y;
  const int64_t Ls = len_s[b], Ln = len_n[b];
  const float* s = speech + (size_t)b * ld_s;
  const float* n = noise + (size_t)b * ld_n + (off_n ? off_n[b] : 0);
  // normalize_wav_decibel: x * 10^(level/20) / (rms + 1e-10)
  float cs = 1.f, cn = 1.f;
  if (normalize) {
    const float lvl = powf(10.f, target_level_db / 20.f);
    cs = lvl / ((float)sqrt(sums[b * 3 + 0] / (double)(Ls > 0 ? Ls : 1)) + 1e-10f);
    cn = lvl / ((float)sqrt(sums[b * 3 + 1] / (double)(Ln > 0 ? Ln : 1)) + 1e-10f);
  }
  // add_noise: scalar = sqrt(speech_power / (10^(snr/10) noise_power + eps)) on the normalised, tiled signals
  const float sp = cs * cs * (float)sums[b * 3 + 0], np_ = cn * cn * (float)sums[b * 3 + 2];
  const float scalar = sqrtf(sp / (powf(10.f, snr_db[b] / 10.f) * np_ + eps)) * cn;
  float* o = wavs + (size_t)b * 3 * T_out;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < T_out; i += (int64_t)gridDim.x * 256) {
    float clean = 0.f, sn = 0.f;
    if (i < Ls) {
      clean = cs * s[i];
      sn = scalar * n[Ln > 0 ? i % Ln : 0];
    }
    o[i] = clean + sn;
    o[(size_t)T_out + i] = clean;
    o[2 * (size_t)T_out + i] = sn;
  }
}

// sums[b] = { sum src tar, sum tar^2, sum src^2 } over i < lengths[b]
__global__ __launch_bounds__(256) void sisdr_sums_kernel(const float* __restrict__ src, const float* __restrict__ tar, int ld,
                                                         const int64_t* __restrict__ lengths, double* __restrict__ sums) {
  __shared__ double red[3][4];
  const int b = blockIdx.y;
  const int64_t L = lengths ? (lengths[b] < ld ? lengths[b] : ld) : ld;
  const float* s = src + (size_t)b * ld;
  const float* t = tar + (size_t)b * ld;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x; i < L; i += (int64_t)gridDim.x * 256) {
    const double x = s[i], y = t[i];
    a0 += x * y;
    a1 += y * y;
    a2 += x * x;
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
  if (threadIdx.x < 3) atomicAdd(&sums[b * 3 + threadIdx.x], (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]));
}

// alpha = <s,t> / (<t,t> + eps); ay = alpha t; sisdr = 10 log10(|ay|^2 / (|ay - s|^2 + eps) + eps)
__global__ void sisdr_final_kernel(const double* __restrict__ sums, int B, float eps, float* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double st = sums[b * 3 + 0], tt = sums[b * 3 + 1], ss = sums[b * 3 + 2];
  const double alpha = st / (tt + (double)eps);
  const double ay2 = alpha * alpha * tt;
  double err = ay2 - 2.0 * alpha * st + ss;      // |alpha t - s|^2 expanded (fp64: no cancellation trouble at speech SNRs)
  if (err < 0.0) err = 0.0;
  out[b] = (float)(10.0 * log10(ay2 / (err + (double)eps) + (double)eps));
}

}  // namespace se

extern "C" int se_mix_f32(const float* speech, int ld_s, const int64_t* len_s, const float* noise, int ld_n, const int64_t* len_n,
                          const int64_t* off_n, const float* snr_db, int B, int T_out, int normalize, float target_level_db, float eps,
                          float* wavs, double* sums, void* stream) {
  SE_REQUIRE(speech && len_s && noise && len_n && snr_db && wavs && sums, "se_mix_f32: null argument");
  SE_REQUIRE(B > 0 && B <= 65535 && T_out > 0 && ld_s > 0 && ld_n > 0, "se_mix_f32: bad shape");
  hipStream_t st = se::as_stream(stream);
  { const int zrc_ = se::zero_async(sums, sizeof(double) * 3 * B, st); if (zrc_) return zrc_; }
  const int span = ld_s > ld_n ? ld_s : ld_n;
  const int chunks = std::max(1, std::min(64, (span + 4095) / 4096));
  hipLaunchKernelGGL(se::mix_sums_kernel, dim3(chunks, B), dim3(256), 0, st, speech, ld_s, len_s, noise, ld_n, len_n, off_n, sums);
  SE_LAUNCH_CHECK();
  const int wchunks = std::max(1, std::min(64, (T_out + 2047) / 2048));
  hipLaunchKernelGGL(se::mix_write_kernel, dim3(wchunks, B), dim3(256), 0, st, speech, ld_s, len_s, noise, ld_n, len_n, off_n, snr_db, sums, T_out,
                     normalize, target_level_db, eps, wavs);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_sisdr_f32(const float* src, const float* tar, int ld, const int64_t* lengths, int B, float eps, double* sums, float* sisdr,
                            void* stream) {
  SE_REQUIRE(src && tar && sums && sisdr && B > 0 && B <= 65535 && ld > 0, "se_sisdr_f32: bad argument");
  hipStream_t st = se::as_stream(stream);
  { const int zrc_ = se::zero_async(sums, sizeof(double) * 3 * B, st); if (zrc_) return zrc_; }
  const int chunks = std::max(1, std::min(64, (ld + 4095) / 4096));
  hipLaunchKernelGGL(se::sisdr_sums_kernel, dim3(chunks, B), dim3(256), 0, st, src, tar, ld, lengths, sums);
  SE_LAUNCH_CHECK();
  hipLaunchKernelGGL(se::sisdr_final_kernel, dim3((B + 255) / 256), dim3(256), 0, st, sums, B, eps, sisdr);
  SE_LAUNCH_CHECK();
  return SE_OK;
}
