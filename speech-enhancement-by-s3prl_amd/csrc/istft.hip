// istft.hip -- rows A6 + D1 + D2: power/phase -> waveform (inverse 400-point real FFT, window, overlap-add,
// envelope division, trim) with the masked square-sum for the dB normalisation fused in; plus the
// dB-normalise scale pass, the masked square-sum of a reference waveform and the length masks.
//
// iSTFT workgroup = 256 threads = 29 hop-blocks (4 640 output samples) of one utterance; it inverse-
// transforms the 32 frames that overlap that span (3 of them shared with the neighbours, re-read from L2):
//   load   : X[k] = sqrt(P) (cos phi, sin phi), Im X[0] = Im X[200] = 0 (c2r semantics); X[200].re rides in X[0].y
//   fold   : Z[k] = E[k] + i O[k]  in place on pairs (k, 200-k)
//   pass A / pass B : fft200.h with DIR = +1
//   ola    : out[n] = sum_f w[n - 160 f] z_f[n - 160 f] / sum_f w^2[n - 160 f]      (1/200 folded into w)
// LDS 51 200 B -> 3 workgroups per CU.  Bound: HBM (2 249 608 B per utterance).
#include "plan.h"
#include "prof.h"
#include "fft200.h"

namespace se {

constexpr int kIFR = 32;                 // frames transformed per workgroup
constexpr int kIHB = 29;                 // hop-blocks of output per workgroup
constexpr int kISpan = kIHB * kHop;      // 4640 samples
constexpr int kIThreads = 256;

__global__ __launch_bounds__(kIThreads) void istft_kernel(
    const float* __restrict__ power, const float* __restrict__ phase, int F, float inv_lp,
    const float* __restrict__ window_inv, const float* __restrict__ window_sq,
    const float2* __restrict__ tw200g, const float2* __restrict__ tw400,
    float* __restrict__ wav, int wav_stride, const int64_t* __restrict__ lengths, float* __restrict__ sumsq) {
  __shared__ float2 Y[kIFR * kHalf];
  __shared__ float2 tw200[kHalf];
  __shared__ float red[kIThreads / 64];

  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int o0 = blockIdx.x * kISpan;                 // first output sample of this workgroup
  const int n_out = kHop * (F - 1);
  const int fbase = blockIdx.x * kIHB - 1;            // first frame overlapping the span (may be -1)
  const int flo = max(fbase, 0);
  const int fhi = min(fbase + kIFR, F);               // exclusive
  if (tid < kHalf) tw200[tid] = tw200g[tid];

  // ---- load + polar:  item (f, k), k = 0..200, contiguous in (B, F, K)
  {
    const size_t gbase = ((size_t)b * F + flo) * kBins;
    const int nitems = (fhi - flo) * kBins;
    for (int it = tid; it < nitems; it += kIThreads) {
      const int fl = it / kBins, k = it - fl * kBins;
      const float p = power[gbase + it];
      const float ph = phase[gbase + it];
      const float mag = (inv_lp == 0.5f) ? sqrtf(p) : powf(p, inv_lp);
      float s, c;
      sincosf(ph, &s, &c);
      float2* Z = Y + (flo + fl - fbase) * kHalf;
      if (k == 0) Z[0].x = mag * c;
      else if (k == kHalf) Z[0].y = mag * c;
      else Z[k] = make_float2(mag * c, mag * s);
    }
  }
  __syncthreads();

  // ---- fold pairs (k, 200-k): Z[k] = E + iO, Z[200-k] = conj(E) + i conj(O)
  for (int it = tid; it < (fhi - flo) * 101; it += kIThreads) {
    const int fl = it / 101, k = it - fl * 101;
    float2* Z = Y + (flo + fl - fbase) * kHalf;
    if (k == 0) {
      const float a = Z[0].x, c = Z[0].y;
      Z[0] = make_float2(0.5f * (a + c), 0.5f * (a - c));
    } else if (k == 100) {
      const float2 v = Z[100];
      Z[100] = make_float2(v.x, -v.y);
    } else {
      const float2 xk = Z[k], xn = Z[kHalf - k];
      const float2 E = make_float2(0.5f * (xk.x + xn.x), 0.5f * (xk.y - xn.y));
      const float2 D = make_float2(0.5f * (xk.x - xn.x), 0.5f * (xk.y + xn.y));
      const float2 w = tw400[k];                                   // W^-k = (cos, +sin)
      const float2 O = make_float2(D.x * w.x - D.y * w.y, D.x * w.y + D.y * w.x);
      Z[k] = make_float2(E.x - O.y, E.y + O.x);                    // E + iO
      Z[kHalf - k] = make_float2(E.x + O.y, -E.y + O.x);           // conj(E) + i conj(O)
    }
  }
  __syncthreads();

  for (int it = tid; it < (fhi - flo) * 25; it += kIThreads) {
    const int fl = it / 25, j = it - fl * 25;
    fft200_pass_a<+1>(Y + (flo + fl - fbase) * kHalf, j, tw200);
  }
  __syncthreads();

  {
    const int f = tid >> 3, q = tid & 7;
    const bool active = (fbase + f >= flo) && (fbase + f < fhi);
    float2 y[25];
    if (active) {
#pragma unroll
      for (int j = 0; j < 25; ++j) y[j] = Y[f * kHalf + 25 * q + j];
      fft25<+1>(y);
    }
    __syncthreads();
    if (active) {
      const float2* w2 = reinterpret_cast<const float2*>(window_inv);
#pragma unroll
      for (int c = 0; c < 5; ++c)
#pragma unroll
        for (int d = 0; d < 5; ++d) {
          const int n = q + 8 * (c + 5 * d);
          const float2 w = w2[n];
          Y[f * kHalf + n] = make_float2(y[5 * c + d].x * w.x, y[5 * c + d].y * w.y);
        }
    }
  }
  __syncthreads();

  // ---- overlap-add + envelope + masked square sum
  const float* xs = reinterpret_cast<const float*>(Y);
  const int len_b = lengths ? (int)min((int64_t)n_out, lengths[b]) : 0;
  float ss = 0.f;
  for (int o = tid; o < kISpan; o += kIThreads) {
    const int n = o0 + o;
    if (n >= n_out) break;
    const int p = n + kHalf;                       // padded index
    const int f_last = min(p / kHop, F - 1);
    float acc = 0.f, env = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int f = f_last - t;
      const int r = p - f * kHop;
      if (f >= 0 && r < kNfft) {
        acc += xs[(f - fbase) * kNfft + r];
        env += window_sq[r];
      }
    }
    const float v = acc / env;
    wav[(size_t)b * wav_stride + n] = v;
    if (n < len_b) ss = fmaf(v, v, ss);
  }
  // right-pad region [n_out, wav_stride) -- zero-filled by the last workgroup of the row
  if (blockIdx.x == gridDim.x - 1)
    for (int n = n_out + tid; n < wav_stride; n += kIThreads) wav[(size_t)b * wav_stride + n] = 0.f;

  if (sumsq) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    if (tid == 0) atomicAdd(&sumsq[b], red[0] + red[1] + red[2] + red[3]);
  }
}

__global__ __launch_bounds__(256) void masked_sumsq_kernel(const float* __restrict__ x, int T, int x_stride,
                                                           const int64_t* __restrict__ lengths, float* __restrict__ sums) {
  __shared__ float red[4];
  const int b = blockIdx.y;
  const int len = (int)min((int64_t)T, lengths[b]);
  const float* row = x + (size_t)b * x_stride;
  float ss = 0.f;
  for (int n = blockIdx.x * 256 + threadIdx.x; n < len; n += gridDim.x * 256) {
    const float v = row[n];
    ss = fmaf(v, v, ss);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&sums[b], red[0] + red[1] + red[2] + red[3]);
}

__global__ __launch_bounds__(256) void dbnorm_kernel(float* __restrict__ wav, int T, int wav_stride,
                                                     const int64_t* __restrict__ lengths, const float* __restrict__ wav_sumsq,
                                                     const float* __restrict__ ref_sumsq, float fixed_db, float eps) {
  const int b = blockIdx.y;
  // masked_mean = sum / (count + eps)   (utils.py:28); count = min(len, T) ones in the length mask
  const float cnt = (float)min((int64_t)T, lengths[b]);
  const float denom = cnt + eps;
  const float target_db = ref_sumsq ? 10.0f * log10f(ref_sumsq[b] / denom) : fixed_db;
  const float scale = sqrtf(powf(10.0f, target_db / 10.0f) / (wav_sumsq[b] / denom + eps));
  float* row = wav + (size_t)b * wav_stride;
  for (int n = blockIdx.x * 256 + threadIdx.x; n < T; n += gridDim.x * 256) row[n] *= scale;
}

__global__ void length_masks_kernel(const int64_t* __restrict__ lengths, int max_len, int64_t* __restrict__ masks) {
  const int b = blockIdx.y;
  const int64_t len = lengths[b];
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < max_len; t += gridDim.x * blockDim.x)
    masks[(size_t)b * max_len + t] = (t < len) ? 1 : 0;
}

}  // namespace se

extern "C" int se_istft_f32(const se_plan* plan, const float* power, const float* phase, int B, int F,
                            float linear_power, float* wav_out, int wav_stride,
                            const int64_t* lengths, float* sumsq_out, void* stream) {
  SE_REQUIRE(plan && power && phase && wav_out, "se_istft_f32: null argument");
  SE_REQUIRE(B > 0 && B <= 65535 && F >= 2, "se_istft_f32: bad B=%d F=%d", B, F);
  const int n_out = se::kHop * (F - 1);
  SE_REQUIRE(wav_stride >= n_out, "se_istft_f32: wav_stride=%d < %d output samples", wav_stride, n_out);
  SE_REQUIRE(linear_power > 0.f, "se_istft_f32: linear_power must be positive");
  SE_REQUIRE(sumsq_out == nullptr || lengths != nullptr, "se_istft_f32: sumsq_out needs lengths");
  hipStream_t st = se::as_stream(stream);
  if (sumsq_out) SE_HIP(hipMemsetAsync(sumsq_out, 0, sizeof(float) * B, st));
  dim3 grid((n_out + se::kISpan - 1) / se::kISpan, B);
  se::ProfScope prof(se::kProfIstft, (double)B * (8.0 * F * se::kBins + 4.0 * n_out), st);
  hipLaunchKernelGGL(se::istft_kernel, grid, dim3(se::kIThreads), 0, st, power, phase, F, 1.0f / linear_power,
                     plan->d_window_inv, plan->d_window_sq, plan->d_tw200, plan->d_tw400, wav_out, wav_stride, lengths, sumsq_out);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_masked_sumsq_f32(const float* x, int B, int T, int x_stride, const int64_t* lengths, float* sums, void* stream) {
  SE_REQUIRE(x && lengths && sums && B > 0 && B <= 65535 && T > 0 && x_stride >= T, "se_masked_sumsq_f32: bad argument");
  hipStream_t st = se::as_stream(stream);
  SE_HIP(hipMemsetAsync(sums, 0, sizeof(float) * B, st));
  dim3 grid(std::min(64, (T + 255) / 256), B);
  hipLaunchKernelGGL(se::masked_sumsq_kernel, grid, dim3(256), 0, st, x, T, x_stride, lengths, sums);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_dbnorm_f32(float* wav, int B, int T, int wav_stride, const int64_t* lengths,
                             const float* wav_sumsq, const float* ref_sumsq, float fixed_db, float eps, void* stream) {
  SE_REQUIRE(wav && lengths && wav_sumsq && B > 0 && B <= 65535 && T > 0 && wav_stride >= T, "se_dbnorm_f32: bad argument");
  dim3 grid(std::min(64, (T + 255) / 256), B);
  hipLaunchKernelGGL(se::dbnorm_kernel, grid, dim3(256), 0, se::as_stream(stream), wav, T, wav_stride, lengths, wav_sumsq,
                     ref_sumsq, fixed_db, eps);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_length_masks_i64(const int64_t* lengths, int B, int max_len, int64_t* masks, void* stream) {
  SE_REQUIRE(lengths && masks && B > 0 && B <= 65535 && max_len > 0, "se_length_masks_i64: bad argument");
  dim3 grid(std::min(256, (max_len + 255) / 256), B);
  hipLaunchKernelGGL(se::length_masks_kernel, grid, dim3(256), 0, se::as_stream(stream), lengths, max_len, masks);
  SE_LAUNCH_CHECK();
  return SE_OK;
}
