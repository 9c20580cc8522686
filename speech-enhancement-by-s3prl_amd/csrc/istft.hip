// istft.hip -- rows A6 + D1 + D2: power/phase -> waveform (inverse 400-point real FFT, window, overlap-add,
// envelope division, trim) with the masked square-sum for the dB normalisation fused in; plus the
// dB-normalise scale pass, the masked square-sum of a reference waveform and the length masks.
//
// iSTFT workgroup = 256 threads = 27 hop-blocks (4 320 output samples) of one utterance; it inverse-
// transforms the 30 frames that overlap that span (3 of them shared with the neighbours, re-read from L2):
//   polar + fold : one item = the bin PAIR (k, 200-k) of a frame, k = 0..100: X = sqrt(P) (cos phi, sin phi) for both bins (hardware
//                  sin / cos / sqrt), then Z[k] = E + i O, Z[200-k] = conj(E) + i conj(O) straight into LDS.  Im X[0] = Im X[200] = 0
//                  (c2r semantics).  The first version wrote X to LDS and folded in a second pass (a barrier, an LDS round trip, and a
//                  25-times unrolled loop with the k = 0 / k = 200 special cases as branches); it also carried both the sqrt and the
//                  general pow() of linear_power in every iteration -- ~6 000 vector instructions per thread, which at 3 workgroups per
//                  CU IS the kernel's time (4 cycles each x 3 waves per SIMD = 30 us per round of 768 workgroups, 1.6 rounds at B = 32).
//                  Now: the kernel is specialised on linear_power == 2 and the pair loop is branch-free.
//   pass A / pass B : fft200.h with DIR = +1; pass B multiplies by window / 400 (1/200 of the inverse transform, 1/2 of the fold)
//   ola    : out[n] = sum_f w[n - 160 f] z_f[n - 160 f] / sum_f w^2[n - 160 f]
// All tables (twiddles, window) live in LDS and every global load of a thread is issued up front, so the output loop
// holds stores only (a load there would drain the stores every iteration: one in-order vmcnt on gfx950).
// LDS 52.8 KB -> 3 workgroups per CU.  Bound: vector issue (the HBM floor, 2 249 608 B per utterance, is ~9 us at B = 32).
#include <stdlib.h>
#include "plan.h"
#include "prof.h"
#include "fft200.h"

namespace se {

#ifndef SE_ISTFT_FR
#define SE_ISTFT_FR 30
#endif
constexpr int kIFR = SE_ISTFT_FR;        // frames transformed per workgroup (30 x 25 = 750 pass-A items = 3 full rounds; 3 workgroups / CU)
constexpr int kIHB = kIFR - 3;           // hop-blocks of output per workgroup (three frames are shared with the neighbours)
constexpr int kISpan = kIHB * kHop;      // 4320 samples
constexpr int kIThreads = 256;
constexpr int kIPairs = 101;             // bin pairs (k, 200 - k) per frame
constexpr int kIPairIters = (kIFR * kIPairs + kIThreads - 1) / kIThreads;   // 12
constexpr int kIRefIters = (kISpan / 4 + kIThreads - 1) / kIThreads;         // 5 output quads per thread
constexpr float kIScale = 1.0f / 400.0f; // 1/200 (inverse transform) x 1/2 (E, O of the fold are kept doubled)

// ENC = 1: `phase` holds the encoded words of se_stft_tphase_f32 (stft.hip: encode_phase): (cos, sin) = (+-(1 - t^2), 2 t) / (1 + t^2)
template <int SQRT, int ENC>
__global__ __launch_bounds__(kIThreads) void istft_kernel(
    const float* __restrict__ power, const float* __restrict__ phase, int F, float inv_lp,
    const float* __restrict__ window, const float2* __restrict__ tw400g, const float2* __restrict__ tw200g,
    float* __restrict__ wav, int wav_stride, const int64_t* __restrict__ lengths, float* __restrict__ sumsq,
    const float* __restrict__ ref, int ref_stride, float* __restrict__ ref_sumsq) {
  __shared__ __attribute__((aligned(16))) float2 Y[kIFR * kHalf + 1];   // +1: the dump slot of the k = 0 pair's second write
  __shared__ float2 tw[kHalf];            // (cos, sin)(2 pi k / 400), k < 200: fold twiddles
  __shared__ float2 tw2[kHalf];           // (cos, sin)(2 pi t / 200): pass-A twiddles W200^(j q), j q <= 168
  __shared__ __attribute__((aligned(16))) float win[kNfft];             // window / 400
  __shared__ float red[2][kIThreads / 64];

  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int o0 = blockIdx.x * kISpan;                 // first output sample of this workgroup
  const int n_out = kHop * (F - 1);
  const int fbase = blockIdx.x * kIHB - 1;            // first frame overlapping the span (may be -1)
  const int flo = max(fbase, 0);
  const int fhi = min(fbase + kIFR, F);               // exclusive
  const int nfr = fhi - flo;

  float rq[kIRefIters][4];
  float rtail = 0.f;
#pragma unroll
  for (int i = 0; i < kIRefIters; ++i) rq[i][0] = rq[i][1] = rq[i][2] = rq[i][3] = 0.f;
  // ---- tables -> LDS first (their loads lead the queue), then every spectrum load of the thread back to back
  {
    const float2 twv = tw400g[min(tid, kHalf - 1)], tw2v = tw200g[min(tid, kHalf - 1)];
    const float wv0 = window[tid], wv1 = window[min(tid + kIThreads, kNfft - 1)];
    const size_t gbase = ((size_t)b * F + flo) * kBins;
    float p0[kIPairIters], h0[kIPairIters], p1[kIPairIters], h1[kIPairIters];
    // the reference waveform of the level normalisation (runner.py:570: wav_tar): its masked square sum rides along (D2), loaded up front
    // with everything else -- a load inside the output loop would drain that loop's stores
    if (ref) {
      const float* rrow = ref + (size_t)b * ref_stride;
      const int rlen = lengths ? (int)min((int64_t)n_out, lengths[b]) : 0;
#pragma unroll
      for (int i = 0; i < kIRefIters; ++i) {
        const int n = o0 + 4 * tid + 4 * kIThreads * i;
#pragma unroll
        for (int e = 0; e < 4; ++e) rq[i][e] = (4 * tid + 4 * kIThreads * i < kISpan && n + e < rlen) ? rrow[n + e] : 0.f;
      }
      // the mask of utils.py:26-46 runs to lengths[b], which may exceed the hop (F - 1) samples the inverse transform returns (T not a multiple of
      // the hop: the output row is zero-padded there, the reference is not): the row's last workgroup takes that tail (< 160 samples)
      if (blockIdx.x == gridDim.x - 1 && lengths) {
        const int rfull = (int)min((int64_t)min(wav_stride, ref_stride), lengths[b]);
        float t = 0.f;
        for (int n = n_out + tid; n < rfull; n += kIThreads) { const float v = rrow[n]; t = fmaf(v, v, t); }
        rtail = t;
      }
    }
#pragma unroll
    for (int r = 0; r < kIPairIters; ++r) {
      const int it = tid + kIThreads * r;
      const int fl = it / kIPairs, k = it - fl * kIPairs;
      const bool ok = fl < nfr;
      const size_t g = gbase + (size_t)(ok ? fl : 0) * kBins;
      p0[r] = power[g + k];
      h0[r] = phase[g + k];
      p1[r] = power[g + kHalf - k];
      h1[r] = phase[g + kHalf - k];
    }
    if (tid < kHalf) { tw[tid] = twv; tw2[tid] = tw2v; }
    win[tid] = wv0 * kIScale;
    if (tid + kIThreads < kNfft) win[tid + kIThreads] = wv1 * kIScale;
    __syncthreads();
    float2* Z0 = Y + (flo - fbase) * kHalf;
#pragma unroll
    for (int r = 0; r < kIPairIters; ++r) {
      const int it = tid + kIThreads * r;
      const int fl = it / kIPairs, k = it - fl * kIPairs;
      const float m0 = SQRT ? __builtin_amdgcn_sqrtf(p0[r]) : powf(p0[r], inv_lp);
      const float m1 = SQRT ? __builtin_amdgcn_sqrtf(p1[r]) : powf(p1[r], inv_lp);
      const bool k0 = (k == 0);
      float2 xk, xn;
      if (ENC) {
        // bit 0 of the word (the sign of cos) stays in t: 6e-8 relative
        const float t0 = h0[r], t1 = h1[r];
        const float s0 = t0 * t0, s1 = t1 * t1;
        const float g0 = m0 * __builtin_amdgcn_rcpf(1.0f + s0), g1 = m1 * __builtin_amdgcn_rcpf(1.0f + s1);
        xk = make_float2(__uint_as_float(__float_as_uint((1.0f - s0) * g0) ^ (__float_as_uint(t0) << 31)), k0 ? 0.f : (t0 + t0) * g0);
        xn = make_float2(__uint_as_float(__float_as_uint((1.0f - s1) * g1) ^ (__float_as_uint(t1) << 31)), k0 ? 0.f : (t1 + t1) * g1);
      } else {
        // sin / cos through the hardware units (argument in revolutions, reduced to [-0.5, 0.5])
        float r0 = h0[r] * 0.15915494309189535f, r1 = h1[r] * 0.15915494309189535f;
        r0 -= rintf(r0);
        r1 -= rintf(r1);
        xk = make_float2(m0 * __builtin_amdgcn_cosf(r0), k0 ? 0.f : m0 * __builtin_amdgcn_sinf(r0));
        xn = make_float2(m1 * __builtin_amdgcn_cosf(r1), k0 ? 0.f : m1 * __builtin_amdgcn_sinf(r1));
      }
      // doubled E, D (the 1/2 rides in the window scale); k = 100 pairs the bin with itself and both writes agree
      const float2 E = make_float2(xk.x + xn.x, xk.y - xn.y);
      const float2 D = make_float2(xk.x - xn.x, xk.y + xn.y);
      const float2 w = tw[k];                                       // W^-k = (cos, +sin)
      const float2 O = make_float2(D.x * w.x - D.y * w.y, D.x * w.y + D.y * w.x);
      if (fl < nfr) {
        float2* Z = Z0 + fl * kHalf;
        Z[k] = make_float2(E.x - O.y, E.y + O.x);                   // E + iO
        (k0 ? Y + kIFR * kHalf : Z + (kHalf - k))[0] = make_float2(E.x + O.y, O.x - E.y);   // conj(E) + i conj(O); k = 0 has no partner slot
      }
    }
  }
  __syncthreads();

  // ---- pass A (inverse)
  for (int it = tid; it < nfr * 25; it += kIThreads) {
    const int fl = it / 25, j = it - fl * 25;
    fft200_pass_a<+1>(Y + (flo + fl - fbase) * kHalf, j, tw2);
  }
  __syncthreads();

  {
    const int f = tid >> 3, q = tid & 7;
    const bool active = (f < kIFR) && (fbase + f >= flo) && (fbase + f < fhi);
    float2 y[25];
    if (active) {
#pragma unroll
      for (int j = 0; j < 25; ++j) y[j] = Y[f * kHalf + 25 * q + j];
      fft25<+1>(y);
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int c = 0; c < 5; ++c)
#pragma unroll
        for (int d = 0; d < 5; ++d) {
          const int n = q + 8 * (c + 5 * d);
          const float2 w = *reinterpret_cast<const float2*>(win + 2 * n);
          Y[f * kHalf + n] = make_float2(y[5 * c + d].x * w.x, y[5 * c + d].y * w.y);
        }
    }
  }
  __syncthreads();

  // ---- overlap-add + envelope + masked square sum; the loop contains LDS reads and global STORES only.  Four consecutive samples
  //      per thread: hop, n_fft/2 and the workgroup span are multiples of 4, so a quad shares its (up to three) frames, the LDS reads
  //      are aligned ds_read_b128 and the store is one 16-B store (scalar fallback when the output row is not 16-B aligned).
  //      The LDS window is w / 400: sum w^2 = 160 000 x sum win^2.
  const float* xs = reinterpret_cast<const float*>(Y);
  const int len_b = lengths ? (int)min((int64_t)n_out, lengths[b]) : 0;
  float ss = 0.f;
  float* wrow = wav + (size_t)b * wav_stride;
  const bool vec_out = ((reinterpret_cast<uintptr_t>(wrow) & 15) == 0);
  constexpr float kEnvScale = kIScale * kIScale;
#pragma unroll 1
  for (int o = 4 * tid; o < kISpan; o += 4 * kIThreads) {
    const int n = o0 + o;
    if (n >= n_out) break;                            // n_out = 160 (F - 1): a multiple of 4, a quad is valid or invalid as a whole
    const int p = n + kHalf;                          // padded index
    const int f_last = min(p / kHop, F - 1);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), env = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int f = f_last - t;
      const int rr = p - f * kHop;
      if (f >= 0 && rr < kNfft) {
        const float4 x4 = *reinterpret_cast<const float4*>(xs + (f - fbase) * kNfft + rr);
        const float4 w4 = *reinterpret_cast<const float4*>(win + rr);
        acc.x += x4.x; acc.y += x4.y; acc.z += x4.z; acc.w += x4.w;
        env.x = fmaf(w4.x, w4.x, env.x); env.y = fmaf(w4.y, w4.y, env.y);
        env.z = fmaf(w4.z, w4.z, env.z); env.w = fmaf(w4.w, w4.w, env.w);
      }
    }
    // v_rcp_f32 (1 ulp) instead of the IEEE division sequence: ~10 instructions per sample in a VALU-bound kernel
    const float4 v = make_float4(acc.x * (kEnvScale * __builtin_amdgcn_rcpf(env.x)), acc.y * (kEnvScale * __builtin_amdgcn_rcpf(env.y)),
                                 acc.z * (kEnvScale * __builtin_amdgcn_rcpf(env.z)), acc.w * (kEnvScale * __builtin_amdgcn_rcpf(env.w)));
    if (vec_out) {
      *reinterpret_cast<float4*>(wrow + n) = v;
    } else {
      wrow[n] = v.x; wrow[n + 1] = v.y; wrow[n + 2] = v.z; wrow[n + 3] = v.w;
    }
    if (n < len_b) ss = fmaf(v.x, v.x, ss);
    if (n + 1 < len_b) ss = fmaf(v.y, v.y, ss);
    if (n + 2 < len_b) ss = fmaf(v.z, v.z, ss);
    if (n + 3 < len_b) ss = fmaf(v.w, v.w, ss);
  }
  // right-pad region [n_out, wav_stride) -- zero-filled by the last workgroup of the row
  if (blockIdx.x == gridDim.x - 1)
    for (int n = n_out + tid; n < wav_stride; n += kIThreads) wav[(size_t)b * wav_stride + n] = 0.f;

  if (sumsq) {
    float sr = rtail;
    if (ref_sumsq) {
#pragma unroll
      for (int i = 0; i < kIRefIters; ++i) sr += rq[i][0] * rq[i][0] + rq[i][1] * rq[i][1] + rq[i][2] * rq[i][2] + rq[i][3] * rq[i][3];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      ss += __shfl_down(ss, off);
      sr += __shfl_down(sr, off);
    }
    if ((tid & 63) == 0) { red[0][tid >> 6] = ss; red[1][tid >> 6] = sr; }
    __syncthreads();
    if (tid == 0) atomicAdd(&sumsq[b], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    if (tid == 64 && ref_sumsq) atomicAdd(&ref_sumsq[b], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

__global__ __launch_bounds__(256) void masked_sumsq_kernel(const float* __restrict__ x, int T, int x_stride,
                                                           const int64_t* __restrict__ lengths, float* __restrict__ sums) {
  __shared__ float red[4];
  const int b = blockIdx.y;
  const int len = (int)min((int64_t)T, lengths[b]);
  const float* row = x + (size_t)b * x_stride;
  float ss = 0.f;
  if ((reinterpret_cast<uintptr_t>(row) & 15) == 0) {          // 16-B loads, two per thread in flight
    const int len4 = len >> 2, stride = gridDim.x * 256;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int n = blockIdx.x * 256 + threadIdx.x; n < len4; n += 2 * stride) {
      const float4 a = reinterpret_cast<const float4*>(row)[n];
      const float4 c = (n + stride < len4) ? reinterpret_cast<const float4*>(row)[n + stride] : make_float4(0.f, 0.f, 0.f, 0.f);
      acc.x = fmaf(a.x, a.x, fmaf(c.x, c.x, acc.x));
      acc.y = fmaf(a.y, a.y, fmaf(c.y, c.y, acc.y));
      acc.z = fmaf(a.z, a.z, fmaf(c.z, c.z, acc.z));
      acc.w = fmaf(a.w, a.w, fmaf(c.w, c.w, acc.w));
    }
    ss = (acc.x + acc.y) + (acc.z + acc.w);
    if (blockIdx.x == 0 && threadIdx.x < (len & 3)) {
      const float v = row[4 * len4 + threadIdx.x];
      ss = fmaf(v, v, ss);
    }
  } else {
    for (int n = blockIdx.x * 256 + threadIdx.x; n < len; n += gridDim.x * 256) {
      const float v = row[n];
      ss = fmaf(v, v, ss);
    }
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
  if (sumsq_out) { const int zrc_ = se::zero_async(sumsq_out, sizeof(float) * B, st); if (zrc_) return zrc_; }
  dim3 grid((n_out + se::kISpan - 1) / se::kISpan, B);
  se::ProfScope prof(se::kProfIstft, (double)B * (8.0 * F * se::kBins + 4.0 * n_out), st);
  if (linear_power == 2.0f)
    hipLaunchKernelGGL((se::istft_kernel<1, 0>), grid, dim3(se::kIThreads), 0, st, power, phase, F, 0.5f, plan->d_window, plan->d_tw400, plan->d_tw200,
                       wav_out, wav_stride, lengths, sumsq_out, nullptr, 0, nullptr);
  else
    hipLaunchKernelGGL((se::istft_kernel<0, 0>), grid, dim3(se::kIThreads), 0, st, power, phase, F, 1.0f / linear_power, plan->d_window, plan->d_tw400,
                       plan->d_tw200, wav_out, wav_stride, lengths, sumsq_out, nullptr, 0, nullptr);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

// the persistent kernel of istft2.hip: the path of log_input != 0 (log_predicted planes: exp in the load); as a replacement of the default kernel it
// measured equal or slower (SE_AMD_STFT2=1 in experiment builds)
extern "C" int se_istft2p_tphase_f32(const se_plan* plan, const float* power, const unsigned* tphase, int B, int F, int log_input,
                                     float* wav_out, int wav_stride, const int64_t* lengths, float* sumsq_out, void* stream);

extern "C" int se_istft_tphase_f32(const se_plan* plan, const float* power, const unsigned* tphase, int B, int F, int log_input,
                                   float* wav_out, int wav_stride, const int64_t* lengths, float* sumsq_out,
                                   const float* ref, int ref_stride, float* ref_sumsq_out, void* stream) {
  SE_REQUIRE(plan && power && tphase && wav_out, "se_istft_tphase_f32: null argument");
  SE_REQUIRE(B > 0 && B <= 65535 && F >= 2, "se_istft_tphase_f32: bad B=%d F=%d", B, F);
  const int n_out = se::kHop * (F - 1);
  SE_REQUIRE(wav_stride >= n_out, "se_istft_tphase_f32: wav_stride=%d < %d output samples", wav_stride, n_out);
  SE_REQUIRE(sumsq_out == nullptr || lengths != nullptr, "se_istft_tphase_f32: sumsq_out needs lengths");
  SE_REQUIRE(ref_sumsq_out == nullptr || (ref && sumsq_out && ref_stride >= n_out), "se_istft_tphase_f32: ref_sumsq_out needs ref (row stride >= %d) and sumsq_out", n_out);
#ifdef SE_AMD_EXPERIMENTS
  static const bool use2 = getenv("SE_AMD_STFT2") != nullptr;
#else
  const bool use2 = false;
#endif
  hipStream_t st = se::as_stream(stream);
  if (use2 || log_input) {
    int rc = se_istft2p_tphase_f32(plan, power, tphase, B, F, log_input, wav_out, wav_stride, lengths, sumsq_out, stream);
    // the reference's sum runs over [0, min(stride, length)) -- as in the default kernel and in masked_normalize_decibel (utils.py:31-46) -- not over
    // the n_out = hop (F - 1) samples the inverse transform produces: up to hop - 1 samples of the reference lie past that when T % hop != 0
    if (rc == SE_OK && ref_sumsq_out) rc = se_masked_sumsq_f32(ref, B, std::min(wav_stride, ref_stride), ref_stride, lengths, ref_sumsq_out, stream);
    return rc;
  }
  if (sumsq_out) {      // one clearing launch for both accumulators when they sit side by side
    int zrc_;
    if (ref_sumsq_out == sumsq_out + B) zrc_ = se::zero_async(sumsq_out, sizeof(float) * 2 * B, st);
    else {
      zrc_ = se::zero_async(sumsq_out, sizeof(float) * B, st);
      if (!zrc_ && ref_sumsq_out) zrc_ = se::zero_async(ref_sumsq_out, sizeof(float) * B, st);
    }
    if (zrc_) return zrc_;
  }
  dim3 grid((n_out + se::kISpan - 1) / se::kISpan, B);
  se::ProfScope prof(se::kProfIstft, (double)B * (8.0 * F * se::kBins + 4.0 * n_out), st);
  hipLaunchKernelGGL((se::istft_kernel<1, 1>), grid, dim3(se::kIThreads), 0, st, power, reinterpret_cast<const float*>(tphase), F, 0.5f, plan->d_window,
                     plan->d_tw400, plan->d_tw200, wav_out, wav_stride, lengths, sumsq_out, ref_sumsq_out ? ref : nullptr, ref_stride, ref_sumsq_out);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_masked_sumsq_f32(const float* x, int B, int T, int x_stride, const int64_t* lengths, float* sums, void* stream) {
  SE_REQUIRE(x && lengths && sums && B > 0 && B <= 65535 && T > 0 && x_stride >= T, "se_masked_sumsq_f32: bad argument");
  hipStream_t st = se::as_stream(stream);
  { const int zrc_ = se::zero_async(sums, sizeof(float) * B, st); if (zrc_) return zrc_; }
  dim3 grid(std::min(32, (T + 2047) / 2048), B);
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
