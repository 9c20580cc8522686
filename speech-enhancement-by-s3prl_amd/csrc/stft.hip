// stft.hip -- rows A1+A2(+A3): framing + Hann + 400-point real FFT -> power / phase / complex / raw mel.
//
// One workgroup (256 threads) transforms FR = 32 consecutive frames of one (utterance, channel):
//   fill   : z[n] = (w[2n] x[2n], w[2n+1] x[2n+1]) straight from global (reflect padding at the edges);
//            neighbouring frames overlap by 60 %, the re-reads are L1/L2 hits, HBM sees each sample once
//   pass A : 25 in-place radix-8 butterflies per frame          (fft200.h)
//   pass B : 8 in-register 25-point DFTs per frame
//   post   : X[k], X[200-k] from Z[k], Z[200-k]; power / phase written time-major -- the 32 frames of a
//            workgroup are ONE contiguous span of (B, F, K), so stores are perfectly coalesced
//   mel    : sparse HTK triangles over the power kept in LDS, written feature-major (B, n_mels, F)
// LDS: 32 x 200 float2 = 51 200 B  -> 3 workgroups per CU.  Bound: HBM (2 249 608 B per utterance-channel).
#include "plan.h"
#include "prof.h"
#include "fft200.h"

namespace se {

constexpr int kFR = 32;          // frames per workgroup
constexpr int kThreads = 256;

__device__ __forceinline__ int reflect(int i, int T) {
  // numpy / torch 'reflect' (no edge repeat); valid for |overshoot| < T
  if (i < 0) i = -i;
  if (i >= T) i = 2 * (T - 1) - i;
  return i;
}

__global__ __launch_bounds__(kThreads) void stft_kernel(
    const float* __restrict__ wavs, int C, int T, int channel, int F,
    const float* __restrict__ window, const float2* __restrict__ tw200g, const float2* __restrict__ tw400,
    const int* __restrict__ mel_start, const int* __restrict__ mel_len, const float* __restrict__ mel_w, int n_mels,
    float* __restrict__ power, float* __restrict__ phase, float* __restrict__ complx, float* __restrict__ mel) {
  __shared__ float2 Y[kFR * kHalf];
  __shared__ float2 tw200[kHalf];

  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int f0 = blockIdx.x * kFR;
  const int nf = min(kFR, F - f0);
  const float* x = wavs + ((size_t)b * C + channel) * (size_t)T;

  if (tid < kHalf) tw200[tid] = tw200g[tid];

  // ---- fill: windowed samples as packed complex
  const bool interior = (f0 * kHop - kHalf >= 0) && ((f0 + nf - 1) * kHop + kHalf <= T);
  const float2* win2 = reinterpret_cast<const float2*>(window);
  if (interior) {
    for (int it = tid; it < nf * kHalf; it += kThreads) {
      const int f = it / kHalf, n = it - f * kHalf;
      const float2 xv = *reinterpret_cast<const float2*>(x + (f0 + f) * kHop - kHalf + 2 * n);
      const float2 w = win2[n];
      Y[it] = make_float2(xv.x * w.x, xv.y * w.y);
    }
  } else {
    for (int it = tid; it < nf * kHalf; it += kThreads) {
      const int f = it / kHalf, n = it - f * kHalf;
      const int s = (f0 + f) * kHop - kHalf + 2 * n;
      const float2 w = win2[n];
      Y[it] = make_float2(x[reflect(s, T)] * w.x, x[reflect(s + 1, T)] * w.y);
    }
  }
  __syncthreads();

  // ---- pass A
  for (int it = tid; it < nf * 25; it += kThreads) {
    const int f = it / 25, j = it - f * 25;
    fft200_pass_a<-1>(Y + f * kHalf, j, tw200);
  }
  __syncthreads();

  // ---- pass B (8 items per frame; 32 frames x 8 = 256 threads)
  {
    const int f = tid >> 3, q = tid & 7;
    float2 y[25];
    const bool active = f < nf;
    if (active) {
#pragma unroll
      for (int j = 0; j < 25; ++j) y[j] = Y[f * kHalf + 25 * q + j];
      fft25<-1>(y);
    }
    __syncthreads();
    if (active) {
#pragma unroll
      for (int c = 0; c < 5; ++c)
#pragma unroll
        for (int d = 0; d < 5; ++d) Y[f * kHalf + q + 8 * (c + 5 * d)] = y[5 * c + d];
    }
  }
  __syncthreads();

  // ---- post: pairs (k, 200-k), k = 0..100
  const size_t obase = ((size_t)b * F + f0) * kBins;
  for (int it = tid; it < nf * 101; it += kThreads) {
    const int f = it / 101, k = it - f * 101;
    float2* Z = Y + f * kHalf;
    const float2 zk = Z[k];
    const float2 zn = Z[k == 0 ? 0 : kHalf - k];
    // E = (zk + conj(zn))/2 ; O = (zk - conj(zn))/(2i) ; P = W^k O, W^k = (c, -s)
    const float2 E = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
    const float2 O = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
    const float2 w = tw400[k];
    const float2 P = make_float2(O.x * w.x + O.y * w.y, O.y * w.x - O.x * w.y);
    const float2 X1 = make_float2(E.x + P.x, E.y + P.y);          // X[k]
    const float2 X2 = make_float2(E.x - P.x, -(E.y - P.y));       // X[200-k] = conj(E - P)
    const float p1 = X1.x * X1.x + X1.y * X1.y;
    const float p2 = X2.x * X2.x + X2.y * X2.y;
    const size_t o1 = obase + (size_t)f * kBins + k;
    const size_t o2 = obase + (size_t)f * kBins + (kHalf - k);
    if (power) {
      power[o1] = p1;
      if (k != 100) power[o2] = p2;
    }
    if (phase) {
      phase[o1] = atan2f(X1.y, X1.x);
      if (k != 100) phase[o2] = atan2f(X2.y, X2.x);
    }
    if (complx) {
      reinterpret_cast<float2*>(complx)[o1] = X1;
      if (k != 100) reinterpret_cast<float2*>(complx)[o2] = X2;
    }
    // power back into LDS for the mel stage: bin k -> Z[k].x ; bin 200 has zero weight in every HTK filter
    Z[k].x = p1;
    if (k != 0 && k != 100) Z[kHalf - k].x = p2;
  }

  if (mel == nullptr) return;
  __syncthreads();
  for (int it = tid; it < n_mels * kFR; it += kThreads) {
    const int m = it / kFR, fl = it - m * kFR;
    if (fl >= nf) continue;
    const int st = mel_start[m], len = mel_len[m];
    const float* wrow = mel_w + m * kMelMaxW;
    float acc = 0.f;
    for (int i = 0; i < len; ++i) {
      const int k = st + i;
      const float pk = (k < kHalf) ? Y[fl * kHalf + k].x : 0.f;
      acc = fmaf(wrow[i], pk, acc);
    }
    mel[((size_t)b * n_mels + m) * F + f0 + fl] = acc;
  }
}

}  // namespace se

extern "C" int se_stft_f32(const se_plan* plan, const float* wavs, int B, int C, int T, int channel,
                           float* power, float* phase, float* complx, float* mel, void* stream) {
  SE_REQUIRE(plan && wavs, "se_stft_f32: null plan / wavs");
  SE_REQUIRE(B > 0 && C > 0 && channel >= 0 && channel < C, "se_stft_f32: bad B=%d C=%d channel=%d", B, C, channel);
  SE_REQUIRE(T > se::kHalf, "se_stft_f32: T=%d must exceed n_fft/2=%d (reflect padding)", T, se::kHalf);
  SE_REQUIRE(B <= 65535, "se_stft_f32: B=%d exceeds grid.y limit", B);
  const int F = T / se::kHop + 1;
  dim3 grid((F + se::kFR - 1) / se::kFR, B);
  // algorithmic bytes: 4 T in + 4 F K per written plane
  se::ProfScope prof(se::kProfStft, (double)B * (4.0 * T + 4.0 * F * se::kBins * ((power != nullptr) + (phase != nullptr) + 2 * (complx != nullptr)) + (mel ? 4.0 * F * plan->geom.n_mels : 0.0)), se::as_stream(stream));
  hipLaunchKernelGGL(se::stft_kernel, grid, dim3(se::kThreads), 0, se::as_stream(stream), wavs, C, T, channel, F,
                     plan->d_window, plan->d_tw200, plan->d_tw400, plan->d_mel_start, plan->d_mel_len, plan->d_mel_w,
                     plan->geom.n_mels, power, phase, complx, mel);
  SE_LAUNCH_CHECK();
  return SE_OK;
}
