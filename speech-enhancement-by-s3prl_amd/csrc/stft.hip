// stft.hip -- rows A1+A2(+A3): framing + Hann + 400-point real FFT -> power / phase / complex / raw mel.
//
// One workgroup (256 threads) transforms FR = 30 consecutive frames of one (utterance, channel):
//   fill   : z[n] = (w[2n] x[2n], w[2n+1] x[2n+1]); all of a thread's global loads are issued back to back (one wait),
//            neighbouring frames overlap by 60 %: the re-reads are L1/L2 hits, HBM sees each sample once
//   pass A : 25 in-place radix-8 butterflies per frame          (fft200.h)
//   pass B : 8 in-register 25-point DFTs per frame
//   post   : X[k], X[200-k] from Z[k], Z[200-k]; (power, atan2 phase) held in registers across a barrier, then written into two
//            LDS PLANES laid out exactly like the output span (frame-major, 201 bins, shifted by the span's offset mod 4)
//   write  : the frames of a workgroup are ONE contiguous span of (B, F, K) in each plane: a linear LDS -> global copy with
//            aligned 16-B loads and stores (3 instructions per 4 outputs; the per-element de-interleave of the first version
//            was 16 % of the kernel's instructions).  No global load sits in any output loop (all tables live in LDS): a load there would force
//            vmcnt(0) per iteration, which on gfx950 also drains the previous iteration's stores (one in-order counter)
//            and serialises the loop on the HBM write latency.
//   mel    : sparse HTK triangles over the power kept in LDS; the filter table is staged into the unused .y halves
//            of the power slots; written feature-major (B, n_mels, F)
// LDS: planes 2 x 6036 floats (>= the 30 x 200 float2 FFT buffer) + two twiddle tables + window = 53 088 B -> 3 workgroups per CU.  Bound: HBM (2 249 608 B per
// utterance-channel).
#include <stdlib.h>
#include "plan.h"
#include "prof.h"
#include "fft200.h"

namespace se {

#ifndef SE_STFT_FR
#define SE_STFT_FR 20
#endif
// frames per workgroup.  Round 1-2: 30 (750 pass-A items = 3 full rounds of 256 threads; LDS 53 KB -> 3 workgroups per CU).  Round 3 sweep
// (tools/stft_fr.sh, profiles/r03_stft_fr.txt): the launch is latency-bound (every phase of a workgroup waits on the one before), so MORE, SMALLER
// workgroups per CU win although each is less efficient: 20 frames (500 pass-A items = 2 rounds; 4 workgroups per CU with the mel table's
// 4 360-float floor on the plane size) for launches that produce a mel plane, 10 frames without that floor (stft_small.hip: the same source
// compiled with SE_STFT_FR=10, SE_STFT_MELPLANE=0 -> 17.7 KB of LDS, 8 workgroups per CU) for launches that do not.
constexpr int kFR = SE_STFT_FR;
constexpr int kThreads = 256;
constexpr int kFillIters = (kFR * kHalf + kThreads - 1) / kThreads;     // 25
constexpr int kPostIters = (kFR * 101 + kThreads - 1) / kThreads;       // 13
#ifndef SE_STFT_MELPLANE
#define SE_STFT_MELPLANE 1
#endif
// floats per output plane in LDS: >= 3 (alignment shift) + kFR x 201, a multiple of 4; the mel stage parks its filter table in the phase plane
// (kMelMax x kMelMaxW weights + 2 kMelMax indices behind 4 floats of slack), which needs 4 360 floats whatever kFR
constexpr int kPlaneOut = (3 + kFR * 201 + 3) / 4 * 4;
constexpr int kPlane = (SE_STFT_MELPLANE && kPlaneOut < 4360) ? 4360 : kPlaneOut;     // floats per output plane in LDS: >= 3 (alignment shift) + 30 x 201, a multiple of 4

__device__ __forceinline__ int reflect(int i, int T) {
  // numpy / torch 'reflect' (no edge repeat); valid for |overshoot| < T
  if (i < 0) i = -i;
  if (i >= T) i = 2 * (T - 1) - i;
  return i;
}

// atan2 with the libm minimax polynomial for atan on [0, 1] but a one-instruction reciprocal range reduction and no
// inf / nan / denormal special cases (inputs are finite FFT outputs): ~20 instructions instead of ~40.
// |error| <= ~2e-7 rad.  (0, 0) -> 0 (libm returns +-pi for x = -0; the phase of a zero-magnitude bin is immaterial).
__device__ __forceinline__ float fast_atan2(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  float a = mn * __builtin_amdgcn_rcpf(mx);
  a = (mx == 0.f) ? 0.f : a;
  const float s = a * a;
  float p = fmaf(s, 0.00264226692f, -0.0152803790f);      // 0x3b2d2a58, 0xbc7a590c
  p = fmaf(s, p, 0.0414993092f);                          // 0x3d29fb3f
  p = fmaf(s, p, -0.0741364285f);                         // 0xbd97d4d7
  p = fmaf(s, p, 0.106052168f);                           // 0x3dd931b2
  p = fmaf(s, p, -0.141971514f);                          // 0xbe1160e6
  p = fmaf(s, p, 0.199923798f);                           // 0x3e4cb8bf
  p = fmaf(s, p, -0.333331168f);                          // 0xbeaaaa62
  float r = fmaf(a * s, p, a);
  r = (ay > ax) ? (1.57079632679f - r) : r;
  r = (x < 0.f) ? (3.14159265359f - r) : r;
  return copysignf(r, y);
}

// phase of X = (x, y), q = x^2 + y^2, as ONE word from which (cos, sin) follow rationally: t = y / (|X| + |x|) (= tan of half the angle of
// (|x|, y), in [-1, 1]) as fp32 with bit 0 = (x < 0):  cos = +-(1 - t^2) / (1 + t^2), sin = 2 t / (1 + t^2)  (istft.hip decodes it).
// ~8 instructions instead of ~22 for atan2, and no sin / cos on the consuming side.  X == 0 -> 0 -> (1, 0), the reference's atan2(0, 0) = 0.
__device__ __forceinline__ float encode_phase(float x, float y, float q) {
  const float m = q * __builtin_amdgcn_rsqf(q);             // |X|  (NaN for q == 0: handled by the select below)
  const float d = m + fabsf(x);
  const float t = (q > 0.f) ? y * __builtin_amdgcn_rcpf(d) : 0.f;
  return __uint_as_float((__float_as_uint(t) & ~1u) | (__float_as_uint(x) >> 31));
}

// blockIdx.z selects one of up to two (channel, output set) jobs of the same launch: the reference transforms the noisy AND the clean channel
// of every batch (runner.py:433,558); as two launches each was 1.4 rounds of the 768 resident workgroups at B = 32, together 2.8
// enc != 0: the `phase` plane receives the encoded phase words of se_stft_tphase_f32 (encode_phase) instead of atan2
struct StftOut { float* power; float* phase; float* complx; float* mel; int channel; int vec_ok; int enc; };

#ifdef SE_STFT_TU_SMALL
#define stft_kernel stft_small_kernel
#endif
__global__ __launch_bounds__(kThreads) void stft_kernel(
    const float* __restrict__ wavs, int C, int T, int F,
    const float* __restrict__ window, const float2* __restrict__ tw400g, const float2* __restrict__ tw200g,
    const int* __restrict__ mel_start, const int* __restrict__ mel_len, const float* __restrict__ mel_w, int n_mels,
    StftOut job0, StftOut job1, unsigned long long* __restrict__ dbgbuf) {
  const StftOut job = blockIdx.z ? job1 : job0;
  const int channel = job.channel, vec_ok = job.vec_ok;
  float* __restrict__ power = job.power;
  float* __restrict__ phase = job.phase;
  float* __restrict__ complx = job.complx;
  float* __restrict__ mel = job.mel;
  __shared__ __attribute__((aligned(16))) float2 Y[kPlane];   // FFT buffer (30 x 200 complex), later the two output planes
  __shared__ float2 tw[kHalf];        // (cos, sin)(2 pi k / 400), k < 200: the recombination twiddles
  __shared__ float2 tw2[kHalf];       // (cos, sin)(2 pi t / 200): pass-A twiddles W200^(j q), j q <= 168
  __shared__ float win[kNfft];

  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int f0 = blockIdx.x * kFR;
  const int nf = min(kFR, F - f0);
  // developer stamps (SE_AMD_STFT_DBG=1): thread 0 of workgroups (x = 5, y < 8) records s_memtime at phase boundaries
  const bool st_on = dbgbuf != nullptr && tid == 0 && blockIdx.x == 5 && b < 8;
  int st_i = 0;
#ifdef SE_AMD_STAMPS
#define SE_STAMP() do { if (st_on) dbgbuf[b * 16 + st_i++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SE_STAMP() do { (void)st_on; (void)st_i; } while (0)
#endif
  SE_STAMP();
  const float* x = wavs + ((size_t)b * C + channel) * (size_t)T;

  // table loads are issued together with the frame loads below and written to LDS afterwards (one memory latency)
  const float2 twv = tw400g[min(tid, kHalf - 1)], tw2v = tw200g[min(tid, kHalf - 1)];
  const float wv0 = window[tid], wv1 = window[min(tid + kThreads, kNfft - 1)];

  // ---- fill: windowed samples as packed complex.  item = tid + 256 r -> (frame, n) advanced incrementally.
  const bool interior = (f0 * kHop - kHalf >= 0) && ((f0 + nf - 1) * kHop + kHalf <= T);
  if (interior) {
    float2 xv[kFillIters];
    {
      int f = tid / kHalf, n = tid - f * kHalf;          // one division per thread
#pragma unroll
      for (int r = 0; r < kFillIters; ++r) {
        xv[r] = (f < nf) ? *reinterpret_cast<const float2*>(x + (f0 + f) * kHop - kHalf + 2 * n) : make_float2(0.f, 0.f);
        n += kThreads - kHalf;                            // 256 = 200 + 56
        f += 1;
        if (n >= kHalf) { n -= kHalf; f += 1; }
      }
    }
    if (tid < kHalf) { tw[tid] = twv; tw2[tid] = tw2v; }
    win[tid] = wv0;
    if (tid + kThreads < kNfft) win[tid + kThreads] = wv1;
    __syncthreads();                                      // window table visible
    {
      int f = tid / kHalf, n = tid - f * kHalf;
#pragma unroll
      for (int r = 0; r < kFillIters; ++r) {
        if (f < nf) {
          const float2 w = *reinterpret_cast<const float2*>(win + 2 * n);
          Y[f * kHalf + n] = make_float2(xv[r].x * w.x, xv[r].y * w.y);
        }
        n += kThreads - kHalf;
        f += 1;
        if (n >= kHalf) { n -= kHalf; f += 1; }
      }
    }
  } else {
    if (tid < kHalf) { tw[tid] = twv; tw2[tid] = tw2v; }
    win[tid] = wv0;
    if (tid + kThreads < kNfft) win[tid + kThreads] = wv1;
    __syncthreads();
    for (int it = tid; it < nf * kHalf; it += kThreads) {
      const int f = it / kHalf, n = it - f * kHalf;
      const int s = (f0 + f) * kHop - kHalf + 2 * n;
      Y[it] = make_float2(x[reflect(s, T)] * win[2 * n], x[reflect(s + 1, T)] * win[2 * n + 1]);
    }
  }
  __syncthreads();
  SE_STAMP();   // fill done

  // ---- pass A (twiddle W200^(j q) from the 400-table)
  for (int it = tid; it < nf * 25; it += kThreads) {
    const int f = it / 25, j = it - f * 25;
    float2* frame = Y + f * kHalf;
    float2 v[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = frame[25 * m + j];
    fft8<-1>(v);
#pragma unroll
    for (int q = 1; q < 8; ++q) {
      const float2 w = tw2[j * q];                         // j q <= 168
      v[q] = cmul(v[q], make_float2(w.x, -w.y));           // forward: exp(-i ...)
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) frame[25 * q + j] = v[q];
  }
  __syncthreads();
  SE_STAMP();   // pass A done

  // ---- pass B (8 items per frame)
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
  SE_STAMP();   // pass B done

  // ---- post: pairs (k, 200-k), k = 0..100 -> (power, phase) of bins k and 200-k, kept in registers until every thread has read its
  //      inputs (the planes written next overlay OTHER frames' FFT outputs)
  const size_t obase = ((size_t)b * F + f0) * kBins;
  const int pad = (int)(obase & 3);             // LDS float index = pad + (output index - obase): same 16-B phase as the global span
  float2 r1[kPostIters], r2[kPostIters];
  {
    int f = tid / 101, k = tid - f * 101;
#pragma unroll
    for (int r = 0; r < kPostIters; ++r) {
      r1[r] = make_float2(0.f, 0.f);
      r2[r] = make_float2(0.f, 0.f);
      if (f < nf) {
        const float2* Z = Y + f * kHalf;
        const float2 zk = Z[k];
        const float2 zn = Z[k == 0 ? 0 : kHalf - k];
        // E = (zk + conj(zn))/2 ; O = (zk - conj(zn))/(2i) ; P = W^k O, W^k = (c, -s)
        const float2 E = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));
        const float2 O = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
        const float2 w = tw[k];
        const float2 P = make_float2(O.x * w.x + O.y * w.y, O.y * w.x - O.x * w.y);
        const float2 X1 = make_float2(E.x + P.x, E.y + P.y);          // X[k]
        const float2 X2 = make_float2(E.x - P.x, -(E.y - P.y));       // X[200-k] = conj(E - P)
        if (complx) {       // rarely requested ('complx' features): direct 8-B stores (no loads in this loop)
          const size_t o1 = obase + (size_t)f * kBins + k;
          reinterpret_cast<float2*>(complx)[o1] = X1;
          if (k != 100) reinterpret_cast<float2*>(complx)[o1 + (kHalf - 2 * k)] = X2;
        }
        r1[r] = make_float2(X1.x * X1.x + X1.y * X1.y, 0.f);
        r2[r] = make_float2(X2.x * X2.x + X2.y * X2.y, 0.f);
        if (phase) {        // a job without a phase plane (the clean channel of the training batch: magnitude target only) skips 2 x ~22 instructions per pair
          if (job.enc) {
            r1[r].y = encode_phase(X1.x, X1.y, r1[r].x);
            r2[r].y = encode_phase(X2.x, X2.y, r2[r].x);
          } else {
            r1[r].y = fast_atan2(X1.y, X1.x);
            r2[r].y = fast_atan2(X2.y, X2.x);
          }
        }
      }
      k += kThreads - 2 * 101;                              // 256 = 2 * 101 + 54
      f += 2;
      if (k >= 101) { k -= 101; f += 1; }
    }
  }
  __syncthreads();
  float* Pw = reinterpret_cast<float*>(Y) + pad;            // power plane: Pw[f * 201 + k]
  float* Ph = Pw + kPlane;                                  // phase plane
  {
    int f = tid / 101, k = tid - f * 101;
#pragma unroll
    for (int r = 0; r < kPostIters; ++r) {
      if (f < nf) {
        const int o = f * kBins + k;
        Pw[o] = r1[r].x;
        Ph[o] = r1[r].y;
        if (k != 100) {                                     // k = 0 pairs with bin 200
          Pw[o + (kHalf - 2 * k)] = r2[r].x;
          Ph[o + (kHalf - 2 * k)] = r2[r].y;
        }
      }
      k += kThreads - 2 * 101;
      f += 2;
      if (k >= 101) { k -= 101; f += 1; }
    }
  }
  __syncthreads();
  SE_STAMP();   // post done

  // ---- write-out: the workgroup's nf x 201 outputs are ONE contiguous span of each plane, and the LDS planes have the same layout
  //      and 16-B phase: scalar head / tail (rows of 201 floats are not 16-B multiples), aligned float4 body
  if (power || phase) {
    const int total = nf * kBins;
    const int head = vec_ok ? min(total, (4 - pad) & 3) : total;
    const int nvec = (total - head) >> 2;
    if (tid < head) {
      if (power) power[obase + tid] = Pw[tid];
      if (phase) phase[obase + tid] = Ph[tid];
    }
    if (!vec_ok) {
      for (int i = tid + kThreads; i < total; i += kThreads) {
        if (power) power[obase + i] = Pw[i];
        if (phase) phase[obase + i] = Ph[i];
      }
    } else {
      for (int i = head + 4 * nvec + tid; i < total; i += kThreads) {
        if (power) power[obase + i] = Pw[i];
        if (phase) phase[obase + i] = Ph[i];
      }
      for (int v4 = tid; v4 < nvec; v4 += kThreads) {
        const int i = head + 4 * v4;
        if (power) *reinterpret_cast<float4*>(power + obase + i) = *reinterpret_cast<const float4*>(Pw + i);
        if (phase) *reinterpret_cast<float4*>(phase + obase + i) = *reinterpret_cast<const float4*>(Ph + i);
      }
    }
  }

  if (mel == nullptr) return;
  __syncthreads();
  SE_STAMP();
  // ---- mel: sparse HTK triangles over the power plane; the filter table is staged into the (now dead) phase plane
  // table = [n_mels x kMelMaxW weights][n_mels starts][n_mels lengths]: only the rows this bank has (40 in the reference's configs: 6 KiB; the
  // launcher checks that it fits behind the 4 floats of slack -- stft_small.hip's 10-frame planes take banks of up to 59 filters)
  float* Tb = reinterpret_cast<float*>(Y) + kPlane + 4;
  const int moff = n_mels * kMelMaxW;
  {
    constexpr int kTbIters = kMelMax * kMelMaxW / kThreads;      // 16 covers the largest bank (128 filters)
    float wv[kTbIters];
#pragma unroll
    for (int r = 0; r < kTbIters; ++r) {
      const int i = tid + kThreads * r;
      wv[r] = (i < moff) ? mel_w[i] : 0.f;
    }
    const int ms = (tid < n_mels) ? mel_start[tid] : 0, ml = (tid < n_mels) ? mel_len[tid] : 0;
#pragma unroll
    for (int r = 0; r < kTbIters; ++r) {
      const int i = tid + kThreads * r;
      if (i < moff) Tb[i] = wv[r];
    }
    if (tid < n_mels) {
      Tb[moff + tid] = __int_as_float(ms);
      Tb[moff + n_mels + tid] = __int_as_float(ml);
    }
  }
  __syncthreads();
  for (int it = tid; it < n_mels * 32; it += kThreads) {
    const int m = it >> 5, fl = it & 31;                    // lane <-> frame: power reads at stride 201 floats (odd: conflict-free)
    if (fl >= nf) continue;
    const int st = __float_as_int(Tb[moff + m]);
    const int len = __float_as_int(Tb[moff + n_mels + m]);
    const float* pr = Pw + fl * kBins + st;
    const float* wr_ = Tb + m * kMelMaxW;
    float acc = 0.f;
    for (int i = 0; i < len; ++i) acc = fmaf(wr_[i], (st + i < kHalf) ? pr[i] : 0.f, acc);
    mel[((size_t)b * n_mels + m) * F + f0 + fl] = acc;
  }
}

}  // namespace se

#ifdef SE_STFT_TU_SMALL
// the no-mel build of this file (stft_small.hip): only the kernel and this launcher
extern "C" int se_stft_small_plane_floats(void) { return se::kPlane; }
extern "C" int se_stft_launch_small(const se_plan* plan, const float* wavs, int B, int C, int T, const void* jobs_, int njobs, void* stream) {
  const se::StftOut* jobs = static_cast<const se::StftOut*>(jobs_);
  const int F = T / se::kHop + 1;
  dim3 grid((F + se::kFR - 1) / se::kFR, B, njobs);
  hipLaunchKernelGGL(se::stft_kernel, grid, dim3(se::kThreads), 0, se::as_stream(stream), wavs, C, T, F, plan->d_window, plan->d_tw400, plan->d_tw200,
                     plan->d_mel_start, plan->d_mel_len, plan->d_mel_w, plan->geom.n_mels, jobs[0], jobs[njobs > 1 ? 1 : 0], nullptr);
  SE_LAUNCH_CHECK();
  return SE_OK;
}
#else
extern "C" int se_stft_launch_small(const se_plan* plan, const float* wavs, int B, int C, int T, const void* jobs, int njobs, void* stream);
extern "C" int se_stft_small_plane_floats(void);

static int stft_launch(const se_plan* plan, const float* wavs, int B, int C, int T, const se::StftOut* jobs, int njobs, unsigned long long* dbgbuf, void* stream) {
  const int F = T / se::kHop + 1;
  dim3 grid((F + se::kFR - 1) / se::kFR, B, njobs);
  double bytes = 0.0;
  for (int j = 0; j < njobs; ++j)      // algorithmic bytes: 4 T in + 4 F K per written plane
    bytes += (double)B * (4.0 * T + 4.0 * F * se::kBins * ((jobs[j].power != nullptr) + (jobs[j].phase != nullptr) + 2 * (jobs[j].complx != nullptr)) +
                          (jobs[j].mel ? 4.0 * F * plan->geom.n_mels : 0.0));
  se::ProfScope prof(se::kProfStft, bytes, se::as_stream(stream));
  static const bool small_ok = getenv("SE_AMD_STFT_SMALL") == nullptr || atoi(getenv("SE_AMD_STFT_SMALL")) != 0;      // 0: always this file's kernel (A/B)
  bool any_mel = false;
  for (int j = 0; j < njobs; ++j) any_mel = any_mel || jobs[j].mel != nullptr;
  // 10-frame workgroups (stft_small.hip): planes of 2 016 floats; a mel bank fits its table (34 floats per filter + 4) there up to 59 filters
  const bool small_fits = !any_mel || plan->geom.n_mels * (se::kMelMaxW + 2) + 4 <= se_stft_small_plane_floats();
  if (small_ok && small_fits && !dbgbuf) return se_stft_launch_small(plan, wavs, B, C, T, jobs, njobs, stream);
  hipLaunchKernelGGL(se::stft_kernel, grid, dim3(se::kThreads), 0, se::as_stream(stream), wavs, C, T, F, plan->d_window, plan->d_tw400, plan->d_tw200,
                     plan->d_mel_start, plan->d_mel_len, plan->d_mel_w, plan->geom.n_mels, jobs[0], jobs[njobs > 1 ? 1 : 0], dbgbuf);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_stft_f32(const se_plan* plan, const float* wavs, int B, int C, int T, int channel,
                           float* power, float* phase, float* complx, float* mel, void* stream) {
  SE_REQUIRE(plan && wavs, "se_stft_f32: null plan / wavs");
  SE_REQUIRE(B > 0 && C > 0 && channel >= 0 && channel < C, "se_stft_f32: bad B=%d C=%d channel=%d", B, C, channel);
  SE_REQUIRE(T > se::kHalf, "se_stft_f32: T=%d must exceed n_fft/2=%d (reflect padding)", T, se::kHalf);
  SE_REQUIRE(B <= 65535, "se_stft_f32: B=%d exceeds grid.y limit", B);
  unsigned long long* dbgbuf = nullptr;
  if (getenv("SE_AMD_STFT_DBG") && complx) {      // developer stamps ride in the `complx` buffer
    dbgbuf = reinterpret_cast<unsigned long long*>(complx);
    complx = nullptr;
  }
  const se::StftOut job{power, phase, complx, mel, channel, (((uintptr_t)power | (uintptr_t)phase) % 16) == 0, 0};
  return stft_launch(plan, wavs, B, C, T, &job, 1, dbgbuf, stream);
}

extern "C" int se_stft2_f32(const se_plan* plan, const float* wavs, int B, int C, int T, int channel_a, float* power_a, float* phase_a, float* complx_a,
                            float* mel_a, int channel_b, float* power_b, float* phase_b, float* complx_b, float* mel_b, void* stream) {
  SE_REQUIRE(plan && wavs, "se_stft2_f32: null plan / wavs");
  SE_REQUIRE(B > 0 && C > 0 && channel_a >= 0 && channel_a < C && channel_b >= 0 && channel_b < C, "se_stft2_f32: bad B=%d C=%d channels=%d,%d", B, C, channel_a, channel_b);
  SE_REQUIRE(T > se::kHalf, "se_stft2_f32: T=%d must exceed n_fft/2=%d (reflect padding)", T, se::kHalf);
  SE_REQUIRE(B <= 65535, "se_stft2_f32: B=%d exceeds grid.y limit", B);
  const se::StftOut jobs[2] = {{power_a, phase_a, complx_a, mel_a, channel_a, (((uintptr_t)power_a | (uintptr_t)phase_a) % 16) == 0, 0},
                               {power_b, phase_b, complx_b, mel_b, channel_b, (((uintptr_t)power_b | (uintptr_t)phase_b) % 16) == 0, 0}};
  return stft_launch(plan, wavs, B, C, T, jobs, 2, nullptr, stream);
}

#ifdef SE_AMD_EXPERIMENTS
// the experimental persistent kernels of stft2.hip (SE_AMD_STFT2=1; measured equal or slower)
extern "C" int se_stft2p_tphase_f32(const se_plan* plan, const float* wavs, int B, int C, int T, int channel_a, float* power_a, unsigned* tphase_a, float* mel_a,
                                    int channel_b, float* power_b, unsigned* tphase_b, float* mel_b, void* stream);
#endif

extern "C" int se_stft_tphase_f32(const se_plan* plan, const float* wavs, int B, int C, int T, int channel_a, float* power_a, unsigned* tphase_a, float* mel_a,
                                  int channel_b, float* power_b, unsigned* tphase_b, float* mel_b, void* stream) {
  SE_REQUIRE(plan && wavs, "se_stft_tphase_f32: null plan / wavs");
  const int njobs = channel_b >= 0 ? 2 : 1;
  SE_REQUIRE(B > 0 && C > 0 && channel_a >= 0 && channel_a < C && channel_b < C, "se_stft_tphase_f32: bad B=%d C=%d channels=%d,%d", B, C, channel_a, channel_b);
  SE_REQUIRE(T > se::kHalf, "se_stft_tphase_f32: T=%d must exceed n_fft/2=%d (reflect padding)", T, se::kHalf);
  SE_REQUIRE(B <= 65535, "se_stft_tphase_f32: B=%d exceeds grid.y limit", B);
#ifdef SE_AMD_EXPERIMENTS
  static const bool use2 = getenv("SE_AMD_STFT2") != nullptr;
  if (use2) return se_stft2p_tphase_f32(plan, wavs, B, C, T, channel_a, power_a, tphase_a, mel_a, channel_b, power_b, tphase_b, mel_b, stream);
#endif
  float* pa = reinterpret_cast<float*>(tphase_a);
  float* pb = reinterpret_cast<float*>(tphase_b);
  const se::StftOut jobs[2] = {{power_a, pa, nullptr, mel_a, channel_a, (((uintptr_t)power_a | (uintptr_t)pa) % 16) == 0, 1},
                               {power_b, pb, nullptr, mel_b, njobs > 1 ? channel_b : channel_a, (((uintptr_t)power_b | (uintptr_t)pb) % 16) == 0, 1}};
  return stft_launch(plan, wavs, B, C, T, jobs, njobs, nullptr, stream);
}
#endif  // SE_STFT_TU_SMALL
