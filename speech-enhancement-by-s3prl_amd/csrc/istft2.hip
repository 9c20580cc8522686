// istft2.hip -- rows A6 + D2 (reduce), the form every INTERNAL consumer uses: enhanced power + the noisy channel's ENCODED phase -> waveform.
//
//   X'[k] = sqrt(predicted[k]) * (cos, sin),  (cos, sin) = (+-(1 - t^2), 2 t) / (1 + t^2) from the word stft2.hip wrote (t = tan of the half
//   angle, bit 0 = cos < 0): one v_rcp_f32 instead of range reduction + v_sin_f32 + v_cos_f32, and no atan2 on the producing side
// then exactly istft.hip: fold to a 200-point complex spectrum, inverse FFT-200 (fft200.h, DIR = +1), window / 400, overlap-add of <= 3 frames
// per sample / sum w^2, masked sum of squares for the dB normalisation (one atomic per chunk).  c2r semantics (Im X[0], Im X[200] ignored).
// Persistent workgroups (3 per CU) walk the chunk list; the spectra of chunk i+1 are loaded into registers while chunk i is transformed and are
// first touched after pass B (see stft2.hip for why the wait has to sit there); hop-blocks per chunk are chosen per launch so that the
// chunk count fills whole rounds of the resident workgroups.  LDS 52.8 KB -> 3 workgroups per CU.
#include <stdlib.h>
#include "plan.h"
#include "prof.h"
#include "fft200.h"

namespace se {

constexpr int kQFR = 30;                 // max frames transformed per chunk
constexpr int kQThreads = 256;
constexpr int kQPairs = 101;             // bin pairs (k, 200 - k) per frame
constexpr int kQIters = (kQFR * kQPairs + kQThreads - 1) / kQThreads;   // 12
constexpr float kQScale = 1.0f / 400.0f; // 1/200 (inverse transform) x 1/2 (E, O of the fold are kept doubled)

// LOGIN = 1: `power` holds log_predicted of a log-target spec head (model.py:108-124): predicted = relu(exp(x)) = exp(x), taken here
template <int LOGIN>
__global__ __launch_bounds__(kQThreads, 3) void istftp_kernel(
    const float* __restrict__ power, const unsigned* __restrict__ tphase, int B, int F, int HB /* hop-blocks per chunk */, int cpu, int total,
    const float* __restrict__ window, const float2* __restrict__ tw400g, const float2* __restrict__ tw200g,
    float* __restrict__ wav, int wav_stride, const int64_t* __restrict__ lengths, float* __restrict__ sumsq) {
  __shared__ __attribute__((aligned(16))) float2 Y[kQFR * kHalf + 1];   // +1: the dump slot of the k = 0 pair's second write
  __shared__ float2 tw[kHalf];            // (cos, sin)(2 pi k / 400), k < 200: fold twiddles
  __shared__ float2 tw2[kHalf];           // (cos, sin)(2 pi t / 200): pass-A twiddles W200^(j q), j q <= 168
  __shared__ __attribute__((aligned(16))) float win[kNfft];             // window / 400
  __shared__ float red[kQThreads / 64];

  const int tid = threadIdx.x;
  const int n_out = kHop * (F - 1);
  {
    const float2 twv = tw400g[min(tid, kHalf - 1)], tw2v = tw200g[min(tid, kHalf - 1)];
    const float wv0 = window[tid], wv1 = window[min(tid + kQThreads, kNfft - 1)];
    if (tid < kHalf) { tw[tid] = twv; tw2[tid] = tw2v; }
    win[tid] = wv0 * kQScale;
    if (tid + kQThreads < kNfft) win[tid + kQThreads] = wv1 * kQScale;
  }

  float p0[kQIters], p1[kQIters];
  unsigned h0[kQIters], h1[kQIters];
  auto chunk_geom = [&](int c, int& b, int& blk, int& fbase, int& flo, int& nfr) {
    b = c / cpu;
    blk = c - b * cpu;
    fbase = blk * HB - 1;                                   // first frame overlapping the chunk's span (may be -1)
    flo = max(fbase, 0);
    nfr = min(fbase + HB + 3, F) - flo;
  };
  auto load_chunk = [&](int c, int lt) {
    int b, blk, fbase, flo, nfr;
    chunk_geom(c, b, blk, fbase, flo, nfr);
    const size_t gbase = ((size_t)b * F + flo) * kBins;
#pragma unroll
    for (int r = 0; r < kQIters; ++r) {
      const int it = lt + kQThreads * r;
      const int fl = it / kQPairs, k = it - fl * kQPairs;
      const size_t g = gbase + (size_t)(fl < nfr ? fl : 0) * kBins;
      p0[r] = power[g + k];
      p1[r] = power[g + kHalf - k];
      h0[r] = tphase[g + k];
      h1[r] = tphase[g + kHalf - k];
    }
  };

  int c = blockIdx.x;
  if (c < total) load_chunk(c, tid);
  __syncthreads();
  for (; c < total;) {
    // lane-derived indices must not be hoisted out of the persistent loop (they would be kept live across it and spilled)
    int lt = tid;
    asm volatile("" : "+v"(lt));
    int b, blk, fbase, flo, nfr;
    chunk_geom(c, b, blk, fbase, flo, nfr);
    const int fhi = flo + nfr;
    const int o0 = blk * HB * kHop;                         // first output sample of the chunk
    // ---- polar + fold, one item = the bin pair (k, 200 - k) of a frame
    {
      float2* Z0 = Y + (flo - fbase) * kHalf;
#pragma unroll
      for (int r = 0; r < kQIters; ++r) {
        const int it = lt + kQThreads * r;
        const int fl = it / kQPairs, k = it - fl * kQPairs;
        const float m0 = LOGIN ? __expf(0.5f * p0[r]) : __builtin_amdgcn_sqrtf(p0[r]);
        const float m1 = LOGIN ? __expf(0.5f * p1[r]) : __builtin_amdgcn_sqrtf(p1[r]);
        const bool k0 = (k == 0);
        // (cos, sin) = (+-(1 - t^2), 2 t) / (1 + t^2); bit 0 of the word (the sign of cos) stays in t: 6e-8 relative
        const float t0 = __uint_as_float(h0[r]), t1 = __uint_as_float(h1[r]);
        const float s0 = t0 * t0, s1 = t1 * t1;
        const float g0 = m0 * __builtin_amdgcn_rcpf(1.0f + s0), g1 = m1 * __builtin_amdgcn_rcpf(1.0f + s1);
        const float cx0 = __uint_as_float(__float_as_uint((1.0f - s0) * g0) ^ (h0[r] << 31));
        const float cx1 = __uint_as_float(__float_as_uint((1.0f - s1) * g1) ^ (h1[r] << 31));
        const float2 xk = make_float2(cx0, k0 ? 0.f : (t0 + t0) * g0);
        const float2 xn = make_float2(cx1, k0 ? 0.f : (t1 + t1) * g1);
        // doubled E, D (the 1/2 rides in the window scale); k = 100 pairs the bin with itself and both writes agree
        const float2 E = make_float2(xk.x + xn.x, xk.y - xn.y);
        const float2 D = make_float2(xk.x - xn.x, xk.y + xn.y);
        const float2 w = tw[k];                                       // W^-k = (cos, +sin)
        const float2 O = make_float2(D.x * w.x - D.y * w.y, D.x * w.y + D.y * w.x);
        if (fl < nfr) {
          float2* Z = Z0 + fl * kHalf;
          Z[k] = make_float2(E.x - O.y, E.y + O.x);                   // E + iO
          (k0 ? Y + kQFR * kHalf : Z + (kHalf - k))[0] = make_float2(E.x + O.y, O.x - E.y);   // conj(E) + i conj(O); k = 0 has no partner slot
        }
      }
    }
    const int cn = c + gridDim.x;
    if (cn < total) load_chunk(cn, lt);                     // in flight under pass A and pass B
    __syncthreads();

    // ---- pass A (inverse)
    for (int it = lt; it < nfr * 25; it += kQThreads) {
      const int fl = it / 25, j = it - fl * 25;
      fft200_pass_a<+1>(Y + (flo + fl - fbase) * kHalf, j, tw2);
    }
    __syncthreads();

    {
      const int f = lt >> 3, q = lt & 7;
      const bool active = (f < kQFR) && (fbase + f >= flo) && (fbase + f < fhi);
      float2 y[25];
      if (active) {
#pragma unroll
        for (int j = 0; j < 25; ++j) y[j] = Y[f * kHalf + 25 * q + j];
        fft25<+1>(y);
      }
      __syncthreads();
      if (active) {
#pragma unroll
        for (int cc = 0; cc < 5; ++cc)
#pragma unroll
          for (int d = 0; d < 5; ++d) {
            const int n = q + 8 * (cc + 5 * d);
            const float2 w = *reinterpret_cast<const float2*>(win + 2 * n);
            Y[f * kHalf + n] = make_float2(y[5 * cc + d].x * w.x, y[5 * cc + d].y * w.y);
          }
      }
    }
    // first "use" of the prefetched spectra: the compiler's wait lands here, with no store younger than the loads in the queue
#pragma unroll
    for (int r = 0; r < kQIters; ++r) asm volatile("" : "+v"(p0[r]), "+v"(p1[r]), "+v"(h0[r]), "+v"(h1[r]));
    __syncthreads();

    // ---- overlap-add + envelope + masked square sum: LDS reads and global STORES only, four consecutive samples per thread
    const float* xs = reinterpret_cast<const float*>(Y);
    const int len_b = lengths ? (int)min((int64_t)n_out, lengths[b]) : 0;
    float ss = 0.f;
    float* wrow = wav + (size_t)b * wav_stride;
    const bool vec_out = ((reinterpret_cast<uintptr_t>(wrow) & 15) == 0);
    constexpr float kEnvScale = kQScale * kQScale;
    const int span = HB * kHop;
#pragma unroll 1
    for (int o = 4 * lt; o < span; o += 4 * kQThreads) {
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
    // right-pad region [n_out, wav_stride) -- zero-filled by the last chunk of the row
    if (blk == cpu - 1)
      for (int n = n_out + lt; n < wav_stride; n += kQThreads) wrow[n] = 0.f;

    if (sumsq) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off);
      if ((lt & 63) == 0) red[lt >> 6] = ss;
    }
    c = cn;
    __syncthreads();                                      // OLA reads of Y done (and `red` complete) before the next chunk's fold
    if (sumsq && lt == 0) atomicAdd(&sumsq[b], red[0] + red[1] + red[2] + red[3]);
  }
}

int resident_workgroups_q(int per_cu) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
    else cus = 256;
  }
  if (const char* e = getenv("SE_AMD_STFT_SLOTS")) return std::max(1, atoi(e));      // developer A/B: e.g. 1000000 = one chunk per workgroup (non-persistent)
  return cus * per_cu;
}

}  // namespace se

// hop-blocks per chunk: the HB in [20, 27] that minimises (rounds of the resident workgroups) x (frames per chunk + a fixed share)
static int pick_hop_blocks(int n_blocks, long long B, int slots) {
  int best = 27;
  double best_cost = 1e30;
  for (int hb = 27; hb >= 20; --hb) {
    const long long chunks = B * ((n_blocks + hb - 1) / hb);
    const long long rounds = (chunks + slots - 1) / slots;
    const double cost = (double)rounds * (hb + 3 + 4);
    if (cost < best_cost - 1e-9) { best_cost = cost; best = hb; }
  }
  return best;
}

extern "C" int se_istft2p_tphase_f32(const se_plan* plan, const float* power, const unsigned* tphase, int B, int F, int log_input,
                                   float* wav_out, int wav_stride, const int64_t* lengths, float* sumsq_out, void* stream) {
  SE_REQUIRE(plan && power && tphase && wav_out, "se_istft2p_tphase_f32: null argument");
  SE_REQUIRE(B > 0 && F >= 2, "se_istft2p_tphase_f32: bad B=%d F=%d", B, F);
  const int n_out = se::kHop * (F - 1);
  SE_REQUIRE(wav_stride >= n_out, "se_istft2p_tphase_f32: wav_stride=%d < %d output samples", wav_stride, n_out);
  SE_REQUIRE(sumsq_out == nullptr || lengths != nullptr, "se_istft2p_tphase_f32: sumsq_out needs lengths");
  hipStream_t st = se::as_stream(stream);
  if (sumsq_out) { const int zrc_ = se::zero_async(sumsq_out, sizeof(float) * B, st); if (zrc_) return zrc_; }
  const int slots = se::resident_workgroups_q(3);
  const int HB = pick_hop_blocks(F - 1, B, slots);
  const int cpu = (F - 1 + HB - 1) / HB;
  const long long total = (long long)B * cpu;
  SE_REQUIRE(total < (1ll << 30), "se_istft2p_tphase_f32: too many chunks");
  // algorithmic bytes as SURVEY 8d states them (power + phase in, waveform out)
  se::ProfScope prof(se::kProfIstft, (double)B * (8.0 * F * se::kBins + 4.0 * n_out), st);
  const dim3 grid((unsigned)std::min<long long>(total, slots));
  if (log_input)
    hipLaunchKernelGGL(se::istftp_kernel<1>, grid, dim3(se::kQThreads), 0, st, power, tphase, B, F, HB, cpu, (int)total,
                       plan->d_window, plan->d_tw400, plan->d_tw200, wav_out, wav_stride, lengths, sumsq_out);
  else
    hipLaunchKernelGGL(se::istftp_kernel<0>, grid, dim3(se::kQThreads), 0, st, power, tphase, B, F, HB, cpu, (int)total,
                       plan->d_window, plan->d_tw400, plan->d_tw200, wav_out, wav_stride, lengths, sumsq_out);
  SE_LAUNCH_CHECK();
  return SE_OK;
}
