// stft2.hip -- rows A1+A2(+A3), the form every INTERNAL consumer uses: framing + Hann + 400-point real FFT -> power, UNIT PHASOR, raw mel.
//
// What changed against stft.hip (which stays for the boundary's on-demand outputs: `phase`, `complx`, the 128-filter MFCC bank):
//   * no atan2.  The only consumer of the noisy phase in every pipeline of the reference is the iSTFT (runner.py:267), which turns it
//     straight back into (cos, sin).  The kernel writes the phase as ONE 32-bit word per bin from which (cos, sin) follow rationally:
//         t = tan(phi' / 2) = y / (|X| + |x|)  in [-1, 1]   (phi' = the angle of (|x|, y)),  bit 0 of the fp32 pattern = (x < 0)
//         cos = +-(1 - t^2) / (1 + t^2),  sin = 2 t / (1 + t^2)
//     -- ~8 instructions per bin instead of ~22 per atan2, exact to ~1e-7, and the SAME 4 bytes per bin as a phase plane (a float2 unit
//     phasor was measured first: cheaper still in instructions, but +4 B / bin on both kernels costs more than it saves from B = 64 up).
//     `phase` = atan2 itself is an on-demand output of the Python boundary (preprocessor.LazyPhase).
//   * persistent workgroups with a register prefetch: a workgroup used to spend ~40 % of its life waiting for its first loads (HBM latency under
//     load, 3 workgroups per CU cannot cover it).  Now 3 x #CU workgroups walk the chunk list; the samples of chunk i+1 are loaded into
//     registers right after chunk i's fill consumed the previous set and are first touched (an opaque use, so that the compiler puts its
//     s_waitcnt there) after pass B -- two FFT passes later, with no store younger than them in the queue (gfx950 has ONE in-order vmcnt:
//     a wait placed after the write-out would also wait for every store of the chunk).
//   * 16-B fills (two packed complex samples per load / LDS write), the sparse mel table lives in registers for the workgroup's lifetime,
//     frames per chunk chosen per launch so that the chunk count fills whole rounds of the resident workgroups.
// Layout of a chunk in LDS and the FFT itself are those of stft.hip (fft200.h).  LDS 53 088 B -> 3 workgroups per CU.
#include <stdlib.h>
#include "plan.h"
#include "prof.h"
#include "fft200.h"

namespace se {

constexpr int kPFR = 30;                 // max frames per chunk
constexpr int kPThreads = 256;
constexpr int kPQuads = ((kPFR - 1) * kHop + kNfft) / 4;                 // 1260 sample quads cover a chunk's 30 overlapping frames
constexpr int kPFill = (kPQuads + kPThreads - 1) / kPThreads;           // 5 16-B loads per thread: every sample is loaded ONCE and fanned out to its <= 3 frames
constexpr int kPPost = (kPFR * 101 + kPThreads - 1) / kPThreads;      // 12
constexpr int kPPlane = 6036;            // floats per output plane in LDS: >= 3 (alignment shift) + 30 x 201, a multiple of 4

struct StftJob { float* power; unsigned* tphase; float* mel; int channel; int vec_ok; };

__device__ __forceinline__ int reflect2(int i, int T) {
  if (i < 0) i = -i;
  if (i >= T) i = 2 * (T - 1) - i;
  return i;
}

// phase of X = (x, y), q = x^2 + y^2, as one word: t = y / (|X| + |x|) (= tan of half the angle of (|x|, y), in [-1, 1]) with bit 0 = (x < 0).
// X == 0 -> t = 0, flag 0 -> (cos, sin) = (1, 0), the reference's atan2(0, 0) = 0.
__device__ __forceinline__ unsigned encode_phase(float x, float y, float q) {
  const float m = q * __builtin_amdgcn_rsqf(q);             // |X|  (NaN for q == 0: handled by the select below)
  const float d = m + fabsf(x);
  const float t = (q > 0.f) ? y * __builtin_amdgcn_rcpf(d) : 0.f;
  return (__float_as_uint(t) & ~1u) | (__float_as_uint(x) >> 31);
}

// NTB = mel-table floats per thread held in registers: 5 (n_mels <= 40) or 16 (n_mels <= 128)
template <int NTB>
__global__ __launch_bounds__(kPThreads, 3) void stftp_kernel(
    const float* __restrict__ wavs, int B, int C, int T, int F, int FR, int cpu /* chunks per utterance */, int total,
    const float* __restrict__ window, const float2* __restrict__ tw400g, const float2* __restrict__ tw200g,
    const int* __restrict__ mel_start, const int* __restrict__ mel_len, const float* __restrict__ mel_w, int n_mels,
    StftJob job0, StftJob job1, int dbg /* developer ablation mask (SE_AMD_STFT_ABLATE): 1 pass A, 2 pass B, 4 post, 8 stores, 16 loads, 32 fill, 64 mel */) {
  __shared__ __attribute__((aligned(16))) float2 Y[kPPlane];  // FFT buffer (30 x 200 complex), later the power plane + the mel table
  __shared__ float2 tw[kHalf];        // (cos, sin)(2 pi k / 400), k < 200: the recombination twiddles
  __shared__ float2 tw2[kHalf];       // (cos, sin)(2 pi t / 200): pass-A twiddles W200^(j q), j q <= 168
  __shared__ __attribute__((aligned(16))) float win[kNfft];

  const int tid = threadIdx.x;
  // ---- tables: LDS (twiddles, window) and registers (mel), once per workgroup
  {
    const float2 twv = tw400g[min(tid, kHalf - 1)], tw2v = tw200g[min(tid, kHalf - 1)];
    const float wv0 = window[tid], wv1 = window[min(tid + kPThreads, kNfft - 1)];
    if (tid < kHalf) { tw[tid] = twv; tw2[tid] = tw2v; }
    win[tid] = wv0;
    if (tid + kPThreads < kNfft) win[tid + kPThreads] = wv1;
  }
  float melw[NTB];
#pragma unroll
  for (int r = 0; r < NTB; ++r) {
    const int i = tid + kPThreads * r;
    melw[r] = (i < n_mels * kMelMaxW) ? mel_w[i] : 0.f;
  }
  const int mel_s = (tid < n_mels) ? mel_start[tid] : 0, mel_l = (tid < n_mels) ? mel_len[tid] : 0;

  float4 xv[kPFill];
  auto chunk_geom = [&](int c, int& jb, int& b, int& f0, int& nf) {
    jb = c / (B * cpu);
    const int r = c - jb * (B * cpu);
    b = r / cpu;
    f0 = (r - b * cpu) * FR;
    nf = min(FR, F - f0);
  };
  // all sample loads of one chunk, issued back to back: quad i = tid + 256 r holds samples [a, a + 4), a = f0 * 160 - 200 + 4 i
  auto load_chunk = [&](int c, int lt) {
    int jb, b, f0, nf;
    chunk_geom(c, jb, b, f0, nf);
    const int channel = jb ? job1.channel : job0.channel;
    const float* x = wavs + ((size_t)b * C + channel) * (size_t)T;
    const bool interior = (f0 * kHop - kHalf >= 0) && ((f0 + nf - 1) * kHop + kHalf <= T);
    const int L = (nf - 1) * kHop + kNfft;
#pragma unroll
    for (int r = 0; r < kPFill; ++r) {
      const int s = 4 * (lt + kPThreads * r);
      const int a = f0 * kHop - kHalf + s;
      xv[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (s < L) {
        if (interior) {
          xv[r] = *reinterpret_cast<const float4*>(x + a);
        } else {
          xv[r] = make_float4(x[reflect2(a, T)], x[reflect2(a + 1, T)], x[reflect2(a + 2, T)], x[reflect2(a + 3, T)]);
        }
      }
    }
  };

  int c = blockIdx.x;
  // developer knob: de-phase the workgroups that share a CU (dbg >> 8 = delay per group in units of 64 x 64 cycles)
  for (int i = ((int)blockIdx.x / 256 % 3) * (dbg >> 8); i > 0; --i) __builtin_amdgcn_s_sleep(64);
  if (c < total && !(dbg & 16)) load_chunk(c, tid);
  __syncthreads();                                          // tables visible
  for (; c < total;) {
    // lane-derived indices must not be hoisted out of the persistent loop (they would be kept live across it: ~120 spilled registers)
    int lt = tid;
    asm volatile("" : "+v"(lt));
    int jb, b, f0, nf;
    chunk_geom(c, jb, b, f0, nf);
    const StftJob job = jb ? job1 : job0;
    // ---- fill: every sample quad goes, windowed, to the <= 3 frames that contain it (hop 160, n_fft 400) as two packed complex values:
    //      one 16-B LDS window read + one 16-B LDS write per (quad, frame)
    {
      const int L = (dbg & 32) ? 0 : (nf - 1) * kHop + kNfft;
#pragma unroll
      for (int r = 0; r < kPFill; ++r) {
        const int s = 4 * (lt + kPThreads * r);
        if (s < L) {
          const int f_hi = min(s / kHop, nf - 1);
#pragma unroll
          for (int t = 0; t < 3; ++t) {
            const int f = f_hi - t, rr = s - f * kHop;
            if (f >= 0 && rr < kNfft) {
              const float4 w = *reinterpret_cast<const float4*>(win + rr);
              *reinterpret_cast<float4*>(Y + f * kHalf + (rr >> 1)) = make_float4(xv[r].x * w.x, xv[r].y * w.y, xv[r].z * w.z, xv[r].w * w.w);
            }
          }
        }
      }
    }
    const int cn = c + gridDim.x;
    if (cn < total && !(dbg & 16)) load_chunk(cn, lt);                         // in flight under pass A and pass B
    __syncthreads();

    // ---- pass A
    for (int it = lt; it < ((dbg & 1) ? 0 : nf * 25); it += kPThreads) {
      const int f = it / 25, j = it - f * 25;
      fft200_pass_a<-1>(Y + f * kHalf, j, tw2);
    }
    __syncthreads();

    // ---- pass B (8 items per frame)
    {
      const int f = lt >> 3, q = lt & 7;
      float2 y[25];
      const bool active = f < nf && !(dbg & 2);
      if (active) {
#pragma unroll
        for (int j = 0; j < 25; ++j) y[j] = Y[f * kHalf + 25 * q + j];
        fft25<-1>(y);
      }
      __syncthreads();
      if (active) {
#pragma unroll
        for (int cc = 0; cc < 5; ++cc)
#pragma unroll
          for (int d = 0; d < 5; ++d) Y[f * kHalf + q + 8 * (cc + 5 * d)] = y[5 * cc + d];
      }
    }
    __syncthreads();

    // ---- post: pairs (k, 200-k), k = 0..100 -> (power, encoded phase) of bins k and 200-k, kept in registers until every thread has read its
    //      inputs (the planes written next overlay OTHER frames' FFT outputs)
    const size_t obase = ((size_t)b * F + f0) * kBins;
    const int pad = (int)(obase & 3);             // LDS float index = pad + (output index - obase): same 16-B phase as the global span
    float p1[kPPost], p2[kPPost];
    unsigned h1[kPPost], h2[kPPost];
    {
      const bool want_ph = job.tphase != nullptr;
      int f = lt / 101, k = lt - f * 101;
#pragma unroll
      for (int r = 0; r < kPPost; ++r) {
        p1[r] = 0.f;
        p2[r] = 0.f;
        h1[r] = 0u;
        h2[r] = 0u;
        if (f < nf && !(dbg & 4)) {
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
          const float q1 = X1.x * X1.x + X1.y * X1.y, q2 = X2.x * X2.x + X2.y * X2.y;
          p1[r] = q1;
          p2[r] = q2;
          if (want_ph) {
            h1[r] = encode_phase(X1.x, X1.y, q1);
            h2[r] = encode_phase(X2.x, X2.y, q2);
          }
        }
        k += kPThreads - 2 * 101;                              // 256 = 2 * 101 + 54
        f += 2;
        if (k >= 101) { k -= 101; f += 1; }
      }
    }
    __syncthreads();
    float* Pw = reinterpret_cast<float*>(Y) + pad;            // power plane: Pw[f * 201 + k]
    unsigned* Ph = reinterpret_cast<unsigned*>(Pw + kPPlane); // encoded-phase plane
    {
      int f = lt / 101, k = lt - f * 101;
#pragma unroll
      for (int r = 0; r < kPPost; ++r) {
        if (f < nf) {
          const int o = f * kBins + k;
          Pw[o] = p1[r];
          Ph[o] = h1[r];
          if (k != 100) {                                     // k = 0 pairs with bin 200
            Pw[o + (kHalf - 2 * k)] = p2[r];
            Ph[o + (kHalf - 2 * k)] = h2[r];
          }
        }
        k += kPThreads - 2 * 101;
        f += 2;
        if (k >= 101) { k -= 101; f += 1; }
      }
    }
    __syncthreads();

    // the prefetched samples are first "used" here, as late as possible before the chunk's first store: the compiler's wait for them lands
    // where nothing younger is in the queue (the ablation: pass A + pass B alone, ~3 us, do not cover an HBM round trip under load)
#pragma unroll
    for (int r = 0; r < kPFill; ++r) asm volatile("" : "+v"(xv[r].x), "+v"(xv[r].y), "+v"(xv[r].z), "+v"(xv[r].w));
    // ---- write-out: the chunk's nf x 201 outputs are ONE contiguous span of each plane, and the LDS planes have the same layout
    //      and 16-B phase: scalar head / tail (rows of 201 words are not 16-B multiples), aligned 16-B body
    {
      float* __restrict__ power = (dbg & 8) ? nullptr : job.power;
      unsigned* __restrict__ tphase = (dbg & 8) ? nullptr : job.tphase;
      const int totalo = nf * kBins;
      const int head = job.vec_ok ? min(totalo, (4 - pad) & 3) : totalo;
      const int nvec = (totalo - head) >> 2;
      if (lt < head) {
        if (power) power[obase + lt] = Pw[lt];
        if (tphase) tphase[obase + lt] = Ph[lt];
      }
      if (!job.vec_ok) {
        for (int i = lt + kPThreads; i < totalo; i += kPThreads) {
          if (power) power[obase + i] = Pw[i];
          if (tphase) tphase[obase + i] = Ph[i];
        }
      } else {
        for (int i = head + 4 * nvec + lt; i < totalo; i += kPThreads) {
          if (power) power[obase + i] = Pw[i];
          if (tphase) tphase[obase + i] = Ph[i];
        }
        for (int v4 = lt; v4 < nvec; v4 += kPThreads) {
          const int i = head + 4 * v4;
          if (power) *reinterpret_cast<float4*>(power + obase + i) = *reinterpret_cast<const float4*>(Pw + i);
          if (tphase) *reinterpret_cast<uint4*>(tphase + obase + i) = *reinterpret_cast<const uint4*>(Ph + i);
        }
      }
    }
    // ---- mel: sparse HTK triangles over the power plane; the filter table (registers) is staged into the now dead phase plane
    if (job.mel && !(dbg & 64)) {
      __syncthreads();
      float* Tb = reinterpret_cast<float*>(Y) + kPPlane + 4;  // [n_mels * 32 weights][kMelMax starts][kMelMax lengths]
#pragma unroll
      for (int r = 0; r < NTB; ++r) Tb[lt + kPThreads * r] = melw[r];
      if (lt < kMelMax) {
        Tb[NTB * kPThreads + lt] = __int_as_float(mel_s);
        Tb[NTB * kPThreads + kMelMax + lt] = __int_as_float(mel_l);
      }
      __syncthreads();
      float* __restrict__ mel = job.mel;
      for (int it = lt; it < n_mels * 32; it += kPThreads) {
        const int m = it >> 5, fl = it & 31;                  // lane <-> frame: power reads at stride 201 floats (odd: conflict-free)
        if (fl >= nf) continue;
        const int st = __float_as_int(Tb[NTB * kPThreads + m]);
        const int len = __float_as_int(Tb[NTB * kPThreads + kMelMax + m]);
        const float* pr = Pw + fl * kBins + st;
        const float* wr_ = Tb + m * kMelMaxW;
        float acc = 0.f;
        for (int i = 0; i < len; ++i) acc = fmaf(wr_[i], (st + i < kHalf) ? pr[i] : 0.f, acc);
        mel[((size_t)b * n_mels + m) * F + f0 + fl] = acc;
      }
    }
    c = cn;
    __syncthreads();                                          // plane reads done before the next fill
  }
}

static int resident_workgroups(int per_cu) {
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

// frames per chunk: the FR in [24, 30] that minimises (rounds of the resident workgroups) x (FR + a fixed per-chunk share)
static int pick_frames_per_chunk(int F, long long utt_jobs, int slots) {
  int best = se::kPFR;
  double best_cost = 1e30;
  for (int fr = se::kPFR; fr >= 24; --fr) {
    const long long chunks = utt_jobs * ((F + fr - 1) / fr);
    const long long rounds = (chunks + slots - 1) / slots;
    const double cost = (double)rounds * (fr + 4);
    if (cost < best_cost - 1e-9) { best_cost = cost; best = fr; }
  }
  return best;
}

extern "C" int se_stft2p_tphase_f32(const se_plan* plan, const float* wavs, int B, int C, int T, int channel_a, float* power_a, unsigned* tphase_a, float* mel_a,
                                  int channel_b, float* power_b, unsigned* tphase_b, float* mel_b, void* stream) {
  SE_REQUIRE(plan && wavs, "se_stft2p_tphase_f32: null plan / wavs");
  const int njobs = channel_b >= 0 ? 2 : 1;
  SE_REQUIRE(B > 0 && C > 0 && channel_a >= 0 && channel_a < C && channel_b < C, "se_stft2p_tphase_f32: bad B=%d C=%d channels=%d,%d", B, C, channel_a, channel_b);
  SE_REQUIRE(T > se::kHalf, "se_stft2p_tphase_f32: T=%d must exceed n_fft/2=%d (reflect padding)", T, se::kHalf);
  const int F = T / se::kHop + 1;
  const int slots = se::resident_workgroups(3);
  const int FR = pick_frames_per_chunk(F, (long long)B * njobs, slots);
  const int cpu = (F + FR - 1) / FR;
  const long long total = (long long)njobs * B * cpu;
  SE_REQUIRE(total < (1ll << 30), "se_stft2p_tphase_f32: too many chunks");
  const se::StftJob j0{power_a, tphase_a, mel_a, channel_a, (((uintptr_t)power_a | (uintptr_t)tphase_a) % 16) == 0};
  const se::StftJob j1{power_b, tphase_b, mel_b, njobs > 1 ? channel_b : channel_a, (((uintptr_t)power_b | (uintptr_t)tphase_b) % 16) == 0};
  double bytes = 0.0;      // algorithmic bytes: 4 T in + 4 F K per written plane
  bytes += (double)B * (4.0 * T + 4.0 * F * se::kBins * ((power_a != nullptr) + (tphase_a != nullptr)) + (mel_a ? 4.0 * F * plan->geom.n_mels : 0.0));
  if (njobs > 1)
    bytes += (double)B * (4.0 * T + 4.0 * F * se::kBins * ((power_b != nullptr) + (tphase_b != nullptr)) + (mel_b ? 4.0 * F * plan->geom.n_mels : 0.0));
  hipStream_t st = se::as_stream(stream);
  se::ProfScope prof(se::kProfStft, bytes, st);
  const dim3 grid((unsigned)std::min<long long>(total, slots));
  const char* ab = getenv("SE_AMD_STFT_ABLATE");
  const char* sg = getenv("SE_AMD_STFT_STAGGER");
  const int dbg = (ab ? atoi(ab) : 0) | ((sg ? atoi(sg) : 0) << 8);
  if (plan->geom.n_mels <= 40)
    hipLaunchKernelGGL((se::stftp_kernel<5>), grid, dim3(se::kPThreads), 0, st, wavs, B, C, T, F, FR, cpu, (int)total, plan->d_window, plan->d_tw400,
                       plan->d_tw200, plan->d_mel_start, plan->d_mel_len, plan->d_mel_w, plan->geom.n_mels, j0, j1, dbg);
  else
    hipLaunchKernelGGL((se::stftp_kernel<16>), grid, dim3(se::kPThreads), 0, st, wavs, B, C, T, F, FR, cpu, (int)total, plan->d_window, plan->d_tw400,
                       plan->d_tw200, plan->d_mel_start, plan->d_mel_len, plan->d_mel_w, plan->geom.n_mels, j0, j1, dbg);
  SE_LAUNCH_CHECK();
  return SE_OK;
}
