// plan.hip -- se_plan_create / destroy, error plumbing, device probing.  Host code only.
#include <math.h>
#include <string>
#include "plan.h"

namespace se {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  if (e == hipErrorOutOfMemory) {
    // keep the substring the reference's skip-batch handler matches (runner.py:505,606)
    set_error("CUDA out of memory (HIP: %s) in %s at %s:%d", hipGetErrorString(e), what, file, line);
    return SE_ERR_OOM;
  }
  set_error("HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
  (void)hipGetLastError();
  return SE_ERR_HIP;
}
}  // namespace se

extern "C" const char* se_last_error(void) { return se::g_err; }
extern "C" const char* se_version(void) { return "se_amd 0.1 (gfx950)"; }

extern "C" int se_device_available(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  if (n <= 0) return 0;
  hipDeviceProp_t p;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
  return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

// HTK mel bank of torchaudio-0.6 create_fb_matrix (f_min 0, f_max sr//2, no norm), float64 math.
static void build_mel(const se_geometry& g, std::vector<float>& fb) {
  const int K = g.n_freq, M = g.n_mels;
  fb.assign((size_t)K * M, 0.f);
  const double fmax = (double)(g.sample_rate / 2);
  const double m_min = 2595.0 * log10(1.0 + 0.0 / 700.0);
  const double m_max = 2595.0 * log10(1.0 + fmax / 700.0);
  std::vector<double> f_pts(M + 2);
  for (int i = 0; i < M + 2; ++i) {
    // torch.linspace(m_min, m_max, M+2) in float64: start + step*i for the first half, end - step*(n-1-i) after
    const int n = M + 2;
    const double step = (m_max - m_min) / (double)(n - 1);
    const double m = (i < n / 2) ? (m_min + step * i) : (m_max - step * (n - 1 - i));
    f_pts[i] = 700.0 * (pow(10.0, m / 2595.0) - 1.0);
  }
  for (int k = 0; k < K; ++k) {
    const int n = K;
    const double step = (fmax - 0.0) / (double)(n - 1);
    const double freq = (k < n / 2) ? (0.0 + step * k) : (fmax - step * (n - 1 - k));
    for (int m = 0; m < M; ++m) {
      const double down = -(f_pts[m] - freq) / (f_pts[m + 1] - f_pts[m]);
      const double up = (f_pts[m + 2] - freq) / (f_pts[m + 2] - f_pts[m + 1]);
      const double v = fmax > 0 ? fmin(down, up) : 0.0;
      fb[(size_t)k * M + m] = (float)(v > 0.0 ? v : 0.0);
    }
  }
}

extern "C" int se_plan_create(const se_geometry* geom, se_plan** out) {
  SE_REQUIRE(geom && out, "se_plan_create: null argument");
  if (geom->n_freq != se::kBins || geom->hop != se::kHop || geom->win > se::kNfft || geom->win < 2 ||
      geom->n_mels < 1 || geom->n_mels > se::kMelMax) {
    se::set_error("se_plan_create: unsupported geometry (n_freq=%d hop=%d win=%d n_mels=%d); the gfx950 kernels are "
                  "specialised for n_freq=201, hop=160, win<=400, n_mels<=128", geom->n_freq, geom->hop, geom->win, geom->n_mels);
    return SE_ERR_UNSUPPORTED;
  }
  se_plan* p = new se_plan();
  p->geom = *geom;
  p->d_blob = nullptr;
  // window: periodic Hann of length win, centred in n_fft (torch.stft pads win_length < n_fft on both sides)
  p->h_window.assign(se::kNfft, 0.f);
  const int left = (se::kNfft - geom->win) / 2;
  for (int n = 0; n < geom->win; ++n)
    p->h_window[left + n] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * (double)n / (double)geom->win));
  build_mel(*geom, p->h_melfb);

  std::vector<float> winv(se::kNfft), wsq(se::kNfft);
  for (int n = 0; n < se::kNfft; ++n) {
    winv[n] = p->h_window[n] / (float)se::kHalf;
    wsq[n] = p->h_window[n] * p->h_window[n];
  }
  std::vector<float2> tw200(se::kHalf), tw400(se::kHalf);
  for (int t = 0; t < se::kHalf; ++t) tw200[t] = make_float2((float)cos(2.0 * M_PI * t / 200.0), (float)sin(2.0 * M_PI * t / 200.0));
  for (int k = 0; k < se::kHalf; ++k) tw400[k] = make_float2((float)cos(2.0 * M_PI * k / 400.0), (float)sin(2.0 * M_PI * k / 400.0));
  std::vector<int> mstart(se::kMelMax, 0), mlen(se::kMelMax, 0);
  std::vector<float> mw((size_t)se::kMelMax * se::kMelMaxW, 0.f);
  for (int m = 0; m < geom->n_mels; ++m) {
    int lo = -1, hi = -1;
    for (int k = 0; k < se::kBins; ++k)
      if (p->h_melfb[(size_t)k * geom->n_mels + m] != 0.f) {
        if (lo < 0) lo = k;
        hi = k;
      }
    if (lo < 0) continue;
    if (hi - lo + 1 > se::kMelMaxW) {
      delete p;
      se::set_error("se_plan_create: mel filter %d spans %d bins (> %d)", m, hi - lo + 1, se::kMelMaxW);
      return SE_ERR_UNSUPPORTED;
    }
    mstart[m] = lo;
    mlen[m] = hi - lo + 1;
    for (int k = lo; k <= hi; ++k) mw[(size_t)m * se::kMelMaxW + (k - lo)] = p->h_melfb[(size_t)k * geom->n_mels + m];
  }

  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    delete p;
    se::set_error("se_plan_create: no HIP device visible (the product path has no CPU fallback)");
    return SE_ERR_NO_DEVICE;
  }
  SE_HIP(hipGetDevice(&p->device));
  // one blob, 256-B aligned sections
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  size_t o_win = 0, o_winv = o_win + al(400 * 4), o_wsq = o_winv + al(400 * 4), o_t200 = o_wsq + al(400 * 4),
         o_t400 = o_t200 + al(200 * 8), o_ms = o_t400 + al(200 * 8), o_ml = o_ms + al(se::kMelMax * 4),
         o_mw = o_ml + al(se::kMelMax * 4), total = o_mw + al(mw.size() * 4);
  e = hipMalloc(&p->d_blob, total);
  if (e != hipSuccess) {
    delete p;
    return se::hip_fail(e, "hipMalloc(plan tables)", __FILE__, __LINE__);
  }
  char* base = (char*)p->d_blob;
  p->d_window = (float*)(base + o_win);
  p->d_window_inv = (float*)(base + o_winv);
  p->d_window_sq = (float*)(base + o_wsq);
  p->d_tw200 = (float2*)(base + o_t200);
  p->d_tw400 = (float2*)(base + o_t400);
  p->d_mel_start = (int*)(base + o_ms);
  p->d_mel_len = (int*)(base + o_ml);
  p->d_mel_w = (float*)(base + o_mw);
  SE_HIP(hipMemcpy(p->d_window, p->h_window.data(), 400 * 4, hipMemcpyHostToDevice));
  SE_HIP(hipMemcpy(p->d_window_inv, winv.data(), 400 * 4, hipMemcpyHostToDevice));
  SE_HIP(hipMemcpy(p->d_window_sq, wsq.data(), 400 * 4, hipMemcpyHostToDevice));
  SE_HIP(hipMemcpy(p->d_tw200, tw200.data(), 200 * 8, hipMemcpyHostToDevice));
  SE_HIP(hipMemcpy(p->d_tw400, tw400.data(), 200 * 8, hipMemcpyHostToDevice));
  SE_HIP(hipMemcpy(p->d_mel_start, mstart.data(), se::kMelMax * 4, hipMemcpyHostToDevice));
  SE_HIP(hipMemcpy(p->d_mel_len, mlen.data(), se::kMelMax * 4, hipMemcpyHostToDevice));
  SE_HIP(hipMemcpy(p->d_mel_w, mw.data(), mw.size() * 4, hipMemcpyHostToDevice));
  *out = p;
  return SE_OK;
}

extern "C" void se_plan_destroy(se_plan* plan) {
  if (!plan) return;
  if (plan->d_blob) (void)hipFree(plan->d_blob);
  delete plan;
}

extern "C" int se_plan_tables(const se_plan* plan, float* window, float* mel_fb) {
  SE_REQUIRE(plan, "se_plan_tables: null plan");
  if (window) memcpy(window, plan->h_window.data(), sizeof(float) * se::kNfft);
  if (mel_fb) memcpy(mel_fb, plan->h_melfb.data(), sizeof(float) * plan->h_melfb.size());
  return SE_OK;
}

extern "C" int se_num_frames(const se_plan* plan, int n_samples) {
  if (!plan || n_samples < 0) return -1;
  return n_samples / plan->geom.hop + 1;
}
