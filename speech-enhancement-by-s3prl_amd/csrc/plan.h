// plan.h -- preprocessor plan: geometry + device tables (window, twiddles, sparse mel bank).
#pragma once
#include <vector>
#include "common.h"

namespace se {
constexpr int kNfft = 400;
constexpr int kHalf = 200;      // complex FFT size
constexpr int kBins = 201;
constexpr int kHop = 160;
constexpr int kMelMaxW = 32;    // max bins under one mel triangle
constexpr int kMelMax = 128;     // 40 (the reference's configs) ... 128 (the MFCC branch's own mel bank, torchaudio's default)
}  // namespace se

struct se_plan {
  se_geometry geom;
  int device;
  // host copies
  std::vector<float> h_window;   // n_fft (win centred, zero padded)
  std::vector<float> h_melfb;    // (n_freq, n_mels)
  // device tables (one allocation)
  void* d_blob;
  float* d_window;      // [400]   forward window
  float* d_window_inv;  // [400]   window / 200 (inverse transform scale folded in)
  float* d_window_sq;   // [400]   window^2
  float2* d_tw200;      // [200]   (cos, sin)(2 pi t / 200)
  float2* d_tw400;      // [200]   (cos, sin)(2 pi k / 400), k < 200 (also serves W200^t = tw400[2t] / -tw400[2t-200])
  int* d_mel_start;     // [kMelMax]
  int* d_mel_len;       // [kMelMax]
  float* d_mel_w;       // [kMelMax][kMelMaxW]
};
