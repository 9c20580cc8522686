// fft200.h -- 200-point complex FFT in LDS for the 400-point real transform (n_fft = 400 = 2 x 8 x 25).
//
// A 400-point real DFT is done as a 200-point complex FFT of z[n] = x[2n] + i x[2n+1] plus an O(N)
// recombination.  The 200-point FFT is a two-pass Cooley-Tukey split n = 25 m + j, k = q + 8 r:
//   pass A: 25 radix-8 butterflies per frame (over m), twiddle W200^(j q), IN PLACE (slot 25 q + j)
//   pass B: 8 in-register 25-point DFTs per frame (5 x 5, compile-time twiddles), out slot q + 8 r
// DIR = -1: forward (exp(-i...)), DIR = +1: inverse (un-normalised).
#pragma once
#include <hip/hip_runtime.h>
#include "fft_consts.h"

namespace se {

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
// multiply by exp(DIR * i * pi/2) = DIR * i
template <int DIR>
__device__ __forceinline__ float2 mul_i(float2 a) {
  return DIR < 0 ? make_float2(a.y, -a.x) : make_float2(-a.y, a.x);
}

template <int DIR>
__device__ __forceinline__ void fft4(float2 u0, float2 u1, float2 u2, float2 u3,
                                     float2& y0, float2& y1, float2& y2, float2& y3) {
  float2 s0 = cadd(u0, u2), s1 = csub(u0, u2), s2 = cadd(u1, u3), s3 = mul_i<DIR>(csub(u1, u3));
  y0 = cadd(s0, s2);
  y2 = csub(s0, s2);
  y1 = cadd(s1, s3);
  y3 = csub(s1, s3);
}

// v[q] <- sum_m v[m] exp(DIR * 2 pi i m q / 8)
template <int DIR>
__device__ __forceinline__ void fft8(float2 (&v)[8]) {
  constexpr float h = kSqrtHalf;
  float2 a0 = cadd(v[0], v[4]), a1 = cadd(v[1], v[5]), a2 = cadd(v[2], v[6]), a3 = cadd(v[3], v[7]);
  float2 b0 = csub(v[0], v[4]), b1 = csub(v[1], v[5]), b2 = csub(v[2], v[6]), b3 = csub(v[3], v[7]);
  b1 = cmul(b1, make_float2(h, DIR * h));
  b2 = mul_i<DIR>(b2);
  b3 = cmul(b3, make_float2(-h, DIR * h));
  fft4<DIR>(a0, a1, a2, a3, v[0], v[2], v[4], v[6]);
  fft4<DIR>(b0, b1, b2, b3, v[1], v[3], v[5], v[7]);
}

// (x0..x4) <- 5-point DFT, exp(DIR * 2 pi i n k / 5)
template <int DIR>
__device__ __forceinline__ void fft5(float2& x0, float2& x1, float2& x2, float2& x3, float2& x4) {
  float2 t1 = cadd(x1, x4), t2 = cadd(x2, x3), t3 = csub(x1, x4), t4 = csub(x2, x3);
  float2 m1 = make_float2(x0.x + kC5_1 * t1.x + kC5_2 * t2.x, x0.y + kC5_1 * t1.y + kC5_2 * t2.y);
  float2 m2 = make_float2(x0.x + kC5_2 * t1.x + kC5_1 * t2.x, x0.y + kC5_2 * t1.y + kC5_1 * t2.y);
  float2 n1 = make_float2(kS5_1 * t3.x + kS5_2 * t4.x, kS5_1 * t3.y + kS5_2 * t4.y);
  float2 n2 = make_float2(kS5_2 * t3.x - kS5_1 * t4.x, kS5_2 * t3.y - kS5_1 * t4.y);
  x0 = make_float2(x0.x + t1.x + t2.x, x0.y + t1.y + t2.y);
  // X1 = m1 + s*i*n1, X4 = m1 - s*i*n1 (s = DIR); i*n = (-n.y, n.x)
  constexpr float s = (float)DIR;
  x1 = make_float2(m1.x - s * n1.y, m1.y + s * n1.x);
  x4 = make_float2(m1.x + s * n1.y, m1.y - s * n1.x);
  x2 = make_float2(m2.x - s * n2.y, m2.y + s * n2.x);
  x3 = make_float2(m2.x + s * n2.y, m2.y - s * n2.x);
}

// pass A for one (frame, j): in-place radix-8 over slots 25 m + j, then twiddle exp(DIR 2 pi i j q / 200).
// tw200[t] = (cos(2 pi t / 200), sin(2 pi t / 200)), t < 200 (LDS or global).
template <int DIR>
__device__ __forceinline__ void fft200_pass_a(float2* __restrict__ frame, int j, const float2* __restrict__ tw200) {
  float2 v[8];
#pragma unroll
  for (int m = 0; m < 8; ++m) v[m] = frame[25 * m + j];
  fft8<DIR>(v);
#pragma unroll
  for (int q = 1; q < 8; ++q) {
    float2 w = tw200[j * q];
    w.y *= (float)DIR;
    v[q] = cmul(v[q], w);
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) frame[25 * q + j] = v[q];
}

// pass B compute for one (frame, q): y[j] (j = 0..24, read from slot 25 q + j) -> y[5 c + d] = OUT[q + 8 (c + 5 d)]
template <int DIR>
__device__ __forceinline__ void fft25(float2 (&y)[25]) {
#pragma unroll
  for (int b = 0; b < 5; ++b) {
    fft5<DIR>(y[b], y[5 + b], y[10 + b], y[15 + b], y[20 + b]);   // y[5c+b] = T[b][c]
#pragma unroll
    for (int c = 1; c < 5; ++c) {
      if (b > 0) y[5 * c + b] = cmul(y[5 * c + b], make_float2(kTw25Cos[b][c], DIR * kTw25Sin[b][c]));
    }
  }
#pragma unroll
  for (int c = 0; c < 5; ++c) fft5<DIR>(y[5 * c], y[5 * c + 1], y[5 * c + 2], y[5 * c + 3], y[5 * c + 4]);
}

}  // namespace se
