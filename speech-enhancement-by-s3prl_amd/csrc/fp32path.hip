// fp32path.hip -- the EXACT-fp32 building blocks of the encoder's parity mode (rows B1-B4 at the tolerance north_star states for enhanced
// magnitudes, 1e-4 relative; BASELINE.json's bench configuration is bf16 and stays the default).
//
// The bf16 kernels round every GEMM operand to 8 mantissa bits; no amount of care in them reaches 1e-4 against the reference's fp32
// PyTorch path.  This file is the other end of the trade: fp32 operands, fp32 products, fp32 sums, libm-grade erf / exp, the score matrix
// materialised as the reference does (runner.py:273-284 -> S3PRL's BERT attention: scores + additive -10000 mask -> softmax -> P V).
//   se_gemm_f32        C = epilogue(A . W^T + bias [+ residual]) on v_mfma_f32_32x32x2_f32 (fp32 in / fp32 accumulate: bit for bit a k-ordered
//                      fmaf chain, 1/16 of the bf16 matrix rate -- gfx950 has no TF32), batched over two levels (utterance, head) with free
//                      strides, W either (N, K) row-major (nn.Linear, K^T of attention) or (K, N) row-major (the V operand of P . V)
//   se_softmax_rows_f32  in-place masked row softmax of the score tensor (exact expf, fp32 sums; keys >= length get the reference's
//                      additive -10000 before the maximum, i.e. exp() == 0 in fp32)
// Throughput is not the point (a 10 s utterance through the 6-layer encoder takes tens of ms); the tile is a plain 64 x 64 x 16 LDS-staged loop.
#include <algorithm>
#include "common.h"
#include "bf16.h"

namespace se {

typedef __attribute__((ext_vector_type(16))) float f32x16_;

constexpr int kFT = 64, kFK = 16;       // output tile 64 x 64, K-tile 16; 256 threads = 4 waves as 2 x 2, wave tile 32 x 32 (one MFMA accumulator)

__device__ __forceinline__ float gelu_exact(float v) { return v * 0.5f * (1.0f + erff(v * 0.70710678118654752440f)); }

// element (m, k) of A at A[m * lda + k]; element (n, k) of W at W[n * ldw + k] (WK = 0) or W[k * ldw + n] (WK = 1)
template <int WK>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, long lda, const float* __restrict__ W, long ldw,
                                                       const float* __restrict__ bias, const float* __restrict__ residual, int res_mod, int M, int N,
                                                       int K, int act, float alpha, float* __restrict__ C, long ldc, int inner, long sA0, long sA1,
                                                       long sW0, long sW1, long sC0, long sC1) {
  __shared__ float As[kFK][kFT + 1];      // [k][m]
  __shared__ float Ws[kFK][kFT + 1];      // [k][n]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int z = blockIdx.z, z0 = z / inner, z1 = z - z0 * inner;
  A += z0 * sA0 + z1 * sA1;
  W += z0 * sW0 + z1 * sW1;
  C += z0 * sC0 + z1 * sC1;
  const int m0 = blockIdx.y * kFT, n0 = blockIdx.x * kFT;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  f32x16_ acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // v_mfma_f32_32x32x2_f32: A operand lane l -> row (l & 31), k = l >> 5; B operand lane l -> column (l & 31), k = l >> 5;
  // D: lane l -> column (l & 31), rows (r & 3) + 8 (r >> 2) + 4 (l >> 5)
  const int lr = lane & 31, lk = lane >> 5;
  for (int k0 = 0; k0 < K; k0 += kFK) {
    // stage: 64 x 16 elements of each operand, 4 per thread; out-of-range -> 0
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 256 * i;
      const int kk = e & 15, mm = e >> 4;             // k fastest: contiguous for row-major A / (N, K) W
      const int gm = m0 + mm, gk = k0 + kk;
      As[kk][mm] = (gm < M && gk < K) ? A[(long)gm * lda + gk] : 0.f;
      if (WK == 0) {
        const int gn = n0 + mm;
        Ws[kk][mm] = (gn < N && gk < K) ? W[(long)gn * ldw + gk] : 0.f;
      } else {
        const int nn = e & 63, k2 = e >> 6;           // n fastest: contiguous for (K, N) W
        const int gn = n0 + nn, gk2 = k0 + k2;
        Ws[k2][nn] = (gn < N && gk2 < K) ? W[(long)gk2 * ldw + gn] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < kFK; ks += 2) {
      const float a = As[ks + lk][wm + lr];
      const float b = Ws[ks + lk][wn + lr];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  const int gn = n0 + wn + lr;
  if (gn >= N) return;
  const float bv = bias ? bias[gn] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int gm = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * lk;
    if (gm >= M) continue;
    float v = fmaf(acc[r], alpha, bv);
    if (act == SE_ACT_GELU) v = gelu_exact(v);
    if (residual) v += residual[(long)(res_mod > 0 ? gm % res_mod : gm) * N + gn];
    C[(long)gm * ldc + gn] = v;
  }
}

// S: (B, heads, T, T) scores; row (b, h, i): p_j = softmax_j(S_ij + (j >= len[b] ? -10000 : 0)).  One wave per row.
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ S, const int32_t* __restrict__ lengths, int heads, int T, long rows) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const int b = (int)(row / ((long)heads * T));
  const int len = lengths ? min(max(lengths[b], 0), T) : T;
  float* s = S + row * T;
  float mx = -INFINITY;
  for (int j = lane; j < T; j += 64) mx = fmaxf(mx, s[j] + (j >= len ? -10000.0f : 0.f));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sum = 0.f;
  for (int j = lane; j < T; j += 64) {
    const float e = expf(s[j] + (j >= len ? -10000.0f : 0.f) - mx);
    s[j] = e;
    sum += e;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float inv = 1.0f / sum;
  for (int j = lane; j < T; j += 64) s[j] *= inv;
}

// ---- three-term bf16 split ("bf16x3"): x = x1 + x2 + r with x1 = bf16(x), x2 = bf16(x - x1), |r| <= 2^-17 |x|.  A product of two fp32 numbers is
// x1 w1 + x1 w2 + x2 w1 up to 3 . 2^-18 relative; with the three operand slices laid side by side along K the sum of the three products is ONE
// bf16 GEMM of depth 3 K on the fast kernels (exact bf16 products, fp32 accumulation):  [x1 | x1 | x2] . [w1 | w2 | w1]^T.
// which = 0: activation layout [x1 | x1 | x2], which = 1: weight layout [w1 | w2 | w1]; each slice Kp >= cols columns wide, zero padded.
template <int VEC>
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, long ld, int rows, int cols, int Kp, int which, uint16_t* __restrict__ out) {
  // one thread = 8 columns of one row (two 16-B loads when VEC: ld and cols multiples of 4, base 16-B aligned; three 16-B stores)
  const int cpr = Kp >> 3;                                       // 8-column chunks per row
  const long total = (long)rows * cpr;
  for (long idx = blockIdx.x * 256L + threadIdx.x; idx < total; idx += gridDim.x * 256L) {
    const long row = idx / cpr;
    const int c = (int)(idx - row * cpr) * 8;
    const float* xr = x + row * ld;
    uint16_t* o = out + row * 3 * (long)Kp;
    float v[8];
    if (VEC && c + 8 <= cols) {
      const float4 a = *reinterpret_cast<const float4*>(xr + c), b = *reinterpret_cast<const float4*>(xr + c + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (c + j < cols) ? xr[c + j] : 0.f;
    }
    bf16x8 hi, mid;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const __bf16 h = (__bf16)v[j];
      hi[j] = h;
      mid[j] = (__bf16)(v[j] - (float)h);
    }
    *reinterpret_cast<bf16x8*>(o + c) = hi;
    *reinterpret_cast<bf16x8*>(o + Kp + c) = which ? mid : hi;
    *reinterpret_cast<bf16x8*>(o + 2 * (long)Kp + c) = which ? hi : mid;
  }
}


// LayerNorm in fp32 (the arithmetic of encoder.hip: layernorm_kernel) that also writes its output as the three-term activation operand [y1 | y1 | y2]
// (row stride 3 Kp, zero beyond H only if the caller cleared the buffer: Kp == H in every use) -- round 4: the bf16x3 mode's split pass behind every
// LayerNorm read the fp32 rows back.  One wave per row, H = 256 NV.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_x3_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, int M, float eps,
                                                           float* __restrict__ out_f32, uint16_t* __restrict__ out3, int Kp) {
  constexpr int H = 256 * NV;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (size_t)row * H;
  float4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const float4*>(xr + (i * 64 + lane) * 4);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  const float mean = s * (1.0f / H);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
    q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
  const float rstd = 1.0f / sqrtf(q * (1.0f / H) + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    const float4 ww = *reinterpret_cast<const float4*>(w + c), bb = *reinterpret_cast<const float4*>(b + c);
    float y[4];
    y[0] = ww.x * (v[i].x * rstd) + bb.x; y[1] = ww.y * (v[i].y * rstd) + bb.y;
    y[2] = ww.z * (v[i].z * rstd) + bb.z; y[3] = ww.w * (v[i].w * rstd) + bb.w;
    if (out_f32) *reinterpret_cast<float4*>(out_f32 + (size_t)row * H + c) = make_float4(y[0], y[1], y[2], y[3]);
    uint16_t hi[4], mid[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const __bf16 h = (__bf16)y[e];
      hi[e] = __builtin_bit_cast(uint16_t, h);
      mid[e] = __builtin_bit_cast(uint16_t, (__bf16)(y[e] - (float)h));
    }
    const uint2 ph = make_uint2(hi[0] | ((uint32_t)hi[1] << 16), hi[2] | ((uint32_t)hi[3] << 16));
    const uint2 pm = make_uint2(mid[0] | ((uint32_t)mid[1] << 16), mid[2] | ((uint32_t)mid[3] << 16));
    uint16_t* o = out3 + (size_t)row * 3 * (size_t)Kp + c;
    *reinterpret_cast<uint2*>(o) = ph;
    *reinterpret_cast<uint2*>(o + Kp) = ph;
    *reinterpret_cast<uint2*>(o + 2 * (size_t)Kp) = pm;
  }
}

}  // namespace se

extern "C" int se_layernorm_x3_f32(const float* x, const float* w, const float* b, int M, int H, float eps, float* out_f32, uint16_t* out3, int Kp,
                                   void* stream) {
  SE_REQUIRE(x && w && b && out3 && M > 0, "se_layernorm_x3_f32: null argument");
  SE_REQUIRE(H == 768 || H == 256 || H == 512 || H == 1024, "se_layernorm_x3_f32: built for H = 256, 512, 768, 1024 (got %d)", H);
  // Kp == H: this producer writes columns [0, H) of each slice only; a wider slice would leave pad columns that the next projection sums over
  SE_REQUIRE(Kp == H && Kp % 8 == 0 && (((uintptr_t)x | (uintptr_t)out_f32 | (uintptr_t)out3 | (uintptr_t)w | (uintptr_t)b) % 16) == 0,
             "se_layernorm_x3_f32: Kp = %d must be == H (no pad columns: they are not written), a multiple of 8; buffers 16-B aligned", Kp);
  hipStream_t st = se::as_stream(stream);
  const dim3 grid((M + 3) / 4);
  switch (H) {
    case 256: hipLaunchKernelGGL((se::layernorm_x3_kernel<1>), grid, dim3(256), 0, st, x, w, b, M, eps, out_f32, out3, Kp); break;
    case 512: hipLaunchKernelGGL((se::layernorm_x3_kernel<2>), grid, dim3(256), 0, st, x, w, b, M, eps, out_f32, out3, Kp); break;
    case 768: hipLaunchKernelGGL((se::layernorm_x3_kernel<3>), grid, dim3(256), 0, st, x, w, b, M, eps, out_f32, out3, Kp); break;
    default: hipLaunchKernelGGL((se::layernorm_x3_kernel<4>), grid, dim3(256), 0, st, x, w, b, M, eps, out_f32, out3, Kp); break;
  }
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_split3_bf16(const float* x, long ld, int rows, int cols, int Kp, int which, uint16_t* out, void* stream) {
  SE_REQUIRE(x && out && rows > 0 && cols > 0 && Kp >= cols && Kp % 8 == 0 && ld >= cols, "se_split3_bf16: bad argument (rows=%d cols=%d Kp=%d)", rows, cols, Kp);
  SE_REQUIRE(rows <= 0x7fffffff / 2 && ((uintptr_t)out % 16) == 0, "se_split3_bf16: bad argument");
  SE_REQUIRE(which == 0 || which == 1, "se_split3_bf16: which = 0 (activations) or 1 (weights)");
  const bool vec = (ld % 4 == 0) && (cols % 4 == 0) && (((uintptr_t)x) % 16 == 0);
  const long total = (long)rows * (Kp / 8);
  const unsigned grid = (unsigned)std::min<long>((total + 255) / 256, 1 << 20);
  if (vec)
    hipLaunchKernelGGL(se::split3_kernel<1>, dim3(grid), dim3(256), 0, se::as_stream(stream), x, ld, rows, cols, Kp, which, out);
  else
    hipLaunchKernelGGL(se::split3_kernel<0>, dim3(grid), dim3(256), 0, se::as_stream(stream), x, ld, rows, cols, Kp, which, out);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_gemm_f32(const float* A, long lda, const float* W, long ldw, int w_kmajor, const float* bias, const float* residual, int res_mod,
                           int M, int N, int K, int act, float alpha, float* C, long ldc, int batch_outer, int batch_inner, long strideA_outer,
                           long strideA_inner, long strideW_outer, long strideW_inner, long strideC_outer, long strideC_inner, void* stream) {
  SE_REQUIRE(A && W && C, "se_gemm_f32: null argument");
  SE_REQUIRE(M > 0 && N > 0 && K > 0 && batch_outer > 0 && batch_inner > 0, "se_gemm_f32: bad shape M=%d N=%d K=%d batch=%d x %d", M, N, K, batch_outer, batch_inner);
  SE_REQUIRE((long)batch_outer * batch_inner <= 65535, "se_gemm_f32: batch %ld exceeds grid.z", (long)batch_outer * batch_inner);
  SE_REQUIRE(act == SE_ACT_IDENTITY || act == SE_ACT_GELU, "se_gemm_f32: activation %d not supported (identity / gelu)", act);
  SE_REQUIRE(!residual || ((long)batch_outer * batch_inner == 1), "se_gemm_f32: a residual is only defined for unbatched calls");
  dim3 grid((N + se::kFT - 1) / se::kFT, (M + se::kFT - 1) / se::kFT, batch_outer * batch_inner);
  SE_REQUIRE(grid.y <= 65535, "se_gemm_f32: M=%d too large", M);
  hipStream_t st = se::as_stream(stream);
  if (w_kmajor)
    hipLaunchKernelGGL(se::gemm_f32_kernel<1>, grid, dim3(256), 0, st, A, lda, W, ldw, bias, residual, res_mod, M, N, K, act, alpha, C, ldc, batch_inner,
                       strideA_outer, strideA_inner, strideW_outer, strideW_inner, strideC_outer, strideC_inner);
  else
    hipLaunchKernelGGL(se::gemm_f32_kernel<0>, grid, dim3(256), 0, st, A, lda, W, ldw, bias, residual, res_mod, M, N, K, act, alpha, C, ldc, batch_inner,
                       strideA_outer, strideA_inner, strideW_outer, strideW_inner, strideC_outer, strideC_inner);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_softmax_rows_f32(float* scores, const int32_t* lengths, int B, int heads, int T, void* stream) {
  SE_REQUIRE(scores && B > 0 && heads > 0 && T > 0, "se_softmax_rows_f32: bad argument");
  const long rows = (long)B * heads * T;
  SE_REQUIRE((rows + 3) / 4 <= 0x7fffffffL, "se_softmax_rows_f32: too many rows");
  hipLaunchKernelGGL(se::softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, se::as_stream(stream), scores, lengths, heads, T, rows);
  SE_LAUNCH_CHECK();
  return SE_OK;
}
