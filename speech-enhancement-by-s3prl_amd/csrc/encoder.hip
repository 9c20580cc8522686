// encoder.hip -- rows B1-B4 + C3: the S3PRL TRANSFORMER encoder (BERT layout) and TransformerSpecPredictionHead
// orchestrated over the bf16 MFMA GEMM (gemm.hip), the flash MHSA kernel (mhsa.hip) and the fused
// row kernels below.  GEMM operands are bf16, accumulation / residual stream / LayerNorm / softmax are fp32.
//
// Per layer (M = B*T rows, H = 768):
//   qkv  = x_bf16 Wqkv^T + b           (M, 3H) bf16        one fused 768 -> 2304 GEMM
//   ctx  = MHSA(qkv)                   (M, H)  bf16
//   a    = ctx Wo^T + b + x_f32        (M, H)  fp32        residual fused in the GEMM epilogue
//   x    = LayerNorm(a)                fp32 + bf16 copies  (TF style, eps inside the sqrt)
//   h    = gelu(x_bf16 W1^T + b)       (M, 4H) bf16        GELU(erf) fused in the GEMM epilogue
//   o    = h W2^T + b + x_f32          (M, H)  fp32
//   x    = LayerNorm(o)
#include <math.h>
#include <stdlib.h>
#include <utility>
#include <vector>
#include "common.h"
#include "bf16.h"
#include "encoder_impl.h"

extern "C" int se_gemm2_splitk_launch(const uint16_t* A, int lda, const uint16_t* W, int ldw, int M, int N, int Kc, int splits,
                                      float* partials, void* stream);

namespace se {

// one wave per row; H = 64 * 4 * NV
template <int NV, int GELU_IN = 0>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ pe, int T,
                                                        const float* __restrict__ w, const float* __restrict__ b, int M, float eps,
                                                        float* __restrict__ out_f32, uint16_t* __restrict__ out_bf16) {
  constexpr int H = 256 * NV;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (size_t)row * H;
  float4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const float4*>(xr + (i * 64 + lane) * 4);
  if (GELU_IN) {
#pragma unroll
    for (int i = 0; i < NV; ++i) { v[i].x = gelu_erf(v[i].x); v[i].y = gelu_erf(v[i].y); v[i].z = gelu_erf(v[i].z); v[i].w = gelu_erf(v[i].w); }
  }
  if (pe) {
    const float* pr = pe + (size_t)(row % T) * H;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float4 p = *reinterpret_cast<const float4*>(pr + (i * 64 + lane) * 4);
      v[i].x += p.x; v[i].y += p.y; v[i].z += p.z; v[i].w += p.w;
    }
  }
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
    float4 y;
    y.x = ww.x * (v[i].x * rstd) + bb.x; y.y = ww.y * (v[i].y * rstd) + bb.y;
    y.z = ww.z * (v[i].z * rstd) + bb.z; y.w = ww.w * (v[i].w * rstd) + bb.w;
    if (out_f32) *reinterpret_cast<float4*>(out_f32 + (size_t)row * H + c) = y;
    if (out_bf16) *reinterpret_cast<uint2*>(out_bf16 + (size_t)row * H + c) = make_uint2(pack_bf16x2(y.x, y.y), pack_bf16x2(y.z, y.w));
  }
}

// split-K finish for the small-batch path: x = LayerNorm(sum_s partial_s + bias + residual); one wave per row, H = 256 NV.
// The K = 3072 projection of a single utterance makes 48 workgroups; split four ways over K it fills the chip, and the sum of
// the four fp32 slabs costs nothing extra here (the LayerNorm pass reads its input anyway).
template <int NV>
__global__ __launch_bounds__(256) void ln_reduce_kernel(const float* __restrict__ partials, int nslab, size_t slab_stride,
                                                        const float* __restrict__ bias, const float* __restrict__ residual,
                                                        const float* __restrict__ w, const float* __restrict__ b, int M, float eps,
                                                        float* __restrict__ out_f32, uint16_t* __restrict__ out_bf16) {
  constexpr int H = 256 * NV;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    const float4 bb = *reinterpret_cast<const float4*>(bias + c);
    const float4 rr = *reinterpret_cast<const float4*>(residual + (size_t)row * H + c);
    v[i] = make_float4(bb.x + rr.x, bb.y + rr.y, bb.z + rr.z, bb.w + rr.w);
    for (int sidx = 0; sidx < nslab; ++sidx) {
      const float4 p = *reinterpret_cast<const float4*>(partials + sidx * slab_stride + (size_t)row * H + c);
      v[i].x += p.x; v[i].y += p.y; v[i].z += p.z; v[i].w += p.w;
    }
  }
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
    float4 y;
    y.x = ww.x * (v[i].x * rstd) + bb.x; y.y = ww.y * (v[i].y * rstd) + bb.y;
    y.z = ww.z * (v[i].z * rstd) + bb.z; y.w = ww.w * (v[i].w * rstd) + bb.w;
    if (out_f32) *reinterpret_cast<float4*>(out_f32 + (size_t)row * H + c) = y;
    if (out_bf16) *reinterpret_cast<uint2*>(out_bf16 + (size_t)row * H + c) = make_uint2(pack_bf16x2(y.x, y.y), pack_bf16x2(y.z, y.w));
  }
}

// generic-H fallback: one workgroup (256 threads) per row
__global__ __launch_bounds__(256) void layernorm_generic_kernel(const float* __restrict__ x, const float* __restrict__ pe, int T,
                                                                const float* __restrict__ w, const float* __restrict__ b, int H, float eps,
                                                                float* __restrict__ out_f32, uint16_t* __restrict__ out_bf16) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* xr = x + (size_t)row * H;
  const float* pr = pe ? pe + (size_t)(row % T) * H : nullptr;
  float s = 0.f;
  for (int c = tid; c < H; c += 256) s += xr[c] + (pr ? pr[c] : 0.f);
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)H;
  __syncthreads();
  float q = 0.f;
  for (int c = tid; c < H; c += 256) {
    const float d = xr[c] + (pr ? pr[c] : 0.f) - mean;
    q += d * d;
  }
  for (int off = 32; off > 0; off >>= 1) q += __shfl_down(q, off);
  if ((tid & 63) == 0) red[tid >> 6] = q;
  __syncthreads();
  const float rstd = 1.0f / sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)H + eps);
  for (int c = tid; c < H; c += 256) {
    const float y = w[c] * ((xr[c] + (pr ? pr[c] : 0.f) - mean) * rstd) + b[c];
    if (out_f32) out_f32[(size_t)row * H + c] = y;
    if (out_bf16) out_bf16[(size_t)row * H + c] = f2bf(y);
  }
}

// fp32 (rows, cols) -> bf16 (rows, ld_out) zero-padded; 4 elements per thread
__global__ __launch_bounds__(256) void cast_pad_kernel(const float* __restrict__ x, size_t rows, int cols, int ld_out,
                                                       uint16_t* __restrict__ out) {
  const size_t n4 = rows * (size_t)(ld_out / 4);
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const size_t r = i / (ld_out / 4);
    const int c = (int)(i - r * (ld_out / 4)) * 4;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    const float* xr = x + r * cols;
    if (c + 3 < cols && (cols % 4 == 0)) {
      const float4 t = *reinterpret_cast<const float4*>(xr + c);
      v0 = t.x; v1 = t.y; v2 = t.z; v3 = t.w;
    } else {
      if (c < cols) v0 = xr[c];
      if (c + 1 < cols) v1 = xr[c + 1];
      if (c + 2 < cols) v2 = xr[c + 2];
      if (c + 3 < cols) v3 = xr[c + 3];
    }
    *reinterpret_cast<uint2*>(out + r * ld_out + c) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
  }
}

// lengths[b] = #frames whose feature sum != 0 (S3PRL process_input_data).  One wave per frame (coalesced row read,
// shuffle reduction), 4 frames per workgroup pass; per-utterance counts by one integer atomic per workgroup.
__global__ __launch_bounds__(256) void valid_lengths_kernel(const float* __restrict__ feats, int T, int D, int32_t* __restrict__ lengths) {
  __shared__ int red[4];
  const int b = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int cnt = 0;
  for (int t = blockIdx.x * 4 + wv; t < T; t += gridDim.x * 4) {
    const float* fr = feats + ((size_t)b * T + t) * D;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) s += fr[d];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    cnt += (s != 0.f) ? 1 : 0;
  }
  if (lane == 0) red[wv] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&lengths[b], red[0] + red[1] + red[2] + red[3]);
}

// SpecHead.forward epilogue (model.py:121-125)
__global__ __launch_bounds__(256) void spec_epilogue_kernel(const float* __restrict__ p, size_t n, int log_target, int act, float eps,
                                                            float* __restrict__ predicted, float* __restrict__ log_predicted) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = p[i];
    float pred, lp;
    if (log_target == 2) {        // LSTM.forward (model.py:56-58): log_predicted = act(p), predicted = exp(log_predicted)
      lp = (act == SE_ACT_RELU) ? fmaxf(v, 0.f) : (act == SE_ACT_SIGMOID) ? 1.f / (1.f + expf(-v)) : v;
      pred = expf(lp);
    } else {
      if (log_target) {
        pred = expf(v);
        lp = v;
      } else {
        pred = v;
        lp = logf(v + eps);
      }
      if (act == SE_ACT_RELU) pred = fmaxf(pred, 0.f);
      else if (act == SE_ACT_SIGMOID) pred = 1.f / (1.f + expf(-pred));
    }
    if (predicted) predicted[i] = pred;
    if (log_predicted) log_predicted[i] = lp;
  }
}

// the same on a row-padded input p (M, ld): the 201-column output GEMM writes rows padded to a multiple of 4 floats so that it takes
// the specialised 16-B-store epilogue (with ldc = 201 it falls back to the all-run-time one: 59 vs 35 us at M = 32 032)
__global__ __launch_bounds__(256) void spec_epilogue_rows_kernel(const float* __restrict__ p, int M, int N, int ld, int log_target, int act, float eps,
                                                                 float* __restrict__ predicted, float* __restrict__ log_predicted) {
  for (int r = blockIdx.x; r < M; r += gridDim.x) {
    for (int c = threadIdx.x; c < N; c += 256) {
      const float v = p[(size_t)r * ld + c];
      float pred, lp;
      if (log_target) {
        pred = expf(v);
        lp = v;
      } else {
        pred = v;
        lp = logf(v + eps);
      }
      if (act == SE_ACT_RELU) pred = fmaxf(pred, 0.f);
      else if (act == SE_ACT_SIGMOID) pred = 1.f / (1.f + expf(-pred));
      const size_t o = (size_t)r * N + c;
      if (predicted) predicted[o] = pred;
      if (log_predicted) log_predicted[o] = lp;
    }
  }
}

// backward of SpecHead.forward's epilogue (model.py:121-125) wrt the raw linear output p (log_target 2: LSTM.forward's, model.py:56-58):
//   log_target: predicted = act(exp(p)), log_predicted = p      -> dp = d_logp + d_pred act'(exp p) exp(p)
//   else      : predicted = act(p),      log_predicted = log(p+eps) -> dp = d_pred act'(p) + d_logp / (p + eps)
// writes dp as fp32 (M, N) and as bf16 (M, ldp) with columns N..ldp-1 zeroed (GEMM operand padding)
__global__ __launch_bounds__(256) void spec_epilogue_bwd_kernel(const float* __restrict__ p, const float* __restrict__ d_pred,
                                                                const float* __restrict__ d_logp, int M, int N, int ldp, int log_target,
                                                                int act, float eps, float* __restrict__ dp_f32, uint16_t* __restrict__ dp_bf16) {
  const size_t n = (size_t)M * ldp;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / ldp;
    const int c = (int)(i - r * ldp);
    float g = 0.f;
    if (c < N) {
      const size_t j = r * N + c;
      const float v = p[j];
      const float gp = d_pred ? d_pred[j] : 0.f, gl = d_logp ? d_logp[j] : 0.f;
      if (log_target == 2) {      // lp = act(p), pred = exp(lp): dp = (d_logp + d_pred exp(lp)) act'(p)
        const float sg = 1.f / (1.f + expf(-v));
        const float lp = (act == SE_ACT_RELU) ? fmaxf(v, 0.f) : (act == SE_ACT_SIGMOID) ? sg : v;
        const float da = (act == SE_ACT_RELU) ? (v > 0.f ? 1.f : 0.f) : (act == SE_ACT_SIGMOID) ? sg * (1.f - sg) : 1.f;
        g = (gl + gp * expf(lp)) * da;
      } else if (log_target) {
        const float e = expf(v);
        const float da = (act == SE_ACT_RELU) ? (e > 0.f ? 1.f : 0.f) : 1.f;
        g = gl + gp * da * e;
      } else {
        const float da = (act == SE_ACT_RELU) ? (v > 0.f ? 1.f : 0.f) : 1.f;
        g = gp * da + gl / (v + eps);
      }
      if (dp_f32) dp_f32[j] = g;
    }
    if (dp_bf16) dp_bf16[i] = f2bf(g);
  }
}

}  // namespace se

namespace se {
// inference QKV copy of one layer: rows [0, H) = bf16(kQScale * W_q) from the fp32 master, rows [H, 3H) copied from the bf16 K / V rows
__global__ __launch_bounds__(256) void qkv_inf_kernel(const float* __restrict__ q_w, const float* __restrict__ q_b, const uint16_t* __restrict__ qkv_w,
                                                      const float* __restrict__ qkv_b, int H, uint16_t* __restrict__ w_inf, float* __restrict__ b_inf) {
  const size_t hh = (size_t)H * H;
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < 3 * hh; e += (size_t)gridDim.x * 256)
    w_inf[e] = e < hh ? f2bf(q_w[e] * kQScale) : qkv_w[e];
  for (int e = blockIdx.x * 256 + threadIdx.x; e < 3 * H; e += gridDim.x * 256) b_inf[e] = e < H ? q_b[e] * kQScale : qkv_b[e];
}
int launch_qkv_inf(const float* q_w, const float* q_b, const uint16_t* qkv_w, const float* qkv_b, int H, uint16_t* qkv_w_inf, float* qkv_b_inf, hipStream_t st) {
  hipLaunchKernelGGL(qkv_inf_kernel, dim3(512), dim3(256), 0, st, q_w, q_b, qkv_w, qkv_b, H, qkv_w_inf, qkv_b_inf);
  SE_LAUNCH_CHECK();
  return SE_OK;
}
}  // namespace se

static inline uint16_t host_f2bf(float f) {   // round to nearest even, NaN kept quiet
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

int se::launch_cast_pad(const float* x, size_t rows, int cols, int ld_out, uint16_t* out, hipStream_t st) {
  const int grid = (int)std::min<size_t>((rows * (size_t)(ld_out / 4) + 255) / 256, 8192);
  hipLaunchKernelGGL(se::cast_pad_kernel, dim3(grid), dim3(256), 0, st, x, rows, cols, ld_out, out);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

int se::launch_layernorm(const float* x, const float* pe, int T, const float* w, const float* b, int M, int H, float eps,
                         float* out_f32, uint16_t* out_bf16, hipStream_t st) {
  if (H == 768) {
    hipLaunchKernelGGL((se::layernorm_kernel<3>), dim3((M + 3) / 4), dim3(256), 0, st, x, pe, T, w, b, M, eps, out_f32, out_bf16);
  } else if (H == 256) {
    hipLaunchKernelGGL((se::layernorm_kernel<1>), dim3((M + 3) / 4), dim3(256), 0, st, x, pe, T, w, b, M, eps, out_f32, out_bf16);
  } else if (H == 512) {
    hipLaunchKernelGGL((se::layernorm_kernel<2>), dim3((M + 3) / 4), dim3(256), 0, st, x, pe, T, w, b, M, eps, out_f32, out_bf16);
  } else if (H == 1024) {
    hipLaunchKernelGGL((se::layernorm_kernel<4>), dim3((M + 3) / 4), dim3(256), 0, st, x, pe, T, w, b, M, eps, out_f32, out_bf16);
  } else {
    hipLaunchKernelGGL(se::layernorm_generic_kernel, dim3(M), dim3(256), 0, st, x, pe, T, w, b, H, eps, out_f32, out_bf16);
  }
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_layernorm_f32(const float* x, const float* w, const float* b, int M, int H, float eps,
                                float* out_f32, uint16_t* out_bf16, void* stream) {
  SE_REQUIRE(x && w && b && (out_f32 || out_bf16) && M > 0 && H > 0, "se_layernorm_f32: bad argument");
  return se::launch_layernorm(x, nullptr, 1, w, b, M, H, eps, out_f32, out_bf16, se::as_stream(stream));
}

extern "C" int se_gelu_layernorm_f32(const float* pre, const float* w, const float* b, int M, int H, float eps,
                                     float* out_f32, uint16_t* out_bf16, void* stream) {
  SE_REQUIRE(pre && w && b && (out_f32 || out_bf16) && M > 0, "se_gelu_layernorm_f32: bad argument");
  SE_REQUIRE(H == 768 || H == 256, "se_gelu_layernorm_f32: only H = 768 / 256 are built (got %d)", H);
  if (H == 768)
    hipLaunchKernelGGL((se::layernorm_kernel<3, 1>), dim3((M + 3) / 4), dim3(256), 0, se::as_stream(stream), pre, nullptr, 1, w, b, M, eps, out_f32, out_bf16);
  else
    hipLaunchKernelGGL((se::layernorm_kernel<1, 1>), dim3((M + 3) / 4), dim3(256), 0, se::as_stream(stream), pre, nullptr, 1, w, b, M, eps, out_f32, out_bf16);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_spec_epilogue_f32(const float* p, size_t n, int log_target, int act, float eps, float* predicted, float* log_predicted,
                                    void* stream) {
  SE_REQUIRE(p && n > 0 && (predicted || log_predicted), "se_spec_epilogue_f32: bad argument");
  const int grid = (int)std::min<size_t>((n + 255) / 256, 8192);
  hipLaunchKernelGGL(se::spec_epilogue_kernel, dim3(grid), dim3(256), 0, se::as_stream(stream), p, n, log_target, act, eps, predicted, log_predicted);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_spec_epilogue_bwd_f32(const float* p, const float* d_pred, const float* d_logp, int M, int N, int ldp, int log_target,
                                        int act, float eps, float* dp_f32, uint16_t* dp_bf16, void* stream) {
  SE_REQUIRE(p && (d_pred || d_logp) && (dp_f32 || dp_bf16) && M > 0 && N > 0 && ldp >= N, "se_spec_epilogue_bwd_f32: bad argument");
  SE_REQUIRE(act == SE_ACT_RELU || act == SE_ACT_IDENTITY || (log_target == 2 && act == SE_ACT_SIGMOID),
             "se_spec_epilogue_bwd_f32: activation must be ReLU or Identity (Sigmoid only with log_target 2)");
  const size_t n = (size_t)M * ldp;
  const int grid = (int)std::min<size_t>((n + 255) / 256, 8192);
  hipLaunchKernelGGL(se::spec_epilogue_bwd_kernel, dim3(grid), dim3(256), 0, se::as_stream(stream), p, d_pred, d_logp, M, N, ldp, log_target,
                     act, eps, dp_f32, dp_bf16);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_cast_f32_bf16(const float* x, size_t n, uint16_t* out, void* stream) {
  SE_REQUIRE(x && out && n > 0 && n % 4 == 0, "se_cast_f32_bf16: n must be a positive multiple of 4");
  const size_t rows = n / 4;
  const int grid = (int)std::min<size_t>((rows + 255) / 256, 4096);
  hipLaunchKernelGGL(se::cast_pad_kernel, dim3(grid), dim3(256), 0, se::as_stream(stream), x, rows, 4, 4, out);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_valid_lengths_i32(const float* feats, int B, int T, int D, int32_t* lengths, void* stream) {
  SE_REQUIRE(feats && lengths && B > 0 && T > 0 && D > 0, "se_valid_lengths_i32: bad argument");
  { const int zrc_ = se::zero_async(lengths, sizeof(int32_t) * B, se::as_stream(stream)); if (zrc_) return zrc_; }
  hipLaunchKernelGGL(se::valid_lengths_kernel, dim3(std::min(32, (T + 3) / 4), B), dim3(256), 0, se::as_stream(stream), feats, T, D, lengths);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_encoder_create(const se_encoder_config* cfg, const se_encoder_weights* w, se_encoder** out) {
  SE_REQUIRE(cfg && w && out, "se_encoder_create: null argument");
  const int H = cfg->hidden, I = cfg->intermediate, L = cfg->layers, D = cfg->input_dim;
  if (cfg->heads <= 0 || H != cfg->heads * 64 || H % 64 != 0 || I % 64 != 0 || D <= 0 || D > se::kInPad || L <= 0) {
    se::set_error("se_encoder_create: unsupported config (hidden=%d heads=%d intermediate=%d input_dim=%d): head dim must be 64, "
                  "hidden/intermediate multiples of 64, input_dim <= %d", H, cfg->heads, I, D, se::kInPad);
    return SE_ERR_UNSUPPORTED;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    se::set_error("se_encoder_create: no HIP device visible (no CPU fallback)");
    return SE_ERR_NO_DEVICE;
  }
  const bool has_head = cfg->spec_out > 0 && w->sh_dense_w;
  // ---- host staging blob
  std::vector<char> host;
  auto reserve = [&](size_t bytes) {
    const size_t off = (host.size() + 255) & ~(size_t)255;
    host.resize(off + bytes, 0);
    return off;
  };
  auto put_f32 = [&](const float* src, size_t n) {
    const size_t off = reserve(n * 4);
    memcpy(host.data() + off, src, n * 4);
    return off;
  };
  auto put_bf16 = [&](const float* src, int rows, int cols, int ld) {   // (rows, cols) fp32 -> (rows, ld) bf16 zero padded
    const size_t off = reserve((size_t)rows * ld * 2);
    uint16_t* d = reinterpret_cast<uint16_t*>(host.data() + off);
    for (int r = 0; r < rows; ++r)
      for (int c = 0; c < cols; ++c) d[(size_t)r * ld + c] = host_f2bf(src[(size_t)r * cols + c]);
    return off;
  };
  struct Off { size_t in_w, in_b, in_ln_w, in_ln_b, pe; std::vector<std::vector<size_t>> l; size_t shd_w, shd_b, shl_w, shl_b, sho_w, sho_b; } o;
  o.in_w = put_bf16(w->in_w, H, D, se::kInPad);
  o.in_b = put_f32(w->in_b, H);
  o.in_ln_w = put_f32(w->in_ln_w, H);
  o.in_ln_b = put_f32(w->in_ln_b, H);
  {
    o.pe = reserve((size_t)se::kMaxPos * H * 4);
    float* pe = reinterpret_cast<float*>(host.data() + o.pe);
    for (int pos = 0; pos < se::kMaxPos; ++pos)
      for (int j = 0; j < H; ++j) {
        const double ang = (double)pos / pow(10000.0, 2.0 * (double)(j / 2) / (double)H);
        pe[(size_t)pos * H + j] = (float)((j & 1) ? cos(ang) : sin(ang));
      }
  }
  for (int i = 0; i < L; ++i) {
    std::vector<size_t> lo;
    // fused QKV weight (3H, H) and bias (3H)
    {
      const size_t off = reserve((size_t)3 * H * H * 2);
      uint16_t* d = reinterpret_cast<uint16_t*>(host.data() + off);
      const float* srcs[3] = {w->q_w[i], w->k_w[i], w->v_w[i]};
      for (int p = 0; p < 3; ++p)
        for (size_t e = 0; e < (size_t)H * H; ++e) d[(size_t)p * H * H + e] = host_f2bf(srcs[p][e]);
      lo.push_back(off);
      const size_t boff = reserve((size_t)3 * H * 4);
      float* bd = reinterpret_cast<float*>(host.data() + boff);
      memcpy(bd, w->q_b[i], H * 4);
      memcpy(bd + H, w->k_b[i], H * 4);
      memcpy(bd + 2 * H, w->v_b[i], H * 4);
      lo.push_back(boff);
    }
    size_t inf_w, inf_b;
    {   // inference copy: query rows / bias scaled by log2(e) / sqrt(64) in fp32, rounded to bf16 once
      inf_w = reserve((size_t)3 * H * H * 2);
      uint16_t* d = reinterpret_cast<uint16_t*>(host.data() + inf_w);
      for (size_t e = 0; e < (size_t)H * H; ++e) d[e] = host_f2bf(w->q_w[i][e] * se::kQScale);
      for (size_t e = 0; e < (size_t)H * H; ++e) d[(size_t)H * H + e] = host_f2bf(w->k_w[i][e]);
      for (size_t e = 0; e < (size_t)H * H; ++e) d[(size_t)2 * H * H + e] = host_f2bf(w->v_w[i][e]);
      inf_b = reserve((size_t)3 * H * 4);
      float* bd = reinterpret_cast<float*>(host.data() + inf_b);
      for (int e = 0; e < H; ++e) bd[e] = w->q_b[i][e] * se::kQScale;
      memcpy(bd + H, w->k_b[i], H * 4);
      memcpy(bd + 2 * H, w->v_b[i], H * 4);
    }
    lo.push_back(put_bf16(w->ao_w[i], H, H, H));
    lo.push_back(put_f32(w->ao_b[i], H));
    lo.push_back(put_f32(w->aln_w[i], H));
    lo.push_back(put_f32(w->aln_b[i], H));
    lo.push_back(put_bf16(w->ff1_w[i], I, H, H));
    lo.push_back(put_f32(w->ff1_b[i], I));
    lo.push_back(put_bf16(w->ff2_w[i], H, I, I));
    lo.push_back(put_f32(w->ff2_b[i], H));
    lo.push_back(put_f32(w->oln_w[i], H));
    lo.push_back(put_f32(w->oln_b[i], H));
    lo.push_back(inf_w);
    lo.push_back(inf_b);
    o.l.push_back(lo);
  }
  if (has_head) {
    o.shd_w = put_bf16(w->sh_dense_w, H, H, H);
    o.shd_b = put_f32(w->sh_dense_b, H);
    o.shl_w = put_f32(w->sh_ln_w, H);
    o.shl_b = put_f32(w->sh_ln_b, H);
    o.sho_w = put_bf16(w->sh_out_w, cfg->spec_out, H, H);
    reserve((size_t)3 * H * 2);                    // up to three zero rows behind it: the output GEMM runs with N rounded up to a multiple of 4
    o.sho_b = put_f32(w->sh_out_b, cfg->spec_out);
    reserve(3 * 4);
  }
  se_encoder* e = new se_encoder();
  e->cfg = *cfg;
  if (!has_head) e->cfg.spec_out = 0;
  e->blob_bytes = host.size() + 256;
  hipError_t err = hipMalloc(&e->blob, e->blob_bytes);
  if (err != hipSuccess) {
    delete e;
    return se::hip_fail(err, "hipMalloc(encoder weights)", __FILE__, __LINE__);
  }
  err = hipMemcpy(e->blob, host.data(), host.size(), hipMemcpyHostToDevice);
  if (err != hipSuccess) {
    (void)hipFree(e->blob);
    delete e;
    return se::hip_fail(err, "hipMemcpy(encoder weights)", __FILE__, __LINE__);
  }
  char* base = reinterpret_cast<char*>(e->blob);
  e->in_w = (uint16_t*)(base + o.in_w);
  e->in_b = (float*)(base + o.in_b);
  e->in_ln_w = (float*)(base + o.in_ln_w);
  e->in_ln_b = (float*)(base + o.in_ln_b);
  e->pe = (float*)(base + o.pe);
  for (int i = 0; i < L; ++i) {
    const auto& lo = o.l[i];
    se_encoder::Layer y;
    y.qkv_w = (uint16_t*)(base + lo[0]); y.qkv_b = (float*)(base + lo[1]);
    y.ao_w = (uint16_t*)(base + lo[2]); y.ao_b = (float*)(base + lo[3]);
    y.aln_w = (float*)(base + lo[4]); y.aln_b = (float*)(base + lo[5]);
    y.ff1_w = (uint16_t*)(base + lo[6]); y.ff1_b = (float*)(base + lo[7]);
    y.ff2_w = (uint16_t*)(base + lo[8]); y.ff2_b = (float*)(base + lo[9]);
    y.oln_w = (float*)(base + lo[10]); y.oln_b = (float*)(base + lo[11]);
    y.qkv_w_inf = (uint16_t*)(base + lo[12]); y.qkv_b_inf = (float*)(base + lo[13]);
    e->layers.push_back(y);
  }
  if (has_head) {
    e->sh_dense_w = (uint16_t*)(base + o.shd_w); e->sh_dense_b = (float*)(base + o.shd_b);
    e->sh_ln_w = (float*)(base + o.shl_w); e->sh_ln_b = (float*)(base + o.shl_b);
    e->sh_out_w = (uint16_t*)(base + o.sho_w); e->sh_out_b = (float*)(base + o.sho_b);
  }
  *out = e;
  return SE_OK;
}

extern "C" void se_encoder_destroy(se_encoder* enc) {
  if (!enc) return;
  if (enc->blob) (void)hipFree(enc->blob);
  delete enc;
}

namespace {
struct Ws {
  uint16_t *xin, *x_bf, *qkv, *ctx, *h;
  float *x_f32, *tmp, *slabs;
  uint8_t* lo;           // low bytes of the 24-bit residual stream (row-complete path)
  char* g8;              // gemm8's statistics slabs + flags
  size_t total;
};
constexpr size_t kSplitKRows = 3072;   // small-batch path: split-K FFN2 up to this many rows (B = 1: 0.73 vs 0.85 ms, B = 2: 0.84 vs 0.93; B = 4: no gain)
constexpr int kSplitK = 4;
using se::al256;
Ws carve(const se_encoder* e, size_t M, char* base) {
  const size_t H = e->cfg.hidden, I = e->cfg.intermediate;
  Ws w;
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += al256(bytes); return p; };
  w.xin = (uint16_t*)take(M * se::kInPad * 2);
  w.x_bf = (uint16_t*)take(M * H * 2);
  w.x_f32 = (float*)take(M * H * 4);
  w.tmp = (float*)take(M * H * 4);
  w.qkv = (uint16_t*)take(M * 3 * H * 2);
  w.ctx = (uint16_t*)take(M * H * 2);
  w.h = (uint16_t*)take(M * I * 2);
  w.slabs = (float*)take(M <= kSplitKRows ? (size_t)kSplitK * M * H * 4 : 0);
  w.lo = (uint8_t*)take(H == 768 ? se::gemm4_lo_bytes((int)M) : 0);
  w.g8 = take(H == 768 ? se::gemm8_scratch_bytes() : 0);
  w.total = off;
  return w;
}
}  // namespace

extern "C" size_t se_encoder_workspace_bytes(const se_encoder* enc, int B, int T) {
  if (!enc || B <= 0 || T <= 0) return 0;
  return carve(enc, (size_t)B * T, nullptr).total + 256;
}

extern "C" int se_encoder_fwd_bf16(const se_encoder* enc, const float* feats, const int32_t* lengths, int B, int T,
                                   float* hidden, void* workspace, size_t workspace_bytes, void* stream) {
  return se_encoder_fwd2_bf16(enc, feats, nullptr, lengths, B, T, hidden, workspace, workspace_bytes, stream);
}

extern "C" int se_encoder_fwd2_bf16(const se_encoder* enc, const float* feats, const uint16_t* feats_bf16_pad, const int32_t* lengths, int B, int T,
                                    float* hidden, void* workspace, size_t workspace_bytes, void* stream) {
  SE_REQUIRE(enc && (feats || feats_bf16_pad) && hidden && workspace, "se_encoder_fwd_bf16: null argument");
  SE_REQUIRE((uintptr_t)feats_bf16_pad % 16 == 0, "se_encoder_fwd2_bf16: the bf16 feature rows must be 16-B aligned");
  SE_REQUIRE(B > 0 && B <= 65535 && T > 0 && T <= se::kMaxPos, "se_encoder_fwd_bf16: bad shape B=%d T=%d (T <= %d)", B, T, se::kMaxPos);
  SE_REQUIRE(workspace_bytes >= se_encoder_workspace_bytes(enc, B, T), "se_encoder_fwd_bf16: workspace too small");
  SE_REQUIRE((uintptr_t)workspace % 256 == 0 && (uintptr_t)hidden % 16 == 0, "se_encoder_fwd_bf16: workspace must be 256-B aligned");
  const int H = enc->cfg.hidden, I = enc->cfg.intermediate, D = enc->cfg.input_dim;
  const size_t Mz = (size_t)B * T;
  SE_REQUIRE(Mz <= 0x7fffffff / 4, "se_encoder_fwd_bf16: B*T too large");
  const int M = (int)Mz;
  hipStream_t st = se::as_stream(stream);
  Ws w = carve(enc, Mz, reinterpret_cast<char*>(workspace));
  int rc;
  // B1: input projection + positional encoding + LayerNorm
  if (feats_bf16_pad) {       // the feature kernel already wrote the projection's operand (se_features2_f32: bf16 rows zero-padded to kInPad columns)
    w.xin = const_cast<uint16_t*>(feats_bf16_pad);
  } else {
    const int grid = (int)std::min<size_t>((Mz * (se::kInPad / 4) + 255) / 256, 8192);
    hipLaunchKernelGGL(se::cast_pad_kernel, dim3(grid), dim3(256), 0, st, feats, Mz, D, se::kInPad, w.xin);
    SE_LAUNCH_CHECK();
  }
  const int L = enc->cfg.layers;
  static int fuse_env = -1;
  if (fuse_env < 0) {
    const char* e = getenv("SE_AMD_FUSED_LN");
    fuse_env = e ? atoi(e) : 1;
  }
  // the row-complete GEMM + LayerNorm kernel owns 128 x 768 outputs per workgroup: M / 128 workgroups.  It needs ~a full round of the
  // 256 CUs to pay off (B = 32: 251 workgroups).  Threshold re-measured with the round-2 kernels (ms per pass, unfused / row-complete):
  // B = 12: 2.28 / 2.42, B = 16: 2.66 / 2.61, B = 20: 3.10 / 2.87, B = 24: 3.94 / 3.28 -- the round-1 threshold of 24 576 rows left B = 17..24 on
  // the slower side; it is now 16 000 rows (B = 1: 8 workgroups, 0.85 vs 1.70 ms).  SE_AMD_FUSED_LN = 2 forces it.  Between its launches the residual stream travels as
  // bf16 + int8 (24 bits, gemm4.hip) instead of fp32 + bf16: 196 instead of 295 MB per K = 768 launch.
  const bool fused = fuse_env && H == 768 && I % 32 == 0 && I >= 128 && (fuse_env == 2 || M >= (enc->cfg.fused_ln_min_rows > 0 ? enc->cfg.fused_ln_min_rows : 16000));
#ifdef SE_AMD_EXPERIMENTS
  static int use8 = -1;
  if (use8 < 0) { const char* e8 = getenv("SE_AMD_GEMM8"); use8 = e8 ? atoi(e8) : 0; }
  if (fused && use8) {      // gemm8's pair flags + error word start every pass at zero (each launch also leaves them zero); the kernel is off by default
    const size_t fo = se::gemm8_scratch_bytes() - (128 * 2 * 4 + 256);
    if ((rc = se::zero_async(w.g8 + fo, 128 * 2 * 4 + 256, st))) return rc;
  }
#endif
  if (fused) {
    // one row-complete kernel: the positional table rides the residual input (row index modulo T)
    if ((rc = se::launch_gemm_pos_ln(w.xin, se::kInPad, enc->in_w, se::kInPad, enc->in_b, enc->pe, T, enc->in_ln_w, enc->in_ln_b, enc->cfg.ln_eps, M, H,
                                     se::kInPad, nullptr, w.x_bf, w.lo, st))) return rc;
  } else {
    if ((rc = se_gemm_bf16(w.xin, se::kInPad, enc->in_w, se::kInPad, enc->in_b, nullptr, M, H, se::kInPad, SE_ACT_IDENTITY, nullptr, w.tmp, H, stream))) return rc;
    if ((rc = se::launch_layernorm(w.tmp, enc->pe, T, enc->in_ln_w, enc->in_ln_b, M, H, enc->cfg.ln_eps, w.x_f32, w.x_bf, st))) return rc;
  }
  for (int i = 0; i < L; ++i) {
    const se_encoder::Layer& y = enc->layers[i];
    // B2
    // (inference copy of the projection: the queries come out pre-scaled by log2(e) / sqrt(64), see mhsa.hip PRE)
    if ((rc = se_gemm_bf16(w.x_bf, H, y.qkv_w_inf, H, y.qkv_b_inf, nullptr, M, 3 * H, H, SE_ACT_IDENTITY, w.qkv, nullptr, 3 * H, stream))) return rc;
    if ((rc = se_mhsa_fwd_prescaled_bf16(w.qkv, lengths, B, T, enc->cfg.heads, w.ctx, stream))) return rc;
    // attention output projection + residual + LayerNorm: one fused row-complete kernel when H == 768 (ping-pong x buffers:
    // the residual is read while the new stream is written), else GEMM + LayerNorm
    if (fused) {      // in place: a workgroup reads the residual rows of its own tile before it writes them
      if ((rc = se::launch_gemm_res24_ln(w.ctx, H, y.ao_w, H, y.ao_b, w.x_bf, w.lo, y.aln_w, y.aln_b, enc->cfg.ln_eps, M, H, H, nullptr, w.x_bf, w.lo, st))) return rc;
    } else {
      if ((rc = se_gemm_bf16(w.ctx, H, y.ao_w, H, y.ao_b, w.x_f32, M, H, H, SE_ACT_IDENTITY, nullptr, w.tmp, H, stream))) return rc;
      if ((rc = se::launch_layernorm(w.tmp, nullptr, 1, y.aln_w, y.aln_b, M, H, enc->cfg.ln_eps, w.x_f32, w.x_bf, st))) return rc;
    }
    // B3
    if ((rc = se_gemm_bf16(w.x_bf, H, y.ff1_w, H, y.ff1_b, nullptr, M, I, H, SE_ACT_GELU, w.h, nullptr, I, stream))) return rc;
    if (fused) {      // the last layer leaves the stream as the fp32 `hidden` the caller asked for
      float* of = (i == L - 1) ? hidden : nullptr;
      uint16_t* ob = w.x_bf;      // the last layer writes it too (next to the fp32 `hidden`): the spec head's operand, se_spechead_fwd2_bf16(x_bf_valid = 1)
      uint8_t* ol = (i == L - 1) ? nullptr : w.lo;
      // 256 x 384 tiles with the LayerNorm statistics exchanged between the two column halves (gemm8.hip) where it applies, else 128 x 768
      rc = 1;
#ifdef SE_AMD_EXPERIMENTS
      rc = se::launch_gemm8_res24_ln(w.h, I, y.ff2_w, I, y.ff2_b, w.x_bf, w.lo, y.oln_w, y.oln_b, enc->cfg.ln_eps, M, H, I, of, ob, ol, w.g8, st);
#endif
      if (rc == 1) rc = se::launch_gemm_res24_ln(w.h, I, y.ff2_w, I, y.ff2_b, w.x_bf, w.lo, y.oln_w, y.oln_b, enc->cfg.ln_eps, M, H, I, of, ob, ol, st);
      if (rc) return rc;
    } else {
      float* xo = (i == L - 1) ? hidden : w.x_f32;
      if (Mz <= kSplitKRows && H == 768 && I % (kSplitK * 64) == 0 && I / kSplitK >= 128) {
        // serving-size batch: split the K = 3072 reduction four ways (4 x the workgroups), finish in the LayerNorm pass
        if ((rc = se_gemm2_splitk_launch(w.h, I, y.ff2_w, I, M, H, I / kSplitK, kSplitK, w.slabs, stream))) return rc;
        hipLaunchKernelGGL((se::ln_reduce_kernel<3>), dim3((M + 3) / 4), dim3(256), 0, st, w.slabs, kSplitK, Mz * H, y.ff2_b, w.x_f32, y.oln_w, y.oln_b, M,
                           enc->cfg.ln_eps, xo, w.x_bf);
        SE_LAUNCH_CHECK();
      } else {
        if ((rc = se_gemm_bf16(w.h, I, y.ff2_w, I, y.ff2_b, w.x_f32, M, H, I, SE_ACT_IDENTITY, nullptr, w.tmp, H, stream))) return rc;
        if ((rc = se::launch_layernorm(w.tmp, nullptr, 1, y.oln_w, y.oln_b, M, H, enc->cfg.ln_eps, xo, w.x_bf, st))) return rc;
      }
    }
  }
  return SE_OK;
}

extern "C" int se_spechead_fwd_bf16(const se_encoder* enc, const float* hidden, int B, int T, int log_target, int act, float eps,
                                    float* predicted, float* log_predicted, float* raw,
                                    void* workspace, size_t workspace_bytes, void* stream) {
  return se_spechead_fwd2_bf16(enc, hidden, B, T, log_target, act, eps, predicted, log_predicted, raw, workspace, workspace_bytes, 0, stream);
}

extern "C" int se_spechead_fwd2_bf16(const se_encoder* enc, const float* hidden, int B, int T, int log_target, int act, float eps,
                                     float* predicted, float* log_predicted, float* raw,
                                     void* workspace, size_t workspace_bytes, int x_bf_valid, void* stream) {
  SE_REQUIRE(enc && hidden && workspace && (predicted || log_predicted || raw), "se_spechead_fwd_bf16: null argument");
  SE_REQUIRE(enc->cfg.spec_out > 0, "se_spechead_fwd_bf16: encoder was created without a spec head");
  SE_REQUIRE(B > 0 && T > 0, "se_spechead_fwd_bf16: bad shape");
  SE_REQUIRE(workspace_bytes >= se_encoder_workspace_bytes(enc, B, T), "se_spechead_fwd_bf16: workspace too small");
  const int H = enc->cfg.hidden, N = enc->cfg.spec_out;
  const size_t Mz = (size_t)B * T;
  const int M = (int)Mz;
  hipStream_t st = se::as_stream(stream);
  Ws w = carve(enc, Mz, reinterpret_cast<char*>(workspace));
  int rc;
  if (!x_bf_valid) {      // else: `hidden` is the output of the last se_encoder_fwd_bf16 on this workspace, whose final launch left its bf16 copy in x_bf
    const int grid = (int)std::min<size_t>((Mz * (H / 4) + 255) / 256, 8192);
    hipLaunchKernelGGL(se::cast_pad_kernel, dim3(grid), dim3(256), 0, st, hidden, Mz, H, H, w.x_bf);
    SE_LAUNCH_CHECK();
  }
  // dense -> gelu -> LayerNorm: one row-complete kernel at large M (as the encoder's projections), GEMM + LayerNorm otherwise
  static int fuse_env = -1;
  if (fuse_env < 0) {
    const char* e = getenv("SE_AMD_FUSED_LN");
    fuse_env = e ? atoi(e) : 1;
  }
  if (fuse_env && H == 768 && (fuse_env == 2 || M >= (enc->cfg.fused_ln_min_rows > 0 ? enc->cfg.fused_ln_min_rows : 16000))) {
    if ((rc = se::launch_gemm_gelu_ln(w.x_bf, H, enc->sh_dense_w, H, enc->sh_dense_b, enc->sh_ln_w, enc->sh_ln_b, enc->cfg.ln_eps, M, H, H, nullptr,
                                      w.ctx, st))) return rc;
  } else {
    if ((rc = se_gemm_bf16(w.x_bf, H, enc->sh_dense_w, H, enc->sh_dense_b, nullptr, M, H, H, SE_ACT_GELU, nullptr, w.tmp, H, stream))) return rc;
    if ((rc = se::launch_layernorm(w.tmp, nullptr, 1, enc->sh_ln_w, enc->sh_ln_b, M, H, enc->cfg.ln_eps, nullptr, w.ctx, st))) return rc;
  }
  // raw linear output p (M, N) fp32: into `raw` if given, else the (free) x_f32 workspace
  float* p = raw ? raw : w.x_f32;
  // internal buffer: N rounded up to a multiple of 4 (zero weight rows / bias entries follow the real ones in the blob) and rows padded
  // to 32 B, so the GEMM takes its specialised 16-B-store epilogue; x_f32 holds M x H floats
  const int Np = raw ? N : (N + 3) & ~3;
  const int ldp = raw ? N : (N + 7) & ~7;
  SE_REQUIRE(ldp <= H, "se_spechead_fwd_bf16: spec_out %d exceeds the hidden size %d", N, H);
  if ((rc = se_gemm_bf16(w.ctx, H, enc->sh_out_w, H, enc->sh_out_b, nullptr, M, Np, H, SE_ACT_IDENTITY, nullptr, p, ldp, stream))) return rc;
  if (predicted || log_predicted) {
    hipLaunchKernelGGL(se::spec_epilogue_rows_kernel, dim3(std::min(M, 8192)), dim3(256), 0, st, p, M, N, ldp, log_target, act, eps, predicted,
                       log_predicted);
    SE_LAUNCH_CHECK();
  }
  return SE_OK;
}
