// head.hip -- rows C1 / C2: LinearResidual.forward (model.py:28-34) and Linear.forward (model.py:14-17):
//   [CMVN over time] -> x W^T + b -> activation -> [ (.) noisy power ]
// fused into one pass over the features, in EXACT fp32 (v_mfma_f32_32x32x2_f32 == k-ordered fmaf chain).
//
//   colstats : per (utterance, feature dim) mean and unbiased std over time, exact two-pass
//   head     : workgroup = 128 frames x all N outputs; 4 waves, wave w owns frames [32w, 32w+32) and
//              keeps NT = ceil(N/32) accumulator tiles (7 for N = 201).  K is walked in chunks of 40
//              through LDS: normalised feature tile [128][41] and weight chunk [NT*32][41] (odd pitch ->
//              conflict-free ds_read_b32 operand fetches).
// Bound: HBM (4*F*(D + 2N) bytes per utterance, +4*F*N when `offset` is stored); the f32 MFMA rate
// (157 TF) puts the GEMM itself at about the same time, so the kernel is balanced, not MFMA-bound.
#include "common.h"
#include "prof.h"

namespace se {

using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case SE_ACT_RELU: return fmaxf(v, 0.f);
    case SE_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    case SE_ACT_GELU: return v * 0.5f * (1.0f + erff(v * 0.70710678118654752f));
    case SE_ACT_EXP: return expf(v);
    default: return v;
  }
}

// stats[(b*D + d)*2] = mean over time, [..+1] = 1 / (unbiased std + eps)
__global__ __launch_bounds__(256) void colstats_kernel(const float* __restrict__ feats, int F, int D, float eps,
                                                       float* __restrict__ stats) {
  __shared__ float red[4][64];
  const int b = blockIdx.y, d = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
  const bool ok = d < D;
  const float* base = feats + (size_t)b * F * D + d;
  float s = 0.f;
  if (ok)
    for (int t = ph; t < F; t += 4) s += base[(size_t)t * D];
  red[ph][threadIdx.x & 63] = s;
  __syncthreads();
  const float mean = (red[0][threadIdx.x & 63] + red[1][threadIdx.x & 63] + red[2][threadIdx.x & 63] + red[3][threadIdx.x & 63]) / (float)F;
  __syncthreads();
  float q = 0.f;
  if (ok)
    for (int t = ph; t < F; t += 4) {
      const float c = base[(size_t)t * D] - mean;
      q = fmaf(c, c, q);
    }
  red[ph][threadIdx.x & 63] = q;
  __syncthreads();
  if (ph == 0 && ok) {
    const float var = (red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]) / (float)(F - 1);
    stats[((size_t)b * D + d) * 2] = mean;
    stats[((size_t)b * D + d) * 2 + 1] = 1.0f / (sqrtf(var) + eps);
  }
}

constexpr int kHM = 128;     // frames per workgroup
constexpr int kHK = 40;      // K chunk
constexpr int kHP = kHK + 1; // LDS pitch (odd)

template <int NT>
__global__ __launch_bounds__(256) void head_kernel(const float* __restrict__ feats, const float* __restrict__ W,
                                                   const float* __restrict__ bias, const float* __restrict__ linears,
                                                   const float* __restrict__ stats, int rows, int F, int D, int N, int act,
                                                   float* __restrict__ predicted, float* __restrict__ offset) {
  __shared__ float As[kHM * kHP];
  __shared__ float Ws[NT * 32 * kHP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * kHM;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  for (int k0 = 0; k0 < D; k0 += kHK) {
    const int kc = min(kHK, D - k0);
    __syncthreads();
    // stage normalised features: item (r, kk)
    for (int it = tid; it < kHM * kHK; it += 256) {
      const int r = it / kHK, kk = it - r * kHK;
      const int row = row0 + r;
      float v = 0.f;
      if (row < rows && kk < kc) {
        v = feats[(size_t)row * D + k0 + kk];
        if (stats) {
          const int b = row / F;
          const float2 ms = *reinterpret_cast<const float2*>(stats + ((size_t)b * D + k0 + kk) * 2);
          v = (v - ms.x) * ms.y;
        }
      }
      As[r * kHP + kk] = v;
    }
    for (int it = tid; it < NT * 32 * kHK; it += 256) {
      const int n = it / kHK, kk = it - n * kHK;
      Ws[n * kHP + kk] = (n < N && kk < kc) ? W[(size_t)n * D + k0 + kk] : 0.f;
    }
    __syncthreads();
    const float* ap = As + (wave * 32 + (lane & 31)) * kHP + (lane >> 5);
    const float* wp = Ws + (lane & 31) * kHP + (lane >> 5);
#pragma unroll 4
    for (int kk = 0; kk < kHK; kk += 2) {
      const float a = ap[kk];
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wp[t * 32 * kHP + kk], acc[t], 0, 0, 0);
    }
  }

  // epilogue: C/D map of 32x32: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = t * 32 + (lane & 31);
    if (n >= N) continue;
    const float bn = bias ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row >= rows) continue;
      const float o = apply_act(acc[t][r] + bn, act);
      const size_t idx = (size_t)row * N + n;
      if (offset) offset[idx] = o;
      if (predicted) predicted[idx] = linears ? linears[idx] * o : o;
    }
  }
}

}  // namespace se

// internal (not part of the public header): column statistics shared with head_bwd.hip
extern "C" int se_head_colstats_f32(const float* feats, int B, int F, int D, float eps, float* stats, void* stream) {
  hipLaunchKernelGGL(se::colstats_kernel, dim3((D + 63) / 64, B), dim3(256), 0, se::as_stream(stream), feats, F, D, eps, stats);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" size_t se_head_workspace_bytes(int B, int F, int D, int N) {
  (void)F;
  // [stats: B*D*2 floats][bwd scratch: g_pre (B*F*N) is NOT kept here; see se_head_linear_bwd_f32]
  return (size_t)B * D * 2 * sizeof(float) + 256;
}

template <int NT>
static int launch_head(const float* feats, const float* W, const float* bias, const float* linears, const float* stats,
                       int rows, int F, int D, int N, int act, float* predicted, float* offset, hipStream_t st) {
  // algorithmic bytes: features + noisy power in, predicted + offset out
  se::ProfScope prof(se::kProfHead, 4.0 * rows * ((double)D + (linears ? N : 0) + (predicted ? N : 0) + (offset ? N : 0)), st);
  hipLaunchKernelGGL((se::head_kernel<NT>), dim3((rows + se::kHM - 1) / se::kHM), dim3(256), 0, st, feats, W, bias, linears,
                     stats, rows, F, D, N, act, predicted, offset);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_head_linear_f32(const float* feats, const float* W, const float* bias, const float* linears,
                                  int B, int F, int D, int N, int act, int cmvn, float eps,
                                  float* predicted, float* offset, void* workspace, size_t workspace_bytes, void* stream) {
  SE_REQUIRE(feats && W && (predicted || offset), "se_head_linear_f32: null argument");
  SE_REQUIRE(B > 0 && B <= 65535 && F >= 2 && D > 0 && N > 0 && N <= 256, "se_head_linear_f32: bad shape B=%d F=%d D=%d N=%d (N <= 256)", B, F, D, N);
  hipStream_t st = se::as_stream(stream);
  float* stats = nullptr;
  if (cmvn) {
    SE_REQUIRE(workspace && workspace_bytes >= se_head_workspace_bytes(B, F, D, N), "se_head_linear_f32: workspace too small");
    stats = reinterpret_cast<float*>(workspace);
    int rc = se_head_colstats_f32(feats, B, F, D, eps, stats, stream);
    if (rc) return rc;
  }
  const int rows = B * F;
  const int nt = (N + 31) / 32;
  switch (nt) {
    case 1: return launch_head<1>(feats, W, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
    case 2: return launch_head<2>(feats, W, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
    case 3: return launch_head<3>(feats, W, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
    case 4: return launch_head<4>(feats, W, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
    case 5: return launch_head<5>(feats, W, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
    case 6: return launch_head<6>(feats, W, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
    case 7: return launch_head<7>(feats, W, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
    default: return launch_head<8>(feats, W, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
  }
}
