// head_bwd.hip -- autograd of rows C1 / C2 wrt the head parameters (runner.py:459 loss.backward()):
//   g_pre = (grad_predicted (.) linears + grad_offset) (.) act'(offset);   gW[n][d] = sum_rows g_pre[row][n] xn[row][d];  gb[n] = sum_rows g_pre[row][n]
// (xn = CMVN-normalised features, recomputed from the forward's statistics; the normalisation has no parameters).
// A long-K reduction (K = B*F rows) done in exact fp32 on v_mfma_f32_32x32x2_f32: each workgroup reduces 256 rows
// into NT x 4 accumulator tiles (n-tiles x 128 feature dims) and adds them to gW with one fp32 atomic per element
// (256-B contiguous per wave-instruction).  Summation order across workgroups is not fixed (atomics).
#include <algorithm>
#include "common.h"

namespace se {

using f32x16b = __attribute__((ext_vector_type(16))) float;
constexpr int kBR = 256;     // rows per workgroup
constexpr int kBS = 32;      // rows per LDS stage
constexpr int kBD = 128;     // feature dims per workgroup (grid.y chunks)

__device__ __forceinline__ float act_grad_from_out(float o, int act) {
  switch (act) {
    case SE_ACT_RELU: return o > 0.f ? 1.f : 0.f;
    case SE_ACT_SIGMOID: return o * (1.f - o);
    case SE_ACT_EXP: return o;
    default: return 1.f;        // identity (GELU is not invertible from its output; rejected on the host)
  }
}

template <int NT>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ feats, const float* __restrict__ linears,
                                                       const float* __restrict__ offset, const float* __restrict__ gp,
                                                       const float* __restrict__ goff, const float* __restrict__ stats, int rows, int F, int D, int N, int act,
                                                       float* __restrict__ gW, float* __restrict__ gb) {
  constexpr int GP = NT * 32 + 1;     // odd pitches
  constexpr int XP = kBD + 1;
  __shared__ float Gs[kBS * GP];
  __shared__ float Xs[kBS * XP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * kBR, d0 = blockIdx.y * kBD;
  constexpr int TILES = NT * 4, PER_WAVE = (TILES + 3) / 4;
  f32x16b acc[PER_WAVE];
#pragma unroll
  for (int t = 0; t < PER_WAVE; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float gb_acc = 0.f;      // thread tid < NT*32 owns output n = tid

  for (int s0 = 0; s0 < kBR; s0 += kBS) {
    __syncthreads();
    for (int it = tid; it < kBS * NT * 32; it += 256) {
      const int r = it / (NT * 32), n = it - r * (NT * 32);
      const int row = row0 + s0 + r;
      float g = 0.f;
      if (row < rows && n < N) {
        const size_t idx = (size_t)row * N + n;
        g = gp ? gp[idx] : 0.f;                      // d loss / d predicted, predicted = linears (.) offset
        if (linears) g *= linears[idx];
        if (goff) g += goff[idx];                      // d loss / d offset directly (WSD scores the mask itself)
        g *= act_grad_from_out(offset[idx], act);
      }
      Gs[r * GP + n] = g;
    }
    for (int it = tid; it < kBS * kBD; it += 256) {
      const int r = it / kBD, dd = it - r * kBD;
      const int row = row0 + s0 + r, d = d0 + dd;
      float v = 0.f;
      if (row < rows && d < D) {
        v = feats[(size_t)row * D + d];
        if (stats) {
          const int b = row / F;
          v = (v - stats[((size_t)b * D + d) * 2]) * stats[((size_t)b * D + d) * 2 + 1];
        }
      }
      Xs[r * XP + dd] = v;
    }
    __syncthreads();
    if (blockIdx.y == 0 && tid < NT * 32) {
#pragma unroll 8
      for (int r = 0; r < kBS; ++r) gb_acc += Gs[r * GP + tid];
    }
    // A[i = n][k = r] = G[r][n];  B[k = r][j = d] = X[r][d]
#pragma unroll 2
    for (int k = 0; k < kBS; k += 2) {
      const int r = k + (lane >> 5);
#pragma unroll
      for (int t = 0; t < PER_WAVE; ++t) {
        const int tile = wave + 4 * t;
        if (tile < TILES) {
          const int nt = tile >> 2, dt = tile & 3;
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(Gs[r * GP + nt * 32 + (lane & 31)], Xs[r * XP + dt * 32 + (lane & 31)], acc[t], 0, 0, 0);
        }
      }
    }
  }
  // C/D: col = lane & 31 -> d ; row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) -> n
#pragma unroll
  for (int t = 0; t < PER_WAVE; ++t) {
    const int tile = wave + 4 * t;
    if (tile >= TILES) continue;
    const int nt = tile >> 2, dt = tile & 3;
    const int d = d0 + dt * 32 + (lane & 31);
    if (d >= D) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = nt * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (n < N) atomicAdd(&gW[(size_t)n * D + d], acc[t][r]);
    }
  }
  if (blockIdx.y == 0 && tid < NT * 32 && tid < N) atomicAdd(&gb[tid], gb_acc);
}

// g_pre = (grad_predicted (.) linears + grad_offset) (.) act'(offset) as a bf16 GEMM operand (M, ldp), columns N.. zeroed
__global__ __launch_bounds__(256) void head_gpre_kernel(const float* __restrict__ linears, const float* __restrict__ offset,
                                                        const float* __restrict__ gp, const float* __restrict__ goff, size_t rows, int N, int ldp,
                                                        int act, uint16_t* __restrict__ out) {
  const size_t n = rows * (size_t)ldp;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = i / ldp;
    const int c = (int)(i - r * ldp);
    float g = 0.f;
    if (c < N) {
      const size_t idx = r * N + c;
      g = gp ? gp[idx] : 0.f;
      if (linears) g *= linears[idx];
      if (goff) g += goff[idx];
      g *= act_grad_from_out(offset[idx], act);
    }
    __bf16 h = (__bf16)g;
    out[i] = __builtin_bit_cast(uint16_t, h);
  }
}

// backward of the CMVN over time y = (x - mean_t x) / (std_t x + eps) (unbiased std), in place on g (B, F, D):
//   dx = r (g - mean_t g) - (x - mu) r^2 / (s (F - 1)) sum_t g_t (x_t - mu),   r = 1 / (s + eps)  [stats = (mu, r) per (b, d)]
__global__ __launch_bounds__(256) void cmvn_time_bwd_kernel(const float* __restrict__ x, const float* __restrict__ stats, int F, int D, float eps,
                                                            float* __restrict__ g) {
  __shared__ float red[2][4][64];
  const int b = blockIdx.y, d = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
  const bool ok = d < D;
  const float mu = ok ? stats[((size_t)b * D + d) * 2] : 0.f, r = ok ? stats[((size_t)b * D + d) * 2 + 1] : 0.f;
  const size_t base = (size_t)b * F * D + d;
  float s0 = 0.f, s1 = 0.f;
  if (ok)
    for (int t = rg; t < F; t += 4) {
      const float gv = g[base + (size_t)t * D];
      s0 += gv;
      s1 += gv * (x[base + (size_t)t * D] - mu);
    }
  red[0][rg][threadIdx.x & 63] = s0;
  red[1][rg][threadIdx.x & 63] = s1;
  __syncthreads();
  const int c = threadIdx.x & 63;
  const float sg = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
  const float sgx = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
  if (!ok) return;
  const float sd = 1.f / r - eps;                       // the standard deviation itself
  const float k2 = sd > 0.f ? r * r / (sd * (float)(F - 1)) * sgx : 0.f;
  const float mg = sg / (float)F;
  for (int t = rg; t < F; t += 4) {
    const size_t o = base + (size_t)t * D;
    g[o] = r * (g[o] - mg) - (x[o] - mu) * k2;
  }
}

}  // namespace se

// defined in head.hip
extern "C" int se_head_colstats_f32(const float* feats, int B, int F, int D, float eps, float* stats, void* stream);

template <int NT>
static int launch_bwd(const float* feats, const float* linears, const float* offset, const float* gp, const float* goff, const float* stats,
                      int rows, int F, int D, int N, int act, float* gW, float* gb, hipStream_t st) {
  dim3 grid((rows + se::kBR - 1) / se::kBR, (D + se::kBD - 1) / se::kBD);
  hipLaunchKernelGGL((se::head_bwd_kernel<NT>), grid, dim3(256), 0, st, feats, linears, offset, gp, goff, stats, rows, F, D, N, act, gW, gb);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_head_linear_bwd_f32(const float* feats, const float* linears, const float* offset, const float* grad_predicted,
                                      const float* grad_offset, int B, int F, int D, int N, int act, int cmvn, float eps,
                                      float* gW, float* gb, void* workspace, size_t workspace_bytes, void* stream) {
  SE_REQUIRE(feats && offset && (grad_predicted || grad_offset) && gW && gb, "se_head_linear_bwd_f32: null argument");
  SE_REQUIRE(B > 0 && F >= 2 && D > 0 && N > 0 && N <= 256, "se_head_linear_bwd_f32: bad shape (N <= 256)");
  SE_REQUIRE(act == SE_ACT_IDENTITY || act == SE_ACT_RELU || act == SE_ACT_SIGMOID || act == SE_ACT_EXP,
             "se_head_linear_bwd_f32: activation %d has no output-only derivative", act);
  hipStream_t st = se::as_stream(stream);
  { const int zrc_ = se::zero_async(gW, sizeof(float) * (size_t)N * D, st); if (zrc_) return zrc_; }
  { const int zrc_ = se::zero_async(gb, sizeof(float) * (size_t)N, st); if (zrc_) return zrc_; }
  float* stats = nullptr;
  if (cmvn) {
    SE_REQUIRE(workspace && workspace_bytes >= se_head_workspace_bytes(B, F, D, N), "se_head_linear_bwd_f32: workspace too small");
    stats = reinterpret_cast<float*>(workspace);
    int rc = se_head_colstats_f32(feats, B, F, D, eps, stats, stream);
    if (rc) return rc;
  }
  const int rows = B * F;
  switch ((N + 31) / 32) {
    case 1: return launch_bwd<1>(feats, linears, offset, grad_predicted, grad_offset, stats, rows, F, D, N, act, gW, gb, st);
    case 2: return launch_bwd<2>(feats, linears, offset, grad_predicted, grad_offset, stats, rows, F, D, N, act, gW, gb, st);
    case 3: return launch_bwd<3>(feats, linears, offset, grad_predicted, grad_offset, stats, rows, F, D, N, act, gW, gb, st);
    case 4: return launch_bwd<4>(feats, linears, offset, grad_predicted, grad_offset, stats, rows, F, D, N, act, gW, gb, st);
    case 5: return launch_bwd<5>(feats, linears, offset, grad_predicted, grad_offset, stats, rows, F, D, N, act, gW, gb, st);
    case 6: return launch_bwd<6>(feats, linears, offset, grad_predicted, grad_offset, stats, rows, F, D, N, act, gW, gb, st);
    case 7: return launch_bwd<7>(feats, linears, offset, grad_predicted, grad_offset, stats, rows, F, D, N, act, gW, gb, st);
    default: return launch_bwd<8>(feats, linears, offset, grad_predicted, grad_offset, stats, rows, F, D, N, act, gW, gb, st);
  }
}

extern "C" size_t se_head_dx_workspace_bytes(int B, int F, int D, int N) {
  const size_t M = (size_t)B * F, NP = ((size_t)N + 63) / 64 * 64;
  return se_head_workspace_bytes(B, F, D, N) + M * NP * 2 + 256 + (size_t)D * NP * 2 + 256;
}

// d loss / d features of rows C1 / C2 (needed when the head sits on top of a trainable stack: the Residual head's LSTM):
//   dx = CMVN'( g_pre . W ),   g_pre as in se_head_linear_bwd_f32.  The product runs on the bf16 GEMM.
extern "C" int se_head_linear_dx_f32(const float* feats, const float* linears, const float* offset, const float* grad_predicted,
                                     const float* grad_offset, const float* W, int B, int F, int D, int N, int act, int cmvn, float eps,
                                     float* dx, void* workspace, size_t workspace_bytes, void* stream) {
  SE_REQUIRE(feats && offset && (grad_predicted || grad_offset) && W && dx && workspace, "se_head_linear_dx_f32: null argument");
  SE_REQUIRE(B > 0 && B <= 65535 && F >= 2 && D > 0 && N > 0 && N <= 256, "se_head_linear_dx_f32: bad shape (N <= 256)");
  SE_REQUIRE(workspace_bytes >= se_head_dx_workspace_bytes(B, F, D, N) && (uintptr_t)workspace % 256 == 0, "se_head_linear_dx_f32: workspace too small / unaligned");
  SE_REQUIRE(act == SE_ACT_IDENTITY || act == SE_ACT_RELU || act == SE_ACT_SIGMOID || act == SE_ACT_EXP,
             "se_head_linear_dx_f32: activation %d has no output-only derivative", act);
  hipStream_t st = se::as_stream(stream);
  const size_t M = (size_t)B * F;
  const int NP = (N + 63) / 64 * 64;
  char* base = reinterpret_cast<char*>(workspace);
  float* stats = reinterpret_cast<float*>(base);
  size_t off = (se_head_workspace_bytes(B, F, D, N) + 255) & ~(size_t)255;
  uint16_t* gpre = reinterpret_cast<uint16_t*>(base + off);
  off += (M * NP * 2 + 255) & ~(size_t)255;
  uint16_t* wt = reinterpret_cast<uint16_t*>(base + off);
  hipLaunchKernelGGL(se::head_gpre_kernel, dim3((unsigned)std::min<size_t>((M * NP + 255) / 256, 8192)), dim3(256), 0, st, linears, offset,
                     grad_predicted, grad_offset, M, N, NP, act, gpre);
  SE_LAUNCH_CHECK();
  int rc = se_transpose_f32_bf16(W, N, D, D, wt, NP, stream);          // (D, NP) = W^T, zero padded
  if (rc) return rc;
  rc = se_gemm_bf16(gpre, NP, wt, NP, nullptr, nullptr, (int)M, D, NP, SE_ACT_IDENTITY, nullptr, dx, D, stream);
  if (rc) return rc;
  if (cmvn) {
    rc = se_head_colstats_f32(feats, B, F, D, eps, stats, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(se::cmvn_time_bwd_kernel, dim3((D + 63) / 64, B), dim3(256), 0, st, feats, stats, F, D, eps, dx);
    SE_LAUNCH_CHECK();
  }
  return SE_OK;
}
