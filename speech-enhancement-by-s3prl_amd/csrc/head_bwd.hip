// head_bwd.hip -- autograd of rows C1 / C2 wrt the head parameters (runner.py:459 loss.backward()):
//   g_pre = (grad_predicted (.) linears + grad_offset) (.) act'(offset);   gW[n][d] = sum_rows g_pre[row][n] xn[row][d];  gb[n] = sum_rows g_pre[row][n]
// (xn = CMVN-normalised features, recomputed from the forward's statistics; the normalisation has no parameters).
// A long-K reduction (K = B*F rows) done in exact fp32 on v_mfma_f32_32x32x2_f32: each workgroup reduces 256 rows
// into NT x 4 accumulator tiles (n-tiles x 128 feature dims) and adds them to gW with one fp32 atomic per element
// (256-B contiguous per wave-instruction).  Summation order across workgroups is not fixed (atomics).
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
  SE_HIP(hipMemsetAsync(gW, 0, sizeof(float) * (size_t)N * D, st));
  SE_HIP(hipMemsetAsync(gb, 0, sizeof(float) * (size_t)N, st));
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
