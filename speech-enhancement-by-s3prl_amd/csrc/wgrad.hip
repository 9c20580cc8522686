// wgrad.hip -- weight gradient dW[N,K] = dY[M,N]^T . X[M,K] straight from the ROW-MAJOR activations (row E2).
//
// The reduction dim is M = B*T (~32 000 rows), the slow index of both operands, so neither is in the K-contiguous layout
// the forward GEMM kernels stage.  Instead of transposing both operands through HBM (15 % of a fine-tune step), this kernel
// stages row-major [64 m][64 col] panels with LDS-DMA and takes the MFMA fragments out of them with the hardware-transposed
// LDS read (ds_read_b64_tr_b16), the way the flash-attention kernels read V^T / Q^T:
//   tile     : 256 (n) x 128 (k) outputs per 512-thread workgroup, 8 waves as 4 x 2, wave tile 64 x 64 = 2 x 2
//              v_mfma_f32_32x32x16_bf16 accumulators; 64 rows of m per stage
//   staging  : global_load_lds_dwordx4 into a 3-deep ring of 48 KiB stages = 4 dY panels + 2 X panels of [64 m][64 col]
//              bf16 (128-B rows, the kv_off swizzle of mhsa_tile.h: conflict-free for the tr reads); wave w loads rows
//              8w..8w+7 of every panel; rows past the split's end read a 16-B zero word instead (no tail pass, no padding)
//   fragments: per 16 rows of m, 2 + 2 fragments = 8 tr reads feed 4 MFMAs; both operands see the same permuted m order
//   split     : grid.y splits of the m range, each writing its own fp32 slab [N][K]; a slab reduce finishes (deterministic)
#include <stdlib.h>
#include <algorithm>
#include "common.h"
#include "prof.h"
#include "bf16.h"
#include "mhsa_tile.h"

namespace se {

constexpr int kWN = 256, kWK = 128, kWM = 64;
constexpr int kWPanel = kWM * 64 * 2;                       // 8 KiB
constexpr int kWStage = (kWN / 64 + kWK / 64) * kWPanel;    // 48 KiB
constexpr int kWStages = 3;
constexpr int kWLds = kWStages * kWStage;

__device__ uint4 g_wgrad_zero = {0u, 0u, 0u, 0u};

typedef __attribute__((address_space(3))) void* w_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* w_glb_ptr_t;
#define SE_WTR(ptr) __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(ptr))

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_tn_kernel(
    const uint16_t* __restrict__ dY, int ldy, const uint16_t* __restrict__ X, int ldx, int M, int N, int K, int m_per_split,
    int tiles_k, int splits_total, float* __restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wk = wave & 1;
  const int l31 = lane & 31, hh = lane >> 5;
  // XCD-aware work mapping: workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with its own 4 MB L2.  All
  // output tiles of ONE m-split re-read the same 64-row slices of dY and X, so a split is pinned to one XCD (split = xcd +
  // 8 q): its tiles run side by side there and every slice is fetched from HBM once per launch instead of once per tile.
  int tile, split;
  {
    const int lin = blockIdx.x, tiles = gridDim.x / splits_total;
    if ((splits_total & 7) == 0) {
      const int xcd = lin & 7, i = lin >> 3;
      split = xcd + 8 * (i / tiles);
      tile = i % tiles;
    } else {
      split = lin / tiles;
      tile = lin - split * tiles;
    }
  }
  const int tn = tile / tiles_k, tk = tile - tn * tiles_k;
  const int n0 = tn * kWN, k0 = tk * kWK;
  const int m_begin = split * m_per_split, m_end = min(M, m_begin + m_per_split);
  const int nt = m_end > m_begin ? (m_end - m_begin + kWM - 1) / kWM : 0;

  // ---- DMA sources: piece i of this wave = rows 8 wave .. +7 of panel i; lane -> row, LDS slot lane & 7 holds logical
  //      16-B chunk (lane & 7) ^ f(row) (the swizzle is applied on the source side: the DMA destination is lane-linear)
  const int row = 8 * wave + (lane >> 3);
  const int f = (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
  const int col8 = ((lane & 7) ^ f) * 8;
  const uint16_t* src[6];
  int ld[6];
  bool col_ok[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    if (i < 4) {
      const int c = n0 + 64 * i + col8;
      col_ok[i] = c < N;
      src[i] = dY + (col_ok[i] ? c : 0);
      ld[i] = ldy;
    } else {
      const int c = k0 + 64 * (i - 4) + col8;
      col_ok[i] = c < K;
      src[i] = X + (col_ok[i] ? c : 0);
      ld[i] = ldx;
    }
  }
  const uint16_t* zero = reinterpret_cast<const uint16_t*>(&g_wgrad_zero);
#define SEW_ISSUE(t, st)                                                                                        \
  do {                                                                                                          \
    const int m = m_begin + (t) * kWM + row;                                                                    \
    char* sb = smem + (st) * kWStage + wave * 1024;                                                             \
    _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                                             \
      const uint16_t* p = (m < m_end && col_ok[i]) ? src[i] + (size_t)m * ld[i] : zero;                         \
      __builtin_amdgcn_global_load_lds((w_glb_ptr_t)p, (w_lds_ptr_t)(sb + i * kWPanel), 16, 0, 0);              \
    }                                                                                                           \
  } while (0)

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transposed-read offsets inside a panel (see mhsa.hip): lane addresses row 4 hh + tq (+8), columns 32 blk + 16 g1 + 4 tp
  const int tq = (lane & 15) >> 2, tp = lane & 3, g1 = (lane >> 4) & 1;
  int toff[2][2];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {
    const int dcol = blk * 32 + 16 * g1 + 4 * tp;
    toff[blk][0] = kv_off(4 * hh + tq, dcol >> 3) + (dcol & 7) * 2;
    toff[blk][1] = kv_off(4 * hh + tq + 8, dcol >> 3) + (dcol & 7) * 2;
  }
  const int a_base = wn * kWPanel, b_base = (4 + wk) * kWPanel;

  if (nt > 0) SEW_ISSUE(0, 0);
  if (nt > 1) SEW_ISSUE(1, 1);
  int st = 0;
  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");      // this wave's pieces of tile t landed; t+1 stays in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();        // everyone's pieces landed AND everyone finished the stage refilled below
    if (t + 2 < nt) {
      const int st2 = (st + 2 >= kWStages) ? st + 2 - kWStages : st + 2;
      SEW_ISSUE(t + 2, st2);
    }
    const char* sb = smem + st * kWStage;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x8 af[2], bfr[2];
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        const bf16x4 alo = SE_WTR(sb + a_base + toff[blk][0] + s * 2048);
        const bf16x4 ahi = SE_WTR(sb + a_base + toff[blk][1] + s * 2048);
        const bf16x4 blo = SE_WTR(sb + b_base + toff[blk][0] + s * 2048);
        const bf16x4 bhi = SE_WTR(sb + b_base + toff[blk][1] + s * 2048);
        af[blk] = (bf16x8){alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
        bfr[blk] = (bf16x8){blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    st = (st + 1 == kWStages) ? 0 : st + 1;
  }
#undef SEW_ISSUE

  // ---- epilogue: acc[i][j][r] = dW[n0 + 64 wn + 32 i + (r&3) + 8 (r>>2) + 4 hh][k0 + 64 wk + 32 j + l31]
  float* out = partials + (size_t)split * N * K;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kc = k0 + 64 * wk + 32 * j + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + 64 * wn + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hh;
        if (n < N && kc < K) out[(size_t)n * K + kc] = acc[i][j][r];
      }
    }
}

__global__ __launch_bounds__(256) void wgrad_slab_reduce_kernel(const float* __restrict__ partials, int splits, size_t n4, int accumulate,
                                                                float* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 s = accumulate ? reinterpret_cast<const float4*>(out)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < splits; ++k) {
      const float4 p = reinterpret_cast<const float4*>(partials)[(size_t)k * n4 + i];
      s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
    }
    reinterpret_cast<float4*>(out)[i] = s;
  }
}

}  // namespace se

// dW[N,K] (+)= dY[M,N]^T . X[M,K]; dY / X row-major bf16 (ldy, ldx in elements, multiples of 8; N, K multiples of 8);
// workspace >= splits * N * K floats.
extern "C" int se_wgrad_tn_bf16(const uint16_t* dY, int ldy, const uint16_t* X, int ldx, int M, int N, int K, int splits, float* dW,
                                int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  SE_REQUIRE(dY && X && dW && workspace, "se_wgrad_tn_bf16: null argument");
  SE_REQUIRE(M > 0 && N > 0 && K > 0 && splits >= 1 && splits <= 65535, "se_wgrad_tn_bf16: bad shape");
  SE_REQUIRE(N % 8 == 0 && K % 8 == 0 && ldy % 8 == 0 && ldx % 8 == 0 && ldy >= N && ldx >= K,
             "se_wgrad_tn_bf16: N, K and the leading dimensions must be multiples of 8");
  SE_REQUIRE((((uintptr_t)dY | (uintptr_t)X | (uintptr_t)workspace | (uintptr_t)dW) % 16) == 0, "se_wgrad_tn_bf16: operands must be 16-B aligned");
  SE_REQUIRE(workspace_bytes >= (size_t)splits * N * K * sizeof(float), "se_wgrad_tn_bf16: workspace too small");
  hipStream_t st = se::as_stream(stream);
  static bool attr_set = false;
  if (!attr_set) {
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::wgrad_tn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, se::kWLds));
    attr_set = true;
  }
  const int tiles_n = (N + se::kWN - 1) / se::kWN, tiles_k = (K + se::kWK - 1) / se::kWK;
  const int tiles_m = (M + se::kWM - 1) / se::kWM;
  const int m_per_split = (tiles_m + splits - 1) / splits * se::kWM;
  float* partials = reinterpret_cast<float*>(workspace);
  {
    se::ProfScope prof(se::kProfGemm, 2.0 * M * (double)N * K, st);
    hipLaunchKernelGGL(se::wgrad_tn_kernel, dim3(tiles_n * tiles_k * splits), dim3(512), se::kWLds, st, dY, ldy, X, ldx, M, N, K, m_per_split,
                       tiles_k, splits, partials);
    SE_LAUNCH_CHECK();
  }
  const size_t n4 = (size_t)N * K / 4;
  hipLaunchKernelGGL(se::wgrad_slab_reduce_kernel, dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 4096)), dim3(256), 0, st, partials, splits,
                     n4, accumulate, dW);
  SE_LAUNCH_CHECK();
  return SE_OK;
}
