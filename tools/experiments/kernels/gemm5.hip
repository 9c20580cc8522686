// gemm5.hip -- 256 x 256 bf16 GEMM with ONE wave per SIMD: 4 waves x (128 x 128) wave tiles, v_mfma_f32_32x32x16_bf16, 256 accumulator
// registers per lane (C = epilogue(A . W^T + bias), bf16 output; contract: gemm.hip).  The follow-up DESIGN.md section 5 names:
// gemm3's 8 waves x (128 x 64) read 96 KiB of fragments per K-step from LDS, this shape 64 KiB, and the reads are software-pipelined
// between the MFMAs of the same wave instead of living in a partner wave's phase.
//
//   tile     : 256 x 256, K walked in 32-deep steps; 256 threads = 4 waves as 2 (M) x 2 (N); wave tile 128 x 128 = 4 x 4 MFMA tiles of
//              32 x 32, two 16-deep k-slices per step: 32 MFMAs = 1 024 matrix-pipe cycles per K-step
//   LDS      : gemm3's ring: 4 stages x (A 256 x 32 + W 256 x 32) bf16 = 4 x 32 KiB, 64-B rows, chunk ^ ((-(row >> 2)) & 3)
//              (conflict-free for the 32-row fragment reads as well: a 16-lane service group covers 4 row quads with 4 distinct XORs)
//   pipeline : K-step g multiplies from registers while (a) the 16 fragment reads of step g+1 and (b) the 8 LDS-DMA pieces of step g+4
//              are issued between its MFMAs (program order pinned with sched_barrier); one barrier per K-step; counted vmcnt
//   epilogue : swapped operands (C^T accumulators): lane = output row, 4 consecutive columns per accumulator quad; lanes l, l ^ 32 trade
//              quads so every lane stores 16 B; bias / GELU fused; compile-time specialised (identity / GELU, bf16 output)
#include <stdlib.h>
#include "common.h"
#include "bf16.h"
#include "prof.h"

namespace se {

constexpr int k5BM = 256, k5BN = 256, k5BK = 32, k5Threads = 256, k5Stages = 4;
constexpr int k5ABytes = k5BM * k5BK * 2, k5Stage = 2 * k5ABytes, k5Lds = k5Stages * k5Stage;     // 16 KiB, 32 KiB, 128 KiB

typedef __attribute__((address_space(3))) void* lds5_ptr_t;
typedef const __attribute__((address_space(1))) void* glb5_ptr_t;

__device__ __forceinline__ int swz5_f(int row) { return (-(row >> 2)) & 3; }
__device__ __forceinline__ int swz5(int row, int chunk) { return row * 64 + ((chunk ^ swz5_f(row)) << 4); }

template <int GELU>
__global__ __launch_bounds__(k5Threads) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm5_bf16_kernel(
    const uint16_t* __restrict__ A, int lda, const uint16_t* __restrict__ W, int ldw, const float* __restrict__ bias, int M, int N, int K,
    uint16_t* __restrict__ out, int ldc, int tiles_m, int tiles_n, int group_m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, hh = lane >> 5;

  const int nwg = tiles_m * tiles_n;
  int id;
  {
    const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  int tm, tn;
  {
    const int per_group = group_m * tiles_n, grp = id / per_group, first_m = grp * group_m;
    const int gsz = min(tiles_m - first_m, group_m), in = id - grp * per_group;
    tn = in / gsz;
    tm = first_m + (in - tn * gsz);
  }
  const int m0 = tm * k5BM, n0 = tn * k5BN;

  // ---- DMA sources: a stage half (A or W) = 16 chunks of 1 KiB = 16 rows x 64 B each; wave w issues chunks w, w+4, w+8, w+12 of A and of W.
  //      lane -> row 16 c + (lane >> 2); LDS position lane & 3 holds logical chunk (lane & 3) ^ f(row)
  const int r16 = lane >> 2, pos = lane & 3;
  const uint16_t* a_src[4];
  const uint16_t* b_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 16 * (4 * i + wave) + r16;
    const int lc = (pos ^ swz5_f(row)) << 3;
    a_src[i] = A + (size_t)min(m0 + row, M - 1) * lda + lc;
    b_src[i] = W + (size_t)min(n0 + row, N - 1) * ldw + lc;
  }
#define SE5_DMA_A(g, i) \
  __builtin_amdgcn_global_load_lds((glb5_ptr_t)(a_src[i] + (g) * k5BK), (lds5_ptr_t)(smem + ((g) & 3) * k5Stage + (4 * (i) + wave) * 1024), 16, 0, 0)
#define SE5_DMA_B(g, i) \
  __builtin_amdgcn_global_load_lds((glb5_ptr_t)(b_src[i] + (g) * k5BK), (lds5_ptr_t)(smem + ((g) & 3) * k5Stage + k5ABytes + (4 * (i) + wave) * 1024), 16, 0, 0)
#define SE5_DMA_STAGE(g)                          \
  do {                                            \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) { \
      SE5_DMA_A(g, i_);                           \
      SE5_DMA_B(g, i_);                           \
    }                                             \
  } while (0)

  // fragment byte offsets inside a stage: X (activation) rows wm * 128 + 32 i + l31, W rows wn * 128 + 32 j + l31; k-slice s: chunk 2 s + hh
  int x_off[4][2], w_off[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      x_off[i][s] = swz5(wm * 128 + 32 * i + l31, 2 * s + hh);
      w_off[i][s] = k5ABytes + swz5(wn * 128 + 32 * i + l31, 2 * s + hh);
    }

  f32x16 acc[4][4];                        // [j: W tile][i: X tile]; D^T: row = output column, col = output row
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  const int nk = K / k5BK;                 // >= 4 (launcher)
  SE5_DMA_STAGE(0);
  SE5_DMA_STAGE(1);
  SE5_DMA_STAGE(2);
  SE5_DMA_STAGE(3);
  asm volatile("s_waitcnt vmcnt(24)" ::: "memory");        // stage 0 landed (this wave's pieces)
  __builtin_amdgcn_s_barrier();

  bf16x8 xf[2][4][2], wf[2][4][2];         // [buffer][tile][k-slice]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      xf[0][i][s] = *reinterpret_cast<const bf16x8*>(smem + x_off[i][s]);
      wf[0][i][s] = *reinterpret_cast<const bf16x8*>(smem + w_off[i][s]);
    }

  // one K-step with register buffer CUR: MFMAs of step g from buffer CUR; reads of step g+1 into buffer CUR ^ 1; DMA of step g+4
#define SE5_STEP(CUR, STEADY)                                                                                              \
  {                                                                                                                        \
    /* stage g+1 must have landed everywhere; every wave's reads of stage g (issued during step g-1) must be complete */   \
    if (STEADY || g + 3 < nk) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");                                 \
    else if (g + 2 < nk) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");                                       \
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                       \
    __builtin_amdgcn_s_barrier();                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                                     \
    const char* sn = smem + ((g + 1) & 3) * k5Stage;                                                                       \
    const bool more = STEADY || g + 1 < nk, dma = STEADY || g + 4 < nk;      /* STEADY: no run-time branches between the MFMA groups */ \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                                        \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                      \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                      \
          acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[CUR][j][s], xf[CUR][i][s], acc[j][i], 0, 0, 0);           \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
        /* after each group of four MFMAs: two fragment reads of step g+1 and one DMA piece of step g+4 */                 \
        if (more) {                                                                                                        \
          xf[CUR ^ 1][j][s] = *reinterpret_cast<const bf16x8*>(sn + x_off[j][s]);                                          \
          wf[CUR ^ 1][j][s] = *reinterpret_cast<const bf16x8*>(sn + w_off[j][s]);                                          \
        }                                                                                                                  \
        if (dma) {                                                                                                         \
          if (s == 0) SE5_DMA_A(g + 4, j);                                                                                 \
          else SE5_DMA_B(g + 4, j);                                                                                        \
        }                                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                                 \
      }                                                                                                                    \
    }                                                                                                                      \
  }
  int g = 0;
  for (; g + 5 < nk; g += 2) {          // steady state: steps g and g+1 both refill the ring (g + 1 + 4 < nk)
    SE5_STEP(0, true)
    ++g;
    SE5_STEP(1, true)
    --g;
  }
  for (; g + 1 < nk; g += 2) {
    SE5_STEP(0, false)
    ++g;
    SE5_STEP(1, false)
    --g;
  }
  if (g < nk) SE5_STEP(0, false)
#undef SE5_STEP

  // ---- epilogue.  acc[j][i][r]: output row m = m0 + wm * 128 + 32 i + l31, column n = n0 + wn * 128 + 32 j + 8 (r >> 2) + 4 hh + (r & 3).
  //      Quad pairs (q, q + 1) = (r >> 2): lane hh = 0 keeps quad q and gets the partner's quad q (columns + 4..7), lane hh = 1 keeps quad
  //      q + 1 and gets the partner's: each lane then owns 8 consecutive columns = one 16-B store.
  const bool interior = (m0 + k5BM <= M);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float4 bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n0 + wn * 128 + 32 * j + 8 * q + 4 * hh;
      bq[q] = bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + wm * 128 + 32 * i + l31;
      uint2 pk[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v0 = acc[j][i][4 * q] + bq[q].x, v1 = acc[j][i][4 * q + 1] + bq[q].y, v2 = acc[j][i][4 * q + 2] + bq[q].z, v3 = acc[j][i][4 * q + 3] + bq[q].w;
        if (GELU) {
          const f32x2 ga = gelu_erf2((f32x2){v0, v1}), gb = gelu_erf2((f32x2){v2, v3});
          v0 = ga.x; v1 = ga.y; v2 = gb.x; v3 = gb.y;
        }
        pk[q] = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
      }
      uint16_t* orow = out + (size_t)min(m, M - 1) * ldc + n0 + wn * 128 + 32 * j;
#pragma unroll
      for (int p2 = 0; p2 < 2; ++p2) {
        const uint2 keep = hh ? pk[2 * p2 + 1] : pk[2 * p2];
        const uint2 send = hh ? pk[2 * p2] : pk[2 * p2 + 1];
        uint2 recv;
        recv.x = __shfl_xor(send.x, 32);
        recv.y = __shfl_xor(send.y, 32);
        // hh = 0: columns 8 (2 p2) + {0..3 own, 4..7 partner}; hh = 1: columns 8 (2 p2 + 1) + {0..3 partner, 4..7 own}
        const uint4 o16 = hh ? make_uint4(recv.x, recv.y, keep.x, keep.y) : make_uint4(keep.x, keep.y, recv.x, recv.y);
        if (interior || m < M) *reinterpret_cast<uint4*>(orow + 8 * (2 * p2 + hh)) = o16;
      }
    }
  }
}

}  // namespace se

// returns 1 if this kernel does not handle the call (caller falls back to gemm3), 0 on success, < 0 on error
extern "C" int se_gemm5_launch(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const float* residual_f32,
                               int M, int N, int K, int act, uint16_t* out_bf16, float* out_f32, int ldc, int vec_ok, void* stream) {
  if (!vec_ok || residual_f32 || out_f32 || !out_bf16 || N % se::k5BN != 0 || K % se::k5BK != 0 || K < 4 * se::k5BK || ldc % 8 != 0 ||
      (act != SE_ACT_IDENTITY && act != SE_ACT_GELU))
    return 1;
  static int group_m = 0;
  static bool attr_set = false;
  if (!attr_set) {
    const char* gm = getenv("SE_AMD_GEMM_GROUPM");
    group_m = gm ? atoi(gm) : 4;
    if (group_m < 1) group_m = 1;
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm5_bf16_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, se::k5Lds));
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm5_bf16_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, se::k5Lds));
    attr_set = true;
  }
  const int tiles_m = (M + se::k5BM - 1) / se::k5BM, tiles_n = N / se::k5BN;
  hipStream_t st = se::as_stream(stream);
  se::ProfScope prof(se::kProfGemm, 2.0 * M * (double)N * K, st);
  if (act == SE_ACT_GELU)
    hipLaunchKernelGGL((se::gemm5_bf16_kernel<1>), dim3(tiles_m * tiles_n), dim3(se::k5Threads), se::k5Lds, st, A, lda, W, ldw, bias, M, N, K, out_bf16,
                       ldc, tiles_m, tiles_n, group_m);
  else
    hipLaunchKernelGGL((se::gemm5_bf16_kernel<0>), dim3(tiles_m * tiles_n), dim3(se::k5Threads), se::k5Lds, st, A, lda, W, ldw, bias, M, N, K, out_bf16,
                       ldc, tiles_m, tiles_n, group_m);
  SE_LAUNCH_CHECK();
  return SE_OK;
}
