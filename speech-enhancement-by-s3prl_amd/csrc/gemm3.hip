// gemm3.hip -- 256 x 256 bf16 GEMM for the encoder's large projections (C = epilogue(A . W^T + bias); contract: gemm.hip)
//
// Why this shape: in-kernel s_memtime stamps of the 256 x 128 kernel (gemm2.hip) showed the operand feed, not the
// matrix pipe, setting the pace: the L1/TA path delivers ~18-20 B/clk/CU into LDS, so a tile needs >= ~128 flop per
// loaded byte (256 x 256) to keep the MFMAs fed; and every LDS-DMA instruction stalls its wave 60-100 cycles, so DMA
// issue must sit in a wave's LDS-read phase (overlapped by its SIMD partner's MFMA phase), never in its MFMA phase.
//
//   tile     : 256 x 256, K walked in 32-deep steps; 512 threads = 8 waves as 2 (M) x 4 (N), wave tile 128 x 64
//              = 8 x 4 v_mfma_f32_16x16x32_bf16 tiles, 128 accumulator registers; one workgroup per CU
//   LDS      : ring of 4 stages x (A 256 x 32 + B 256 x 32) bf16 = 4 x 32 KiB; rows are 64 B, 16-B chunk index XOR-ed
//              with (-(row >> 2)) & 3 (on the DMA source address and again on the ds_read_b128 side): conflict-free
//   staging  : global_load_lds_dwordx4, 4 per wave per K-step, K-steps g+1 .. g+3 in flight while g is multiplied;
//              counted s_waitcnt vmcnt, raw s_barrier only
//   schedule : ping-pong -- waves 4-7 run one barrier behind waves 0-3; per K-step two {LDS reads | barrier | 16 MFMAs |
//              barrier} pairs, so on every SIMD one wave multiplies while its partner reads / issues DMA
//   epilogue : swapped operands (C^T accumulators): 4 consecutive output columns per lane, stored straight from
//              registers; bias / GELU / fp32 residual fused; compile-time specialised
#include <stdlib.h>
#include "common.h"
#include "bf16.h"
#include "prof.h"

// The raised wave priority around the MFMA clusters (s_setprio 1 ... 0) is OFF: A/B on one box, fine-tune step 19.73 -> 19.52 ms without it
// (the same finding as for the row-complete and the persistent kernels).  SE_AMD_EXTRA_DEFINES=-DSE_AMD_SETPRIO python build.py --force brings it back.
#ifdef SE_AMD_SETPRIO
#define SE_SETPRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define SE_SETPRIO(x) ((void)0)
#endif

namespace se {

constexpr int k3BM = 256, k3BN = 256, k3BK = 32, k3Threads = 512, k3Stages = 4;
constexpr int k3ABytes = k3BM * k3BK * 2, k3Stage = 2 * k3ABytes, k3Lds = k3Stages * k3Stage;     // 16 KiB, 32 KiB, 128 KiB

typedef __attribute__((address_space(3))) void* lds3_ptr_t;
typedef const __attribute__((address_space(1))) void* glb3_ptr_t;

// 16-B chunk swizzle for 64-B rows: chunk' = chunk ^ f(row), f = (-(row >> 2)) & 3.  ds_read_b128 is serviced in
// 16-lane groups {0-3, 12-15, 20-27}, ... i.e. rows {0-3, 12-15} at chunk c TOGETHER WITH rows {4-11} at chunk c^1;
// f must make those 16 addresses hit 16 distinct 16-B slots of the 256-B bank row (the obvious (row >> 2) & 3 is 2-way).
__device__ __forceinline__ int swz3_f(int row) { return (-(row >> 2)) & 3; }
__device__ __forceinline__ int swz3(int row, int chunk) { return row * 64 + ((chunk ^ swz3_f(row)) << 4); }

__device__ __forceinline__ float act3(float v, int act) {
  if (act == SE_ACT_GELU) return gelu_erf(v);
  if (act == SE_ACT_RELU) return fmaxf(v, 0.f);
  if (act == SE_ACT_EXP) return __expf(v);
  if (act == SE_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
  return v;
}

// diagnostic stamps (dbg bit 4, SE_AMD_GEMM_DBG=16): lane 0 of every wave of workgroups 0..7 appends s_memtime values to the
// buffer passed as `residual` (timing-only run; results are garbage)
__device__ __forceinline__ void stamp3(unsigned long long* buf, int& idx, bool on) {
#ifdef SE_AMD_STAMPS
  if (on) buf[idx++] = __builtin_amdgcn_s_memtime();
#endif
}

// ACT: compile-time activation; EF bit 0: fp32 residual, bit 1: bf16 output, bit 2: fp32 output (N % 4 == 0, 16-B rows)
// SCHED 0: two {reads | barrier | 16 MFMAs | barrier} pairs per K-step; SCHED 1: one {12 reads + DMA issue + waits | barrier |
// 32 MFMAs | barrier} pair per K-step (half the barriers; the LDS latency is absorbed in the read phase)
template <int ACT, int EF, int SCHED>
__global__ __launch_bounds__(k3Threads) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm3_bf16_kernel(
    const uint16_t* __restrict__ A, int lda, const uint16_t* __restrict__ W, int ldw, const float* __restrict__ bias,
    const float* __restrict__ residual, int M, int N, int K, uint16_t* __restrict__ out_bf16, float* __restrict__ out_f32,
    int ldc, int tiles_m, int tiles_n, int dbg, int group_m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  const int nwg = tiles_m * tiles_n;
  int id;
  {
    const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  // grouped order inside the XCD's contiguous id range: GROUP_M tile rows form a super-tile that is swept n-major with
  // m fastest, so the ~32 workgroups in flight on an XCD share few A panels AND few weight-row slices (both must fit 4 MB)
  int tm, tn;
  {
    const int per_group = group_m * tiles_n, grp = id / per_group, first_m = grp * group_m;
    const int gsz = min(tiles_m - first_m, group_m), in = id - grp * per_group;
    tn = in / gsz;
    tm = first_m + (in - tn * gsz);
  }
  const int m0 = tm * k3BM, n0 = tn * k3BN;

  // ---- DMA sources: a stage half (A or B) = 16 chunks of 1 KiB = 16 rows x 64 B each; wave w issues chunks w, w+8 of A and of B.
  //      lane -> row 16 c + (lane >> 2); LDS position lane & 3 holds logical chunk (lane & 3) ^ ((row >> 2) & 3)
  const int r16 = lane >> 2, pos = lane & 3;
  const uint16_t* a_src[2];
  const uint16_t* b_src[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 16 * (i * 8 + wave) + r16;
    const int lc = (pos ^ swz3_f(row)) << 3;
    a_src[i] = A + (size_t)min(m0 + row, M - 1) * lda + lc;
    b_src[i] = W + (size_t)min(n0 + row, N - 1) * ldw + lc;
  }
#define SE3_ISSUE(g, st)                                                                                                   \
  do {                                                                                                                     \
    char* sb = smem + (st) * k3Stage + wave * 1024;                                                                        \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                        \
      __builtin_amdgcn_global_load_lds((glb3_ptr_t)(a_src[i] + (g) * k3BK), (lds3_ptr_t)(sb + i * 8192), 16, 0, 0);        \
      __builtin_amdgcn_global_load_lds((glb3_ptr_t)(b_src[i] + (g) * k3BK), (lds3_ptr_t)(sb + k3ABytes + i * 8192), 16, 0, 0); \
    }                                                                                                                      \
  } while (0)

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment byte offsets inside a stage: lane -> row (lane & 15) of the 16-row tile, logical chunk lane >> 4
  const int frow = lane & 15, fch = lane >> 4;
  int a_off[8], b_off[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) a_off[i] = swz3(wr * 128 + i * 16 + frow, fch);
#pragma unroll
  for (int j = 0; j < 4; ++j) b_off[j] = k3ABytes + swz3(wc * 64 + j * 16 + frow, fch);

  const int nk = K / k3BK;                 // >= 3 (launcher)
  SE3_ISSUE(0, 0);
  SE3_ISSUE(1, 1);
  SE3_ISSUE(2, 2);
  const bool late = wave >= 4;
  const bool st_on = (dbg & 16) && lane == 0 && blockIdx.x < 8;
  unsigned long long* st_buf = reinterpret_cast<unsigned long long*>(const_cast<float*>(residual)) + ((size_t)blockIdx.x * 8 + wave) * 256;
  int st_i = 0;
  stamp3(st_buf, st_i, st_on);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // K-step 0 landed (this wave's pieces)
  __builtin_amdgcn_s_barrier();
  if (late) __builtin_amdgcn_s_barrier();                  // stagger

  if constexpr (SCHED == 0) {
  int st = 0;
  stamp3(st_buf, st_i, st_on);
  for (int g = 0; g < nk; ++g) {
    const char* sb = smem + st * k3Stage;
    bf16x8 af[4], bfr[4];
    // ---------------- half 0: m-tiles 0-3
#pragma unroll
    for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sb + b_off[j]);
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sb + a_off[i]);
    __builtin_amdgcn_sched_barrier(0);
    stamp3(st_buf, st_i, st_on && g < 12);
    __builtin_amdgcn_s_barrier();
    stamp3(st_buf, st_i, st_on && g < 12);
    __builtin_amdgcn_sched_barrier(0);
    SE_SETPRIO(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    SE_SETPRIO(0);
    __builtin_amdgcn_sched_barrier(0);
    stamp3(st_buf, st_i, st_on && g < 12);
    __builtin_amdgcn_s_barrier();
    stamp3(st_buf, st_i, st_on && g < 12);
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- half 1: m-tiles 4-7 (B fragments stay in registers)
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sb + a_off[4 + i]);
    // Ring slot (g+3)%4 == (g-1)%4: its last readers (waves 4-7, half 1 of step g-1) finished before the barrier this
    // wave passed after its half-0 reads above.  Issued HERE (read phase) so the DMA issue stalls overlap the SIMD
    // partner's MFMA phase.
    stamp3(st_buf, st_i, st_on && g < 12);
    if (g + 3 < nk) {
      SE3_ISSUE(g + 3, (st + 3) & 3);
      stamp3(st_buf, st_i, st_on && g < 12);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // K-step g+1 landed; g+2, g+3 (8 DMAs) stay in flight
    } else if (g + 2 < nk) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    stamp3(st_buf, st_i, st_on && g < 12);
    __builtin_amdgcn_s_barrier();
    stamp3(st_buf, st_i, st_on && g < 12);
    __builtin_amdgcn_sched_barrier(0);
    SE_SETPRIO(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[4 + i][j], 0, 0, 0);
    SE_SETPRIO(0);
    __builtin_amdgcn_sched_barrier(0);
    stamp3(st_buf, st_i, st_on && g < 12);
    __builtin_amdgcn_s_barrier();
    stamp3(st_buf, st_i, st_on && g < 12);
    __builtin_amdgcn_sched_barrier(0);
    st = (st + 1) & 3;
  }
  } else {
  int st = 0;
  for (int g = 0; g < nk; ++g) {
    const char* sb = smem + st * k3Stage;
    bf16x8 af[8], bfr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sb + b_off[j]);
#pragma unroll
    for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sb + a_off[i]);
    // ring slot (g+3)%4 == (g-1)%4: every wave finished (lgkmcnt(0) below) its reads of step g-1 before the barrier
    // that ended ITS read phase, and this wave has passed that barrier -> safe to refill here
    if (g + 3 < nk) {
      SE3_ISSUE(g + 3, (st + 3) & 3);
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");     // K-step g+1 landed; fragments of step g in registers
    } else if (g + 2 < nk) {
      asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    SE_SETPRIO(1);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    SE_SETPRIO(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    st = (st + 1) & 3;
  }
  }
  if (!late) __builtin_amdgcn_s_barrier();                 // re-align the two groups (barrier counts must match)

  // ---- epilogue: C^T accumulators: col = lane & 15 -> output row, row = 4 (lane >> 4) + r -> 4 consecutive columns
  constexpr bool RES = EF & 1, OBF = EF & 2, OF32 = EF & 4;
  const int mrow = lane & 15, ncol = 4 * (lane >> 4);
  float4 bb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int gn = min(n0 + wc * 64 + j * 16 + ncol, N - 4);
    bb[j] = bias ? *reinterpret_cast<const float4*>(bias + gn) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // wave-uniform; the interior path's bf16 stores are 16 B wide
  const bool interior = (m0 + k3BM <= M) && (n0 + k3BN <= N) && (!OBF || ((ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(out_bf16) & 15) == 0));
#define SE3_EPILOGUE_BODY(PRED)                                                                                            \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                          \
    const int gm = m0 + wr * 128 + i * 16 + mrow;                                                                          \
    const bool mok = gm < M;                                                                                               \
    const size_t orow = (size_t)min(gm, M - 1) * ldc;                                                                      \
    float4 rr[4];                                                                                                          \
    if constexpr (RES) {                                                                                                   \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                      \
        const int gn = min(n0 + wc * 64 + j * 16 + ncol, N - 4);                                                           \
        rr[j] = *reinterpret_cast<const float4*>(residual + orow + gn);                                                    \
      }                                                                                                                    \
    }                                                                                                                      \
    uint2 pk[4];                                                                                                           \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                        \
      const int gn = n0 + wc * 64 + j * 16 + ncol;                                                                         \
      float v0 = acc[i][j][0] + bb[j].x, v1 = acc[i][j][1] + bb[j].y, v2 = acc[i][j][2] + bb[j].z, v3 = acc[i][j][3] + bb[j].w; \
      if constexpr (ACT == SE_ACT_GELU) {                                                                                  \
        const f32x2 ga = gelu_erf2((f32x2){v0, v1}), gb = gelu_erf2((f32x2){v2, v3});                                      \
        v0 = ga.x; v1 = ga.y; v2 = gb.x; v3 = gb.y;                                                                        \
      } else {                                                                                                             \
        v0 = act3(v0, ACT); v1 = act3(v1, ACT); v2 = act3(v2, ACT); v3 = act3(v3, ACT);                                    \
      }                                                                                                                    \
      if constexpr (RES) { v0 += rr[j].x; v1 += rr[j].y; v2 += rr[j].z; v3 += rr[j].w; }                                   \
      pk[j] = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));                                                        \
      if (!(PRED) || (mok && gn < N)) {                                                                                    \
        if constexpr (OF32) *reinterpret_cast<float4*>(out_f32 + orow + gn) = make_float4(v0, v1, v2, v3);                 \
        if constexpr (OBF) {                                                                                               \
          if (PRED) *reinterpret_cast<uint2*>(out_bf16 + orow + gn) = pk[j];                                               \
        }                                                                                                                  \
      }                                                                                                                    \
    }                                                                                                                      \
    if constexpr (OBF) {                                                                                                   \
      if (!(PRED)) {                                                                                                       \
        /* 16-B stores: lane pairs (g, g ^ 1) = lanes l, l ^ 16 trade 4-column pieces of two neighbouring MFMA tiles, so each   \
           lane owns 8 consecutive bf16 columns and a wave instruction writes 16 rows x 64 contiguous bytes instead of      \
           16 rows x 32: 5.2 vs 3.2 TB/s for the store pattern alone (tools/micro) */                    \
        _Pragma("unroll") for (int p2 = 0; p2 < 2; ++p2) {                                                                 \
          const uint2 keep = godd ? pk[2 * p2 + 1] : pk[2 * p2];                                                           \
          const uint2 send = godd ? pk[2 * p2] : pk[2 * p2 + 1];                                                           \
          uint2 recv;                                                                                                      \
          recv.x = __shfl_xor(send.x, 16);                                                                                 \
          recv.y = __shfl_xor(send.y, 16);                                                                                 \
          const uint4 o16 = godd ? make_uint4(recv.x, recv.y, keep.x, keep.y) : make_uint4(keep.x, keep.y, recv.x, recv.y); \
          *reinterpret_cast<uint4*>(out_bf16 + orow + n0 + wc * 64 + 16 * (2 * p2 + (godd ? 1 : 0)) + ncol8) = o16;        \
        }                                                                                                                  \
      }                                                                                                                    \
    }                                                                                                                      \
  }
  const bool godd = (lane >> 4) & 1;                      /* odd lane group: keeps tile 2p+1, gets the left 4 columns from its partner */
  const int ncol8 = 4 * ((lane >> 4) & ~1);               /* first of this lane's 8 consecutive columns inside the 16-column tile */
  if (interior) {
    SE3_EPILOGUE_BODY(false)
  } else {
    SE3_EPILOGUE_BODY(true)
  }
#undef SE3_EPILOGUE_BODY
}


// ------------------------------------------------------------------------------------------------------------------
// Persistent form: gridDim.x workgroups (one per CU, a multiple of 8) each walk a list of output tiles.  The LDS ring
// runs CONTINUOUSLY across tile boundaries (the last three K-steps of a tile already fetch the first three of the
// next), and the epilogue's stores are left in flight under the next tile's main loop -- counted vmcnt waits include
// them -- so neither the pipeline fill nor the HBM-bound store phase leaves the matrix pipe idle.
// Tile order: XCD x (= blockIdx & 7) owns a contiguous range of tile ids (n fastest); its 32 workgroups take ids
// range_start + (blockIdx >> 3) + 32 i, so the tiles in flight on one XCD share A panels / weight rows in its L2.
// ------------------------------------------------------------------------------------------------------------------
template <int ACT, int EF>
__global__ __launch_bounds__(k3Threads) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm3p_bf16_kernel(
    const uint16_t* __restrict__ A, int lda, const uint16_t* __restrict__ W, int ldw, const float* __restrict__ bias,
    const float* __restrict__ residual, int M, int N, int K, uint16_t* __restrict__ out_bf16, float* __restrict__ out_f32,
    int ldc, int tiles_m, int tiles_n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // ---- this workgroup's tile list
  const int nwg = tiles_m * tiles_n;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wpx = gridDim.x >> 3;
  const int q = nwg >> 3, rem = nwg & 7;
  const int range_start = (xcd < rem) ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
  const int range_count = q + (xcd < rem ? 1 : 0);
  const int my_tiles = (range_count > slot) ? (range_count - slot + wpx - 1) / wpx : 0;
  if (my_tiles == 0) return;

  const int r16 = lane >> 2, pos = lane & 3;
  const uint16_t* a_cur[2];
  const uint16_t* b_cur[2];
  const uint16_t* a_nxt[2];
  const uint16_t* b_nxt[2];
#define SE3P_SET_PTRS(ap, bp, tile_id)                                             \
  do {                                                                             \
    const int tm_ = (tile_id) / tiles_n, tn_ = (tile_id) - tm_ * tiles_n;          \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                \
      const int row = 16 * (i * 8 + wave) + r16;                                   \
      const int lc = (pos ^ swz3_f(row)) << 3;                                     \
      ap[i] = A + (size_t)min(tm_ * k3BM + row, M - 1) * lda + lc;                 \
      bp[i] = W + (size_t)min(tn_ * k3BN + row, N - 1) * ldw + lc;                 \
    }                                                                              \
  } while (0)
#define SE3P_ISSUE(ap, bp, g, st)                                                                                          \
  do {                                                                                                                     \
    char* sb_ = smem + (st) * k3Stage + wave * 1024;                                                                       \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                        \
      __builtin_amdgcn_global_load_lds((glb3_ptr_t)(ap[i] + (g) * k3BK), (lds3_ptr_t)(sb_ + i * 8192), 16, 0, 0);          \
      __builtin_amdgcn_global_load_lds((glb3_ptr_t)(bp[i] + (g) * k3BK), (lds3_ptr_t)(sb_ + k3ABytes + i * 8192), 16, 0, 0); \
    }                                                                                                                      \
  } while (0)

  const int frow = lane & 15, fch = lane >> 4;
  int a_off[8], b_off[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) a_off[i] = swz3(wr * 128 + i * 16 + frow, fch);
#pragma unroll
  for (int j = 0; j < 4; ++j) b_off[j] = k3ABytes + swz3(wc * 64 + j * 16 + frow, fch);

  const int nk = K / k3BK;                 // >= 3 (launcher)
  int tile_id = range_start + slot;
  SE3P_SET_PTRS(a_cur, b_cur, tile_id);
  SE3P_ISSUE(a_cur, b_cur, 0, 0);
  SE3P_ISSUE(a_cur, b_cur, 1, 1);
  SE3P_ISSUE(a_cur, b_cur, 2, 2);
  const bool late = wave >= 4;
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (late) __builtin_amdgcn_s_barrier();

  constexpr bool RES = EF & 1, OBF = EF & 2, OF32 = EF & 4;
  // store instructions of one interior-tile epilogue: 8 x 4 fp32 (16 B per lane), or 8 x 2 bf16 after the lane-pair exchange
  constexpr int kStores = OBF ? 16 : 32;
  const int mrow = lane & 15, ncol = 4 * (lane >> 4);
  const bool godd = (lane >> 4) & 1;
  const int ncol8 = 4 * ((lane >> 4) & ~1);
  int st = 0;
  bool stores_pending = false;             // the previous tile's epilogue left exactly kStores stores in flight
  for (int ti = 0; ti < my_tiles; ++ti) {
    const bool has_next = ti + 1 < my_tiles;
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int g = 0; g < nk; ++g) {
      const char* sb = smem + st * k3Stage;
      bf16x8 af[8], bfr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sb + b_off[j]);
#pragma unroll
      for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sb + a_off[i]);
      // refill ring slot (st+3)&3 with the K-step three ahead in the continuous stream (this tile's or the next tile's)
      const int gi = g + 3;
      bool issued = false;
      if (gi < nk) {
        SE3P_ISSUE(a_cur, b_cur, gi, (st + 3) & 3);
        issued = true;
      } else if (has_next) {
        if (gi == nk) SE3P_SET_PTRS(a_nxt, b_nxt, tile_id + wpx);
        SE3P_ISSUE(a_nxt, b_nxt, gi - nk, (st + 3) & 3);
        issued = true;
      }
      // wait until the NEXT K-step of the stream has landed.  VMEM ops younger than its 4 DMAs, in issue order:
      //   [previous tile's epilogue stores, only while g < 2] + the DMA groups of the two following steps.
      if (issued) {
        if (stores_pending && g < 2) {
          if constexpr (kStores == 16) asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory");   // 8 + kStores
          else asm volatile("s_waitcnt vmcnt(40) lgkmcnt(0)" ::: "memory");
        }
        else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      } else if (g + 2 < nk) {
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");      // stream tail: only step g+2 remains younger
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      SE_SETPRIO(1);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      SE_SETPRIO(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      st = (st + 1) & 3;
    }

    // ---- epilogue of this tile (no LDS, no barrier); its stores stay in flight under the next tile's first K-steps
    const int tm = tile_id / tiles_n, tn = tile_id - tm * tiles_n;
    const int m0 = tm * k3BM, n0 = tn * k3BN;
    float4 bb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gn = min(n0 + wc * 64 + j * 16 + ncol, N - 4);
      bb[j] = bias ? *reinterpret_cast<const float4*>(bias + gn) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const bool interior = (m0 + k3BM <= M) && (n0 + k3BN <= N) && (!OBF || ((ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(out_bf16) & 15) == 0));
#define SE3P_EPILOGUE_BODY(PRED)                                                                                           \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                          \
    const int gm = m0 + wr * 128 + i * 16 + mrow;                                                                          \
    const bool mok = gm < M;                                                                                               \
    const size_t orow = (size_t)min(gm, M - 1) * ldc;                                                                      \
    float4 rr[4];                                                                                                          \
    if constexpr (RES) {                                                                                                   \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                      \
        const int gn = min(n0 + wc * 64 + j * 16 + ncol, N - 4);                                                           \
        rr[j] = *reinterpret_cast<const float4*>(residual + orow + gn);                                                    \
      }                                                                                                                    \
    }                                                                                                                      \
    uint2 pk[4];                                                                                                           \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                        \
      const int gn = n0 + wc * 64 + j * 16 + ncol;                                                                         \
      float v0 = acc[i][j][0] + bb[j].x, v1 = acc[i][j][1] + bb[j].y, v2 = acc[i][j][2] + bb[j].z, v3 = acc[i][j][3] + bb[j].w; \
      if constexpr (ACT == SE_ACT_GELU) {                                                                                  \
        const f32x2 ga = gelu_erf2((f32x2){v0, v1}), gb = gelu_erf2((f32x2){v2, v3});                                      \
        v0 = ga.x; v1 = ga.y; v2 = gb.x; v3 = gb.y;                                                                        \
      } else {                                                                                                             \
        v0 = act3(v0, ACT); v1 = act3(v1, ACT); v2 = act3(v2, ACT); v3 = act3(v3, ACT);                                    \
      }                                                                                                                    \
      if constexpr (RES) { v0 += rr[j].x; v1 += rr[j].y; v2 += rr[j].z; v3 += rr[j].w; }                                   \
      pk[j] = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));                                                        \
      if (!(PRED) || (mok && gn < N)) {                                                                                    \
        if constexpr (OF32) *reinterpret_cast<float4*>(out_f32 + orow + gn) = make_float4(v0, v1, v2, v3);                 \
        if constexpr (OBF) {                                                                                               \
          if (PRED) *reinterpret_cast<uint2*>(out_bf16 + orow + gn) = pk[j];                                               \
        }                                                                                                                  \
      }                                                                                                                    \
    }                                                                                                                      \
    if constexpr (OBF) {                                                                                                   \
      if (!(PRED)) {      /* 16-B stores through the lane-pair exchange, as in the one-tile kernel above */                 \
        _Pragma("unroll") for (int p2 = 0; p2 < 2; ++p2) {                                                                 \
          const uint2 keep = godd ? pk[2 * p2 + 1] : pk[2 * p2];                                                           \
          const uint2 send = godd ? pk[2 * p2] : pk[2 * p2 + 1];                                                           \
          uint2 recv;                                                                                                      \
          recv.x = __shfl_xor(send.x, 16);                                                                                 \
          recv.y = __shfl_xor(send.y, 16);                                                                                 \
          const uint4 o16 = godd ? make_uint4(recv.x, recv.y, keep.x, keep.y) : make_uint4(keep.x, keep.y, recv.x, recv.y); \
          *reinterpret_cast<uint4*>(out_bf16 + orow + n0 + wc * 64 + 16 * (2 * p2 + (godd ? 1 : 0)) + ncol8) = o16;        \
        }                                                                                                                  \
      }                                                                                                                    \
    }                                                                                                                      \
  }
    if (interior) {
      SE3P_EPILOGUE_BODY(false)
      stores_pending = true;               // exactly kStores store instructions were issued by this wave
    } else {
      SE3P_EPILOGUE_BODY(true)
      // exec-masked stores: the instruction count is not fixed, so drain everything (the stores are the YOUNGEST
      // operations, in-order retirement means only vmcnt(0) covers them) before the counted waits resume
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stores_pending = false;
    }
#undef SE3P_EPILOGUE_BODY
    if (has_next) {
      tile_id += wpx;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a_cur[i] = a_nxt[i];
        b_cur[i] = b_nxt[i];
      }
    }
  }
  if (!late) __builtin_amdgcn_s_barrier();                 // re-align the two groups (barrier counts must match)
#undef SE3P_SET_PTRS
#undef SE3P_ISSUE
}

}  // namespace se

namespace {
struct G3Args {
  const uint16_t* A; int lda; const uint16_t* W; int ldw; const float* bias; const float* residual; int M, N, K;
  uint16_t* out_bf16; float* out_f32; int ldc; hipStream_t st;
};
template <int ACT, int EF>
int launch3s(const G3Args& g, int sched);

template <int ACT, int EF, int SCHED>
int launch3(const G3Args& g) {
  const int tiles_m = (g.M + se::k3BM - 1) / se::k3BM, tiles_n = (g.N + se::k3BN - 1) / se::k3BN;
  static int dbg = -1, group_m = 0;
  if (dbg < 0) {
    const char* e = getenv("SE_AMD_GEMM_DBG");
    dbg = e ? atoi(e) : 0;
    const char* gm = getenv("SE_AMD_GEMM_GROUPM");
    group_m = gm ? atoi(gm) : 4;
    if (group_m < 1) group_m = 1;
  }
  static bool attr_set = false;
  if (!attr_set) {
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm3_bf16_kernel<ACT, EF, SCHED>), hipFuncAttributeMaxDynamicSharedMemorySize, se::k3Lds));
    attr_set = true;
  }
  hipLaunchKernelGGL((se::gemm3_bf16_kernel<ACT, EF, SCHED>), dim3(tiles_m * tiles_n), dim3(se::k3Threads), se::k3Lds, g.st, g.A, g.lda, g.W, g.ldw,
                     g.bias, g.residual, g.M, g.N, g.K, g.out_bf16, g.out_f32, g.ldc, tiles_m, tiles_n, dbg, group_m);
  SE_LAUNCH_CHECK();
  return SE_OK;
}
template <int ACT, int EF>
int launch3p(const G3Args& g) {
  const int tiles_m = (g.M + se::k3BM - 1) / se::k3BM, tiles_n = (g.N + se::k3BN - 1) / se::k3BN;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    SE_HIP(hipGetDevice(&dev));
    SE_HIP(hipGetDeviceProperties(&prop, dev));
    n_cu = prop.multiProcessorCount & ~7;          // one workgroup per CU, a multiple of the 8 XCDs
    if (n_cu < 8) n_cu = 8;
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm3p_bf16_kernel<ACT, EF>), hipFuncAttributeMaxDynamicSharedMemorySize, se::k3Lds));
  }
  static bool attr_set = false;
  if (!attr_set) {
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm3p_bf16_kernel<ACT, EF>), hipFuncAttributeMaxDynamicSharedMemorySize, se::k3Lds));
    attr_set = true;
  }
  const int grid = n_cu;
  hipLaunchKernelGGL((se::gemm3p_bf16_kernel<ACT, EF>), dim3(grid), dim3(se::k3Threads), se::k3Lds, g.st, g.A, g.lda, g.W, g.ldw, g.bias,
                     g.residual, g.M, g.N, g.K, g.out_bf16, g.out_f32, g.ldc, tiles_m, tiles_n);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

template <int ACT, int EF>
int launch3s(const G3Args& g, int sched) {
  if (sched == 2) return launch3p<ACT, EF>(g);
  return sched ? launch3<ACT, EF, 1>(g) : launch3<ACT, EF, 0>(g);
}
}  // namespace

// returns 1 if this kernel does not handle the call (caller falls back to gemm2), 0 on success, < 0 on error
extern "C" int se_gemm3_launch(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const float* residual_f32,
                               int M, int N, int K, int act, uint16_t* out_bf16, float* out_f32, int ldc, int vec_ok, void* stream) {
  const bool vec = vec_ok && (N % 4 == 0) && (ldc % 4 == 0);
  const bool one_out = (out_bf16 != nullptr) != (out_f32 != nullptr);
  if (!vec || !one_out || K % se::k3BK != 0 || K < 3 * se::k3BK || (act != SE_ACT_IDENTITY && act != SE_ACT_GELU)) return 1;
  G3Args g{A, lda, W, ldw, bias, residual_f32, M, N, K, out_bf16, out_f32, ldc, se::as_stream(stream)};
  const bool gelu = act == SE_ACT_GELU, res = residual_f32 != nullptr, obf = out_bf16 != nullptr;
  se::ProfScope prof(se::kProfGemm, 2.0 * M * (double)N * K, g.st);
  static int sched = -1;
  if (sched < 0) {
    const char* e = getenv("SE_AMD_GEMM3_SCHED");
    sched = e ? atoi(e) : 1;        // 0 / 1: one workgroup per tile (two / one [default] barrier pairs per K-step); 2: persistent (measured equal)
  }
  if (!gelu && !res && obf) return launch3s<SE_ACT_IDENTITY, 2>(g, sched);
  if (gelu && !res && obf) return launch3s<SE_ACT_GELU, 2>(g, sched);
  if (!gelu && res && !obf) return launch3s<SE_ACT_IDENTITY, 4 | 1>(g, sched);
  if (!gelu && !res && !obf) return launch3s<SE_ACT_IDENTITY, 4>(g, sched);
  if (gelu && !res && !obf) return launch3s<SE_ACT_GELU, 4>(g, sched);
  return 1;
}
