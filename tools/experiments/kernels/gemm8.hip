// gemm8.hip -- row-complete GEMM + bias + 24-bit residual + LayerNorm for the FFN-output projection (N = 768, K = 3072) on 256 x 384 tiles:
//      x = LayerNorm(A[M,K] . W[768,K]^T + bias + residual) * ln_w + ln_b          (contract and 24-bit stream layout: gemm4.hip)
//
// Why: the 128 x 768 tile of gemm4 / gemm7 makes every workgroup stream the WHOLE weight matrix (4.7 MB at K = 3072) for 128 rows: 110 flop per
// byte staged into LDS, and its K loop runs at a marginal 0.85 PFLOP/s -- the no-DMA ablation of gemm7 runs at 1.29, i.e. the loop is
// feed-bound.  A 256 x 384 tile stages 80 KiB per 64-deep K-tile for 12.6 MFLOP = 157 flop per byte (more than the 128 of the 256 x 256
// GEMMs that reach 1.4 PFLOP/s marginal), with the same 192 accumulator registers per lane.  The price: a row's 768 columns live in TWO
// workgroups, so the LayerNorm statistics cross a workgroup boundary once per launch:
//   grid     : 256 workgroups = 128 row blocks x 2 column halves; block id -> (xcd = id & 7, slot = id >> 3): row block (slot >> 1) * 8 + xcd,
//              half slot & 1 -- both halves of a row block on ONE XCD (speed only; the protocol is placement-independent), all co-resident
//              (one 160-KiB-LDS workgroup per CU, grid <= CU count: checked by the launcher)
//   exchange : each half computes the exact two-pass (mean, M2) of its 384 columns per row, publishes 256 x (mean, M2) to a scratch slab
//              [plain stores -> every wave s_waitcnt vmcnt(0) -> barrier -> lane 0: agent-scope release fence -> vmcnt(0) -> relaxed agent flag
//              store], polls the partner's flag (one lane, relaxed agent load, bounded spin), agent-scope acquire fence -> vmcnt(0) ->
//              barrier -> plain loads, and combines by Chan's formula: mean = (m_a + m_b) / 2, M2 = M2_a + M2_b + 192 (m_a - m_b)^2.
//              The consumer clears the flag it consumed, so every launch leaves the flags zero (hipGraph replays need no host reset).
//   tile     : 256 x 384 x 64; 8 waves as 2 (M) x 4 (N), wave tile 128 x 96 = 8 x 6 MFMA tiles (v_mfma_f32_16x16x32_bf16)
//   LDS      : 2 K-tile buffers x (A_0, A_1: 128 rows each; B_0, B_1: 192 rows each) x 128-B rows = 160 KiB; chunk ^ ((row >> 1) & 7)
//   phases   : per K-tile eight {reads | barrier | 12 MFMAs | barrier} phases: quadrants (0,0) (0,1) (1,1) (1,0) on k-slice 0, then back
//              (1,0) (1,1) (0,1) (0,0) on k-slice 1, so each phase loads ONE new operand piece (4 A or 3 B fragments; 7 at a slice start);
//              waves 4-7 run one barrier behind waves 0-3
//   staging  : K-tile t+1 goes to the other buffer during phases 1-3 of tile t (10 LDS-DMA parts per wave); one wait (vmcnt(0)) in phase 8
//
// MEASURED (round 2, B = 32, same box, tools/bench_kernels.py res24): parity-green (tests/test_gpu_encoder_blocks.py::
// test_gemm_res24_pair_exchange_vs_single_tile) but SLOWER than the 128 x 768 kernel: K = 3072 233 vs 187 us, K = 768 95 vs 75 us; whole step
// 4.40 vs 4.13 ms.  The marginal rate per K is 0.82 vs 1.01 PFLOP/s: twelve-MFMA phases (two barriers per 192 matrix-pipe cycles) with the
// ten LDS-DMA issues of a K-tile bunched into three read sections lose more than the better flop-per-byte ratio gains, and the launch's
// fixed part (residual read + LayerNorm + 150 MB of output, + the exchange) is 20 us longer.  Off by default (SE_AMD_GEMM8=1 enables it for
// K >= 1536); kept as the measured answer to "a wider tile for FFN2" and as the tested reference of the agent-scope pair exchange.
#include <stdlib.h>
#include "common.h"
#include "bf16.h"
#include "prof.h"

namespace se {

constexpr int k8BM = 256, k8BN = 384, k8N = 768, k8Threads = 512;
constexpr int k8ASlot = 128 * 128, k8BSlot = 192 * 128, k8Buf = 2 * k8ASlot + 2 * k8BSlot, k8Lds = 2 * k8Buf;      // 16 KiB, 24 KiB, 80 KiB, 160 KiB
constexpr int k8SpinLimit = 1 << 22;

typedef __attribute__((address_space(3))) void* lds8_ptr_t;

// the 24-bit codec of gemm4.hip (dec24 / enc24_lo4: low byte = bits 15..8 of the fp32 pattern as a signed correction to the RNE bf16)
__device__ __forceinline__ float dec24_8(uint32_t hi16, int lo8) { return __uint_as_float((hi16 << 16) + (uint32_t)(lo8 << 8)); }
__device__ __forceinline__ uint32_t enc24_lo8(float y, uint32_t hi16) {
  return (uint32_t)min(((int)(__float_as_uint(y) - (hi16 << 16)) + 128) >> 8, 127) & 0xffu;
}
// byte offset of the 4 low bytes of (row, 4 columns starting at col4, col4 % 4 == 0) in gemm4's tile-major layout:
// [128-row block][wave = 2 (64-row half) x 4 (192-column group)][16-row tile][16-column tile][lane = 16 (col quad) + row][4 B]
__device__ __forceinline__ size_t lo_off8(int row, int col4) {
  const int id = row >> 7, wr4 = (row >> 6) & 1, i = (row >> 4) & 3, wc4 = col4 / 192, c = col4 - wc4 * 192, t = c >> 4;
  const int ln = ((c >> 2) & 3) * 16 + (row & 15);
  return ((((size_t)id * 8 + wr4 * 4 + wc4) * 4 + i) * 12 + t) * 256 + ln * 4;
}

// ROUT: 1 = the stream leaves as (bf16 hi, int8 lo); 0 = fp32 rows (last layer)
template <int ROUT>
__global__ __launch_bounds__(k8Threads) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm8_res24_ln_kernel(
    const uint16_t* __restrict__ A, int lda, const uint16_t* __restrict__ W, int ldw, const float* __restrict__ bias,
    const uint16_t* __restrict__ res_hi, const uint8_t* __restrict__ res_lo, const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps,
    int M, int K, float* __restrict__ out_f32, uint16_t* __restrict__ out_bf16, uint8_t* __restrict__ out_lo, int nrb,
    float* __restrict__ xchg, int* __restrict__ flags, int* __restrict__ err) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int rb = (slot >> 1) * 8 + xcd, half = slot & 1;
  if (rb >= nrb) return;                          // whole workgroup (uniform): fewer than 128 row blocks
  const int m0 = rb * k8BM, n0 = half * k8BN;

  // ---- DMA sources as 32-bit byte offsets.  1-KiB part = 8 slot rows x 128 B; lane -> slot row 8 q + (lane >> 3), logical chunk (lane & 7) ^ ((row >> 1) & 7).
  //      A_h slot row rho <-> tile row (rho >> 6) * 128 + h * 64 + (rho & 63); B_h slot row rho <-> tile column (rho / 48) * 96 + h * 48 + rho % 48
  //      The ten per-lane offsets of a K-tile are REBUILT at every issue from two registers (first slot row, chunk column): kept as ten
  //      loop-invariant registers they were spilled, and every reload drained the LDS-DMA pipeline (scratch loads share vmcnt)
  const uint32_t lds_wave = (uint32_t)(size_t)(lds8_ptr_t)smem + wave * 1024;
#define SE8_DMA1(base_bytes, off32, lds_dst)                                                                               \
  do {                                                                                                                     \
    uint32_t keep_;                                                                                                        \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"           \
                 : "=&s"(keep_) : "v"(off32), "s"(base_bytes), "s"(lds_dst) : "memory");                                   \
  } while (0)
  // the wave's 10 parts of K-tile kt into buffer buf, in the order A_0 A_0 B_0 B_0 B_0 A_1 A_1 B_1 B_1 B_1; [first, last) selects a range
#define SE8_STAGE(buf, kt, first, last)                                                                                    \
  do {                                                                                                                     \
    const char* sa_ = reinterpret_cast<const char*>(A) + (size_t)(kt) * 128;                                               \
    const char* sw_ = reinterpret_cast<const char*>(W) + (size_t)(kt) * 128;                                               \
    const uint32_t d_ = lds_wave + (uint32_t)((buf) * k8Buf);                                                              \
    /* the lane id is re-derived at every issue (v_mbcnt on an opaque zero): a lane-constant register kept across the K loop was   \
       spilled, and each reload's compiler-inserted vmcnt(0) drained the LDS-DMA pipeline */                                  \
    unsigned z_ = 0u;                                                                                                      \
    asm volatile("v_mov_b32 %0, 0" : "=v"(z_));                                                                            \
    const int ln_ = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z_));                               \
    const int r0_ = 8 * wave + (ln_ >> 3);                                                                                 \
    const int lc2_ = (((ln_ & 7) ^ ((r0_ >> 1) & 7)) << 3) * 2;                                                            \
    _Pragma("unroll") for (int e_ = (first); e_ < (last); ++e_) {                                                          \
      const int isb_ = (e_ >= 2 && e_ < 5) || e_ >= 7, h_ = e_ >= 5, p_ = isb_ ? (h_ ? e_ - 7 : e_ - 2) : (h_ ? e_ - 5 : e_); \
      const int rho_ = r0_ + 64 * p_;                                                                                      \
      if (!isb_) {                                                                                                         \
        const int trow_ = (rho_ >> 6) * 128 + h_ * 64 + (rho_ & 63);                                                       \
        const uint32_t of_ = (uint32_t)(min(m0 + trow_, M - 1) * lda) * 2u + (uint32_t)lc2_;                               \
        SE8_DMA1(sa_, of_, d_ + (uint32_t)(h_ * k8ASlot + p_ * 8192));                                                     \
      } else {                                                                                                             \
        const int wq_ = rho_ / 48, tcol_ = wq_ * 96 + h_ * 48 + (rho_ - wq_ * 48);                                         \
        const uint32_t of_ = (uint32_t)((n0 + tcol_) * ldw) * 2u + (uint32_t)lc2_;                                         \
        SE8_DMA1(sw_, of_, d_ + (uint32_t)(2 * k8ASlot + h_ * k8BSlot + p_ * 8192));                                       \
      }                                                                                                                    \
    }                                                                                                                      \
  } while (0)

  const int nk = K / 64;                    // >= 2 (launcher)
  SE8_STAGE(0, 0, 0, 10);                   // ring prologue first: the residual rows read below travel beside it

  // ---- C^T accumulators acc[i8][j6]: row m0 + wr * 128 + 16 i8 + (lane & 15), columns n0 + wc * 96 + 16 j6 + 4 (lane >> 4) + {0..3};
  //      initialised with bias + the 24-bit residual (hi = bf16 rows, lo = gemm4's tile-major bytes)
  f32x4 acc[8][6];
  {
    const int mrow = lane & 15, cq = lane >> 4;
    const int colw = n0 + wc * 96 + 4 * cq;
    float4 bb[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) bb[j] = bias ? *reinterpret_cast<const float4*>(bias + colw + 16 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int gm = min(m0 + wr * 128 + i * 16 + mrow, M - 1);
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int col = colw + 16 * j;
        const uint2 h4 = *reinterpret_cast<const uint2*>(res_hi + (size_t)gm * k8N + col);
        const uint32_t l4 = *reinterpret_cast<const uint32_t*>(res_lo + lo_off8(gm, col));
        acc[i][j] = (f32x4){dec24_8(h4.x & 0xffffu, (int)(int8_t)(l4 & 0xffu)) + bb[j].x, dec24_8(h4.x >> 16, (int)(int8_t)((l4 >> 8) & 0xffu)) + bb[j].y,
                            dec24_8(h4.y & 0xffffu, (int)(int8_t)((l4 >> 16) & 0xffu)) + bb[j].z, dec24_8(h4.y >> 16, (int)(int8_t)(l4 >> 24)) + bb[j].w};
      }
    }
  }

  // ---- fragment addresses (lane -> row lane & 15 of a 16-row tile, logical chunk 4 s + (lane >> 4)); tiles 16 rows apart share the swizzle term
  const int frow = lane & 15, fch = lane >> 4;
  // (k-slice 1 = chunk + 4 = byte offset ^ 64 under the XOR swizzle: one address register per operand, the other is one v_xor away)
  int a_ad0, b_ad0;
  {
    const int ra = wr * 64 + frow, rbw = wc * 48 + frow;
    a_ad0 = ra * 128 + ((fch ^ ((ra >> 1) & 7)) << 4);
    b_ad0 = 2 * k8ASlot + rbw * 128 + ((fch ^ ((rbw >> 1) & 7)) << 4);
  }
  bf16x8 af[4], bfr[3];
#define SE8_READ_A(buf, h, s)                                                                                              \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                                         \
    af[i_] = *reinterpret_cast<const bf16x8*>(smem + (buf) * k8Buf + (h) * k8ASlot + (a_ad0 ^ ((s) * 64)) + i_ * 2048);
#define SE8_READ_B(buf, h, s)                                                                                              \
  _Pragma("unroll") for (int j_ = 0; j_ < 3; ++j_)                                                                         \
    bfr[j_] = *reinterpret_cast<const bf16x8*>(smem + (buf) * k8Buf + (h) * k8BSlot + (b_ad0 ^ ((s) * 64)) + j_ * 2048);
#define SE8_MMA(ha, hb)                                                                                                    \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) _Pragma("unroll") for (int j_ = 0; j_ < 3; ++j_)                        \
    acc[(ha) * 4 + i_][(hb) * 3 + j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j_], af[i_], acc[(ha) * 4 + i_][(hb) * 3 + j_], 0, 0, 0);
#define SE8_SYNC_A()                                                                                                       \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                       \
  __builtin_amdgcn_sched_barrier(0);                                                                                       \
  __builtin_amdgcn_s_barrier();                                                                                            \
  __builtin_amdgcn_sched_barrier(0);                                                                                       \
  __builtin_amdgcn_s_setprio(1);
#define SE8_SYNC_B()                                                                                                       \
  __builtin_amdgcn_s_setprio(0);                                                                                           \
  __builtin_amdgcn_sched_barrier(0);                                                                                       \
  __builtin_amdgcn_s_barrier();                                                                                            \
  __builtin_amdgcn_sched_barrier(0);
#define SE8_PHASE(READS, ISSUE, WAIT, ha, hb)                                                                              \
  READS ISSUE; WAIT; SE8_SYNC_A() SE8_MMA(ha, hb) SE8_SYNC_B()
#define SE8_NOP ((void)0)
  // K-tile from buffer BUF; tile T + 1 (if it exists) staged into the other buffer during phases 1-3
#define SE8_TILE(BUF, T)                                                                                                   \
  {                                                                                                                        \
    const bool more_ = (T) + 1 < nk;                                                                                       \
    SE8_PHASE(SE8_READ_A(BUF, 0, 0) SE8_READ_B(BUF, 0, 0), if (more_) SE8_STAGE((BUF) ^ 1, (T) + 1, 0, 4), SE8_NOP, 0, 0)  \
    SE8_PHASE(SE8_READ_B(BUF, 1, 0), if (more_) SE8_STAGE((BUF) ^ 1, (T) + 1, 4, 7), SE8_NOP, 0, 1)                        \
    SE8_PHASE(SE8_READ_A(BUF, 1, 0), if (more_) SE8_STAGE((BUF) ^ 1, (T) + 1, 7, 10), SE8_NOP, 1, 1)                       \
    SE8_PHASE(SE8_READ_B(BUF, 0, 0), SE8_NOP, SE8_NOP, 1, 0)                                                               \
    SE8_PHASE(SE8_READ_A(BUF, 1, 1) SE8_READ_B(BUF, 0, 1), SE8_NOP, SE8_NOP, 1, 0)                                         \
    SE8_PHASE(SE8_READ_B(BUF, 1, 1), SE8_NOP, SE8_NOP, 1, 1)                                                               \
    SE8_PHASE(SE8_READ_A(BUF, 0, 1), SE8_NOP, SE8_NOP, 0, 1)                                                               \
    SE8_PHASE(SE8_READ_B(BUF, 0, 1), SE8_NOP, asm volatile("s_waitcnt vmcnt(0)" ::: "memory"), 0, 0)                       \
  }
  const bool late = wave >= 4;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // K-tile 0 landed (and the residual loads above)
  __builtin_amdgcn_s_barrier();
  if (late) __builtin_amdgcn_s_barrier();                   // stagger: waves 4-7 one barrier behind
  // (one K-tile per loop iteration with a run-time buffer index: the two-tile unrolled body with compile-time buffer offsets made the
  //  register allocator spill accumulators inside the loop)
  for (int t = 0; t < nk; ++t) {
    const int bsel = t & 1;
    SE8_TILE(bsel, t)
  }
  if (!late) __builtin_amdgcn_s_barrier();                  // re-align the two groups
  __syncthreads();                                          // ring is dead: reuse it below
#undef SE8_TILE
#undef SE8_PHASE
#undef SE8_STAGE
#undef SE8_DMA1

  // (the epilogue rebuilds its lane-derived terms from an opaque copy of the lane id: otherwise the 48 residual addresses of the prologue are
  //  kept alive -- spilled -- across the K loop as common subexpressions of the 48 output addresses)
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));
  const int mrow = lane_e & 15, cq = lane_e >> 4;
  const int colw = n0 + wc * 96 + 4 * cq;
  // ---- epilogue: LayerNorm statistics.  Local (this half's 384 columns): exact two-pass mean and M2 per row
  float* colv = reinterpret_cast<float*>(smem);             // [2][384]: ln_w, ln_b of this half
  for (int c = tid; c < k8BN; c += k8Threads) {
    colv[c] = ln_w[n0 + c];
    colv[k8BN + c] = ln_b[n0 + c];
  }
  float* red = reinterpret_cast<float*>(smem + 4096);       // [pass][wr][wc][128]
  float* stat = reinterpret_cast<float*>(smem + 16384);     // [256][2]: this half's (mean, M2); later the partner's
  float mean[8], rstd[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) s += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (cq == 0) red[(wr * 4 + wc) * 128 + i * 16 + mrow] = s;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float* rp = red + wr * 512 + i * 16 + mrow;
    mean[i] = (rp[0] + rp[128] + rp[256] + rp[384]) * (1.0f / k8BN);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float d = acc[i][j][r] - mean[i];
        q = fmaf(d, d, q);
      }
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
    if (cq == 0) red[1024 + (wr * 4 + wc) * 128 + i * 16 + mrow] = q;
  }
  __syncthreads();
  // publish (mean, M2) of the 256 rows: wave wc == 0 of each wave row writes its 128 rows (lanes cq == 0 hold the row index)
  float* mine = xchg + ((size_t)rb * 2 + half) * 512;
  const float* theirs = xchg + ((size_t)rb * 2 + (half ^ 1)) * 512;
  float m2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float* rp = red + 1024 + wr * 512 + i * 16 + mrow;
    m2[i] = rp[0] + rp[128] + rp[256] + rp[384];
    if (wc == 0 && cq == 0) {
      const int r = wr * 128 + i * 16 + mrow;
      *reinterpret_cast<float2*>(mine + 2 * r) = make_float2(mean[i], m2[i]);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its stores ...
  __syncthreads();                                           // ... before the one lane that publishes for all of them
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store(flags + rb * 2 + half, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // wait for the partner half (co-resident by construction; the spin is bounded all the same)
    int* pf = flags + rb * 2 + (half ^ 1);
    int spins = 0;
    while (__hip_atomic_load(pf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > k8SpinLimit) {
        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  // partner's statistics -> LDS (plain loads after the acquire), then Chan's combination
  if (tid < 256) *reinterpret_cast<float2*>(stat + 2 * tid) = *reinterpret_cast<const float2*>(theirs + 2 * tid);
  __syncthreads();
  if (tid == 0) __hip_atomic_store(flags + rb * 2 + (half ^ 1), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // consumed: leave the flag clear for the next launch
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float2 o = *reinterpret_cast<const float2*>(stat + 2 * (wr * 128 + i * 16 + mrow));
    const float d = mean[i] - o.x;
    const float var = (m2[i] + o.y + d * d * (0.5f * k8BN)) * (1.0f / k8N);
    mean[i] = 0.5f * (mean[i] + o.x);
    rstd[i] = 1.0f / sqrtf(var + eps);
  }
  // ---- normalise + store (16-B bf16 stores through the lane-pair exchange of gemm3 / gemm4)
  const bool interior = m0 + k8BM <= M;
  const bool godd = cq & 1;
  const int lo_rows = (M + 127) & ~127;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int gm = m0 + wr * 128 + i * 16 + mrow;
    const bool ok = interior || gm < M;
    const int gmc = min(gm, M - 1);
    const size_t o = (size_t)gmc * k8N + colw;
    const size_t orow8 = (size_t)gmc * k8N + n0 + wc * 96 + 4 * (cq & ~1);
    uint2 pk_prev = make_uint2(0u, 0u);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const float4 lw = *reinterpret_cast<const float4*>(colv + wc * 96 + 4 * cq + 16 * j);
      const float4 lb = *reinterpret_cast<const float4*>(colv + k8BN + wc * 96 + 4 * cq + 16 * j);
      const float y0 = lw.x * ((acc[i][j][0] - mean[i]) * rstd[i]) + lb.x;
      const float y1 = lw.y * ((acc[i][j][1] - mean[i]) * rstd[i]) + lb.y;
      const float y2 = lw.z * ((acc[i][j][2] - mean[i]) * rstd[i]) + lb.z;
      const float y3 = lw.w * ((acc[i][j][3] - mean[i]) * rstd[i]) + lb.w;
      if (!ROUT) {
        if (ok) *reinterpret_cast<float4*>(out_f32 + o + 16 * j) = make_float4(y0, y1, y2, y3);
      } else {
        const uint2 pk = make_uint2(pack_bf16x2(y0, y1), pack_bf16x2(y2, y3));
        // tile-major lo bytes: rows past M inside the last 128-row tile are written too (the buffer covers whole 128-row tiles, never read as
        // real rows); rows beyond it (this kernel's tiles are 256 rows) are not
        if (gm < lo_rows) *reinterpret_cast<uint32_t*>(out_lo + lo_off8(gm, colw + 16 * j)) =
            enc24_lo8(y0, pk.x & 0xffffu) | (enc24_lo8(y1, pk.x >> 16) << 8) | (enc24_lo8(y2, pk.y & 0xffffu) << 16) | (enc24_lo8(y3, pk.y >> 16) << 24);
        if (j & 1) {
          const uint2 keep = godd ? pk : pk_prev, send = godd ? pk_prev : pk;
          uint2 recv;
          recv.x = __shfl_xor(send.x, 16);
          recv.y = __shfl_xor(send.y, 16);
          const uint4 o16 = godd ? make_uint4(recv.x, recv.y, keep.x, keep.y) : make_uint4(keep.x, keep.y, recv.x, recv.y);
          if (ok) *reinterpret_cast<uint4*>(out_bf16 + orow8 + 16 * (godd ? j : j - 1)) = o16;
        }
        pk_prev = pk;
      }
    }
  }
#undef SE8_READ_A
#undef SE8_READ_B
#undef SE8_MMA
#undef SE8_SYNC_A
#undef SE8_SYNC_B
}

size_t gemm8_scratch_bytes() { return (size_t)128 * 2 * 512 * 4 + 128 * 2 * 4 + 256; }      // statistics slabs + flags + error word

// returns 1 when the call is not for this kernel (the caller uses gemm7), 0 on success, < 0 on error
int launch_gemm8_res24_ln(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const uint16_t* res_hi, const uint8_t* res_lo,
                          const float* ln_w, const float* ln_b, float eps, int M, int N, int K, float* out_f32, uint16_t* out_bf16, uint8_t* out_lo,
                          void* scratch, hipStream_t st, int force) {
  static int use8 = -1, n_cu = 0;
  if (use8 < 0) {
    const char* e = getenv("SE_AMD_GEMM8");
    use8 = e ? atoi(e) : 0;        // measured SLOWER than the 128 x 768 kernel (header): off unless SE_AMD_GEMM8=1
    int dev = 0;
    hipDeviceProp_t prop;
    SE_HIP(hipGetDevice(&dev));
    SE_HIP(hipGetDeviceProperties(&prop, dev));
    n_cu = prop.multiProcessorCount;
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8_res24_ln_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, k8Lds));
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm8_res24_ln_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, k8Lds));
  }
  const int nrb = (M + k8BM - 1) / k8BM;
  // the pair protocol needs every workgroup resident at once: one 160-KiB workgroup per CU, 256 of them; long K only (at K = 768 the
  // launch is dominated by its residual / output phases and gemm7 is as fast)
  const bool shape_ok = force ? (K >= 128 && nrb >= 1) : (K >= 1536 && nrb >= 64);
  if ((!use8 && !force) || !scratch || N != k8N || K % 64 != 0 || !shape_ok || nrb > 128 || n_cu < 256 || (size_t)M * lda >= (1u << 31) || (size_t)k8N * ldw >= (1u << 31) ||
      (out_bf16 != nullptr) == (out_f32 != nullptr) || (out_bf16 && !out_lo))
    return 1;
  float* xchg = reinterpret_cast<float*>(scratch);
  int* flags = reinterpret_cast<int*>(reinterpret_cast<char*>(scratch) + (size_t)128 * 2 * 512 * 4);
  int* err = flags + 256;
  ProfScope prof(kProfGemm, 2.0 * M * (double)N * K, st);
  if (out_bf16)
    hipLaunchKernelGGL(gemm8_res24_ln_kernel<1>, dim3(256), dim3(k8Threads), k8Lds, st, A, lda, W, ldw, bias, res_hi, res_lo, ln_w, ln_b, eps, M, K, out_f32, out_bf16,
                       out_lo, nrb, xchg, flags, err);
  else
    hipLaunchKernelGGL(gemm8_res24_ln_kernel<0>, dim3(256), dim3(k8Threads), k8Lds, st, A, lda, W, ldw, bias, res_hi, res_lo, ln_w, ln_b, eps, M, K, out_f32, out_bf16,
                       out_lo, nrb, xchg, flags, err);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

}  // namespace se
