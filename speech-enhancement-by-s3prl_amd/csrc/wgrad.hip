// wgrad.hip -- weight gradient dW[N,K] = dY[M,N]^T . X[M,K] straight from the ROW-MAJOR activations (row E2).
//
// The reduction dim is M = B*T (~32 000 rows), the slow index of both operands, so neither is in the K-contiguous layout
// the forward GEMM kernels stage.  Instead of transposing both operands through HBM (15 % of a fine-tune step), this kernel
// stages row-major [32 m][64 col] panels with LDS-DMA and takes the MFMA fragments out of them with the hardware-transposed
// LDS read (ds_read_b64_tr_b16), the way the flash-attention kernels read V^T / Q^T:
//   tile     : 256 (n) x 256 (k) outputs per 512-thread workgroup, 8 waves as 2 x 4, wave tile 128 x 64 = 4 x 2
//              v_mfma_f32_32x32x16_bf16 accumulators (128 registers); 32 rows of m per stage
//   staging  : global_load_lds_dwordx4 into a 4-deep ring of 32 KiB stages = 4 dY panels + 4 X panels of [32 m][64 col]
//              bf16 (128-B rows, the kv_off swizzle of mhsa_tile.h: conflict-free for the tr reads), two stages in flight
//              behind the one being multiplied; rows past the split's end read a 16-B zero word instead (no tail pass)
//   fragments: per 16 rows of m, 4 + 2 fragments = 12 tr reads feed 8 MFMAs (128 x 64 wave tile: 1.5 reads per MFMA; the first
//              64 x 64 version needed 2 and took 575 clk per 16-row step against 512 clk of MFMA work).
//              Round 4 correction: rounds 1-3 read that 575 as "the b64 transposed read runs at half the LDS rate of ds_read_b128
//              (~8 clk per wave instruction): LDS is this kernel's limiter".  tools/micro/lds_rate.hip measures ds_read_b64_tr_b16
//              at 2.0-2.1 LDS cycles per wave instruction = 247-250 B/clk/CU, the SAME bytes per clock as ds_read_b128 (4.0 cycles,
//              256 B/clk/CU) -- the ~8 clk seen from inside one wave is the instruction's issue / return cadence for ONE wave, not
//              the array's rate.  At 8 waves x 12 reads x 2 cycles = 192 LDS-array cycles per 256-cycle MFMA step the array is
//              busy, not saturated; what the step waits for is not established (no stamps of this loop since the correction).
//   split    : the m range is split so that tiles x splits ~ one workgroup per CU; every split writes its own fp32 slab
//              [N][K] and a slab reduce finishes (deterministic).  Work items are dealt to the XCDs in contiguous ranges
//              (bijective remap), so the tiles of one split -- which re-read the same dY / X slices -- share an L2.
#include "clkprobe.h"
#include <stdlib.h>
#include <algorithm>
#include "common.h"
#include "prof.h"
#include "bf16.h"
#include "mhsa_tile.h"

SE_CLKPROBE_DECL(clkprobe_wgrad)
namespace se {

constexpr int kWN = 256, kWK = 256, kWM = 32;
constexpr int kWPanel = kWM * 64 * 2;                       // 4 KiB
constexpr int kWPanels = kWN / 64 + kWK / 64;               // 8
constexpr int kWStage = kWPanels * kWPanel;                 // 32 KiB
constexpr int kWStages = 4;
constexpr int kWLds = kWStages * kWStage;
constexpr int kWLdsStag = 5 * kWStage;                     // the staggered form's five-stage ring: all 160 KiB
constexpr int kWDma = kWStage / 1024 / 8;                   // DMA instructions per wave per stage (4)

__device__ uint4 g_wgrad_zero = {0u, 0u, 0u, 0u};

typedef __attribute__((address_space(3))) void* w_lds_ptr_t;
typedef const __attribute__((address_space(1))) void* w_glb_ptr_t;
#define SE_WTR(ptr) __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(ptr))
typedef unsigned int h4 __attribute__((ext_vector_type(2)));      // one transposed 64-bit fragment half (asm reads of STAG 3)

// STAG = 1 (round 4): waves 4-7 take the step's one barrier in the MIDDLE of their step (between the two 16-row sub-steps), so the two waves of a
// SIMD run half a step apart instead of re-aligning at every step top (the attention forward gained 7 % that way, mhsa8.hip).  Costs one ring stage:
// a late wave reads tile t right after barrier t - 1, so every wave certifies its pieces of tile t + 1 -- not t -- before barrier t, and the
// refill of step t goes to the stage of tile t - 2 (five stages = all 160 KiB).
template <int STAG>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_tn_kernel(
    const uint16_t* __restrict__ dY, int ldy, const uint16_t* __restrict__ X, int ldx, int M, int N, int K, int m_per_split,
    int tiles_k, int tiles, int work, float* __restrict__ partials, unsigned long long* __restrict__ stamps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  SE_CLKPROBE_BEGIN();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 2, wk = wave & 3;
  const int l31 = lane & 31, hh = lane >> 5;
  // XCD-aware bijective remap: workgroup ids are dealt round-robin to the 8 XCDs; XCD x takes the contiguous range of work
  // items (split-major) [x W/8, (x+1) W/8), i.e. about one split: its tiles re-read the same slices of dY and X from one L2
  int id;
  {
    const int orig = blockIdx.x, xcd = orig & 7, q = work >> 3, r = work & 7;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  const int split = id / tiles, tile = id - split * tiles;
  const int tn = tile / tiles_k, tk = tile - tn * tiles_k;
  const int n0 = tn * kWN, k0 = tk * kWK;
  const int m_begin = split * m_per_split, m_end = min(M, m_begin + m_per_split);
  const int nt = m_end > m_begin ? (m_end - m_begin + kWM - 1) / kWM : 0;

  // ---- DMA sources: piece i of this wave = rows 8 rg .. +7 of panel 2 i + (wave >> 2), rg = wave & 3; lane -> row, LDS slot
  //      lane & 7 holds logical 16-B chunk (lane & 7) ^ f(row) (swizzle on the source side: the DMA destination is lane-linear)
  const int row = 8 * (wave & 3) + (lane >> 3);
  const int f = (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
  const int col8 = ((lane & 7) ^ f) * 8;
  const uint16_t* src[kWDma];
  int ld[kWDma];
  bool col_ok[kWDma];
#pragma unroll
  for (int i = 0; i < kWDma; ++i) {
    const int panel = 2 * i + (wave >> 2);
    if (panel < 4) {
      const int c = n0 + 64 * panel + col8;
      col_ok[i] = c < N;
      src[i] = dY + (col_ok[i] ? c : 0);
      ld[i] = ldy;
    } else {
      const int c = k0 + 64 * (panel - 4) + col8;
      col_ok[i] = c < K;
      src[i] = X + (col_ok[i] ? c : 0);
      ld[i] = ldx;
    }
  }
  const uint16_t* zero = reinterpret_cast<const uint16_t*>(&g_wgrad_zero);
  const int piece_off = (wave >> 2) * kWPanel + (wave & 3) * 1024;
  // LDS-DMA by inline asm (M0 = the wave-uniform LDS destination): through __builtin_amdgcn_global_load_lds the compiler orders every later
  // ds_read behind the DMA with s_waitcnt vmcnt(0) -- rounds 1-4 shipped exactly that in front of each step's first fragment read, so the
  // counted waits below never applied and the "three stages in flight" were one (found with the clock probe: 2.2 GHz in-kernel, 0.39 of the
  // matrix rate at that clock = stalled, not power-bound; the step took 2 200 cycles for 1 024 of MFMA)
  const uint32_t lds_piece = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(w_lds_ptr_t)smem + (uint32_t)piece_off);
#define SEW_ISSUE(t, st)                                                                                        \
  do {                                                                                                          \
    const int m = m_begin + (t) * kWM + row;                                                                    \
    _Pragma("unroll") for (int i = 0; i < kWDma; ++i) {                                                         \
      const uint16_t* p = (m < m_end && col_ok[i]) ? src[i] + (size_t)m * ld[i] : zero;                         \
      uint32_t keep_;                                                                                           \
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                   : "=&s"(keep_) : "v"(p), "s"(lds_piece + (uint32_t)((st) * kWStage + 2 * i * kWPanel)) : "memory"); \
    }                                                                                                           \
  } while (0)

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
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
  const int a_base = 2 * wn * kWPanel, b_base = (4 + wk) * kWPanel;     // this wave's 2 dY panels (128 n) and its X panel (64 k)

  // developer stamps (SE_AMD_WGRAD_STAMPS=1): lane 0 of every wave of workgroups 0..7 appends s_memtime values
  const bool st_on = stamps && lane == 0 && blockIdx.x < 8;
  unsigned long long* st_buf = stamps + ((size_t)blockIdx.x * 8 + wave) * 256;
  int st_i = 0;
#ifdef SE_AMD_STAMPS      // -DSE_AMD_STAMPS builds only: the exec-masked sites cost the hot loop even when switched off
#define SEW_STAMP()                                                        \
  do {                                                                     \
    if (st_on && st_i < 256) st_buf[st_i++] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define SEW_STAMP() do { (void)st_on; (void)st_buf; (void)st_i; } while (0)
#endif
  SEW_STAMP();
  if (nt > 0) SEW_ISSUE(0, 0);
  if (nt > 1) SEW_ISSUE(1, 1);
  if (nt > 2) SEW_ISSUE(2, 2);
  constexpr int kRing = STAG ? 5 : kWStages;
  constexpr bool PIPE = STAG == 2;      // STAG 2: the staggered form with the refill in front of the reads and counted LDS waits
  const bool late = STAG && wave >= 4;
  if (STAG) {
    // tile 0 certified for everybody (the late half reads it before its first in-loop barrier)
    if (nt > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (nt > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  int st = 0;
  // ---- STAG 3 only: asm fragment reads.  Per-lane LDS addresses of the four (block parity, row half) transposed-read positions inside this wave's
  //      dY panels (a) and X panel (b); the stage base is added per step, the panel / sub-step offsets are instruction immediates
  const uint32_t lds_base = (uint32_t)(size_t)(w_lds_ptr_t)smem;
  uint32_t ra[2][2], rb[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y) { ra[x][y] = (uint32_t)(a_base + toff[x][y]); rb[x][y] = (uint32_t)(b_base + toff[x][y]); }
  h4 fa_lo[2][4], fa_hi[2][4], fb_lo[2][2], fb_hi[2][2];
  (void)ra; (void)rb; (void)lds_base;
#define SEW3_RD(dst_, addr_, off_) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst_) : "v"(addr_), "n"(off_))
  // set `k`, stage base sb_ (uniform), sub-step s_: blocks 0..3 of dY = panel (blk >> 1), 32-row half (blk & 1); blocks 0..1 of X
#define SEW3_READ(k, sb_, s_)                                                                                   \
  do {                                                                                                          \
    const uint32_t a00 = ra[0][0] + (sb_), a01 = ra[0][1] + (sb_), a10 = ra[1][0] + (sb_), a11 = ra[1][1] + (sb_); \
    const uint32_t b00 = rb[0][0] + (sb_), b01 = rb[0][1] + (sb_), b10 = rb[1][0] + (sb_), b11 = rb[1][1] + (sb_); \
    SEW3_RD(fa_lo[k][0], a00, (s_) * 2048);           SEW3_RD(fa_hi[k][0], a01, (s_) * 2048);                   \
    SEW3_RD(fa_lo[k][1], a10, (s_) * 2048);           SEW3_RD(fa_hi[k][1], a11, (s_) * 2048);                   \
    SEW3_RD(fa_lo[k][2], a00, kWPanel + (s_) * 2048); SEW3_RD(fa_hi[k][2], a01, kWPanel + (s_) * 2048);         \
    SEW3_RD(fa_lo[k][3], a10, kWPanel + (s_) * 2048); SEW3_RD(fa_hi[k][3], a11, kWPanel + (s_) * 2048);         \
    SEW3_RD(fb_lo[k][0], b00, (s_) * 2048);           SEW3_RD(fb_hi[k][0], b01, (s_) * 2048);                   \
    SEW3_RD(fb_lo[k][1], b10, (s_) * 2048);           SEW3_RD(fb_hi[k][1], b11, (s_) * 2048);                   \
  } while (0)
  // all but the CNT youngest LDS operations complete; set k's registers are operands, so that nothing that uses them is scheduled above the wait
#define SEW3_WAIT(k, CNT)                                                                                       \
  asm volatile("s_waitcnt lgkmcnt(" CNT ")"                                                                     \
               : "+v"(fa_lo[k][0]), "+v"(fa_hi[k][0]), "+v"(fa_lo[k][1]), "+v"(fa_hi[k][1]), "+v"(fa_lo[k][2]), "+v"(fa_hi[k][2]),           \
                 "+v"(fa_lo[k][3]), "+v"(fa_hi[k][3]), "+v"(fb_lo[k][0]), "+v"(fb_hi[k][0]), "+v"(fb_lo[k][1]), "+v"(fb_hi[k][1]) :: "memory")
#define SEW3_MMA(k)                                                                                             \
  do {                                                                                                          \
    bf16x8 bq_[2];                                                                                              \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                               \
      bq_[j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(fb_lo[k][j], fb_hi[k][j], 0, 1, 2, 3));       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                             \
      const bf16x8 aq_ = __builtin_bit_cast(bf16x8, __builtin_shufflevector(fa_lo[k][i], fa_hi[k][i], 0, 1, 2, 3)); \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                             \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq_, bq_[j], acc[i][j], 0, 0, 0);                   \
    }                                                                                                           \
  } while (0)
  if constexpr (STAG == 3) {
    if (nt > 0) SEW3_READ(0, lds_base, 0);          // tile 0 is certified (the barrier above)
  }
  for (int t = 0; t < nt; ++t) {
    SEW_STAMP();
    if (!STAG) {
      // this wave's pieces of tile t landed; tiles t+1, t+2 stay in flight
      if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (t + 1 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      SEW_STAMP();
      __builtin_amdgcn_s_barrier();        // everyone's pieces landed AND everyone finished the stage refilled below
    } else if (!late) {
      // early half, barrier t at the step top: its pieces of tile t + 1 landed (issued so far: .. t + 2)
      if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      SEW_STAMP();
      __builtin_amdgcn_s_barrier();
    }
    SEW_STAMP();
    const char* sb = smem + st * kWStage;
#define SEW_READ(af_, bfr_, sb_, s_)                                                                            \
    do {                                                                                                        \
      _Pragma("unroll") for (int blk = 0; blk < 4; ++blk) {                                                     \
        const int off = a_base + (blk >> 1) * kWPanel + (s_) * 2048;                                            \
        const bf16x4 lo = SE_WTR((sb_) + off + toff[blk & 1][0]);                                               \
        const bf16x4 hi = SE_WTR((sb_) + off + toff[blk & 1][1]);                                               \
        af_[blk] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};                            \
      }                                                                                                         \
      _Pragma("unroll") for (int blk = 0; blk < 2; ++blk) {                                                     \
        const bf16x4 lo = SE_WTR((sb_) + b_base + (s_) * 2048 + toff[blk][0]);                                  \
        const bf16x4 hi = SE_WTR((sb_) + b_base + (s_) * 2048 + toff[blk][1]);                                  \
        bfr_[blk] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};                           \
      }                                                                                                         \
    } while (0)
#define SEW_MMA(af_, bfr_)                                                                                      \
    do {                                                                                                        \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                             \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                           \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af_[i], bfr_[j], acc[i][j], 0, 0, 0);             \
    } while (0)
#define SEW_REFILL()                                                                                            \
    do {                                                                                                        \
      /* refill: tile t + 3 into the slot of stage t - 1 (STAG: t - 2), free since the last barrier this wave passed.  Issued behind the sub-step's \
         fragment reads: every global_load_lds stalls the issuing wave ~110 clk, which overlaps the LDS pipe serving those reads */ \
      __builtin_amdgcn_sched_barrier(0);                                                                        \
      if (t + 3 < nt) { int sn = st + 3; if (sn >= kRing) sn -= kRing; SEW_ISSUE(t + 3, sn); }                  \
      __builtin_amdgcn_sched_barrier(0);                                                                        \
      SEW_STAMP();                                                                                              \
    } while (0)
#define SEW_LATE_BARRIER()                                                                                      \
    do {                                                                                                        \
      if (STAG && late) {                                                                                       \
        /* late half, barrier t in mid-step: its pieces of tile t + 1 landed (issued so far: .. t + 3) */       \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        const int after = (t + 2 < nt ? 1 : 0) + (t + 3 < nt ? 1 : 0);                                          \
        if (after == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                        \
        else if (after == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                   \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                   \
        __builtin_amdgcn_s_barrier();                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
      }                                                                                                         \
    } while (0)
    if constexpr (STAG == 3) {
      // STAG 3 (A/B, not the default: 3-5 % SLOWER than STAG 1 -- FFN 181-186 / 167-171 us against 174-177 / 162-165, fine-tune step 19.5 against 19.2 ms,
      // profiles/r04_wgrad_stag3.txt -- so the fragment-read latency inside a wave is not what the step waits for once two staggered waves share a SIMD):
      // the software pipeline the compiler would not keep -- the fragment reads are inline asm (the compiler neither waits for them nor
      // sees them as LDS traffic), the waits are written by hand with the fragment registers as operands (so no MFMA moves above its wait), and
      // while one sub-step's 8 MFMAs run, the NEXT sub-step's 12 transposed reads are already in the LDS queue: set 0 = (t, sub-step 0), read under the
      // previous step's second sub-step; set 1 = (t, 1), read here.  A read only writes registers whose last MFMA has issued (in-order issue; the
      // data returns >= 64 cycles later, the MFMA takes its operands in its first cycles).
      const uint32_t sbase = lds_base + (uint32_t)(st * kWStage);
      SEW3_READ(1, sbase, 1);
      SEW_REFILL();
      SEW3_WAIT(0, "12");                  // set 0 landed; the 12 reads of set 1 may stay in flight
      SEW3_MMA(0);
      SEW_LATE_BARRIER();
      {
        // tile t + 1: certified by barrier t, which this wave has passed.  UNCONDITIONAL (after the last tile it reads a stage nobody needs): a branch
        // here made the compiler merge the two paths' fragment registers with copies placed in FRONT of one path's wait
        const uint32_t sbn = lds_base + (uint32_t)((st + 1 == kRing ? 0 : st + 1) * kWStage);
        SEW3_READ(0, sbn, 0);
        SEW3_WAIT(1, "12");
      }
      SEW3_MMA(1);
    } else
    if constexpr (!PIPE) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 af[4], bfr[2];
        SEW_READ(af, bfr, sb, s);
        if (s == 0) SEW_REFILL();
        SEW_MMA(af, bfr);
        if (s == 0) SEW_LATE_BARRIER();
      }
    } else {
      // STAG 2 (A/B, not the default: a tie with STAG 1, FFN 174-181 / 158-161 us against 175-178 / 163-168, fine-tune step 19.75 against 19.52 ms):
      // the refill is issued IN FRONT of the sub-step's reads and the X fragments are requested first, so nothing stands between the reads and the
      // MFMAs and the compiler can count the LDS returns (lgkmcnt(n)) instead of draining them: the first MFMA pair starts when 4 of 6 fragments are in.
      // Measured against two other orders on one box (profiles/r04_wgrad_stag2.txt): BOTH sub-steps' 24 reads at the step top, one wait per 16 MFMAs:
      // 12 % SLOWER (FFN 195 / 177 us against 170 / 158) -- the wait gets longer, not rarer; the next sub-step's reads under this one's MFMAs (a software
      // pipeline) does not survive the compiler: its s_waitcnt lgkmcnt(0) before a block's first MFMA also waits for reads issued just before it, and
      // pinning reads between MFMAs with sched_barrier / sched_group_barrier sent the register allocator to 256 registers + 163 spills.
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if (s == 0) SEW_REFILL();
        bf16x8 af[4], bfr[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
          const bf16x4 lo = SE_WTR(sb + b_base + s * 2048 + toff[blk][0]);
          const bf16x4 hi = SE_WTR(sb + b_base + s * 2048 + toff[blk][1]);
          bfr[blk] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
          const int off = a_base + (blk >> 1) * kWPanel + s * 2048;
          const bf16x4 lo = SE_WTR(sb + off + toff[blk & 1][0]);
          const bf16x4 hi = SE_WTR(sb + off + toff[blk & 1][1]);
          af[blk] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
        SEW_MMA(af, bfr);
        if (s == 0) SEW_LATE_BARRIER();
      }
    }
    st = st + 1 == kRing ? 0 : st + 1;
  }
  if constexpr (STAG == 3) SEW3_WAIT(0, "0");      // the last step's look-ahead reads have written their registers before anything reuses them
  SE_CLKPROBE_END(clkprobe_wgrad);
  SEW_STAMP();
#undef SEW_ISSUE
#undef SEW3_MMA
#undef SEW3_WAIT
#undef SEW3_READ
#undef SEW3_RD
#undef SEW_READ
#undef SEW_MMA
#undef SEW_REFILL
#undef SEW_LATE_BARRIER

  // ---- epilogue: acc[i][j][r] = dW[n0 + 128 wn + 32 i + (r&3) + 8 (r>>2) + 4 hh][k0 + 64 wk + 32 j + l31]
  float* out = partials + (size_t)split * N * K;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kc = k0 + 64 * wk + 32 * j + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + 128 * wn + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hh;
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

static int wgrad_stag() {
  static int v = -1;
  // A/B: 0 = every wave's barrier at the step top (four-stage ring), 1 = waves 4-7 half a step behind (five stages), 2 = 1 with the
  // refill in front of the sub-step's reads, X fragments first (counted lgkmcnt waits), 3 = 1 with asm fragment reads pipelined across sub-steps
  if (v < 0) { const char* e = getenv("SE_AMD_WGRAD_STAG"); v = e ? atoi(e) : 1; if (v < 0 || v > 3) v = 1; }
  return v;
}

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
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::wgrad_tn_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, se::kWLds));
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::wgrad_tn_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, se::kWLdsStag));
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::wgrad_tn_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, se::kWLdsStag));
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::wgrad_tn_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, se::kWLdsStag));
    attr_set = true;
  }
  const int tiles_n = (N + se::kWN - 1) / se::kWN, tiles_k = (K + se::kWK - 1) / se::kWK;
  const int tiles_m = (M + se::kWM - 1) / se::kWM;
  const int m_per_split = (tiles_m + splits - 1) / splits * se::kWM;
  const int tiles = tiles_n * tiles_k, work = tiles * splits;
  float* partials = reinterpret_cast<float*>(workspace);
  static int stamp_env = -1;
  if (stamp_env < 0) {
    const char* e = getenv("SE_AMD_WGRAD_STAMPS");       // developer switch: the 128 KiB after the slabs receive cycle stamps
    stamp_env = e ? atoi(e) : 0;
  }
  unsigned long long* stamps = nullptr;
  if (stamp_env) {
    SE_REQUIRE(workspace_bytes >= (size_t)splits * N * K * sizeof(float) + 8 * 8 * 256 * 8, "se_wgrad_tn_bf16: stamp build needs 128 KiB more workspace");
    stamps = reinterpret_cast<unsigned long long*>(partials + (size_t)splits * N * K);
  }
  {
    se::ProfScope prof(se::kProfGemm, 2.0 * M * (double)N * K, st);
    if (wgrad_stag() == 3) hipLaunchKernelGGL(se::wgrad_tn_kernel<3>, dim3(work), dim3(512), se::kWLdsStag, st, dY, ldy, X, ldx, M, N, K, m_per_split, tiles_k,
                                              tiles, work, partials, stamps);
    else if (wgrad_stag() == 2) hipLaunchKernelGGL(se::wgrad_tn_kernel<2>, dim3(work), dim3(512), se::kWLdsStag, st, dY, ldy, X, ldx, M, N, K, m_per_split, tiles_k,
                                              tiles, work, partials, stamps);
    else if (wgrad_stag()) hipLaunchKernelGGL(se::wgrad_tn_kernel<1>, dim3(work), dim3(512), se::kWLdsStag, st, dY, ldy, X, ldx, M, N, K, m_per_split, tiles_k, tiles,
                                         work, partials, stamps);
    else hipLaunchKernelGGL(se::wgrad_tn_kernel<0>, dim3(work), dim3(512), se::kWLds, st, dY, ldy, X, ldx, M, N, K, m_per_split, tiles_k, tiles, work,
                            partials, stamps);
    SE_LAUNCH_CHECK();
  }
  const size_t n4 = (size_t)N * K / 4;
  hipLaunchKernelGGL(se::wgrad_slab_reduce_kernel, dim3((unsigned)std::min<size_t>((n4 + 255) / 256, 4096)), dim3(256), 0, st, partials, splits,
                     n4, accumulate, dW);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

// Per-group weight gradients: slabs[g][N][K] = dY[g R .. (g+1) R)^T . X[same rows]  for g = 0 .. ceil(M / R) - 1 -- the m-split
// slabs of the kernel above with the split boundaries put on utterance boundaries (R = frames per utterance) and NO reduce:
// one launch gives every utterance's own gradient (active-sampling scoring, sampler.py:59-111, which the reference obtains by
// B sequential backward passes with retain_graph).
extern "C" int se_wgrad_tn_slabs_bf16(const uint16_t* dY, int ldy, const uint16_t* X, int ldx, int M, int N, int K, int rows_per_slab,
                                      float* slabs, void* stream) {
  SE_REQUIRE(dY && X && slabs, "se_wgrad_tn_slabs_bf16: null argument");
  SE_REQUIRE(M > 0 && N > 0 && K > 0 && rows_per_slab > 0, "se_wgrad_tn_slabs_bf16: bad shape");
  SE_REQUIRE(N % 8 == 0 && K % 8 == 0 && ldy % 8 == 0 && ldx % 8 == 0 && ldy >= N && ldx >= K,
             "se_wgrad_tn_slabs_bf16: N, K and the leading dimensions must be multiples of 8");
  SE_REQUIRE((((uintptr_t)dY | (uintptr_t)X | (uintptr_t)slabs) % 16) == 0, "se_wgrad_tn_slabs_bf16: operands must be 16-B aligned");
  const int groups = (M + rows_per_slab - 1) / rows_per_slab;
  SE_REQUIRE(groups <= 65535, "se_wgrad_tn_slabs_bf16: too many groups");
  hipStream_t st = se::as_stream(stream);
  static bool attr_set = false;
  if (!attr_set) {
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::wgrad_tn_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, se::kWLds));
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::wgrad_tn_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, se::kWLdsStag));
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::wgrad_tn_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, se::kWLdsStag));
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::wgrad_tn_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, se::kWLdsStag));
    attr_set = true;
  }
  const int tiles_n = (N + se::kWN - 1) / se::kWN, tiles_k = (K + se::kWK - 1) / se::kWK;
  const int tiles = tiles_n * tiles_k, work = tiles * groups;
  se::ProfScope prof(se::kProfGemm, 2.0 * M * (double)N * K, st);
  if (wgrad_stag() == 3) hipLaunchKernelGGL(se::wgrad_tn_kernel<3>, dim3(work), dim3(512), se::kWLdsStag, st, dY, ldy, X, ldx, M, N, K, rows_per_slab, tiles_k, tiles,
                                       work, slabs, (unsigned long long*)nullptr);
  else if (wgrad_stag() == 2) hipLaunchKernelGGL(se::wgrad_tn_kernel<2>, dim3(work), dim3(512), se::kWLdsStag, st, dY, ldy, X, ldx, M, N, K, rows_per_slab, tiles_k, tiles,
                                       work, slabs, (unsigned long long*)nullptr);
  else if (wgrad_stag()) hipLaunchKernelGGL(se::wgrad_tn_kernel<1>, dim3(work), dim3(512), se::kWLdsStag, st, dY, ldy, X, ldx, M, N, K, rows_per_slab, tiles_k, tiles,
                                       work, slabs, (unsigned long long*)nullptr);
  else hipLaunchKernelGGL(se::wgrad_tn_kernel<0>, dim3(work), dim3(512), se::kWLds, st, dY, ldy, X, ldx, M, N, K, rows_per_slab, tiles_k, tiles, work,
                          slabs, (unsigned long long*)nullptr);
  SE_LAUNCH_CHECK();
  return SE_OK;
}
