// gemm6q.hip -- the persistent 256 x 256 x 64 eight-phase bf16 GEMM (gemm6.hip: gemm6p_bf16_kernel) with its epilogue PIPELINED INTO THE K LOOP
// (round 5; C = act(A . W^T + bias) as bf16 rows: the encoder's QKV and FFN1 projections)
//
// What round 4 measured on gemm6p (profiles/r04_gemm6p_ablation.txt): 27 of 111 us (QKV) and 47 of 152 us (FFN1 + GELU) are the epilogue, fully
// exposed: at a tile's end both wave groups re-align, every wave converts its 128 accumulators and issues 16 stores back to back with the matrix
// pipe idle, and the counted waits of the next tile's first K-tiles then also wait for those stores (vmcnt retires in order).
//
// Here no wave ever stops multiplying for the epilogue:
//   * a wave's 128 x 64 output is four 64 x 32 quadrants that the K-tile's four phases visit in the order (0,0) (0,1) (1,1) (1,0).  In the LAST
//     K-tile of an output tile quadrant (0,0) is therefore final after phase 1, (0,1) after phase 2, ... and none of them is written again before
//     the same phase of the NEXT tile's first K-tile.  The epilogue is cut into small pieces -- the activation of ONE MFMA tile in place; the bf16
//     conversion + ONE store of a 16-row tile pair -- placed by tools/gen_gemm6q_sched.py (gemm6q_sched.h) into the 13 sections that lie between:
//     in READ sections behind the LDS-DMA issue (they run while the fragment reads are in flight, beside the SIMD partner's MFMAs) and in MFMA
//     sections ONE PIECE AFTER EVERY SECOND MFMA.  (First version: a quadrant's pieces as a block behind eight MFMAs -- the wave issues in order, so
//     only the last MFMA's 16 cycles shadow the block: every vector / scalar instruction of it delayed the next MFMAs;
//     profiles/r05_gemm6q_steps.txt.)
//   * the accumulators are not cleared: the first K-tile's MFMAs take the BIAS as their C operand (so the epilogue has no bias add either);
//   * the wave groups keep their one-barrier stagger across tile boundaries (no re-alignment), the LDS-DMA stream is continuous as before;
//   * NO LANE EXCHANGE: the B fragments are read with a COLUMN PERMUTATION -- MFMA row n of column tile j (of half h) is output column
//     32 h + 8 (n >> 2) + 4 j + (n & 3) of the wave tile -- so a lane's accumulators of the two tiles of a half are 8 CONSECUTIVE columns: one
//     16-B store per lane, a wave instruction = 16 rows x 64 B, straight from the converted registers.  The B slots' 16-B chunk swizzle becomes
//     chunk ^ ((row & 3) | ((row >> 1) & 4)) (conflict-free for these reads: tools/micro/swizzle_check.py).
//     (Whole-line stores -- 8 rows x 128 B through a DPP lane ^ 8 exchange -- were built and measured too: a CU that is bound by its vector-memory
//     path stores whole lines 4 x faster than half lines (tools/micro/store_cu.hip), but this kernel is not, and the extra 12 vector instructions
//     per 16 rows cost more than the store shape gained: 110-112 vs 107 us.)
//   * the stores are inline asm (SGPR base + 32-bit lane offset; two wait states behind each -- the compiler cannot see that a 16-B store still
//     reads its data registers when the next vector instruction overwrites them: lanes 12-15 of every row of 16 stored the NEXT piece's first
//     register until the s_nop went in) and every wait of the boundary K-tiles counts them by hand (gemm6q_sched.h):
//       wait of K-tile t (phase 4) = vmcnt(6 + stores issued after B_0 (t + 1))   [everything up to B_0 (t + 1) has landed]
//   * ONE instruction stream for every tile, so that the counts are static.  PAD = true (the caller's output buffer has ceil(M / 256) x 256 rows:
//     the encoder's workspace): rows past M are simply written.  PAD = false: rows past M are a wave-uniform matter per store (its 16 rows are all
//     there, all missing, or cut) -- a missing group's store goes to a scratch dump instead of the output (SGPR base select), a cut group's store
//     runs under an EXEC mask of its existing rows.  The first tile's K-tile 0 (nothing to finish yet) is peeled in front of the tile loop; the
//     last tile's pieces ride a phantom K-tile behind it.
// A-fragment layout, LDS ring, DMA stream and tile order are gemm6p's (gemm6.hip); results differ from it only by the place of the bias in the
// fp32 sum.
#include "clkprobe.h"
#include <stdlib.h>
#include "common.h"
#include "bf16.h"
#include "prof.h"

SE_CLKPROBE_DECL(clkprobe_gemm6q)
namespace se {

constexpr int kqBM = 256, kqBN = 256, kqBK = 64, kqThreads = 512;
constexpr int kqSlot = 128 * 128, kqBuf = 4 * kqSlot, kqLds = 2 * kqBuf;       // 16 KiB, 64 KiB, 128 KiB
constexpr int kqA0 = 0, kqA1 = kqSlot, kqB0 = 2 * kqSlot, kqB1 = 3 * kqSlot;  // slot offsets inside a buffer

typedef __attribute__((address_space(3))) void* ldsq_ptr_t;
typedef unsigned int u32x4q __attribute__((ext_vector_type(4)));

// -DSE6Q_ABL=<mask> (timing only, results wrong): 1 no store instructions (the conversion stays), 2 no epilogue pieces at all, 4 every store goes to the
// dump (L2-resident: the store instructions stay, their HBM traffic goes), 8 the waits of the two K-tiles behind a tile boundary do not cover the stores
// (nor, then, the youngest DMA pieces: wrong data, but no wave ever waits for a store)
#ifndef SE6Q_ABL
#define SE6Q_ABL 0
#endif
// placement of the micro-steps: 0 = between the MFMA halves of the phases that follow (default), 1 = all of a quadrant's four right after the phase
// that finishes it (A/B), 2 = in the READ sections that follow
#ifndef SE6Q_PLACE
#define SE6Q_PLACE 0
#endif
// store shape: 0 = 16 rows x 64 B per wave instruction straight from the converted registers, 1 = 8 rows x 128 B (whole lines) through a DPP lane ^ 8
// exchange; the piece placement follows (gemm6q_sched.h)
#ifndef SE6Q_WL
#define SE6Q_WL 1
#endif
#include "gemm6q_sched.h"

template <int ACT, bool PAD>
__global__ __launch_bounds__(kqThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm6q_bf16_kernel(
    const uint16_t* __restrict__ A, int lda, const uint16_t* __restrict__ W, int ldw, const float* __restrict__ bias, int M, int N, int K,
    uint16_t* __restrict__ out_bf16, int ldc, int tiles_m, int tiles_n, int group_m, int late_start, char* __restrict__ dump) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  SE_CLKPROBE_BEGIN();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // ---- this workgroup's tile list (gemm6p's: XCD x owns a contiguous id range, its workgroups take ids start + slot + wpx i)
  const int nwg = tiles_m * tiles_n;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wpx = gridDim.x >> 3;
  const int q = nwg >> 3, rem = nwg & 7;
  const int range_start = (xcd < rem) ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
  const int range_count = q + (xcd < rem ? 1 : 0);
  const int my_tiles = (range_count > slot) ? (range_count - slot + wpx - 1) / wpx : 0;
  if (my_tiles == 0) return;
  if (late_start > 0 && my_tiles < (range_count + wpx - 1) / wpx)
    for (int i = 0; i < late_start; ++i) __builtin_amdgcn_s_sleep(127);

  uint32_t a_of[2][2], b_of[2][2];
#define SEQ_SET_SRC(id_, m0_, n0_)                                                                                         \
  do {                                                                                                                     \
    const int per_group_ = group_m * tiles_n, grp_ = (id_) / per_group_, first_m_ = grp_ * group_m;                        \
    const int gsz_ = min(tiles_m - first_m_, group_m), in_ = (id_) - grp_ * per_group_;                                    \
    const int tn_ = in_ / gsz_, tm_ = first_m_ + (in_ - tn_ * gsz_);                                                       \
    m0_ = tm_ * kqBM;                                                                                                      \
    n0_ = tn_ * kqBN;                                                                                                      \
    int ln_ = lane;                                                                                                        \
    asm volatile("" : "+v"(ln_));     /* opaque: keeps the lane-derived terms from being hoisted out of the tile loop and spilled */ \
    _Pragma("unroll") for (int p_ = 0; p_ < 2; ++p_) {                                                                     \
      const int rho_ = 8 * (wave + 8 * p_) + (ln_ >> 3);                                                                   \
      const int lc_ = ((ln_ & 7) ^ ((rho_ >> 1) & 7)) << 3;            /* A slots: chunk ^ ((row >> 1) & 7) */                \
      const int lb_ = ((ln_ & 7) ^ ((rho_ & 3) | ((rho_ >> 1) & 4))) << 3;   /* B slots: chunk ^ ((row & 3) | ((row >> 1) & 4)) */  \
      _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) {                                                                   \
        const int trow_ = (rho_ >> 6) * 128 + h_ * 64 + (rho_ & 63);                                                       \
        const int tcol_ = (rho_ >> 5) * 64 + h_ * 32 + (rho_ & 31);                                                        \
        a_of[h_][p_] = (uint32_t)(min(m0_ + trow_, M - 1) * lda + lc_) * 2u;                                               \
        b_of[h_][p_] = (uint32_t)(min(n0_ + tcol_, N - 1) * ldw + lb_) * 2u;                                               \
      }                                                                                                                    \
    }                                                                                                                      \
  } while (0)
  const uint32_t lds_wave = (uint32_t)(size_t)(ldsq_ptr_t)smem + wave * 1024;
#define SEQ_DMA(base, of, slot_off, buf, kt)                                                                               \
  do {                                                                                                                     \
    const char* sb_ = reinterpret_cast<const char*>(base) + (size_t)(kt) * (kqBK * 2);                                     \
    _Pragma("unroll") for (int p_ = 0; p_ < 2; ++p_) {                                                                     \
      uint32_t keep_;                                                                                                      \
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"      \
                   : "=&s"(keep_)                                                                                          \
                   : "v"(of[p_]), "s"(sb_), "s"(lds_wave + (uint32_t)((buf) * kqBuf + (slot_off) + p_ * 8192))             \
                   : "memory");                                                                                            \
    }                                                                                                                      \
  } while (0)

  const int frow = lane & 15, fch = lane >> 4;
  int a_ad[2], b_ad[2][2];                  // [k-slice] / [k-slice][column tile]: the B rows of the two tiles are interleaved (column permutation)
  {
    const int ra = wr * 64 + frow;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      a_ad[s] = ra * 128 + (((4 * s + fch) ^ ((ra >> 1) & 7)) << 4);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int rb = wc * 32 + 8 * (frow >> 2) + 4 * j + (frow & 3);
        b_ad[s][j] = rb * 128 + (((4 * s + fch) ^ ((rb & 3) | ((rb >> 1) & 4))) << 4);
      }
    }
  }
  bf16x8 af[4][2], bfr[2][2];
  f32x4 acc[8][4];
  f32x4 bb[2];                              // bias of this lane's 4 columns in the phase's two MFMA column tiles: the C operand of the first K-tile's MFMAs
#define SEQ_READ_A(buf, h)                                                                                                 \
  _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                        \
    af[i_][s_] = *reinterpret_cast<const bf16x8*>(smem + (buf) * kqBuf + ((h) ? kqA1 : kqA0) + a_ad[s_] + i_ * 2048);
#define SEQ_READ_B(buf, h)                                                                                                 \
  _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                        \
    bfr[j_][s_] = *reinterpret_cast<const bf16x8*>(smem + (buf) * kqBuf + ((h) ? kqB1 : kqB0) + b_ad[s_][j_]);
  // a phase's 16 MFMAs: k-slice 0 then 1, row tiles i = 0..3, two column tiles each; FIRST (K-tile 0 of an output tile, slice 0): C = bias
  // instead of the old accumulator.  H0 .. H7: epilogue pieces, one hook point after every second MFMA.
#define SEQ_MMA2(mq, nq, s_, i_, FIRST)                                                                                    \
  _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                                         \
    acc[(mq) * 4 + (i_)][(nq) * 2 + j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                                         \
        bfr[j_][s_], af[i_][s_], ((FIRST) && (s_) == 0) ? bb[j_] : acc[(mq) * 4 + (i_)][(nq) * 2 + j_], 0, 0, 0);
#define SEQ_SB __builtin_amdgcn_sched_barrier(0);
#define SEQ_MSEC(mq, nq, FIRST, TAG, PH)                                                                                   \
  SEQ_MMA2(mq, nq, 0, 0, FIRST) SEQ_SB SEQ_##TAG##_P##PH##M0 SEQ_SB SEQ_MMA2(mq, nq, 0, 1, FIRST) SEQ_SB SEQ_##TAG##_P##PH##M1 SEQ_SB    \
  SEQ_MMA2(mq, nq, 0, 2, FIRST) SEQ_SB SEQ_##TAG##_P##PH##M2 SEQ_SB SEQ_MMA2(mq, nq, 0, 3, FIRST) SEQ_SB SEQ_##TAG##_P##PH##M3 SEQ_SB    \
  SEQ_MMA2(mq, nq, 1, 0, FIRST) SEQ_SB SEQ_##TAG##_P##PH##M4 SEQ_SB SEQ_MMA2(mq, nq, 1, 1, FIRST) SEQ_SB SEQ_##TAG##_P##PH##M5 SEQ_SB    \
  SEQ_MMA2(mq, nq, 1, 2, FIRST) SEQ_SB SEQ_##TAG##_P##PH##M6 SEQ_SB SEQ_MMA2(mq, nq, 1, 3, FIRST) SEQ_SB SEQ_##TAG##_P##PH##M7 SEQ_SB
#define SEQ_SYNC_A()                                                                                                       \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                       \
  __builtin_amdgcn_sched_barrier(0);                                                                                       \
  __builtin_amdgcn_s_barrier();                                                                                            \
  __builtin_amdgcn_sched_barrier(0);
#define SEQ_SYNC_B()                                                                                                       \
  __builtin_amdgcn_sched_barrier(0);                                                                                       \
  __builtin_amdgcn_s_barrier();                                                                                            \
  __builtin_amdgcn_sched_barrier(0);

  // ---- epilogue pieces.  C^T accumulators with the column permutation: lane (rho = lane & 15, g = lane >> 4) of acc[k][2 h + j] holds output row
  //      16 k + rho, columns 32 h + 8 g + 4 j + (0..3) of the wave tile.
  //      SEQ_ACT(mq, nq, b0, b1): the activation of the quadrant's MFMA tiles b0 .. b1 - 1 (b = 2 i + j), in place (nothing for the identity).
  //      SEQ_GH(mq, nq, i): rows 16 (4 mq + i) .. + 15, columns 32 nq .. + 31: four conversions, one store of 16 rows x 64 B.
  //      ep_voff = this lane's byte offset inside such a store: tile-independent; the tile enters through the SGPR base.
  uint32_t ep_voff;
  {
    int le_ = lane;
    asm volatile("" : "+v"(le_));
    ep_voff = (uint32_t)((le_ & 15) * ldc + wc * 64 + 8 * (le_ >> 4)) * 2u;
  }
  const uint32_t ep_row_step = (uint32_t)ldc * 32u;      // 16 rows, bytes
  const char* ep_base = nullptr;            // output address of the wave tile's row 0, column 0 (tile being finished)
  int ep_rows = 0;                          // how many of the wave tile's 128 rows exist (M - first row; may be <= 0 or >= 128)
#define SEQ_SET_EP(m0_, n0_)                                                                                               \
  do {                                                                                                                     \
    ep_base = reinterpret_cast<const char*>(out_bf16) + ((size_t)((m0_) + wr * 128) * ldc + (n0_)) * 2;                    \
    ep_rows = M - (m0_) - wr * 128;                                                                                        \
  } while (0)
#define SEQ_ACT(mq, nq, b0, b1)                                                                                            \
  if constexpr (ACT == SE_ACT_GELU && !(SE6Q_ABL & 2)) {                                                                   \
    _Pragma("unroll") for (int b_ = (b0); b_ < (b1); ++b_) {                                                               \
      f32x4& c_ = acc[(mq) * 4 + (b_ >> 1)][(nq) * 2 + (b_ & 1)];                                                          \
      const f32x2 g0_ = gelu_poly2((f32x2){c_[0], c_[1]}), g1_ = gelu_poly2((f32x2){c_[2], c_[3]});                        \
      c_ = (f32x4){g0_.x, g0_.y, g1_.x, g1_.y};                                                                            \
    }                                                                                                                      \
  }
#define SEQ_GH(mq, nq, i)                                                                                                  \
  if (SE6Q_ABL & 2) {        /* ablation: no epilogue work; the empty asm keeps the accumulators (and so the MFMAs) alive */ \
    asm volatile("" ::"v"(acc[(mq) * 4 + (i)][(nq) * 2]), "v"(acc[(mq) * 4 + (i)][(nq) * 2 + 1]));                         \
  } else {                                                                                                                 \
    const f32x4 c0_ = acc[(mq) * 4 + (i)][(nq) * 2], c1_ = acc[(mq) * 4 + (i)][(nq) * 2 + 1];                             \
    const u32x4q o_ = {pack_bf16x2(c0_[0], c0_[1]), pack_bf16x2(c0_[2], c0_[3]), pack_bf16x2(c1_[0], c1_[1]), pack_bf16x2(c1_[2], c1_[3])}; \
    if constexpr (PAD) {                                                                                                   \
      const char* sb_ = ep_base + (size_t)((mq) * 4 + (i)) * ep_row_step;                                                  \
      if (SE6Q_ABL & 1)                                                                                                    \
        asm volatile("" ::"v"(ep_voff), "v"(o_), "s"(sb_) : "memory");                                                     \
      else if ((nq) == 0)                                                                                                  \
        asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(ep_voff), "v"(o_), "s"(sb_) : "memory");             \
      else                                                                                                                 \
        asm volatile("global_store_dwordx4 %0, %1, %2 offset:64\n\ts_nop 1" ::"v"(ep_voff), "v"(o_), "s"(sb_) : "memory");  \
    } else {                                                                                                               \
      const int have_ = ep_rows - 16 * ((mq) * 4 + (i));      /* existing rows of this 16-row group (wave-uniform) */       \
      const char* sb_ = ((have_ > 0 && !(SE6Q_ABL & 4)) ? ep_base : dump) + (size_t)((mq) * 4 + (i)) * ep_row_step + (nq) * 64; \
      const unsigned long long em_ = (have_ <= 0 || have_ >= 16) ? ~0ull : ((1ull << have_) - 1ull) * 0x0001000100010001ull; \
      if (SE6Q_ABL & 1)                                                                                                    \
        asm volatile("" ::"v"(ep_voff), "v"(o_), "s"(sb_), "s"(em_) : "memory");                                           \
      else                                                                                                                 \
        asm volatile("s_mov_b64 exec, %3\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_mov_b64 exec, -1\n\ts_nop 0" ::"v"(ep_voff), "v"(o_), "s"(sb_), "s"(em_) : "memory"); \
    }                                                                                                                      \
  }
  // whole-line form.  SEQ_GW(mq, i): rows 16 (4 mq + i) .. + 15, all 64 columns: U = this lane's 8 columns of the left half (chunk g), V = of the
  // right half (chunk 4 + g); row_ror:8 = lane ^ 8 inside each row of 16 lanes, bank_mask picks the 4-lane banks that take the rotated value:
  //   X: rows 0-7  -- lanes rho < 8 own U, lanes rho >= 8 the V of lane rho - 8      Y: rows 8-15 -- lanes rho < 8 the U of lane rho + 8, lanes rho >= 8 own V
  // ep_voffw = this lane's byte offset inside a store's 8 rows x 128 B
  uint32_t ep_voffw;
  {
    int le_ = lane;
    asm volatile("" : "+v"(le_));
    ep_voffw = (uint32_t)((le_ & 7) * ldc + wc * 64 + (((le_ >> 3) & 1) * 4 + (le_ >> 4)) * 8) * 2u;
  }
#define SEQ_STORE8(r8_, o_)                                                                                                \
  do {                                                                                                                     \
    if constexpr (PAD) {                                                                                                   \
      const char* sb_ = ep_base + (size_t)(r8_) * (ep_row_step >> 1);                                                      \
      if (SE6Q_ABL & 1)                                                                                                    \
        asm volatile("" ::"v"(ep_voffw), "v"(o_), "s"(sb_) : "memory");                                                    \
      else                                                                                                                 \
        asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(ep_voffw), "v"(o_), "s"(sb_) : "memory");          \
    } else {                                                                                                               \
      const int have_ = ep_rows - 8 * (r8_);                                                                               \
      const char* sb_ = ((have_ > 0 && !(SE6Q_ABL & 4)) ? ep_base : dump) + (size_t)(r8_) * (ep_row_step >> 1);            \
      const unsigned long long em_ = (have_ <= 0 || have_ >= 8) ? ~0ull : ((1ull << have_) - 1ull) * 0x0101010101010101ull; \
      if (SE6Q_ABL & 1)                                                                                                    \
        asm volatile("" ::"v"(ep_voffw), "v"(o_), "s"(sb_), "s"(em_) : "memory");                                          \
      else                                                                                                                 \
        asm volatile("s_mov_b64 exec, %3\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_mov_b64 exec, -1\n\ts_nop 0" ::"v"(ep_voffw), "v"(o_), "s"(sb_), "s"(em_) : "memory"); \
    }                                                                                                                      \
  } while (0)
#define SEQ_GW(mq, i)                                                                                                      \
  if (SE6Q_ABL & 2) {                                                                                                      \
    asm volatile("" ::"v"(acc[(mq) * 4 + (i)][0]), "v"(acc[(mq) * 4 + (i)][1]));                                           \
    asm volatile("" ::"v"(acc[(mq) * 4 + (i)][2]), "v"(acc[(mq) * 4 + (i)][3]));                                           \
  } else {                                                                                                                 \
    const f32x4 c0_ = acc[(mq) * 4 + (i)][0], c1_ = acc[(mq) * 4 + (i)][1], c2_ = acc[(mq) * 4 + (i)][2], c3_ = acc[(mq) * 4 + (i)][3]; \
    const uint32_t u_[4] = {pack_bf16x2(c0_[0], c0_[1]), pack_bf16x2(c0_[2], c0_[3]), pack_bf16x2(c1_[0], c1_[1]), pack_bf16x2(c1_[2], c1_[3])}; \
    const uint32_t v_[4] = {pack_bf16x2(c2_[0], c2_[1]), pack_bf16x2(c2_[2], c2_[3]), pack_bf16x2(c3_[0], c3_[1]), pack_bf16x2(c3_[2], c3_[3])}; \
    u32x4q x_, y_;                                                                                                         \
    _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) {                                                                     \
      x_[r_] = (uint32_t)__builtin_amdgcn_update_dpp((int)u_[r_], (int)v_[r_], 0x128, 0xf, 0xc, false);                    \
      y_[r_] = (uint32_t)__builtin_amdgcn_update_dpp((int)v_[r_], (int)u_[r_], 0x128, 0xf, 0x3, false);                    \
    }                                                                                                                      \
    SEQ_STORE8(2 * ((mq) * 4 + (i)), x_);                                                                                  \
    SEQ_STORE8(2 * ((mq) * 4 + (i)) + 1, y_);                                                                              \
  }
#define SEQ_NOP ((void)0)
  // one K-tile from buffer BUF.  I1..I4: the LDS-DMA issue of each phase's read section; WAIT: the phase-4 wait; FIRST: K-tile 0 of an output tile;
  // TAG: which set of epilogue pieces (gemm6q_sched.h: P = none, L = last K-tile of a tile, N = first K-tile of the next); X4: an extra statement in
  // phase 4's read section
#define SEQ_KT(BUF, I1, I2, I3, I4, WAIT, FIRST, TAG, X4)                                                                  \
  {                                                                                                                        \
    SEQ_READ_B(BUF, 0)                                                                                                     \
    SEQ_READ_A(BUF, 0)                                                                                                     \
    if (FIRST) SEQ_LOAD_BB(0);                                                                                             \
    I1;                                                                                                                    \
    SEQ_##TAG##_P1R;                                                                                                       \
    SEQ_SYNC_A()                                                                                                           \
    SEQ_MSEC(0, 0, FIRST, TAG, 1)                                                                                          \
    SEQ_SYNC_B()                                                                                                           \
    SEQ_READ_B(BUF, 1)                                                                                                     \
    if (FIRST) SEQ_LOAD_BB(1);                                                                                             \
    I2;                                                                                                                    \
    SEQ_##TAG##_P2R;                                                                                                       \
    SEQ_SYNC_A()                                                                                                           \
    SEQ_MSEC(0, 1, FIRST, TAG, 2)                                                                                          \
    SEQ_SYNC_B()                                                                                                           \
    SEQ_READ_A(BUF, 1)                                                                                                     \
    if (FIRST) SEQ_LOAD_BB(1);                                                                                             \
    I3;                                                                                                                    \
    SEQ_##TAG##_P3R;                                                                                                       \
    SEQ_SYNC_A()                                                                                                           \
    SEQ_MSEC(1, 1, FIRST, TAG, 3)                                                                                          \
    SEQ_SYNC_B()                                                                                                           \
    SEQ_READ_B(BUF, 0)                                                                                                     \
    if (FIRST) SEQ_LOAD_BB(0);                                                                                             \
    I4;                                                                                                                    \
    X4;                                                                                                                    \
    SEQ_##TAG##_P4R;                                                                                                       \
    WAIT;                                                                                                                  \
    SEQ_SYNC_A()                                                                                                           \
    SEQ_MSEC(1, 0, FIRST, TAG, 4)                                                                                          \
    SEQ_SYNC_B()                                                                                                           \
  }
#define SEQ_W_(n_) asm volatile("s_waitcnt vmcnt(" #n_ ")" ::: "memory")
#define SEQ_W(n_) SEQ_W_(n_)

  // bias of every output column, staged once in the LDS beside the ring (no compiler-visible VMEM load inside the tile loop)
  float* bias_lds = reinterpret_cast<float*>(smem + kqLds);
  for (int c = tid; c < N; c += kqThreads) bias_lds[c] = bias ? bias[c] : 0.f;
  __syncthreads();
#define SEQ_LOAD_BB(nq_)                                                                                                  \
  do {                                                                                                                     \
    int lb_ = lane;                                                                                                        \
    asm volatile("" : "+v"(lb_));                                                                                          \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                                       \
      bb[j_] = *reinterpret_cast<const f32x4*>(bias_lds + cn0 + wc * 64 + (nq_) * 32 + 8 * (lb_ >> 4) + 4 * j_);           \
  } while (0)

  const int nk = K / kqBK;                  // even, >= 4 (launcher)
  int tile_id = range_start + slot;
  int m0, n0;                               // the tile whose K-tiles are being ISSUED (moves on two K-tiles before the computed tile does)
  int cm0, cn0;                             // the tile being COMPUTED
  SEQ_SET_SRC(tile_id, m0, n0);
  // ---- prologue: K-tile 0 and A_0 / B_1 / A_1 of K-tile 1 of the first tile (B_0 (1) is issued by K-tile 0's phase 1 like every later one)
  SEQ_DMA(A, a_of[0], kqA0, 0, 0);
  SEQ_DMA(W, b_of[0], kqB0, 0, 0);
  SEQ_DMA(W, b_of[1], kqB1, 0, 0);
  SEQ_DMA(A, a_of[1], kqA1, 0, 0);
  SEQ_DMA(A, a_of[0], kqA0, 1, 1);
  SEQ_DMA(W, b_of[1], kqB1, 1, 1);
  SEQ_DMA(A, a_of[1], kqA1, 1, 1);
  SEQ_W(6);
  __builtin_amdgcn_s_barrier();
  const bool late = wave >= 4;
  if (late) __builtin_amdgcn_s_barrier();                  // stagger: waves 4-7 run one barrier behind, from here to the kernel's end

  cm0 = m0;
  cn0 = n0;
  // K-tile 0 of the first tile (C = bias), peeled: nothing to finish yet, the store-free wait
  SEQ_KT(0, SEQ_DMA(W, b_of[0], kqB0, 1, 1), SEQ_DMA(A, a_of[0], kqA0, 0, 2), SEQ_DMA(W, b_of[1], kqB1, 0, 2), SEQ_DMA(A, a_of[1], kqA1, 0, 2), SEQ_W(6), true,
         P, SEQ_NOP)
  for (int ti = 0; ti < my_tiles; ++ti) {
    const bool has_next = ti + 1 < my_tiles;
    SEQ_KT(1, SEQ_DMA(W, b_of[0], kqB0, 0, 2), SEQ_DMA(A, a_of[0], kqA0, 1, 3), SEQ_DMA(W, b_of[1], kqB1, 1, 3), SEQ_DMA(A, a_of[1], kqA1, 1, 3), SEQ_W(6),
           false, P, SEQ_NOP)
    for (int t = 2; t + 2 < nk; t += 2) {
      SEQ_KT(0, SEQ_DMA(W, b_of[0], kqB0, 1, t + 1), SEQ_DMA(A, a_of[0], kqA0, 0, t + 2), SEQ_DMA(W, b_of[1], kqB1, 0, t + 2),
             SEQ_DMA(A, a_of[1], kqA1, 0, t + 2), SEQ_W(6), false, P, SEQ_NOP)
      SEQ_KT(1, SEQ_DMA(W, b_of[0], kqB0, 0, t + 2), SEQ_DMA(A, a_of[0], kqA0, 1, t + 3), SEQ_DMA(W, b_of[1], kqB1, 1, t + 3),
             SEQ_DMA(A, a_of[1], kqA1, 1, t + 3), SEQ_W(6), false, P, SEQ_NOP)
    }
    // last two K-tiles: after B_0 (nk - 1) every issue belongs to the NEXT output tile, so the eight source offsets are rewritten in place (the
    // last tile of the list "prefetches" itself: one instruction stream; those bytes are never read and are drained before the kernel ends)
    const int next_id = has_next ? tile_id + wpx : tile_id;
    SEQ_KT(0, SEQ_DMA(W, b_of[0], kqB0, 1, nk - 1); tile_id = next_id; SEQ_SET_SRC(tile_id, m0, n0), SEQ_DMA(A, a_of[0], kqA0, 0, 0),
           SEQ_DMA(W, b_of[1], kqB1, 0, 0), SEQ_DMA(A, a_of[1], kqA1, 0, 0), SEQ_W(6), false, P, SEQ_SET_EP(cm0, cn0))
    // last K-tile: the pieces of gemm6q_sched.h's tag L; its wait counts the stores issued before it
    SEQ_KT(1, SEQ_DMA(W, b_of[0], kqB0, 0, 0), SEQ_DMA(A, a_of[0], kqA0, 1, 1), SEQ_DMA(W, b_of[1], kqB1, 1, 1), SEQ_DMA(A, a_of[1], kqA1, 1, 1),
           SEQ_W(SEQ_WAIT_L), false, L, SEQ_NOP)
    cm0 = m0;
    cn0 = n0;
    // K-tile 0 of the next tile (C = bias) with the rest of this tile's epilogue (tag N).  After the LAST tile this is a phantom K-tile (the tile
    // "prefetched" itself; its products are never stored): an exit path that finished the last tile's pieces outside the loop made the register
    // allocator spill 30-50 accumulator registers around every tile boundary, which costs more than one K-tile in ~50 per workgroup
    SEQ_KT(0, SEQ_DMA(W, b_of[0], kqB0, 1, 1), SEQ_DMA(A, a_of[0], kqA0, 0, 2), SEQ_DMA(W, b_of[1], kqB1, 0, 2), SEQ_DMA(A, a_of[1], kqA1, 0, 2),
           SEQ_W(SEQ_WAIT_N), true, N, SEQ_NOP)
  }
  if (!late) __builtin_amdgcn_s_barrier();                 // barrier counts of the two wave groups must match
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // no LDS-DMA may outlive the workgroup's LDS allocation
  SE_CLKPROBE_END(clkprobe_gemm6q);
#undef SEQ_KT
#undef SEQ_GH
#undef SEQ_GW
#undef SEQ_STORE8
#undef SEQ_ACT
#undef SEQ_MSEC
#undef SEQ_MMA2
#undef SEQ_SET_SRC
#undef SEQ_DMA
#undef SEQ_READ_A
#undef SEQ_READ_B
#undef SEQ_MMA_HALF
#undef SEQ_SYNC_A
#undef SEQ_SYNC_B
}

}  // namespace se

// returns 1 when the shape is not this kernel's (the caller falls back to gemm6p), 0 on success, < 0 on error.  bf16 output, no residual,
// act = identity or GELU; conditions as the persistent kernel of gemm6.hip plus a 32-bit output offset.  out_rows_alloc: how many rows the output
// buffer really has (>= M); from ceil(M / 256) x 256 on the kernel writes whole tiles (no masks, no dump).  with_prof: open a profiling scope here
// (direct callers; se_gemm6_launch has its own).
namespace {
template <int ACT, bool PAD>
void launch6q(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, int M, int N, int K, uint16_t* out_bf16, int ldc, int tiles_m, int tiles_n,
              int group_m, int late_start, char* dump, int grid, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm6q_bf16_kernel<ACT, PAD>), hipFuncAttributeMaxDynamicSharedMemorySize, se::kqLds + 32768);
    attr_set = true;
  }
  hipLaunchKernelGGL((se::gemm6q_bf16_kernel<ACT, PAD>), dim3(grid), dim3(se::kqThreads), se::kqLds + N * 4, st, A, lda, W, ldw, bias, M, N, K, out_bf16, ldc, tiles_m,
                     tiles_n, group_m, late_start, dump);
}
}  // namespace

extern "C" int se_gemm6q_launch(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, int M, int N, int K, int act,
                                uint16_t* out_bf16, int ldc, int out_rows_alloc, int with_prof, void* stream) {
  if (!(N % se::kqBN == 0 && N <= 8192 && K % se::kqBK == 0 && (K / se::kqBK) % 2 == 0 && K >= 4 * se::kqBK && (ldc % 8) == 0 && ldc >= N && lda >= K &&
        ldw >= K && lda % 8 == 0 && ldw % 8 == 0 && (size_t)M * lda < (1u << 31) && (size_t)N * ldw < (1u << 31) && (size_t)(M + 256) * ldc < (1u << 30) &&
        (size_t)((M + se::kqBM - 1) / se::kqBM) * (N / se::kqBN) > 256 && (((uintptr_t)A | (uintptr_t)W | (uintptr_t)out_bf16) % 16) == 0 &&
        (act == SE_ACT_IDENTITY || act == SE_ACT_GELU) && ldc <= 8192))
    return 1;
  const int tiles_m = (M + se::kqBM - 1) / se::kqBM, tiles_n = N / se::kqBN;
  static int group_m = 0, n_cu = 0, late_start = 0, force_ragged = 0;
  static bool init = false;
  static char* dump = nullptr;              // where the stores of row groups past M go (PAD = false): 128 rows x ldc <= 8192 bf16 + the lane offsets
  if (!init) {
    SE_HIP(hipMalloc(reinterpret_cast<void**>(&dump), (size_t)136 * 8192 * 2));
    const char* gm = getenv("SE_AMD_GEMM_GROUPM");
    group_m = gm ? atoi(gm) : 4;
    if (group_m < 1) group_m = 1;
    const char* ls = getenv("SE_AMD_GEMM6P_LATE");
    late_start = ls ? atoi(ls) : 2;
    if (const char* fr = getenv("SE_AMD_GEMM6Q_RAGGED")) force_ragged = atoi(fr);      // A/B: 1 = the masked / dump form even where the output is padded
    int dev = 0;
    hipDeviceProp_t prop;
    SE_HIP(hipGetDevice(&dev));
    SE_HIP(hipGetDeviceProperties(&prop, dev));
    n_cu = prop.multiProcessorCount & ~7;          // one workgroup per CU, a multiple of the 8 XCDs
    if (n_cu < 8) n_cu = 8;
    init = true;
  }
  hipStream_t st = se::as_stream(stream);
  const int grid = std::min(n_cu, (tiles_m * tiles_n + 7) & ~7);
  const bool pad = !force_ragged && out_rows_alloc >= tiles_m * se::kqBM;
  const int pslot = (with_prof && se::prof_on()) ? se::prof_begin(se::kProfGemm, 2.0 * M * (double)N * K, st) : -1;
  if (act == SE_ACT_GELU) {
    if (pad) launch6q<SE_ACT_GELU, true>(A, lda, W, ldw, bias, M, N, K, out_bf16, ldc, tiles_m, tiles_n, group_m, late_start, dump, grid, st);
    else launch6q<SE_ACT_GELU, false>(A, lda, W, ldw, bias, M, N, K, out_bf16, ldc, tiles_m, tiles_n, group_m, late_start, dump, grid, st);
  } else {
    if (pad) launch6q<SE_ACT_IDENTITY, true>(A, lda, W, ldw, bias, M, N, K, out_bf16, ldc, tiles_m, tiles_n, group_m, late_start, dump, grid, st);
    else launch6q<SE_ACT_IDENTITY, false>(A, lda, W, ldw, bias, M, N, K, out_bf16, ldc, tiles_m, tiles_n, group_m, late_start, dump, grid, st);
  }
  if (pslot >= 0) se::prof_end(pslot, st);
  SE_LAUNCH_CHECK();
  return SE_OK;
}
