// gemm6.hip -- 256 x 256 bf16 GEMM, 64-deep K-tiles, eight phases per pair of K-tiles (C = epilogue(A . W^T + bias); contract: gemm.hip)
//
// Why another main loop: gemm3's 32-deep K-steps stage rows of 64 B, so every LDS-DMA wave instruction touches 16 cache lines and uses
// half of each (the "fragment-shaped load" of the CDNA guide: twice the TA work per byte), and the whole family sat at ~0.9 PFLOP/s at
// 4096^3 / 8192^3 where the guide's 256^2 eight-phase structure reaches 1.3-1.5 on the same chip.  This kernel is that structure:
//
//   tile     : 256 x 256 x 64; 512 threads = 8 waves as 2 (M) x 4 (N), wave tile 128 x 64 = 8 x 4 v_mfma_f32_16x16x32_bf16 tiles
//   LDS      : 2 K-tile buffers x 4 half-tile slots x 16 KiB = 128 KiB.  A slot holds 128 rows x 128 B (one full cache line of k per
//              row), 16-B chunk index XOR-ed with (row >> 1) & 7 (on the DMA source address and on the ds_read_b128 side): conflict-free.
//              Half-tiles are cut so that each one is read in exactly ONE phase (B_0 in two):
//                A_h = rows {wr * 128 + h * 64 + [0, 64)} of both wave rows, B_h = columns {wc * 64 + h * 32 + [0, 32)} of all four wave columns
//   phases   : a wave's 128 x 64 output is four 64 x 32 quadrants; per K-tile the quadrants run (0,0) (0,1) (1,1) (1,0), so consecutive
//              phases share one operand in registers: reads per phase = B_0 + A_0 | B_1 | A_1 | B_0 (12 | 4 | 8 | 4 ds_read_b128), 16 MFMAs each.
//              phase = { fragment reads ; one half-tile of LDS-DMA (2 per wave) ; [counted vmcnt] | barrier | 16 MFMAs | barrier };
//              waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave multiplies while its partner reads / issues DMA.
//   staging  : a continuous stream of half-tiles, one per phase, each into the slot whose last reader finished one phase earlier:
//                phase 1 of tile t : B_0 (t+1)     phase 2 : A_0 (t+2)     phase 3 : B_1 (t+2)     phase 4 : A_1 (t+2)
//              The only wait of a K-tile is vmcnt(6) in phase 4 (three half-tiles stay in flight): in-order retirement then guarantees
//              every half-tile of tile t+1; it is read from the next phase on (one barrier later for the lagging wave group).
//   epilogue : gemm3's (swapped operands, C^T accumulators, 16-B stores, bias / GELU / fp32 residual fused, compile-time specialised)
#include "clkprobe.h"
#include <stdlib.h>
#include "common.h"
#include "bf16.h"
#include "prof.h"

SE_CLKPROBE_DECL(clkprobe_gemm6)
namespace se {

constexpr int k6BM = 256, k6BN = 256, k6BK = 64, k6Threads = 512;
constexpr int k6Slot = 128 * 128, k6Buf = 4 * k6Slot, k6Lds = 2 * k6Buf;       // 16 KiB, 64 KiB, 128 KiB
constexpr int k6A0 = 0, k6A1 = k6Slot, k6B0 = 2 * k6Slot, k6B1 = 3 * k6Slot;  // slot offsets inside a buffer

typedef __attribute__((address_space(3))) void* lds6_ptr_t;
typedef const __attribute__((address_space(1))) void* glb6_ptr_t;

__device__ __forceinline__ float act6(float v, int act) {
  if (act == SE_ACT_GELU) return gelu_erf(v);
  if (act == SE_ACT_RELU) return fmaxf(v, 0.f);
  return v;
}

// timing-only ablation (-DSE6_ABL_SAMESRC, results wrong): every tile of the persistent kernel fetches the operands of tile (0, 0) -- the same 2 x 256 rows
// from every CU, i.e. the identical instruction stream with all operand traffic served by L2 / L1 hits: separates "bound by the memory side" from
// "bound by the LDS-DMA path or the issue stream"
// -DSE6_ABL=<mask> (persistent kernel only, timing only, results wrong): 1 no LDS-DMA (and nothing to wait for), 2 no s_barrier, 4 no LDS fragment
// reads (the MFMAs run on whatever the fragment registers hold), 8 no epilogue at all, 16 the epilogue's arithmetic and lane exchange but no store instructions, 32 all tiles store to the first 256 output rows (the stores stay, their HBM traffic goes)
#ifndef SE6_ABL
#define SE6_ABL 0
#endif
#ifdef SE6_ABL_SAMESRC
#define SE6_ABL_SRC(x_) 0
#else
#define SE6_ABL_SRC(x_) (x_)
#endif

// A/B (-DSE6_NT_STORES): the tile's 16-B stores as non-temporal (global_store ... nt) -- the dirty lines of a kernel's output are otherwise written back
// from the eight L2s at the kernel boundary, which the next launch waits for
typedef unsigned int se6_u32x4 __attribute__((ext_vector_type(4)));
#ifdef SE6_NT_STORES
#define SE6_STORE16(ptr_, v_) __builtin_nontemporal_store((se6_u32x4){(v_).x, (v_).y, (v_).z, (v_).w}, reinterpret_cast<se6_u32x4*>(ptr_))
#else
#define SE6_STORE16(ptr_, v_) do { if (!(SE6_ABL & 16) || M < 0) *reinterpret_cast<uint4*>(ptr_) = (v_); } while (0)      /* ablation 16: the tile is converted and exchanged but never stored */
#endif

// Epilogue of a 256 x 256 tile held as C^T accumulators acc[8][4] (col = lane & 15 -> output row, row = 4 (lane >> 4) + r -> 4 consecutive
// columns); expects m0, n0, wr, wc, mrow, ncol, ncol8, godd, bb[4] (bias, zero when the accumulators already carry it) in scope.
#define SE6_EPILOGUE_BODY(PRED)                                                                                            \
  _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                                          \
    const int gm = m0 + wr * 128 + i * 16 + mrow;                                                                          \
    const bool mok = gm < M;                                                                                               \
    const size_t orow = (size_t)((SE6_ABL & 32) ? (gm & 255) : min(gm, M - 1)) * ldc;       /* ablation 32: every tile writes rows 0 .. 255 (L2-resident) */ \
    float4 rr[4];                                                                                                          \
    if constexpr (RES) {                                                                                                   \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                      \
        const int gn = min(n0 + wc * 64 + j * 16 + ncol, N - 4);                                                           \
        rr[j] = *reinterpret_cast<const float4*>(residual + orow + gn);                                                    \
      }                                                                                                                    \
    }                                                                                                                      \
    uint2 pk[4], pm[4];                                                                                                    \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                        \
      const int gn = n0 + wc * 64 + j * 16 + ncol;                                                                         \
      float v0 = acc[i][j][0] + bb[j].x, v1 = acc[i][j][1] + bb[j].y, v2 = acc[i][j][2] + bb[j].z, v3 = acc[i][j][3] + bb[j].w; \
      if constexpr (ACT == SE_ACT_GELU) {                                                                                  \
        /* bf16 output: the transcendental-free polynomial (bf16.h); fp32 output keeps the erf form */                      \
        const f32x2 ga = (OBF && !OF32 && !X3) ? gelu_poly2((f32x2){v0, v1}) : gelu_erf2((f32x2){v0, v1});                 \
        const f32x2 gb = (OBF && !OF32 && !X3) ? gelu_poly2((f32x2){v2, v3}) : gelu_erf2((f32x2){v2, v3});                 \
        v0 = ga.x; v1 = ga.y; v2 = gb.x; v3 = gb.y;                                                                        \
      } else {                                                                                                             \
        v0 = act6(v0, ACT); v1 = act6(v1, ACT); v2 = act6(v2, ACT); v3 = act6(v3, ACT);                                    \
      }                                                                                                                    \
      if constexpr (RES) { v0 += rr[j].x; v1 += rr[j].y; v2 += rr[j].z; v3 += rr[j].w; }                                   \
      pk[j] = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));                                                        \
      if constexpr (DUAL) {                                                                                                \
        /* second output at out_bf16 + dual_off: gelu (erf form) of the bf16-ROUNDED value -- what se_gelu_bf16 computes from the stored rows */ \
        const f32x2 da = gelu_erf2((f32x2){__uint_as_float(pk[j].x << 16), __uint_as_float(pk[j].x & 0xffff0000u)});       \
        const f32x2 db = gelu_erf2((f32x2){__uint_as_float(pk[j].y << 16), __uint_as_float(pk[j].y & 0xffff0000u)});       \
        pm[j] = make_uint2(pack_bf16x2(da.x, da.y), pack_bf16x2(db.x, db.y));                                              \
        if (PRED) {                                                                                                        \
          if (mok && gn < N) *reinterpret_cast<uint2*>(out_bf16 + dual_off + orow + gn) = pm[j];                           \
        }                                                                                                                  \
      }                                                                                                                    \
      if constexpr (X3) {                                                                                                  \
        /* the residual term y2 = bf16(y - y1) of the three-term operand; stored below with the 16-B pair exchange */      \
        const float r0 = v0 - __uint_as_float(pk[j].x << 16), r1 = v1 - __uint_as_float(pk[j].x & 0xffff0000u);            \
        const float r2 = v2 - __uint_as_float(pk[j].y << 16), r3 = v3 - __uint_as_float(pk[j].y & 0xffff0000u);            \
        pm[j] = make_uint2(pack_bf16x2(r0, r1), pack_bf16x2(r2, r3));                                                      \
        if (PRED) {                                                                                                        \
          if (mok && gn < N) {                                                                                             \
            uint16_t* o3 = out_bf16 + orow + gn;                                                                           \
            *reinterpret_cast<uint2*>(o3) = pk[j];                                                                         \
            *reinterpret_cast<uint2*>(o3 + ldc / 3) = pk[j];                                                               \
            *reinterpret_cast<uint2*>(o3 + 2 * (ldc / 3)) = pm[j];                                                         \
          }                                                                                                                \
        }                                                                                                                  \
      } else                                                                                                               \
      if (!(PRED) || (mok && gn < N)) {                                                                                    \
        if constexpr (OF32) *reinterpret_cast<float4*>(out_f32 + orow + gn) = make_float4(v0, v1, v2, v3);                 \
        if constexpr (OBF) {                                                                                               \
          if (PRED) *reinterpret_cast<uint2*>(out_bf16 + orow + gn) = pk[j];                                               \
        }                                                                                                                  \
      }                                                                                                                    \
    }                                                                                                                      \
    if constexpr (OBF) {                                                                                                   \
      if (!(PRED)) {                                                                                                       \
        /* 16-B stores: lanes l, l ^ 16 trade 4-column pieces of two neighbouring MFMA tiles, so each lane owns 8 consecutive   \
           bf16 columns and a wave instruction writes 16 rows x 64 contiguous bytes (gemm3.hip); X3: the y1 slice twice, then y2 */ \
        _Pragma("unroll") for (int sl = 0; sl < (X3 ? 3 : DUAL ? 2 : 1); ++sl) {                                           \
          _Pragma("unroll") for (int p2 = 0; p2 < 2; ++p2) {                                                               \
            const bool second = (X3 && sl == 2) || (DUAL && sl == 1);                                                      \
            const uint2 e0 = second ? pm[2 * p2] : pk[2 * p2], e1 = second ? pm[2 * p2 + 1] : pk[2 * p2 + 1];               \
            const uint2 keep = godd ? e1 : e0;                                                                             \
            const uint2 send = godd ? e0 : e1;                                                                             \
            uint2 recv;                                                                                                    \
            recv.x = __shfl_xor(send.x, 16);                                                                               \
            recv.y = __shfl_xor(send.y, 16);                                                                               \
            const uint4 o16 = godd ? make_uint4(recv.x, recv.y, keep.x, keep.y) : make_uint4(keep.x, keep.y, recv.x, recv.y); \
            SE6_STORE16(out_bf16 + orow + (X3 ? sl * (ldc / 3) : (DUAL && sl == 1) ? dual_off : 0) + n0 + wc * 64 + 16 * (2 * p2 + (godd ? 1 : 0)) + ncol8, o16); \
          }                                                                                                                \
        }                                                                                                                  \
      }                                                                                                                    \
    }                                                                                                                      \
  }

// ACT: compile-time activation; EF bit 0: fp32 residual, bit 1: bf16 output, bit 2: fp32 output (N % 4 == 0, 16-B rows), bit 3 (with bit 1): the bf16
// output is the three-term operand [y1 | y1 | y2] of the next bf16x3 projection (row stride ldc = 3 x slice width; exact erf GELU)
template <int ACT, int EF>
__global__ __launch_bounds__(k6Threads) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm6_bf16_kernel(
    const uint16_t* __restrict__ A, int lda, const uint16_t* __restrict__ W, int ldw, const float* __restrict__ bias,
    const float* __restrict__ residual, int M, int N, int K, uint16_t* __restrict__ out_bf16, float* __restrict__ out_f32,
    int ldc, int tiles_m, int tiles_n, int group_m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  SE_CLKPROBE_BEGIN();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // XCD-aware bijective remap + grouped order (as gemm3): each XCD walks a contiguous id range, GROUP_M tile rows swept n-major
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
  const int m0 = tm * k6BM, n0 = tn * k6BN;

  // ---- DMA sources.  A half-tile slot = 16 pieces of 1 KiB = 8 slot rows x 128 B each; wave w issues pieces w and w + 8.
  //      lane -> slot row rho = 8 q + (lane >> 3); LDS position lane & 7 holds logical chunk (lane & 7) ^ ((rho >> 1) & 7).
  //      A_h slot row rho <-> tile row (rho >> 6) * 128 + h * 64 + (rho & 63);  B_h slot row rho <-> tile column (rho >> 5) * 64 + h * 32 + (rho & 31)
  const uint16_t* a_src[2][2];      // [h][piece]
  const uint16_t* b_src[2][2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int rho = 8 * (wave + 8 * p) + (lane >> 3);
    const int lc = ((lane & 7) ^ ((rho >> 1) & 7)) << 3;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int trow = (rho >> 6) * 128 + h * 64 + (rho & 63);
      const int tcol = (rho >> 5) * 64 + h * 32 + (rho & 31);
      a_src[h][p] = A + (size_t)min(m0 + trow, M - 1) * lda + lc;
      b_src[h][p] = W + (size_t)min(n0 + tcol, N - 1) * ldw + lc;
    }
  }
#define SE6_DMA(src, slot_off, buf, kt)                                                                                    \
  do {                                                                                                                     \
    _Pragma("unroll") for (int p_ = 0; p_ < 2; ++p_)                                                                       \
      __builtin_amdgcn_global_load_lds((glb6_ptr_t)(src[p_] + (size_t)(kt) * k6BK),                                        \
                                       (lds6_ptr_t)(smem + (buf) * k6Buf + (slot_off) + (wave + 8 * p_) * 1024), 16, 0, 0); \
  } while (0)

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- fragment addresses: lane -> row (lane & 15) of a 16-row MFMA tile, logical chunk 4 s + (lane >> 4) of k-slice s.
  //      Tiles 16 rows apart share the swizzle term ((rho >> 1) & 7 only sees rho mod 16): tile i sits at base + 2 KiB * i.
  const int frow = lane & 15, fch = lane >> 4;
  int a_ad[2], b_ad[2];
  {
    const int ra = wr * 64 + frow, rb = wc * 32 + frow;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      a_ad[s] = ra * 128 + (((4 * s + fch) ^ ((ra >> 1) & 7)) << 4);
      b_ad[s] = rb * 128 + (((4 * s + fch) ^ ((rb >> 1) & 7)) << 4);
    }
  }
  bf16x8 af[4][2], bfr[2][2];          // [tile][k-slice]
#define SE6_READ_A(buf, h)                                                                                                 \
  _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                        \
    af[i_][s_] = *reinterpret_cast<const bf16x8*>(smem + (buf) * k6Buf + ((h) ? k6A1 : k6A0) + a_ad[s_] + i_ * 2048);
#define SE6_READ_B(buf, h)                                                                                                 \
  _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                        \
    bfr[j_][s_] = *reinterpret_cast<const bf16x8*>(smem + (buf) * k6Buf + ((h) ? k6B1 : k6B0) + b_ad[s_] + j_ * 2048);
#define SE6_MMA(mq, nq)                                                                                                    \
  _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                        \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                                       \
      acc[(mq) * 4 + i_][(nq) * 2 + j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j_][s_], af[i_][s_], acc[(mq) * 4 + i_][(nq) * 2 + j_], 0, 0, 0);
  // reads retired BEFORE the barrier: the slot read in this phase is refilled by DMA one phase later (WAR), and the MFMAs start right
  // after the barrier
#define SE6_SYNC_A()                                                                                                       \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                       \
  __builtin_amdgcn_sched_barrier(0);                                                                                       \
  __builtin_amdgcn_s_barrier();                                                                                            \
  __builtin_amdgcn_sched_barrier(0);                                                                                       \
  __builtin_amdgcn_s_setprio(1);
#define SE6_SYNC_B()                                                                                                       \
  __builtin_amdgcn_s_setprio(0);                                                                                           \
  __builtin_amdgcn_sched_barrier(0);                                                                                       \
  __builtin_amdgcn_s_barrier();                                                                                            \
  __builtin_amdgcn_sched_barrier(0);

  const int nk = K / k6BK;                  // >= 2 (launcher)
  // ---- prologue: all of tile 0, then tile 1's A_0, B_1, A_1 (what phases 2-4 of "tile -1" would have issued)
  SE6_DMA(a_src[0], k6A0, 0, 0);
  SE6_DMA(b_src[0], k6B0, 0, 0);
  SE6_DMA(b_src[1], k6B1, 0, 0);
  SE6_DMA(a_src[1], k6A1, 0, 0);
  SE6_DMA(a_src[0], k6A0, 1, 1);
  SE6_DMA(b_src[1], k6B1, 1, 1);
  SE6_DMA(a_src[1], k6A1, 1, 1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");        // tile 0 landed (this wave's pieces)
  __builtin_amdgcn_s_barrier();
  const bool late = wave >= 4;
  if (late) __builtin_amdgcn_s_barrier();                  // stagger: waves 4-7 run one barrier behind

  // one K-tile from buffer BUF (compile-time); T = its index; STEADY: tile T + 2 exists (no run-time branches in the body)
#define SE6_TILE(BUF, T, STEADY)                                                                                           \
  {                                                                                                                        \
    /* phase 1: quadrant (0,0) */                                                                                          \
    SE6_READ_B(BUF, 0)                                                                                                     \
    SE6_READ_A(BUF, 0)                                                                                                     \
    if (STEADY || (T) + 1 < nk) SE6_DMA(b_src[0], k6B0, (BUF) ^ 1, (T) + 1);                                               \
    SE6_SYNC_A()                                                                                                           \
    SE6_MMA(0, 0)                                                                                                          \
    SE6_SYNC_B()                                                                                                           \
    /* phase 2: quadrant (0,1); A_0 of this buffer was read one phase ago -> refill it with tile T + 2 */                  \
    SE6_READ_B(BUF, 1)                                                                                                     \
    if (STEADY || (T) + 2 < nk) SE6_DMA(a_src[0], k6A0, BUF, (T) + 2);                                                     \
    SE6_SYNC_A()                                                                                                           \
    SE6_MMA(0, 1)                                                                                                          \
    SE6_SYNC_B()                                                                                                           \
    /* phase 3: quadrant (1,1) */                                                                                          \
    SE6_READ_A(BUF, 1)                                                                                                     \
    if (STEADY || (T) + 2 < nk) SE6_DMA(b_src[1], k6B1, BUF, (T) + 2);                                                     \
    SE6_SYNC_A()                                                                                                           \
    SE6_MMA(1, 1)                                                                                                          \
    SE6_SYNC_B()                                                                                                           \
    /* phase 4: quadrant (1,0); the K-tile's only DMA wait: everything up to B_0 (T + 1) has landed, 3 half-tiles of T + 2 stay in flight */ \
    SE6_READ_B(BUF, 0)                                                                                                     \
    if (STEADY || (T) + 2 < nk) {                                                                                          \
      SE6_DMA(a_src[1], k6A1, BUF, (T) + 2);                                                                               \
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                                     \
    } else {                                                                                                               \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                     \
    }                                                                                                                      \
    SE6_SYNC_A()                                                                                                           \
    SE6_MMA(1, 0)                                                                                                          \
    SE6_SYNC_B()                                                                                                           \
  }
  int t = 0;
  for (; t + 3 < nk; t += 2) {              // steady state: tiles t and t + 1 both have a tile two ahead
    SE6_TILE(0, t, true)
    SE6_TILE(1, t + 1, true)
  }
  for (; t + 1 < nk; t += 2) {
    SE6_TILE(0, t, false)
    SE6_TILE(1, t + 1, false)
  }
  if (t < nk) SE6_TILE(0, t, false)
#undef SE6_TILE
  if (!late) __builtin_amdgcn_s_barrier();                 // re-align the two groups (barrier counts must match)

  // ---- epilogue: C^T accumulators: col = lane & 15 -> output row, row = 4 (lane >> 4) + r -> 4 consecutive columns
  constexpr bool RES = EF & 1, OBF = EF & 2, OF32 = EF & 4, X3 = EF & 8, DUAL = false;
  constexpr long long dual_off = 0;
  (void)dual_off;
  const int mrow = lane & 15, ncol = 4 * (lane >> 4);
  float4 bb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int gn = min(n0 + wc * 64 + j * 16 + ncol, N - 4);
    bb[j] = bias ? *reinterpret_cast<const float4*>(bias + gn) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // wave-uniform; the interior path's bf16 stores are 16 B wide
  const bool interior = (m0 + k6BM <= M) && (n0 + k6BN <= N) && (!OBF || ((ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(out_bf16) & 15) == 0));
  const bool godd = (lane >> 4) & 1;                      // odd lane group: keeps tile 2p+1, gets the left 4 columns from its partner
  const int ncol8 = 4 * ((lane >> 4) & ~1);               // first of this lane's 8 consecutive columns inside the 16-column tile
  if (interior) {
    SE6_EPILOGUE_BODY(false)
  } else {
    SE6_EPILOGUE_BODY(true)
  }
  SE_CLKPROBE_END(clkprobe_gemm6);
#undef SE6_DMA
#undef SE6_READ_A
#undef SE6_READ_B
#undef SE6_MMA
#undef SE6_SYNC_A
#undef SE6_SYNC_B
}


// ------------------------------------------------------------------------------------------------------------------
// Persistent form (bf16 output, no residual: the QKV and FFN1 projections).  The K-sweep of the one-tile-per-workgroup kernel above
// (tools/bench_kernels.py ksweep) shows a marginal rate of 1.4-1.5 PFLOP/s and a FIXED cost of 40 (QKV) / 72 (FFN1) us per launch at
// K = 768: pipeline fill, epilogue and the 147 / 197 MB output write of every round, none of it overlapped with MFMAs because a
// 128-KiB-LDS workgroup owns its CU.  Here gridDim.x workgroups (one per CU) walk lists of output tiles and
//   * the half-tile stream runs CONTINUOUSLY across tile boundaries (the last two K-tiles of a tile already stage the first two of the next),
//   * the epilogue's 16 store instructions per wave stay in flight under the next tile's first K-tiles (counted vmcnt includes them),
//   * workgroups whose list is one tile shorter start `late_start` x ~4 us late, so the write bursts of the two populations fall into each
//     other's main loops instead of all CUs writing at once.
// Tile order: XCD x (= blockIdx & 7) owns a contiguous id range (gemm6's grouped order); its workgroups take ids start + slot + 32 i.
// ------------------------------------------------------------------------------------------------------------------
// DUAL_ = 1 (training forward of FFN1, ACT = identity): the tile is stored twice -- the pre-activation at out_bf16 and its GELU at out_bf16 + dual_off
// (elements) -- so that the activation is not a second launch that re-reads the rows it has just written
template <int ACT, int INM = 0, int DUAL_ = 0>
__global__ __launch_bounds__(k6Threads) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm6p_bf16_kernel(
    const uint16_t* __restrict__ A, int lda, const uint16_t* __restrict__ W, int ldw, const float* __restrict__ bias, int M, int N, int K,
    uint16_t* __restrict__ out_bf16, int ldc, int tiles_m, int tiles_n, int group_m, int late_start, long long dual_off) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  SE_CLKPROBE_BEGIN();
  constexpr bool RES = false, OBF = true, OF32 = false, X3 = false, DUAL = DUAL_ != 0;
  (void)dual_off;
  const float* residual = nullptr;          // named by the shared epilogue body inside discarded branches only
  float* out_f32 = nullptr;
  (void)residual; (void)out_f32;
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
  if ((late_start & 0xff) > 0 && my_tiles < (range_count + wpx - 1) / wpx)
    for (int i = 0; i < (late_start & 0xff); ++i) __builtin_amdgcn_s_sleep(127);
  // A/B (SE_AMD_GEMM6P_SKEW = late_start >> 8): workgroup slot s of every XCD starts (s % 8) x skew x ~0.27 us late, so that the CUs' store bursts
  // at the tile boundaries do not all hit the memory system in the same microsecond
  for (int i = 0; i < (late_start >> 8) * (slot & 7); ++i) __builtin_amdgcn_s_sleep(8);

  // ---- DMA sources as 32-bit byte offsets from A / W (the launcher checks they fit): 8 registers per tile, rewritten in place when the
  //      stream moves on to the next tile.  Mapping as in gemm6_bf16_kernel.
  uint32_t a_of[2][2], b_of[2][2];
#define SE6P_SET_SRC(id_, m0_, n0_)                                                                                        \
  do {                                                                                                                     \
    const int per_group_ = group_m * tiles_n, grp_ = (id_) / per_group_, first_m_ = grp_ * group_m;                        \
    const int gsz_ = min(tiles_m - first_m_, group_m), in_ = (id_) - grp_ * per_group_;                                    \
    const int tn_ = in_ / gsz_, tm_ = first_m_ + (in_ - tn_ * gsz_);                                                       \
    m0_ = tm_ * k6BM;                                                                                                      \
    n0_ = tn_ * k6BN;                                                                                                      \
    int ln_ = lane;                                                                                                        \
    asm volatile("" : "+v"(ln_));     /* opaque: keeps the lane-derived terms from being hoisted out of the tile loop and spilled */ \
    _Pragma("unroll") for (int p_ = 0; p_ < 2; ++p_) {                                                                     \
      const int rho_ = 8 * (wave + 8 * p_) + (ln_ >> 3);                                                                   \
      const int lc_ = ((ln_ & 7) ^ ((rho_ >> 1) & 7)) << 3;                                                                \
      _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) {                                                                   \
        const int trow_ = (rho_ >> 6) * 128 + h_ * 64 + (rho_ & 63);                                                       \
        const int tcol_ = (rho_ >> 5) * 64 + h_ * 32 + (rho_ & 31);                                                        \
        a_of[h_][p_] = (uint32_t)(min(SE6_ABL_SRC(m0_) + trow_, M - 1) * lda + lc_) * 2u;                                  \
        b_of[h_][p_] = (uint32_t)(min(SE6_ABL_SRC(n0_) + tcol_, N - 1) * ldw + lc_) * 2u;                                  \
      }                                                                                                                    \
    }                                                                                                                      \
  } while (0)
  // one half-tile: `of` = a_of[h] / b_of[h], kt = K-tile index inside the tile the offsets belong to.  Hand-written issue: SGPR base +
  // 32-bit per-lane offset (the builtin widened every offset to a 64-bit VGPR pair and spilled them); M0 = LDS destination, saved and
  // restored around the instruction.  These loads are invisible to the compiler's vmcnt bookkeeping: every wait on them is counted
  // by hand below (compiler-issued loads / stores in between only make ITS waits more conservative).
  const uint32_t lds_wave = (uint32_t)(size_t)(lds6_ptr_t)smem + wave * 1024;
#define SE6P_DMA(base, of, slot_off, buf, kt)                                                                              \
  do {                                                                                                                     \
    const char* sb_ = reinterpret_cast<const char*>(base) + (size_t)(kt) * (k6BK * 2);                                     \
    if (SE6_ABL & 1) break;                                                                                                \
    _Pragma("unroll") for (int p_ = 0; p_ < 2; ++p_) {                                                                     \
      uint32_t keep_;                                                                                                      \
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"      \
                   : "=&s"(keep_)                                                                                          \
                   : "v"(of[p_]), "s"(sb_), "s"(lds_wave + (uint32_t)((buf) * k6Buf + (slot_off) + p_ * 8192))             \
                   : "memory");                                                                                            \
    }                                                                                                                      \
  } while (0)

  const int frow = lane & 15, fch = lane >> 4;
  int a_ad[2], b_ad[2];
  {
    const int ra = wr * 64 + frow, rb = wc * 32 + frow;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      a_ad[s] = ra * 128 + (((4 * s + fch) ^ ((ra >> 1) & 7)) << 4);
      b_ad[s] = rb * 128 + (((4 * s + fch) ^ ((rb >> 1) & 7)) << 4);
    }
  }
  bf16x8 af[4][2], bfr[2][2];
  if (SE6_ABL & 4) {      // ablation: the fragments are never read; give them lane-dependent, non-trivial contents once (zeros would raise the clock)
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
#pragma unroll
      for (int i_ = 0; i_ < 4; ++i_)
#pragma unroll
        for (int e_ = 0; e_ < 8; ++e_) af[i_][s_][e_] = (__bf16)(0.01f * (float)((lane * 7 + i_ * 3 + e_ + s_) % 13 - 6));
#pragma unroll
      for (int j_ = 0; j_ < 2; ++j_)
#pragma unroll
        for (int e_ = 0; e_ < 8; ++e_) bfr[j_][s_][e_] = (__bf16)(0.01f * (float)((lane * 5 + j_ + e_ * 3 + s_) % 11 - 5));
    }
  }
#define SE6_READ_A(buf, h)                                                                                                 \
  if (!(SE6_ABL & 4)) _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)    \
    af[i_][s_] = *reinterpret_cast<const bf16x8*>(smem + (buf) * k6Buf + ((h) ? k6A1 : k6A0) + a_ad[s_] + i_ * 2048);
#define SE6_READ_B(buf, h)                                                                                                 \
  if (!(SE6_ABL & 4)) _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)    \
    bfr[j_][s_] = *reinterpret_cast<const bf16x8*>(smem + (buf) * k6Buf + ((h) ? k6B1 : k6B0) + b_ad[s_] + j_ * 2048);
#define SE6_MMA(mq, nq)                                                                                                    \
  _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                        \
    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                                                       \
      acc[(mq) * 4 + i_][(nq) * 2 + j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j_][s_], af[i_][s_], acc[(mq) * 4 + i_][(nq) * 2 + j_], 0, 0, 0);
#define SE6_SYNC_A()                                                                                                       \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                       \
  __builtin_amdgcn_sched_barrier(0);                                                                                       \
  __builtin_amdgcn_s_barrier();                                                                                            \
  __builtin_amdgcn_sched_barrier(0);                                                                                       \
  __builtin_amdgcn_s_setprio(1);
#define SE6_SYNC_B()                                                                                                       \
  __builtin_amdgcn_s_setprio(0);                                                                                           \
  __builtin_amdgcn_sched_barrier(0);                                                                                       \
  __builtin_amdgcn_s_barrier();                                                                                            \
  __builtin_amdgcn_sched_barrier(0);
  // one K-tile from buffer BUF; I1..I4: what each phase issues (statements), WAIT: the phase-4 wait (statement).
  // INM = 1: the half-tile of a phase is issued between its two groups of eight MFMAs instead of in the read section (the wait of phase 4
  // then sees one half-tile fewer in flight: SE6P_W6 / the stores_pending wait are defined accordingly).
#define SE6_MMA_HALF(mq, nq, s_)                                                                                           \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                        \
    acc[(mq) * 4 + i_][(nq) * 2 + j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j_][s_], af[i_][s_], acc[(mq) * 4 + i_][(nq) * 2 + j_], 0, 0, 0);
#undef SE6_SYNC_A
#undef SE6_SYNC_B
#define SE6_SYNC_A()                                                                                                       \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                       \
  __builtin_amdgcn_sched_barrier(0);                                                                                       \
  if (!(SE6_ABL & 2)) __builtin_amdgcn_s_barrier();                                                                                            \
  __builtin_amdgcn_sched_barrier(0);                                                                                       \
  if (INM != 2) __builtin_amdgcn_s_setprio(1);
#define SE6_SYNC_B()                                                                                                       \
  if (INM != 2) __builtin_amdgcn_s_setprio(0);                                                                             \
  __builtin_amdgcn_sched_barrier(0);                                                                                       \
  if (!(SE6_ABL & 2)) __builtin_amdgcn_s_barrier();                                                                                            \
  __builtin_amdgcn_sched_barrier(0);
#define SE6P_ISSUE_MID_BEGIN if (INM == 1) { __builtin_amdgcn_sched_barrier(0);
#define SE6P_ISSUE_MID_END __builtin_amdgcn_sched_barrier(0); }
#define SE6P_TILE(BUF, I1, I2, I3, I4, WAIT)                                                                               \
  {                                                                                                                        \
    SE6_READ_B(BUF, 0)                                                                                                     \
    SE6_READ_A(BUF, 0)                                                                                                     \
    if (INM != 1) { I1; }                                                                                                    \
    SE6_SYNC_A()                                                                                                           \
    SE6_MMA_HALF(0, 0, 0)                                                                                                  \
    SE6P_ISSUE_MID_BEGIN I1; SE6P_ISSUE_MID_END                                                                            \
    SE6_MMA_HALF(0, 0, 1)                                                                                                  \
    SE6_SYNC_B()                                                                                                           \
    SE6_READ_B(BUF, 1)                                                                                                     \
    if (INM != 1) { I2; }                                                                                                    \
    SE6_SYNC_A()                                                                                                           \
    SE6_MMA_HALF(0, 1, 0)                                                                                                  \
    SE6P_ISSUE_MID_BEGIN I2; SE6P_ISSUE_MID_END                                                                            \
    SE6_MMA_HALF(0, 1, 1)                                                                                                  \
    SE6_SYNC_B()                                                                                                           \
    SE6_READ_A(BUF, 1)                                                                                                     \
    if (INM != 1) { I3; }                                                                                                    \
    SE6_SYNC_A()                                                                                                           \
    SE6_MMA_HALF(1, 1, 0)                                                                                                  \
    SE6P_ISSUE_MID_BEGIN I3; SE6P_ISSUE_MID_END                                                                            \
    SE6_MMA_HALF(1, 1, 1)                                                                                                  \
    SE6_SYNC_B()                                                                                                           \
    SE6_READ_B(BUF, 0)                                                                                                     \
    if (INM != 1) { I4; }                                                                                                    \
    WAIT;                                                                                                                  \
    SE6_SYNC_A()                                                                                                           \
    SE6_MMA_HALF(1, 0, 0)                                                                                                  \
    SE6P_ISSUE_MID_BEGIN I4; SE6P_ISSUE_MID_END                                                                            \
    SE6_MMA_HALF(1, 0, 1)                                                                                                  \
    SE6_SYNC_B()                                                                                                           \
  }
#define SE6P_NOP ((void)0)
#define SE6P_W6 do { if (INM == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); } while (0)
#define SE6P_W0 asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

  // bias of every output column, staged once in the LDS beside the ring: the tile loop then contains no compiler-visible VMEM load at all
  // (a per-tile global load made hipcc put s_waitcnt vmcnt(0) at the loop header, which drained the stores this kernel leaves in flight)
  float* bias_lds = reinterpret_cast<float*>(smem + k6Lds);
  for (int c = tid; c < N; c += k6Threads) bias_lds[c] = bias ? bias[c] : 0.f;
  __syncthreads();

  const int nk = K / k6BK;                  // even, >= 4 (launcher)
  int tile_id = range_start + slot;
  int m0, n0;
  SE6P_SET_SRC(tile_id, m0, n0);
  // ---- prologue: K-tile 0 and all of K-tile 1 of the first tile (later tiles arrive with the same state: their K-tile 0 landed,
  //      A_0 / B_1 / A_1 of K-tile 1 issued by the previous tile's last phases, B_0 of K-tile 1 issued just before the epilogue)
  SE6P_DMA(A, a_of[0], k6A0, 0, 0);
  SE6P_DMA(W, b_of[0], k6B0, 0, 0);
  SE6P_DMA(W, b_of[1], k6B1, 0, 0);
  SE6P_DMA(A, a_of[1], k6A1, 0, 0);
  SE6P_DMA(A, a_of[0], k6A0, 1, 1);
  SE6P_DMA(W, b_of[1], k6B1, 1, 1);
  SE6P_DMA(A, a_of[1], k6A1, 1, 1);
  SE6P_DMA(W, b_of[0], k6B0, 1, 1);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  if (!(SE6_ABL & 2)) __builtin_amdgcn_s_barrier();
  const bool late = wave >= 4;
  if (!(SE6_ABL & 2) && (late)) __builtin_amdgcn_s_barrier();                  // stagger: waves 4-7 run one barrier behind

  bool stores_pending = false;              // the previous tile's epilogue left exactly 16 store instructions per wave in flight
  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int ti = 0; ti < my_tiles; ++ti) {
    const bool has_next = ti + 1 < my_tiles;

    // K-tile 0: B_0 (1) is already on its way.  Its wait must not cover the previous tile's stores (issued after B_0 (1)): 6 + 16.
    SE6P_TILE(0, SE6P_NOP, SE6P_DMA(A, a_of[0], k6A0, 0, 2), SE6P_DMA(W, b_of[1], k6B1, 0, 2), SE6P_DMA(A, a_of[1], k6A1, 0, 2),
              if (stores_pending) { if (INM == 1) asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(22)" ::: "memory"); } else SE6P_W6)
    SE6P_TILE(1, SE6P_DMA(W, b_of[0], k6B0, 0, 2), SE6P_DMA(A, a_of[0], k6A0, 1, 3), SE6P_DMA(W, b_of[1], k6B1, 1, 3), SE6P_DMA(A, a_of[1], k6A1, 1, 3), SE6P_W6)
    for (int t = 2; t + 2 < nk; t += 2) {
      SE6P_TILE(0, SE6P_DMA(W, b_of[0], k6B0, 1, t + 1), SE6P_DMA(A, a_of[0], k6A0, 0, t + 2), SE6P_DMA(W, b_of[1], k6B1, 0, t + 2),
                SE6P_DMA(A, a_of[1], k6A1, 0, t + 2), SE6P_W6)
      SE6P_TILE(1, SE6P_DMA(W, b_of[0], k6B0, 0, t + 2), SE6P_DMA(A, a_of[0], k6A0, 1, t + 3), SE6P_DMA(W, b_of[1], k6B1, 1, t + 3),
                SE6P_DMA(A, a_of[1], k6A1, 1, t + 3), SE6P_W6)
    }
    // last two K-tiles: after B_0 (nk - 1) every remaining issue belongs to the NEXT output tile (K-tiles 0 and 1), so the eight source
    // offsets are rewritten in place; the epilogue keeps this tile's m0 / n0
    const int m0c = m0, n0c = n0;
    // (the last tile of the list "prefetches" itself: same instruction stream, no second copy of the 128-accumulator code path for the
    //  register allocator to trip over; those bytes are never read and are drained before the kernel ends)
    const int next_id = has_next ? tile_id + wpx : tile_id;
    SE6P_TILE(0, SE6P_DMA(W, b_of[0], k6B0, 1, nk - 1); tile_id = next_id; SE6P_SET_SRC(tile_id, m0, n0),
              SE6P_DMA(A, a_of[0], k6A0, 0, 0), SE6P_DMA(W, b_of[1], k6B1, 0, 0), SE6P_DMA(A, a_of[1], k6A1, 0, 0), SE6P_W6)
    SE6P_TILE(1, SE6P_DMA(W, b_of[0], k6B0, 0, 0), SE6P_DMA(A, a_of[0], k6A0, 1, 1), SE6P_DMA(W, b_of[1], k6B1, 1, 1), SE6P_DMA(A, a_of[1], k6A1, 1, 1), SE6P_W6)
    if (!(SE6_ABL & 2) && (!late)) __builtin_amdgcn_s_barrier();               // re-align the wave groups: both run the epilogue together
    SE6P_DMA(W, b_of[0], k6B0, 1, 1);                      // B_0 of the next tile's K-tile 1 (its slot was last read in the phase just finished)

    // ---- epilogue of tile (m0c, n0c); no LDS, no barrier
    {
      const int m0 = m0c, n0 = n0c;
      int le_ = lane;
      asm volatile("" : "+v"(le_));      // opaque (see SE6P_SET_SRC): the epilogue's lane constants are rebuilt per tile, not kept live
      const int mrow = le_ & 15, ncol = 4 * (le_ >> 4);
      const bool godd = (le_ >> 4) & 1;
      const int ncol8 = 4 * ((le_ >> 4) & ~1);
      float4 bb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bb[j] = *reinterpret_cast<const float4*>(bias_lds + n0 + wc * 64 + j * 16 + ncol);
      const bool interior = m0 + k6BM <= M;                // N is a multiple of 256 and the bf16 rows are 16-B aligned (launcher)
      if ((SE6_ABL & 8) && acc[0][0][0] != 12345.678f) {
        // ablation: no stores (the comparison keeps the accumulators alive)
      } else if (interior) {
        SE6_EPILOGUE_BODY(false)
        stores_pending = true;
      } else {
        SE6_EPILOGUE_BODY(true)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // exec-masked stores: their count is not fixed, so drain before the counted waits resume
        stores_pending = false;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (!(SE6_ABL & 2) && (late && has_next)) __builtin_amdgcn_s_barrier();    // stagger again
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // no LDS-DMA may outlive the workgroup's LDS allocation
  SE_CLKPROBE_END(clkprobe_gemm6);
#undef SE6P_TILE
#undef SE6P_SET_SRC
#undef SE6P_DMA
#undef SE6_READ_A
#undef SE6_READ_B
#undef SE6_MMA
#undef SE6_SYNC_A
#undef SE6_SYNC_B
}

}  // namespace se

namespace {
struct G6Args {
  const uint16_t* A; int lda; const uint16_t* W; int ldw; const float* bias; const float* residual; int M, N, K;
  uint16_t* out_bf16; float* out_f32; int ldc; hipStream_t st;
};

template <int ACT, int EF>
int launch6(const G6Args& g) {
  const int tiles_m = (g.M + se::k6BM - 1) / se::k6BM, tiles_n = (g.N + se::k6BN - 1) / se::k6BN;
  static int group_m = 0;
  static bool attr_set = false;
  if (!attr_set) {
    const char* gm = getenv("SE_AMD_GEMM_GROUPM");
    group_m = gm ? atoi(gm) : 4;
    if (group_m < 1) group_m = 1;
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm6_bf16_kernel<ACT, EF>), hipFuncAttributeMaxDynamicSharedMemorySize, se::k6Lds));
    attr_set = true;
  }
  hipLaunchKernelGGL((se::gemm6_bf16_kernel<ACT, EF>), dim3(tiles_m * tiles_n), dim3(se::k6Threads), se::k6Lds, g.st, g.A, g.lda, g.W, g.ldw,
                     g.bias, g.residual, g.M, g.N, g.K, g.out_bf16, g.out_f32, g.ldc, tiles_m, tiles_n, group_m);
  SE_LAUNCH_CHECK();
  return SE_OK;
}
}  // namespace

namespace {
template <int ACT>
int launch6p(const G6Args& g) {
  const int tiles_m = (g.M + se::k6BM - 1) / se::k6BM, tiles_n = g.N / se::k6BN;
  static int group_m = 0, n_cu = 0, late_start = 0;
  static bool attr_set = false;
  if (!attr_set) {
    const char* gm = getenv("SE_AMD_GEMM_GROUPM");
    group_m = gm ? atoi(gm) : 4;
    if (group_m < 1) group_m = 1;
    const char* ls = getenv("SE_AMD_GEMM6P_LATE");
    late_start = ls ? atoi(ls) : 2;
    if (const char* sk = getenv("SE_AMD_GEMM6P_SKEW")) late_start |= (atoi(sk) & 0xff) << 8;
    int dev = 0;
    hipDeviceProp_t prop;
    SE_HIP(hipGetDevice(&dev));
    SE_HIP(hipGetDeviceProperties(&prop, dev));
    n_cu = prop.multiProcessorCount & ~7;          // one workgroup per CU, a multiple of the 8 XCDs
    if (n_cu < 8) n_cu = 8;
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm6p_bf16_kernel<ACT, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, se::k6Lds + 32768));
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm6p_bf16_kernel<ACT, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, se::k6Lds + 32768));
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm6p_bf16_kernel<ACT, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, se::k6Lds + 32768));
    attr_set = true;
  }
  static int inm = -1;
  // 0: the first form (DMA issue in the read section, raised wave priority around the MFMA groups); 1: DMA issue between the MFMA groups
  // (A/B: tie); 2 (default): the first form without the priority change (A/B, whole step: 4.018 -> 4.003 ms)
  if (inm < 0) { const char* e = getenv("SE_AMD_GEMM6P_INM"); inm = e ? atoi(e) : 2; }
  const int grid = std::min(n_cu, (tiles_m * tiles_n + 7) & ~7);
  if (inm == 2)
    hipLaunchKernelGGL((se::gemm6p_bf16_kernel<ACT, 2>), dim3(grid), dim3(se::k6Threads), se::k6Lds + g.N * 4, g.st, g.A, g.lda, g.W, g.ldw, g.bias, g.M, g.N,
                       g.K, g.out_bf16, g.ldc, tiles_m, tiles_n, group_m, late_start, 0LL);
  else if (inm)
    hipLaunchKernelGGL((se::gemm6p_bf16_kernel<ACT, 1>), dim3(grid), dim3(se::k6Threads), se::k6Lds + g.N * 4, g.st, g.A, g.lda, g.W, g.ldw, g.bias, g.M, g.N,
                       g.K, g.out_bf16, g.ldc, tiles_m, tiles_n, group_m, late_start, 0LL);
  else
    hipLaunchKernelGGL((se::gemm6p_bf16_kernel<ACT, 0>), dim3(grid), dim3(se::k6Threads), se::k6Lds + g.N * 4, g.st, g.A, g.lda, g.W, g.ldw, g.bias, g.M, g.N,
                       g.K, g.out_bf16, g.ldc, tiles_m, tiles_n, group_m, late_start, 0LL);
  SE_LAUNCH_CHECK();
  return SE_OK;
}
}  // namespace

// returns 1 if this kernel does not handle the call (caller falls back to gemm3 / gemm2), 0 on success, < 0 on error
extern "C" int se_gemm6_launch(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const float* residual_f32,
                               int M, int N, int K, int act, uint16_t* out_bf16, float* out_f32, int ldc, int vec_ok, void* stream) {
  const bool vec = vec_ok && (N % 4 == 0) && (ldc % 4 == 0);
  const bool one_out = (out_bf16 != nullptr) != (out_f32 != nullptr);
  if (!vec || !one_out || K % se::k6BK != 0 || K < 2 * se::k6BK || (act != SE_ACT_IDENTITY && act != SE_ACT_GELU)) return 1;
  G6Args g{A, lda, W, ldw, bias, residual_f32, M, N, K, out_bf16, out_f32, ldc, se::as_stream(stream)};
  const bool gelu = act == SE_ACT_GELU, res = residual_f32 != nullptr, obf = out_bf16 != nullptr;
  se::ProfScope prof(se::kProfGemm, 2.0 * M * (double)N * K, g.st);
  static int persistent = -1;
  if (persistent < 0) {
    const char* e = getenv("SE_AMD_GEMM6P");          // 0: one workgroup per tile for every shape
    persistent = e ? atoi(e) : 1;
  }
  // persistent form: bf16 output without residual, several tiles per CU, whole column tiles, an even number of K-tiles, 32-bit source offsets
  if (persistent && obf && !res && N % se::k6BN == 0 && N <= 8192 && (K / se::k6BK) % 2 == 0 && K >= 4 * se::k6BK && (ldc % 8) == 0 &&
      ((uintptr_t)out_bf16 % 16) == 0 && (size_t)M * lda < (1u << 31) && (size_t)N * ldw < (1u << 31) &&
      (size_t)((M + se::k6BM - 1) / se::k6BM) * (N / se::k6BN) > 256)
    return gelu ? launch6p<SE_ACT_GELU>(g) : launch6p<SE_ACT_IDENTITY>(g);
  if (!gelu && !res && obf) return launch6<SE_ACT_IDENTITY, 2>(g);
  if (gelu && !res && obf) return launch6<SE_ACT_GELU, 2>(g);
  if (!gelu && res && !obf) return launch6<SE_ACT_IDENTITY, 4 | 1>(g);
  if (!gelu && !res && !obf) return launch6<SE_ACT_IDENTITY, 4>(g);
  if (gelu && !res && !obf) return launch6<SE_ACT_GELU, 4>(g);
  return 1;
}

// y = A . W^T + bias as bf16 rows at out_pre AND gelu(y) (erf form, of the bf16-rounded y) at out_act, one launch of the persistent kernel: the
// training forward of the FFN's first projection, which keeps the pre-activation for the backward pass (encoder_train.hip; was se_gemm_bf16 +
// se_gelu_bf16).  Returns 1 when the shape is not the persistent kernel's (the caller then runs the two launches).
extern "C" int se_gemm6_dual_gelu_launch(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, int M, int N, int K, uint16_t* out_pre,
                                         uint16_t* out_act, int ldc, void* stream) {
  if (!(N % se::k6BN == 0 && N <= 8192 && K % se::k6BK == 0 && (K / se::k6BK) % 2 == 0 && K >= 4 * se::k6BK && (ldc % 8) == 0 && ldc >= N &&
        lda >= K && ldw >= K && lda % 8 == 0 && ldw % 8 == 0 && (size_t)M * lda < (1u << 31) && (size_t)N * ldw < (1u << 31) &&
        (size_t)((M + se::k6BM - 1) / se::k6BM) * (N / se::k6BN) > 256 &&
        (((uintptr_t)A | (uintptr_t)W | (uintptr_t)out_pre | (uintptr_t)out_act | (uintptr_t)bias) % 16) == 0))
    return 1;
  static int on = -1;
  if (on < 0) { const char* e = getenv("SE_AMD_GEMM6_DUAL"); on = e ? atoi(e) : 1; }      // A/B: 0 = two launches
  if (!on) return 1;
  const int tiles_m = (M + se::k6BM - 1) / se::k6BM, tiles_n = N / se::k6BN;
  static int group_m = 0, n_cu = 0, late_start = 0;
  static bool attr_set = false;
  if (!attr_set) {
    const char* gm = getenv("SE_AMD_GEMM_GROUPM");
    group_m = gm ? atoi(gm) : 4;
    if (group_m < 1) group_m = 1;
    const char* ls = getenv("SE_AMD_GEMM6P_LATE");
    late_start = ls ? atoi(ls) : 2;
    int dev = 0;
    hipDeviceProp_t prop;
    SE_HIP(hipGetDevice(&dev));
    SE_HIP(hipGetDeviceProperties(&prop, dev));
    n_cu = prop.multiProcessorCount & ~7;
    if (n_cu < 8) n_cu = 8;
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm6p_bf16_kernel<SE_ACT_IDENTITY, 2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, se::k6Lds + 32768));
    attr_set = true;
  }
  hipStream_t st = se::as_stream(stream);
  se::ProfScope prof(se::kProfGemm, 2.0 * M * (double)N * K, st);
  const int grid = std::min(n_cu, (tiles_m * tiles_n + 7) & ~7);
  hipLaunchKernelGGL((se::gemm6p_bf16_kernel<SE_ACT_IDENTITY, 2, 1>), dim3(grid), dim3(se::k6Threads), se::k6Lds + N * 4, st, A, lda, W, ldw, bias, M, N, K, out_pre,
                     ldc, tiles_m, tiles_n, group_m, late_start, (long long)(out_act - out_pre));
  SE_LAUNCH_CHECK();
  return SE_OK;
}

// out3 (M, 3 Kp) bf16 = the three-term activation operand [y1 | y1 | y2] of y = act(A . W^T + bias) (se_split3_bf16's layout, which = 0): the bf16x3
// parity mode's projections hand their result to the next projection in this form instead of writing fp32 and splitting it in another pass
// (round 4: 14 % of that mode's pass).  A / W are themselves three-term operands ([x1 | x1 | x2] . [w1 | w2 | w1]^T, depth K = 3 x the layer's);
// act: SE_ACT_IDENTITY or SE_ACT_GELU (exact erf form).
extern "C" int se_gemm_x3out_bf16(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, int M, int N, int K, int act,
                                  uint16_t* out3, int Kp, void* stream) {
  SE_REQUIRE(A && W && out3, "se_gemm_x3out_bf16: null argument");
  SE_REQUIRE(M > 0 && N > 0 && N % 4 == 0 && K % se::k6BK == 0 && K >= 2 * se::k6BK, "se_gemm_x3out_bf16: bad shape M=%d N=%d K=%d", M, N, K);
  // Kp == N: the epilogue writes columns [0, N) of each slice only (se_split3_bf16 zero-fills a pad, this producer does not: ADVICE r4)
  SE_REQUIRE(Kp == N && Kp % 8 == 0 && lda >= K && ldw >= K && lda % 8 == 0 && ldw % 8 == 0, "se_gemm_x3out_bf16: bad leading dimensions (Kp must equal N = %d)", N);
  SE_REQUIRE((((uintptr_t)A | (uintptr_t)W | (uintptr_t)out3 | (uintptr_t)bias) % 16) == 0, "se_gemm_x3out_bf16: operands must be 16-B aligned");
  SE_REQUIRE(act == SE_ACT_IDENTITY || act == SE_ACT_GELU, "se_gemm_x3out_bf16: act %d (identity or GELU)", act);
  G6Args g{A, lda, W, ldw, bias, nullptr, M, N, K, out3, nullptr, 3 * Kp, se::as_stream(stream)};
  se::ProfScope prof(se::kProfGemm, 2.0 * M * (double)N * K, g.st);
  return act == SE_ACT_GELU ? launch6<SE_ACT_GELU, 2 | 8>(g) : launch6<SE_ACT_IDENTITY, 2 | 8>(g);
}
