// mhsa.hip -- row B2 core: softmax(Q K^T / sqrt(64) + pad-mask) V for 64-wide heads, flash style (the
// (B, heads, T, T) score tensor the reference materialises -- 48 MB per utterance and layer -- never exists).
//
// qkv is the fused projection output (B*T, 3H) bf16 = [Q | K | V]; head h owns columns 64h..64h+63 of each.
// Workgroup = 4 waves = 128 query rows of one (utterance, head); wave = 32 query rows.  Per 64-key tile:
//   S^T = K Q^T   v_mfma_f32_32x32x16_bf16 with K as the A operand ("swapped"): each lane then holds 32 of the
//                 64 scores of ONE query row, so row max / sum are in-register (+1 exchange with lane^32)
//   P             exp2 with 1/sqrt(64) log2(e) folded in; online softmax, running max / sum per lane
//   O^T += V^T P^T  the S^T accumulator registers are used directly as the B operand (no LDS round trip, no
//                 lane movement); V^T fragments come from the row-major V tile through ds_read_b64_tr_b16
//                 (hardware-transposed LDS read), 4 consecutive keys x one d column per lane, in the same
//                 permuted k order as the accumulator registers.  O^T keeps the query on the lane, so the
//                 online-softmax rescale is lane-local.
// K / V tiles: global -> registers -> LDS, double buffered, XOR-swizzled 16-B chunks (conflict-free b128 and
// tr_b16 reads).  Keys >= lengths[b] are excluded (the reference adds -10000, i.e. exp() == 0 in fp32).
// Bound: MFMA (4 T^2 64 flop per head) -- with d = 64 the exp stream (v_exp_f32) is the co-limiter.
#include <stdlib.h>
#include "common.h"
#include "prof.h"
#include "bf16.h"
#include "mhsa_tile.h"
#include "dropout.h"

// developer ablation (timing only, results are wrong): build with SE_AMD_EXTRA_DEFINES=-DSE_MHSA_ABL=<mask>: 1 no K/V staging, 2 no barrier,
// 4 no exponentials, 8 no PV MFMAs, 16 no QK MFMAs, 32 no LDS fragment reads, 64 every workgroup stages the K / V of (utterance 0, head 0): all L2 hits
#ifndef SE_MHSA_ABL
#define SE_MHSA_ABL 0
#endif

namespace se {

// DROP = 1 (training): the attention probabilities are dropped (counter-based mask of dropout.h, site key `dkey`) AFTER the
// row sum, i.e. O = (P . mask / (1 - p)) V with P normalised by the full sum -- torch's dropout(softmax(.)) V
// PRE = 1 (inference): the queries arrive PRE-SCALED by log2(e) / sqrt(64) (the encoder's inference copy of the QKV weights folds it into the
// query rows before their bf16 rounding), so K Q'^T is already the base-2 exponent, and the running reference starts at 0 instead of -inf.
// Tiles are then SPECULATIVE: P = exp2(S) against the reference 0 with no row maximum, no lane exchange, no reference update and no
// rescale test (~50 of the ~190 vector instructions of a wave's key tile, in a loop whose vector issue is the bound: PMC 56-63 % vector
// issue against 36 % matrix pipe, profiles/r02*_pmc_sq_mhsa.json); the scores stay intact in their accumulators, and a row sum outside
// [2^-60, 2^60) (overflow, inf / nan, or -- first tile -- a row far below the reference) sends the wave to the exact online-softmax tile
// for this tile and every later one.  fp32 / bf16 keep their relative precision anywhere in that range.  Same box: 133 us speculative
// against 147 us always-exact (SE_AMD_MHSA_SPEC=0) per B = 32 launch.
// DMA = 1: the K / V tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write, no wait in front of the writes); the
// destination of a wave instruction is lane-linear (8 rows x 128 B), so the XOR swizzle of kv_off() is applied to the per-lane SOURCE chunk.
// The compile-time ablation (-DSE_MHSA_ABL) put the register staging at 27 % of the launch.
// NW = waves per workgroup (4, or 8 with the register staging only): 8 waves share one staged K / V tile, so each wave stages half as much per tile
template <int OCC, int DROP, int PRE, int DMA = 0, int NW = 4>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void mhsa_fwd_kernel(
    const uint16_t* __restrict__ qkv, const int32_t* __restrict__ lengths, int T, int H, uint16_t* __restrict__ ctx,
    float* __restrict__ lse, uint32_t dkey, uint32_t thr16, float dscale) {
  __shared__ __attribute__((aligned(16))) char smem[(DMA == 2 ? 3 : 2) * 2 * kAK * kHD * 2];   // 2 (DMA == 2: 3) slots x (K, V) x 8 KiB = 32 (48) KiB

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
#ifdef SE_AMD_STAMPS
  // developer build only (SE_AMD_BUILD_STAMPS=1; tools/mhsa_stamps.py): per-wave sums of the s_memtime spent in each phase of the key tile.  The
  // PRE instantiations take the stamp buffer through `lse` (unused in inference).  Phases: 0 top -> QK^T MFMAs issued, 1 -> first exponential
  // issued (= matrix results back), 2 -> last bf16 pack issued, 3 -> PV MFMAs issued (V^T fragment reads inside), 4 -> next tile written to LDS
  // (global-load wait inside), 5 -> barrier passed; 6 prologue, 7 whole kernel
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
  unsigned long long st_prev = st_t0;
#define SE_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_prev; st_prev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define SE_STAMP_PIN2(a, b) asm volatile("" :: "v"(a), "v"(b))
#else
#define SE_STAMP(i) do { } while (0)
#define SE_STAMP_PIN2(a, b) do { } while (0)
#endif
  // XCD-aware work mapping: workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with its own L2.  All
  // query tiles of one (utterance, head) re-read the same K / V, so they are placed on ONE XCD, consecutive in its
  // dispatch order: workgroup (xcd, i) -> pair 8 (i / nqt) + xcd, query tile i % nqt   [needs pairs % 8 == 0]
  int b, head, qt;
  {
    const int nqt = gridDim.x, pairs = gridDim.y * gridDim.z;
    const int lin = blockIdx.x + nqt * (blockIdx.y + gridDim.y * blockIdx.z);
    if ((pairs & 7) == 0) {
      const int xcd = lin & 7, i = lin >> 3;
      const int pair = 8 * (i / nqt) + xcd;
      qt = i % nqt;
      head = pair % gridDim.y;
      b = pair / gridDim.y;
    } else {
      qt = blockIdx.x; head = blockIdx.y; b = blockIdx.z;
    }
  }
  static_assert(NW == 4 || (NW == 8 && DMA == 0), "8-wave workgroups: register staging only");
  const int q0 = qt * (NW * 32) + wave * 32;
  const int ld = 3 * H;
  const int len = lengths ? min(max(lengths[b], 1), T) : T;
  const int nkt = (len + kAK - 1) / kAK;
  const uint16_t* base = qkv + (size_t)b * T * ld + head * kHD;

  // ---- Q fragments (B operand of S^T = K Q^T): lane -> query row q0 + l31, d = 16 s + 8 hh .. +7
  bf16x8 qf[4];
  {
    const int q = min(q0 + l31, T - 1);
    const uint16_t* qp = base + (size_t)q * ld + 8 * hh;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
  }

  // ---- staging: K and V tiles are 64 rows x 128 B; 256 threads x 16 B = 32 rows per pass
  const int srow = tid >> 3, sch = tid & 7;
  const uint16_t* kvbase = (SE_MHSA_ABL & 64) ? qkv : base;
  const uint16_t* kp = kvbase + H + sch * 8;
  const uint16_t* vp = kvbase + 2 * H + sch * 8;
  uint4 rk0, rk1, rv0, rv1;
  const int so0 = kv_off(srow, sch), so1 = kv_off((srow + 32) & 63, sch);      // NW == 8: 512 threads x 16 B = the whole 64-row tile in one pass
#define SE_A_ISSUE(kt)                                                                       \
  do {                                                                                       \
    const size_t r0 = (size_t)min((kt) * kAK + srow, T - 1) * ld;                            \
    const size_t r1 = (size_t)min((kt) * kAK + srow + 32, T - 1) * ld;                       \
    rk0 = *reinterpret_cast<const uint4*>(kp + r0);                                          \
    if (NW == 4) rk1 = *reinterpret_cast<const uint4*>(kp + r1);                             \
    rv0 = *reinterpret_cast<const uint4*>(vp + r0);                                          \
    if (NW == 4) rv1 = *reinterpret_cast<const uint4*>(vp + r1);                             \
  } while (0)
#define SE_A_WRITE(buf)                                                 \
  do {                                                                  \
    char* k_w = smem + (buf) * 16384;                                   \
    char* v_w = k_w + 8192;                                             \
    *reinterpret_cast<uint4*>(k_w + so0) = rk0;                         \
    if (NW == 4) *reinterpret_cast<uint4*>(k_w + so1) = rk1;            \
    *reinterpret_cast<uint4*>(v_w + so0) = rv0;                         \
    if (NW == 4) *reinterpret_cast<uint4*>(v_w + so1) = rv1;            \
  } while (0)
  // LDS-DMA form: wave w brings rows [16 w, 16 w + 16) of the K and of the V tile, two 1-KiB pieces (8 rows x 128 B) each; lane l of a piece
  // writes slot l & 7 of row l >> 3, i.e. it must FETCH chunk (l & 7) ^ f(row) (kv_off: slot = chunk ^ f)
  typedef __attribute__((address_space(1))) const void* glb_a_t;
  typedef __attribute__((address_space(3))) void* lds_a_t;
  const int drow = wave * 16 + (lane >> 3);                 // tile row of this lane in piece 0 (piece 1: + 8)
  const int dch0 = ((lane & 7) ^ (kv_off(drow, 0) >> 4 & 7)) * 8, dch1 = ((lane & 7) ^ (kv_off(drow + 8, 0) >> 4 & 7)) * 8;
#define SE_A_DMA(kt, buf)                                                                                                  \
  do {                                                                                                                     \
    const size_t r0 = (size_t)min((kt) * kAK + drow, T - 1) * ld;                                                          \
    const size_t r1 = (size_t)min((kt) * kAK + drow + 8, T - 1) * ld;                                                      \
    char* k_w = smem + (buf) * 16384 + wave * 2048;                                                                        \
    __builtin_amdgcn_global_load_lds((glb_a_t)(base + H + r0 + dch0), (lds_a_t)(k_w), 16, 0, 0);                           \
    __builtin_amdgcn_global_load_lds((glb_a_t)(base + H + r1 + dch1), (lds_a_t)(k_w + 1024), 16, 0, 0);                    \
    __builtin_amdgcn_global_load_lds((glb_a_t)(base + 2 * H + r0 + dch0), (lds_a_t)(k_w + 8192), 16, 0, 0);                \
    __builtin_amdgcn_global_load_lds((glb_a_t)(base + 2 * H + r1 + dch1), (lds_a_t)(k_w + 8192 + 1024), 16, 0, 0);         \
  } while (0)

  const f32x16 kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 o0, o1;                      // O^T d-blocks 0 / 1: col = query (lane & 31), row = d
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = PRE ? 0.f : -INFINITY, l_run = 0.f;
  bool slow = dscale < 0.f;           // wave-uniform: the speculative (no row maximum) path failed once (PRE only); dscale < 0: never speculate (A/B switch)
  const float c = PRE ? 1.0f : 0.125f * 1.44269504088896340736f;     // 1/sqrt(64) * log2(e), unless the queries carry it already
  constexpr float kDefer = 8.f;
  typedef float f2 __attribute__((ext_vector_type(2)));

  // ---- loop-invariant LDS byte offsets (buffer / key-block / k-step parts are compile-time immediates below)
  //   K fragment (b128): row l31 (+32 for key block 1 = +4096 B), logical chunk 2 s + hh -> 4 per-lane values
  //   V^T fragment (tr_b16): lane addresses key tq + 4 hh (+8 for elements 4..7), columns 32 dblk + 16 g1 + 4 tp;
  //   adding 16 s + 32 kb to the key leaves the swizzle term unchanged (it reads key bits 1-3) -> +2048 s + 4096 kb
  int koff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) koff[s] = kv_off(l31, 2 * s + hh);
  const int tq = (lane & 15) >> 2, tp = lane & 3, g1 = (lane >> 4) & 1;
  int voff[2][2];                     // [dblk][lo / hi]
#pragma unroll
  for (int dblk = 0; dblk < 2; ++dblk) {
    const int dcol = dblk * 32 + 16 * g1 + 4 * tp;
    voff[dblk][0] = 8192 + kv_off(4 * hh + tq, dcol >> 3) + (dcol & 7) * 2;
    voff[dblk][1] = 8192 + kv_off(4 * hh + tq + 8, dcol >> 3) + (dcol & 7) * 2;
  }

  // dropout: pair index of (this lane's query row, key) = row_id * ceil(T / 2) + key / 2
  const uint32_t drop_row = ((uint32_t)(b * (H / kHD) + head) * (uint32_t)T + (uint32_t)min(q0 + l31, T - 1)) * (uint32_t)((T + 1) >> 1);
  // DROP == 2: the mask comes as the query-major bit matrix of dropmask.hip -- per 64-key tile one 8-byte load per lane (fetched one tile ahead),
  // word .x = the tile's 32 even keys, .y = its odd keys; the kept probabilities are NOT scaled here: 1 / (1 - p) goes into the final 1 / l
  const uint32_t* mrow = nullptr;
  uint2 mw_nxt = make_uint2(0u, 0u);
  if (DROP == 2) {
    // the matrix address travels in the two hash arguments this mode does not use (dkey = low, thr16 = high word): one kernel signature for all
    // modes -- two more kernel arguments moved the register allocation of the inference instantiation past its 168-register budget (2 spills)
    const uint32_t* dmask = reinterpret_cast<const uint32_t*>(((uint64_t)thr16 << 32) | (uint64_t)dkey);
    const int dmask_w = 4 * ((T + 127) >> 7);
    mrow = dmask + ((size_t)(b * (H / kHD) + head) * T + min(q0 + l31, T - 1)) * dmask_w;
    mw_nxt = *reinterpret_cast<const uint2*>(mrow);
  }

  if (DMA == 2) {
    // three slots, tiles kt + 1 AND kt + 2 in flight: the first toucher of a K / V tile takes an L2 miss (8 query tiles share it: 12.5 % compulsory
    // misses), and one tile of lookahead does not cover that round trip under load
    SE_A_DMA(0, 0);
    if (nkt > 1) {
      SE_A_DMA(1, 1);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
  } else if (DMA) {
    SE_A_DMA(0, 0);
  } else {
    SE_A_ISSUE(0);
    SE_A_WRITE(0);
  }
  if (DMA != 2) __syncthreads();
  SE_STAMP(6);

#ifndef SE_MHSA_PRIO
#define SE_MHSA_PRIO 0          /* bit 0: raised wave priority around the QK^T products, bit 1: around the PV products */
#endif
#ifndef SE_MHSA_PRIO_LVL
#define SE_MHSA_PRIO_LVL 3
#endif
#define SE_A_TILE(CUR)                                                                                                     \
  {                                                                                                                        \
    if (DMA == 2) { if (kt + 2 < nkt) SE_A_DMA(kt + 2, (CUR) >= 1 ? (CUR) - 1 : 2); }                                      \
    else if (DMA == 3) { if (kt + 1 < nkt) SE_A_DMA(kt + 1, (CUR) ^ 1); }                                                  \
    else if (kt + 1 < nkt && !(SE_MHSA_ABL & 1)) { if (DMA) SE_A_DMA(kt + 1, (CUR) ^ 1); else SE_A_ISSUE(kt + 1); }       \
    uint2 mw_cur = mw_nxt;                                                                                                 \
    if (DROP == 2 && kt + 1 < nkt) mw_nxt = *reinterpret_cast<const uint2*>(mrow + 2 * (kt + 1));                          \
    const char* t_s = smem + (CUR) * 16384;                                                                                \
    f32x16 s0, s1;                                                                                                         \
    if (SE_MHSA_PRIO & 1) __builtin_amdgcn_s_setprio(SE_MHSA_PRIO_LVL);                                                                    \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                                        \
      const bf16x8 ka = (SE_MHSA_ABL & 32) ? qf[s] : *reinterpret_cast<const bf16x8*>(t_s + koff[s]);                      \
      const bf16x8 kb_ = (SE_MHSA_ABL & 32) ? qf[3 - s] : *reinterpret_cast<const bf16x8*>(t_s + koff[s] + 4096);          \
      /* first product of the chain takes the inline constant 0 as C: no 32 v_mov per tile to clear the accumulators */      \
      if (SE_MHSA_ABL & 16) { if (s == 0) { _Pragma("unroll") for (int r = 0; r < 16; ++r) { s0[r] = __builtin_bit_cast(f32x4, ka)[r & 3] * 1e-30f; s1[r] = __builtin_bit_cast(f32x4, kb_)[r & 3] * 1e-30f; } } } else { \
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[s], s == 0 ? kZero16 : s0, 0, 0, 0);                              \
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb_, qf[s], s == 0 ? kZero16 : s1, 0, 0, 0); }                           \
    }                                                                                                                      \
    if (SE_MHSA_PRIO & 1) __builtin_amdgcn_s_setprio(0);                                                                   \
    SE_STAMP(0);                                                                                                           \
    if ((kt + 1) * kAK > len) {                                                                                            \
      const int kbase = kt * kAK + 4 * hh;                                                                                 \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                                     \
        const int key = kbase + (r & 3) + 8 * (r >> 2);                                                                    \
        if (key >= len) s0[r] = -INFINITY;                                                                                 \
        if (key + 32 >= len) s1[r] = -INFINITY;                                                                            \
      }                                                                                                                    \
    }                                                                                                                      \
    bf16x8 pf[2][2];                                                                                                       \
    bool spec_ok = false;                                                                                                  \
    if (PRE && !DROP && !slow) {                                                                                           \
      /* SPECULATIVE tile: probabilities against the initial reference 0, no row maximum at all -- 16 v_max3, the lane exchange,  */ \
      /* the reference update and the rescale test are ~30 of the tile's ~135 vector instructions (8 % of the launch, measured).  */ \
      /* The scores stay intact in s0 / s1 (the exponentials go straight into the bf16 fragments), and the row sums tell whether   */ \
      /* the speculation held: a partial sum that left [2^-60, 2^60) (overflow / inf / nan; on the first tile also underflow: a    */ \
      /* row far below the reference) sends the wave down the exact online-softmax path below, for this tile and all later ones.  */ \
      float rs0 = 0.f, rs1 = 0.f;                                                                                          \
      _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                                        \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                                    \
          const float a0 = (SE_MHSA_ABL & 4) ? s0[8 * s + j] * 0.5f + 1.0f : __builtin_amdgcn_exp2f(s0[8 * s + j]);        \
          const float a1 = (SE_MHSA_ABL & 4) ? s1[8 * s + j] * 0.5f + 1.0f : __builtin_amdgcn_exp2f(s1[8 * s + j]);        \
          rs0 += a0;                                                                                                       \
          rs1 += a1;                                                                                                       \
          pf[0][s][j] = (__bf16)a0;                                                                                        \
          pf[1][s][j] = (__bf16)a1;                                                                                        \
          if (s == 0 && j == 0) { SE_STAMP_PIN2(a0, a1); SE_STAMP(1); }                                                    \
        }                                                                                                                  \
      SE_STAMP_PIN2(pf[0][0], pf[0][1]); SE_STAMP_PIN2(pf[1][0], pf[1][1]);                                                \
      SE_STAMP(2);                                                                                                         \
      const float rs = rs0 + rs1;                                                                                          \
      const bool bad = !(rs < 0x1p60f) || (kt == 0 && rs < 0x1p-60f);                                                      \
      if (!__any(bad)) {                                                                                                   \
        l_run += rs;                                                                                                       \
        spec_ok = true;                                                                                                    \
      } else {                                                                                                             \
        slow = true;                                                                                                       \
      }                                                                                                                    \
    }                                                                                                                      \
    if (!spec_ok) {                                                                                                        \
    float mx = fmaxf(s0[0], s1[0]);                                                                                        \
    _Pragma("unroll") for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s0[r]), s1[r]);       /* v_max3_f32 */            \
    {   /* the other 32 keys of this query row live in lane ^ 32: v_permlane32_swap (VALU) instead of ds_bpermute (an LDS round trip */ \
        /* of ~100 cycles in the middle of the tile's dependency chain)                                                            */ \
      const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);           \
      mx = fmaxf(__uint_as_float(sw_[0]), __uint_as_float(sw_[1]));                                                        \
    }                                                                                                                      \
    /* deferred rescale: the reference maximum only moves when the tile maximum exceeds it by > 2^kDefer (log2 domain), */  \
    /* so P <= 2^kDefer instead of 1 (bf16 / fp32 keep their relative precision) and the O-wide multiply is rare        */  \
    float m_new = ((mx - m_run) * c > kDefer) ? mx : m_run;                                                                \
    if (PRE && kt == 0 && mx < -64.f) m_new = mx;        /* a first tile far below the initial reference 0 (later tiles cannot matter) */ \
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);                                                       \
    /* scalar fp32 ops on purpose (this file is built with -fno-slp-vectorize): v_pk_fma_f32 / v_pk_add_f32 cost more issue time than   */  \
    /* the two plain instructions they replace in this VALU-bound loop (148 -> 141 us per launch)                               */  \
    const float mc = -m_new * c;                                                                                           \
    float rs0 = 0.f, rs1 = 0.f;                                                                                            \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                                       \
      const float a0 = __builtin_amdgcn_exp2f(PRE ? s0[r] + mc : fmaf(s0[r], c, mc));                                      \
      const float a1 = __builtin_amdgcn_exp2f(PRE ? s1[r] + mc : fmaf(s1[r], c, mc));                                      \
      rs0 += a0;                                                                                                           \
      rs1 += a1;                                                                                                           \
      s0[r] = a0; s1[r] = a1;                                                                                              \
    }                                                                                                                      \
    const f2 rs2 = {rs0, rs1};                                                                                             \
    l_run = fmaf(l_run, alpha, rs2.x + rs2.y);                                                                             \
    m_run = m_new;                                                                                                         \
    if (__any(alpha != 1.0f)) {                                                                                            \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }                                   \
    }                                                                                                                      \
    if (DROP == 2) {                                                                                                       \
      const uint32_t we_ = mw_cur.x >> (2 * hh), wo_ = mw_cur.y >> (2 * hh);                                               \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                                     \
        /* key 4 hh + (r & 3) + 8 (r >> 2) (+ 32): pair 2 hh + ((r & 3) >> 1) + 4 (r >> 2) (+ 16), parity r & 1 */         \
        const int pos = ((r & 3) >> 1) + 4 * (r >> 2);                                                                     \
        const uint32_t w_ = (r & 1) ? wo_ : we_;                                                                           \
        s0[r] = __uint_as_float(__float_as_uint(s0[r]) & (uint32_t)((int32_t)(w_ << (31 - pos)) >> 31));                   \
        s1[r] = __uint_as_float(__float_as_uint(s1[r]) & (uint32_t)((int32_t)(w_ << (15 - pos)) >> 31));                   \
      }                                                                                                                    \
    } else if (DROP) {                                                                                                     \
      const uint32_t pb = drop_row + (uint32_t)((kt * kAK + 4 * hh) >> 1);                                                 \
      _Pragma("unroll") for (int r = 0; r < 16; r += 2) {                                                                  \
        const uint32_t off = (uint32_t)(((r & 3) + 8 * (r >> 2)) >> 1);                                                    \
        const uint32_t b0 = dropout_bits(dkey, pb + off), b1 = dropout_bits(dkey, pb + off + 16);                          \
        s0[r] *= dropout_mul(b0, 0, thr16, dscale); s0[r + 1] *= dropout_mul(b0, 1, thr16, dscale);                        \
        s1[r] *= dropout_mul(b1, 0, thr16, dscale); s1[r + 1] *= dropout_mul(b1, 1, thr16, dscale);                        \
      }                                                                                                                    \
    }                                                                                                                      \
    _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                                          \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                                      \
        pf[0][s][j] = (__bf16)s0[8 * s + j];                                                                               \
        pf[1][s][j] = (__bf16)s1[8 * s + j];                                                                               \
      }                                                                                                                    \
    }                                                                                                                      \
    if (SE_MHSA_PRIO & 2) __builtin_amdgcn_s_setprio(SE_MHSA_PRIO_LVL);                                                    \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                                       \
      _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                                      \
        _Pragma("unroll") for (int dblk = 0; dblk < 2; ++dblk) {                                                           \
          bf16x8 va = qf[2 * kb + s];                                                                                       \
          if (!(SE_MHSA_ABL & 32)) {                                                                                       \
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(                                                      \
              (__attribute__((address_space(3))) bf16x4*)(t_s + voff[dblk][0] + kb * 4096 + s * 2048));                    \
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(                                                      \
              (__attribute__((address_space(3))) bf16x4*)(t_s + voff[dblk][1] + kb * 4096 + s * 2048));                    \
          va = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};                                             \
          }                                                                                                                \
          if (SE_MHSA_ABL & 8) { o0[2 * kb + s] += __builtin_bit_cast(f32x4, va)[0] * __builtin_bit_cast(f32x4, pf[kb][s])[dblk]; } else { \
          if (dblk == 0) o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pf[kb][s], o0, 0, 0, 0);                         \
          else o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pf[kb][s], o1, 0, 0, 0); }                                 \
        }                                                                                                                  \
      }                                                                                                                    \
    if (SE_MHSA_PRIO & 2) __builtin_amdgcn_s_setprio(0);                                                                   \
    SE_STAMP(3);                                                                                                           \
    if (!DMA && kt + 1 < nkt && !(SE_MHSA_ABL & 1)) SE_A_WRITE((CUR) ^ 1);                                                 \
    SE_STAMP(4);                                                                                                           \
    if (DMA == 2) {                                                                                                        \
      if (kt + 2 < nkt) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  \
      __builtin_amdgcn_s_barrier();                                                                                        \
    } else if (!(SE_MHSA_ABL & 2)) __syncthreads();                                                                        \
    SE_STAMP(5);                                                                                                           \
  }

  int kt = 0;
  if (DMA == 2) {
    // ONE tile body with a run-time slot (three unrolled bodies spill at 3 waves per SIMD)
    int slot = 0;
    for (; kt < nkt; ++kt) {
      SE_A_TILE(slot)
      slot = slot == 2 ? 0 : slot + 1;
    }
  } else if (DMA == 3) {
    int slot = 0;
    for (; kt < nkt; ++kt) {
      SE_A_TILE(slot)
      slot ^= 1;
    }
  } else {
    // unrolled by two so the double-buffer offset is an immediate
    for (; kt + 1 < nkt; kt += 2) {
      SE_A_TILE(0)
      ++kt;
      SE_A_TILE(1)
      --kt;
    }
    if (kt < nkt) SE_A_TILE(0)
  }
#undef SE_A_TILE

  // ---- epilogue: O / l ; lane holds query q0 + l31, d = 32 dblk + (r&3) + 8 (r>>2) + 4 hh
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = (DROP == 2 ? dscale : 1.0f) / l_tot;
  const int q = q0 + l31;
  // training: log2-domain log-sum-exp of the scaled scores, P = exp2(c s - lse) in the backward kernels
#ifdef SE_AMD_STAMPS
  const bool lse_is_stamps = PRE;          // stamp builds: the PRE instantiations receive the stamp buffer through `lse`
#else
  const bool lse_is_stamps = false;
#endif
  if (lse && !lse_is_stamps && q < T && hh == 0) lse[((size_t)b * (H / kHD) + head) * T + q] = fmaf(m_run, c, __builtin_amdgcn_logf(l_tot));
  if (q < T) {
    uint16_t* op = ctx + ((size_t)b * T + q) * H + head * kHD + 4 * hh;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint2 w0 = make_uint2(pack_bf16x2(o0[4 * g] * inv, o0[4 * g + 1] * inv), pack_bf16x2(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv));
      uint2 w1 = make_uint2(pack_bf16x2(o1[4 * g] * inv, o1[4 * g + 1] * inv), pack_bf16x2(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv));
      *reinterpret_cast<uint2*>(op + 8 * g) = w0;
      *reinterpret_cast<uint2*>(op + 32 + 8 * g) = w1;
    }
  }
#ifdef SE_AMD_STAMPS
  if (PRE && lse) {
    st_acc[7] = __builtin_amdgcn_s_memtime() - st_t0;
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    unsigned long long* sp = reinterpret_cast<unsigned long long*>(lse) + ((size_t)lin * NW + wave) * 8;
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) sp[i] = st_acc[i];
    }
  }
#endif
}


}  // namespace se

int se_mhsaN_fwd_launch(const uint16_t* qkv, const int32_t* lengths, int B, int T, int heads, uint16_t* ctx, int never_speculate, int nw, int wpe, hipStream_t st);   // mhsa8.hip
#ifdef SE_AMD_EXPERIMENTS      // the parked attention forwards (tools/experiments/kernels/, mhsa8.hip's other layouts): developer builds only
int se_mhsa_fwd_pipe_launch(const uint16_t* qkv, const int32_t* lengths, int B, int T, int heads, uint16_t* ctx, int occ, hipStream_t st);   // mhsa_pipe.hip
int se_mhsa2_fwd_launch(const uint16_t* qkv, const int32_t* lengths, int B, int T, int heads, uint16_t* ctx, int never_speculate, hipStream_t st);   // mhsa2.hip
int se_mhsa3_fwd_launch(const uint16_t* qkv, const int32_t* lengths, int B, int T, int heads, uint16_t* ctx, int never_speculate, hipStream_t st);   // mhsa3.hip
int se_mhsa8_fwd_launch(const uint16_t* qkv, const int32_t* lengths, int B, int T, int heads, uint16_t* ctx, int never_speculate, hipStream_t st);   // mhsa8.hip
int se_mhsa9_fwd_launch(const uint16_t* qkv, const int32_t* lengths, int B, int T, int heads, uint16_t* ctx, int never_speculate, hipStream_t st);   // mhsa9.hip
int se_mhsaP_fwd_launch(const uint16_t* qkv, const int32_t* lengths, int B, int T, int heads, uint16_t* ctx, int never_speculate, int wgs, hipStream_t st);   // mhsa8.hip
#endif

static int mhsa_fwd_launch(const uint16_t* qkv, const int32_t* lengths, int B, int T, int heads, uint16_t* ctx, float* lse, float dropout_p,
                           uint64_t seed, uint32_t site, void* stream) {
  SE_REQUIRE(qkv && ctx, "se_mhsa_fwd_bf16: null argument");
  SE_REQUIRE(B > 0 && B <= 65535 && T > 0 && heads > 0 && heads <= 65535, "se_mhsa_fwd_bf16: bad shape B=%d T=%d heads=%d", B, T, heads);
  const int H = heads * se::kHD;
  dim3 grid((T + se::kAQ - 1) / se::kAQ, heads, B);
  se::ProfScope prof(se::kProfMhsa, 4.0 * B * (double)heads * T * (double)T * se::kHD, se::as_stream(stream));
  static int occ = -1;
  if (occ < 0) {
    const char* e = getenv("SE_AMD_MHSA_OCC");
    occ = e ? atoi(e) : 3;
  }
  const se::DropoutCfg d = se::make_dropout(dropout_p, seed);
  if (d.thr16) {
    SE_REQUIRE((double)B * heads * T * ((T + 1) / 2) < 4294967296.0, "se_mhsa_fwd: dropout pair index exceeds 32 bits");
    hipLaunchKernelGGL((se::mhsa_fwd_kernel<2, 1, 0>), grid, dim3(256), 0, se::as_stream(stream), qkv, lengths, T, H, ctx, lse,
                       se::dropout_key(seed, site), d.thr16, d.scale);
  } else if (occ == 2) {
    hipLaunchKernelGGL((se::mhsa_fwd_kernel<2, 0, 0>), grid, dim3(256), 0, se::as_stream(stream), qkv, lengths, T, H, ctx, lse, 0u, 0u, 1.f);
  } else {
    hipLaunchKernelGGL((se::mhsa_fwd_kernel<3, 0, 0>), grid, dim3(256), 0, se::as_stream(stream), qkv, lengths, T, H, ctx, lse, 0u, 0u, 1.f);
  }
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_mhsa_fwd_bf16(const uint16_t* qkv, const int32_t* lengths, int B, int T, int heads, uint16_t* ctx, void* stream) {
  return mhsa_fwd_launch(qkv, lengths, B, T, heads, ctx, nullptr, 0.f, 0, 0, stream);
}

static int mhsa_prescaled_launch(const uint16_t* qkv, const int32_t* lengths, int B, int T, int heads, uint16_t* ctx, int pipe, void* stream) {
  SE_REQUIRE(qkv && ctx, "se_mhsa_fwd_prescaled_bf16: null argument");
  SE_REQUIRE(B > 0 && B <= 65535 && T > 0 && heads > 0 && heads <= 65535, "se_mhsa_fwd_prescaled_bf16: bad shape B=%d T=%d heads=%d", B, T, heads);
  const int H = heads * se::kHD;
  dim3 grid((T + se::kAQ - 1) / se::kAQ, heads, B);
  se::ProfScope prof(se::kProfMhsa, 4.0 * B * (double)heads * T * (double)T * se::kHD, se::as_stream(stream));
  static int spec = -1;
  if (spec < 0) { const char* e = getenv("SE_AMD_MHSA_SPEC"); spec = e ? atoi(e) : 1; }      // 0: always the exact online-softmax tile (A/B)
  // 10: 8 free-running waves on one LDS-DMA staged tile, two workgroups per CU, waves 4-7 half a tile behind (mhsa8.hip; the default from 768 workgroups on)
  if (pipe == 10) return se_mhsaN_fwd_launch(qkv, lengths, B, T, heads, ctx, spec ? 0 : 1, 8, 4, se::as_stream(stream));
#ifdef SE_AMD_EXPERIMENTS
  static int pipe_occ = -1;
  if (pipe_occ < 0) {
    const char* o = getenv("SE_AMD_MHSA_PIPE_OCC");
    pipe_occ = o ? atoi(o) : 2;
  }
  if (pipe == 1) return se_mhsa_fwd_pipe_launch(qkv, lengths, B, T, heads, ctx, pipe_occ, se::as_stream(stream));
  if (pipe == 12) return se_mhsa9_fwd_launch(qkv, lengths, B, T, heads, ctx, spec ? 0 : 1, se::as_stream(stream));      // 8 waves, software-pipelined over the key tiles (mhsa9.hip)
  if (pipe == 11 || pipe == 13 || pipe == 14)      // variant 10 as persistent workgroups (mhsa8.hip); 13 / 14: with 8 / 5 workgroups (tests: long item lists, both list forms)
    return se_mhsaP_fwd_launch(qkv, lengths, B, T, heads, ctx, spec ? 0 : 1, pipe == 11 ? 0 : (pipe == 13 ? 8 : 5), se::as_stream(stream));
  if (pipe == 9 || pipe == 16) return se_mhsaN_fwd_launch(qkv, lengths, B, T, heads, ctx, spec ? 0 : 1, pipe == 16 ? 16 : 8, pipe == 9 ? 2 : 4, se::as_stream(stream));
  if (pipe == 8) return se_mhsa8_fwd_launch(qkv, lengths, B, T, heads, ctx, spec ? 0 : 1, se::as_stream(stream));      // 8-wave alternating segments (mhsa8.hip)
  if (pipe == 3) return se_mhsa3_fwd_launch(qkv, lengths, B, T, heads, ctx, spec ? 0 : 1, se::as_stream(stream));      // interleaved matrix / vector stream, two query blocks per wave (mhsa3.hip)
  if (pipe == 2) return se_mhsa2_fwd_launch(qkv, lengths, B, T, heads, ctx, spec ? 0 : 1, se::as_stream(stream));      // two query blocks per wave (mhsa2.hip)
  static int dma = -1;
  if (dma < 0) { const char* e = getenv("SE_AMD_MHSA_DMA"); dma = e ? atoi(e) : 0; }        // A/B: LDS-DMA rings inside this file's kernel (all measured equal or slower)
  static int nw8 = -1;
  if (nw8 < 0) { const char* e = getenv("SE_AMD_MHSA_NW"); nw8 = (e && atoi(e) == 8) ? 1 : 0; }      // A/B: 8-wave workgroups (256 queries share a staged tile)
  if (dma == 3)
    hipLaunchKernelGGL((se::mhsa_fwd_kernel<4, 0, 1, 3>), grid, dim3(256), 0, se::as_stream(stream), qkv, lengths, T, H, ctx, nullptr, 0u, 0u, spec ? 1.f : -1.f);
  else if (dma == 4)
    hipLaunchKernelGGL((se::mhsa_fwd_kernel<3, 0, 1, 3>), grid, dim3(256), 0, se::as_stream(stream), qkv, lengths, T, H, ctx, nullptr, 0u, 0u, spec ? 1.f : -1.f);
  else if (dma == 2)
    hipLaunchKernelGGL((se::mhsa_fwd_kernel<3, 0, 1, 2>), grid, dim3(256), 0, se::as_stream(stream), qkv, lengths, T, H, ctx, nullptr, 0u, 0u, spec ? 1.f : -1.f);
  else if (dma)
    hipLaunchKernelGGL((se::mhsa_fwd_kernel<3, 0, 1, 1>), grid, dim3(256), 0, se::as_stream(stream), qkv, lengths, T, H, ctx, nullptr, 0u, 0u, spec ? 1.f : -1.f);
  else if (nw8) {
    dim3 grid8((T + 255) / 256, heads, B);
    hipLaunchKernelGGL((se::mhsa_fwd_kernel<2, 0, 1, 0, 8>), grid8, dim3(512), 0, se::as_stream(stream), qkv, lengths, T, H, ctx, nullptr, 0u, 0u, spec ? 1.f : -1.f);
  } else
#else
  if (pipe != 0) {
    se::set_error("se_mhsa_fwd_prescaled: variant %d is a parked experiment (developer builds: SE_AMD_BUILD_EXPERIMENTS=kernels); the product has 0 and 10", pipe);
    return SE_ERR_UNSUPPORTED;
  }
#endif
    hipLaunchKernelGGL((se::mhsa_fwd_kernel<3, 0, 1, 0>), grid, dim3(256), 0, se::as_stream(stream), qkv, lengths, T, H, ctx, nullptr, 0u, 0u, spec ? 1.f : -1.f);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

#ifdef SE_AMD_STAMPS
// developer entry of stamp builds (not in include/se_amd.h): the inference kernel at 1 / 2 / 3 waves per SIMD with the per-phase cycle sums of every
// wave written to `stamps` (grid workgroups x 4 waves x 8 uint64)
extern "C" int se_mhsa_fwd_stamps_bf16(const uint16_t* qkv, const int32_t* lengths, int B, int T, int heads, uint16_t* ctx, void* stamps, int occ, void* stream) {
  const int H = heads * se::kHD;
  dim3 grid((T + se::kAQ - 1) / se::kAQ, heads, B);
  float* sp = reinterpret_cast<float*>(stamps);
  if (occ == 1) hipLaunchKernelGGL((se::mhsa_fwd_kernel<1, 0, 1, 0>), grid, dim3(256), 0, se::as_stream(stream), qkv, lengths, T, H, ctx, sp, 0u, 0u, 1.f);
  else if (occ == 2) hipLaunchKernelGGL((se::mhsa_fwd_kernel<2, 0, 1, 0>), grid, dim3(256), 0, se::as_stream(stream), qkv, lengths, T, H, ctx, sp, 0u, 0u, 1.f);
  else hipLaunchKernelGGL((se::mhsa_fwd_kernel<3, 0, 1, 0>), grid, dim3(256), 0, se::as_stream(stream), qkv, lengths, T, H, ctx, sp, 0u, 0u, 1.f);
  SE_LAUNCH_CHECK();
  return SE_OK;
}
#endif

extern "C" int se_mhsa_fwd_prescaled_bf16(const uint16_t* qkv, const int32_t* lengths, int B, int T, int heads, uint16_t* ctx, void* stream) {
  static int pipe = -2, min_wgs = 768;
  if (pipe == -2) {
    const char* e = getenv("SE_AMD_MHSA_PIPE");
    pipe = e ? atoi(e) : -1;     // A/B: force one variant (0 = this file's kernel, 10 = mhsa8.hip's; developer builds: the parked experiments too)
    if (const char* m = getenv("SE_AMD_MHSA8_MIN_WGS")) min_wgs = atoi(m);
  }
  if (pipe >= 0) return mhsa_prescaled_launch(qkv, lengths, B, T, heads, ctx, pipe, stream);
  // default (round 4): 8-wave workgroups on one LDS-DMA staged K / V tile, two per CU, half of each SIMD's waves half a tile behind (mhsa8.hip:
  // mhsaN_fwd_kernel<8, 4, 1>; 107 vs 118 us at B = 32) from 768 such workgroups (1.5 per CU slot pair) on -- B >= 16 at T = 1001; below that this
  // file's 4-wave workgroups (three per CU, twice as many of them) cover the CUs better (profiles/r04_mhsa_bsweep.txt)
  const long wgs8 = (long)((T + 255) / 256) * heads * B;
  return mhsa_prescaled_launch(qkv, lengths, B, T, heads, ctx, wgs8 >= min_wgs ? 10 : 0, stream);
}

// test / measurement surface: variant 0 = this file's kernel (the default below 768 workgroups), 10 = mhsa8.hip's (the default from there on)
extern "C" int se_mhsa_fwd_prescaled_variant_bf16(const uint16_t* qkv, const int32_t* lengths, int B, int T, int heads, uint16_t* ctx, int variant,
                                                  void* stream) {
  SE_REQUIRE(variant >= 0 && variant <= 16, "se_mhsa_fwd_prescaled_variant_bf16: variant %d (0 or 10)", variant);
  return mhsa_prescaled_launch(qkv, lengths, B, T, heads, ctx, variant, stream);
}

extern "C" int se_mhsa_fwd_lse_bf16(const uint16_t* qkv, const int32_t* lengths, int B, int T, int heads, uint16_t* ctx, float* lse,
                                    float dropout_p, uint64_t seed, uint32_t site, void* stream) {
  SE_REQUIRE(lse, "se_mhsa_fwd_lse_bf16: null lse");
  return mhsa_fwd_launch(qkv, lengths, B, T, heads, ctx, lse, dropout_p, seed, site, stream);
}

#ifdef SE_AMD_EXPERIMENTS
// Training forward with the dropout mask of dropmask.hip (se_mhsa_dropmask: query-major bit matrix `mask_r`) instead of the in-kernel hash: same
// mask, same result up to the place of the 1 / (1 - p) factor (applied to the fp32 context sums instead of to each probability before its bf16
// rounding).
extern "C" int se_mhsa_fwd_lse_masked_bf16(const uint16_t* qkv, const int32_t* lengths, int B, int T, int heads, uint16_t* ctx, float* lse,
                                           const uint32_t* mask_r, float dropout_p, void* stream) {
  SE_REQUIRE(qkv && ctx && lse && mask_r, "se_mhsa_fwd_lse_masked_bf16: null argument");
  SE_REQUIRE(B > 0 && B <= 65535 && T > 0 && heads > 0 && heads <= 65535, "se_mhsa_fwd_lse_masked_bf16: bad shape B=%d T=%d heads=%d", B, T, heads);
  SE_REQUIRE((uintptr_t)mask_r % 16 == 0, "se_mhsa_fwd_lse_masked_bf16: mask must be 16-B aligned");
  const se::DropoutCfg d = se::make_dropout(dropout_p, 0);
  SE_REQUIRE(d.thr16 != 0, "se_mhsa_fwd_lse_masked_bf16: dropout_p must be > 0 (use se_mhsa_fwd_lse_bf16 without dropout)");
  const int H = heads * se::kHD;
  dim3 grid((T + se::kAQ - 1) / se::kAQ, heads, B);
  se::ProfScope prof(se::kProfMhsa, 4.0 * B * (double)heads * T * (double)T * se::kHD, se::as_stream(stream));
  hipLaunchKernelGGL((se::mhsa_fwd_kernel<2, 2, 0>), grid, dim3(256), 0, se::as_stream(stream), qkv, lengths, T, H, ctx, lse,
                     (uint32_t)((uintptr_t)mask_r & 0xffffffffu), (uint32_t)((uintptr_t)mask_r >> 32), d.scale);
  SE_LAUNCH_CHECK();
  return SE_OK;
}
#endif
