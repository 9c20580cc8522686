// mhsa8.hip -- row B2 core, inference form (pre-scaled queries): the flash attention forward as an 8-wave workgroup whose SIMD partners
// ALTERNATE a matrix segment with a load / softmax segment (MI355X_MICROARCH.md, "Two waves per SIMD"), built in round 4 after the in-kernel
// stamps of mhsa.hip (tools/mhsa_stamps.py, profiles/r04_mhsa_stamps.txt) showed where that kernel's waves wait: a wave ALONE on its SIMD needs
// ~2 200 cycles per 64-key tile for 512 cycles of MFMA and ~430 of softmax issue -- the K fragment reads in front of QK^T (~300 exposed), the
// matrix results in front of the first exponential (~250), the V^T fragment reads inside PV (~160), the global -> register -> LDS staging and its
// barrier (~200) -- and three such waves per SIMD only overlap each other by chance (1 330 cycles per tile and wave at the SIMD).
//
// Workgroup = 8 waves = 256 query rows of one (utterance, head); wave w owns 32 rows, waves w and w + 4 share a SIMD.  Per 64-key tile a wave runs
//   C1  S^T = K Q^T            8 MFMAs on registers only (the K fragments were read in the previous L2)
//   L1  V^T fragment reads (16 ds_read_b64_tr_b16) + the tile's softmax (speculative exp2 / exact online form, as mhsa.hip) + the LDS-DMA of tile t + 2
//   C2  O^T += V^T P^T         8 MFMAs on registers only
//   L2  K fragment reads of tile t + 1 (8 ds_read_b128)
// with one s_barrier after every segment; waves 4-7 run ONE segment behind waves 0-3, so on every SIMD one wave multiplies while its partner
// loads / exponentiates:   slot:      0      1      2      3
//                          waves 0-3  C1     L1     C2     L2
//                          waves 4-7  L2'    C1     L1     C2        (' = of the previous tile)
// K / V tiles arrive by LDS-DMA (inline asm: through the builtin the compiler orders every later ds_read behind the DMA with vmcnt(0)) into a
// three-slot ring, tile t + 2 issued in L1(t); each wave waits for its own pieces of tile t + 1 with a counted vmcnt at the end of L1(t), and the
// barrier behind it publishes them before any wave's L2(t).  Tile layout, swizzle and fragment addressing are mhsa.hip's (mhsa_tile.h: kv_off).
// 256 queries per staged tile halve the staging per query of mhsa.hip; there is no ds_write and no VGPR staging at all.
#include "clkprobe.h"
#include <stdlib.h>
#include <algorithm>
#include "common.h"
#include "prof.h"
#include "bf16.h"
#include "mhsa_tile.h"

// developer ablation of mhsaN_fwd_kernel (timing only, results are wrong): -DSE_MHSAN_ABL=<mask>: 1 no LDS-DMA staging (and no waits for it),
// 2 no tile barrier, 4 no LDS fragment reads (the MFMAs take the query fragments), 8 no exponentials
#ifndef SE_MHSAN_ABL
#define SE_MHSAN_ABL 0
#endif
// A/B: 1 = speculative exponentials in place + packed row-sum adds (16 v_pk_add_f32 instead of 32 v_add_f32 per tile, 118 instead of 128
// registers), 0 = separate registers + scalar adds (what ships).  Measured on one box, interleaved (profiles/r04_mhsa_pkadd_ab.txt): the packed form
// is 9-12 % SLOWER (8 waves 119.8-123.7 vs 110.1-111.3 us, 16 waves 131-134 vs 122-124): on this chip a packed fp32 add is not a cheaper issue
// than two scalar ones next to a busy matrix pipe, and the in-place chain exp -> add -> pack shortens the distance between dependent instructions
#ifndef SE_MHSAN_PKADD
#define SE_MHSAN_PKADD 0
#endif

SE_CLKPROBE_DECL(clkprobe_mhsa)
namespace se {
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int k8Q = 256;        // query rows per workgroup
constexpr int k8Slot = 16384;   // one ring slot: K tile (8 KiB) + V tile (8 KiB)

#define SE8_BAR()                                   \
  do {                                              \
    __builtin_amdgcn_sched_barrier(0);              \
    asm volatile("s_barrier" ::: "memory");         \
    __builtin_amdgcn_sched_barrier(0);              \
  } while (0)

// SPLIT = 0: the four-segment schedule above.  SPLIT = 1: the softmax is cut in two (keys 0-31 / 32-63 of the tile) and the halves ride L1 and L2
// (see the loop): both load segments then carry ~half of the vector work instead of L1 carrying all of it.
template <int SPLIT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void mhsa8_fwd_kernel(
    const uint16_t* __restrict__ qkv, const int32_t* __restrict__ lengths, int T, int H, uint16_t* __restrict__ ctx, float dscale
#ifdef SE_AMD_STAMPS
    , unsigned long long* __restrict__ stamps
#endif
    ) {
  __shared__ __attribute__((aligned(16))) char smem[3 * k8Slot];
#ifdef SE_AMD_STAMPS
  // developer build only: per-wave cycle sums of the four segments (0 C1, 2 L1, 4 C2, 6 L2) and of the barrier wait behind each (1, 3, 5, 7)
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_prev = 0;
#define SE_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc[i] += t_ - st_prev; st_prev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SE_STAMP(i) do { } while (0)
#endif

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  // XCD-aware work mapping (as mhsa.hip): all query tiles of one (utterance, head) on ONE XCD, consecutive in its dispatch order
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
  const int q0 = qt * k8Q + wave * 32;
  const int ld = 3 * H;
  const int len = lengths ? min(max(lengths[b], 1), T) : T;
  const int nkt = (len + kAK - 1) / kAK;
  const uint16_t* base = qkv + (size_t)b * T * ld + head * kHD;

  // ---- LDS-DMA: wave w brings rows [8 w, 8 w + 8) of the K and of the V tile, one 1-KiB piece (8 rows x 128 B) each; lane l writes slot l & 7 of
  //      row l >> 3 of the piece, so it FETCHES chunk (l & 7) ^ f(row)   (kv_off: slot = chunk ^ f)
  typedef __attribute__((address_space(3))) char* lds_c_t;
  const int drow = wave * 8 + (lane >> 3);
  const uint32_t dch = (uint32_t)(((lane & 7) ^ (kv_off(drow, 0) >> 4 & 7)) * 16);
  const uint32_t lds_wave = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(lds_c_t)smem + wave * 1024);
  const char* gbase = reinterpret_cast<const char*>(base);
#define SE8_DMA(kt, slot)                                                                                                   \
  do {                                                                                                                      \
    const uint32_t row_ = (uint32_t)(min((kt) * kAK + drow, T - 1) * ld);                                                   \
    const uint32_t ok_ = (row_ + (uint32_t)H) * 2u + dch, ov_ = (row_ + 2u * (uint32_t)H) * 2u + dch;                       \
    uint32_t keep_;                                                                                                         \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"    \
                 : "=&s"(keep_) : "v"(ok_), "s"(gbase), "s"(lds_wave + (uint32_t)((slot) * k8Slot)) : "memory");            \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"    \
                 : "=&s"(keep_) : "v"(ov_), "s"(gbase), "s"(lds_wave + (uint32_t)((slot) * k8Slot + 8192)) : "memory");     \
  } while (0)

  SE8_DMA(0, 0);
  if (nkt > 1) SE8_DMA(1, 1);

  // ---- Q fragments (B operand of S^T = K Q^T): lane -> query row q0 + l31, d = 16 s + 8 hh .. +7
  bf16x8 qf[4];
  {
    const int q = min(q0 + l31, T - 1);
    const uint16_t* qp = base + (size_t)q * ld + 8 * hh;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
  }

  const f32x16 kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 o0, o1;                      // O^T d-blocks 0 / 1: col = query (lane & 31), row = d
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = 0.f, l_run = 0.f;
  bool slow = dscale < 0.f;           // wave-uniform: the speculative (no row maximum) path failed once; dscale < 0: never speculate (A/B switch)
  constexpr float kDefer = 8.f;

  // ---- loop-invariant LDS byte offsets (mhsa.hip)
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

  bf16x8 kf[8];                       // K fragments of the coming tile: [2 s + key block]
  bf16x8 vf[8];                       // V^T fragments of the current tile: [(2 kb + s) * 2 + dblk]
#define SE8_READ_K(slot)                                                                                    \
  do {                                                                                                      \
    const char* t_ = smem + (slot) * k8Slot;                                                                \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                         \
      kf[2 * s] = *reinterpret_cast<const bf16x8*>(t_ + koff[s]);                                           \
      kf[2 * s + 1] = *reinterpret_cast<const bf16x8*>(t_ + koff[s] + 4096);                                \
    }                                                                                                       \
  } while (0)
#define SE8_PIN8(a)                                                                                         \
  asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]))

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // tiles 0 / 1 and the Q rows
  SE8_BAR();
  SE8_READ_K(0);
  SE8_PIN8(kf);
  if (wave >= 4) SE8_BAR();           // waves 4-7 run one segment behind
#ifdef SE_AMD_STAMPS
  st_prev = __builtin_amdgcn_s_memtime();
  const unsigned long long st_t0 = st_prev;
#endif

  int slot = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    const char* t_s = smem + slot * k8Slot;
    const int slot1 = slot == 2 ? 0 : slot + 1, slot2 = slot == 0 ? 2 : slot - 1;      // (slot + 1) % 3, (slot + 2) % 3
    // ================= C1: S^T = K Q^T
    f32x16 s0, s1;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2 * s], qf[s], s == 0 ? kZero16 : s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2 * s + 1], qf[s], s == 0 ? kZero16 : s1, 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    SE_STAMP(0);
    SE8_BAR();
    SE_STAMP(1);
    // ================= L1: staging of tile kt + 2, V^T fragments, softmax
    if (kt + 2 < nkt) SE8_DMA(kt + 2, slot2);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int dblk = 0; dblk < 2; ++dblk) {
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
              (__attribute__((address_space(3))) bf16x4*)(t_s + voff[dblk][0] + kb * 4096 + s * 2048));
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
              (__attribute__((address_space(3))) bf16x4*)(t_s + voff[dblk][1] + kb * 4096 + s * 2048));
          vf[(2 * kb + s) * 2 + dblk] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
    if ((kt + 1) * kAK > len) {
      const int kbase = kt * kAK + 4 * hh;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kbase + (r & 3) + 8 * (r >> 2);
        if (key >= len) s0[r] = -INFINITY;
        if (key + 32 >= len) s1[r] = -INFINITY;
      }
    }
    bf16x8 pf[2][2];
    bool spec_ok = false;
    if (!slow) {
      // SPECULATIVE tile (mhsa.hip): probabilities against the initial reference 0, no row maximum; the row sums tell whether it held
      float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float a0 = __builtin_amdgcn_exp2f(s0[8 * s + j]);
          const float a1 = __builtin_amdgcn_exp2f(s1[8 * s + j]);
          rs0 += a0;
          rs1 += a1;
          pf[0][s][j] = (__bf16)a0;
          pf[1][s][j] = (__bf16)a1;
        }
      const float rs = rs0 + rs1;
      const bool bad = !(rs < 0x1p60f) || (kt == 0 && rs < 0x1p-60f);
      if (!__any(bad)) {
        l_run += rs;
        spec_ok = true;
      } else {
        slow = true;
      }
    }
    if (!spec_ok) {
      float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s0[r]), s1[r]);       /* v_max3_f32 */
      {
        const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = fmaxf(__uint_as_float(sw_[0]), __uint_as_float(sw_[1]));
      }
      float m_new = ((mx - m_run) > kDefer) ? mx : m_run;
      if (kt == 0 && mx < -64.f) m_new = mx;        /* a first tile far below the initial reference 0 (later tiles cannot matter) */
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      const float mc = -m_new;
      float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float a0 = __builtin_amdgcn_exp2f(s0[r] + mc);
        const float a1 = __builtin_amdgcn_exp2f(s1[r] + mc);
        rs0 += a0;
        rs1 += a1;
        s0[r] = a0; s1[r] = a1;
      }
      l_run = fmaf(l_run, alpha, rs0 + rs1);
      m_run = m_new;
      if (__any(alpha != 1.0f)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          pf[0][s][j] = (__bf16)s0[8 * s + j];
          pf[1][s][j] = (__bf16)s1[8 * s + j];
        }
    }
    SE8_PIN8(vf);
    asm volatile("" : "+v"(pf[0][0]), "+v"(pf[0][1]), "+v"(pf[1][0]), "+v"(pf[1][1]));
    // this wave's pieces of tile kt + 1 (issued one tile ago) are in LDS once all but the two pieces just issued have completed
    if (kt + 2 < nkt) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SE_STAMP(2);
    SE8_BAR();
    SE_STAMP(3);
    // ================= C2: O^T += V^T P^T
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[(2 * kb + s) * 2], pf[kb][s], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[(2 * kb + s) * 2 + 1], pf[kb][s], o1, 0, 0, 0);
      }
    __builtin_amdgcn_s_setprio(0);
    SE_STAMP(4);
    SE8_BAR();
    SE_STAMP(5);
    // ================= L2: K fragments of the next tile
    if (kt + 1 < nkt) {
      SE8_READ_K(slot1);
      SE8_PIN8(kf);
    }
    SE_STAMP(6);
    SE8_BAR();
    SE_STAMP(7);
    slot = slot1;
  }
  if (wave < 4) SE8_BAR();            // waves 0-3 match the extra barrier of waves 4-7

  // ---- epilogue: O / l ; lane holds query q0 + l31, d = 32 dblk + (r&3) + 8 (r>>2) + 4 hh
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  const int q = q0 + l31;
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
  if (stamps) {
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    unsigned long long* sp = stamps + ((size_t)lin * 8 + wave) * 10;
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) sp[i] = st_acc[i];
      sp[8] = __builtin_amdgcn_s_memtime() - st_t0;
      sp[9] = (unsigned long long)nkt;
    }
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// mhsaN_fwd_kernel<NW>: NW = 8 or 16 waves (256 / 512 query rows) SHARE one LDS-DMA staged K / V tile and otherwise run FREE (one barrier per
// tile, no segment hand-shake): what the stamps of the alternating kernel above said about head dim 64 -- the softmax of a 64-key tile is ~520
// issue cycles of ONE wave (a single wave issues a vector instruction every ~5 cycles, a SIMD with 3-4 waves one every ~2), against 512 matrix
// cycles, so a strict two-wave alternation leaves the matrix pipe idle for half of every load segment -- and what the round-3 ablation said
// about mhsa.hip (staging = 27 % of the launch at 128 queries per staged tile).  With 512 queries per tile a wave issues ONE 1-KiB LDS-DMA piece
// per tile (NW = 16; two at NW = 8) and nothing else for the staging: no global loads into registers, no ds_write, 1/4 of the L2 -> LDS bytes.
// Three-slot ring, tile t + 2 issued at the top of tile t, own piece(s) of tile t + 1 awaited (counted vmcnt) in front of the tile's one barrier.
// WPE = waves per SIMD the register allocation is held to: <8, 2> one 8-wave workgroup per CU, <8, 4> TWO (128 registers; two independent barrier
// domains), <16, 4> one 16-wave workgroup
// STAG = 1: the waves with (wave & 4) != 0 -- one half of every SIMD's residents -- take the tile's barrier in the MIDDLE of their tile (between the
// softmax and the PV products) instead of at its end: a per-tile barrier otherwise re-aligns all waves of a SIMD at every tile top (they then queue
// for the matrix pipe together and exponentiate together); with the shifted barrier the two halves run half a tile apart (MI355X_MICROARCH.md, "Two
// waves per SIMD", item 9).  One more ring slot (4): a late wave still reads V of tile t - 1 after the barrier that lets the early ones start tile t.
template <int NW, int WPE, int STAG = 0>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void mhsaN_fwd_kernel(
    const uint16_t* __restrict__ qkv, const int32_t* __restrict__ lengths, int T, int H, uint16_t* __restrict__ ctx, float dscale) {
  constexpr int kRing = STAG ? 4 : 3;
  __shared__ __attribute__((aligned(16))) char smem[kRing * k8Slot];
  SE_CLKPROBE_BEGIN();
  static_assert(NW == 8 || NW == 16, "8 or 16 waves");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
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
  const int q0 = qt * (NW * 32) + wave * 32;
  const int ld = 3 * H;
  const int len = lengths ? min(max(lengths[b], 1), T) : T;
  const int nkt = (len + kAK - 1) / kAK;
  const uint16_t* base = qkv + (size_t)b * T * ld + head * kHD;

  // LDS-DMA pieces (8 rows x 128 B = 1 KiB): NW = 16: wave w < 8 brings K piece w, wave w >= 8 V piece w - 8; NW = 8: K piece w and V piece w
  typedef __attribute__((address_space(3))) char* lds_c_t;
  const int piece = wave & 7;
  const int drow = piece * 8 + (lane >> 3);
  const uint32_t dch = (uint32_t)(((lane & 7) ^ (kv_off(drow, 0) >> 4 & 7)) * 16);
  const uint32_t dpk = (uint32_t)drow | (dch << 8);          // one register for both lane constants of the DMA address (unpacked per issue, opaquely:
                                                              // hoisted copies spilled, and a spill reload waits vmcnt(0) in front of the DMA)
  const uint32_t which0 = (NW == 16 && wave >= 8) ? 2u : 1u;                       // 1 = K block of the fused row, 2 = V block
  const uint32_t lds_piece = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(lds_c_t)smem + piece * 1024 + (which0 == 2u ? 8192u : 0u));
  const char* gbase = reinterpret_cast<const char*>(base);
#define SEN_DMA(kt, slot)                                                                                                     \
  do {                                                                                                                        \
    uint32_t dp_ = dpk;                                                                                                       \
    asm volatile("" : "+v"(dp_));                                                                                             \
    const uint32_t row_ = (uint32_t)(min((kt) * kAK + (int)(dp_ & 0xffu), T - 1) * ld);                                       \
    const uint32_t o0_ = (row_ + which0 * (uint32_t)H) * 2u + (dp_ >> 8);                                                     \
    uint32_t keep_;                                                                                                           \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"      \
                 : "=&s"(keep_) : "v"(o0_), "s"(gbase), "s"(lds_piece + (uint32_t)((slot) * k8Slot)) : "memory");             \
    if (NW == 8) {                                                                                                            \
      const uint32_t o1_ = (row_ + 2u * (uint32_t)H) * 2u + (dp_ >> 8);                                                       \
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"    \
                   : "=&s"(keep_) : "v"(o1_), "s"(gbase), "s"(lds_piece + (uint32_t)((slot) * k8Slot + 8192)) : "memory");    \
    }                                                                                                                         \
  } while (0)
  constexpr int kPer = NW == 8 ? 2 : 1;       // DMA instructions per wave and tile

  if (!(SE_MHSAN_ABL & 1)) {
  SEN_DMA(0, 0);
  if (nkt > 1) SEN_DMA(1, 1);
  }

  bf16x8 qf[4];
  {
    const int q = min(q0 + l31, T - 1);
    const uint16_t* qp = base + (size_t)q * ld + 8 * hh;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
  }
  const f32x16 kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 o0, o1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = 0.f, l_run = 0.f;
  bool slow = dscale < 0.f;
  constexpr float kDefer = 8.f;
  // K fragment address of k-step s = kv_off(l31, 2 s + hh) = koff0 ^ (s << 5): 2 s and hh occupy disjoint bits of the chunk index, so the XOR with
  // the swizzle term commutes -- ONE address register instead of four (the 16-wave form has 128 registers per lane)
  const int koff0 = kv_off(l31, hh);
  const int tq = (lane & 15) >> 2, tp = lane & 3, g1 = (lane >> 4) & 1;
  int voff[2][2];
#pragma unroll
  for (int dblk = 0; dblk < 2; ++dblk) {
    const int dcol = dblk * 32 + 16 * g1 + 4 * tp;
    voff[dblk][0] = 8192 + kv_off(4 * hh + tq, dcol >> 3) + (dcol & 7) * 2;
    voff[dblk][1] = 8192 + kv_off(4 * hh + tq + 8, dcol >> 3) + (dcol & 7) * 2;
  }

  // tile 0 landed (this wave's piece(s); tile 1's may still fly) -> publish
  if (nkt > 1) { if (kPer == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SE8_BAR();

  const bool late = STAG && (wave & 4) != 0;
#define SEN_SYNC()                                                                                                            \
  do {                                                                                                                        \
    if (!(SE_MHSAN_ABL & 1)) {                                                                                                \
      if (kt + 2 < nkt) { if (kPer == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); } \
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                   \
    }                                                                                                                         \
    if (!(SE_MHSAN_ABL & 2)) SE8_BAR();                                                                                       \
  } while (0)
  int slot = 0;
  for (int kt = 0; kt < nkt; ++kt) {
    const char* t_s = smem + slot * k8Slot;
    const int slot1 = slot == kRing - 1 ? 0 : slot + 1, slot2 = slot1 == kRing - 1 ? 0 : slot1 + 1;      // (slot + 1), (slot + 2) mod ring
    if (kt + 2 < nkt && !(SE_MHSAN_ABL & 1)) SEN_DMA(kt + 2, slot2);       // the slot of tile kt - 1: every wave left it before the barrier that ended tile kt - 1
    f32x16 s0, s1;
    // S^T = K Q^T of this tile + the key mask of a ragged last tile.  A macro: the exact path re-issues it when the speculative exponentials (taken in
    // place, below) failed their range test -- rare, and the K tile is still in its slot
#define SEN_QK()                                                                                                              \
    do {                                                                                                                      \
      _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                                         \
        const bf16x8 ka = (SE_MHSAN_ABL & 4) ? qf[s] : *reinterpret_cast<const bf16x8*>(t_s + (koff0 ^ (s << 5)));            \
        const bf16x8 kb_ = (SE_MHSAN_ABL & 4) ? qf[3 - s] : *reinterpret_cast<const bf16x8*>(t_s + (koff0 ^ (s << 5)) + 4096); \
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[s], s == 0 ? kZero16 : s0, 0, 0, 0);                              \
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb_, qf[s], s == 0 ? kZero16 : s1, 0, 0, 0);                             \
      }                                                                                                                       \
      if ((kt + 1) * kAK > len) {                                                                                             \
        const int kbase = kt * kAK + 4 * hh;                                                                                  \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                                      \
          const int key = kbase + (r & 3) + 8 * (r >> 2);                                                                     \
          if (key >= len) s0[r] = -INFINITY;                                                                                  \
          if (key + 32 >= len) s1[r] = -INFINITY;                                                                             \
        }                                                                                                                     \
      }                                                                                                                       \
    } while (0)
    SEN_QK();
    // barrier positions of the staggered forms: STAG 1: early waves at the tile end, late ones after the softmax; 2: end / after QK^T;
    // 3: after QK^T / after the softmax
    if ((STAG == 2 && late) || (STAG == 3 && !late)) SEN_SYNC();
    bf16x8 pf[2][2];
    bool spec_ok = false, redo = false;
    if (!slow) {
#if SE_MHSAN_PKADD
      // exponentials IN PLACE in the score registers; neighbouring registers of one accumulator block are an aligned pair, so the row sum
      // takes ONE v_pk_add_f32 per two exponentials (16 instead of 32 adds per tile)
      f32x2 rs2 = {0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          s0[2 * k + e] = (SE_MHSAN_ABL & 8) ? s0[2 * k + e] * 0.5f + 1.0f : __builtin_amdgcn_exp2f(s0[2 * k + e]);
          s1[2 * k + e] = (SE_MHSAN_ABL & 8) ? s1[2 * k + e] * 0.5f + 1.0f : __builtin_amdgcn_exp2f(s1[2 * k + e]);
        }
        rs2 += f32x2{s0[2 * k], s0[2 * k + 1]};
        rs2 += f32x2{s1[2 * k], s1[2 * k + 1]};
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          pf[0][s][j] = (__bf16)s0[8 * s + j];
          pf[1][s][j] = (__bf16)s1[8 * s + j];
        }
      const float rs = rs2[0] + rs2[1];
#else
      float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float a0 = (SE_MHSAN_ABL & 8) ? s0[8 * s + j] * 0.5f + 1.0f : __builtin_amdgcn_exp2f(s0[8 * s + j]);
          const float a1 = (SE_MHSAN_ABL & 8) ? s1[8 * s + j] * 0.5f + 1.0f : __builtin_amdgcn_exp2f(s1[8 * s + j]);
          rs0 += a0;
          rs1 += a1;
          pf[0][s][j] = (__bf16)a0;
          pf[1][s][j] = (__bf16)a1;
        }
      const float rs = rs0 + rs1;
#endif
      const bool bad = !(rs < 0x1p60f) || (kt == 0 && rs < 0x1p-60f);
      if (!__any(bad)) {
        l_run += rs;
        spec_ok = true;
      } else {
        slow = true;
        redo = true;
      }
    }
    if (!spec_ok) {
      if (redo && SE_MHSAN_PKADD) SEN_QK();                  // the scores again: the failed speculation exponentiated them in place
      float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s0[r]), s1[r]);
      {
        const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
        mx = fmaxf(__uint_as_float(sw_[0]), __uint_as_float(sw_[1]));
      }
      float m_new = ((mx - m_run) > kDefer) ? mx : m_run;
      if (kt == 0 && mx < -64.f) m_new = mx;
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      const float mc = -m_new;
      float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float a0 = __builtin_amdgcn_exp2f(s0[r] + mc);
        const float a1 = __builtin_amdgcn_exp2f(s1[r] + mc);
        rs0 += a0;
        rs1 += a1;
        s0[r] = a0; s1[r] = a1;
      }
      l_run = fmaf(l_run, alpha, rs0 + rs1);
      m_run = m_new;
      if (__any(alpha != 1.0f)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          pf[0][s][j] = (__bf16)s0[8 * s + j];
          pf[1][s][j] = (__bf16)s1[8 * s + j];
        }
    }
    if ((STAG == 1 || STAG == 3) && late) SEN_SYNC();
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int dblk = 0; dblk < 2; ++dblk) {
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
              (__attribute__((address_space(3))) bf16x4*)(t_s + voff[dblk][0] + kb * 4096 + s * 2048));
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
              (__attribute__((address_space(3))) bf16x4*)(t_s + voff[dblk][1] + kb * 4096 + s * 2048));
          bf16x8 va = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          if (SE_MHSAN_ABL & 4) va = qf[2 * kb + s];
          if (dblk == 0) o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pf[kb][s], o0, 0, 0, 0);
          else o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pf[kb][s], o1, 0, 0, 0);
        }
    // own piece(s) of tile kt + 1 (issued one tile ago) in LDS: all but the kPer just issued have completed
    if (STAG == 0 || ((STAG == 1 || STAG == 2) && !late)) SEN_SYNC();
    slot = slot1;
  }

#undef SEN_QK
  SE_CLKPROBE_END(clkprobe_mhsa);
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  const int q = q0 + l31;
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
}

#ifdef SE_AMD_EXPERIMENTS
// ---------------------------------------------------------------------------------------------------------------------------------------------
// mhsaP_fwd_kernel: mhsaN_fwd_kernel<8, 4, 1> as a PERSISTENT workgroup.  The in-kernel clock probe (csrc/clkprobe.h, profiles/r04_clk_probe.txt)
// put numbers on the launch: a 256-query workgroup lives 56 950 cycles = 30.2 us at the 1.885 GHz the chip holds in this kernel, the 1 536
// workgroups of the bench shape are exactly three rounds of 2 per CU = 90.6 us, and the launch takes 108 us from the first workgroup's start to
// the last one's end -- 16 % of the launch is workgroup turnover (dispatch, an empty ring, the first tiles' latency, the drain) between rounds.
// Here 2 workgroups per CU are launched ONCE and each walks its list of (utterance, head, query tile) items; the K / V tiles of all its items
// form ONE stream through the four-slot ring (the LDS-DMA look-ahead of two tiles crosses the item boundary, so the ring never drains), the next
// item's Q fragments are fetched (inline asm, so their place among the counted LDS-DMA operations is fixed) right after the last QK^T of the
// current item, and nothing but the epilogue's stores separates two items.  Same arithmetic, same tile order per item: bit-identical results.
// Item order: XCD x (= blockIdx & 7) owns the pairs = x (mod 8); its workgroups take list positions slot, slot + n, slot + 2 n ... of
// (pair-major, query tile minor), as the dispatcher would have handed them out.
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void mhsaP_fwd_kernel(
    const uint16_t* __restrict__ qkv, const int32_t* __restrict__ lengths, int T, int H, uint16_t* __restrict__ ctx, float dscale, int nqt, int heads,
    int pairs) {
  constexpr int kRing = 4;
  __shared__ __attribute__((aligned(16))) char smem[kRing * k8Slot];
  SE_CLKPROBE_BEGIN();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hh = lane >> 5;
  const int ld = 3 * H;

  // ---- this workgroup's item list: positions j0 + i * jstride, i < n_items
  int j0, jstride, n_items, xcd = -1;
  {
    const int w = blockIdx.x, G = gridDim.x;
    int per;
    if ((pairs & 7) == 0 && (G & 7) == 0) { xcd = w & 7; j0 = w >> 3; jstride = G >> 3; per = nqt * (pairs >> 3); }
    else { j0 = w; jstride = G; per = nqt * pairs; }
    n_items = j0 < per ? (per - j0 + jstride - 1) / jstride : 0;
  }
  if (n_items == 0) return;
#define SEP_DECODE(i, b_, head_, qt_)                                      \
  do {                                                                     \
    const int j_ = j0 + (i) * jstride;                                     \
    const int pr_ = xcd >= 0 ? 8 * (j_ / nqt) + xcd : j_ / nqt;            \
    qt_ = j_ % nqt; head_ = pr_ % heads; b_ = pr_ / heads;                 \
  } while (0)

  // ---- the K / V tile stream (all items back to back): cursor (d_i, d_kt), its item's base address and tile count
  typedef __attribute__((address_space(3))) char* lds_c_t;
  const int drow = wave * 8 + (lane >> 3);
  const uint32_t dch = (uint32_t)(((lane & 7) ^ (kv_off(drow, 0) >> 4 & 7)) * 16);
  const uint32_t dpk = (uint32_t)drow | (dch << 8);
  const uint32_t lds_piece = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(lds_c_t)smem + wave * 1024);
  int d_i = 0, d_kt = 0, d_nkt = 0;
  const char* d_base = nullptr;
#define SEP_DSETUP()                                                                                  \
  do {                                                                                                \
    int b_, head_, qt_;                                                                               \
    SEP_DECODE(d_i, b_, head_, qt_);                                                                  \
    (void)qt_;                                                                                        \
    const int len_ = lengths ? min(max(lengths[b_], 1), T) : T;                                       \
    d_nkt = (len_ + kAK - 1) / kAK;                                                                   \
    d_base = reinterpret_cast<const char*>(qkv + (size_t)b_ * T * ld + head_ * kHD);                  \
  } while (0)
  SEP_DSETUP();
  bool iss = false;
  // one tile of the stream (K piece + V piece of this wave) into ring slot `slot_`; iss = whether the stream still had a tile
#define SEP_DMA_NEXT(slot_)                                                                                                     \
  do {                                                                                                                          \
    iss = d_i < n_items;                                                                                                        \
    if (iss) {                                                                                                                  \
      uint32_t dp_ = dpk;                                                                                                       \
      asm volatile("" : "+v"(dp_));                                                                                             \
      const uint32_t row_ = (uint32_t)(min(d_kt * kAK + (int)(dp_ & 0xffu), T - 1) * ld);                                       \
      const uint32_t o0_ = (row_ + (uint32_t)H) * 2u + (dp_ >> 8), o1_ = (row_ + 2u * (uint32_t)H) * 2u + (dp_ >> 8);           \
      uint32_t keep_;                                                                                                           \
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"      \
                   : "=&s"(keep_) : "v"(o0_), "s"(d_base), "s"(lds_piece + (uint32_t)((slot_) * k8Slot)) : "memory");           \
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"      \
                   : "=&s"(keep_) : "v"(o1_), "s"(d_base), "s"(lds_piece + (uint32_t)((slot_) * k8Slot + 8192)) : "memory");    \
      if (++d_kt == d_nkt) {                                                                                                    \
        d_kt = 0;                                                                                                               \
        if (++d_i < n_items) SEP_DSETUP();                                                                                      \
      }                                                                                                                         \
    }                                                                                                                           \
  } while (0)

  // ---- item 0: Q fragments first (plain loads, older than every LDS-DMA), then the stream's first two tiles
  int b, head, qt;
  SEP_DECODE(0, b, head, qt);
  bf16x8 qf[4];
  {
    const int q = min(qt * k8Q + wave * 32 + l31, T - 1);
    const uint16_t* qp = qkv + ((size_t)b * T + q) * ld + head * kHD + 8 * hh;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
  }
  SEP_DMA_NEXT(0);
  SEP_DMA_NEXT(1);
  if (iss) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SE8_BAR();

  const f32x16 kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  constexpr float kDefer = 8.f;
  const int koff0 = kv_off(l31, hh);
  const int tq = (lane & 15) >> 2, tp = lane & 3, g1 = (lane >> 4) & 1;
  int voff[2][2];
#pragma unroll
  for (int dblk = 0; dblk < 2; ++dblk) {
    const int dcol = dblk * 32 + 16 * g1 + 4 * tp;
    voff[dblk][0] = 8192 + kv_off(4 * hh + tq, dcol >> 3) + (dcol & 7) * 2;
    voff[dblk][1] = 8192 + kv_off(4 * hh + tq + 8, dcol >> 3) + (dcol & 7) * 2;
  }
  const bool late = (wave & 4) != 0;
  // the tile's barrier: this wave's pieces of the NEXT stream tile (issued one tile ago) have landed when all but the operations issued since
  // -- this tile's LDS-DMA pair (iss) and the four Q loads of the next item (qpre) -- are complete
#define SEP_SYNC()                                                                                                            \
  do {                                                                                                                        \
    if (iss) { if (qpre) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); } \
    else { if (qpre) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } \
    SE8_BAR();                                                                                                                \
  } while (0)

  int slot = 0;
  for (int it = 0; it < n_items; ++it) {
    if (it > 0) SEP_DECODE(it, b, head, qt);
    const int q0 = qt * k8Q + wave * 32;
    const int len = lengths ? min(max(lengths[b], 1), T) : T;
    const int nkt = (len + kAK - 1) / kAK;
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float m_run = 0.f, l_run = 0.f;
    bool slow = dscale < 0.f;
    bool qpre = false;
    for (int kt = 0; kt < nkt; ++kt) {
      const char* t_s = smem + slot * k8Slot;
      const int slot1 = (slot + 1) & 3, slot2 = (slot + 2) & 3;
      SEP_DMA_NEXT(slot2);                 // the slot of stream tile t - 2: every wave left it before the barrier(s) that ended tile t - 1
      f32x16 s0, s1;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bf16x8 ka = *reinterpret_cast<const bf16x8*>(t_s + (koff0 ^ (s << 5)));
        const bf16x8 kb_ = *reinterpret_cast<const bf16x8*>(t_s + (koff0 ^ (s << 5)) + 4096);
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[s], s == 0 ? kZero16 : s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb_, qf[s], s == 0 ? kZero16 : s1, 0, 0, 0);
      }
      if (kt == nkt - 1 && it + 1 < n_items) {
        // the item's last QK^T is issued: its Q registers take the next item's fragments (waited for with vmcnt(0) behind this tile)
        int nb, nh, nq;
        SEP_DECODE(it + 1, nb, nh, nq);
        const char* sb = reinterpret_cast<const char*>(qkv + (size_t)nb * T * ld + nh * kHD);       // scalar base of the item; lane part below
        // the lane id re-derived HERE (mbcnt): a hoisted address pair, or `lane` itself, is a register the tile loop does not have -- it was spilled,
        // and its reload's vmcnt(0) waited for the LDS-DMA just issued
        int lq;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lq));
        const uint32_t vo = (uint32_t)(min(nq * k8Q + wave * 32 + (lq & 31), T - 1) * ld + 8 * (lq >> 5)) * 2u;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(qf[0]) : "v"(vo), "s"(sb) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:32" : "=v"(qf[1]) : "v"(vo), "s"(sb) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:64" : "=v"(qf[2]) : "v"(vo), "s"(sb) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:96" : "=v"(qf[3]) : "v"(vo), "s"(sb) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        qpre = true;
      }
      if ((kt + 1) * kAK > len) {
        const int kbase = kt * kAK + 4 * hh;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase + (r & 3) + 8 * (r >> 2);
          if (key >= len) s0[r] = -INFINITY;
          if (key + 32 >= len) s1[r] = -INFINITY;
        }
      }
      bf16x8 pf[2][2];
      bool spec_ok = false;
      if (!slow) {
        float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float a0 = __builtin_amdgcn_exp2f(s0[8 * s + j]);
            const float a1 = __builtin_amdgcn_exp2f(s1[8 * s + j]);
            rs0 += a0;
            rs1 += a1;
            pf[0][s][j] = (__bf16)a0;
            pf[1][s][j] = (__bf16)a1;
          }
        const float rs = rs0 + rs1;
        const bool bad = !(rs < 0x1p60f) || (kt == 0 && rs < 0x1p-60f);
        if (!__any(bad)) {
          l_run += rs;
          spec_ok = true;
        } else {
          slow = true;
        }
      }
      if (!spec_ok) {
        float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s0[r]), s1[r]);
        {
          const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
          mx = fmaxf(__uint_as_float(sw_[0]), __uint_as_float(sw_[1]));
        }
        float m_new = ((mx - m_run) > kDefer) ? mx : m_run;
        if (kt == 0 && mx < -64.f) m_new = mx;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        const float mc = -m_new;
        float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float a0 = __builtin_amdgcn_exp2f(s0[r] + mc);
          const float a1 = __builtin_amdgcn_exp2f(s1[r] + mc);
          rs0 += a0;
          rs1 += a1;
          s0[r] = a0; s1[r] = a1;
        }
        l_run = fmaf(l_run, alpha, rs0 + rs1);
        m_run = m_new;
        if (__any(alpha != 1.0f)) {
#pragma unroll
          for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            pf[0][s][j] = (__bf16)s0[8 * s + j];
            pf[1][s][j] = (__bf16)s1[8 * s + j];
          }
      }
      if (late) SEP_SYNC();
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int dblk = 0; dblk < 2; ++dblk) {
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (__attribute__((address_space(3))) bf16x4*)(t_s + voff[dblk][0] + kb * 4096 + s * 2048));
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                (__attribute__((address_space(3))) bf16x4*)(t_s + voff[dblk][1] + kb * 4096 + s * 2048));
            const bf16x8 va = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            if (dblk == 0) o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pf[kb][s], o0, 0, 0, 0);
            else o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pf[kb][s], o1, 0, 0, 0);
          }
      if (!late) SEP_SYNC();
      slot = slot1;
    }
    if (qpre) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the next item's Q fragments are in their registers

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    const int q = q0 + l31;
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
  }
  SE_CLKPROBE_END(clkprobe_mhsa);
#undef SEP_SYNC
#undef SEP_DMA_NEXT
#undef SEP_DSETUP
#undef SEP_DECODE
}
#endif      // SE_AMD_EXPERIMENTS

}  // namespace se

// variant 9: 8 waves free-running (one workgroup per CU), 10: the same held to 128 registers (two per CU), 16: 16 waves (mhsaN_fwd_kernel)
int se_mhsaN_fwd_launch(const uint16_t* qkv, const int32_t* lengths, int B, int T, int heads, uint16_t* ctx, int never_speculate, int nw, int wpe, hipStream_t st) {
  const int H = heads * se::kHD;
  SE_REQUIRE((double)T * 3.0 * H * 2.0 < 2147483648.0, "se_mhsaN: T * 3 H * 2 = %.0f bytes exceeds the 31-bit DMA offset", (double)T * 3.0 * H * 2.0);
  dim3 grid((T + nw * 32 - 1) / (nw * 32), heads, B);
#ifdef SE_AMD_EXPERIMENTS
  static const int stag = getenv("SE_AMD_MHSA_STAG") ? atoi(getenv("SE_AMD_MHSA_STAG")) : 1;      // A/B: 0 = every wave's barrier at the tile end; 1 .. 3: the staggered forms
  if (nw == 16 && stag) hipLaunchKernelGGL((se::mhsaN_fwd_kernel<16, 4, 1>), grid, dim3(1024), 0, st, qkv, lengths, T, H, ctx, never_speculate ? -1.f : 1.f);
  else if (nw == 8 && wpe == 4 && stag == 2) hipLaunchKernelGGL((se::mhsaN_fwd_kernel<8, 4, 2>), grid, dim3(512), 0, st, qkv, lengths, T, H, ctx, never_speculate ? -1.f : 1.f);
  else if (nw == 8 && wpe == 4 && stag == 3) hipLaunchKernelGGL((se::mhsaN_fwd_kernel<8, 4, 3>), grid, dim3(512), 0, st, qkv, lengths, T, H, ctx, never_speculate ? -1.f : 1.f);
  else if (nw == 8 && wpe == 4 && stag) hipLaunchKernelGGL((se::mhsaN_fwd_kernel<8, 4, 1>), grid, dim3(512), 0, st, qkv, lengths, T, H, ctx, never_speculate ? -1.f : 1.f);
  else if (nw == 16) hipLaunchKernelGGL((se::mhsaN_fwd_kernel<16, 4>), grid, dim3(1024), 0, st, qkv, lengths, T, H, ctx, never_speculate ? -1.f : 1.f);
  else if (wpe == 4) hipLaunchKernelGGL((se::mhsaN_fwd_kernel<8, 4>), grid, dim3(512), 0, st, qkv, lengths, T, H, ctx, never_speculate ? -1.f : 1.f);
  else hipLaunchKernelGGL((se::mhsaN_fwd_kernel<8, 2>), grid, dim3(512), 0, st, qkv, lengths, T, H, ctx, never_speculate ? -1.f : 1.f);
#else
  SE_REQUIRE(nw == 8 && wpe == 4, "se_mhsaN: the product library has the <8 waves, 4 per SIMD, staggered> form only");
  hipLaunchKernelGGL((se::mhsaN_fwd_kernel<8, 4, 1>), grid, dim3(512), 0, st, qkv, lengths, T, H, ctx, never_speculate ? -1.f : 1.f);
#endif
  SE_LAUNCH_CHECK();
  return SE_OK;
}

#ifdef SE_AMD_EXPERIMENTS
// variant 11: mhsaP_fwd_kernel -- two persistent workgroups per CU (SE_AMD_MHSAP_WGS overrides the grid for A/B); variants 13 / 14: 8 / 5 workgroups
int se_mhsaP_fwd_launch(const uint16_t* qkv, const int32_t* lengths, int B, int T, int heads, uint16_t* ctx, int never_speculate, int wgs, hipStream_t st) {
  const int H = heads * se::kHD;
  SE_REQUIRE((double)T * 3.0 * H * 2.0 < 2147483648.0, "se_mhsaP: T * 3 H * 2 = %.0f bytes exceeds the 31-bit DMA offset", (double)T * 3.0 * H * 2.0);
  static int want = 0;
  if (want == 0) {
    if (const char* e = getenv("SE_AMD_MHSAP_WGS")) want = std::max(1, atoi(e));
    else {
      int dev = 0;
      hipDeviceProp_t prop;
      SE_HIP(hipGetDevice(&dev));
      SE_HIP(hipGetDeviceProperties(&prop, dev));
      want = 2 * std::max(8, prop.multiProcessorCount & ~7);      // 64 KiB of LDS and 128 registers per lane: exactly two resident per CU
    }
  }
  const int nqt = (T + se::k8Q - 1) / se::k8Q;
  const long items = (long)nqt * heads * B;
  SE_REQUIRE(items < 2147483647L, "se_mhsaP: %ld work items", items);
  int grid = (int)std::min<long>(wgs > 0 ? wgs : want, items);       // wgs > 0: the tests' short grids (many items per workgroup, both list forms)
  if (grid >= 8) grid &= ~7;                    // whole XCD rounds (the kernel's XCD-aware list needs gridDim % 8 == 0)
  hipLaunchKernelGGL(se::mhsaP_fwd_kernel, dim3(grid), dim3(512), 0, st, qkv, lengths, T, H, ctx, never_speculate ? -1.f : 1.f, nqt, heads, heads * B);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

int se_mhsa8_fwd_launch(const uint16_t* qkv, const int32_t* lengths, int B, int T, int heads, uint16_t* ctx, int never_speculate, hipStream_t st) {
  const int H = heads * se::kHD;
  // the 32-bit lane offsets of the LDS-DMA address one (utterance, head) block: 2 (T - 1) 3 H + 6 H bytes
  SE_REQUIRE((double)T * 3.0 * H * 2.0 < 2147483648.0, "se_mhsa8: T * 3 H * 2 = %.0f bytes exceeds the 31-bit DMA offset", (double)T * 3.0 * H * 2.0);
  dim3 grid((T + se::k8Q - 1) / se::k8Q, heads, B);
#ifdef SE_AMD_STAMPS
  hipLaunchKernelGGL((se::mhsa8_fwd_kernel<0>), grid, dim3(512), 0, st, qkv, lengths, T, H, ctx, never_speculate ? -1.f : 1.f, (unsigned long long*)nullptr);
#else
  hipLaunchKernelGGL((se::mhsa8_fwd_kernel<0>), grid, dim3(512), 0, st, qkv, lengths, T, H, ctx, never_speculate ? -1.f : 1.f);
#endif
  SE_LAUNCH_CHECK();
  return SE_OK;
}

#ifdef SE_AMD_STAMPS
// developer entry of stamp builds (not in include/se_amd.h): `stamps` = grid workgroups x 8 waves x 10 uint64
extern "C" int se_mhsa8_fwd_stamps_bf16(const uint16_t* qkv, const int32_t* lengths, int B, int T, int heads, uint16_t* ctx, void* stamps, void* stream) {
  const int H = heads * se::kHD;
  dim3 grid((T + se::k8Q - 1) / se::k8Q, heads, B);
  hipLaunchKernelGGL((se::mhsa8_fwd_kernel<0>), grid, dim3(512), 0, se::as_stream(stream), qkv, lengths, T, H, ctx, 1.f, (unsigned long long*)stamps);
  SE_LAUNCH_CHECK();
  return SE_OK;
}
#endif
#endif      // SE_AMD_EXPERIMENTS
