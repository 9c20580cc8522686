// mhsa3.hip -- the inference flash MHSA forward (pre-scaled queries) as ONE interleaved matrix / vector instruction stream per wave.
//
// What the measurements of this round say (tools/mhsa_ablate.sh, tools/micro/coissue.hip, profiles/r03_*):
//   * in mhsa.hip the parts of a key tile ADD: MFMAs 41 %, staging 27 %, fragment reads 19 %, exponentials 11 %, rest 34 % of the launch;
//   * a wave's vector instructions issue under its OWN in-flight MFMA (1 MFMA + 8 v_fma per iteration: 49 cycles, the vector time alone), but are
//     held back by ANOTHER wave's MFMA on the same SIMD (+32 cycles per partner MFMA: fully additive).  Three waves per SIMD that each run
//     "8 MFMAs, then ~130 vector instructions, then 8 MFMAs" therefore serialise matrix and vector time, whatever the occupancy;
//   * two waves per SIMD that EACH interleave 1 MFMA : 8 vector instructions reach 36.7 cycles per MFMA (87 % of the matrix pipe).
// So: every wave owns two 32-query blocks A and B, and the softmax of one block is issued between the MFMAs of the other:
//     step 1   8 x { MFMA of S_B(t) = K(t) Q_B^T      ;  1/8 of the softmax of A: 4 x (exp2, row-sum add), 2 x bf16 pack }
//     step 2   8 x { MFMA of O_A += V(t)^T P_A(t)^T   ;  1/8 of the softmax of B }
//     step 3   8 x   MFMA of O_B += V(t)^T P_B(t)^T ,  8 x MFMA of S_A(t+1) = K(t+1) Q_A^T        (no vector work left: the partner wave's turn)
// with the LDS fragment of each MFMA read one slot ahead and sched_barrier fences between the slots so that the compiler keeps the interleave.
// Probabilities are SPECULATIVE (reference 0, no row maximum: mhsa.hip PRE); a row sum outside [2^-60, 2^60) anywhere in the workgroup makes the
// whole workgroup redo its queries with the exact online softmax (second pass below; never taken on bounded scores, tested with forced ones).
// K ring of 3 slots (S_A(t+1) needs K(t+1) during tile t), V ring of 2, LDS-DMA staging two / one tile ahead, ONE barrier per tile.
// Workgroup = 4 waves = 256 queries, 2 waves per SIMD (<= 256 registers), LDS 40 KiB.
#include <stdlib.h>
#include "common.h"
#include "prof.h"
#include "bf16.h"
#include "mhsa_tile.h"

namespace se {

constexpr int kAQ3 = 256;              // queries per workgroup (4 waves x 2 blocks x 32)
constexpr int kKSlots = 3, kVSlots = 2;
constexpr int kVBase = kKSlots * 8192; // V slots behind the K slots

#define SE_SB3() __builtin_amdgcn_sched_barrier(0)
// developer ablation (timing only): -DSE_MHSA3_ABL=<mask>: 1 no staging in the loop, 2 no barrier / DMA wait, 4 no softmax share in the slots, 8 no step-3 MFMAs,
// 16 no fragment reads (one fragment reused)
#ifndef SE_MHSA3_ABL
#define SE_MHSA3_ABL 0
#endif

// 1/8 of a block's speculative softmax: elements 4 (J & 3) .. + 3 of s0 (J < 4) or s1 (J >= 4): exp2, row-sum, bf16 pack into pf[J >> 2][(J >> 1) & 1]
template <int J>
__device__ __forceinline__ void sm_slot(const f32x16& s0, const f32x16& s1, bf16x8 (&pf)[2][2], float& rs) {
  const f32x16& src = (J < 4) ? s0 : s1;
  constexpr int e0 = 4 * (J & 3);
  const float a0 = __builtin_amdgcn_exp2f(src[e0]), a1 = __builtin_amdgcn_exp2f(src[e0 + 1]);
  const float a2 = __builtin_amdgcn_exp2f(src[e0 + 2]), a3 = __builtin_amdgcn_exp2f(src[e0 + 3]);
  rs += (a0 + a1) + (a2 + a3);
  bf16x8& d = pf[J >> 2][(J >> 1) & 1];
  constexpr int j0 = 4 * (J & 1);
  d[j0] = (__bf16)a0;
  d[j0 + 1] = (__bf16)a1;
  d[j0 + 2] = (__bf16)a2;
  d[j0 + 3] = (__bf16)a3;
}

// keeps a value (and everything it was computed from) inside the slot that produced it: without it LLVM sinks a slot's whole softmax share into the
// block of its first use, out from under the MFMAs it was written beside (compiler pitfalls)
__device__ __forceinline__ void pin_frag(bf16x8& x) {
  typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
  u32x4_ t = __builtin_bit_cast(u32x4_, x);
  asm volatile("" : "+v"(t));
  x = __builtin_bit_cast(bf16x8, t);
}

__device__ __forceinline__ bf16x8 ld_k(const char* tk, const int (&koff)[4], int i) {      // fragment of QK MFMA i: k-step i >> 1, key half i & 1
  return *reinterpret_cast<const bf16x8*>(tk + koff[i >> 1] + (i & 1) * 4096);
}
__device__ __forceinline__ bf16x8 ld_v(const char* tv, const int (&voff)[2][2], int i) {    // fragment of PV MFMA i: key block i >> 2, k-step (i >> 1) & 1, d-block i & 1
  const int o = (i >> 2) * 4096 + ((i >> 1) & 1) * 2048;
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(tv + voff[i & 1][0] + o));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(tv + voff[i & 1][1] + o));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void mhsa3_fwd_kernel(
    const uint16_t* __restrict__ qkv, const int32_t* __restrict__ lengths, int T, int H, uint16_t* __restrict__ ctx, int never_speculate) {
  __shared__ __attribute__((aligned(16))) char smem[(kKSlots + kVSlots) * 8192 + 16];
  int* fail_word = reinterpret_cast<int*>(smem + (kKSlots + kVSlots) * 8192);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  int b, head, qt;
  {      // XCD-aware work mapping (mhsa.hip): all query tiles of one (utterance, head) on ONE XCD, consecutive in its dispatch order
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
  const int q0 = qt * kAQ3 + wave * 64;
  const int ld = 3 * H;
  const int len = lengths ? min(max(lengths[b], 1), T) : T;
  const int nkt = (len + kAK - 1) / kAK;
  const uint16_t* base = qkv + (size_t)b * T * ld + head * kHD;
  if (tid == 0) *fail_word = 0;

  // ---- Q fragments of both blocks: lane -> query row q0 + 32 X + l31, d = 16 s + 8 hh .. +7
  bf16x8 qfA[4], qfB[4];
  {
    const uint16_t* qa = base + (size_t)min(q0 + l31, T - 1) * ld + 8 * hh;
    const uint16_t* qb = base + (size_t)min(q0 + 32 + l31, T - 1) * ld + 8 * hh;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qfA[s] = *reinterpret_cast<const bf16x8*>(qa + 16 * s);
      qfB[s] = *reinterpret_cast<const bf16x8*>(qb + 16 * s);
    }
  }

  // ---- LDS-DMA staging: wave w brings rows [16 w, 16 w + 16) of a K or V tile as two 1-KiB pieces (8 rows x 128 B); lane l of a piece writes slot
  //      l & 7 of row l >> 3, so it FETCHES chunk (l & 7) ^ f(row) (kv_off: slot = chunk ^ f)
  // Issued as inline asm (saddr + 32-bit lane offset, M0 = LDS address): through the builtin the compiler orders every later ds_read behind the
  // DMA with s_waitcnt vmcnt(0) -- in the middle of the tile, which drains the lookahead -- while here the waits are the counted ones below.
  typedef __attribute__((address_space(3))) char* lds_c_t;
  const int drow = wave * 16 + (lane >> 3);
  const uint32_t dch0 = (uint32_t)(((lane & 7) ^ (kv_off(drow, 0) >> 4 & 7)) * 16), dch1 = (uint32_t)(((lane & 7) ^ (kv_off(drow + 8, 0) >> 4 & 7)) * 16);
  const uint32_t lds_wave = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(lds_c_t)smem + wave * 2048);
  const char* gbase = reinterpret_cast<const char*>(base);
  auto stage = [&](int kt, int which /* 1 = K, 2 = V */, int lds_off) {
    const uint32_t o0 = (uint32_t)(min(kt * kAK + drow, T - 1) * ld + which * H) * 2u + dch0;
    const uint32_t o1 = (uint32_t)(min(kt * kAK + drow + 8, T - 1) * ld + which * H) * 2u + dch1;
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(o0), "s"(gbase), "s"(lds_wave + (uint32_t)lds_off) : "memory");
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(o1), "s"(gbase), "s"(lds_wave + (uint32_t)lds_off + 1024u) : "memory");
  };

  int koff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) koff[s] = kv_off(l31, 2 * s + hh);
  const int tq = (lane & 15) >> 2, tp = lane & 3, g1 = (lane >> 4) & 1;
  int voff[2][2];                     // [dblk][lo / hi]
#pragma unroll
  for (int dblk = 0; dblk < 2; ++dblk) {
    const int dcol = dblk * 32 + 16 * g1 + 4 * tp;
    voff[dblk][0] = kv_off(4 * hh + tq, dcol >> 3) + (dcol & 7) * 2;
    voff[dblk][1] = kv_off(4 * hh + tq + 8, dcol >> 3) + (dcol & 7) * 2;
  }

  const f32x16 kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 oA0 = kZero16, oA1 = kZero16, oB0 = kZero16, oB1 = kZero16;      // O^T d-blocks: col = query (lane & 31), row = d
  float lA = 0.f, lB = 0.f;
  bool fail = never_speculate != 0;

  // ================================ pass 0: speculative, interleaved ================================
  if (!fail) {
    stage(0, 1, 0);
    stage(0, 2, kVBase);
    if (nkt > 1) stage(1, 1, 8192);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // the compiler's own wait for the Q loads must sit HERE: inside the loop it would be a vmcnt(0) per tile that drains the hand-issued DMA
#pragma unroll
    for (int s = 0; s < 4; ++s) { pin_frag(qfA[s]); pin_frag(qfB[s]); }
    f32x16 sA0, sA1, sB0, sB1;
    // S_A(0): eight bare MFMAs
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bf16x8 kf = ld_k(smem, koff, i);
      if (i & 1) sA1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qfA[i >> 1], i < 2 ? kZero16 : sA1, 0, 0, 0);
      else sA0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qfA[i >> 1], i < 2 ? kZero16 : sA0, 0, 0, 0);
    }
    int kslot = 0, vslot = 0;           // slots of K(kt), V(kt)
    for (int kt = 0; kt < nkt; ++kt) {
      const int kslot1 = kslot == kKSlots - 1 ? 0 : kslot + 1;          // K(kt + 1)
      const int kslot2 = kslot1 == kKSlots - 1 ? 0 : kslot1 + 1;        // K(kt + 2): the slot K(kt - 1) leaves
      if (!(SE_MHSA3_ABL & 1)) {
      if (kt + 2 < nkt) stage(kt + 2, 1, kslot2 * 8192);
      if (kt + 1 < nkt) stage(kt + 1, 2, kVBase + (vslot ^ 1) * 8192);
      }
      const char* tk = smem + kslot * 8192;
      const char* tk1 = smem + kslot1 * 8192;
      const char* tv = smem + kVBase + vslot * 8192;
      const bool tail = (kt + 1) * kAK > len;
      if (tail) {      // keys >= len of block A's scores (computed in the previous iteration's step 3): -inf -> probability 0
        const int kbase = kt * kAK + 4 * hh;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase + (r & 3) + 8 * (r >> 2);
          if (key >= len) sA0[r] = -INFINITY;
          if (key + 32 >= len) sA1[r] = -INFINITY;
        }
      }
      bf16x8 pfA[2][2], pfB[2][2];
      float rsA = 0.f, rsB = 0.f;
      // ---- step 1: S_B(t) MFMAs, softmax of A between them
      {
        // fragments are read TWO slots ahead: with eight waves reading, an LDS fragment takes ~190 cycles from issue to use (ablation: a slot
        // that waits for a read issued one slot earlier lasts as long as that latency, whatever it computes)
        bf16x8 kf = ld_k(tk, koff, 0), kf1 = ld_k(tk, koff, 1);
#define SE3_S1(I)                                                                                                                      \
        {                                                                                                                              \
          bf16x8 kn = kf1;                                                                                                             \
          if ((I) < 6 && !(SE_MHSA3_ABL & 16)) kn = ld_k(tk, koff, (I) + 2);                                                           \
          if ((I) & 1) sB1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qfB[(I) >> 1], (I) < 2 ? kZero16 : sB1, 0, 0, 0);             \
          else sB0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qfB[(I) >> 1], (I) < 2 ? kZero16 : sB0, 0, 0, 0);                     \
          SE_SB3();            /* the fragment read and this slot's MFMA go first, the vector share behind them */                    \
          if (!(SE_MHSA3_ABL & 4) || (I) == 0) sm_slot<(I)>(sA0, sA1, pfA, rsA);                                                       \
          asm volatile("" : "+v"(rsA));                                                                                                \
          if ((I) & 1) pin_frag(pfA[(I) >> 2][((I) >> 1) & 1]);                                                                        \
          kf = kf1;                                                                                                                    \
          kf1 = kn;                                                                                                                    \
          SE_SB3();                                                                                                                    \
        }
        SE3_S1(0) SE3_S1(1) SE3_S1(2) SE3_S1(3) SE3_S1(4) SE3_S1(5) SE3_S1(6) SE3_S1(7)
#undef SE3_S1
      }
      if (tail) {
        const int kbase = kt * kAK + 4 * hh;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kbase + (r & 3) + 8 * (r >> 2);
          if (key >= len) sB0[r] = -INFINITY;
          if (key + 32 >= len) sB1[r] = -INFINITY;
        }
      }
      // ---- step 2: O_A += V^T P_A^T MFMAs, softmax of B between them
      {
        bf16x8 vf = ld_v(tv, voff, 0), vf1 = ld_v(tv, voff, 1);
#define SE3_S2(I)                                                                                                                      \
        {                                                                                                                              \
          bf16x8 vn = vf1;                                                                                                             \
          if ((I) < 6 && !(SE_MHSA3_ABL & 16)) vn = ld_v(tv, voff, (I) + 2);                                                           \
          if ((I) & 1) oA1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pfA[(I) >> 2][((I) >> 1) & 1], oA1, 0, 0, 0);                  \
          else oA0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pfA[(I) >> 2][((I) >> 1) & 1], oA0, 0, 0, 0);                          \
          SE_SB3();                                                                                                                    \
          if (!(SE_MHSA3_ABL & 4) || (I) == 0) sm_slot<(I)>(sB0, sB1, pfB, rsB);                                                       \
          asm volatile("" : "+v"(rsB));                                                                                                \
          if ((I) & 1) pin_frag(pfB[(I) >> 2][((I) >> 1) & 1]);                                                                        \
          vf = vf1;                                                                                                                    \
          vf1 = vn;                                                                                                                    \
          SE_SB3();                                                                                                                    \
        }
        SE3_S2(0) SE3_S2(1) SE3_S2(2) SE3_S2(3) SE3_S2(4) SE3_S2(5) SE3_S2(6) SE3_S2(7)
#undef SE3_S2
      }
      // ---- step 3: O_B += V^T P_B^T and S_A(t + 1) = K(t + 1) Q_A^T: sixteen MFMAs, no vector work of this wave left
      if (!(SE_MHSA3_ABL & 8)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const bf16x8 vf = ld_v(tv, voff, i);
          if (i & 1) oB1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pfB[i >> 2][(i >> 1) & 1], oB1, 0, 0, 0);
          else oB0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pfB[i >> 2][(i >> 1) & 1], oB0, 0, 0, 0);
        }
        if (kt + 1 < nkt) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const bf16x8 kf = ld_k(tk1, koff, i);
            if (i & 1) sA1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qfA[i >> 1], i < 2 ? kZero16 : sA1, 0, 0, 0);
            else sA0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qfA[i >> 1], i < 2 ? kZero16 : sA0, 0, 0, 0);
          }
        }
      }
      // row sums tell whether the speculation held (overflow / inf / nan; on the first tile also a row far below the reference 0)
      const float rsa = rsA + __shfl_xor(rsA, 32), rsb = rsB + __shfl_xor(rsB, 32);
      fail = fail || !(rsa < 0x1p60f) || !(rsb < 0x1p60f) || (kt == 0 && (rsa < 0x1p-60f || rsb < 0x1p-60f));
      lA += rsA;
      lB += rsB;
      kslot = kslot1;
      vslot ^= 1;
      if (!(SE_MHSA3_ABL & 2)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this tile's DMA (K(t + 2), V(t + 1)) has landed
      __builtin_amdgcn_s_barrier();
      }
    }
    if (__any(fail)) *fail_word = 1;
    __syncthreads();
    fail = *fail_word != 0;            // workgroup-uniform
  }

  // ================================ pass 1 (rare): exact online softmax, one block after the other ================================
  if (fail) {
    __syncthreads();
#pragma unroll 1
    for (int X = 0; X < 2; ++X) {
      bf16x8 qf[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) qf[s] = X ? qfB[s] : qfA[s];
      f32x16 o0 = kZero16, o1 = kZero16;
      float m_run = 0.f, l_run = 0.f;
      constexpr float kDefer = 8.f;
      stage(0, 1, 0);
      stage(0, 2, kVBase);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#pragma unroll 1
      for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) {
          stage(kt + 1, 1, (cur ^ 1) * 8192);
          stage(kt + 1, 2, kVBase + (cur ^ 1) * 8192);
        }
        const char* tk = smem + cur * 8192;
        const char* tv = smem + kVBase + cur * 8192;
        f32x16 s0, s1;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const bf16x8 kf = ld_k(tk, koff, i);
          if (i & 1) s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[i >> 1], i < 2 ? kZero16 : s1, 0, 0, 0);
          else s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[i >> 1], i < 2 ? kZero16 : s0, 0, 0, 0);
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
        float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s0[r]), s1[r]);
        {
          const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
          mx = fmaxf(__uint_as_float(sw_[0]), __uint_as_float(sw_[1]));
        }
        float m_new = ((mx - m_run) > kDefer) ? mx : m_run;
        if (kt == 0 && mx < -64.f) m_new = mx;        // a first tile far below the initial reference 0 (later tiles cannot matter)
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float rs0 = 0.f, rs1 = 0.f;
        bf16x8 pf[2][2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float a0 = __builtin_amdgcn_exp2f(s0[r] - m_new);
          const float a1 = __builtin_amdgcn_exp2f(s1[r] - m_new);
          rs0 += a0;
          rs1 += a1;
          pf[0][r >> 3][r & 7] = (__bf16)a0;
          pf[1][r >> 3][r & 7] = (__bf16)a1;
        }
        l_run = fmaf(l_run, alpha, rs0 + rs1);
        m_run = m_new;
        if (__any(alpha != 1.0f)) {
#pragma unroll
          for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const bf16x8 vf = ld_v(tv, voff, i);
          if (i & 1) o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[i >> 2][(i >> 1) & 1], o1, 0, 0, 0);
          else o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[i >> 2][(i >> 1) & 1], o0, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      if (X) { oB0 = o0; oB1 = o1; lB = l_run; } else { oA0 = o0; oA1 = o1; lA = l_run; }
    }
  }

  // ---- epilogue: O / l ; lane holds query q0 + 32 X + l31, d = 32 dblk + (r&3) + 8 (r>>2) + 4 hh
#pragma unroll
  for (int X = 0; X < 2; ++X) {
    const float lx = X ? lB : lA;
    const f32x16& o0 = X ? oB0 : oA0;
    const f32x16& o1 = X ? oB1 : oA1;
    const float l_tot = lx + __shfl_xor(lx, 32);
    const float inv = 1.0f / l_tot;
    const int q = q0 + 32 * X + l31;
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
}

}  // namespace se

int se_mhsa3_fwd_launch(const uint16_t* qkv, const int32_t* lengths, int B, int T, int heads, uint16_t* ctx, int never_speculate, hipStream_t st) {
  const int H = heads * se::kHD;
  dim3 grid((T + se::kAQ3 - 1) / se::kAQ3, heads, B);
  hipLaunchKernelGGL(se::mhsa3_fwd_kernel, grid, dim3(256), 0, st, qkv, lengths, T, H, ctx, never_speculate);
  SE_LAUNCH_CHECK();
  return SE_OK;
}
