// mhsa_pipe.hip -- EXPERIMENT, off by default (SE_AMD_MHSA_PIPE=1 enables it behind se_mhsa_fwd_prescaled_bf16): the inference flash MHSA
// forward as a SOFTWARE-PIPELINED loop over 32-key half tiles.  Same data layout, fragments and LDS swizzle as mhsa.hip.
//
// Idea.  mhsa.hip runs, per wave and 64-key tile, three phases one after the other: 8 MFMAs (S^T = K Q^T), ~160 vector instructions
// (online softmax), 8 MFMAs (O^T += V^T P^T); matrix and vector work of ONE wave never overlap, only those of the 3 waves of a SIMD do
// when their phases happen to differ.  Here one wave interleaves, in program order, the vector work of half tile h with the matrix work
// around it:
//     body(h):   [matrix]  S(h+1) = K(h+1) Q^T  (4 MFMAs)      and      O += V(h-1)^T P(h-1)^T  (4 MFMAs)
//                [vector]  P(h) = exp2(S(h) - m), row max / sum, bf16 pack
// eight slots of {1 MFMA, ~10 vector instructions}, fenced with sched_barrier so the compiler keeps the interleave; the LDS fragment
// reads of a slot's MFMA are issued two slots earlier.  The online softmax runs per HALF tile (32 keys): S(h) and S(h+1) are the two
// accumulator blocks the old kernel already had, P(h-1) / P(h) two 8-register bf16 fragments.  The rare O rescale (deferred: only
// when a half tile's maximum exceeds the running reference by 2^8) happens at the top of the next body, after the in-flight O MFMAs.
// K / V ring: two 16 KiB slots as before, but K(t+1) lands at the end of body(2t) (one barrier per tile, there) and V(t+1) at the end of
// body(2t+1): K(t+1) is needed half a tile before V(t) is dead.
//
// Result (B = 32, T = 1001, 12 heads, same box, tools/bench_kernels.py mhsa): 169.5 us against 143.4 us for mhsa.hip -- SLOWER, results
// identical to 7e-3 of the unscaled kernel.  Why, from the ablation switches below and tools/micro/valu_rate.hip:
//   * a SIMD issues ONE plain vector instruction per 4 cycles and one v_exp_f32 per 8, whatever the number of resident waves (micro:
//     16 v_add_f32 cost 67-74 cycles with 1, 2 or 4 waves on the SIMD; 16 v_exp_f32 132-155).  The 170 vector instructions of a
//     (32 query x 64 key) wave tile therefore cost >= 816 cycles of SIMD time against 512 for its 16 MFMAs: at head dim 64 this kernel is
//     VECTOR-ISSUE bound by construction, and an MFMA issued into a busy vector stream still adds 13-20 cycles (micro: +13-15 per MFMA;
//     here: the vector + LDS stream alone runs in 90 us = its issue floor, adding the MFMAs +32 us, their LDS fragment waits +10-19 us).
//   * what is left is the staging of the K / V tiles (global loads + LDS writes: +35 us here) and the barrier (+8 us); with 2 waves per
//     SIMD (182 registers) these stalls are exposed, while mhsa.hip's 3 waves cover each other's.  Interleaving inside a wave buys
//     nothing once the vector pipe is the bound: every cycle it hides is a matrix-pipe cycle, and the matrix pipe had 64 % slack anyway.
// Floor for this instruction mix at d = 64: exp 32 x 8 + row sums 32 x 4 + max 16 x 4 + pack 16 x 4 + MFMA issue 16 x 8 + LDS issue
// 24 x 4 + ~60 other = ~800 cycles per wave tile = 77 us per launch at 2.0 GHz with NO stall of any kind (0.51 of the 2.5 PF
// peak); mhsa.hip runs at 1 430 (0.28-0.32).  SE_AMD_MHSA_ABL: 1 no staging, 2 no barrier; SE_AMD_MHSA_ABLK: 4 no fragment reads,
// 8 no exponentials, 16 no MFMAs (timing only: results are garbage).
#include <stdlib.h>
#include "common.h"
#include "prof.h"
#include "bf16.h"
#include "mhsa_tile.h"

namespace se {

#define SE_SB() __builtin_amdgcn_sched_barrier(0)

struct PipeWave {
  f32x16 o0, o1;          // O^T d-blocks: col = query (lane & 31), row = d
  f32x16 S[2];            // S^T of the half tile being exponentiated / being produced
  bf16x8 qf[4];
  bf16x8 pf[2][2];        // [half parity][k-step]
  bf16x8 va0;             // V^T fragment (k-step 0, d-block 0) of the NEXT body's first O MFMA
  float m_run, l_run, alpha_prev;
};

__device__ __forceinline__ bf16x8 ld_vt(const char* p_lo, const char* p_hi) {
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p_lo));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(p_hi));
  return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__device__ __forceinline__ void pin_frag(bf16x8& x) {
  typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
  u32x4_ t = __builtin_bit_cast(u32x4_, x);
  asm volatile("" : "+v"(t));
  x = __builtin_bit_cast(bf16x8, t);
}

// One half step.  KB: parity of the half tile whose scores are exponentiated (S[KB] -> pf[KB]); the MFMAs produce S[KB ^ 1] from the K half
// at `kq` (DOQK) and add P(h-1) V(h-1) from the V half at `vp` (pf[KB ^ 1]).  MASK: keys >= len get -inf (last tile only).
template <int KB, int MASK, int DOQK, int ABL>
__device__ __forceinline__ void half_step(PipeWave& w, const char* kq, const char* vp, const char* vp_next, const int (&koff)[4], const int (&voff)[2][2], int key0,
                                          int len, bool first) {
  const f32x16 kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  constexpr float kDefer = 8.f;
  f32x16& Sc = w.S[KB];
  f32x16& Sn = w.S[KB ^ 1];
  bf16x8(&pfp)[2] = w.pf[KB ^ 1];
  bf16x8(&pfc)[2] = w.pf[KB];
#define SE_LDK(s) ((ABL & 4) ? w.qf[s] : *reinterpret_cast<const bf16x8*>(kq + koff[s]))
#define SE_LDV(s, d) ((ABL & 4) ? w.qf[(s) + 2 * (d)] : ld_vt(vp + voff[d][0] + (s) * 2048, vp + voff[d][1] + (s) * 2048))
#define SE_PAIR(i)                                                   \
  {                                                                  \
    const float a_ = (ABL & 8) ? (Sc[2 * (i)] + mc) * 0.5f : __builtin_amdgcn_exp2f(Sc[2 * (i)] + mc);       \
    const float b_ = (ABL & 8) ? (Sc[2 * (i) + 1] + mc) * 0.5f : __builtin_amdgcn_exp2f(Sc[2 * (i) + 1] + mc);   \
    rsa += a_;                                                       \
    rsb += b_;                                                       \
    Sc[2 * (i)] = a_;                                                \
    Sc[2 * (i) + 1] = b_;                                            \
  }
  // the O MFMAs of the previous body have to land before a rescale: rare (deferred maximum), so a branch
  if (__any(w.alpha_prev != 1.0f)) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { w.o0[r] *= w.alpha_prev; w.o1[r] *= w.alpha_prev; }
  }
  if (MASK) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (key0 + (r & 3) + 8 * (r >> 2) >= len) Sc[r] = -INFINITY;
  }
  // fragment reads run two slots ahead of the MFMA that consumes them; the first O fragment was read by the previous body (w.va0)
  // (KB = 1 reads its own first O fragment here: V(t) only becomes visible with the barrier in front of this body)
  if (KB == 1) w.va0 = SE_LDV(0, 0);
  bf16x8 ka = SE_LDK(0), kb_ = SE_LDK(1), kc, kd;
  bf16x8 va, vb, vc;
  SE_SB();
  // slot 0: O (k-step 0, d-block 0)
  if (!(ABL & 16)) w.o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.va0, pfp[0], w.o0, 0, 0, 0);
  vb = SE_LDV(0, 1);
  float mxa = fmaxf(fmaxf(Sc[0], Sc[1]), Sc[2]), mxb = fmaxf(fmaxf(Sc[8], Sc[9]), Sc[10]);
  mxa = fmaxf(fmaxf(mxa, Sc[3]), Sc[4]);
  mxb = fmaxf(fmaxf(mxb, Sc[11]), Sc[12]);
  SE_SB();
  // slot 1: S k-step 0
  if (DOQK && !(ABL & 16)) Sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, w.qf[0], kZero16, 0, 0, 0);
  kc = SE_LDK(2);
  mxa = fmaxf(fmaxf(mxa, Sc[5]), Sc[6]);
  mxb = fmaxf(fmaxf(mxb, Sc[13]), Sc[14]);
  float mx = fmaxf(fmaxf(mxa, Sc[7]), fmaxf(mxb, Sc[15]));
  {   // the other 16 keys of this query row live in lane ^ 32
    const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
    mx = fmaxf(__uint_as_float(sw_[0]), __uint_as_float(sw_[1]));
  }
  SE_SB();
  // slot 2: S k-step 1
  if (DOQK && !(ABL & 16)) Sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb_, w.qf[1], Sn, 0, 0, 0);
  va = SE_LDV(1, 0);
  float m_new = (mx - w.m_run > kDefer) ? mx : w.m_run;
  if (KB == 0) m_new = (first && mx < -64.f) ? mx : m_new;      // a first half tile far below the initial reference 0
  const float alpha = __builtin_amdgcn_exp2f(w.m_run - m_new);
  const float mc = -m_new;
  float rsa = 0.f, rsb = 0.f;
  SE_PAIR(0)
  SE_SB();
  // slot 3: O (0, 1)
  if (!(ABL & 16)) w.o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb, pfp[0], w.o1, 0, 0, 0);
  kd = SE_LDK(3);
  SE_PAIR(1)
  SE_PAIR(2)
  SE_SB();
  // slot 4: S k-step 2
  if (DOQK && !(ABL & 16)) Sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kc, w.qf[2], Sn, 0, 0, 0);
  vc = SE_LDV(1, 1);
  SE_PAIR(3)
  SE_PAIR(4)
  SE_SB();
  // slot 5: O (1, 0)
  if (!(ABL & 16)) w.o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pfp[1], w.o0, 0, 0, 0);
  SE_PAIR(5)
  SE_PAIR(6)
  SE_SB();
  // slot 6: S k-step 3
  if (DOQK && !(ABL & 16)) Sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kd, w.qf[3], Sn, 0, 0, 0);
  if (KB == 1) w.va0 = (ABL & 4) ? w.qf[0] : ld_vt(vp_next + voff[0][0], vp_next + voff[0][1]);      // the next body's first O fragment
  SE_PAIR(7)
  w.l_run = fmaf(w.l_run, alpha, rsa + rsb);
  w.m_run = m_new;
  w.alpha_prev = alpha;
  SE_SB();
  // slot 7: O (1, 1)
  if (!(ABL & 16)) w.o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vc, pfp[1], w.o1, 0, 0, 0);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    pfc[0][j] = (__bf16)Sc[j];
    pfc[1][j] = (__bf16)Sc[8 + j];
  }
  // opaque uses: without them LLVM's code sinking moves the whole exponential stream into the NEXT basic block (its only users,
  // the next body's MFMAs, live there), i.e. out from under this body's MFMAs
  pin_frag(pfc[0]);
  pin_frag(pfc[1]);
  pin_frag(w.va0);
  asm volatile("" : "+v"(w.l_run), "+v"(w.m_run), "+v"(w.alpha_prev));
  SE_SB();
#undef SE_LDK
#undef SE_LDV
#undef SE_PAIR
}

template <int OCC, int ABL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void mhsa_fwd_pipe_kernel(
    const uint16_t* __restrict__ qkv, const int32_t* __restrict__ lengths, int T, int H, uint16_t* __restrict__ ctx, int abl) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * kAK * kHD * 2];   // 2 slots x (K, V) x 8 KiB

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
  const int q0 = qt * kAQ + wave * 32;
  const int ld = 3 * H;
  const int len = lengths ? min(max(lengths[b], 1), T) : T;
  const int nkt = (len + kAK - 1) / kAK;
  const uint16_t* base = qkv + (size_t)b * T * ld + head * kHD;

  PipeWave w;
  {
    const int q = min(q0 + l31, T - 1);
    const uint16_t* qp = base + (size_t)q * ld + 8 * hh;
#pragma unroll
    for (int s = 0; s < 4; ++s) w.qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
  }

  // ---- staging: K and V tiles are 64 rows x 128 B; 256 threads x 16 B = 32 rows per pass
  const int srow = tid >> 3, sch = tid & 7;
  const uint16_t* kp = base + H + sch * 8;
  const uint16_t* vp = base + 2 * H + sch * 8;
  uint4 rk0, rk1, rv0, rv1;
  const int so0 = kv_off(srow, sch), so1 = kv_off(srow + 32, sch);
#define SE_P_ISSUE_K(kt)                                                                     \
  do {                                                                                       \
    rk0 = *reinterpret_cast<const uint4*>(kp + (size_t)min((kt) * kAK + srow, T - 1) * ld);      \
    rk1 = *reinterpret_cast<const uint4*>(kp + (size_t)min((kt) * kAK + srow + 32, T - 1) * ld); \
  } while (0)
#define SE_P_ISSUE_V(kt)                                                                     \
  do {                                                                                       \
    rv0 = *reinterpret_cast<const uint4*>(vp + (size_t)min((kt) * kAK + srow, T - 1) * ld);      \
    rv1 = *reinterpret_cast<const uint4*>(vp + (size_t)min((kt) * kAK + srow + 32, T - 1) * ld); \
  } while (0)
#define SE_P_WRITE_K(slot)                                                       \
  do {                                                                           \
    *reinterpret_cast<uint4*>(smem + (slot) * 16384 + so0) = rk0;                \
    *reinterpret_cast<uint4*>(smem + (slot) * 16384 + so1) = rk1;                \
  } while (0)
#define SE_P_WRITE_V(slot)                                                       \
  do {                                                                           \
    *reinterpret_cast<uint4*>(smem + (slot) * 16384 + 8192 + so0) = rv0;         \
    *reinterpret_cast<uint4*>(smem + (slot) * 16384 + 8192 + so1) = rv1;         \
  } while (0)
  // LDS-only synchronisation: the global loads of the next tile stay in flight across it
#define SE_P_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

  // ---- loop-invariant LDS byte offsets (as mhsa.hip)
  int koff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) koff[s] = kv_off(l31, 2 * s + hh);
  const int tq = (lane & 15) >> 2, tp = lane & 3, g1 = (lane >> 4) & 1;
  int voff[2][2];
#pragma unroll
  for (int dblk = 0; dblk < 2; ++dblk) {
    const int dcol = dblk * 32 + 16 * g1 + 4 * tp;
    voff[dblk][0] = 8192 + kv_off(4 * hh + tq, dcol >> 3) + (dcol & 7) * 2;
    voff[dblk][1] = 8192 + kv_off(4 * hh + tq + 8, dcol >> 3) + (dcol & 7) * 2;
  }

#pragma unroll
  for (int r = 0; r < 16; ++r) { w.o0[r] = 0.f; w.o1[r] = 0.f; }
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) w.pf[1][s][j] = (__bf16)0.f;      // P(-1) = 0: body(0) multiplies it with ...
  w.m_run = 0.f;
  w.l_run = 0.f;
  w.alpha_prev = 1.0f;

  SE_P_ISSUE_K(0);
  SE_P_ISSUE_V(0);
  // ... the V half of slot 1, which nothing has written yet: zero it (0 x garbage could be NaN)
  *reinterpret_cast<uint4*>(smem + 16384 + 8192 + tid * 32) = make_uint4(0u, 0u, 0u, 0u);
  *reinterpret_cast<uint4*>(smem + 16384 + 8192 + tid * 32 + 16) = make_uint4(0u, 0u, 0u, 0u);
  SE_P_WRITE_K(0);
  SE_P_WRITE_V(0);
  SE_P_BARRIER();
  {   // S(0) = K(0, half 0) Q^T
    const f32x16 kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s)
      w.S[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(smem + koff[s]), w.qf[s], s == 0 ? kZero16 : w.S[0], 0, 0, 0);
  }

  SE_P_ISSUE_K(1);      // rows are clamped to T - 1: a load past the last tile is harmless (and never written to LDS)
  w.va0 = ld_vt(smem + 16384 + 4096 + voff[0][0], smem + 16384 + 4096 + voff[0][1]);      // of the zeroed V half that body(0) multiplies with P(-1) = 0

  // tile kt with its K / V in slot CUR.  LAST: no next tile (no LDS fills, no S(next)), keys >= len masked.
  // K(kt + 1) was requested one tile ago (it is needed in the middle of this tile), V(kt + 1) is requested at the top of this one.
#define SE_P_TILE(CUR, LAST)                                                                                                          \
  {                                                                                                                                   \
    if (!(LAST) && !(abl & 1)) SE_P_ISSUE_V(kt + 1);                                                                                  \
    half_step<0, LAST, 1, ABL>(w, smem + (CUR) * 16384 + 4096, smem + ((CUR) ^ 1) * 16384 + 4096, smem + (CUR) * 16384, koff, voff,   \
                               kt * kAK + 4 * hh, len, kt == 0);                                                                      \
    if (!(LAST) && !(abl & 1)) SE_P_WRITE_K((CUR) ^ 1);                                                                               \
    if (!(abl & 2)) SE_P_BARRIER();                                                                                                   \
    if (!(LAST) && !(abl & 1)) SE_P_ISSUE_K(kt + 2);                                                                                  \
    half_step<1, LAST, !(LAST), ABL>(w, smem + ((CUR) ^ 1) * 16384, smem + (CUR) * 16384, smem + (CUR) * 16384 + 4096, koff, voff,    \
                                     kt * kAK + 32 + 4 * hh, len, false);                                                             \
    if (!(LAST) && !(abl & 1)) SE_P_WRITE_V((CUR) ^ 1);                                                                               \
  }
  // the last half's P V product: P(2 nkt - 1) with the V half 1 of the last tile's slot
#define SE_P_FINAL(CUR)                                                                                                   \
  {                                                                                                                       \
    if (__any(w.alpha_prev != 1.0f)) {                                                                                    \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) { w.o0[r] *= w.alpha_prev; w.o1[r] *= w.alpha_prev; }                \
    }                                                                                                                     \
    const char* v_s = smem + (CUR) * 16384 + 4096;                                                                        \
    w.o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.va0, w.pf[1][0], w.o0, 0, 0, 0);                                     \
    w.o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_vt(v_s + voff[1][0], v_s + voff[1][1]), w.pf[1][0], w.o1, 0, 0, 0); \
    w.o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_vt(v_s + voff[0][0] + 2048, v_s + voff[0][1] + 2048), w.pf[1][1], w.o0, 0, 0, 0); \
    w.o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_vt(v_s + voff[1][0] + 2048, v_s + voff[1][1] + 2048), w.pf[1][1], w.o1, 0, 0, 0); \
  }

  int kt = 0;
  for (; kt + 2 < nkt; kt += 2) {
    SE_P_TILE(0, 0)
    ++kt;
    SE_P_TILE(1, 0)
    --kt;
  }
  if (nkt - kt == 2) {
    SE_P_TILE(0, 0)
    ++kt;
    SE_P_TILE(1, 1)
    SE_P_FINAL(1)
  } else {
    SE_P_TILE(0, 1)
    SE_P_FINAL(0)
  }
#undef SE_P_TILE
#undef SE_P_FINAL

  // ---- epilogue: O / l ; lane holds query q0 + l31, d = 32 dblk + (r&3) + 8 (r>>2) + 4 hh
  const float l_tot = w.l_run + __shfl_xor(w.l_run, 32);
  const float inv = 1.0f / l_tot;
  const int q = q0 + l31;
  if (q < T) {
    uint16_t* op = ctx + ((size_t)b * T + q) * H + head * kHD + 4 * hh;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint2 w0 = make_uint2(pack_bf16x2(w.o0[4 * g] * inv, w.o0[4 * g + 1] * inv), pack_bf16x2(w.o0[4 * g + 2] * inv, w.o0[4 * g + 3] * inv));
      uint2 w1 = make_uint2(pack_bf16x2(w.o1[4 * g] * inv, w.o1[4 * g + 1] * inv), pack_bf16x2(w.o1[4 * g + 2] * inv, w.o1[4 * g + 3] * inv));
      *reinterpret_cast<uint2*>(op + 8 * g) = w0;
      *reinterpret_cast<uint2*>(op + 32 + 8 * g) = w1;
    }
  }
}

}  // namespace se

// launched by se_mhsa_fwd_prescaled_bf16 (mhsa.hip) unless SE_AMD_MHSA_PIPE=0
int se_mhsa_fwd_pipe_launch(const uint16_t* qkv, const int32_t* lengths, int B, int T, int heads, uint16_t* ctx, int occ, hipStream_t st) {
  const int H = heads * se::kHD;
  dim3 grid((T + se::kAQ - 1) / se::kAQ, heads, B);
  static int abl = -1;
  if (abl < 0) { const char* e = getenv("SE_AMD_MHSA_ABL"); abl = e ? atoi(e) : 0; }
  static int ablk = -1;
  if (ablk < 0) { const char* e = getenv("SE_AMD_MHSA_ABLK"); ablk = e ? atoi(e) : 0; }
#define SE_PIPE_LAUNCH(O, A) hipLaunchKernelGGL((se::mhsa_fwd_pipe_kernel<O, A>), grid, dim3(256), 0, st, qkv, lengths, T, H, ctx, abl)
  if (occ == 3) SE_PIPE_LAUNCH(3, 0);
  else if (ablk == 4) SE_PIPE_LAUNCH(2, 4);
  else if (ablk == 8) SE_PIPE_LAUNCH(2, 8);
  else if (ablk == 16) SE_PIPE_LAUNCH(2, 16);
  else if (ablk == 12) SE_PIPE_LAUNCH(2, 12);
  else if (ablk == 20) SE_PIPE_LAUNCH(2, 20);
  else SE_PIPE_LAUNCH(2, 0);
  SE_LAUNCH_CHECK();
  return SE_OK;
}
