// mhsa9.hip -- row B2 core, inference form (pre-scaled queries): 8 free-running waves on one LDS-DMA staged K / V tile, each wave SOFTWARE-PIPELINED
// over the key tiles: the score products of tile t + 1 are issued in front of the softmax of tile t (two score buffers), and every LDS fragment is
// read one phase ahead of the MFMAs that consume it.  Why (round 4, tools/micro/attn_skel.hip = the tile's MFMAs + exponentials + sums + packs with
// their true dependencies and NO memory traffic): in program order (QK^T | softmax | PV, what mhsa.hip / mhsa8.hip do) that skeleton alone costs
// 1 470 cycles per (32 query x 64 key) unit for one wave and 780 / 735 / 700 at 2 / 3 / 4 waves per SIMD; with the next tile's QK^T in front of this
// tile's softmax 740 for ONE wave and 600 at 2 or 3 per SIMD (the matrix floor is 512).  mhsa.hip runs at ~1 300 per unit.
//
// Workgroup = 8 waves = 256 query rows of one (utterance, head), one workgroup per CU (<= 256 registers per lane).  Five-slot ring of 16-KiB K / V
// tiles; per tile t a wave
//   a  issues its two LDS-DMA pieces of tile t + 4 (the slot of tile t - 1: every wave left it before the barrier that ended tile t - 1)
//   b  S(t+1) = K(t+1) Q^T        8 MFMAs on the K fragments read during tile t - 1     } one basic block: the compiler / the hardware run the
//   c  softmax(t) -> P(t)         32 exp2, 32 adds, 16 packs on S(t)                      } vector work of c beside the matrix work of b
//   d  O += V(t)^T P(t)^T         8 MFMAs, their V^T fragments read as they go (prefetching them too does not fit 256 registers: 147 spills)
//   e  reads the K(t+2) fragments (8 LDS reads), awaits its own pieces of tile t + 3 (counted vmcnt), one barrier
// The speculative (no row maximum) softmax of mhsa.hip is the fast loop; a wave whose row sums leave [2^-60, 2^60) redoes that tile and does all later
// ones in the exact online-softmax loop (same steps, same barrier / DMA protocol, so the other waves of the workgroup are not disturbed).
#include <stdlib.h>
#include "common.h"
#include "prof.h"
#include "bf16.h"
#include "mhsa_tile.h"

namespace se {

constexpr int k9Q = 256;        // query rows per workgroup
constexpr int k9Slot = 16384;   // one ring slot: K tile (8 KiB) + V tile (8 KiB)
constexpr int k9Ring = 5;      // tiles t (V still read by PV), t + 1, t + 2, t + 3 resident + t + 4 arriving

#define SE9_BAR()                                   \
  do {                                              \
    __builtin_amdgcn_sched_barrier(0);              \
    asm volatile("s_barrier" ::: "memory");         \
    __builtin_amdgcn_sched_barrier(0);              \
  } while (0)
#define SE9_PIN8(a)                                                                                         \
  asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]))

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void mhsa9_fwd_kernel(
    const uint16_t* __restrict__ qkv, const int32_t* __restrict__ lengths, int T, int H, uint16_t* __restrict__ ctx, float dscale) {
  __shared__ __attribute__((aligned(16))) char smem[k9Ring * k9Slot];

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
  const int q0 = qt * k9Q + wave * 32;
  const int ld = 3 * H;
  const int len = lengths ? min(max(lengths[b], 1), T) : T;
  const int nkt = (len + kAK - 1) / kAK;
  const uint16_t* base = qkv + (size_t)b * T * ld + head * kHD;

  // ---- LDS-DMA: wave w brings rows [8 w, 8 w + 8) of the K and of the V tile (one 1-KiB piece each); lane l writes slot l & 7 of row l >> 3
  typedef __attribute__((address_space(3))) char* lds_c_t;
  const int drow = wave * 8 + (lane >> 3);
  const uint32_t dch = (uint32_t)(((lane & 7) ^ (kv_off(drow, 0) >> 4 & 7)) * 16);
  const uint32_t lds_wave = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(lds_c_t)smem + wave * 1024);
  const char* gbase = reinterpret_cast<const char*>(base);
#define SE9_DMA(kt, slot)                                                                                                   \
  do {                                                                                                                      \
    const uint32_t row_ = (uint32_t)(min((kt) * kAK + drow, T - 1) * ld);                                                   \
    const uint32_t ok_ = (row_ + (uint32_t)H) * 2u + dch, ov_ = (row_ + 2u * (uint32_t)H) * 2u + dch;                       \
    uint32_t keep_;                                                                                                         \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"    \
                 : "=&s"(keep_) : "v"(ok_), "s"(gbase), "s"(lds_wave + (uint32_t)((slot) * k9Slot)) : "memory");            \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"    \
                 : "=&s"(keep_) : "v"(ov_), "s"(gbase), "s"(lds_wave + (uint32_t)((slot) * k9Slot + 8192)) : "memory");     \
  } while (0)

  // tiles 0 .. 3 (rows past the length are clamped: they only feed masked keys or tiles that are never used)
  SE9_DMA(0, 0); SE9_DMA(1, 1); SE9_DMA(2, 2); SE9_DMA(3, 3);

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
  constexpr float kDefer = 8.f;

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

  bf16x8 kf[8];                       // K fragments [2 s + key block] of the tile whose scores are computed next
#define SE9_READ_K(slot)                                                                                    \
  do {                                                                                                      \
    const char* t_ = smem + (slot) * k9Slot;                                                                \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                         \
      kf[2 * s] = *reinterpret_cast<const bf16x8*>(t_ + koff[s]);                                           \
      kf[2 * s + 1] = *reinterpret_cast<const bf16x8*>(t_ + koff[s] + 4096);                                \
    }                                                                                                       \
  } while (0)
#define SE9_QK(S0, S1)                                                                                      \
  do {                                                                                                      \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                         \
      S0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2 * s], qf[s], s == 0 ? kZero16 : S0, 0, 0, 0);       \
      S1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2 * s + 1], qf[s], s == 0 ? kZero16 : S1, 0, 0, 0);   \
    }                                                                                                       \
  } while (0)
#define SE9_MASK(S0, S1, kt_)                                                                               \
  do {                                                                                                      \
    if (((kt_) + 1) * kAK > len) {                                                                          \
      const int kbase = (kt_) * kAK + 4 * hh;                                                               \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                      \
        const int key = kbase + (r & 3) + 8 * (r >> 2);                                                     \
        if (key >= len) S0[r] = -INFINITY;                                                                  \
        if (key + 32 >= len) S1[r] = -INFINITY;                                                             \
      }                                                                                                     \
    }                                                                                                       \
  } while (0)
#define SE9_PV(slot)                                                                                        \
  do {                                                                                                      \
    const char* t_ = smem + (slot) * k9Slot;                                                                \
    _Pragma("unroll") for (int kb = 0; kb < 2; ++kb)                                                        \
      _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                         \
        _Pragma("unroll") for (int dblk = 0; dblk < 2; ++dblk) {                                            \
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(                                       \
              (__attribute__((address_space(3))) bf16x4*)(t_ + voff[dblk][0] + kb * 4096 + s * 2048));      \
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(                                       \
              (__attribute__((address_space(3))) bf16x4*)(t_ + voff[dblk][1] + kb * 4096 + s * 2048));      \
          const bf16x8 va = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};                 \
          if (dblk == 0) o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pf[kb][s], o0, 0, 0, 0);          \
          else o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pf[kb][s], o1, 0, 0, 0);                    \
        }                                                                                                   \
  } while (0)
  // exact online softmax of one tile (mhsa.hip): row maximum, deferred rescale of O, P in sa / sb, then packed
#define SE9_EXACT(SA, SB, kt_)                                                                              \
  do {                                                                                                      \
    float mx = fmaxf(SA[0], SB[0]);                                                                         \
    _Pragma("unroll") for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, SA[r]), SB[r]);                     \
    {                                                                                                       \
      const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false); \
      mx = fmaxf(__uint_as_float(sw_[0]), __uint_as_float(sw_[1]));                                         \
    }                                                                                                       \
    float m_new = ((mx - m_run) > kDefer) ? mx : m_run;                                                     \
    if ((kt_) == 0 && mx < -64.f) m_new = mx;                                                               \
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);                                              \
    const float mc = -m_new;                                                                                \
    float rs0 = 0.f, rs1 = 0.f;                                                                             \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                                        \
      const float a0 = __builtin_amdgcn_exp2f(SA[r] + mc);                                                  \
      const float a1 = __builtin_amdgcn_exp2f(SB[r] + mc);                                                  \
      rs0 += a0;                                                                                            \
      rs1 += a1;                                                                                            \
      SA[r] = a0; SB[r] = a1;                                                                               \
    }                                                                                                       \
    l_run = fmaf(l_run, alpha, rs0 + rs1);                                                                  \
    m_run = m_new;                                                                                          \
    if (__any(alpha != 1.0f)) {                                                                             \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }                    \
    }                                                                                                       \
    _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                           \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                       \
        pf[0][s][j] = (__bf16)SA[8 * s + j];                                                                \
        pf[1][s][j] = (__bf16)SB[8 * s + j];                                                                \
      }                                                                                                     \
  } while (0)
  // step e: next fragments, own DMA pieces of tile kt + 3 landed, one barrier
#define SE9_TAIL(kt_)                                                                                       \
  do {                                                                                                      \
    SE9_READ_K(((kt_) + 2) % k9Ring);                                                                            \
    SE9_PIN8(kf);                                                                                           \
    if ((kt_) + 4 < nkt) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
    SE9_BAR();                                                                                              \
  } while (0)

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // tiles 0 .. 3 and the Q rows
  SE9_BAR();
  f32x16 sa0, sa1, sb0, sb1;          // scores of the current (a) and of the next (b) tile; swapped by unrolling the loop by two
  SE9_READ_K(0);
  SE9_PIN8(kf);
  SE9_QK(sa0, sa1);
  SE9_MASK(sa0, sa1, 0);
  SE9_READ_K(1);
  SE9_PIN8(kf);

  bf16x8 pf[2][2];
  int kt = 0;
  bool slow = dscale < 0.f;           // wave-uniform: dscale < 0 = never speculate (A/B switch)
  // ---------------- fast loop: speculative tiles, two per trip (the score buffers trade places)
#define SE9_FAST(SC0, SC1, SN0, SN1)                                                                        \
  {                                                                                                         \
    if (kt + 4 < nkt) SE9_DMA(kt + 4, (kt + 4) % k9Ring);                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    SE9_QK(SN0, SN1);                              /* b: scores of tile kt + 1 (garbage past the last tile: never used) */ \
    float rs0 = 0.f, rs1 = 0.f;                    /* c: speculative probabilities of tile kt */            \
    _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                           \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                       \
        const float a0 = __builtin_amdgcn_exp2f(SC0[8 * s + j]);                                            \
        const float a1 = __builtin_amdgcn_exp2f(SC1[8 * s + j]);                                            \
        rs0 += a0;                                                                                          \
        rs1 += a1;                                                                                          \
        pf[0][s][j] = (__bf16)a0;                                                                           \
        pf[1][s][j] = (__bf16)a1;                                                                           \
      }                                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    const float rs = rs0 + rs1;                                                                             \
    const bool bad = !(rs < 0x1p60f) || (kt == 0 && rs < 0x1p-60f);                                         \
    if (__any(bad)) { slow = true; break; }        /* the scores of tile kt are intact and nothing of it is accumulated: the exact loop redoes it */ \
    l_run += rs;                                                                                            \
    SE9_PV(kt % k9Ring);                                /* d */                                                  \
    __builtin_amdgcn_sched_barrier(0);             /* the next fragments are read AFTER the MFMAs that free their registers */ \
    SE9_MASK(SN0, SN1, kt + 1);                                                                             \
    SE9_TAIL(kt);                                  /* e */                                                  \
    ++kt;                                                                                                   \
  }
  bool odd = false;                   // which buffer pair holds the current tile's scores (false: a)
  while (kt < nkt && !slow) {
    SE9_FAST(sa0, sa1, sb0, sb1)
    odd = true;
    if (kt >= nkt) break;
    SE9_FAST(sb0, sb1, sa0, sa1)
    odd = false;
  }
  // ---------------- exact loop (rare): not pipelined, one score buffer.  Entry state as at the top of a fast tile: scores of tile kt (in b when
  // `odd`), kf = K(kt + 1), vf = V(kt); the fast loop's S(kt + 1) is simply recomputed.
  if (kt < nkt) {
    if (odd) { sa0 = sb0; sa1 = sb1; }
    for (; kt < nkt; ++kt) {
      if (kt + 4 < nkt) SE9_DMA(kt + 4, (kt + 4) % k9Ring);
      SE9_EXACT(sa0, sa1, kt);
      SE9_PV(kt % k9Ring);
      SE9_QK(sa0, sa1);
      SE9_MASK(sa0, sa1, kt + 1);
      SE9_TAIL(kt);
    }
  }

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
}

}  // namespace se

int se_mhsa9_fwd_launch(const uint16_t* qkv, const int32_t* lengths, int B, int T, int heads, uint16_t* ctx, int never_speculate, hipStream_t st) {
  const int H = heads * se::kHD;
  SE_REQUIRE((double)T * 3.0 * H * 2.0 < 2147483648.0, "se_mhsa9: T * 3 H * 2 = %.0f bytes exceeds the 31-bit DMA offset", (double)T * 3.0 * H * 2.0);
  dim3 grid((T + se::k9Q - 1) / se::k9Q, heads, B);
  hipLaunchKernelGGL(se::mhsa9_fwd_kernel, grid, dim3(512), 0, st, qkv, lengths, T, H, ctx, never_speculate ? -1.f : 1.f);
  SE_LAUNCH_CHECK();
  return SE_OK;
}
