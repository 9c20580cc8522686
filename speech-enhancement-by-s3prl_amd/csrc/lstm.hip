// lstm.hip -- recurrent core of the (Bi)LSTM downstream heads (model.py:37-91, the head the
// reference's own scripts train: run_active.sh `--downstream LSTM`, pseudo_noise.yaml:50-53 hidden 256 x 3 layers, bidirectional).
//
// The input projection x W_ih^T + b and every gradient GEMM run on the bf16 GEMM / weight-gradient kernels; what is left is
// the strictly sequential part: T = 1001 steps of gates = xproj_t + W_hh h_{t-1} per utterance and direction.  It is latency
// bound (262 k MAC per step), so a step touches no memory beyond LDS, and the matrix-vector product runs on the MATRIX pipe:
//   one 512-thread workgroup (8 waves) = one (utterance, direction).  W_hh (512 KiB as bf16 = the CU's whole register file)
//   stays ON CHIP for all T steps as MFMA A-fragments: per lane 64 fragments of 8 bf16, 45-46 in registers + 18-19 in LDS (144-152 KiB).
//   The vector (h_{t-1}, or the gate gradients in the backward) is the B operand with all 16 columns equal, so 15/16 of
//   v_mfma_f32_16x16x32_bf16 is wasted -- and it is still twice as fast as v_dot2c_f32_bf16 (half rate on gfx950: 128 of
//   them per thread made the first version VALU-bound at 3.1 us per step): 64 MFMAs per wave = 2 048 matrix-pipe cycles per step.
//   Wave w owns hidden units 32 w .. 32 w + 31 with ALL FOUR gates (row tiles g x 256 + 32 w + 16 half), so the accumulators of
//   one lane hold the four gate pre-activations of the same units: the cell update is lane-local (lanes with (lane & 15) < 8
//   take one unit each), c_t lives in a register, h_t goes to a double-buffered LDS vector: ONE barrier per step.
// Backward (BPTT) mirrors it with W_hh^T (wave w owns dh_{t-1}[32 w ..]); the gate-gradient vector (1024 bf16) is the B operand.
// dgates are written for the batched GEMMs that follow (dW_ih, dW_hh, dx).
// Padded frames are processed like real ones, as nn.LSTM on the reference's padded batches does (no packing in model.py).
#include "common.h"
#include "bf16.h"

namespace se {

// 4-B-per-lane LDS-DMA by inline asm (M0 = the wave-uniform LDS destination; lane l lands at +4 l).  NOT __builtin_amdgcn_global_load_lds: through the
// builtin the compiler orders every later LDS read behind the DMA with s_waitcnt vmcnt(0), and __syncthreads() carries a vmcnt(0) of its own -- both
// kernels below waited out the full latency of the piece they had just issued, once per time step (found in round 4 with tools/isa_wait_lint.py after
// the same pattern in wgrad.hip).  The step barrier is therefore SE_L_BAR: LDS writes drained (lgkmcnt), global stores and DMA left in flight.
#define SE_L_DMA4(gptr_, lds_u32_)                                                                                          \
  do {                                                                                                                      \
    uint32_t keep_;                                                                                                         \
    const uint32_t l_ = __builtin_amdgcn_readfirstlane(lds_u32_);      /* (step % 3) is computed on the vector unit */             \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"         \
                 : "=&s"(keep_) : "v"(gptr_), "s"(l_) : "memory");                                                          \
  } while (0)
#define SE_L_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

constexpr int kLH = 256, kLG = 1024;
constexpr int kLFrag = 64;                                               // A-fragments per lane
constexpr int kLRegF = 46, kLRegB = 47;                                  // of them in registers (forward / backward); the rest in LDS
constexpr int kLThreads = 512;
constexpr int lds_w_bytes(int reg) { return 8 * (kLFrag - reg) * 1024; }   // 8 waves x (64 - reg) fragments x 1 KiB

// v_exp_f32 + v_rcp_f32 (1 ulp): the IEEE division sequence was a third of the gate phase's instructions
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return fmaf(2.f, __builtin_amdgcn_rcpf(1.f + __expf(-2.f * x)), -1.f); }

// accumulator element of this lane's unit: tile `half` = li >> 2, register li & 3   (li = lane & 15 < 8)
__device__ __forceinline__ float sel_unit(const f32x4& a0, const f32x4& a1, int li) {
  const f32x4 a = (li & 4) ? a1 : a0;
  const float lo = (li & 1) ? a[1] : a[0], hi = (li & 1) ? a[3] : a[2];
  return (li & 2) ? hi : lo;
}

// Fragment (ks, t8) = rows (t8 >> 1) * 256 + 32 wave + 16 (t8 & 1) + li, k = 32 ks + 8 lg .. + 7.  LDS-resident: t8 = 6, 7 for every ks and
// t8 = 5 for ks < 2 (18 fragments); each is read into one of three register slots right after the MFMA that consumed the slot's
// previous content, i.e. eight MFMAs (>= 128 cycles) before its own use -- the weights do not depend on the step, so the chain runs
// straight across the step boundary.  (Left to the compiler, every LDS fragment became "ds_read, s_waitcnt, v_mfma": 19 exposed LDS
// latencies per step.)
__device__ __forceinline__ constexpr bool lf_in_lds(int ks, int t8) { return t8 >= 6 || (t8 == 5 && ks < 2); }
__device__ __forceinline__ constexpr int lf_lds_index(int ks, int t8) { return t8 == 5 ? 16 + ks : 2 * ks + (t8 - 6); }     // 0..17
__device__ __forceinline__ constexpr int lf_reg_index(int ks, int t8) {                                                  // 0..45
  int n = 0;
  for (int k = 0; k < 8; ++k)
    for (int t = 0; t < 8; ++t) {
      if (k == ks && t == t8) return n;
      if (!lf_in_lds(k, t)) ++n;
    }
  return n;
}

// w_hh: [ndir][1024][256] bf16 row-major (nn.LSTM's weight_hh, gate order i, f, g, o)
__global__ __launch_bounds__(kLThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void lstm_fwd_kernel(
    const uint16_t* __restrict__ w_hh, const float* __restrict__ xproj, int B, int T, int ndir, uint16_t* __restrict__ h_out,
    float* __restrict__ gates_out, float* __restrict__ c_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int kLReg = kLRegF, kLLds = kLFrag - kLReg, kLWBytes = lds_w_bytes(kLReg);
  const int tid = threadIdx.x, lane = tid & 63, lg = lane >> 4, li = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x, dir = blockIdx.y;
  const bool reverse = dir == 1;
  const uint16_t* W = w_hh + (size_t)dir * kLG * kLH;
  char* wl = smem + wave * (kLLds * 1024) + lane * 16;
  uint16_t* hs = reinterpret_cast<uint16_t*>(smem + kLWBytes);             // [2][256] bf16: h_{t-1} / h_t
  float* xring = reinterpret_cast<float*>(smem + kLWBytes + 2 * kLH * 2);  // [3][1024] fp32: input projections of steps s .. s+2
  // xproj rows arrive by LDS-DMA three steps ahead (no VGPRs, no exposed HBM latency: as register prefetches one step ahead they
  // cost 0.4 us per step).  Wave w fetches floats 128 w .. 128 w + 127 of a row with two 256-B pieces.
  typedef __attribute__((address_space(3))) void* ldsp_t;
  const size_t seq = ((size_t)dir * B + b) * T;
  const float* xrow0 = xproj + seq * kLG + 128 * wave + lane;
  const uint32_t xring_u = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(ldsp_t)xring + (uint32_t)wave * 512u);
#define SE_L_DMA(step)                                                                                                     \
  do {                                                                                                                      \
    const int t_ = reverse ? T - 1 - (step) : (step);                                                                       \
    const uint32_t d_ = xring_u + (uint32_t)(((step) % 3) * 4096);                                                          \
    SE_L_DMA4(xrow0 + (size_t)t_ * kLG, d_);                                                                                \
    SE_L_DMA4(xrow0 + (size_t)t_ * kLG + 64, d_ + 256u);                                                                    \
  } while (0)

  bf16x8 wreg[kLReg];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks)
#pragma unroll
    for (int t8 = 0; t8 < 8; ++t8) {
      const int row = (t8 >> 1) * kLH + 32 * wave + 16 * (t8 & 1) + li;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(W + (size_t)row * kLH + 32 * ks + 8 * lg);
      if (lf_in_lds(ks, t8)) *reinterpret_cast<bf16x8*>(wl + lf_lds_index(ks, t8) * 1024) = v;
      else wreg[lf_reg_index(ks, t8)] = v;
    }
  if (tid < kLH) hs[tid] = 0;

  const bool act = li < 8;
  const int u = 32 * wave + 16 * (li >> 2) + 4 * lg + (li & 3);              // this lane's hidden unit (act lanes)
  float c = 0.f;
  const f32x4 kZero4 = {0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the weight loads above (LDS part written below their wait anyway)
  SE_L_DMA(0);
  if (T > 1) SE_L_DMA(1);
  if (T > 2) SE_L_DMA(2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // the three slots start with the LDS fragments of ks = 0 (t8 = 5, 6, 7)
  bf16x8 slot[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) slot[i] = *reinterpret_cast<const bf16x8*>(wl + lf_lds_index(0, 5 + i) * 1024);
  for (int s = 0; s < T; ++s) {
    const int t = reverse ? T - 1 - s : s;
    const uint16_t* hcur = hs + (s & 1) * kLH + 8 * lg;
    f32x4 acc[8];
    bf16x8 hb[2];
    hb[0] = *reinterpret_cast<const bf16x8*>(hcur);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (ks < 7) hb[(ks + 1) & 1] = *reinterpret_cast<const bf16x8*>(hcur + 32 * (ks + 1));
#pragma unroll
      for (int t8 = 0; t8 < 8; ++t8) {
        const bool in_lds = lf_in_lds(ks, t8);
        const bf16x8 a = in_lds ? slot[t8 - 5] : wreg[in_lds ? 0 : lf_reg_index(ks, t8)];
        acc[t8] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, hb[ks & 1], ks == 0 ? kZero4 : acc[t8], 0, 0, 0);
        if (t8 >= 4) __builtin_amdgcn_sched_barrier(0);      // pin: MFMA, then the refill read of the slot it consumed
        if (in_lds || (t8 == 5 && ks == 7)) {
          // refill the slot: the same t8 of the next ks (of the next step after ks = 7); t8 = 5 only has LDS fragments for ks < 3
          const int nks = (ks + 1) & 7;
          if (lf_in_lds(nks, t8)) slot[t8 - 5] = *reinterpret_cast<const bf16x8*>(wl + lf_lds_index(nks, t8) * 1024);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // row s+3 goes to the ring slot of row s (read during the previous step's gate phase, before its barrier); then everything older
    // than these two pieces must have landed: row s+1 (issued two steps ago) and the stores of the previous step
    if (s > 0 && s + 2 < T) {
      SE_L_DMA(s + 2);
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (act) {
      const float* xr = xring + (s % 3) * kLG + u;
      const float gi = sigmoidf_(sel_unit(acc[0], acc[1], li) + xr[0]);
      const float gf = sigmoidf_(sel_unit(acc[2], acc[3], li) + xr[kLH]);
      const float gg = tanhf_(sel_unit(acc[4], acc[5], li) + xr[2 * kLH]);
      const float go = sigmoidf_(sel_unit(acc[6], acc[7], li) + xr[3 * kLH]);
      c = gf * c + gi * gg;
      const float h = go * tanhf_(c);
      const size_t row = seq + t;
      float* gp = gates_out + row * kLG + u;
      gp[0] = gi; gp[kLH] = gf; gp[2 * kLH] = gg; gp[3 * kLH] = go;
      c_out[row * kLH + u] = c;
      const uint16_t hb16 = f2bf(h);
      hs[((s + 1) & 1) * kLH + u] = hb16;
      h_out[((size_t)b * T + t) * (ndir * kLH) + dir * kLH + u] = hb16;
    }
    SE_L_BAR();            // LDS writes of this step visible; stores and the DMA just issued stay in flight
  }
}

// Backward fragments (js, half): row k = 32 wave + 16 half + li of W_hh^T, gate index j = 32 js + 8 lg .. + 7.  LDS-resident: both halves of
// js = 3, 7, .., 31 (two register slots, refilled right after use: eight MFMAs ahead) and (js = 1, half = 0) (read at the top of the step).
__device__ __forceinline__ constexpr bool lb_in_lds(int js, int half) { return (js & 3) == 3 || (js == 1 && half == 0); }
__device__ __forceinline__ constexpr int lb_lds_index(int js, int half) { return (js & 3) == 3 ? 2 * (js >> 2) + half : 16; }      // 0..16
__device__ __forceinline__ constexpr int lb_reg_index(int js, int half) {                                                        // 0..46
  int n = 0;
  for (int k = 0; k < 32; ++k)
    for (int h = 0; h < 2; ++h) {
      if (k == js && h == half) return n;
      if (!lb_in_lds(k, h)) ++n;
    }
  return n;
}

// w_hh_t: [ndir][256][1024] bf16 row-major = W_hh^T
__global__ __launch_bounds__(kLThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void lstm_bwd_kernel(
    const uint16_t* __restrict__ w_hh_t, const float* __restrict__ gates, const float* __restrict__ c_saved, const float* __restrict__ dh_out,
    int ld_dh, int B, int T, int ndir, uint16_t* __restrict__ dgates_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int kLReg = kLRegB, kLLds = kLFrag - kLReg, kLWBytes = lds_w_bytes(kLReg);
  const int tid = threadIdx.x, lane = tid & 63, lg = lane >> 4, li = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x, dir = blockIdx.y;
  const bool reverse = dir == 1;
  const uint16_t* W = w_hh_t + (size_t)dir * kLH * kLG;
  char* wl = smem + wave * (kLLds * 1024) + lane * 16;
  uint16_t* dgs = reinterpret_cast<uint16_t*>(smem + kLWBytes);            // [2][1024] bf16 gate gradients of the current step
  // per-step operands (post-activation gates 4 KiB, c_{t-1} 1 KiB, dL/dh_t 1 KiB) arrive by LDS-DMA two steps ahead in a ring of three
  // rows; a row is complete and visible to every wave one barrier after its issuers' counted wait, i.e. from the step after that wait
  // on -- which is why a row carries c_{t-1} (c_t is the previous step's c_{t-1}, kept in a register)
  char* ring = smem + kLWBytes + 2 * kLG * 2;                              // [3][6144]
  typedef __attribute__((address_space(3))) void* ldsp_t;
  const size_t seq = ((size_t)dir * B + b) * T;
  // wave w: gate floats 128 w .. + 127 (two 256-B pieces) and one 256-B piece of c (waves 0-3) or dh (waves 4-7)
  const float* g_src = gates + seq * kLG + 128 * wave + lane;
  const float* c_src = c_saved + seq * kLH + 64 * (wave & 3) + lane;
  const float* d_src = dh_out + (size_t)b * T * ld_dh + dir * kLH + 64 * (wave & 3) + lane;
  const uint32_t ring_u = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(ldsp_t)ring);
#define SE_LB_DMA(step)                                                                                                     \
  do {                                                                                                                      \
    const int t_ = reverse ? (step) : T - 1 - (step);                                                                       \
    const uint32_t r_ = ring_u + (uint32_t)(((step) % 3) * 6144);                                                           \
    SE_L_DMA4(g_src + (size_t)t_ * kLG, r_ + (uint32_t)wave * 512u);                                                        \
    SE_L_DMA4(g_src + (size_t)t_ * kLG + 64, r_ + (uint32_t)wave * 512u + 256u);                                            \
    const int sn_ = (step) + 1 < T ? (step) + 1 : (step);          /* the ring row of step s carries c of step s+1 = c_{t-1} */     \
    const int tn_ = reverse ? sn_ : T - 1 - sn_;                                                                            \
    if (wave < 4) SE_L_DMA4(c_src + (size_t)tn_ * kLH, r_ + 4096u + (uint32_t)wave * 256u);                                 \
    else SE_L_DMA4(d_src + (size_t)t_ * ld_dh, r_ + 5120u + (uint32_t)(wave - 4) * 256u);                                   \
  } while (0)

  bf16x8 wreg[kLReg];
#pragma unroll
  for (int js = 0; js < 32; ++js)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int row = 32 * wave + 16 * half + li;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(W + (size_t)row * kLG + 32 * js + 8 * lg);
      if (lb_in_lds(js, half)) *reinterpret_cast<bf16x8*>(wl + lb_lds_index(js, half) * 1024) = v;
      else wreg[lb_reg_index(js, half)] = v;
    }

  const bool act = li < 8;
  const int u = 32 * wave + 16 * (li >> 2) + 4 * lg + (li & 3);
  float dh_rec = 0.f, dc_rec = 0.f;
  float ct = act ? c_saved[(seq + (reverse ? 0 : T - 1)) * kLH + u] : 0.f;     // c_t of the first step
  const f32x4 kZero4 = {0.f, 0.f, 0.f, 0.f};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  SE_LB_DMA(0);
  if (T > 1) SE_LB_DMA(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  bf16x8 slot[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) slot[i] = *reinterpret_cast<const bf16x8*>(wl + lb_lds_index(3, i) * 1024);
  for (int s = 0; s < T; ++s) {
    const int t = reverse ? s : T - 1 - s;                 // the forward visited t last-to-first in this order
    uint16_t* dcur = dgs + (s & 1) * kLG;
    // row s+2 goes to the slot of row s-1 (last read before the previous barrier); everything older than its three pieces has landed
    // after the wait: row s+1 and the stores of the previous step
    if (s + 2 < T) {
      SE_LB_DMA(s + 2);
      asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const bf16x8 odd = *reinterpret_cast<const bf16x8*>(wl + lb_lds_index(1, 0) * 1024);      // the 17th LDS fragment, used at js = 1
    if (act) {
      const float* row = reinterpret_cast<const float*>(ring + (s % 3) * 6144);
      const float gi = row[u], gf = row[kLH + u], gg = row[2 * kLH + u], go = row[3 * kLH + u];
      const float cprev = (s + 1 < T) ? row[kLG + u] : 0.f;
      const float dh = row[kLG + kLH + u] + dh_rec;
      const float tc = tanhf_(ct);
      const float d_o = dh * tc * go * (1.f - go);
      const float dc = dh * go * (1.f - tc * tc) + dc_rec;
      const float d_i = dc * gg * gi * (1.f - gi);
      const float d_f = dc * cprev * gf * (1.f - gf);
      const float d_g = dc * gi * (1.f - gg * gg);
      dc_rec = dc * gf;
      ct = cprev;
      const uint16_t bi = f2bf(d_i), bff = f2bf(d_f), bg = f2bf(d_g), bo = f2bf(d_o);
      dcur[u] = bi; dcur[kLH + u] = bff; dcur[2 * kLH + u] = bg; dcur[3 * kLH + u] = bo;
      uint16_t* op = dgates_out + (seq + t) * kLG + u;
      op[0] = bi; op[kLH] = bff; op[2 * kLH] = bg; op[3 * kLH] = bo;
    }
    SE_L_BAR();            // LDS writes of this step visible; stores and the DMA just issued stay in flight
    // dh_{t-1}[k] = sum_j W_hh[j][k] dgates[j]: two partial accumulators per row tile (same accumulator every fourth MFMA)
    f32x4 acc[2][2];
    const uint16_t* dv = dcur + 8 * lg;
    bf16x8 db[2];
    db[0] = *reinterpret_cast<const bf16x8*>(dv);
#pragma unroll
    for (int js = 0; js < 32; ++js) {
      if (js < 31) db[(js + 1) & 1] = *reinterpret_cast<const bf16x8*>(dv + 32 * (js + 1));
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const bool in_slot = (js & 3) == 3;
        const bool is_odd = js == 1 && half == 0;
        const bf16x8 a = in_slot ? slot[half] : is_odd ? odd : wreg[(in_slot || is_odd) ? 0 : lb_reg_index(js, half)];
        acc[half][js & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, db[js & 1], js < 2 ? kZero4 : acc[half][js & 1], 0, 0, 0);
        if (in_slot) {
          __builtin_amdgcn_sched_barrier(0);
          slot[half] = *reinterpret_cast<const bf16x8*>(wl + lb_lds_index((js + 4) & 31, half) * 1024);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if ((js & 3) == 2) __builtin_amdgcn_sched_barrier(0);
    }
    const f32x4 a0 = acc[0][0] + acc[0][1];
    const f32x4 a1 = acc[1][0] + acc[1][1];
    dh_rec = sel_unit(a0, a1, li);
  }
}

constexpr int kLFwdLds = lds_w_bytes(kLRegF) + 2 * kLH * 2 + 3 * kLG * 4;
constexpr int kLBwdLds = lds_w_bytes(kLRegB) + 2 * kLG * 2 + 3 * 6144;

}  // namespace se

extern "C" int se_lstm_fwd_bf16(const uint16_t* w_hh, const float* xproj, int B, int T, int ndir, uint16_t* h_out, float* gates_out,
                                float* c_out, void* stream) {
  SE_REQUIRE(w_hh && xproj && h_out && gates_out && c_out, "se_lstm_fwd_bf16: null argument");
  SE_REQUIRE(B > 0 && B <= 65535 && T > 0 && (ndir == 1 || ndir == 2), "se_lstm_fwd_bf16: bad shape");
  static bool attr_set = false;
  if (!attr_set) {
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::lstm_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, se::kLFwdLds));
    attr_set = true;
  }
  hipLaunchKernelGGL(se::lstm_fwd_kernel, dim3(B, ndir), dim3(se::kLThreads), se::kLFwdLds, se::as_stream(stream), w_hh, xproj, B, T, ndir, h_out,
                     gates_out, c_out);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_lstm_bwd_bf16(const uint16_t* w_hh_t, const float* gates, const float* c_saved, const float* dh_out, int ld_dh, int B,
                                int T, int ndir, uint16_t* dgates_out, void* stream) {
  SE_REQUIRE(w_hh_t && gates && c_saved && dh_out && dgates_out, "se_lstm_bwd_bf16: null argument");
  SE_REQUIRE(B > 0 && B <= 65535 && T > 0 && (ndir == 1 || ndir == 2) && ld_dh >= ndir * se::kLH, "se_lstm_bwd_bf16: bad shape");
  static bool attr_set = false;
  if (!attr_set) {
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::lstm_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, se::kLBwdLds));
    attr_set = true;
  }
  hipLaunchKernelGGL(se::lstm_bwd_kernel, dim3(B, ndir), dim3(se::kLThreads), se::kLBwdLds, se::as_stream(stream), w_hh_t, gates, c_saved, dh_out,
                     ld_dh, B, T, ndir, dgates_out);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

// bias gradient of a bf16 matrix: out[c] = sum_r x[r][c]   (cols, ld multiples of 8)
namespace se { int launch_colsum_bf16(const uint16_t* x, int rows, int cols, int ld, float* out, hipStream_t st); }
extern "C" int se_colsum_bf16(const uint16_t* x, int rows, int cols, int ld, float* out, void* stream) {
  SE_REQUIRE(x && out && rows > 0 && cols > 0 && ld >= cols, "se_colsum_bf16: bad argument");
  return se::launch_colsum_bf16(x, rows, cols, ld, out, se::as_stream(stream));
}
