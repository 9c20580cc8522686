// mhsa2.hip -- the inference flash MHSA forward (pre-scaled queries, mhsa.hip PRE = 1) with QB = 2 query blocks per wave.
//
// Why.  The compile-time ablation of mhsa.hip (-DSE_MHSA_ABL, tools/mhsa_ablate.sh; B = 32, T = 1001, 12 heads) put the launch at
//   K / V staging 27 %, MFMAs 41 %, LDS fragment reads 19 %, exponentials 11 %, barrier 5 %, everything else 34 % of its time -- the parts ADD
// (sum 137 %): the matrix pipe does not run under the other work, and more than half of the launch is per-KEY-TILE work that every wave
// repeats for only 32 queries.  Here a wave owns 64 queries (two 32-query blocks, workgroup = 4 waves = 256 queries): every staged K / V tile,
// every K and V^T fragment read and every barrier now serves twice the MFMA work, and the two blocks are independent instruction streams
// inside one wave (the only place where this chip overlaps matrix and vector work well: tools/micro/valu_rate.hip, one wave per SIMD).
// 2 waves per SIMD (<= 256 registers), grid = ceil(T / 256) x heads x B.
// Data layout, fragments, LDS swizzle, speculative max-free softmax with exact online-softmax fallback: mhsa.hip.
#include <stdlib.h>
#include "common.h"
#include "prof.h"
#include "bf16.h"
#include "mhsa_tile.h"

namespace se {

constexpr int kQ2 = 2;                 // query blocks per wave
constexpr int kAQ2 = 4 * 32 * kQ2;     // queries per workgroup

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void mhsa2_fwd_kernel(
    const uint16_t* __restrict__ qkv, const int32_t* __restrict__ lengths, int T, int H, uint16_t* __restrict__ ctx, int never_speculate) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * kAK * kHD * 2];   // 2 buffers x (K, V) x 8 KiB = 32 KiB

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  // XCD-aware work mapping (mhsa.hip): all query tiles of one (utterance, head) on ONE XCD, consecutive in its dispatch order
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
  const int q0 = qt * kAQ2 + wave * (32 * kQ2);
  const int ld = 3 * H;
  const int len = lengths ? min(max(lengths[b], 1), T) : T;
  const int nkt = (len + kAK - 1) / kAK;
  const uint16_t* base = qkv + (size_t)b * T * ld + head * kHD;

  // ---- Q fragments (B operand of S^T = K Q^T): lane -> query row q0 + 32 qb + l31, d = 16 s + 8 hh .. +7
  bf16x8 qf[kQ2][4];
#pragma unroll
  for (int qb = 0; qb < kQ2; ++qb) {
    const int q = min(q0 + 32 * qb + l31, T - 1);
    const uint16_t* qp = base + (size_t)q * ld + 8 * hh;
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[qb][s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
  }

  // ---- staging by LDS-DMA (no VGPR round trip: the 64-query wave needs its registers): wave w brings rows [16 w, 16 w + 16) of the K and of the V
  //      tile, two 1-KiB pieces (8 rows x 128 B) each; lane l of a piece writes slot l & 7 of row l >> 3, i.e. it FETCHES chunk (l & 7) ^ f(row)
  typedef __attribute__((address_space(1))) const void* glb_a_t;
  typedef __attribute__((address_space(3))) void* lds_a_t;
  const int drow = wave * 16 + (lane >> 3);
  const int dch0 = ((lane & 7) ^ (kv_off(drow, 0) >> 4 & 7)) * 8, dch1 = ((lane & 7) ^ (kv_off(drow + 8, 0) >> 4 & 7)) * 8;
  auto stage = [&](int kt, int buf) {
    const size_t r0 = (size_t)min(kt * kAK + drow, T - 1) * ld;
    const size_t r1 = (size_t)min(kt * kAK + drow + 8, T - 1) * ld;
    char* k_w = smem + buf * 16384 + wave * 2048;
    __builtin_amdgcn_global_load_lds((glb_a_t)(base + H + r0 + dch0), (lds_a_t)(k_w), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_a_t)(base + H + r1 + dch1), (lds_a_t)(k_w + 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_a_t)(base + 2 * H + r0 + dch0), (lds_a_t)(k_w + 8192), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_a_t)(base + 2 * H + r1 + dch1), (lds_a_t)(k_w + 8192 + 1024), 16, 0, 0);
  };

  const f32x16 kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 o0[kQ2], o1[kQ2];            // O^T d-blocks 0 / 1 of each query block: col = query (lane & 31), row = d
#pragma unroll
  for (int qb = 0; qb < kQ2; ++qb)
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[qb][r] = 0.f; o1[qb][r] = 0.f; }
  float m_run[kQ2], l_run[kQ2];
#pragma unroll
  for (int qb = 0; qb < kQ2; ++qb) { m_run[qb] = 0.f; l_run[qb] = 0.f; }
  bool slow = never_speculate != 0;   // wave-uniform: the speculative (no row maximum) path failed once
  constexpr float kDefer = 8.f;

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

  stage(0, 0);
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nkt) stage(kt + 1, cur ^ 1);      // lands under this tile's work; the barrier at the end waits for it (vmcnt(0))
    const char* t_s = smem + cur * 16384;
    // ---- S^T = K Q^T for both query blocks: every K fragment read feeds four MFMAs
    f32x16 s0[kQ2], s1[kQ2];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 ka = *reinterpret_cast<const bf16x8*>(t_s + koff[s]);
      const bf16x8 kb_ = *reinterpret_cast<const bf16x8*>(t_s + koff[s] + 4096);
#pragma unroll
      for (int qb = 0; qb < kQ2; ++qb) {
        s0[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[qb][s], s == 0 ? kZero16 : s0[qb], 0, 0, 0);
        s1[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb_, qf[qb][s], s == 0 ? kZero16 : s1[qb], 0, 0, 0);
      }
    }
    if ((kt + 1) * kAK > len) {
      const int kbase = kt * kAK + 4 * hh;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kbase + (r & 3) + 8 * (r >> 2);
#pragma unroll
        for (int qb = 0; qb < kQ2; ++qb) {
          if (key >= len) s0[qb][r] = -INFINITY;
          if (key + 32 >= len) s1[qb][r] = -INFINITY;
        }
      }
    }
    // ---- softmax per query block (independent streams)
    bf16x8 pf[kQ2][2][2];
    bool spec_ok = false;
    if (!slow) {
      float rs[kQ2];
#pragma unroll
      for (int qb = 0; qb < kQ2; ++qb) {
        float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float a0 = __builtin_amdgcn_exp2f(s0[qb][8 * s + j]);
            const float a1 = __builtin_amdgcn_exp2f(s1[qb][8 * s + j]);
            rs0 += a0;
            rs1 += a1;
            pf[qb][0][s][j] = (__bf16)a0;
            pf[qb][1][s][j] = (__bf16)a1;
          }
        rs[qb] = rs0 + rs1;
      }
      bool bad = false;
#pragma unroll
      for (int qb = 0; qb < kQ2; ++qb) bad = bad || !(rs[qb] < 0x1p60f) || (kt == 0 && rs[qb] < 0x1p-60f);
      if (!__any(bad)) {
#pragma unroll
        for (int qb = 0; qb < kQ2; ++qb) l_run[qb] += rs[qb];
        spec_ok = true;
      } else {
        slow = true;
      }
    }
    if (!spec_ok) {
#pragma unroll
      for (int qb = 0; qb < kQ2; ++qb) {
        float mx = fmaxf(s0[qb][0], s1[qb][0]);
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s0[qb][r]), s1[qb][r]);
        {
          const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
          mx = fmaxf(__uint_as_float(sw_[0]), __uint_as_float(sw_[1]));
        }
        float m_new = ((mx - m_run[qb]) > kDefer) ? mx : m_run[qb];
        if (kt == 0 && mx < -64.f) m_new = mx;        // a first tile far below the initial reference 0 (later tiles cannot matter)
        const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_new);
        const float mc = -m_new;
        float rs0 = 0.f, rs1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float a0 = __builtin_amdgcn_exp2f(s0[qb][r] + mc);
          const float a1 = __builtin_amdgcn_exp2f(s1[qb][r] + mc);
          rs0 += a0;
          rs1 += a1;
          s0[qb][r] = a0;
          s1[qb][r] = a1;
        }
        l_run[qb] = fmaf(l_run[qb], alpha, rs0 + rs1);
        m_run[qb] = m_new;
        if (__any(alpha != 1.0f)) {
#pragma unroll
          for (int r = 0; r < 16; ++r) { o0[qb][r] *= alpha; o1[qb][r] *= alpha; }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            pf[qb][0][s][j] = (__bf16)s0[qb][8 * s + j];
            pf[qb][1][s][j] = (__bf16)s1[qb][8 * s + j];
          }
      }
    }
    // ---- O^T += V^T P^T: every V^T fragment read feeds both query blocks
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int dblk = 0; dblk < 2; ++dblk) {
          const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
              (__attribute__((address_space(3))) bf16x4*)(t_s + voff[dblk][0] + kb * 4096 + s * 2048));
          const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
              (__attribute__((address_space(3))) bf16x4*)(t_s + voff[dblk][1] + kb * 4096 + s * 2048));
          const bf16x8 va = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
          for (int qb = 0; qb < kQ2; ++qb) {
            if (dblk == 0) o0[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pf[qb][kb][s], o0[qb], 0, 0, 0);
            else o1[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, pf[qb][kb][s], o1[qb], 0, 0, 0);
          }
        }
      }
    __syncthreads();
  }

  // ---- epilogue: O / l ; lane holds query q0 + 32 qb + l31, d = 32 dblk + (r&3) + 8 (r>>2) + 4 hh
#pragma unroll
  for (int qb = 0; qb < kQ2; ++qb) {
    const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32);
    const float inv = 1.0f / l_tot;
    const int q = q0 + 32 * qb + l31;
    if (q < T) {
      uint16_t* op = ctx + ((size_t)b * T + q) * H + head * kHD + 4 * hh;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        uint2 w0 = make_uint2(pack_bf16x2(o0[qb][4 * g] * inv, o0[qb][4 * g + 1] * inv), pack_bf16x2(o0[qb][4 * g + 2] * inv, o0[qb][4 * g + 3] * inv));
        uint2 w1 = make_uint2(pack_bf16x2(o1[qb][4 * g] * inv, o1[qb][4 * g + 1] * inv), pack_bf16x2(o1[qb][4 * g + 2] * inv, o1[qb][4 * g + 3] * inv));
        *reinterpret_cast<uint2*>(op + 8 * g) = w0;
        *reinterpret_cast<uint2*>(op + 32 + 8 * g) = w1;
      }
    }
  }
}

}  // namespace se

int se_mhsa2_fwd_launch(const uint16_t* qkv, const int32_t* lengths, int B, int T, int heads, uint16_t* ctx, int never_speculate, hipStream_t st) {
  const int H = heads * se::kHD;
  dim3 grid((T + se::kAQ2 - 1) / se::kAQ2, heads, B);
  hipLaunchKernelGGL(se::mhsa2_fwd_kernel, grid, dim3(256), 0, st, qkv, lengths, T, H, ctx, never_speculate);
  SE_LAUNCH_CHECK();
  return SE_OK;
}
