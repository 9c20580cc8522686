// lstm.hip -- recurrent core of the (Bi)LSTM downstream heads (model.py:37-91, the head the
// reference's own scripts train: run_active.sh `--downstream LSTM`, pseudo_noise.yaml:50-53 hidden 256 x 3 layers, bidirectional).
//
// The input projection x W_ih^T + b and every gradient GEMM run on the bf16 GEMM / weight-gradient kernels; what is left is
// the strictly sequential part: T = 1001 steps of gates = xproj_t + W_hh h_{t-1} per utterance and direction.  It is latency
// bound (262 k MAC per step), so the design goal is that a step touches no memory beyond LDS:
//   one 1024-thread workgroup = one (utterance, direction); thread j owns gate column j of W_hh (4H = 1024 columns of H = 256)
//   and keeps it ON CHIP for all T steps: 92 bf16 pairs in registers + 36 pairs in LDS (144 KiB) -- 1024 threads x 128 VGPRs is
//   the whole register file of a CU, so the split is what makes the 512 KiB matrix resident; h_{t-1} (256 bf16) lives in LDS and
//   is read as broadcasts; v_dot2c_f32_bf16 accumulates in fp32; cell state fp32 in registers of threads 0..255.
//   Per step: 128 dot2 per thread, 2 barriers, one coalesced 4 KiB read of xproj_t (prefetched) and the stores of what the
//   backward needs (post-activation gates, c_t, h_t).
// Backward (BPTT) mirrors it with the contraction over the gate index: thread (k, q) owns W_hh[256 q .. 256 q + 255][k] the same
// way, the four partial sums meet in LDS.  dgates are written for the batched GEMMs that follow (dW_ih, dW_hh, dx).
// Padded frames are processed like real ones, as nn.LSTM on the reference's padded batches does (no packing in model.py).
#include "common.h"
#include "bf16.h"

namespace se {

constexpr int kLH = 256, kLG = 1024, kLPairs = 128, kLLds = 36, kLReg = kLPairs - kLLds;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

__device__ __forceinline__ float dot2(uint32_t w, uint32_t h, float acc) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w), __builtin_bit_cast(bf16x2_t, h), acc, false);
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return 2.f / (1.f + __expf(-2.f * x)) - 1.f; }

// wp: [ndir][128 kk][1024 j] u32 = (W_hh[j][2 kk], W_hh[j][2 kk + 1]) as bf16 pairs
__global__ __launch_bounds__(1024) void lstm_fwd_kernel(const uint32_t* __restrict__ wp, const float* __restrict__ xproj, int B, int T, int ndir,
                                                        uint16_t* __restrict__ h_out, float* __restrict__ gates_out, float* __restrict__ c_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* wl = reinterpret_cast<uint32_t*>(smem);                       // [kLLds][1024]
  float* gs = reinterpret_cast<float*>(smem + kLLds * kLG * 4);            // [1024]
  uint32_t* hs = reinterpret_cast<uint32_t*>(smem + kLLds * kLG * 4 + kLG * 4);   // [128] bf16 pairs of h_{t-1}
  const int j = threadIdx.x, b = blockIdx.x, dir = blockIdx.y;
  const bool reverse = dir == 1;
  const uint32_t* w = wp + (size_t)dir * kLPairs * kLG;
  uint32_t wr[kLReg];
#pragma unroll
  for (int r = 0; r < kLReg; ++r) wr[r] = w[(size_t)(kLLds + r) * kLG + j];
#pragma unroll 4
  for (int kk = 0; kk < kLLds; ++kk) wl[kk * kLG + j] = w[(size_t)kk * kLG + j];
  if (j < kLPairs) hs[j] = 0u;
  float c = 0.f;
  const size_t seq = ((size_t)dir * B + b) * T;
  const float* xp = xproj + seq * kLG + j;
  float x_next = xp[(size_t)(reverse ? T - 1 : 0) * kLG];
  __syncthreads();
  for (int s = 0; s < T; ++s) {
    const int t = reverse ? T - 1 - s : s;
    float acc = x_next;
    if (s + 1 < T) x_next = xp[(size_t)(reverse ? t - 1 : t + 1) * kLG];
#pragma unroll
    for (int q = 0; q < kLLds / 4; ++q) {
      const uint4 hv = reinterpret_cast<const uint4*>(hs)[q];
      acc = dot2(wl[(4 * q + 0) * kLG + j], hv.x, acc);
      acc = dot2(wl[(4 * q + 1) * kLG + j], hv.y, acc);
      acc = dot2(wl[(4 * q + 2) * kLG + j], hv.z, acc);
      acc = dot2(wl[(4 * q + 3) * kLG + j], hv.w, acc);
    }
#pragma unroll
    for (int q = 0; q < kLReg / 4; ++q) {
      const uint4 hv = reinterpret_cast<const uint4*>(hs)[kLLds / 4 + q];
      acc = dot2(wr[4 * q + 0], hv.x, acc);
      acc = dot2(wr[4 * q + 1], hv.y, acc);
      acc = dot2(wr[4 * q + 2], hv.z, acc);
      acc = dot2(wr[4 * q + 3], hv.w, acc);
    }
    gs[j] = acc;
    __syncthreads();
    if (j < kLH) {
      const float gi = sigmoidf_(gs[j]), gf = sigmoidf_(gs[kLH + j]), gg = tanhf_(gs[2 * kLH + j]), go = sigmoidf_(gs[3 * kLH + j]);
      c = gf * c + gi * gg;
      const float h = go * tanhf_(c);
      const size_t row = seq + t;
      float* gp = gates_out + row * kLG + j;
      gp[0] = gi; gp[kLH] = gf; gp[2 * kLH] = gg; gp[3 * kLH] = go;
      c_out[row * kLH + j] = c;
      const uint16_t hb = f2bf(h);
      reinterpret_cast<uint16_t*>(hs)[j] = hb;
      h_out[((size_t)b * T + t) * (ndir * kLH) + dir * kLH + j] = hb;
    }
    __syncthreads();
  }
}

// wq: [ndir][4 q][128 jj][256 k] u32 = (W_hh[256 q + 2 jj][k], W_hh[256 q + 2 jj + 1][k]) as bf16 pairs
__global__ __launch_bounds__(1024) void lstm_bwd_kernel(const uint32_t* __restrict__ wq, const float* __restrict__ gates, const float* __restrict__ c_saved,
                                                        const float* __restrict__ dh_out, int ld_dh, int B, int T, int ndir,
                                                        uint16_t* __restrict__ dgates_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* wl = reinterpret_cast<uint32_t*>(smem);                               // [4][kLLds][256]
  uint32_t* dgp = reinterpret_cast<uint32_t*>(smem + 4 * kLLds * kLH * 4);         // [512] bf16 pairs of dgates (pair p = gates 2p, 2p+1)
  float* part = reinterpret_cast<float*>(smem + 4 * kLLds * kLH * 4 + 512 * 4);    // [4][256]
  const int tid = threadIdx.x, k = tid & (kLH - 1), q = tid >> 8;
  const int b = blockIdx.x, dir = blockIdx.y;
  const bool reverse = dir == 1;
  const uint32_t* w = wq + ((size_t)dir * 4 + q) * kLPairs * kLH;
  uint32_t wr[kLReg];
#pragma unroll
  for (int r = 0; r < kLReg; ++r) wr[r] = w[(size_t)(kLLds + r) * kLH + k];
#pragma unroll 4
  for (int jj = 0; jj < kLLds; ++jj) wl[(q * kLLds + jj) * kLH + k] = w[(size_t)jj * kLH + k];
  float dh_rec = 0.f, dc_rec = 0.f;
  const size_t seq = ((size_t)dir * B + b) * T;
  uint16_t* dg16 = reinterpret_cast<uint16_t*>(dgp);
  __syncthreads();
  for (int s = 0; s < T; ++s) {
    const int t = reverse ? s : T - 1 - s;                 // the forward visited t last-to-first in this order
    const int tp = reverse ? t + 1 : t - 1;                // its previous step (in forward order)
    if (tid < kLH) {
      const size_t row = seq + t;
      const float* gp = gates + row * kLG + tid;
      const float gi = gp[0], gf = gp[kLH], gg = gp[2 * kLH], go = gp[3 * kLH];
      const float ct = c_saved[row * kLH + tid];
      const float cprev = (tp >= 0 && tp < T) ? c_saved[(seq + tp) * kLH + tid] : 0.f;
      const float dh = dh_out[((size_t)b * T + t) * ld_dh + dir * kLH + tid] + dh_rec;
      const float tc = tanhf_(ct);
      const float d_o = dh * tc * go * (1.f - go);
      const float dc = dh * go * (1.f - tc * tc) + dc_rec;
      const float d_i = dc * gg * gi * (1.f - gi);
      const float d_f = dc * cprev * gf * (1.f - gf);
      const float d_g = dc * gi * (1.f - gg * gg);
      dc_rec = dc * gf;
      const uint16_t bi = f2bf(d_i), bff = f2bf(d_f), bg = f2bf(d_g), bo = f2bf(d_o);
      dg16[tid] = bi; dg16[kLH + tid] = bff; dg16[2 * kLH + tid] = bg; dg16[3 * kLH + tid] = bo;
      uint16_t* op = dgates_out + row * kLG + tid;
      op[0] = bi; op[kLH] = bff; op[2 * kLH] = bg; op[3 * kLH] = bo;
    }
    __syncthreads();
    // dh_{t-1}[k] = sum_j dgates[j] W_hh[j][k]; this thread: the 256 gates of quarter q
    float acc = 0.f;
#pragma unroll
    for (int g4 = 0; g4 < kLLds / 4; ++g4) {
      const uint4 dv = reinterpret_cast<const uint4*>(dgp + q * kLPairs)[g4];
      acc = dot2(wl[(q * kLLds + 4 * g4 + 0) * kLH + k], dv.x, acc);
      acc = dot2(wl[(q * kLLds + 4 * g4 + 1) * kLH + k], dv.y, acc);
      acc = dot2(wl[(q * kLLds + 4 * g4 + 2) * kLH + k], dv.z, acc);
      acc = dot2(wl[(q * kLLds + 4 * g4 + 3) * kLH + k], dv.w, acc);
    }
#pragma unroll
    for (int g4 = 0; g4 < kLReg / 4; ++g4) {
      const uint4 dv = reinterpret_cast<const uint4*>(dgp + q * kLPairs)[kLLds / 4 + g4];
      acc = dot2(wr[4 * g4 + 0], dv.x, acc);
      acc = dot2(wr[4 * g4 + 1], dv.y, acc);
      acc = dot2(wr[4 * g4 + 2], dv.z, acc);
      acc = dot2(wr[4 * g4 + 3], dv.w, acc);
    }
    part[q * kLH + k] = acc;
    __syncthreads();
    if (tid < kLH) dh_rec = (part[tid] + part[kLH + tid]) + (part[2 * kLH + tid] + part[3 * kLH + tid]);
  }
}

constexpr int kLFwdLds = kLLds * kLG * 4 + kLG * 4 + kLPairs * 4;
constexpr int kLBwdLds = 4 * kLLds * kLH * 4 + 512 * 4 + 4 * kLH * 4;

}  // namespace se

extern "C" int se_lstm_fwd_bf16(const uint32_t* w_hh_pairs, const float* xproj, int B, int T, int ndir, uint16_t* h_out, float* gates_out,
                                float* c_out, void* stream) {
  SE_REQUIRE(w_hh_pairs && xproj && h_out && gates_out && c_out, "se_lstm_fwd_bf16: null argument");
  SE_REQUIRE(B > 0 && B <= 65535 && T > 0 && (ndir == 1 || ndir == 2), "se_lstm_fwd_bf16: bad shape");
  static bool attr_set = false;
  if (!attr_set) {
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::lstm_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, se::kLFwdLds));
    attr_set = true;
  }
  hipLaunchKernelGGL(se::lstm_fwd_kernel, dim3(B, ndir), dim3(1024), se::kLFwdLds, se::as_stream(stream), w_hh_pairs, xproj, B, T, ndir, h_out,
                     gates_out, c_out);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_lstm_bwd_bf16(const uint32_t* w_hh_gate_pairs, const float* gates, const float* c_saved, const float* dh_out, int ld_dh, int B,
                                int T, int ndir, uint16_t* dgates_out, void* stream) {
  SE_REQUIRE(w_hh_gate_pairs && gates && c_saved && dh_out && dgates_out, "se_lstm_bwd_bf16: null argument");
  SE_REQUIRE(B > 0 && B <= 65535 && T > 0 && (ndir == 1 || ndir == 2) && ld_dh >= ndir * se::kLH, "se_lstm_bwd_bf16: bad shape");
  static bool attr_set = false;
  if (!attr_set) {
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::lstm_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, se::kLBwdLds));
    attr_set = true;
  }
  hipLaunchKernelGGL(se::lstm_bwd_kernel, dim3(B, ndir), dim3(1024), se::kLBwdLds, se::as_stream(stream), w_hh_gate_pairs, gates, c_saved, dh_out,
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
