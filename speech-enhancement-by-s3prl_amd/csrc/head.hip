// head.hip -- rows C1 / C2: LinearResidual.forward (model.py:28-34) and Linear.forward (model.py:14-17):
//   [CMVN over time] -> x W^T + b -> activation -> [ (.) noisy power ]
// fused into one pass over the features, in EXACT fp32 (v_mfma_f32_32x32x2_f32 == k-ordered fmaf chain).
//
//   colstats : per (utterance, feature dim) mean and unbiased std over time, exact two-pass
//   head     : workgroup = 128 frames x all N outputs; 4 waves, wave w owns frames [32w, 32w+32) and
//              keeps NT = ceil(N/32) accumulator tiles (7 for N = 201).  K is walked in chunks of 40
//              through LDS: normalised feature tile [128][41] and weight chunk [NT*32][41] (odd pitch ->
//              conflict-free ds_read_b32 operand fetches).
// Bound: HBM (4*F*(D + 2N) bytes per utterance, +4*F*N when `offset` is stored); the f32 MFMA rate
// (157 TF) puts the GEMM itself at about the same time, so the kernel is balanced, not MFMA-bound.
#include "common.h"
#include "prof.h"

namespace se {

using f32x16 = __attribute__((ext_vector_type(16))) float;

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case SE_ACT_RELU: return fmaxf(v, 0.f);
    // hardware exp2 / rcp (1 ulp each: 2e-7 relative on the mask): the IEEE division and libm expf were ~20 vector instructions per output element
    // of an epilogue that holds 112 of them per lane (360 -> 324 us per launch at B = 256).  Measured without effect on the same launch: register-
    // prefetched staging, and 16-B operand reads from a pitch-44 tile feeding four MFMAs each (340 us)
    case SE_ACT_SIGMOID: return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * v));
    case SE_ACT_GELU: return v * 0.5f * (1.0f + erff(v * 0.70710678118654752f));
    case SE_ACT_EXP: return expf(v);
    default: return v;
  }
}

// stats[(b*D + d)*2] = mean over time, [..+1] = 1 / (unbiased std + eps)
// One pass (round 3; the first version read the features twice): per column fp64 sums of x and x^2 -- the products of two fp32 values are exact in
// fp64 and 1001 of them lose ~1e-13 relative, so the variance equals the two-pass fp32 result to well below fp32 resolution -- four rows in
// flight per thread, 4 row phases per workgroup.
__global__ __launch_bounds__(256) void colstats_kernel(const float* __restrict__ feats, int F, int D, float eps,
                                                       float* __restrict__ stats) {
  __shared__ double red[2][4][64];
  const int b = blockIdx.y, d = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
  const bool ok = d < D;
  const float* base = feats + (size_t)b * F * D + d;
  double s = 0.0, q = 0.0;
  if (ok) {
    int t = ph;
    for (; t + 12 < F; t += 16) {
      const float v0 = base[(size_t)t * D], v1 = base[(size_t)(t + 4) * D], v2 = base[(size_t)(t + 8) * D], v3 = base[(size_t)(t + 12) * D];
      s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
      q += ((double)v0 * v0 + (double)v1 * v1) + ((double)v2 * v2 + (double)v3 * v3);
    }
    for (; t < F; t += 4) {
      const float v = base[(size_t)t * D];
      s += (double)v;
      q += (double)v * v;
    }
  }
  red[0][ph][threadIdx.x & 63] = s;
  red[1][ph][threadIdx.x & 63] = q;
  __syncthreads();
  if (ph == 0 && ok) {
    const int c = threadIdx.x;
    const double S = (red[0][0][c] + red[0][1][c]) + (red[0][2][c] + red[0][3][c]);
    const double Q = (red[1][0][c] + red[1][1][c]) + (red[1][2][c] + red[1][3][c]);
    const double mean = S / (double)F;
    const double var = fmax(Q - S * mean, 0.0) / (double)(F - 1);
    stats[((size_t)b * D + d) * 2] = (float)mean;
    stats[((size_t)b * D + d) * 2 + 1] = 1.0f / ((float)sqrt(var) + eps);
  }
}

constexpr int kHM = 128;     // frames per workgroup
// LDS footprint = occupancy: K chunks of 40 with 16-row epilogue passes are 57.7 KB (2 workgroups per CU); chunks of 20 with 8-row passes are
// 29.6 KB, and the 168 registers of the N = 201 instantiation then allow 3 workgroups per CU: 328 -> 307 us (D = 120), 460 -> 419 us (D = 201)
// per launch at B = 256 (tools/head_occ.sh; either change alone does nothing)
#ifndef SE_HEAD_HK
#define SE_HEAD_HK 20
#endif
#ifndef SE_HEAD_HALF
#define SE_HEAD_HALF 8
#endif
constexpr int kHK = SE_HEAD_HK;      // K chunk (kHK / 4 16-B pieces per row)
constexpr int kHP = kHK + 1; // LDS pitch (odd: conflict-free ds_read_b32 operand fetches)
constexpr int kHHalf = SE_HEAD_HALF;   // rows of a wave's 32 that go out per epilogue pass (16 or 8)

// Round 3 rewrite (the first version staged every element with its own index division and 4-B accesses and stored the outputs as 128-B row
// pieces straight from the accumulator layout: 787 us for 256 utterances, 0.12 of the HBM rate).  Now:
//   staging  : 16-B global loads (features: CMVN applied on the way; weights), no per-element division; LDS keeps the odd pitch
//   epilogue : a wave's 32 x N outputs are ONE contiguous span of the row-major (M, N) tensors (N is the whole row), so the activated mask goes
//              through LDS in that layout, 16 rows at a time, and leaves as aligned 16-B stores with the noisy power read the same way
//              (predicted = linears * offset): every global access of the kernel is a full 16-B lane access on consecutive addresses.
template <int NT>
__global__ __launch_bounds__(256, 2) void head_kernel(const float* __restrict__ feats, const float* __restrict__ W,
                                                      const float* __restrict__ bias, const float* __restrict__ linears,
                                                      const float* __restrict__ stats, int rows, int F, int D, int N, int act,
                                                      float* __restrict__ predicted, float* __restrict__ offset, int vec_io) {
  constexpr int kStage = (kHM + NT * 32) * kHP;                     // floats: feature tile + weight chunk
  constexpr int kEpi = 4 * kHHalf * (NT * 32);                       // floats: per wave 16 rows x (up to NT * 32) outputs
  __shared__ __attribute__((aligned(16))) float smem[kStage > kEpi ? kStage : kEpi];
  float* As = smem;
  float* Ws = smem + kHM * kHP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * kHM;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  for (int k0 = 0; k0 < D; k0 += kHK) {
    __syncthreads();
    // ---- stage the normalised feature tile: item = (row, 16-B piece); 10 pieces per row
    for (int it = tid; it < kHM * (kHK / 4); it += 256) {
      const int r = it / (kHK / 4), c = it - r * (kHK / 4);
      const int row = row0 + r, k = k0 + 4 * c;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < rows && k < D) {
        const float* src = feats + (size_t)row * D + k;
        if (k + 3 < D) {
          v = *reinterpret_cast<const float4*>(src);
        } else {
          v.x = src[0];
          if (k + 1 < D) v.y = src[1];
          if (k + 2 < D) v.z = src[2];
        }
        if (stats) {
          const float* sp = stats + ((size_t)(row / F) * D + k) * 2;           // (mean, 1 / (std + eps)) pairs
          if (k + 3 < D) {
            const float4 s0 = *reinterpret_cast<const float4*>(sp), s1 = *reinterpret_cast<const float4*>(sp + 4);
            v = make_float4((v.x - s0.x) * s0.y, (v.y - s0.z) * s0.w, (v.z - s1.x) * s1.y, (v.w - s1.z) * s1.w);
          } else {
            v.x = (v.x - sp[0]) * sp[1];
            if (k + 1 < D) v.y = (v.y - sp[2]) * sp[3];
            if (k + 2 < D) v.z = (v.z - sp[4]) * sp[5];
          }
        }
      }
      float* d = As + r * kHP + 4 * c;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    // ---- stage the weight chunk: item = (output n, 16-B piece)
    for (int it = tid; it < NT * 32 * (kHK / 4); it += 256) {
      const int n = it / (kHK / 4), c = it - n * (kHK / 4);
      const int k = k0 + 4 * c;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < N && k < D) {
        const float* src = W + (size_t)n * D + k;
        if (k + 3 < D) {
          v = *reinterpret_cast<const float4*>(src);
        } else {
          v.x = src[0];
          if (k + 1 < D) v.y = src[1];
          if (k + 2 < D) v.z = src[2];
        }
      }
      float* d = Ws + n * kHP + 4 * c;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    const float* ap = As + (wave * 32 + (lane & 31)) * kHP + (lane >> 5);
    const float* wp = Ws + (lane & 31) * kHP + (lane >> 5);
#pragma unroll 4
    for (int kk = 0; kk < kHK; kk += 2) {
      const float a = ap[kk];
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wp[t * 32 * kHP + kk], acc[t], 0, 0, 0);
    }
  }

  // ---- epilogue.  C/D map of 32x32: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5): registers 8 h .. 8 h + 7 hold rows 16 h .. 16 h + 15
  __syncthreads();                                              // every wave is done with the staging tiles
  float* Es = smem + wave * (kHHalf * NT * 32);                 // this wave's [16][N] slab, rows packed at pitch N: the global layout of the span
  float bn[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = t * 32 + (lane & 31);
    bn[t] = (bias && n < N) ? bias[n] : 0.f;
  }
#pragma unroll
  for (int h = 0; h < 32 / kHHalf; ++h) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = t * 32 + (lane & 31);
      if (n < N) {
#pragma unroll
        for (int r8 = 0; r8 < kHHalf / 2; ++r8) {
          const int r = (kHHalf / 2) * h + r8;
          const int rl = (r & 3) + (kHHalf == 16 ? 8 * ((r >> 2) & 1) : 0) + 4 * (lane >> 5);       // row within the pass's 16 (8) rows
          Es[rl * N + n] = apply_act(acc[t][r] + bn[t], act);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int rbase = row0 + wave * 32 + kHHalf * h;                            // a multiple of 16: the span starts 64-B aligned for any N
    const int nrow = min(kHHalf, rows - rbase);
    if (nrow > 0) {
      const int cnt = nrow * N;
      const size_t g = (size_t)rbase * N;
      if (vec_io) {
        const int nvec = cnt >> 2;
        for (int i = lane; i < nvec; i += 64) {
          const float4 o = *reinterpret_cast<const float4*>(Es + 4 * i);
          if (offset) *reinterpret_cast<float4*>(offset + g + 4 * i) = o;
          if (predicted) {
            float4 p = o;
            if (linears) {
              const float4 l = *reinterpret_cast<const float4*>(linears + g + 4 * i);
              p = make_float4(o.x * l.x, o.y * l.y, o.z * l.z, o.w * l.w);
            }
            *reinterpret_cast<float4*>(predicted + g + 4 * i) = p;
          }
        }
        for (int i = 4 * nvec + lane; i < cnt; i += 64) {
          const float o = Es[i];
          if (offset) offset[g + i] = o;
          if (predicted) predicted[g + i] = linears ? linears[g + i] * o : o;
        }
      } else {
        for (int i = lane; i < cnt; i += 64) {
          const float o = Es[i];
          if (offset) offset[g + i] = o;
          if (predicted) predicted[g + i] = linears ? linears[g + i] * o : o;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
}

}  // namespace se

// internal (not part of the public header): column statistics shared with head_bwd.hip
extern "C" int se_head_colstats_f32(const float* feats, int B, int F, int D, float eps, float* stats, void* stream) {
  hipLaunchKernelGGL(se::colstats_kernel, dim3((D + 63) / 64, B), dim3(256), 0, se::as_stream(stream), feats, F, D, eps, stats);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" size_t se_head_workspace_bytes(int B, int F, int D, int N) {
  (void)F;
  // [stats: B*D*2 floats][bwd scratch: g_pre (B*F*N) is NOT kept here; see se_head_linear_bwd_f32]
  return (size_t)B * D * 2 * sizeof(float) + 256;
}

template <int NT>
static int launch_head(const float* feats, const float* W, const float* bias, const float* linears, const float* stats,
                       int rows, int F, int D, int N, int act, float* predicted, float* offset, hipStream_t st) {
  const int vec_io = ((((uintptr_t)predicted | (uintptr_t)offset | (uintptr_t)linears) % 16) == 0) ? 1 : 0;
  // algorithmic bytes: features + noisy power in, predicted + offset out
  se::ProfScope prof(se::kProfHead, 4.0 * rows * ((double)D + (linears ? N : 0) + (predicted ? N : 0) + (offset ? N : 0)), st);
  hipLaunchKernelGGL((se::head_kernel<NT>), dim3((rows + se::kHM - 1) / se::kHM), dim3(256), 0, st, feats, W, bias, linears,
                     stats, rows, F, D, N, act, predicted, offset, vec_io);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_head_linear_f32(const float* feats, const float* W, const float* bias, const float* linears,
                                  int B, int F, int D, int N, int act, int cmvn, float eps,
                                  float* predicted, float* offset, void* workspace, size_t workspace_bytes, void* stream) {
  SE_REQUIRE(feats && W && (predicted || offset), "se_head_linear_f32: null argument");
  SE_REQUIRE(B > 0 && B <= 65535 && F >= 2 && D > 0 && N > 0 && N <= 256, "se_head_linear_f32: bad shape B=%d F=%d D=%d N=%d (N <= 256)", B, F, D, N);
  hipStream_t st = se::as_stream(stream);
  float* stats = nullptr;
  if (cmvn) {
    SE_REQUIRE(workspace && workspace_bytes >= se_head_workspace_bytes(B, F, D, N), "se_head_linear_f32: workspace too small");
    stats = reinterpret_cast<float*>(workspace);
    int rc = se_head_colstats_f32(feats, B, F, D, eps, stats, stream);
    if (rc) return rc;
  }
  const int rows = B * F;
  const int nt = (N + 31) / 32;
  switch (nt) {
    case 1: return launch_head<1>(feats, W, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
    case 2: return launch_head<2>(feats, W, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
    case 3: return launch_head<3>(feats, W, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
    case 4: return launch_head<4>(feats, W, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
    case 5: return launch_head<5>(feats, W, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
    case 6: return launch_head<6>(feats, W, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
    case 7: return launch_head<7>(feats, W, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
    default: return launch_head<8>(feats, W, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
  }
}
