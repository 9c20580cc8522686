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
#include <stdlib.h>
#include <algorithm>
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


// ---------------------------------------------------------------------------------------------------------------------------------------------
// Round 4: the SAME fp32 result on the bf16 matrix instruction.  VERDICT r3 #6: the kernel above is not HBM-bound but fp32-MFMA-bound --
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate: at D = 201, 256 utterances, 23 GFLOP / 155 TF/s = 149 us against a 77 us HBM floor.
// Here both operands are split into THREE bf16 terms, x = x1 + x2 + x3 (x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2): the residuals are
// exact in fp32, the dropped remainder is <= 2^-24 |x|), and the product is the six terms whose weight is >= 2^-16:
//     x w  =  x1 w1 + (x1 w2 + x2 w1) + (x2 w2 + x1 w3 + x3 w1)  + O(2^-24)
// -- every bf16 product is exact in the fp32 accumulator, so the sum is an fp32 dot product with a different (and no worse) summation order.  Six
// v_mfma_f32_32x32x16_bf16 (6 x 32 cycles for 16 k) replace eight v_mfma_f32_32x32x2_f32 (8 x 64 cycles): 2.7 x less matrix time, which puts the
// kernel under its HBM floor.  Weights arrive pre-split (head_split_w_kernel: 3 planes [NT * 32][Kp] bf16, zero padded), the features are split
// while they are staged (CMVN first, in fp32).  LDS: 16-k chunks, rows at a 48-B pitch (conflict-free ds_read_b128 of 16 consecutive rows):
// 3 x (128 + NT * 32) x 48 B = 50.7 KB at N = 201 -> 3 workgroups per CU as before.  Epilogue: the kernel above's, unchanged (same C / D map).
constexpr int kH3K = 16;          // k per chunk
// developer ablation of head3_kernel (timing only, results wrong): -DSE_HEAD_ABL=<mask>: 1 no K loop, 2 no epilogue loads / stores (LDS round trip kept),
// 4 no feature / statistics loads, 8 no MFMAs, 16 no activation
#ifndef SE_HEAD_ABL
#define SE_HEAD_ABL 0
#endif
typedef __attribute__((ext_vector_type(8))) __bf16 h3_bf16x8;

__device__ __forceinline__ void split3(float x, uint16_t& a, uint16_t& b, uint16_t& c) {
  const __bf16 x1 = (__bf16)x;
  const float r1 = x - (float)x1;
  const __bf16 x2 = (__bf16)r1;
  const float r2 = r1 - (float)x2;
  const __bf16 x3 = (__bf16)r2;
  a = __builtin_bit_cast(uint16_t, x1);
  b = __builtin_bit_cast(uint16_t, x2);
  c = __builtin_bit_cast(uint16_t, x3);
}

// W (N, D) fp32 -> W3 [chunk = k / 16][plane 0..2][half = (k / 8) & 1][row 0..rows_p) x 8 bf16 (16 B): CHUNK-major, so that the 3 NT KiB a
// workgroup needs per 16-k chunk are ONE contiguous span -- 1-KiB LDS-DMA pieces with lane-linear source AND destination -- and [half][row] inside,
// which is also what the fragment reads want: lane l reads row (l & 31), half (l >> 5): 32 rows x 16 B contiguous per half = conflict-free
// ds_read_b128 without padding.  Rows >= N and k >= D are zero.
__global__ __launch_bounds__(256) void head_split_w_kernel(const float* __restrict__ W, int N, int D, int rows_p, int Kp, uint16_t* __restrict__ W3) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rows_p * Kp) return;
  const int n = i / Kp, k = i - n * Kp;
  const float v = (n < N && k < D) ? W[(size_t)n * D + k] : 0.f;
  uint16_t p[3];
  split3(v, p[0], p[1], p[2]);
  const int c = k >> 4, h = (k >> 3) & 1, e = k & 7;
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) W3[((((size_t)c * 3 + pl) * 2 + h) * rows_p + n) * 8 + e] = p[pl];
}

// Round 4, second version (the first one staged weights and features through registers per chunk and exposed both latencies twice per chunk:
// 254 / 313 us): weights by LDS-DMA one chunk ahead (double-buffered, no registers, no address arithmetic: the chunk is one contiguous span),
// the next chunk's feature rows (+ their CMVN statistics) loaded into registers before this chunk's MFMAs and split / written to LDS at the top of
// the next one (double-buffered): ONE barrier per chunk, no global latency on the critical path.  LDS 2 x 3 NT KiB + 2 x 12 KiB = 66 KiB at
// N = 201 -> two workgroups per CU.
template <int NT>
__global__ __launch_bounds__(256, 2) void head3_kernel(const float* __restrict__ feats, const uint16_t* __restrict__ W3, int Kp,
                                                       const float* __restrict__ bias, const float* __restrict__ linears,
                                                       const float* __restrict__ stats, int rows, int F, int D, int N, int act,
                                                       float* __restrict__ predicted, float* __restrict__ offset, int vec_io) {
  constexpr int kWBuf = 3 * 2 * NT * 32 * 16;                        // bytes of one weight chunk: [plane][half][row] x 16 B
  constexpr int kABuf = 3 * 2 * kHM * 16;                            // one feature chunk, same layout over the 128 rows
  constexpr int kStage = 2 * kWBuf + 2 * kABuf;
  constexpr int kEpi = 4 * kHHalf * (NT * 32) * 4;
  constexpr int kPieces = kWBuf / 1024;                              // 3 NT
  __shared__ __attribute__((aligned(16))) char smem3[kStage > kEpi ? kStage : kEpi];
  char* Ws = smem3;
  char* As = smem3 + 2 * kWBuf;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = blockIdx.x * kHM;
  const int nchunk = (SE_HEAD_ABL & 1) ? 0 : Kp / kH3K;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // ---- weight chunk c -> buffer c & 1: pieces wave, wave + 4, ... of its 3 NT KiB (inline asm: invisible to the compiler's vmcnt bookkeeping;
  //      the one wait on them is the vmcnt(0) at the top of the chunk that reads them)
  typedef __attribute__((address_space(3))) char* lds_h_t;
  const uint32_t lds_w = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(lds_h_t)smem3);
  const char* wbase = reinterpret_cast<const char*>(W3);
#define SEH_DMA_W(c)                                                                                                        \
  do {                                                                                                                      \
    const char* sb_ = wbase + (size_t)(c) * kWBuf;                                                                          \
    for (int j_ = wave; j_ < kPieces; j_ += 4) {                                                                            \
      const uint32_t off_ = (uint32_t)(j_ * 1024 + lane * 16);                                                              \
      uint32_t keep_;                                                                                                       \
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"  \
                   : "=&s"(keep_) : "v"(off_), "s"(sb_), "s"(lds_w + (uint32_t)(((c) & 1) * kWBuf + j_ * 1024)) : "memory"); \
    }                                                                                                                       \
  } while (0)
  // ---- feature chunk c: this thread's two items (row, 4-float piece) and their statistics, into registers
  float4 fv[2], fs0[2], fs1[2];
#define SEH_LOAD_F(c)                                                                                                       \
  _Pragma("unroll") for (int rep = 0; rep < 2; ++rep) {                                                                     \
    const int it = tid + 256 * rep;                                                                                         \
    const int r = it >> 2, c4 = it & 3;                                                                                     \
    const int row = row0 + r, k = (c) * kH3K + 4 * c4;                                                                      \
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f), s0 = make_float4(0.f, 1.f, 0.f, 1.f), s1 = s0;                              \
    if (!(SE_HEAD_ABL & 4) && row < rows && k < D) {                                                                        \
      const float* src = feats + (size_t)row * D + k;                                                                       \
      if (k + 3 < D) {                                                                                                      \
        v = *reinterpret_cast<const float4*>(src);                                                                          \
      } else {                                                                                                              \
        v.x = src[0];                                                                                                       \
        if (k + 1 < D) v.y = src[1];                                                                                        \
        if (k + 2 < D) v.z = src[2];                                                                                        \
      }                                                                                                                     \
      if (stats) {                                                                                                          \
        const float* sp = stats + ((size_t)(row / F) * D + k) * 2;           /* (mean, 1 / (std + eps)) pairs */             \
        if (k + 3 < D) {                                                                                                    \
          s0 = *reinterpret_cast<const float4*>(sp);                                                                        \
          s1 = *reinterpret_cast<const float4*>(sp + 4);                                                                    \
        } else {                                                                                                            \
          s0.x = sp[0]; s0.y = sp[1];                                                                                       \
          if (k + 1 < D) { s0.z = sp[2]; s0.w = sp[3]; }                                                                    \
          if (k + 2 < D) { s1.x = sp[4]; s1.y = sp[5]; }                                                                    \
        }                                                                                                                   \
      }                                                                                                                     \
    }                                                                                                                       \
    fv[rep] = v; fs0[rep] = s0; fs1[rep] = s1;                                                                              \
  }

  SEH_DMA_W(0);
  SEH_LOAD_F(0)
  // The noisy-power rows a pass multiplies with are loaded ONE PASS AHEAD into registers (round 4): a load inside the store loop costs a full memory
  // round trip per iteration (load -> use), ~25 of them per wave, with only two or three workgroups per CU to hide it -- that, not the GEMM, was most
  // of this kernel's time (the matrix work is ~40 us of a 270 us launch).
  constexpr int kVecIt = (kHHalf * NT * 32 / 4 + 63) / 64;      // float4 items per lane and pass (7 at N <= 224, 8 rows)
  float4 lin[2][kVecIt];
  auto load_lin = [&](int h, float4 (&dst)[kVecIt]) {
    const int rbase = row0 + wave * 32 + kHHalf * h;
    const int nrow = min(kHHalf, rows - rbase);
    const int nvec = nrow > 0 ? (nrow * N) >> 2 : 0;
    const size_t g = (size_t)rbase * N;
#pragma unroll
    for (int j = 0; j < kVecIt; ++j) {
      const int i = lane + 64 * j;
      dst[j] = (!(SE_HEAD_ABL & 2) && linears && vec_io && i < nvec) ? *reinterpret_cast<const float4*>(linears + g + 4 * i) : make_float4(1.f, 1.f, 1.f, 1.f);
    }
  };
  load_lin(0, lin[0]);      // the first pass's rows travel under the whole K loop
  for (int c = 0; c < nchunk; ++c) {
    // everything issued one chunk ago has landed: this wave's pieces of weight chunk c, this thread's feature items of chunk c
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- features: CMVN in fp32, the three-term split, 8 B per plane into [plane][half][row]
    char* Ab = As + (c & 1) * kABuf;
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      const int it = tid + 256 * rep;
      const int r = it >> 2, c4 = it & 3;
      float4 v = fv[rep];
      const float4 s0 = fs0[rep], s1 = fs1[rep];
      v = make_float4((v.x - s0.x) * s0.y, (v.y - s0.z) * s0.w, (v.z - s1.x) * s1.y, (v.w - s1.z) * s1.w);      // (0, 1) pairs without CMVN / past D
      uint16_t p0[4], p1[4], p2[4];
      split3(v.x, p0[0], p1[0], p2[0]);
      split3(v.y, p0[1], p1[1], p2[1]);
      split3(v.z, p0[2], p1[2], p2[2]);
      split3(v.w, p0[3], p1[3], p2[3]);
      char* d = Ab + ((c4 >> 1) * kHM + r) * 16 + (c4 & 1) * 8;
      *reinterpret_cast<uint2*>(d) = make_uint2(p0[0] | ((uint32_t)p0[1] << 16), p0[2] | ((uint32_t)p0[3] << 16));
      *reinterpret_cast<uint2*>(d + 2 * kHM * 16) = make_uint2(p1[0] | ((uint32_t)p1[1] << 16), p1[2] | ((uint32_t)p1[3] << 16));
      *reinterpret_cast<uint2*>(d + 4 * kHM * 16) = make_uint2(p2[0] | ((uint32_t)p2[1] << 16), p2[2] | ((uint32_t)p2[3] << 16));
    }
    __syncthreads();      // weight chunk c and feature chunk c visible; every wave is past chunk c - 1 (whose buffers are refilled next)
    if (c + 1 < nchunk) {
      SEH_DMA_W(c + 1);
      SEH_LOAD_F(c + 1)
    }
    const char* ap = Ab + (wave * 32 + (lane & 31)) * 16 + (lane >> 5) * (kHM * 16);
    const char* wp = Ws + (c & 1) * kWBuf + (lane & 31) * 16 + (lane >> 5) * (NT * 32 * 16);
    const h3_bf16x8 a0 = *reinterpret_cast<const h3_bf16x8*>(ap), a1 = *reinterpret_cast<const h3_bf16x8*>(ap + 2 * kHM * 16),
                    a2 = *reinterpret_cast<const h3_bf16x8*>(ap + 4 * kHM * 16);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const char* wt = wp + t * 32 * 16;
      const h3_bf16x8 w0 = *reinterpret_cast<const h3_bf16x8*>(wt), w1 = *reinterpret_cast<const h3_bf16x8*>(wt + 2 * NT * 32 * 16),
                      w2 = *reinterpret_cast<const h3_bf16x8*>(wt + 4 * NT * 32 * 16);
      if (SE_HEAD_ABL & 8) { asm volatile("" :: "v"(w0), "v"(w1), "v"(w2), "v"(a0), "v"(a1), "v"(a2)); continue; }
      // smallest terms first
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, w0, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, w2, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, w1, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, w0, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, w1, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, w0, acc[t], 0, 0, 0);
    }
  }
#undef SEH_DMA_W
#undef SEH_LOAD_F

  // ---- epilogue: as head_kernel (same accumulator map).  The slab is private to the wave and an LDS queue serves one wave's instructions in order: what
  //      the exchange between its lanes needs is the LDS counter drained and the compiler kept from moving LDS accesses across the point.  A
  //      workgroup-scope fence (the round-3 form, still in head_kernel) also waits for vmcnt(0): every pass then sat out the acknowledgement of its
  //      own 16-B stores AND the arrival of the next pass's noisy-power rows it had just requested -- the prefetch hid nothing
#define SEH_LDS_WAVE_SYNC()                                   \
  do {                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        \
    __builtin_amdgcn_wave_barrier();                          \
    asm volatile("" ::: "memory");                            \
  } while (0)
  __syncthreads();
  float* Es = reinterpret_cast<float*>(smem3) + wave * (kHHalf * NT * 32);
  float bn[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = t * 32 + (lane & 31);
    bn[t] = (bias && n < N) ? bias[n] : 0.f;
  }
#pragma unroll
  for (int h = 0; h < 32 / kHHalf; ++h) {
    if (h + 1 < 32 / kHHalf) load_lin(h + 1, lin[(h + 1) & 1]);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = t * 32 + (lane & 31);
      if (n < N) {
#pragma unroll
        for (int r8 = 0; r8 < kHHalf / 2; ++r8) {
          const int r = (kHHalf / 2) * h + r8;
          const int rl = (r & 3) + (kHHalf == 16 ? 8 * ((r >> 2) & 1) : 0) + 4 * (lane >> 5);
          Es[rl * N + n] = (SE_HEAD_ABL & 16) ? acc[t][r] + bn[t] : apply_act(acc[t][r] + bn[t], act);
        }
      }
    }
    SEH_LDS_WAVE_SYNC();
    const int rbase = row0 + wave * 32 + kHHalf * h;
    const int nrow = min(kHHalf, rows - rbase);
    if (nrow > 0) {
      const int cnt = nrow * N;
      const size_t g = (size_t)rbase * N;
      if (vec_io) {
        const int nvec = cnt >> 2;
#pragma unroll
        for (int j = 0; j < kVecIt; ++j) {
          const int i = lane + 64 * j;
          if (i < nvec) {
            const float4 o = *reinterpret_cast<const float4*>(Es + 4 * i);
            if (SE_HEAD_ABL & 2) { asm volatile("" :: "v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w)); continue; }
            if (offset) *reinterpret_cast<float4*>(offset + g + 4 * i) = o;
            if (predicted) {
              const float4 l = lin[h & 1][j];
              *reinterpret_cast<float4*>(predicted + g + 4 * i) = make_float4(o.x * l.x, o.y * l.y, o.z * l.z, o.w * l.w);
            }
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
    SEH_LDS_WAVE_SYNC();
  }
}

#undef SEH_LDS_WAVE_SYNC


// ---------------------------------------------------------------------------------------------------------------------------------------------
// Round 5: head3_kernel without its control flow.  The ablation of head3 at 256 utterances (tools/one_head.py, -DSE_HEAD_ABL: 234 us = K loop 118
// [feature loads 79, MFMAs 44] + epilogue traffic 74 + activation 41, nothing overlapping anything) and its ISA said why: `act` is a run-time switch
// evaluated per output element (branches around every v_exp / v_rcp), the feature / statistics loads sit in nested conditionals whose merges the
// compiler resolves with s_waitcnt vmcnt(0) INSIDE the "prefetch", and the workgroup-scope fences of the epilogue wait for vmcnt(0) too -- each
// pass sat out its own stores and the rows it had just requested.  Here:
//   * ACT is a template parameter; N, alignment and F >= 128 are launcher preconditions (anything else runs head3_kernel)
//   * the CMVN statistics of the (at most two) utterances a 128-row tile touches are staged in LDS once; a chunk's feature items are then ONE
//     unconditional 16-B load each (row and column clamped into the tensor; the column shift is undone in registers), issued THREE chunks ahead into
//     statically named registers (the chunk loop is unrolled by three), with hand-counted vmcnt beside the weight DMA
//   * the epilogue's lane exchange through the wave's private LDS slab waits for the LDS counter only
//   * SIS: the SISDR criterion's three sums (objective.py:81-100: <s, t>, <t, t>, <s, s> of the square-rooted planes over the valid frames) are taken from
//     the products on their way out -- one extra plane read instead of a launch that reads two -- per workgroup and utterance into a slab that
//     sisdr_head_mean_kernel (objectives.hip) folds in a fixed order
template <int ACT>
__device__ __forceinline__ float act_c(float v) {
  if constexpr (ACT == SE_ACT_RELU) return fmaxf(v, 0.f);
  else if constexpr (ACT == SE_ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * v));
  else if constexpr (ACT == SE_ACT_GELU) return v * 0.5f * (1.0f + erff(v * 0.70710678118654752f));
  else if constexpr (ACT == SE_ACT_EXP) return expf(v);
  else return v;
}

constexpr int kH5StatCols = 256;      // Kp <= 256

template <int NT, int ACT, int SIS>
__global__ __launch_bounds__(256, 2) void head5_kernel(const float* __restrict__ feats, const uint16_t* __restrict__ W3, int Kp,
                                                       const float* __restrict__ bias, const float* __restrict__ linears,
                                                       const float* __restrict__ stats, int rows, int F, int D, int N,
                                                       float* __restrict__ predicted, float* __restrict__ offset,
                                                       const float* __restrict__ tar, const int64_t* __restrict__ lengths, int len_div,
                                                       double* __restrict__ slab) {
  constexpr int kWBuf = 3 * 2 * NT * 32 * 16;
  constexpr int kABuf = 3 * 2 * kHM * 16;
  constexpr int kStage = 2 * kWBuf + 2 * kABuf;
  constexpr int kEpi = 4 * kHHalf * (NT * 32) * 4;
  constexpr int kMain = kStage > kEpi ? kStage : kEpi;
  constexpr int kPieces = kWBuf / 1024;                              // 3 NT
  constexpr int kWIss = (kPieces + 3) / 4;                           // weight DMAs per wave and chunk (the last one may repeat a piece)
  __shared__ __attribute__((aligned(16))) char smem5[kMain + 2 * kH5StatCols * 8 + 64];
  char* Ws = smem5;
  char* As = smem5 + 2 * kWBuf;
  float2* St = reinterpret_cast<float2*>(smem5 + kMain);            // [2 utterances][Kp] (mean, 1 / (std + eps))
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row0 = blockIdx.x * kHM;
  const int nchunk = Kp / kH3K;
  const int b0 = row0 / F;
  const int boundary = (b0 + 1) * F;                                 // first row of the tile's second utterance (F >= 128: there is no third)

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  typedef __attribute__((address_space(3))) char* lds_h_t;
  const uint32_t lds_w = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(lds_h_t)smem5);
  const char* wbase = reinterpret_cast<const char*>(W3);
#define SE5_DMA_W(c)                                                                                                        \
  do {                                                                                                                      \
    const char* sb_ = wbase + (size_t)(c) * kWBuf;                                                                          \
    _Pragma("unroll") for (int q_ = 0; q_ < kWIss; ++q_) {                                                                  \
      const int j_ = min(wave + 4 * q_, kPieces - 1);                                                                       \
      const uint32_t off_ = (uint32_t)(j_ * 1024 + lane * 16);                                                              \
      uint32_t keep_;                                                                                                       \
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"  \
                   : "=&s"(keep_) : "v"(off_), "s"(sb_), "s"(lds_w + (uint32_t)(((c) & 1) * kWBuf + j_ * 1024)) : "memory"); \
    }                                                                                                                       \
  } while (0)

  // ---- this thread's two feature items per chunk: rows r and 64 + r, 16-B piece c4
  const int fr = tid >> 2, c4 = tid & 3;
  const float* fp0 = feats + (size_t)min(row0 + fr, rows - 1) * D;
  const float* fp1 = feats + (size_t)min(row0 + 64 + fr, rows - 1) * D;
  const int su0 = (row0 + fr >= boundary) ? Kp : 0, su1 = (row0 + 64 + fr >= boundary) ? Kp : 0;
  float4 fv[3][2];
#define SE5_LOAD_F(c, s)                                                                                                    \
  do {                                                                                                                      \
    const int kk_ = min((c) * kH3K + 4 * c4, D - 4);                                                                        \
    fv[s][0] = *reinterpret_cast<const float4*>(fp0 + kk_);                                                                 \
    fv[s][1] = *reinterpret_cast<const float4*>(fp1 + kk_);                                                                 \
  } while (0)

  // statistics of the tile's two utterances: requested FIRST (they are the oldest loads when their LDS stores wait for them, so that wait leaves the
  // weight DMA and the feature loads behind them in flight), stored to LDS after everything else of the prologue has been issued
  float2 sv[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int i = tid + 256 * q;
    const int u = i >= Kp ? 1 : 0, k = i - u * Kp;
    sv[q] = make_float2(0.f, 1.f);
    if (stats && i < 2 * Kp && k < D && (size_t)(b0 + u) * F < (size_t)rows) sv[q] = *reinterpret_cast<const float2*>(stats + ((size_t)(b0 + u) * D + k) * 2);
  }
  SE5_DMA_W(0);
  SE5_LOAD_F(0, 0);
  if (nchunk > 1) SE5_LOAD_F(1, 1);
  if (nchunk > 2) SE5_LOAD_F(2, 2);
  // the noisy-power rows (and, SIS, the target rows) of a pass: ONE register set each, item j re-requested for the next pass right after pass h has
  // used it (between that pass's stores) -- a pass of look-ahead for both planes in 2 x 28 registers; the first pass's rows travel under the K loop
  constexpr int kVecIt = (kHHalf * NT * 32 / 4 + 63) / 64;
  float4 lin[kVecIt], tv[SIS ? kVecIt : 1];
  auto load_item = [&](const float* __restrict__ src, int h, int j, float fill) -> float4 {
    const int rbase = row0 + wave * 32 + kHHalf * h;
    const int nrow = min(kHHalf, rows - rbase);
    const int nvec = nrow > 0 ? (nrow * N) >> 2 : 0;
    const int i = lane + 64 * j;
    return (src && h < 32 / kHHalf && i < nvec) ? *reinterpret_cast<const float4*>(src + (size_t)rbase * N + 4 * i) : make_float4(fill, fill, fill, fill);
  };
#pragma unroll
  for (int j = 0; j < kVecIt; ++j) {
    lin[j] = load_item(linears, 0, j, 1.f);
    if constexpr (SIS) tv[j] = load_item(tar, 0, j, 0.f);
  }
  St[tid] = sv[0];
  if (tid + 256 < 2 * Kp) St[tid + 256] = sv[1];
  __syncthreads();                                                   // statistics visible (the first split reads them)

  // one chunk: S = c % 3 (static).  Issue order per wave: prologue [W(0) F(0) F(1) F(2) ...], then per chunk c [W(c + 1) F(c + 3)]; at the top of
  // chunk c the only younger group that may stay in flight is F(c + 2) (two loads) -- at c = 0 F(1) and F(2)
#define SE5_CHUNK(c, S)                                                                                                     \
  do {                                                                                                                      \
    if ((c) == 0) {                                                                                                         \
      if (nchunk > 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                      \
      else if (nchunk > 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                                 \
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                 \
    } else if ((c) + 2 < nchunk) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                           \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                   \
    char* Ab = As + ((c) & 1) * kABuf;                                                                                      \
    const int k_ = (c) * kH3K + 4 * c4;                                                                                     \
    const int sh_ = k_ - min(k_, D - 4);              /* 0 inside the row; 1..3 on its last piece; >= 4 past it */           \
    _Pragma("unroll") for (int rep = 0; rep < 2; ++rep) {                                                                   \
      const float4 a = fv[S][rep];                                                                                          \
      float4 v;                                                                                                             \
      v.x = sh_ == 0 ? a.x : sh_ == 1 ? a.y : sh_ == 2 ? a.z : sh_ == 3 ? a.w : 0.f;                                        \
      v.y = sh_ == 0 ? a.y : sh_ == 1 ? a.z : sh_ == 2 ? a.w : 0.f;                                                         \
      v.z = sh_ == 0 ? a.z : sh_ == 1 ? a.w : 0.f;                                                                          \
      v.w = sh_ == 0 ? a.w : 0.f;                                                                                           \
      const float4* sp = reinterpret_cast<const float4*>(St + (rep ? su1 : su0) + k_);                                      \
      const float4 s0 = sp[0], s1 = sp[1];                                                                                  \
      v = make_float4((v.x - s0.x) * s0.y, (v.y - s0.z) * s0.w, (v.z - s1.x) * s1.y, (v.w - s1.z) * s1.w);                  \
      uint16_t p0[4], p1[4], p2[4];                                                                                         \
      split3(v.x, p0[0], p1[0], p2[0]);                                                                                     \
      split3(v.y, p0[1], p1[1], p2[1]);                                                                                     \
      split3(v.z, p0[2], p1[2], p2[2]);                                                                                     \
      split3(v.w, p0[3], p1[3], p2[3]);                                                                                     \
      char* d = Ab + ((c4 >> 1) * kHM + fr + 64 * rep) * 16 + (c4 & 1) * 8;                                                 \
      *reinterpret_cast<uint2*>(d) = make_uint2(p0[0] | ((uint32_t)p0[1] << 16), p0[2] | ((uint32_t)p0[3] << 16));          \
      *reinterpret_cast<uint2*>(d + 2 * kHM * 16) = make_uint2(p1[0] | ((uint32_t)p1[1] << 16), p1[2] | ((uint32_t)p1[3] << 16)); \
      *reinterpret_cast<uint2*>(d + 4 * kHM * 16) = make_uint2(p2[0] | ((uint32_t)p2[1] << 16), p2[2] | ((uint32_t)p2[3] << 16)); \
    }                                                                                                                       \
    __syncthreads();                                                                                                        \
    if ((c) + 1 < nchunk) SE5_DMA_W((c) + 1);                                                                               \
    if ((c) + 3 < nchunk) SE5_LOAD_F((c) + 3, S);                                                                           \
    const char* ap = Ab + (wave * 32 + (lane & 31)) * 16 + (lane >> 5) * (kHM * 16);                                        \
    const char* wp = Ws + ((c) & 1) * kWBuf + (lane & 31) * 16 + (lane >> 5) * (NT * 32 * 16);                              \
    const h3_bf16x8 a0 = *reinterpret_cast<const h3_bf16x8*>(ap), a1 = *reinterpret_cast<const h3_bf16x8*>(ap + 2 * kHM * 16), \
                    a2 = *reinterpret_cast<const h3_bf16x8*>(ap + 4 * kHM * 16);                                            \
    _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                                                        \
      const char* wt = wp + t * 32 * 16;                                                                                    \
      const h3_bf16x8 w0 = *reinterpret_cast<const h3_bf16x8*>(wt), w1 = *reinterpret_cast<const h3_bf16x8*>(wt + 2 * NT * 32 * 16), \
                      w2 = *reinterpret_cast<const h3_bf16x8*>(wt + 4 * NT * 32 * 16);                                      \
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, w0, acc[t], 0, 0, 0);      /* smallest terms first, as head3_kernel */ \
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, w2, acc[t], 0, 0, 0);                                            \
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, w1, acc[t], 0, 0, 0);                                            \
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, w0, acc[t], 0, 0, 0);                                            \
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, w1, acc[t], 0, 0, 0);                                            \
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, w0, acc[t], 0, 0, 0);                                            \
    }                                                                                                                       \
  } while (0)

  for (int c = 0; c < nchunk; c += 3) {
    SE5_CHUNK(c, 0);
    if (c + 1 < nchunk) SE5_CHUNK(c + 1, 1);
    if (c + 2 < nchunk) SE5_CHUNK(c + 2, 2);
  }
#undef SE5_CHUNK
#undef SE5_DMA_W
#undef SE5_LOAD_F

  // ---- epilogue (accumulator map and LDS slab as head3_kernel; 16-B global accesses only: the launcher checked the alignment)
  __syncthreads();
#define SE5_LDS_WAVE_SYNC()                                   \
  do {                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        \
    __builtin_amdgcn_wave_barrier();                          \
    asm volatile("" ::: "memory");                            \
  } while (0)
  float* Es = reinterpret_cast<float*>(smem5) + wave * (kHHalf * NT * 32);
  float bn[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int n = t * 32 + (lane & 31);
    bn[t] = (bias && n < N) ? bias[n] : 0.f;
  }
  // SIS: valid frames of the tile's two utterances, and the running sums {<s,t>, <t,t>, <s,s>} of each
  int len0 = 0, len1 = 0;
  double sa[3] = {0.0, 0.0, 0.0}, sb[3] = {0.0, 0.0, 0.0};
  if constexpr (SIS) {
    const int nutt = (rows + F - 1) / F;
    auto frames = [&](int b) {
      if (b >= nutt) return 0;
      const int wl = (int)min(max(lengths[b], (int64_t)0), (int64_t)0x7fffffff);
      return min(F, len_div > 0 ? wl / len_div + 1 : wl);
    };
    len0 = frames(b0);
    len1 = frames(b0 + 1);
  }
  const uint32_t magic = (1u << 20) / (uint32_t)N + 1u;             // e / N == (e * magic) >> 20 for e < 2 048, N <= 256
  (void)magic;
#pragma unroll
  for (int h = 0; h < 32 / kHHalf; ++h) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int n = t * 32 + (lane & 31);
      if (n < N) {
#pragma unroll
        for (int r8 = 0; r8 < kHHalf / 2; ++r8) {
          const int r = (kHHalf / 2) * h + r8;
          const int rl = (r & 3) + (kHHalf == 16 ? 8 * ((r >> 2) & 1) : 0) + 4 * (lane >> 5);
          Es[rl * N + n] = act_c<ACT>(acc[t][r] + bn[t]);
        }
      }
    }
    SE5_LDS_WAVE_SYNC();
    const int rbase = row0 + wave * 32 + kHHalf * h;
    const int nrow = min(kHHalf, rows - rbase);
    if (nrow > 0) {
      const int cnt = nrow * N;
      const size_t g = (size_t)rbase * N;
      const int nvec = cnt >> 2;
      // SIS: element e of the pass belongs to utterance b0 while e < eb, to b0 + 1 from there on; valid while e < lim0 resp. eb <= e < lim1
      int eb = 0, lim0 = 0, lim1 = 0;
      if constexpr (SIS) {
        eb = min(max(boundary - rbase, 0), nrow) * N;
        lim0 = min(max(len0 - (rbase - b0 * F), 0) * N, eb);
        lim1 = eb + min(max(len1 - max(rbase - boundary, 0), 0), nrow) * N;
        lim1 = min(lim1, cnt);
      }
      // a pass's <= 28 elements per lane are summed in fp32 (hardware square roots, masks as selects: ~12 issue slots per element; the fp64 form of
      // sisdr_spec_slab_kernel -- two precise square roots, three fp64 products and six fp64 fused adds per element -- made this launch 300 us where
      // head + criterion as two launches take 260), the passes in fp64
      float pa[3] = {0.f, 0.f, 0.f}, pb[3] = {0.f, 0.f, 0.f};
      const bool straddle = SIS && eb < cnt && eb > 0;
#pragma unroll
      for (int j = 0; j < kVecIt; ++j) {
        const int i = lane + 64 * j;
        const float4 l = lin[j];
        float4 tq = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (SIS) tq = tv[j];
        lin[j] = load_item(linears, h + 1, j, 1.f);
        if constexpr (SIS) tv[j] = load_item(tar, h + 1, j, 0.f);
        if (i < nvec) {
          const float4 o = *reinterpret_cast<const float4*>(Es + 4 * i);
          if (offset) *reinterpret_cast<float4*>(offset + g + 4 * i) = o;
          const float4 p = make_float4(o.x * l.x, o.y * l.y, o.z * l.z, o.w * l.w);
          if (predicted) *reinterpret_cast<float4*>(predicted + g + 4 * i) = p;
          if constexpr (SIS) {
            const float pe[4] = {p.x, p.y, p.z, p.w}, te[4] = {tq.x, tq.y, tq.z, tq.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int e = 4 * i + q;
              const float sv = __builtin_amdgcn_sqrtf(fmaxf(pe[q], 0.f)), yv = __builtin_amdgcn_sqrtf(fmaxf(te[q], 0.f));
              // eb == 0: the whole pass belongs to the second utterance, eb == cnt: to the first
              const bool in0 = e < lim0;
              const float s0 = in0 ? sv : 0.f, y0 = in0 ? yv : 0.f;
              pa[0] = fmaf(s0, y0, pa[0]); pa[1] = fmaf(y0, y0, pa[1]); pa[2] = fmaf(s0, s0, pa[2]);
              if (eb < cnt) {
                const bool in1 = e >= eb && e < lim1;
                const float s1 = in1 ? sv : 0.f, y1 = in1 ? yv : 0.f;
                pb[0] = fmaf(s1, y1, pb[0]); pb[1] = fmaf(y1, y1, pb[1]); pb[2] = fmaf(s1, s1, pb[2]);
              }
            }
          }
        }
      }
      (void)straddle;
      for (int i = 4 * nvec + lane; i < cnt; i += 64) {               // a ragged last tile only (cnt % 4 != 0)
        const float o = Es[i];
        if (offset) offset[g + i] = o;
        const float p = linears ? linears[g + i] * o : o;
        if (predicted) predicted[g + i] = p;
        if constexpr (SIS) {
          const float sv = __builtin_amdgcn_sqrtf(fmaxf(p, 0.f)), yv = __builtin_amdgcn_sqrtf(fmaxf(tar[g + i], 0.f));
          if (i < lim0) { pa[0] = fmaf(sv, yv, pa[0]); pa[1] = fmaf(yv, yv, pa[1]); pa[2] = fmaf(sv, sv, pa[2]); }
          else if (i >= eb && i < lim1) { pb[0] = fmaf(sv, yv, pb[0]); pb[1] = fmaf(yv, yv, pb[1]); pb[2] = fmaf(sv, sv, pb[2]); }
        }
      }
      if constexpr (SIS) {
#pragma unroll
        for (int q = 0; q < 3; ++q) { sa[q] += (double)pa[q]; sb[q] += (double)pb[q]; }
      }
    }
    SE5_LDS_WAVE_SYNC();
  }
#undef SE5_LDS_WAVE_SYNC
  if constexpr (SIS) {
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        sa[q] += __shfl_xor(sa[q], off);
        sb[q] += __shfl_xor(sb[q], off);
      }
    __syncthreads();                                                 // every wave is past its slab
    double* red = reinterpret_cast<double*>(smem5);                  // [4 waves][6]
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < 3; ++q) { red[wave * 6 + q] = sa[q]; red[wave * 6 + 3 + q] = sb[q]; }
    }
    __syncthreads();
    if (tid < 6) slab[(size_t)blockIdx.x * 6 + tid] = (red[tid] + red[6 + tid]) + (red[12 + tid] + red[18 + tid]);
  }
}

}  // namespace se

// internal (not part of the public header): column statistics shared with head_bwd.hip
extern "C" int se_head_colstats_f32(const float* feats, int B, int F, int D, float eps, float* stats, void* stream) {
  SE_REQUIRE(feats && stats && B > 0 && B <= 65535 && F >= 2 && D > 0, "se_head_colstats_f32: bad argument (B=%d F=%d D=%d)", B, F, D);
  hipLaunchKernelGGL(se::colstats_kernel, dim3((D + 63) / 64, B), dim3(256), 0, se::as_stream(stream), feats, F, D, eps, stats);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" size_t se_head_workspace_bytes(int B, int F, int D, int N) {
  (void)F;
  // [stats: B*D*2 floats][three-term bf16 split of the weights: 3 planes x ceil32(N) x ceil16(D) bf16][bwd scratch: g_pre (B*F*N) is NOT kept
  // here; see se_head_linear_bwd_f32]
  const size_t rows_p = (size_t)((N + 31) / 32) * 32, Kp = (size_t)((D + 15) / 16) * 16;
  return (((size_t)B * D * 2 * sizeof(float) + 255) & ~(size_t)255) + 3 * rows_p * Kp * sizeof(uint16_t) + 256;
}

template <int NT>
static int launch_head3(const float* feats, const uint16_t* W3, int Kp, const float* bias, const float* linears, const float* stats,
                        int rows, int F, int D, int N, int act, float* predicted, float* offset, hipStream_t st) {
  const int vec_io = ((((uintptr_t)predicted | (uintptr_t)offset | (uintptr_t)linears) % 16) == 0) ? 1 : 0;
  se::ProfScope prof(se::kProfHead, 4.0 * rows * ((double)D + (linears ? N : 0) + (predicted ? N : 0) + (offset ? N : 0)), st);
  hipLaunchKernelGGL((se::head3_kernel<NT>), dim3((rows + se::kHM - 1) / se::kHM), dim3(256), 0, st, feats, W3, Kp, bias, linears,
                     stats, rows, F, D, N, act, predicted, offset, vec_io);
  SE_LAUNCH_CHECK();
  return SE_OK;
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
  SE_REQUIRE(workspace && workspace_bytes >= se_head_workspace_bytes(B, F, D, N), "se_head_linear_f32: workspace too small");
  if (cmvn) {
    stats = reinterpret_cast<float*>(workspace);
    int rc = se_head_colstats_f32(feats, B, F, D, eps, stats, stream);
    if (rc) return rc;
  }
  const int rows = B * F;
  const int nt = (N + 31) / 32;
  static const bool fp32_mfma = getenv("SE_AMD_HEAD_F32MFMA") != nullptr;      // developer A/B: the round-3 kernel on v_mfma_f32_32x32x2_f32
  if (!fp32_mfma) {
    const int rows_p = nt * 32, Kp = (D + 15) / 16 * 16;
    uint16_t* W3 = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(workspace) + (((size_t)B * D * 2 * sizeof(float) + 255) & ~(size_t)255));
    hipLaunchKernelGGL(se::head_split_w_kernel, dim3((rows_p * Kp + 255) / 256), dim3(256), 0, st, W, N, D, rows_p, Kp, W3);
    SE_LAUNCH_CHECK();
    switch (nt) {
      case 1: return launch_head3<1>(feats, W3, Kp, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
      case 2: return launch_head3<2>(feats, W3, Kp, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
      case 3: return launch_head3<3>(feats, W3, Kp, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
      case 4: return launch_head3<4>(feats, W3, Kp, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
      case 5: return launch_head3<5>(feats, W3, Kp, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
      case 6: return launch_head3<6>(feats, W3, Kp, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
      case 7: return launch_head3<7>(feats, W3, Kp, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
      default: return launch_head3<8>(feats, W3, Kp, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
    }
  }
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


// ---- round 5: the evaluate()-style pass calls the head once per batch with the SAME weights and with column statistics that the feature launch
// has already produced (se_features3_f32's colstats_out: every feature row is in LDS there).  se_head_linear_f32 re-splits the weights and
// re-reads the features for the statistics on every call (two launches + 123 MB at 256 utterances); these entry points take both ready-made.
extern "C" size_t se_head_w3_bytes(int N, int D) {
  const size_t rows_p = (size_t)((N + 31) / 32) * 32, Kp = (size_t)((D + 15) / 16) * 16;
  return 3 * rows_p * Kp * sizeof(uint16_t);
}

extern "C" int se_head_split_weights_f32(const float* W, int N, int D, uint16_t* W3, void* stream) {
  SE_REQUIRE(W && W3 && N > 0 && N <= 256 && D > 0, "se_head_split_weights_f32: bad argument (N=%d D=%d)", N, D);
  const int rows_p = (N + 31) / 32 * 32, Kp = (D + 15) / 16 * 16;
  hipLaunchKernelGGL(se::head_split_w_kernel, dim3((rows_p * Kp + 255) / 256), dim3(256), 0, se::as_stream(stream), W, N, D, rows_p, Kp, W3);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

// the branch-free kernel's preconditions: N in (192, 224] (7 column tiles: the reference's 201 bins), a compile-time activation it was instantiated for,
// F >= 128 (a tile touches at most two utterances), at most 256 staged statistics columns, 16-B aligned planes, D >= 4
static bool head5_ok(const float* linears, const float* predicted, const float* offset, const float* tar, int F, int D, int N, int act) {
  static const bool on = getenv("SE_AMD_HEAD5") == nullptr || atoi(getenv("SE_AMD_HEAD5")) != 0;      // 0: always head3_kernel (A/B)
  const int Kp = (D + 15) / 16 * 16;
  return on && (N + 31) / 32 == 7 && (act == SE_ACT_SIGMOID || act == SE_ACT_RELU) && F >= se::kHM && Kp <= se::kH5StatCols && D >= 4 &&
         ((((uintptr_t)predicted | (uintptr_t)offset | (uintptr_t)linears | (uintptr_t)tar) % 16) == 0) && (N * 8 * 4) % 16 == 0;
}

static int launch_head5(const float* feats, const uint16_t* W3, const float* bias, const float* linears, const float* stats, int rows, int F, int D, int N,
                        int act, float* predicted, float* offset, const float* tar, const int64_t* lengths, int len_div, double* slab, hipStream_t st) {
  const int Kp = (D + 15) / 16 * 16;
  se::ProfScope prof(se::kProfHead, 4.0 * rows * ((double)D + (linears ? N : 0) + (predicted ? N : 0) + (offset ? N : 0) + (tar ? N : 0)), st);
  const dim3 grid((rows + se::kHM - 1) / se::kHM), block(256);
  if (tar)
    hipLaunchKernelGGL((se::head5_kernel<7, SE_ACT_SIGMOID, 1>), grid, block, 0, st, feats, W3, Kp, bias, linears, stats, rows, F, D, N, predicted, offset, tar,
                       lengths, len_div, slab);
  else if (act == SE_ACT_SIGMOID)
    hipLaunchKernelGGL((se::head5_kernel<7, SE_ACT_SIGMOID, 0>), grid, block, 0, st, feats, W3, Kp, bias, linears, stats, rows, F, D, N, predicted, offset, nullptr,
                       nullptr, 0, nullptr);
  else
    hipLaunchKernelGGL((se::head5_kernel<7, SE_ACT_RELU, 0>), grid, block, 0, st, feats, W3, Kp, bias, linears, stats, rows, F, D, N, predicted, offset, nullptr,
                       nullptr, 0, nullptr);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_head_linear_pre_f32(const float* feats, const uint16_t* W3, const float* bias, const float* linears, const float* stats,
                                      int B, int F, int D, int N, int act, float* predicted, float* offset, void* stream) {
  SE_REQUIRE(feats && W3 && (predicted || offset), "se_head_linear_pre_f32: null argument");
  SE_REQUIRE(B > 0 && B <= 65535 && F >= 2 && D > 0 && N > 0 && N <= 256, "se_head_linear_pre_f32: bad shape B=%d F=%d D=%d N=%d (N <= 256)", B, F, D, N);
  hipStream_t st = se::as_stream(stream);
  const int rows = B * F, nt = (N + 31) / 32, Kp = (D + 15) / 16 * 16;
  if (head5_ok(linears, predicted, offset, nullptr, F, D, N, act))
    return launch_head5(feats, W3, bias, linears, stats, rows, F, D, N, act, predicted, offset, nullptr, nullptr, 0, nullptr, st);
  switch (nt) {
    case 1: return launch_head3<1>(feats, W3, Kp, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
    case 2: return launch_head3<2>(feats, W3, Kp, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
    case 3: return launch_head3<3>(feats, W3, Kp, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
    case 4: return launch_head3<4>(feats, W3, Kp, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
    case 5: return launch_head3<5>(feats, W3, Kp, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
    case 6: return launch_head3<6>(feats, W3, Kp, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
    case 7: return launch_head3<7>(feats, W3, Kp, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
    default: return launch_head3<8>(feats, W3, Kp, bias, linears, stats, rows, F, D, N, act, predicted, offset, st);
  }
}

// The evaluate()-style pass of a mask head scored by objective.SISDR (runner.py:556-575 with vcb.yaml / pseudo_noise.yaml): se_head_linear_pre_f32 AND
// se_sisdr_spec_loss_f32 on its `predicted`, the criterion's sums taken from the products as they leave the head's registers (one extra read of
// linear_tar instead of a launch that reads predicted and linear_tar).  Shapes the fused kernel does not take run the two entry points one after the other.
extern "C" int se_sisdr_head_mean_f32(const double* slab, int B, int F, int tile_rows, float eps, float* loss_b, double* sums_out, float* loss_out, void* stream);

extern "C" size_t se_head_sisdr_scratch_doubles(int B, int F, int N) {
  const size_t wgs = ((size_t)B * F + se::kHM - 1) / se::kHM;
  return std::max(wgs * 6, se_sisdr_spec_loss_scratch_doubles(B, F, N));
}

extern "C" int se_head_linear_sisdr_f32(const float* feats, const uint16_t* W3, const float* bias, const float* linears, const float* stats,
                                        int B, int F, int D, int N, int act, float* predicted, float* offset,
                                        const float* linear_tar, const int64_t* lengths, int len_div, float eps, double* scratch,
                                        float* loss_b, double* sums_out, float* loss_out, void* stream) {
  SE_REQUIRE(feats && W3 && predicted && linear_tar && lengths && scratch && loss_b && sums_out && loss_out, "se_head_linear_sisdr_f32: null argument");
  SE_REQUIRE(B > 0 && B <= 65535 && F >= 2 && D > 0 && N > 0 && N <= 256 && len_div >= 0, "se_head_linear_sisdr_f32: bad shape B=%d F=%d D=%d N=%d", B, F, D, N);
  hipStream_t st = se::as_stream(stream);
  if (act == SE_ACT_SIGMOID && head5_ok(linears, predicted, offset, linear_tar, F, D, N, act)) {
    const int rc = launch_head5(feats, W3, bias, linears, stats, B * F, F, D, N, act, predicted, offset, linear_tar, lengths, len_div, scratch, st);
    if (rc) return rc;
    return se_sisdr_head_mean_f32(scratch, B, F, se::kHM, eps, loss_b, sums_out, loss_out, stream);
  }
  const int rc = se_head_linear_pre_f32(feats, W3, bias, linears, stats, B, F, D, N, act, predicted, offset, stream);
  if (rc) return rc;
  return se_sisdr_spec_loss_f32(predicted, linear_tar, lengths, len_div, B, F, N, eps, scratch, loss_b, sums_out, loss_out, stream);
}
