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

  // ---- epilogue: as head_kernel (same accumulator map)
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
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
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
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
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

extern "C" int se_head_linear_pre_f32(const float* feats, const uint16_t* W3, const float* bias, const float* linears, const float* stats,
                                      int B, int F, int D, int N, int act, float* predicted, float* offset, void* stream) {
  SE_REQUIRE(feats && W3 && (predicted || offset), "se_head_linear_pre_f32: null argument");
  SE_REQUIRE(B > 0 && B <= 65535 && F >= 2 && D > 0 && N > 0 && N <= 256, "se_head_linear_pre_f32: bad shape B=%d F=%d D=%d N=%d (N <= 256)", B, F, D, N);
  hipStream_t st = se::as_stream(stream);
  const int rows = B * F, nt = (N + 31) / 32, Kp = (D + 15) / 16 * 16;
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
