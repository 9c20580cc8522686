// gemm4.hip -- row-complete GEMM + bias + fp32 residual + LayerNorm for the encoder's H = 768 projections
// (attention output and FFN2: rows B2 / B3 "Linear -> +residual -> LayerNorm"):
//      x = LayerNorm(A[M,K] . W[768,K]^T + bias + residual) * ln_w + ln_b   ->  x_f32 (M,768) and x_bf16 (M,768)
//
// Why: with N = 768 a 256 x 256 tiling gives 378 workgroups (1.5 rounds of 256 CUs) and the LayerNorm needs whole rows,
// so the unfused path writes an fp32 (M,768) tensor, reads it back in a separate LayerNorm launch and runs its GEMM at
// 0.5-0.7 PFLOP/s.  Here one workgroup owns 128 COMPLETE rows: M = 32 032 -> 251 workgroups ~ one per CU (no wave
// quantisation), the LayerNorm runs on the accumulators, and 196 MB of HBM traffic + one launch disappear per call.
//
//   tile     : 128 x 768, 32-deep K-steps; 512 threads = 8 waves as 2 (M) x 4 (N); wave tile 64 x 192 = 4 x 12 MFMA tiles
//              (v_mfma_f32_16x16x32_bf16, 192 accumulator registers)
//   LDS      : the weight K-step (768 x 32 bf16 = 48 KiB) does not fit a 3-deep ring, so it is split by the three
//              64-column groups a wave walks: a ring of SIX 16 KiB weight pieces refilled at sub-step granularity (five
//              pieces in flight), plus a 4-deep ring of 8 KiB A pieces: 96 + 32 KiB; 64-B rows, chunk ^ ((-(row>>2))&3)
//   schedule : per K-step three {LDS reads + waits | barrier | 16 MFMAs with the ring-refill DMAs between them | barrier} sub-steps,
//              waves 4-7 one barrier behind waves 0-3 (ping-pong); counted vmcnt (6-8 DMAs younger than the piece needed next)
//   epilogue : swapped operands (C^T accumulators, initialised with bias + residual before the K loop); ln_w / ln_b staged
//              in the (then dead) ring; exact two-pass mean / variance per row (lane group -> wave -> LDS across the 4 column waves);
//              fp32 and bf16 rows stored straight from registers
#include <stdlib.h>
#include "clkprobe.h"
#include "common.h"
#include "bf16.h"
#include "prof.h"
#include "encoder_impl.h"

SE_CLKPROBE_DECL(clkprobe_gemm7)
// -DSE7_ABL=<mask> (gemm7_res_ln_kernel, timing only, results wrong): 1 no LDS-DMA, 2 no s_barrier in the K loop, 4 no LDS fragment reads, 8 no epilogue
// (LayerNorm + stores), 16 no residual / bias read in front of the K loop
#ifndef SE7_ABL
#define SE7_ABL 0
#endif
namespace se {

constexpr int k4BM = 128, k4N = 768, k4BK = 32, k4Threads = 512;
constexpr int k4WPiece = 256 * 64, k4APiece = 128 * 64, k4ASlots = 4;
constexpr int k4ABytes = k4APiece * k4ASlots;                 // 32 768
constexpr int k4_lds(int slots) { return k4WPiece * slots + k4ABytes + 2 * 2 * 4 * 64 * 4; }   // ring + A ring + LN partial sums (4 KiB)

typedef __attribute__((address_space(3))) void* lds4_ptr_t;
typedef const __attribute__((address_space(1))) void* glb4_ptr_t;

__device__ __forceinline__ int swz4_f(int row) { return (-(row >> 2)) & 3; }
__device__ __forceinline__ int swz4(int row, int chunk) { return row * 64 + ((chunk ^ swz4_f(row)) << 4); }

// SLOTS: weight-ring depth (pieces; SLOTS - 1 in flight).  INM: the ring-refill DMAs are issued between the MFMAs (1) or in the read phase (0)
// ---- 24-bit residual stream between the row-complete launches of one encoder pass: value = bf16 hi (the x_bf16 tensor the next GEMM
//      reads anyway, round-to-nearest) + int8 lo = round((y - hi) * 2^(15 - E)), E = exponent of hi: 16 mantissa bits (2^-17 relative; the
//      encoder's error against fp32 is unchanged: 3.177e-3 vs 3.172e-3 relative L2) for 3 B per element instead of 4 + 2.  lo lives in the
//      accumulator's own layout (one contiguous 256-B block per wave, row tile and column tile): written and read by the same lane.
// Round 2 (end): the low byte is defined ON THE BIT PATTERN -- lo = round((bits(y) - bits(hi) << 16) / 256), i.e. bits 15..8 of the fp32 pattern
// as a signed correction to the round-to-nearest bf16 -- instead of through a float scale 2^(15 - E).  Same resolution (1/256 of hi's bf16
// ulp = 2^8 fp32 ulps; numpy emulation on 2 M values: mean 5.5e-6, max 3.0e-5 relative, the maximum when the byte saturates at +127 for a y
// exactly half way that rounded down -- as in the first format), but decoding is ONE integer add per element and encoding three:
//   dec:  bits = (hi << 16) + (lo << 8)                      enc:  lo = min(((bits(y) - (hi << 16)) + 128) >> 8, 127)
// (sign-magnitude patterns: y and hi always share the sign, so the integer difference is the signed difference of the magnitudes and the add
// moves the magnitude the same way).  This codec runs on all 192 accumulators of a wave at both ends of every row-complete launch: the
// float-select form cost ~24 vector instructions per element over both directions, the bit-field float form ~16, this one ~9.
#ifdef SE_AMD_OLD_CODEC      // A/B build switch (SE_AMD_EXTRA_DEFINES=-DSE_AMD_OLD_CODEC python build.py --force): the first, float-scale format
__device__ __forceinline__ float dec24(uint32_t hi16, int lo8) {
  const uint32_t eb = (hi16 >> 7) & 0xffu;
  const float sc = __uint_as_float(eb > 15u ? (eb - 15u) << 23 : 0u);
  return fmaf((float)lo8, sc, __uint_as_float(hi16 << 16));
}
__device__ __forceinline__ uint32_t enc24_lo(float y, uint32_t hi16) {
  const uint32_t eb = (hi16 >> 7) & 0xffu;
  const float sc = __uint_as_float((eb >= 15u && eb <= 254u) ? (269u - eb) << 23 : 0u);
  const float q = rintf((y - __uint_as_float(hi16 << 16)) * sc);
  return (uint32_t)(int)fminf(fmaxf(q, -128.f), 127.f) & 0xffu;
}
__device__ __forceinline__ uint32_t enc24_lo4(float y0, float y1, float y2, float y3, uint2 pk) {
  return enc24_lo(y0, pk.x & 0xffffu) | (enc24_lo(y1, pk.x >> 16) << 8) | (enc24_lo(y2, pk.y & 0xffffu) << 16) | (enc24_lo(y3, pk.y >> 16) << 24);
}
#elif defined(SE_AMD_FLOAT_CODEC)      // A/B: the float-scale format computed on the exponent field (the intermediate version of round 2)
__device__ __forceinline__ float dec24(uint32_t hi16, int lo8) {
  const uint32_t hb = hi16 << 16;
  const uint32_t ef = max(hb & 0x7f800000u, 15u << 23);
  return fmaf((float)lo8, __uint_as_float(ef - (15u << 23)), __uint_as_float(hb));
}
__device__ __forceinline__ uint32_t enc24_lo4(float y0, float y1, float y2, float y3, uint2 pk) {
  const uint32_t hb[4] = {pk.x << 16, pk.x & 0xffff0000u, pk.y << 16, pk.y & 0xffff0000u};
  const float y[4] = {y0, y1, y2, y3};
  uint32_t m[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t ef = max(hb[r] & 0x7f800000u, 15u << 23);
    const float q = (y[r] - __uint_as_float(hb[r])) * __uint_as_float(0x86800000u - ef);
    m[r] = __float_as_uint(__builtin_amdgcn_fmed3f(q, -128.f, 127.f) + 12582912.0f);
  }
  const uint32_t t01 = __builtin_amdgcn_perm(m[1], m[0], 0x0c0c0400u), t23 = __builtin_amdgcn_perm(m[3], m[2], 0x0c0c0400u);
  return __builtin_amdgcn_perm(t23, t01, 0x05040100u);
}
#else
__device__ __forceinline__ float dec24(uint32_t hi16, int lo8) { return __uint_as_float((hi16 << 16) + (uint32_t)(lo8 << 8)); }
// the four lo bytes of (y0..y3) given their packed bf16 pairs (pk.x = hi(y0) | hi(y1) << 16, pk.y = hi(y2) | hi(y3) << 16), as one dword
__device__ __forceinline__ uint32_t enc24_lo4(float y0, float y1, float y2, float y3, uint2 pk) {
  const uint32_t hb[4] = {pk.x << 16, pk.x & 0xffff0000u, pk.y << 16, pk.y & 0xffff0000u};
  const float y[4] = {y0, y1, y2, y3};
  uint32_t m[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) m[r] = (uint32_t)min(((int)(__float_as_uint(y[r]) - hb[r]) + 128) >> 8, 127);
  const uint32_t t01 = __builtin_amdgcn_perm(m[1], m[0], 0x0c0c0400u), t23 = __builtin_amdgcn_perm(m[3], m[2], 0x0c0c0400u);
  return __builtin_amdgcn_perm(t23, t01, 0x05040100u);
}
#endif

// acc[i][t] <- bias (+ residual): fp32 rows, or the 24-bit (bf16 hi row-major + int8 lo tile-major) stream, or a (T, 768) table indexed row % res_mod
template <int GELU, int RIN, int LN = 1>
__device__ __forceinline__ void acc_init4(f32x4 (&acc)[4][12], const float* __restrict__ bias, const float* __restrict__ residual,
                                          const uint8_t* __restrict__ res_lo, int M, int m0, int id, int wave, int wr, int lane, int col0, int res_mod) {
  const int mrow = lane & 15;
  float4 bb[12];
#pragma unroll
  for (int t = 0; t < 12; ++t) bb[t] = bias ? *reinterpret_cast<const float4*>(bias + col0 + 16 * t) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gm = min(m0 + wr * 64 + i * 16 + mrow, M - 1);
    const float* rp = residual + (size_t)(res_mod > 0 ? gm % res_mod : gm) * k4N + col0;      // res_mod = T: a (T, 768) positional table as the residual
#pragma unroll
    for (int t = 0; t < 12; ++t) {
      if (GELU || (!LN && !residual)) {      // LN = 0: the plain projection; its residual is optional
        acc[i][t] = (f32x4){bb[t].x, bb[t].y, bb[t].z, bb[t].w};
      } else if (RIN) {
        const uint2 h4 = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(residual) + (size_t)gm * k4N + col0 + 16 * t);
        const uint32_t l4 = *reinterpret_cast<const uint32_t*>(res_lo + ((((size_t)id * 8 + wave) * 4 + i) * 12 + t) * 256 + lane * 4);
        acc[i][t] = (f32x4){dec24(h4.x & 0xffffu, (int)(int8_t)(l4 & 0xffu)) + bb[t].x, dec24(h4.x >> 16, (int)(int8_t)((l4 >> 8) & 0xffu)) + bb[t].y,
                            dec24(h4.y & 0xffffu, (int)(int8_t)((l4 >> 16) & 0xffu)) + bb[t].z, dec24(h4.y >> 16, (int)(int8_t)(l4 >> 24)) + bb[t].w};
      } else {
        const float4 rr = *reinterpret_cast<const float4*>(rp + 16 * t);
        acc[i][t] = (f32x4){rr.x + bb[t].x, rr.y + bb[t].y, rr.z + bb[t].z, rr.w + bb[t].w};
      }
    }
  }
}

// (optional gelu) + LayerNorm on the accumulators (which hold A.W^T + bias + residual) + the stores.  Called with every wave past its last
// ring access (the caller's __syncthreads): the ring is dead, its first 6 KiB take ln_w / ln_b, `red_off` + 4 KiB the row partial sums.
// X3OUT: out_bf16 is the NEXT bf16x3 projection's three-slice operand [y1 | y1 | y2] (row stride 3 x 768; se_split3_bf16's layout), beside the fp32 rows
template <int GELU, int ROUT, int LN = 1, int X3OUT = 0>
__device__ __forceinline__ void ln_epilogue4(f32x4 (&acc)[4][12], char* smem, int red_off, const float* __restrict__ ln_w, const float* __restrict__ ln_b,
                                             float eps, int M, int m0, int id, int wave, int wr, int wc, int lane, float* __restrict__ out_f32,
                                             uint16_t* __restrict__ out_bf16, uint8_t* __restrict__ out_lo) {
  const int tid = wave * 64 + lane;
  const int mrow = lane & 15, cq = lane >> 4;
  const int col0 = wc * 192 + 4 * cq;
  if (GELU) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int t = 0; t < 12; ++t) {
        const f32x2 ga = gelu_erf2((f32x2){acc[i][t][0], acc[i][t][1]}), gb = gelu_erf2((f32x2){acc[i][t][2], acc[i][t][3]});
        acc[i][t] = (f32x4){ga.x, ga.y, gb.x, gb.y};
      }
  }
  // ---- epilogue: LayerNorm on the accumulators (which already hold A.W^T + bias + residual); LN = 0: the accumulators are the result
  float* colv = reinterpret_cast<float*>(smem);             // [2][768]: ln_w, ln_b staged in the dead ring
  float mean[4], rstd[4];
  if constexpr (LN) {
    for (int c = tid; c < k4N; c += k4Threads) {
      colv[c] = ln_w[c];
      colv[k4N + c] = ln_b[c];
    }
    float* red = reinterpret_cast<float*>(smem + red_off);   // [pass][wr][wc][64]
    // pass 1: row means
  #pragma unroll
    for (int i = 0; i < 4; ++i) {
      float s = 0.f;
  #pragma unroll
      for (int t = 0; t < 12; ++t) s += (acc[i][t][0] + acc[i][t][1]) + (acc[i][t][2] + acc[i][t][3]);
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      if (cq == 0) red[(wr * 4 + wc) * 64 + i * 16 + mrow] = s;
    }
    __syncthreads();
  #pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float* rp = red + wr * 256 + i * 16 + mrow;
      mean[i] = (rp[0] + rp[64] + rp[128] + rp[192]) * (1.0f / k4N);
    }
    // pass 2: variance around the mean (exact two-pass, as the stand-alone LayerNorm kernel)
  #pragma unroll
    for (int i = 0; i < 4; ++i) {
      float q = 0.f;
  #pragma unroll
      for (int t = 0; t < 12; ++t)
  #pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = acc[i][t][r] - mean[i];
          q = fmaf(d, d, q);
        }
      q += __shfl_xor(q, 16);
      q += __shfl_xor(q, 32);
      if (cq == 0) red[512 + (wr * 4 + wc) * 64 + i * 16 + mrow] = q;
    }
    __syncthreads();
  #pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float* rp = red + 512 + wr * 256 + i * 16 + mrow;
      rstd[i] = 1.0f / sqrtf((rp[0] + rp[64] + rp[128] + rp[192]) * (1.0f / k4N) + eps);
    }
  }
  // normalise + store (stores only in this loop; the per-column vectors come from LDS)
  const bool interior = m0 + k4BM <= M;
  const bool godd = cq & 1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gm = m0 + wr * 64 + i * 16 + mrow;
    const bool ok = interior || gm < M;
    const size_t o = (size_t)min(gm, M - 1) * k4N + col0;
    const size_t orow8 = (size_t)min(gm, M - 1) * (X3OUT ? 3 * k4N : k4N) + wc * 192 + 4 * (cq & ~1);      // first of this lane's 8 consecutive bf16 columns
    uint2 pk_prev = make_uint2(0u, 0u), pm_prev = make_uint2(0u, 0u);
    const float nmr = LN ? -mean[i] * rstd[i] : 0.f;
#pragma unroll
    for (int t = 0; t < 12; ++t) {
      if ((t & 3) == 0) __builtin_amdgcn_sched_barrier(0);
      float y0 = acc[i][t][0], y1 = acc[i][t][1], y2 = acc[i][t][2], y3 = acc[i][t][3];
      if constexpr (LN) {
        const float4 lw = *reinterpret_cast<const float4*>(colv + col0 + 16 * t);
        const float4 lb = *reinterpret_cast<const float4*>(colv + k4N + col0 + 16 * t);
        // xhat = acc * rstd - mean * rstd as ONE fma (the per-row product is formed once), then lw * xhat + lb: two instructions per element
        y0 = fmaf(lw.x, fmaf(y0, rstd[i], nmr), lb.x);
        y1 = fmaf(lw.y, fmaf(y1, rstd[i], nmr), lb.y);
        y2 = fmaf(lw.z, fmaf(y2, rstd[i], nmr), lb.z);
        y3 = fmaf(lw.w, fmaf(y3, rstd[i], nmr), lb.w);
      }
      if (!ROUT && ok && out_f32) *reinterpret_cast<float4*>(out_f32 + o + 16 * t) = make_float4(y0, y1, y2, y3);
      if (out_bf16) {
        // 16-B bf16 stores (as gemm3.hip): lanes l, l ^ 16 trade 4-column pieces of the MFMA tile pair (t - 1, t), so a lane owns 8
        // consecutive columns and one wave instruction writes 16 rows x 64 contiguous bytes instead of 16 x 32
        const uint2 pk = make_uint2(pack_bf16x2(y0, y1), pack_bf16x2(y2, y3));
        if (ROUT)      // tile-major lo bytes: rows past M are written too (the buffer covers whole tiles) and never read as real rows
          *reinterpret_cast<uint32_t*>(out_lo + ((((size_t)id * 8 + wave) * 4 + i) * 12 + t) * 256 + lane * 4) =
              enc24_lo4(y0, y1, y2, y3, pk);
        uint2 pm = make_uint2(0u, 0u);
        if constexpr (X3OUT)        // the residual term y2 = bf16(y - y1)
          pm = make_uint2(pack_bf16x2(y0 - __uint_as_float(pk.x << 16), y1 - __uint_as_float(pk.x & 0xffff0000u)),
                          pack_bf16x2(y2 - __uint_as_float(pk.y << 16), y3 - __uint_as_float(pk.y & 0xffff0000u)));
        if (t & 1) {
          const uint2 keep = godd ? pk : pk_prev, send = godd ? pk_prev : pk;
          uint2 recv;
          recv.x = __shfl_xor(send.x, 16);
          recv.y = __shfl_xor(send.y, 16);
          const uint4 o16 = godd ? make_uint4(recv.x, recv.y, keep.x, keep.y) : make_uint4(keep.x, keep.y, recv.x, recv.y);
          if (ok) *reinterpret_cast<uint4*>(out_bf16 + orow8 + 16 * (godd ? t : t - 1)) = o16;
          if constexpr (X3OUT) {
            if (ok) *reinterpret_cast<uint4*>(out_bf16 + orow8 + k4N + 16 * (godd ? t : t - 1)) = o16;
            const uint2 keepm = godd ? pm : pm_prev, sendm = godd ? pm_prev : pm;
            uint2 recvm;
            recvm.x = __shfl_xor(sendm.x, 16);
            recvm.y = __shfl_xor(sendm.y, 16);
            const uint4 o16m = godd ? make_uint4(recvm.x, recvm.y, keepm.x, keepm.y) : make_uint4(keepm.x, keepm.y, recvm.x, recvm.y);
            if (ok) *reinterpret_cast<uint4*>(out_bf16 + orow8 + 2 * k4N + 16 * (godd ? t : t - 1)) = o16m;
          }
        }
        pk_prev = pk;
        pm_prev = pm;
      }
    }
  }
}

// GELU: x = LayerNorm(gelu(A . W^T + bias)) without a residual (the spec head's dense -> act -> LayerNorm, model.py:100-101)
// RIN : the residual comes as (bf16 hi row-major, int8 lo tile-major) instead of fp32;  ROUT: the output goes out as (bf16, lo) instead of (fp32, bf16)
template <int SLOTS, int INM, int GELU, int RIN, int ROUT>
__global__ __launch_bounds__(k4Threads) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm4_res_ln_kernel(
    const uint16_t* __restrict__ A, int lda, const uint16_t* __restrict__ W, int ldw, const float* __restrict__ bias,
    const float* __restrict__ residual, const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps, int M, int K,
    float* __restrict__ out_f32, uint16_t* __restrict__ out_bf16, int ntiles, int res_mod, int stagger,
    const uint8_t* __restrict__ res_lo, uint8_t* __restrict__ out_lo) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int k4WSlots = SLOTS, k4WBytes = k4WPiece * SLOTS, k4RedOff = k4WBytes + k4ABytes;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  int id;
  {
    const int orig = blockIdx.x, xcd = orig & 7, q = ntiles >> 3, r = ntiles & 7;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  const int m0 = id * k4BM;
  // stagger: every second workgroup starts `stagger` x ~3.4 us late, so that the HBM-bound phases of the two halves (residual read at
  // the start, output write at the end) fall into each other's compute phase instead of all 251 workgroups hitting HBM together
  if (stagger > 0 && (blockIdx.x & 8))
    for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(127);

  // ---- DMA sources.  Weight piece (g, j) = rows {wcc*192 + j*64 + 0..63 : wcc = 0..3} = 16 chunks of 16 rows x 64 B;
  //      wave w issues chunks w and w+8.  chunk c -> wcc = c >> 2, piece-local rows 16 c + (lane >> 2).
  const int r16 = lane >> 2, pos = lane & 3;
  const uint16_t* w_src[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = wave + 8 * i, prow = 16 * c + r16;
    const int grow = (c >> 2) * 192 + (c & 3) * 16 + r16;          // + 64 j per piece
    w_src[i] = W + (size_t)grow * ldw + ((pos ^ swz4_f(prow)) << 3);
  }
  const uint16_t* a_src;
  {
    const int prow = 16 * wave + r16;
    a_src = A + (size_t)min(m0 + prow, M - 1) * lda + ((pos ^ swz4_f(prow)) << 3);
  }
  const size_t w_jstep = (size_t)64 * ldw;
#define SE4_ISSUE_W(g, j, slot)                                                                                              \
  do {                                                                                                                       \
    char* sb_ = smem + (slot) * k4WPiece + wave * 1024;                                                                      \
    __builtin_amdgcn_global_load_lds((glb4_ptr_t)(w_src[0] + (j) * w_jstep + (g) * k4BK), (lds4_ptr_t)(sb_), 16, 0, 0);      \
    __builtin_amdgcn_global_load_lds((glb4_ptr_t)(w_src[1] + (j) * w_jstep + (g) * k4BK), (lds4_ptr_t)(sb_ + 8192), 16, 0, 0); \
  } while (0)
#define SE4_ISSUE_W0(g, j, slot)                                                                                             \
  __builtin_amdgcn_global_load_lds((glb4_ptr_t)(w_src[0] + (j) * w_jstep + (g) * k4BK), (lds4_ptr_t)(smem + (slot) * k4WPiece + wave * 1024), 16, 0, 0)
#define SE4_ISSUE_W1(g, j, slot)                                                                                             \
  __builtin_amdgcn_global_load_lds((glb4_ptr_t)(w_src[1] + (j) * w_jstep + (g) * k4BK), (lds4_ptr_t)(smem + (slot) * k4WPiece + wave * 1024 + 8192), 16, 0, 0)
#define SE4_ISSUE_A(g)                                                                                                       \
  do {                                                                                                                       \
    __builtin_amdgcn_global_load_lds((glb4_ptr_t)(a_src + (g) * k4BK),                                                       \
                                     (lds4_ptr_t)(smem + k4WBytes + ((g) & 3) * k4APiece + wave * 1024), 16, 0, 0);         \
  } while (0)

  // ring prologue FIRST: pieces 0..4 = (0,0) (0,1) (0,2) (1,0) (1,1), A steps 0..2 -- the 393 KB of residual rows read below for the
  // accumulator initialisation then travel concurrently with it (one HBM latency instead of two before the first MFMA)
  SE4_ISSUE_A(0);
  SE4_ISSUE_W(0, 0, 0);
  SE4_ISSUE_A(1);
  SE4_ISSUE_A(2);
  SE4_ISSUE_W(0, 1, 1);
  SE4_ISSUE_W(0, 2, 2);
  SE4_ISSUE_W(1, 0, 3);
  SE4_ISSUE_W(1, 1, 4);
  if (SLOTS == 7) SE4_ISSUE_W(1, 2, 5);
  // C^T accumulators: acc[i][t][r] = C[row wr*64 + 16 i + (lane & 15)][col wc*192 + 16 t + 4 (lane >> 4) + r].
  // They are INITIALISED with bias + residual (the registers are free now and the loads overlap the DMA prologue); adding
  // them in the epilogue, next to 192 live accumulators, spilled ~1 KB per lane.
  const int mrow = lane & 15, cq = lane >> 4;
  const int col0 = wc * 192 + 4 * cq;
  f32x4 acc[4][12];
  acc_init4<GELU, RIN>(acc, bias, residual, res_lo, M, m0, id, wave, wr, lane, col0, res_mod);

  const int frow = lane & 15, fch = lane >> 4;
  int a_off[4], b_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_off[i] = k4WBytes + swz4(wr * 64 + i * 16 + frow, fch);
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) b_off[jj] = swz4(wc * 64 + jj * 16 + frow, fch);

  const int nk = K / k4BK;                 // >= 4 (launcher)
  const bool late = wave >= 4;
  if (SLOTS == 7) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");        // A(0), W piece 0 landed
  else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (late) __builtin_amdgcn_s_barrier();                   // stagger: waves 4-7 one barrier behind

  int wslot = 0;                            // ring slot of the piece being multiplied
  for (int g = 0; g < nk; ++g) {
    const char* a_s = smem + (g & 3) * k4APiece;
    bf16x8 af[4];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const char* w_s = smem + wslot * k4WPiece;
      bf16x8 bfr[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) bfr[jj] = *reinterpret_cast<const bf16x8*>(w_s + b_off[jj]);
      if (j == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(a_s + a_off[i]);
      }
      // ring refill: piece s+SLOTS-1 goes to the slot of piece s-1, whose last readers (waves 4-7) completed their reads (lgkmcnt(0)
      // below) before the barrier this wave passed at the end of the previous sub-step; A(g+3) to the slot of A(g-1).  INM: issued
      // between the MFMAs below instead (an LDS-DMA issued beside outstanding ds_reads costs the wave ~130 cycles; which phase has the
      // slack depends on the clock the data lets the matrix pipe run at -- see se_gemm_res_ln_bf16)
      const int gN = g + (j + SLOTS - 1) / 3, jN = (j + SLOTS - 1) % 3;
      int slotN = wslot + SLOTS - 1;
      if (slotN >= k4WSlots) slotN -= k4WSlots;
      if (!INM) {
        if (gN < nk) SE4_ISSUE_W(gN, jN, slotN);
        if (j == 0 && g + 3 < nk) SE4_ISSUE_A(g + 3);
      }
      // piece s+1 must have landed: wait until only the DMAs issued after its second half remain.  The counts come from replaying
      // the per-wave issue order (prologue, then per sub-step [read-phase issues | wait | MFMA-phase issues]); rows: first K-step,
      // steady state, K-steps nk-3, nk-2, nk-1 (issue stops: A after nk-4, pieces after sub-step 3 nk - SLOTS)
      {
        // rows per configuration {first K-step, steady, nk-3, nk-2, nk-1} x j; selected with scalar compares only (a run-time indexed
        // table would become a memory load inside the loop)
        //   6 slots, read-phase issue  {9,9,9} {10,10,9} {9,9,8}    {8,6,4} {2,0,0}
        //   6 slots, MFMA-phase issue  {6,7,7} {7,8,7}   {7,7,6}    {6,6,4} {2,0,0}
        //   7 slots, read-phase issue  {11..}  {12..}    {11..}     {8,6,4} {2,0,0}
        //   7 slots, MFMA-phase issue  {8,9,9} {9,10,10} {9,9,9}    {8,6,4} {2,0,0}
        int n;
        if (g == nk - 1) n = (j == 0) ? 2 : 0;
        else if (g == nk - 2) n = (j == 0) ? ((SLOTS == 6 && INM) ? 6 : 8) : (j == 1 ? 6 : 4);
        else if (SLOTS == 6 && !INM) n = (g == 0) ? 9 : (g == nk - 3) ? (j == 2 ? 8 : 9) : (j == 2 ? 9 : 10);
        else if (SLOTS == 6 && INM) n = (g == 0) ? (j == 0 ? 6 : 7) : (g == nk - 3) ? (j == 2 ? 6 : 7) : (j == 1 ? 8 : 7);
        else if (SLOTS == 7 && !INM) n = (g == 0 || g == nk - 3) ? 11 : 12;
        else n = (g == 0) ? (j == 0 ? 8 : 9) : (g == nk - 3) ? 9 : (j == 0 ? 9 : 10);
        switch (n) {
          case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
          case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); break;
          case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); break;
          case 6: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory"); break;
          case 7: asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory"); break;
          case 8: asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); break;
          case 9: asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory"); break;
          case 10: asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory"); break;
          case 11: asm volatile("s_waitcnt vmcnt(11) lgkmcnt(0)" ::: "memory"); break;
          default: asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory"); break;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
          acc[i][4 * j + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[jj], af[i], acc[i][4 * j + jj], 0, 0, 0);
        if (INM) {
          __builtin_amdgcn_sched_barrier(0);
          if (i == 0 && gN < nk) SE4_ISSUE_W0(gN, jN, slotN);
          if (i == 1 && gN < nk) SE4_ISSUE_W1(gN, jN, slotN);
          if (i == 2 && j == 0 && g + 3 < nk) SE4_ISSUE_A(g + 3);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      wslot = (wslot + 1 == k4WSlots) ? 0 : wslot + 1;
    }
  }
  if (!late) __builtin_amdgcn_s_barrier();                  // re-align the two groups
  __syncthreads();                                          // ring is dead: reuse it for the per-column vectors

  ln_epilogue4<GELU, ROUT>(acc, smem, k4RedOff, ln_w, ln_b, eps, M, m0, id, wave, wr, wc, lane, out_f32, out_bf16, out_lo);
}


// ------------------------------------------------------------------------------------------------------------------
// gemm7: the same row-complete tile (128 x 768, 8 waves as 2 x 4, wave tile 64 x 192) on 64-deep K-tiles with gemm6's staging rules.
// The K-sweep of the kernel above gives a marginal rate of 0.82 PFLOP/s (matrix pipe busy 36 % of the cycles, PMC) where the 256 x 256
// eight-phase loop runs at 1.4-1.5; here
//   pieces   : a K-tile is THREE weight pieces of 256 rows x 128 B (rows {wc * 192 + p * 64 + [0, 64)} of all four wave columns: 32 KiB, one
//              full cache line of k per row) + ONE activation piece (128 rows x 128 B, 16 KiB); 16-B chunk index ^ ((row >> 1) & 7)
//   ring     : 4 weight slots + 2 activation slots = 160 KiB (all of the CU's LDS; the LayerNorm scratch reuses the dead ring)
//   phases   : (piece p, k-slice s): 4 + 4 fragment reads, 16 MFMAs (4 row tiles x the piece's 4 column tiles x 32 k) -- six phases per
//              K-tile; a weight piece is read in two CONSECUTIVE phases and is free after them
//   staging  : during the two phases of piece q the wave issues its four 1-KiB parts of piece q + 3 (two per phase) into the slot piece
//              q - 1 left one phase earlier; the first phase of a K-tile also issues the next K-tile's activation piece.  One counted wait
//              per piece (second phase): piece q + 1 complete, 8-10 DMAs (two pieces) stay in flight across the barriers
//   sync     : {reads ; DMA issue ; [vmcnt] ; lgkmcnt(0) | barrier | 16 MFMAs | barrier}, waves 4-7 one barrier behind waves 0-3
// ------------------------------------------------------------------------------------------------------------------
constexpr int k7WSlot = 256 * 128, k7ASlot = 128 * 128, k7ABase = 4 * k7WSlot, k7Lds = 4 * k7WSlot + 2 * k7ASlot;     // 32 KiB, 16 KiB, 160 KiB

template <int GELU, int RIN, int ROUT, int INM = 0, int LN = 1, int X3OUT = 0>
__global__ __launch_bounds__(k4Threads) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm7_res_ln_kernel(
    const uint16_t* __restrict__ A, int lda, const uint16_t* __restrict__ W, int ldw, const float* __restrict__ bias,
    const float* __restrict__ residual, const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps, int M, int K,
    float* __restrict__ out_f32, uint16_t* __restrict__ out_bf16, int ntiles, int res_mod,
    const uint8_t* __restrict__ res_lo, uint8_t* __restrict__ out_lo) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  SE_CLKPROBE_BEGIN();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  int id;
  {
    const int orig = blockIdx.x, xcd = orig & 7, q = ntiles >> 3, r = ntiles & 7;
    id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  const int m0 = id * k4BM;

  // ---- DMA sources as 32-bit byte offsets (launcher: they fit).  A 1-KiB part = 8 slot rows x 128 B: lane -> slot row 8 c + (lane >> 3),
  //      LDS position lane & 7 holds logical chunk (lane & 7) ^ ((row >> 1) & 7).  Weight slot row rho <-> W row (rho >> 6) * 192 + p * 64 + (rho & 63);
  //      wave w issues parts w, w + 8, w + 16, w + 24 of a weight piece and parts w, w + 8 of an activation piece.
  uint32_t w_of[4], a_of[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rho = 8 * (wave + 8 * i) + (lane >> 3);
    const int lc = ((lane & 7) ^ ((rho >> 1) & 7)) << 3;
    w_of[i] = (uint32_t)(((rho >> 6) * 192 + (rho & 63)) * ldw + lc) * 2u;
    if (i < 2) a_of[i] = (uint32_t)(min(m0 + rho, M - 1) * lda + lc) * 2u;
  }
  const uint32_t lds_wave = (uint32_t)(size_t)(lds4_ptr_t)smem + wave * 1024;
  const size_t w_pstep = (size_t)64 * ldw * 2;            // bytes between the weight rows of consecutive pieces
#define SE7_DMA1(base_bytes, off32, lds_dst)                                                                               \
  do {                                                                                                                     \
    if (SE7_ABL & 1) break;                                                                                                \
    uint32_t keep_;                                                                                                        \
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"           \
                 : "=&s"(keep_) : "v"(off32), "s"(base_bytes), "s"(lds_dst) : "memory");                                   \
  } while (0)
  // half `s` (parts 2 s, 2 s + 1 of this wave's four) of weight piece q = 3 t + p into ring slot q & 3
#define SE7_DMA_W(t, p, s)                                                                                                 \
  do {                                                                                                                     \
    const char* sb_ = reinterpret_cast<const char*>(W) + (size_t)(p) * w_pstep + (size_t)(t) * 128;                        \
    const uint32_t d_ = lds_wave + (uint32_t)(((3 * (t) + (p)) & 3) * k7WSlot);                                            \
    SE7_DMA1(sb_, w_of[2 * (s)], d_ + (uint32_t)((2 * (s)) * 8192));                                                       \
    SE7_DMA1(sb_, w_of[2 * (s) + 1], d_ + (uint32_t)((2 * (s) + 1) * 8192));                                               \
  } while (0)
#define SE7_DMA_A(t)                                                                                                       \
  do {                                                                                                                     \
    const char* sb_ = reinterpret_cast<const char*>(A) + (size_t)(t) * 128;                                                \
    const uint32_t d_ = lds_wave + (uint32_t)(k7ABase + ((t) & 1) * k7ASlot);                                              \
    SE7_DMA1(sb_, a_of[0], d_);                                                                                            \
    SE7_DMA1(sb_, a_of[1], d_ + 8192u);                                                                                    \
  } while (0)

  // ring prologue FIRST (the residual rows read for the accumulator initialisation then travel beside it): A (0), pieces 0, 1, 2
  const int nk = K / 64;                    // >= 2 (launcher)
  SE7_DMA_A(0);
  SE7_DMA_W(0, 0, 0);
  SE7_DMA_W(0, 0, 1);
  SE7_DMA_W(0, 1, 0);
  SE7_DMA_W(0, 1, 1);
  SE7_DMA_W(0, 2, 0);
  SE7_DMA_W(0, 2, 1);

  const int cq = lane >> 4;
  const int col0 = wc * 192 + 4 * cq;
  f32x4 acc[4][12];
  if (SE7_ABL & 16) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int t = 0; t < 12; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  } else
  acc_init4<GELU, RIN, LN>(acc, bias, residual, res_lo, M, m0, id, wave, wr, lane, col0, res_mod);
  // every accumulator is COMPLETE here (opaque uses): with the integer codec the compiler otherwise left residual loads pending into the K loop
  // and protected the hand-issued LDS-DMA operands with s_waitcnt vmcnt(1) / vmcnt(0) INSIDE it -- which drains the DMA ring every K-tile
  // (K = 3072: 214 instead of 170 us per launch)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < 12; ++t) asm volatile("" : "+v"(acc[i][t]));

  // fragment addresses: lane -> row (lane & 15) of a 16-row tile, logical chunk 4 s + (lane >> 4); tiles 16 rows apart share the swizzle term
  const int frow = lane & 15, fch = lane >> 4;
  int a_ad[2], b_ad[2];
  {
    const int ra = wr * 64 + frow, rb = wc * 64 + frow;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      a_ad[s2] = k7ABase + ra * 128 + (((4 * s2 + fch) ^ ((ra >> 1) & 7)) << 4);
      b_ad[s2] = rb * 128 + (((4 * s2 + fch) ^ ((rb >> 1) & 7)) << 4);
    }
  }
  const bool late = wave >= 4;
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");         // A (0) and piece 0 landed (the residual loads above already drained more than that)
  __builtin_amdgcn_s_barrier();
  if (!(SE7_ABL & 2) && late) __builtin_amdgcn_s_barrier();                   // stagger: waves 4-7 one barrier behind

  const int npieces = 3 * nk;
  bf16x8 abl_a[4], abl_b[4];                 // ablation 4 only: loop-invariant stand-ins for the LDS fragments
#pragma unroll
  for (int jj = 0; jj < 4; ++jj)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      abl_b[jj][e] = (__bf16)(0.01f * (float)((lane * 7 + jj * 3 + e) % 13 - 6));
      abl_a[jj][e] = (__bf16)(0.01f * (float)((lane * 5 + jj + e * 3) % 11 - 5));
    }
  (void)abl_a; (void)abl_b;
  for (int t = 0; t < nk; ++t) {
    const int a_sl = (t & 1) * k7ASlot;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const int q = 3 * t + p;
      const int w_sl = (q & 3) * k7WSlot;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 bfr[4], af[4];
        if (SE7_ABL & 4) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) { bfr[jj] = abl_b[jj]; af[jj] = abl_a[jj]; asm volatile("" : "+v"(bfr[jj]), "+v"(af[jj])); }
        } else {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) bfr[jj] = *reinterpret_cast<const bf16x8*>(smem + w_sl + b_ad[s2] + jj * 2048);
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(smem + a_sl + a_ad[s2] + i * 2048);
        }
        // staging (see the header): piece q + 3 -> the slot piece q - 1 left at the end of the previous phase; A (t + 1) -> the slot A (t - 1) left
        if (!INM) {
          if (p == 0 && s2 == 0 && t + 1 < nk) SE7_DMA_A(t + 1);
          if (q + 3 < npieces) SE7_DMA_W(t + 1, p, s2);         // piece q + 3 = piece p of K-tile t + 1
        }
        if (s2 == 1) {
          // piece q + 1 (and, before a new K-tile, its activation piece) complete: everything issued during pieces q - 1 and q may stay in flight
          // (INM: this phase's own half of piece q + 3 is issued AFTER this wait, between the MFMA groups: two parts fewer are in flight here)
          const int n = (q + 2 < npieces ? 4 : 0) + (q + 3 < npieces ? (INM ? 2 : 4) : 0) + ((p == 0 || p == 1) && t + 1 < nk ? 2 : 0);
          switch (n) {
            case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (!(SE7_ABL & 2)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (!INM) __builtin_amdgcn_s_setprio(1);        // with the refill between the MFMA groups the raised priority costs ~1 % of the step (A/B)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
            acc[i][4 * p + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[jj], af[i], acc[i][4 * p + jj], 0, 0, 0);
          if (INM) {          // the ring refill between the MFMA groups instead of in the read phase (A/B, whole step: 4.199 -> 4.147 ms;
                              // one part per group instead of two after the first: 4.17-4.19)
            __builtin_amdgcn_sched_barrier(0);
            if (i == 0 && q + 3 < npieces) SE7_DMA_W(t + 1, p, s2);
            if (i == 1 && p == 0 && s2 == 0 && t + 1 < nk) SE7_DMA_A(t + 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (!INM) __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (!(SE7_ABL & 2)) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (!(SE7_ABL & 2) && !late) __builtin_amdgcn_s_barrier();                  // re-align the two groups
  __syncthreads();                                          // ring is dead: reuse it for the per-column vectors
  if ((SE7_ABL & 8) && acc[0][0][0] != 12345.678f) {
    // ablation: no epilogue (the comparison keeps the accumulators alive)
  } else
  ln_epilogue4<GELU, ROUT, LN, X3OUT>(acc, smem, 8192, ln_w, ln_b, eps, M, m0, id, wave, wr, wc, lane, out_f32, out_bf16, out_lo);
  SE_CLKPROBE_END(clkprobe_gemm7);
#undef SE7_DMA1
#undef SE7_DMA_W
#undef SE7_DMA_A
}

}  // namespace se

static int gemm4_launch(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const float* residual_f32,
                        const float* ln_w, const float* ln_b, float eps, int M, int N, int K,
                        float* out_f32, uint16_t* out_bf16, bool gelu_no_residual, int res_mod, void* stream,
                        const uint16_t* res_hi = nullptr, const uint8_t* res_lo = nullptr, uint8_t* out_lo = nullptr) {
  const bool rin = res_hi != nullptr, rout = out_lo != nullptr;
  SE_REQUIRE(A && W && (residual_f32 || gelu_no_residual || rin) && ln_w && ln_b && (out_f32 || out_bf16), "se_gemm_res_ln_bf16: null argument");
  SE_REQUIRE(!rin || res_lo, "se_gemm_res_ln_bf16: 24-bit residual without its low bytes");
  SE_REQUIRE(!rout || out_bf16, "se_gemm_res_ln_bf16: 24-bit output needs the bf16 tensor");
  if (gelu_no_residual) residual_f32 = ln_w;      // never dereferenced by the GELU instantiation; keeps the pointer checks below uniform
  if (rin) residual_f32 = reinterpret_cast<const float*>(res_hi);
  if (N != se::k4N || K % se::k4BK != 0 || K < 4 * se::k4BK) {
    se::set_error("se_gemm_res_ln_bf16: the fused kernel is specialised for N = 768 and K a multiple of 32, K >= 128 (got N=%d K=%d)", N, K);
    return SE_ERR_UNSUPPORTED;
  }
  SE_REQUIRE(M > 0 && lda >= K && ldw >= K && lda % 8 == 0 && ldw % 8 == 0, "se_gemm_res_ln_bf16: bad leading dimensions");
  SE_REQUIRE((((uintptr_t)A | (uintptr_t)W | (uintptr_t)residual_f32 | (uintptr_t)out_f32 | (uintptr_t)out_bf16 | (uintptr_t)bias |
               (uintptr_t)ln_w | (uintptr_t)ln_b | (uintptr_t)res_lo | (uintptr_t)out_lo) % 16) == 0, "se_gemm_res_ln_bf16: pointers must be 16-B aligned");
  static int cfg = -1, stagger = 0;
  if (cfg < 0) {
    const char* es = getenv("SE_AMD_GEMM4_STAGGER");
    stagger = es ? atoi(es) : 0;        // x ~3.4 us.  With the fp32 stream 2-4 was worth 0.7 % (4.93 -> 4.89 ms per step); with the 24-bit stream it is 0 +- 0.3 %: off
    const char* e = getenv("SE_AMD_GEMM4_CFG");          // developer switch: bit 0 = DMA issue between the MFMAs, bit 1 = 7-piece ring
    cfg = e ? (atoi(e) & 3) : 1;        // default: 6 pieces, MFMA-phase issue -- equal to read-phase issue when A streams from the Infinity Cache (B = 32 bench), 17 % faster when it comes from HBM
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm4_res_ln_kernel<6, 0, 0, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, se::k4_lds(6)));
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm4_res_ln_kernel<6, 1, 0, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, se::k4_lds(6)));
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm4_res_ln_kernel<7, 0, 0, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, se::k4_lds(7)));
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm4_res_ln_kernel<7, 1, 0, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, se::k4_lds(7)));
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm4_res_ln_kernel<6, 1, 1, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, se::k4_lds(6)));
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm4_res_ln_kernel<6, 1, 0, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, se::k4_lds(6)));
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm4_res_ln_kernel<6, 1, 0, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, se::k4_lds(6)));
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm4_res_ln_kernel<6, 1, 0, 1, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, se::k4_lds(6)));
  }
  const int ntiles = (M + se::k4BM - 1) / se::k4BM;
  hipStream_t st = se::as_stream(stream);
  se::ProfScope prof(se::kProfGemm, 2.0 * M * (double)N * K, st);
  static int use7 = -1;
  if (use7 < 0) {
    const char* e7 = getenv("SE_AMD_GEMM7");           // 0: the 32-deep six-piece ring above for every call
    use7 = e7 ? atoi(e7) : 1;
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm7_res_ln_kernel<1, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, se::k7Lds));
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm7_res_ln_kernel<0, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, se::k7Lds));
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm7_res_ln_kernel<0, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, se::k7Lds));
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm7_res_ln_kernel<0, 1, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, se::k7Lds));
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm7_res_ln_kernel<0, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, se::k7Lds));
  }
  // 64-deep K-tiles (gemm7): whole K-tiles, at least two, 32-bit source offsets
  if (use7 && K % 64 == 0 && K >= 128 && (size_t)M * lda < (1u << 31) && (size_t)se::k4N * ldw < (1u << 31)) {
#define SE7_LAUNCH(GE, RI, RO, IM)                                                                                                        \
  hipLaunchKernelGGL((se::gemm7_res_ln_kernel<GE, RI, RO, IM>), dim3(ntiles), dim3(se::k4Threads), se::k7Lds, st, A, lda, W, ldw, bias, residual_f32, \
                     ln_w, ln_b, eps, M, K, out_f32, out_bf16, ntiles, res_mod, res_lo, out_lo)
    static int inm7 = -1;
    if (inm7 < 0) {
      const char* e = getenv("SE_AMD_GEMM7_INM");        // 0: ring refill issued in the read phase (the first version of this kernel)
      inm7 = e ? atoi(e) : 1;
      SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm7_res_ln_kernel<1, 0, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, se::k7Lds));
      SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm7_res_ln_kernel<0, 0, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, se::k7Lds));
      SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm7_res_ln_kernel<0, 0, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, se::k7Lds));
      SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm7_res_ln_kernel<0, 1, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, se::k7Lds));
      SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm7_res_ln_kernel<0, 1, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, se::k7Lds));
    }
#define SE7_PICK(GE, RI, RO) do { if (inm7) SE7_LAUNCH(GE, RI, RO, 1); else SE7_LAUNCH(GE, RI, RO, 0); } while (0)
    if (gelu_no_residual) SE7_PICK(1, 0, 0);
    else if (rin && rout) SE7_PICK(0, 1, 1);
    else if (rin) SE7_PICK(0, 1, 0);
    else if (rout) SE7_PICK(0, 0, 1);
    else SE7_PICK(0, 0, 0);
#undef SE7_PICK
#undef SE7_LAUNCH
    SE_LAUNCH_CHECK();
    return SE_OK;
  }
  const int stg_ = (K >= 768) ? stagger : 0;       // the short K = 128 input stage has no compute phase to hide anything under
#define SE4_LAUNCH(SL, IM, GE, RI, RO)                                                                                                    \
  hipLaunchKernelGGL((se::gemm4_res_ln_kernel<SL, IM, GE, RI, RO>), dim3(ntiles), dim3(se::k4Threads), se::k4_lds(SL), st, A, lda, W, ldw, bias, \
                     residual_f32, ln_w, ln_b, eps, M, K, out_f32, out_bf16, ntiles, res_mod, stg_, res_lo, out_lo)
  if (gelu_no_residual) SE4_LAUNCH(6, 1, 1, 0, 0);
  else if (rin && rout) SE4_LAUNCH(6, 1, 0, 1, 1);
  else if (rin) SE4_LAUNCH(6, 1, 0, 1, 0);
  else if (rout) SE4_LAUNCH(6, 1, 0, 0, 1);
  else switch (cfg) {
    case 0: SE4_LAUNCH(6, 0, 0, 0, 0); break;
    case 2: SE4_LAUNCH(7, 0, 0, 0, 0); break;
    case 3: SE4_LAUNCH(7, 1, 0, 0, 0); break;
    default: SE4_LAUNCH(6, 1, 0, 0, 0); break;
  }
#undef SE4_LAUNCH
  SE_LAUNCH_CHECK();
  return SE_OK;
}

// The row-complete kernel WITHOUT its LayerNorm (round 4): out = A[M,K] . W[768,K]^T (+ bias) (+ residual_f32) as fp32 and / or bf16 rows.  For N = 768
// the 256 x 256-tile kernels run 378 tiles = a second round filled to 48 %; 251 row tiles of 128 x 768 are one full round.  Called by se_gemm_bf16
// (csrc/gemm.hip) for the training path's FFN-output forward and the FFN1 / QKV input gradients (K = 3072 / 2304).  Returns 1 when the shape is not its own.
extern "C" int se_gemm7_plain_launch(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const float* residual_f32, int M, int K,
                                     uint16_t* out_bf16, float* out_f32, void* stream) {
  if (!(K % 64 == 0 && K >= 128 && (size_t)M * lda < (1u << 31) && (size_t)se::k4N * ldw < (1u << 31))) return 1;
  if ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)residual_f32 | (uintptr_t)out_f32 | (uintptr_t)out_bf16 | (uintptr_t)bias) % 16) != 0) return 1;
  static bool attr_set = false;
  if (!attr_set) {
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm7_res_ln_kernel<0, 0, 0, 1, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, se::k7Lds));
    attr_set = true;
  }
  const int ntiles = (M + se::k4BM - 1) / se::k4BM;
  hipStream_t st = se::as_stream(stream);
  se::ProfScope prof(se::kProfGemm, 2.0 * M * (double)se::k4N * K, st);
  hipLaunchKernelGGL((se::gemm7_res_ln_kernel<0, 0, 0, 1, 0>), dim3(ntiles), dim3(se::k4Threads), se::k7Lds, st, A, lda, W, ldw, bias, residual_f32,
                     (const float*)nullptr, (const float*)nullptr, 0.f, M, K, out_f32, out_bf16, ntiles, 0, (const uint8_t*)nullptr, (uint8_t*)nullptr);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

// bf16x3 parity mode: x = LayerNorm(A . W^T + bias + residual_f32) as fp32 rows (the next residual) AND as the next projection's three-slice operand
// out3 (M, 3 x 768) = [x1 | x1 | x2] from the same launch (A / W are three-slice operands themselves, K = 3 x the layer's width); was
// se_gemm_res_ln_bf16 + se_split3_bf16.  N = 768 only; returns SE_ERR_UNSUPPORTED otherwise.
extern "C" int se_gemm_res_ln_x3_bf16(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const float* residual_f32, const float* ln_w,
                                      const float* ln_b, float eps, int M, int N, int K, float* out_f32, uint16_t* out3, void* stream) {
  SE_REQUIRE(A && W && residual_f32 && ln_w && ln_b && out_f32 && out3, "se_gemm_res_ln_x3_bf16: null argument");
  if (N != se::k4N || K % 64 != 0 || K < 128 || (size_t)M * lda >= (1u << 31) || (size_t)se::k4N * ldw >= (1u << 31)) {
    se::set_error("se_gemm_res_ln_x3_bf16: N = 768, K a multiple of 64 (>= 128), 31-bit operand offsets");
    return SE_ERR_UNSUPPORTED;
  }
  SE_REQUIRE(M > 0 && lda >= K && ldw >= K && lda % 8 == 0 && ldw % 8 == 0, "se_gemm_res_ln_x3_bf16: bad leading dimensions");
  SE_REQUIRE((((uintptr_t)A | (uintptr_t)W | (uintptr_t)residual_f32 | (uintptr_t)out_f32 | (uintptr_t)out3 | (uintptr_t)bias | (uintptr_t)ln_w |
               (uintptr_t)ln_b) % 16) == 0, "se_gemm_res_ln_x3_bf16: pointers must be 16-B aligned");
  static bool attr_set = false;
  if (!attr_set) {
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::gemm7_res_ln_kernel<0, 0, 0, 1, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, se::k7Lds));
    attr_set = true;
  }
  const int ntiles = (M + se::k4BM - 1) / se::k4BM;
  hipStream_t st = se::as_stream(stream);
  se::ProfScope prof(se::kProfGemm, 2.0 * M * (double)se::k4N * K, st);
  hipLaunchKernelGGL((se::gemm7_res_ln_kernel<0, 0, 0, 1, 1, 1>), dim3(ntiles), dim3(se::k4Threads), se::k7Lds, st, A, lda, W, ldw, bias, residual_f32, ln_w, ln_b,
                     eps, M, K, out_f32, out3, ntiles, 0, (const uint8_t*)nullptr, (uint8_t*)nullptr);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_gemm_res_ln_bf16(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const float* residual_f32,
                                   const float* ln_w, const float* ln_b, float eps, int M, int N, int K,
                                   float* out_f32, uint16_t* out_bf16, void* stream) {
  return gemm4_launch(A, lda, W, ldw, bias, residual_f32, ln_w, ln_b, eps, M, N, K, out_f32, out_bf16, false, 0, stream);
}

namespace se {
// x = LayerNorm(gelu(A . W^T + bias)): internal helper of the spec head (se_spechead_fwd_bf16)
int launch_gemm_gelu_ln(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const float* ln_w, const float* ln_b, float eps,
                        int M, int N, int K, float* out_f32, uint16_t* out_bf16, hipStream_t st) {
  return gemm4_launch(A, lda, W, ldw, bias, nullptr, ln_w, ln_b, eps, M, N, K, out_f32, out_bf16, true, 0, st);
}
// x = LayerNorm(A . W^T + bias + table[row % T]): the encoder's input stage (projection + sinusoidal positions + LayerNorm); out_lo != nullptr:
// the stream leaves as (bf16, lo) instead of (fp32, bf16)
int launch_gemm_pos_ln(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const float* table, int T, const float* ln_w,
                       const float* ln_b, float eps, int M, int N, int K, float* out_f32, uint16_t* out_bf16, uint8_t* out_lo, hipStream_t st) {
  return gemm4_launch(A, lda, W, ldw, bias, table, ln_w, ln_b, eps, M, N, K, out_f32, out_bf16, false, T, st, nullptr, nullptr, out_lo);
}
// the encoder's projections on the 24-bit stream: residual (res_hi, res_lo) in; (out_bf16, out_lo) or, for the last layer, fp32 out
int launch_gemm_res24_ln(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const uint16_t* res_hi, const uint8_t* res_lo,
                         const float* ln_w, const float* ln_b, float eps, int M, int N, int K, float* out_f32, uint16_t* out_bf16, uint8_t* out_lo,
                         hipStream_t st) {
  return gemm4_launch(A, lda, W, ldw, bias, nullptr, ln_w, ln_b, eps, M, N, K, out_f32, out_bf16, false, 0, st, res_hi, res_lo, out_lo);
}
size_t gemm4_lo_bytes(int M) { return (size_t)((M + k4BM - 1) / k4BM) * k4BM * k4N; }
}  // namespace se

// test / measurement surface of the row-complete projection on the 24-bit stream: variant 0 = what the encoder would pick, 7 = the 128 x 768
// tile (gemm7 / gemm4), 8 = the 256 x 384 pair-exchange tile (gemm8, forced at any size); scratch: se_gemm_res24_scratch_bytes() bytes, zeroed
extern "C" size_t se_gemm_res24_scratch_bytes(void) { return se::gemm8_scratch_bytes(); }
extern "C" size_t se_gemm_res24_lo_bytes(int M) { return se::gemm4_lo_bytes(M); }
extern "C" int se_gemm_res24_ln_bf16(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const uint16_t* res_hi, const uint8_t* res_lo,
                                     const float* ln_w, const float* ln_b, float eps, int M, int N, int K, float* out_f32, uint16_t* out_bf16, uint8_t* out_lo,
                                     int variant, void* scratch, void* stream) {
  SE_REQUIRE(A && W && res_hi && res_lo && ln_w && ln_b && (out_f32 || out_bf16), "se_gemm_res24_ln_bf16: null argument");
  hipStream_t st = se::as_stream(stream);
#ifndef SE_AMD_EXPERIMENTS
  (void)scratch;
  if (variant == 8) {
    se::set_error("se_gemm_res24_ln_bf16: variant 8 (the 256 x 384 pair-exchange experiment) is not part of the product library");
    return SE_ERR_UNSUPPORTED;
  }
#else
  if (variant == 0 || variant == 8) {
    const int rc = se::launch_gemm8_res24_ln(A, lda, W, ldw, bias, res_hi, res_lo, ln_w, ln_b, eps, M, N, K, out_f32, out_bf16, out_lo, scratch, st, variant == 8);
    if (rc != 1) return rc;
    if (variant == 8) {
      se::set_error("se_gemm_res24_ln_bf16: the 256 x 384 pair-exchange kernel does not take this call (N=%d K=%d M=%d)", N, K, M);
      return SE_ERR_UNSUPPORTED;
    }
  }
#endif
  return se::launch_gemm_res24_ln(A, lda, W, ldw, bias, res_hi, res_lo, ln_w, ln_b, eps, M, N, K, out_f32, out_bf16, out_lo, st);
}
