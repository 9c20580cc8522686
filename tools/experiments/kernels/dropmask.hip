// dropmask.hip -- the attention-probability dropout mask of ONE layer as two bit matrices, generated ONCE per layer and step and consumed by the
// three attention kernels of the training path (forward, dQ, dK/dV) instead of being re-hashed inside each of them.
// The mask itself is unchanged -- keep(pair, h) of dropout.h, the function the CPU oracle reproduces bit for bit (oracle/encoder.py: keep_mask;
// reference: the attention_probs dropout of the S3PRL encoder, rate from config/pretrain_sample.yaml:10) -- only WHERE it is evaluated moves:
// the hash was 10 + 8 vector instructions per element pair in the forward / dQ kernels and a full hash per ELEMENT in the key-stationary dK/dV
// kernel (its lane's pair partner lives in the neighbouring lane), i.e. most of the vector work of all three; a lane now fetches the 64 mask
// bits of its tile row with one 8-byte load and applies them with two instructions per element (v_bfe_i32 + v_and_b32).
//
//   R [B * heads][T][Wr]    query-major (forward, dQ: lane = query).  A 64-key tile kt of query row q is the word pair R[..][q][2 kt .. 2 kt + 1]:
//                            word 0 = the 32 EVEN keys of the tile, word 1 = the 32 ODD keys (bit j = key 64 kt + 2 j (+ 1)) -- the order in which
//                            the generator's per-pair lanes deliver them (a v_cmp result IS the ballot of the wave).  Wr = 4 ceil(T / 128).
//   C [B * heads][Tc][Wc]   key-major (dK/dV: lane = key).  Bit j of C[..][k][w] = query 32 w + j.  Tc = 128 ceil(T / 128), Wc = 8 ceil(T / 256).
// Bits of keys / queries >= T are unspecified (the kernels mask them by length).
// Generator: lane = one element PAIR (one hash, two keys), wave = 128 keys, workgroup = 4 waves = 512 keys x 256 queries.  Per query row: one
// hash, three compares whose SGPR results are the R words of the row, two v_addc that shift the lane's own two key bits into its C words.
#include "common.h"
#include "dropout.h"

namespace se {

// rows I .. 31 of one group of 32 query rows (compile-time recursion: the lane select of v_writelane must be an inline constant -- its data operand
// already takes the instruction's one SGPR read)
template <int I>
__device__ __forceinline__ void mask_rows(uint32_t& pr, uint32_t dkey, uint32_t thr16, uint32_t thr_hi, uint32_t ppr, int rows_left, uint32_t& we,
                                          uint32_t& wo, uint32_t& r0, uint32_t& r1, uint32_t& r2, uint32_t& r3) {
  const uint32_t h = mix24(pr ^ dkey);
  // keep <=> 16-bit half >= thr16: the low half through the 16-bit compare, the high half as h >= thr16 << 16 (one instruction each); the
  // compare result in SGPRs is the wave's ballot, and v_addc shifts the lane's own bit into its key word: w = 2 w + bit
  const uint64_t me = __ballot((uint16_t)h >= (uint16_t)thr16);
  const uint64_t mo = __ballot(h >= thr_hi);
  uint64_t junk;
  asm("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(we), "=s"(junk) : "s"(me));
  asm("v_addc_co_u32_e64 %0, %1, %0, %0, %2" : "+v"(wo), "=s"(junk) : "s"(mo));
  // lane I keeps the row's four R words (v_writelane: SGPR value into one lane of a VGPR)
  asm("v_writelane_b32 %0, %1, %2" : "+v"(r0) : "s"((uint32_t)me), "n"(I));
  asm("v_writelane_b32 %0, %1, %2" : "+v"(r1) : "s"((uint32_t)mo), "n"(I));
  asm("v_writelane_b32 %0, %1, %2" : "+v"(r2) : "s"((uint32_t)(me >> 32)), "n"(I));
  asm("v_writelane_b32 %0, %1, %2" : "+v"(r3) : "s"((uint32_t)(mo >> 32)), "n"(I));
  if (I + 1 < rows_left) pr += ppr;                     // rows past the end repeat the last one (never read)
  if constexpr (I + 1 < 32) mask_rows<I + 1>(pr, dkey, thr16, thr_hi, ppr, rows_left, we, wo, r0, r1, r2, r3);
}

__global__ __launch_bounds__(256) void mhsa_dropmask_kernel(uint32_t* __restrict__ R, uint32_t* __restrict__ C, int T, int Wr, int Tc, int Wc,
                                                            uint32_t dkey, uint32_t thr16) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bh = blockIdx.z;
  const int kgrp = blockIdx.x * 4 + wave;                 // 128-key group of this wave
  if (kgrp * 128 >= Tc) return;                           // wave-uniform
  const uint32_t ppr = (uint32_t)((T + 1) >> 1);
  const uint32_t pair = min((uint32_t)(kgrp * 64 + lane), ppr - 1);
  const int qb0 = blockIdx.y * 256;
  const uint32_t thr_hi = thr16 << 16;
  uint32_t* c_even = C + ((size_t)bh * Tc + (size_t)kgrp * 128 + 2 * lane) * Wc + (qb0 >> 5);
  uint32_t* c_odd = c_even + Wc;
  for (int gg = 0; gg < 2; ++gg) {
    uint32_t ce[4], co[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int qg = qb0 + 32 * (4 * gg + g);             // first query of this group of 32 rows
      uint32_t we = 0, wo = 0;                            // the lane's even / odd key: query bits, first row in the MSB
      uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;            // lane i: the four R words of row qg + i
      uint32_t pr = ((uint32_t)bh * (uint32_t)T + (uint32_t)min(qg, T - 1)) * ppr + pair;
      mask_rows<0>(pr, dkey, thr16, thr_hi, ppr, T - qg, we, wo, r0, r1, r2, r3);
      ce[g] = __builtin_bitreverse32(we);
      co[g] = __builtin_bitreverse32(wo);
      const int q = qg + lane;
      if (lane < 32 && q < T) *reinterpret_cast<uint4*>(R + ((size_t)bh * T + q) * Wr + 4 * kgrp) = make_uint4(r0, r1, r2, r3);
    }
    *reinterpret_cast<uint4*>(c_even + 4 * gg) = make_uint4(ce[0], ce[1], ce[2], ce[3]);      // Wc = 8 words per 256-query block: always inside the row
    *reinterpret_cast<uint4*>(c_odd + 4 * gg) = make_uint4(co[0], co[1], co[2], co[3]);
  }
}

}  // namespace se

extern "C" size_t se_mhsa_dropmask_bytes(int B, int T, int heads, int which) {
  if (B <= 0 || T <= 0 || heads <= 0) return 0;
  const size_t bh = (size_t)B * heads;
  if (which == 0) return bh * (size_t)T * (4 * ((T + 127) / 128)) * 4;
  return bh * (size_t)(128 * ((T + 127) / 128)) * (8 * ((T + 255) / 256)) * 4;
}

extern "C" int se_mhsa_dropmask(int B, int T, int heads, float dropout_p, uint64_t seed, uint32_t site, uint32_t* mask_r, uint32_t* mask_c, void* stream) {
  SE_REQUIRE(mask_r && mask_c, "se_mhsa_dropmask: null argument");
  SE_REQUIRE(B > 0 && T > 0 && heads > 0 && (size_t)B * heads <= 65535, "se_mhsa_dropmask: bad shape B=%d T=%d heads=%d", B, T, heads);
  SE_REQUIRE((((uintptr_t)mask_r | (uintptr_t)mask_c) % 16) == 0, "se_mhsa_dropmask: buffers must be 16-B aligned");
  const se::DropoutCfg d = se::make_dropout(dropout_p, seed);
  SE_REQUIRE(d.thr16 != 0, "se_mhsa_dropmask: dropout_p must be > 0");
  SE_REQUIRE((double)B * heads * T * ((T + 1) / 2) < 4294967296.0, "se_mhsa_dropmask: dropout pair index exceeds 32 bits");
  const int Tc = 128 * ((T + 127) / 128), Wr = 4 * (Tc / 128), Wc = 8 * ((T + 255) / 256);
  dim3 grid((Tc / 128 + 3) / 4, (T + 255) / 256, B * heads);
  hipLaunchKernelGGL(se::mhsa_dropmask_kernel, grid, dim3(256), 0, se::as_stream(stream), mask_r, mask_c, T, Wr, Tc, Wc, se::dropout_key(seed, site), d.thr16);
  SE_LAUNCH_CHECK();
  return SE_OK;
}
