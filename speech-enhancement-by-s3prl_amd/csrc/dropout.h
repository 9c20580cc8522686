// dropout.h -- counter-based dropout masks for the training path (BERT dropout sites of the S3PRL encoder: after the input
// LayerNorm, on the attention probabilities, after the attention-output and FFN-output projections; rates from
// config/pretrain_sample.yaml:9-10).  The mask is a pure function of (seed, site, element index), so the backward kernels
// regenerate it instead of storing it, and the CPU oracle reproduces it bit for bit (oracle/encoder.py: keep_mask).
//
//   key(site)     = lowbias32(seed_lo * 0x9E3779B9 + seed_hi * 0x85EBCA6B + site)
//   keep(pair, h) = ((mix24(pair ^ key) >> (16 h)) & 0xffff) >= thr16              thr16 = round(p * 65536)
// mix24 = the lowbias32 structure (xorshift 16, multiply, xorshift 15, multiply, xorshift 16) with 24-bit multiplies
// (v_mul_u32_u24, full rate; v_mul_lo_u32 is quarter rate): 32 instead of 56 issue cycles per element pair in the VALU-bound
// attention kernels.  Not a bijection (dropout does not need one); keep fraction, lag-1..1001 autocorrelation and the
// correlation between the two halves measured at the noise floor (|r| <= 0.002 over 4 M consecutive pairs, 3 keys).
// One 32-bit hash serves TWO neighbouring elements (16 random bits each): `pair` = element index / 2 inside a row-major
// (rows, cols) site with cols even -- for the attention site pair = row_id * ceil(T / 2) + key / 2 -- and h = index & 1.
// Kept elements are scaled by 1 / (1 - p) (torch semantics).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace se {

__host__ __device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}

__host__ __device__ __forceinline__ uint32_t dropout_key(uint64_t seed, uint32_t site) {
  const uint32_t k = lowbias32((uint32_t)seed * 0x9E3779B9U + (uint32_t)(seed >> 32) * 0x85EBCA6BU + site);
  return k ? k : 1u;          // 0 means "no dropout at this site" to the kernels
}

__host__ __device__ __forceinline__ uint32_t mix24(uint32_t x) {
  x ^= x >> 16;
  x = (x & 0xffffffU) * 0x7feb35U;
  x ^= x >> 15;
  x = (x & 0xffffffU) * 0x46ca6bU;
  x ^= x >> 16;
  return x;
}

// the 32 random bits of one element pair
__device__ __forceinline__ uint32_t dropout_bits(uint32_t key, uint32_t pair) { return mix24(pair ^ key); }
// multiplier (0 or `scale`) of half h (0 / 1) of a pair
__device__ __forceinline__ float dropout_mul(uint32_t bits, int h, uint32_t thr16, float scale) {
  const uint32_t v = h ? (bits >> 16) : (bits & 0xffffU);
  return v >= thr16 ? scale : 0.f;
}

struct DropoutCfg {
  uint32_t thr16;     // 0 = dropout off
  float scale;        // 1 / (1 - p)
  uint64_t seed;
};

inline DropoutCfg make_dropout(float p, uint64_t seed) {
  DropoutCfg d;
  d.seed = seed;
  if (!(p > 0.f)) {
    d.thr16 = 0;
    d.scale = 1.f;
  } else {
    const float pc = p > 0.95f ? 0.95f : p;
    d.thr16 = (uint32_t)(pc * 65536.f + 0.5f);
    d.scale = 1.f / (1.f - pc);
  }
  return d;
}

// site ids: 4 per layer (0 attention probabilities, 1 attention-output dense, 2 FFN-output dense), input stage = 4 * layers
__host__ __device__ __forceinline__ uint32_t dropout_site(int layer, int which) { return 4u * (uint32_t)layer + (uint32_t)which; }

}  // namespace se
