// mhsa_tile.h -- tile geometry shared by the flash MHSA forward (mhsa.hip) and backward (mhsa_bwd.hip) kernels.
#pragma once
#include "bf16.h"

namespace se {

constexpr int kAQ = 128;        // rows of the stationary operand per workgroup (4 waves x 32)
constexpr int kAK = 64;         // rows of the streamed operand per LDS tile
constexpr int kHD = 64;         // head dim
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

// byte offset of 16-B chunk `ch` (8 bf16) of row `key` inside a [64][64] bf16 tile; f is a bit permutation of
// (key>>1)&7 chosen so that (a) 16 rows at one chunk hit 16 distinct 16-B slots (b128 row reads) and
// (b) 4 consecutive rows land in 4 distinct 64-B quarters of the 256-B bank row (tr_b16 transposed reads)
__device__ __forceinline__ int kv_off(int key, int ch) {
  const int f = (((key >> 1) & 1) << 2) | ((key >> 2) & 3);
  return key * 128 + ((ch ^ f) << 4);
}

}  // namespace se
