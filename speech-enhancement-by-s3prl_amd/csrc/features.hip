// features.hip -- row A4: select_feat of S3PRL OnlinePreprocessor.forward: log(x+eps), stacked deltas
// (5-tap [-2,-1,0,1,2]/10, replicate padding, delta-of-delta for delta=2), CMVN over time
// (mean, UNBIASED std, +eps outside the sqrt), transposed to time-major (B, F, D*(1+delta)).
//
// Two HBM-bound passes:
//   rows : one workgroup per (utterance, raw feature dim): the whole time row lives in LDS; log + deltas +
//          exact two-pass mean / variance; derived rows written feature-major (coalesced along time)
//   emit : one workgroup per (utterance, 32-frame tile): normalise + LDS transpose -> contiguous time-major rows
#include <stdlib.h>
#include "common.h"

namespace se {

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// derived == nullptr: statistics only (se_features3_f32: the one-pass tile kernel below produces the rows).  stat_inv: the pair is written as
// (mean, 1 / (unbiased std + stat_eps)) -- the form head.hip's CMVN multiplies with (colstats_kernel) -- instead of (mean, unbiased std + eps)
__global__ __launch_bounds__(256) void feat_rows_kernel(const float* __restrict__ raw, int raw_time_major, int D, int F,
                                                        int apply_log, int delta, int cmvn, float eps,
                                                        float* __restrict__ derived, float* __restrict__ stats, int32_t* __restrict__ valid_count,
                                                        int stat_inv = 0, float stat_eps = 0.f) {
  extern __shared__ __attribute__((aligned(16))) float rows[];   // (1+delta) x F
  __shared__ float red[4];
  const int b = blockIdx.y, d = blockIdx.x, tid = threadIdx.x;
  if (valid_count && d == 0 && tid == 0) valid_count[b] = 0;      // the emit launch (next on the stream) counts into it: no clearing launch
  const int Dout = D * (1 + delta);
  for (int t = tid; t < F; t += 256) {
    float v = raw_time_major ? raw[((size_t)b * F + t) * D + d] : raw[((size_t)b * D + d) * F + t];
    if (apply_log) v = logf(v + eps);
    rows[t] = v;
  }
  __syncthreads();
  for (int j = 1; j <= delta; ++j) {
    const float* src = rows + (size_t)(j - 1) * F;
    float* dst = rows + (size_t)j * F;
    for (int t = tid; t < F; t += 256) {
      const float m2 = src[max(t - 2, 0)], m1 = src[max(t - 1, 0)], p1 = src[min(t + 1, F - 1)], p2 = src[min(t + 2, F - 1)];
      // conv1d order of torchaudio compute_deltas: sum_k kernel[k] * x[t-2+k], kernel = [-2,-1,0,1,2], then / 10
      dst[t] = (-2.f * m2 - m1 + p1 + 2.f * p2) / 10.f;
    }
    __syncthreads();
  }
  for (int j = 0; j <= delta; ++j) {
    const float* src = rows + (size_t)j * F;
    const int dd = j * D + d;
    if (derived) {
      float* out = derived + ((size_t)b * Dout + dd) * F;
      for (int t = tid; t < F; t += 256) out[t] = src[t];
    }
    if (cmvn) {
      float s = 0.f;
      for (int t = tid; t < F; t += 256) s += src[t];
      const float mean = block_sum_256(s, red) / (float)F;
      float q = 0.f;
      for (int t = tid; t < F; t += 256) {
        const float c = src[t] - mean;
        q = fmaf(c, c, q);
      }
      const float var = block_sum_256(q, red) / (float)(F - 1);
      if (tid == 0) {
        stats[((size_t)b * Dout + dd) * 2 + 0] = mean;
        stats[((size_t)b * Dout + dd) * 2 + 1] = stat_inv ? 1.0f / (sqrtf(var) + stat_eps) : sqrtf(var) + eps;
      }
    }
  }
}

constexpr int kEmitT = 32;

// Side outputs for the encoder that consumes these features next (transformer.py): the rows as bf16, zero-padded to `ld_pad` columns -- exactly the
// operand of its input projection (the pass se_encoder_fwd_bf16 otherwise runs itself) -- and the number of frames whose feature sum is not zero
// (S3PRL process_input_data's length rule; se_valid_lengths_i32 otherwise), counted into valid_count[b] (cleared by feat_rows_kernel).
__global__ __launch_bounds__(256) void feat_emit_kernel(const float* __restrict__ derived, const float* __restrict__ stats,
                                                        int Dout, int F, int cmvn, float* __restrict__ out,
                                                        uint16_t* __restrict__ out_bf16_pad, int ld_pad, int32_t* __restrict__ valid_count) {
  extern __shared__ __attribute__((aligned(16))) float tile[];   // kEmitT x (Dout+1)
  const int b = blockIdx.y, t0 = blockIdx.x * kEmitT, tid = threadIdx.x;
  const int nt = min(kEmitT, F - t0);
  const int ld = Dout + 1;
  for (int it = tid; it < Dout * kEmitT; it += 256) {
    const int dd = it / kEmitT, tl = it - dd * kEmitT;
    if (tl < nt) {
      float v = derived[((size_t)b * Dout + dd) * F + t0 + tl];
      if (cmvn) {
        const float mean = stats[((size_t)b * Dout + dd) * 2], den = stats[((size_t)b * Dout + dd) * 2 + 1];
        v = (v - mean) / den;
      }
      tile[tl * ld + dd] = v;
    }
  }
  __syncthreads();
  float* o = out + ((size_t)b * F + t0) * Dout;
  for (int it = tid; it < nt * Dout; it += 256) {
    const int tl = it / Dout, dd = it - tl * Dout;
    o[it] = tile[tl * ld + dd];
  }
  if (out_bf16_pad) {
    uint16_t* ob = out_bf16_pad + ((size_t)b * F + t0) * ld_pad;
    const int pairs = ld_pad >> 1;
    for (int it = tid; it < nt * pairs; it += 256) {
      const int tl = it / pairs, c = (it - tl * pairs) * 2;
      const float v0 = c < Dout ? tile[tl * ld + c] : 0.f, v1 = c + 1 < Dout ? tile[tl * ld + c + 1] : 0.f;
      typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
      const bf16x2_ pk = {(__bf16)v0, (__bf16)v1};
      *reinterpret_cast<uint32_t*>(ob + (size_t)tl * ld_pad + c) = __builtin_bit_cast(uint32_t, pk);
    }
  }
  if (valid_count) {
    // 8 threads per frame (kEmitT = 32 frames), shuffle-reduced; one integer atomic per workgroup
    const int tl = tid >> 3, part = tid & 7;
    float sum = 0.f;
    if (tl < nt)
      for (int dd = part; dd < Dout; dd += 8) sum += tile[tl * ld + dd];
    sum += __shfl_xor(sum, 1);
    sum += __shfl_xor(sum, 2);
    sum += __shfl_xor(sum, 4);
    const unsigned long long m = __ballot(part == 0 && tl < nt && sum != 0.f);
    __shared__ int cnts[4];
    if ((tid & 63) == 0) cnts[tid >> 6] = __popcll(m);
    __syncthreads();
    if (tid == 0) atomicAdd(&valid_count[b], cnts[0] + cnts[1] + cnts[2] + cnts[3]);
  }
}

// Statistics-only form of feat_rows_kernel for se_features3_f32: ONE WAVE per (utterance, raw feature dim) row, four rows per workgroup.  The
// row-per-workgroup kernel above spends its time in __syncthreads (two per block sum, six block sums per raw row at delta = 2: 52 us for
// 10 240 rows of 1 001 frames); a wave needs none -- its LDS region is private, the reductions are shuffles.  Same arithmetic (log, 5-tap deltas,
// exact two-pass mean / unbiased variance), another summation order.
__global__ __launch_bounds__(256) void feat_stats_kernel(const float* __restrict__ raw, int raw_time_major, int rows_total, int D, int F,
                                                         int apply_log, int delta, float eps, float* __restrict__ stats,
                                                         int32_t* __restrict__ valid_count, int stat_inv, float stat_eps) {
  extern __shared__ __attribute__((aligned(16))) float rows[];   // 4 waves x (1 + delta) x F
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave;
  if (r >= rows_total) return;                                    // whole waves leave; no workgroup barrier below
  const int b = r / D, d = r - b * D;
  if (valid_count && d == 0 && lane == 0) valid_count[b] = 0;     // the tile launch (next on the stream) counts into it
  float* x = rows + (size_t)wave * (1 + delta) * F;
  const int Dout = D * (1 + delta);
  for (int t = lane; t < F; t += 64) {
    float v = raw_time_major ? raw[((size_t)b * F + t) * D + d] : raw[((size_t)b * D + d) * F + t];
    if (apply_log) v = logf(v + eps);
    x[t] = v;
  }
#define SE_WAVE_SYNC()                                         \
  do {                                                         \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     \
    __builtin_amdgcn_wave_barrier();                           \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");     \
  } while (0)
  SE_WAVE_SYNC();
  for (int j = 1; j <= delta; ++j) {
    const float* src = x + (size_t)(j - 1) * F;
    float* dst = x + (size_t)j * F;
    for (int t = lane; t < F; t += 64) {
      const float m2 = src[max(t - 2, 0)], m1 = src[max(t - 1, 0)], p1 = src[min(t + 1, F - 1)], p2 = src[min(t + 2, F - 1)];
      dst[t] = (-2.f * m2 - m1 + p1 + 2.f * p2) / 10.f;
    }
    SE_WAVE_SYNC();
  }
#undef SE_WAVE_SYNC
  for (int j = 0; j <= delta; ++j) {
    const float* src = x + (size_t)j * F;
    float s = 0.f;
    for (int t = lane; t < F; t += 64) s += src[t];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s / (float)F;
    float q = 0.f;
    for (int t = lane; t < F; t += 64) {
      const float c = src[t] - mean;
      q = fmaf(c, c, q);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
    const float var = q / (float)(F - 1);
    if (lane == 0) {
      const size_t o = ((size_t)b * Dout + j * D + d) * 2;
      stats[o] = mean;
      stats[o + 1] = stat_inv ? 1.0f / (sqrtf(var) + stat_eps) : sqrtf(var) + eps;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------------------
// Round 5: ONE pass from the raw plane to the time-major rows (se_features3_f32).  The two launches above move every derived row through HBM
// twice (rows: raw -> feature-major `derived`; emit: `derived` -> time-major `out`): at 256 utterances of mel / log / delta-2 that is 2 x 123 MB
// around a 41 MB input.  log and the deltas only need +-2 frames per delta order, so a workgroup that owns 32 consecutive frames recomputes
// them from a (32 + 4 delta)-frame window of the raw plane in LDS and writes its rows straight out; what needs the WHOLE time axis -- the CMVN
// statistics -- comes from feat_rows_kernel in its statistics-only form (one read of the raw plane, no `derived`).
//   window : x[level j][d][p], p = frame - (t0 - 2 delta); level 0 = log(raw + eps) at the replicate-clamped frame, level j = the 5-tap delta of
//            level j - 1 EVALUATED AT the clamped frame (compute_deltas pads the sequence it differentiates, level by level: the value left of
//            frame 0 is delta_j[0], not a delta of padded inputs)
//   emit   : as feat_emit_kernel (rows of the tile are one contiguous span of `out`; odd LDS pitch: conflict-free column walks), same side outputs
#ifndef SE_FEAT_TILE
#define SE_FEAT_TILE 32
#endif
constexpr int kTileT = SE_FEAT_TILE;      // frames per workgroup (a multiple of 32: the valid-frame count walks 32 frames per 256 threads)
// developer A/B (-DSE_FEAT_FASTMATH, results differ in the last ulp): hardware log2 x ln 2 and a multiplication by 0.1f in place of logf and the IEEE / 10
#ifdef SE_FEAT_FASTMATH
#define SE_FEAT_LOG(x_) __logf(x_)
#define SE_FEAT_DIV10(x_) ((x_) * 0.1f)
#else
#define SE_FEAT_LOG(x_) logf(x_)
#define SE_FEAT_DIV10(x_) ((x_) / 10.f)
#endif

__global__ __launch_bounds__(256) void feat_tile_kernel(const float* __restrict__ raw, int raw_time_major, int D, int F, int apply_log, int delta,
                                                        int cmvn, float eps, const float* __restrict__ stats, float* __restrict__ out,
                                                        uint16_t* __restrict__ out_bf16_pad, int ld_pad, int32_t* __restrict__ valid_count,
                                                        double* __restrict__ col_part) {
  extern __shared__ __attribute__((aligned(16))) float x[];      // (1 + delta) * D rows of pitch P
  const int b = blockIdx.y, t0 = blockIdx.x * kTileT, tid = threadIdx.x;
  const int H = 2 * delta, W = kTileT + 2 * H, P = W + 1;
  const int Dout = D * (1 + delta);
  const int nt = min(kTileT, F - t0);
  // ---- level 0
  if (raw_time_major) {
    for (int it = tid; it < D * W; it += 256) {
      const int p = it / D, d = it - p * D;
      const int u = min(max(t0 - H + p, 0), F - 1);
      float v = raw[((size_t)b * F + u) * D + d];
      if (apply_log) v = SE_FEAT_LOG(v + eps);
      x[d * P + p] = v;
    }
  } else {
    for (int it = tid; it < D * W; it += 256) {
      const int d = it / W, p = it - d * W;
      const int u = min(max(t0 - H + p, 0), F - 1);
      float v = raw[((size_t)b * D + d) * F + u];
      if (apply_log) v = SE_FEAT_LOG(v + eps);
      x[d * P + p] = v;
    }
  }
  __syncthreads();
  // ---- levels 1 .. delta: positions [2 j, W - 2 j)
  for (int j = 1; j <= delta; ++j) {
    const float* src = x + (size_t)(j - 1) * D * P;
    float* dst = x + (size_t)j * D * P;
    const int wj = W - 4 * j;
    for (int it = tid; it < D * wj; it += 256) {
      const int d = it / wj, p = 2 * j + (it - d * wj);
      const int pc = min(max(t0 - H + p, 0), F - 1) - (t0 - H);      // the position of the clamped frame
      const float* s = src + d * P + pc;
      const float m2 = s[-2], m1 = s[-1], p1 = s[1], p2 = s[2];
      dst[d * P + p] = SE_FEAT_DIV10(-2.f * m2 - m1 + p1 + 2.f * p2);
    }
    __syncthreads();
  }
  // ---- normalise in place (the tile's own frames only): every later read is a plain LDS read
  if (cmvn) {
    for (int it = tid; it < Dout * kTileT; it += 256) {
      const int dd = it / kTileT, tl = it - dd * kTileT;
      const float mean = stats[((size_t)b * Dout + dd) * 2], den = stats[((size_t)b * Dout + dd) * 2 + 1];
      float* e = x + dd * P + H + tl;
      *e = (*e - mean) / den;
    }
    __syncthreads();
  }
  const float* xt = x + H;                                      // xt[dd * P + tl]
  float* o = out + ((size_t)b * F + t0) * Dout;
  const int total = nt * Dout;
  if (((reinterpret_cast<uintptr_t>(o) & 15) == 0)) {
    const int nvec = total >> 2;
    int i = 4 * tid, tl = i / Dout, dd = i - tl * Dout;           // one division per thread; then advanced by 1 024 elements per trip
    const int step_tl = 1024 / Dout, step_dd = 1024 - step_tl * Dout;
    for (int v4 = tid; v4 < nvec; v4 += 256) {
      float e[4];
      int tl_ = tl, dd_ = dd;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        e[r] = xt[dd_ * P + tl_];
        if (++dd_ == Dout) { dd_ = 0; ++tl_; }
      }
      *reinterpret_cast<float4*>(o + 4 * v4) = make_float4(e[0], e[1], e[2], e[3]);
      tl += step_tl;
      dd += step_dd;
      if (dd >= Dout) { dd -= Dout; ++tl; }
    }
    for (int it = 4 * nvec + tid; it < total; it += 256) {
      const int tl2 = it / Dout, dd2 = it - tl2 * Dout;
      o[it] = xt[dd2 * P + tl2];
    }
  } else {
    for (int it = tid; it < total; it += 256) {
      const int tl2 = it / Dout, dd2 = it - tl2 * Dout;
      o[it] = xt[dd2 * P + tl2];
    }
  }
  // column statistics of the rows just written, for the mask head's own CMVN (model.py:29-31): this tile's sum and square sum per output column in
  // fp64 (the products of fp32 values are exact there: the one-pass variance of feat_colstats_fold_kernel equals the two-pass fp32 one to below fp32
  // resolution, as head.hip's colstats_kernel) -- a second pass over log / deltas for them cost as much as this whole launch
  if (col_part) {
    for (int dd = tid; dd < Dout; dd += 256) {
      double sm = 0.0, sq = 0.0;
      for (int tl = 0; tl < nt; ++tl) {
        const double v = (double)xt[dd * P + tl];
        sm += v;
        sq = fma(v, v, sq);
      }
      double* cp = col_part + (((size_t)b * gridDim.x + blockIdx.x) * Dout + dd) * 2;
      cp[0] = sm;
      cp[1] = sq;
    }
  }
  if (out_bf16_pad) {
    uint16_t* ob = out_bf16_pad + ((size_t)b * F + t0) * ld_pad;
    const int pairs = ld_pad >> 1;
    for (int it = tid; it < nt * pairs; it += 256) {
      const int tl = it / pairs, c = (it - tl * pairs) * 2;
      const float v0 = c < Dout ? xt[c * P + tl] : 0.f, v1 = c + 1 < Dout ? xt[(c + 1) * P + tl] : 0.f;
      typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
      const bf16x2_ pk = {(__bf16)v0, (__bf16)v1};
      *reinterpret_cast<uint32_t*>(ob + (size_t)tl * ld_pad + c) = __builtin_bit_cast(uint32_t, pk);
    }
  }
  if (valid_count) {
    // 8 threads per frame, shuffle-reduced; one integer atomic per workgroup (as feat_emit_kernel, same summation order)
    const int part = tid & 7;
    int cnt = 0;
    for (int tl = tid >> 3; tl < kTileT; tl += 32) {
      float sum = 0.f;
      if (tl < nt)
        for (int dd = part; dd < Dout; dd += 8) sum += xt[dd * P + tl];
      sum += __shfl_xor(sum, 1);
      sum += __shfl_xor(sum, 2);
      sum += __shfl_xor(sum, 4);
      cnt += __popcll(__ballot(part == 0 && tl < nt && sum != 0.f));
    }
    __shared__ int cnts[4];
    if ((tid & 63) == 0) cnts[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0) atomicAdd(&valid_count[b], cnts[0] + cnts[1] + cnts[2] + cnts[3]);
  }
}

// stats[(b, dd)] = (mean, 1 / (unbiased std + eps)) from the tiles' partial sums (fixed order)
__global__ __launch_bounds__(256) void feat_colstats_fold_kernel(const double* __restrict__ col_part, int ntile, int Dout, int F, float eps,
                                                                 float* __restrict__ stats) {
  const int b = blockIdx.y, dd = blockIdx.x * 256 + threadIdx.x;
  if (dd >= Dout) return;
  double S = 0.0, Q = 0.0;
  const double* cp = col_part + ((size_t)b * ntile * Dout + dd) * 2;
  for (int t = 0; t < ntile; ++t) {
    S += cp[(size_t)t * Dout * 2];
    Q += cp[(size_t)t * Dout * 2 + 1];
  }
  const double mean = S / (double)F;
  const double var = fmax(Q - S * mean, 0.0) / (double)(F - 1);
  stats[((size_t)b * Dout + dd) * 2] = (float)mean;
  stats[((size_t)b * Dout + dd) * 2 + 1] = 1.0f / ((float)sqrt(var) + eps);
}

}  // namespace se

extern "C" size_t se_features_workspace_bytes(int B, int D, int F, int delta) {
  const size_t Dout = (size_t)D * (1 + delta);
  return ((size_t)B * Dout * F + (size_t)B * Dout * 2) * sizeof(float) + 256;
}

extern "C" int se_features_f32(const float* raw, int raw_time_major, int B, int D, int F,
                               int apply_log, int delta, int cmvn, float eps,
                               float* out, void* workspace, size_t workspace_bytes, void* stream) {
  return se_features2_f32(raw, raw_time_major, B, D, F, apply_log, delta, cmvn, eps, out, workspace, workspace_bytes, nullptr, 0, nullptr, stream);
}

extern "C" int se_features2_f32(const float* raw, int raw_time_major, int B, int D, int F,
                                int apply_log, int delta, int cmvn, float eps,
                                float* out, void* workspace, size_t workspace_bytes,
                                uint16_t* out_bf16_pad, int ld_pad, int32_t* valid_count, void* stream) {
  SE_REQUIRE(raw && out && workspace, "se_features_f32: null argument");
  SE_REQUIRE(out_bf16_pad == nullptr || (ld_pad >= D * (1 + delta) && ld_pad % 2 == 0 && (uintptr_t)out_bf16_pad % 4 == 0), "se_features2_f32: bad bf16 side output (ld_pad=%d)", ld_pad);
  SE_REQUIRE(B > 0 && B <= 65535 && D > 0 && D <= 65535 && F >= 2 && delta >= 0 && delta <= 3, "se_features_f32: bad shape B=%d D=%d F=%d delta=%d", B, D, F, delta);
  SE_REQUIRE(workspace_bytes >= se_features_workspace_bytes(B, D, F, delta), "se_features_f32: workspace too small");
  const size_t rows_lds = (size_t)(1 + delta) * F * sizeof(float);
  const int Dout = D * (1 + delta);
  const size_t tile_lds = (size_t)se::kEmitT * (Dout + 1) * sizeof(float);
  SE_REQUIRE(rows_lds <= 64 * 1024, "se_features_f32: F=%d too long for the LDS row buffer", F);
  SE_REQUIRE(tile_lds <= 120 * 1024, "se_features_f32: D*(1+delta)=%d too wide for the LDS transpose tile", Dout);
  float* derived = reinterpret_cast<float*>(workspace);
  float* stats = derived + (size_t)B * Dout * F;
  hipStream_t st = se::as_stream(stream);
  hipLaunchKernelGGL(se::feat_rows_kernel, dim3(D, B), dim3(256), rows_lds, st, raw, raw_time_major, D, F, apply_log, delta,
                     cmvn, eps, derived, stats, valid_count);
  SE_LAUNCH_CHECK();
  if (tile_lds > 64 * 1024)
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::feat_emit_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds));
  hipLaunchKernelGGL(se::feat_emit_kernel, dim3((F + se::kEmitT - 1) / se::kEmitT, B), dim3(256), tile_lds, st, derived, stats,
                     Dout, F, cmvn, out, out_bf16_pad, ld_pad, valid_count);
  SE_LAUNCH_CHECK();
  return SE_OK;
}


// cmvn: the (mean, std + eps) pairs; colstats_out: the tiles' (sum, square sum) pairs in fp64 (call with the F the launch will get)
extern "C" size_t se_features3_workspace_bytes(int B, int D, int delta) { return (size_t)B * D * (1 + delta) * 2 * sizeof(float) + 256; }
extern "C" size_t se_features3_colstats_workspace_bytes(int B, int D, int F, int delta) {
  return (size_t)B * ((F + se::kTileT - 1) / se::kTileT) * D * (1 + delta) * 2 * sizeof(double) + 256;
}

extern "C" int se_features3_f32(const float* raw, int raw_time_major, int B, int D, int F, int apply_log, int delta, int cmvn, float eps,
                                float* out, void* workspace, size_t workspace_bytes, uint16_t* out_bf16_pad, int ld_pad, int32_t* valid_count,
                                float* colstats_out, float colstats_eps, void* stream) {
  SE_REQUIRE(raw && out, "se_features3_f32: null argument");
  SE_REQUIRE(out_bf16_pad == nullptr || (ld_pad >= D * (1 + delta) && ld_pad % 2 == 0 && (uintptr_t)out_bf16_pad % 4 == 0), "se_features3_f32: bad bf16 side output (ld_pad=%d)", ld_pad);
  SE_REQUIRE(B > 0 && B <= 65535 && D > 0 && D <= 65535 && F >= 2 && delta >= 0 && delta <= 3, "se_features3_f32: bad shape B=%d D=%d F=%d delta=%d", B, D, F, delta);
  SE_REQUIRE(!(cmvn && colstats_out), "se_features3_f32: colstats_out describes un-normalised features (cmvn must be 0)");
  SE_REQUIRE(!cmvn || (workspace && workspace_bytes >= se_features3_workspace_bytes(B, D, delta)), "se_features3_f32: workspace too small");
  SE_REQUIRE(!colstats_out || (workspace && workspace_bytes >= se_features3_colstats_workspace_bytes(B, D, F, delta) && (uintptr_t)workspace % 8 == 0),
             "se_features3_f32: colstats_out needs se_features3_colstats_workspace_bytes(B, D, F, delta) bytes of workspace");
  const size_t rows_lds = (size_t)(1 + delta) * F * sizeof(float);
  const int Dout = D * (1 + delta);
  const size_t tile_lds = (size_t)Dout * (se::kTileT + 4 * delta + 1) * sizeof(float);
  SE_REQUIRE(tile_lds <= 120 * 1024, "se_features3_f32: D*(1+delta)=%d too wide for the LDS window", Dout);
  hipStream_t st = se::as_stream(stream);
  float* stats = cmvn ? reinterpret_cast<float*>(workspace) : nullptr;
  if (stats) {
    SE_REQUIRE(rows_lds <= 64 * 1024, "se_features3_f32: F=%d too long for the LDS row buffer", F);
    static const bool wave_rows = getenv("SE_AMD_FEAT_STATS_WG") == nullptr;      // A/B: the workgroup-per-row statistics launch
    if (wave_rows && 4 * rows_lds <= 64 * 1024) {
      hipLaunchKernelGGL(se::feat_stats_kernel, dim3((B * D + 3) / 4), dim3(256), 4 * rows_lds, st, raw, raw_time_major, B * D, D, F, apply_log, delta, eps,
                         stats, valid_count, cmvn ? 0 : 1, colstats_eps);
    } else {
      hipLaunchKernelGGL(se::feat_rows_kernel, dim3(D, B), dim3(256), rows_lds, st, raw, raw_time_major, D, F, apply_log, delta, 1, eps,
                         static_cast<float*>(nullptr), stats, valid_count, cmvn ? 0 : 1, colstats_eps);
    }
    SE_LAUNCH_CHECK();
  } else if (valid_count) {
    const int zrc_ = se::zero_async(valid_count, sizeof(int32_t) * B, st);
    if (zrc_) return zrc_;
  }
  if (tile_lds > 64 * 1024)
    SE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(se::feat_tile_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tile_lds));
  const int ntile = (F + se::kTileT - 1) / se::kTileT;
  double* col_part = colstats_out ? reinterpret_cast<double*>(workspace) : nullptr;
  hipLaunchKernelGGL(se::feat_tile_kernel, dim3(ntile, B), dim3(256), tile_lds, st, raw, raw_time_major, D, F, apply_log,
                     delta, cmvn, eps, cmvn ? stats : nullptr, out, out_bf16_pad, ld_pad, valid_count, col_part);
  SE_LAUNCH_CHECK();
  if (colstats_out) {
    hipLaunchKernelGGL(se::feat_colstats_fold_kernel, dim3((Dout + 255) / 256, B), dim3(256), 0, st, col_part, ntile, Dout, F, colstats_eps, colstats_out);
    SE_LAUNCH_CHECK();
  }
  return SE_OK;
}
