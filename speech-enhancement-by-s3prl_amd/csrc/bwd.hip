// bwd.hip -- building blocks of the backward pass through the bf16 encoder / spec head (row E2 for C3 / C4):
//   se_transpose_bf16     (rows, cols) -> (cols, ld_out) zero padded           operand layouts for the two GEMM forms below
//   se_wgrad_bf16         dW[N,K] (+)= dY[M,N]^T . X[M,K]   weight gradient: both operands transposed so the reduction
//                         dim (M = B*T) is contiguous, split-K over M on the forward GEMM kernel (gemm2), slab reduce
//   se_colsum_f32         db[N] = sum_rows dY                bias gradient
//   se_layernorm_bwd_f32  dx, dgamma, dbeta of the TF-style LayerNorm; optional GELU on the way in (spec head:
//                         LN(gelu(pre))) with gelu' applied on the way out
// The input gradient dX = dY . W needs no kernel of its own: it is se_gemm_bf16 on W^T (transposed bf16 weight copy).
// Mixed precision as the forward: bf16 GEMM operands, fp32 accumulation / reductions.
#include "common.h"
#include "bf16.h"
#include "encoder_impl.h"
#include "dropout.h"

extern "C" int se_gemm2_splitk_launch(const uint16_t* A, int lda, const uint16_t* W, int ldw, int M, int N, int Kc, int splits,
                                      float* partials, void* stream);

namespace se {

// out[c][r] = in[r][c] for r < rows, 0 for rows <= r < ld_out.  64 x 64 tiles through LDS.
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const uint16_t* __restrict__ in, int rows, int cols, int ld_in,
                                                             uint16_t* __restrict__ out, int ld_out) {
  __shared__ uint16_t tile[64][66];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? in[(size_t)r * ld_in + c] : (uint16_t)0;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < ld_out) out[(size_t)c * ld_out + r] = tile[tx][i];
  }
}

// vector form of the same (ld_in % 8 == 0, ld_out % 8 == 0, 16-B aligned bases): 16-B global loads of row pieces, 16-B
// global stores of 8 consecutive source rows of one column (8 lanes complete a 128-B line of the output row); the
// 2-byte gathers happen in LDS only (132-B row stride: the 8 row-chunks of a store wave hit distinct banks).
__global__ __launch_bounds__(256) void transpose_bf16_vec_kernel(const uint16_t* __restrict__ in, int rows, int cols, int ld_in,
                                                                 uint16_t* __restrict__ out, int ld_out) {
  __shared__ uint32_t tile[64][33];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int t = threadIdx.x;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int rl = (t >> 3) + 32 * pass, r = r0 + rl, c = c0 + (t & 7) * 8;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (r < rows) {
      if (c + 7 < cols) {
        v = *reinterpret_cast<const uint4*>(in + (size_t)r * ld_in + c);
      } else if (c < cols) {
        uint16_t e[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) e[k] = (c + k < cols) ? in[(size_t)r * ld_in + c + k] : (uint16_t)0;
        v = make_uint4(e[0] | ((uint32_t)e[1] << 16), e[2] | ((uint32_t)e[3] << 16), e[4] | ((uint32_t)e[5] << 16), e[6] | ((uint32_t)e[7] << 16));
      }
    }
    uint32_t* d = &tile[rl][(t & 7) * 4];
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
  __syncthreads();
  const uint16_t* t16 = reinterpret_cast<const uint16_t*>(&tile[0][0]);
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int cl = (t >> 3) + 32 * pass, m = (t & 7) * 8;
    const int c = c0 + cl, r = r0 + m;
    if (c < cols && r < ld_out) {
      uint16_t e[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) e[k] = t16[(m + k) * 66 + cl];
      *reinterpret_cast<uint4*>(out + (size_t)c * ld_out + r) =
          make_uint4(e[0] | ((uint32_t)e[1] << 16), e[2] | ((uint32_t)e[3] << 16), e[4] | ((uint32_t)e[5] << 16), e[6] | ((uint32_t)e[7] << 16));
    }
  }
}

// same from an fp32 source (casts on the way): out[c][r] = bf16(in[r][c])
__global__ __launch_bounds__(256) void transpose_f32_bf16_kernel(const float* __restrict__ in, int rows, int cols, int ld_in,
                                                                 uint16_t* __restrict__ out, int ld_out) {
  __shared__ uint16_t tile[64][66];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? f2bf(in[(size_t)r * ld_in + c]) : (uint16_t)0;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < ld_out) out[(size_t)c * ld_out + r] = tile[tx][i];
  }
}

__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ partials, int splits, size_t n, int accumulate,
                                                          float* __restrict__ out) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float s = accumulate ? out[i] : 0.f;
    for (int k = 0; k < splits; ++k) s += partials[(size_t)k * n + i];
    out[i] = s;
  }
}

// colsum[c] += sum_r x[r][c]; grid.x over row chunks, 256 threads = 256 columns per grid.y
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int rows, int cols, int ld, int rows_per_block,
                                                     float* __restrict__ out) {
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= cols) return;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += x[(size_t)r * ld + c];
  atomicAdd(&out[c], s);
}

__device__ __forceinline__ float gelu_grad(float x) {
  // d/dx [x Phi(x)] = Phi(x) + x phi(x)
  const float cdf = 0.5f * (1.0f + erf_as(x * 0.70710678118654752f));
  const float pdf = 0.3989422804014327f * __builtin_amdgcn_exp2f(-0.72134752044448170f * x * x);   // exp(-x^2/2)/sqrt(2 pi)
  return fmaf(x, pdf, cdf);
}

// y = gelu(x), bf16 -> bf16, 8 elements per thread (FFN activation of the training forward, which keeps x)
__global__ __launch_bounds__(256) void gelu_bf16_kernel(const uint16_t* __restrict__ x, size_t n8, uint16_t* __restrict__ y) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const uint4 v = reinterpret_cast<const uint4*>(x)[i];
    const uint32_t in[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = pack_bf16x2(gelu_erf(bf2f((uint16_t)(in[k] & 0xffffu))), gelu_erf(bf2f((uint16_t)(in[k] >> 16))));
    reinterpret_cast<uint4*>(y)[i] = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// dx = dy * gelu'(x), all bf16
__global__ __launch_bounds__(256) void gelu_bwd_bf16_kernel(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ x, size_t n8,
                                                            uint16_t* __restrict__ dx) {
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const uint4 g = reinterpret_cast<const uint4*>(dy)[i];
    const uint4 v = reinterpret_cast<const uint4*>(x)[i];
    const uint32_t gi[4] = {g.x, g.y, g.z, g.w}, vi[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      o[k] = pack_bf16x2(bf2f((uint16_t)(gi[k] & 0xffffu)) * gelu_grad(bf2f((uint16_t)(vi[k] & 0xffffu))),
                         bf2f((uint16_t)(gi[k] >> 16)) * gelu_grad(bf2f((uint16_t)(vi[k] >> 16))));
    reinterpret_cast<uint4*>(dx)[i] = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// Column sums of a bf16 matrix (bias gradients), optionally fused with gelu': GELU_BWD: dx = dy * gelu'(pre) is written and
// summed in the same pass.  Lane = 8 consecutive columns (16-B loads), wave = 512 columns of one row, the 4 waves of a
// workgroup interleave rows; per-workgroup LDS reduction, then one atomic per column.  cols % 8 == 0.
template <int GELU_BWD>
__global__ __launch_bounds__(256) void colsum8_bf16_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ pre,
                                                           uint16_t* __restrict__ dx, int rows, int cols, int ld, int rows_per_block,
                                                           float* __restrict__ out) {
  __shared__ float red[4][512];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = blockIdx.y * 512 + lane * 8;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  if (c < cols) {
    for (int r = r0 + wv; r < r1; r += 4) {
      const size_t o = (size_t)r * ld + c;
      const uint4 v = *reinterpret_cast<const uint4*>(x + o);
      const uint32_t vi[4] = {v.x, v.y, v.z, v.w};
      if (GELU_BWD) {
        const uint4 pz = *reinterpret_cast<const uint4*>(pre + o);
        const uint32_t pi[4] = {pz.x, pz.y, pz.z, pz.w};
        uint32_t ov[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float d0 = bf2f((uint16_t)(vi[k] & 0xffffu)) * gelu_grad(bf2f((uint16_t)(pi[k] & 0xffffu)));
          const float d1 = bf2f((uint16_t)(vi[k] >> 16)) * gelu_grad(bf2f((uint16_t)(pi[k] >> 16)));
          ov[k] = pack_bf16x2(d0, d1);
          // sum what is stored (the bf16-rounded gradient the weight-gradient GEMM will also see)
          acc[2 * k] += bf2f((uint16_t)(ov[k] & 0xffffu));
          acc[2 * k + 1] += bf2f((uint16_t)(ov[k] >> 16));
        }
        *reinterpret_cast<uint4*>(dx + o) = make_uint4(ov[0], ov[1], ov[2], ov[3]);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          acc[2 * k] += bf2f((uint16_t)(vi[k] & 0xffffu));
          acc[2 * k + 1] += bf2f((uint16_t)(vi[k] >> 16));
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) red[wv][lane * 8 + k] = acc[k];
  __syncthreads();
  for (int j = threadIdx.x; j < 512; j += 256) {
    const int cc = blockIdx.y * 512 + j;
    if (cc < cols) atomicAdd(&out[cc], (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]));
  }
}

// One wave per row, H = 256 NV.  x_in: LayerNorm input (or its pre-GELU value when GELU_IN); dy: gradient of the output.
// dx (fp32) and / or dx_bf16 written; dgamma / dbeta accumulated with one atomic per column per workgroup.
template <int NV, int GELU_IN>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ x_in, const float* __restrict__ pe, int T,
                                                            const float* __restrict__ dy,
                                                            const float* __restrict__ w, int M, float eps, float* __restrict__ dx,
                                                            uint16_t* __restrict__ dx_bf16, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ dbias, int rows_per_wave,
                                                            uint32_t key_dy, uint32_t key_dx, uint32_t thr16, float dscale, int group_rows) {
  constexpr int H = 256 * NV;
  // group_rows > 0: blockIdx.y = group g owns rows [g group_rows, (g + 1) group_rows) and its own (dgamma, dbeta, dbias) rows --
  // per-utterance parameter gradients in one launch (scoring.py); 0: one group of M rows
  const int gbase = blockIdx.y * group_rows;
  const int gend = group_rows ? min(M, gbase + group_rows) : M;
  if (dgamma) dgamma += (size_t)blockIdx.y * H;
  if (dbeta) dbeta += (size_t)blockIdx.y * H;
  if (dbias) dbias += (size_t)blockIdx.y * H;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float4 gsum[NV], bsum[NV], xsum[NV];     // column sums of dy xhat (dgamma), dy (dbeta), dx (bias gradient of the producing linear)
#pragma unroll
  for (int i = 0; i < NV; ++i) { gsum[i] = make_float4(0.f, 0.f, 0.f, 0.f); bsum[i] = make_float4(0.f, 0.f, 0.f, 0.f); xsum[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
  float4 ww[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) ww[i] = *reinterpret_cast<const float4*>(w + (i * 64 + lane) * 4);
  const int row0 = gbase + (blockIdx.x * 4 + wv) * rows_per_wave;
  for (int row = row0; row < min(gend, row0 + rows_per_wave); ++row) {
    float4 pre[NV], v[NV], g[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      pre[i] = *reinterpret_cast<const float4*>(x_in + (size_t)row * H + (i * 64 + lane) * 4);
      g[i] = *reinterpret_cast<const float4*>(dy + (size_t)row * H + (i * 64 + lane) * 4);
      if (key_dy) {           // the forward dropped this LayerNorm's OUTPUT (input stage): mask the incoming gradient
        const uint32_t pr = (uint32_t)row * (H / 2) + (uint32_t)(i * 64 + lane) * 2;
        const uint32_t b0 = dropout_bits(key_dy, pr), b1 = dropout_bits(key_dy, pr + 1);
        g[i].x *= dropout_mul(b0, 0, thr16, dscale); g[i].y *= dropout_mul(b0, 1, thr16, dscale);
        g[i].z *= dropout_mul(b1, 0, thr16, dscale); g[i].w *= dropout_mul(b1, 1, thr16, dscale);
      }
      if (pe) {               // input stage: LayerNorm(x W^T + b + positional encoding)
        const float4 pp = *reinterpret_cast<const float4*>(pe + (size_t)(row % T) * H + (i * 64 + lane) * 4);
        pre[i].x += pp.x; pre[i].y += pp.y; pre[i].z += pp.z; pre[i].w += pp.w;
      }
      v[i] = pre[i];
      if (GELU_IN) { v[i].x = gelu_erf(v[i].x); v[i].y = gelu_erf(v[i].y); v[i].z = gelu_erf(v[i].z); v[i].w = gelu_erf(v[i].w); }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    const float mean = s * (1.0f / H);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
      q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off);
    const float rstd = 1.0f / sqrtf(q * (1.0f / H) + eps);
    // xhat = v * rstd ; gh = dy * w ; dx = rstd (gh - mean(gh) - xhat mean(gh xhat))
    float a = 0.f, bq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      v[i].x *= rstd; v[i].y *= rstd; v[i].z *= rstd; v[i].w *= rstd;
      gsum[i].x += g[i].x * v[i].x; gsum[i].y += g[i].y * v[i].y; gsum[i].z += g[i].z * v[i].z; gsum[i].w += g[i].w * v[i].w;
      bsum[i].x += g[i].x; bsum[i].y += g[i].y; bsum[i].z += g[i].z; bsum[i].w += g[i].w;
      g[i].x *= ww[i].x; g[i].y *= ww[i].y; g[i].z *= ww[i].z; g[i].w *= ww[i].w;
      a += (g[i].x + g[i].y) + (g[i].z + g[i].w);
      bq += (g[i].x * v[i].x + g[i].y * v[i].y) + (g[i].z * v[i].z + g[i].w * v[i].w);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); bq += __shfl_xor(bq, off); }
    a *= (1.0f / H);
    bq *= (1.0f / H);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float4 d;
      d.x = rstd * (g[i].x - a - v[i].x * bq); d.y = rstd * (g[i].y - a - v[i].y * bq);
      d.z = rstd * (g[i].z - a - v[i].z * bq); d.w = rstd * (g[i].w - a - v[i].w * bq);
      if (GELU_IN) { d.x *= gelu_grad(pre[i].x); d.y *= gelu_grad(pre[i].y); d.z *= gelu_grad(pre[i].z); d.w *= gelu_grad(pre[i].w); }
      const size_t o = (size_t)row * H + (i * 64 + lane) * 4;
      if (dx) *reinterpret_cast<float4*>(dx + o) = d;           // residual branch: unmasked
      if (key_dx) {           // the forward dropped the producing linear's output before the residual add: its gradient is masked
        const uint32_t pr = (uint32_t)row * (H / 2) + (uint32_t)(i * 64 + lane) * 2;
        const uint32_t b0 = dropout_bits(key_dx, pr), b1 = dropout_bits(key_dx, pr + 1);
        d.x *= dropout_mul(b0, 0, thr16, dscale); d.y *= dropout_mul(b0, 1, thr16, dscale);
        d.z *= dropout_mul(b1, 0, thr16, dscale); d.w *= dropout_mul(b1, 1, thr16, dscale);
      }
      xsum[i].x += d.x; xsum[i].y += d.y; xsum[i].z += d.z; xsum[i].w += d.w;
      if (dx_bf16) *reinterpret_cast<uint2*>(dx_bf16 + o) = make_uint2(pack_bf16x2(d.x, d.y), pack_bf16x2(d.z, d.w));
    }
  }
  // dgamma / dbeta: reduce the 4 waves through LDS, then one atomic per column per workgroup
  __shared__ float red[3][4][H];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    *reinterpret_cast<float4*>(&red[0][wv][(i * 64 + lane) * 4]) = gsum[i];
    *reinterpret_cast<float4*>(&red[1][wv][(i * 64 + lane) * 4]) = bsum[i];
    *reinterpret_cast<float4*>(&red[2][wv][(i * 64 + lane) * 4]) = xsum[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < H; c += 256) {
    if (dgamma) atomicAdd(&dgamma[c], red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
    if (dbeta) atomicAdd(&dbeta[c], red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
    if (dbias) atomicAdd(&dbias[c], red[2][0][c] + red[2][1][c] + red[2][2][c] + red[2][3][c]);
  }
}

}  // namespace se

extern "C" int se_transpose_bf16(const uint16_t* in, int rows, int cols, int ld_in, uint16_t* out, int ld_out, void* stream) {
  SE_REQUIRE(in && out && rows > 0 && cols > 0 && ld_in >= cols && ld_out >= rows, "se_transpose_bf16: bad argument");
  dim3 grid((ld_out + 63) / 64, (cols + 63) / 64);
  SE_REQUIRE(grid.y <= 65535, "se_transpose_bf16: too many columns");
  if (ld_in % 8 == 0 && ld_out % 8 == 0 && (((uintptr_t)in | (uintptr_t)out) & 15) == 0)
    hipLaunchKernelGGL(se::transpose_bf16_vec_kernel, grid, dim3(256), 0, se::as_stream(stream), in, rows, cols, ld_in, out, ld_out);
  else
    hipLaunchKernelGGL(se::transpose_bf16_kernel, grid, dim3(256), 0, se::as_stream(stream), in, rows, cols, ld_in, out, ld_out);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_transpose_f32_bf16(const float* in, int rows, int cols, int ld_in, uint16_t* out, int ld_out, void* stream) {
  SE_REQUIRE(in && out && rows > 0 && cols > 0 && ld_in >= cols && ld_out >= rows, "se_transpose_f32_bf16: bad argument");
  dim3 grid((ld_out + 63) / 64, (cols + 63) / 64);
  SE_REQUIRE(grid.y <= 65535, "se_transpose_f32_bf16: too many columns");
  hipLaunchKernelGGL(se::transpose_f32_bf16_kernel, grid, dim3(256), 0, se::as_stream(stream), in, rows, cols, ld_in, out, ld_out);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

// dW[N,K] (+)= dYt[N,Mp] . Xt[K,Mp]^T ; Mp = splits * Mc, Mc % 64 == 0, Mc >= 128 ; workspace >= splits * N * K floats
extern "C" int se_wgrad_bf16(const uint16_t* dYt, const uint16_t* Xt, int Mp, int N, int K, int splits, float* dW, int accumulate,
                             void* workspace, size_t workspace_bytes, void* stream) {
  SE_REQUIRE(dYt && Xt && dW && workspace, "se_wgrad_bf16: null argument");
  SE_REQUIRE(splits >= 1 && Mp % splits == 0, "se_wgrad_bf16: Mp must be a multiple of splits");
  const int Mc = Mp / splits;
  SE_REQUIRE(Mc % 64 == 0 && Mc >= 128, "se_wgrad_bf16: Mp / splits = %d must be a multiple of 64 and >= 128", Mc);
  SE_REQUIRE(workspace_bytes >= (size_t)splits * N * K * sizeof(float), "se_wgrad_bf16: workspace too small");
  SE_REQUIRE((((uintptr_t)dYt | (uintptr_t)Xt | (uintptr_t)workspace) % 16) == 0, "se_wgrad_bf16: operands must be 16-B aligned");
  float* partials = reinterpret_cast<float*>(workspace);
  int rc = se_gemm2_splitk_launch(dYt, Mp, Xt, Mp, N, K, Mc, splits, partials, stream);
  if (rc) return rc;
  const size_t n = (size_t)N * K;
  hipLaunchKernelGGL(se::slab_reduce_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)), dim3(256), 0, se::as_stream(stream),
                     partials, splits, n, accumulate, dW);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

namespace se {
// out[g][c] = sum over the rows of group g of x[(g * rows + r) * ld + c]: per-utterance bias gradients (active-sampling scoring) in ONE launch.
// Workgroup = (64-column strip, group); 4 waves walk the group's rows 4 apart, lanes own one column each; LDS combine.
template <typename TIn>
__global__ __launch_bounds__(256) void colsum_groups_kernel(const TIn* __restrict__ x, int rows, int cols, int ld, float* __restrict__ out) {
  __shared__ float part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6, g = blockIdx.y;
  float acc = 0.f;
  if (c < cols) {
    const TIn* p = x + (size_t)g * rows * ld + c;
    for (int r = w; r < rows; r += 4) {
      if constexpr (sizeof(TIn) == 2) acc += bf2f(p[(size_t)r * ld]);
      else acc += p[(size_t)r * ld];
    }
  }
  part[w][threadIdx.x & 63] = acc;
  __syncthreads();
  if (w == 0 && c < cols) out[(size_t)g * cols + c] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}
}  // namespace se

extern "C" int se_colsum_groups(const void* x, int is_bf16, int groups, int rows, int cols, int ld, float* out, void* stream) {
  SE_REQUIRE(x && out && groups > 0 && groups <= 65535 && rows > 0 && cols > 0 && ld >= cols, "se_colsum_groups: bad argument");
  dim3 grid((cols + 63) / 64, groups);
  if (is_bf16)
    hipLaunchKernelGGL(se::colsum_groups_kernel<uint16_t>, grid, dim3(256), 0, se::as_stream(stream), static_cast<const uint16_t*>(x), rows, cols, ld, out);
  else
    hipLaunchKernelGGL(se::colsum_groups_kernel<float>, grid, dim3(256), 0, se::as_stream(stream), static_cast<const float*>(x), rows, cols, ld, out);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_colsum_f32(const float* x, int rows, int cols, int ld, float* out, int accumulate, void* stream) {
  SE_REQUIRE(x && out && rows > 0 && cols > 0 && ld >= cols, "se_colsum_f32: bad argument");
  hipStream_t st = se::as_stream(stream);
  if (!accumulate) { const int zrc_ = se::zero_async(out, sizeof(float) * cols, st); if (zrc_) return zrc_; }
  const int rpb = 256;
  dim3 grid((rows + rpb - 1) / rpb, (cols + 255) / 256);
  hipLaunchKernelGGL(se::colsum_kernel, grid, dim3(256), 0, st, x, rows, cols, ld, rpb, out);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

int se::launch_layernorm_bwd(const float* x_in, const float* pe, int T, const float* dy, const float* w, int M, int H, float eps, int gelu_in,
                             float* dx, uint16_t* dx_bf16, float* dgamma, float* dbeta, float* dbias, int accumulate, hipStream_t st,
                             uint32_t key_dy, uint32_t key_dx, uint32_t thr16, float dscale, int group_rows) {
  SE_REQUIRE(H == 768 || H == 256, "layernorm backward: only H = 768 / 256 are built (got %d)", H);
  const int groups = group_rows ? (M + group_rows - 1) / group_rows : 1;
  SE_REQUIRE(groups <= 65535, "layernorm backward: %d groups exceed grid.y", groups);
  if (!accumulate) {
    const size_t zb = sizeof(float) * (size_t)H * groups;
    const int zrc_ = se::zero_async3(dgamma, zb, dbeta, zb, dbias, zb, st);
    if (zrc_) return zrc_;
  }
  const int rows_per_wave = 16;
  const dim3 grid(((group_rows ? group_rows : M) + 4 * rows_per_wave - 1) / (4 * rows_per_wave), groups);
  if (H == 768) {
    if (gelu_in)
      hipLaunchKernelGGL((se::layernorm_bwd_kernel<3, 1>), grid, dim3(256), 0, st, x_in, pe, T, dy, w, M, eps, dx, dx_bf16, dgamma, dbeta, dbias, rows_per_wave, key_dy, key_dx, thr16, dscale, group_rows);
    else
      hipLaunchKernelGGL((se::layernorm_bwd_kernel<3, 0>), grid, dim3(256), 0, st, x_in, pe, T, dy, w, M, eps, dx, dx_bf16, dgamma, dbeta, dbias, rows_per_wave, key_dy, key_dx, thr16, dscale, group_rows);
  } else {
    if (gelu_in)
      hipLaunchKernelGGL((se::layernorm_bwd_kernel<1, 1>), grid, dim3(256), 0, st, x_in, pe, T, dy, w, M, eps, dx, dx_bf16, dgamma, dbeta, dbias, rows_per_wave, key_dy, key_dx, thr16, dscale, group_rows);
    else
      hipLaunchKernelGGL((se::layernorm_bwd_kernel<1, 0>), grid, dim3(256), 0, st, x_in, pe, T, dy, w, M, eps, dx, dx_bf16, dgamma, dbeta, dbias, rows_per_wave, key_dy, key_dx, thr16, dscale, group_rows);
  }
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_layernorm_bwd_f32(const float* x_in, const float* dy, const float* w, int M, int H, float eps, int gelu_in,
                                    float* dx, uint16_t* dx_bf16, float* dgamma, float* dbeta, int accumulate, void* stream) {
  SE_REQUIRE(x_in && dy && w && (dx || dx_bf16) && M > 0, "se_layernorm_bwd_f32: bad argument");
  return se::launch_layernorm_bwd(x_in, nullptr, 1, dy, w, M, H, eps, gelu_in, dx, dx_bf16, dgamma, dbeta, nullptr, accumulate, se::as_stream(stream));
}

// the same with the rows cut into `groups` blocks of `rows` and one (dgamma, dbeta) row per block: the per-utterance LayerNorm parameter
// gradients of the active-sampling scoring (sampler.py:84-104) in ONE launch
extern "C" int se_layernorm_bwd_groups_f32(const float* x_in, const float* dy, const float* w, int groups, int rows, int H, float eps, int gelu_in,
                                           float* dx, uint16_t* dx_bf16, float* dgamma, float* dbeta, void* stream) {
  SE_REQUIRE(x_in && dy && w && (dx || dx_bf16) && groups > 0 && rows > 0, "se_layernorm_bwd_groups_f32: bad argument");
  SE_REQUIRE((long)groups * rows <= 0x7fffffffL, "se_layernorm_bwd_groups_f32: too many rows");
  return se::launch_layernorm_bwd(x_in, nullptr, 1, dy, w, groups * rows, H, eps, gelu_in, dx, dx_bf16, dgamma, dbeta, nullptr, 0, se::as_stream(stream), 0, 0,
                                  0, 1.f, rows);
}

int se::launch_colsum_bf16(const uint16_t* x, int rows, int cols, int ld, float* out, hipStream_t st) {
  SE_REQUIRE(cols % 8 == 0 && ld % 8 == 0, "bf16 column sum: cols and ld must be multiples of 8");
  { const int zrc_ = se::zero_async(out, sizeof(float) * cols, st); if (zrc_) return zrc_; }
  const int rpb = 128;
  dim3 grid((rows + rpb - 1) / rpb, (cols + 511) / 512);
  hipLaunchKernelGGL((se::colsum8_bf16_kernel<0>), grid, dim3(256), 0, st, x, nullptr, nullptr, rows, cols, ld, rpb, out);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

// dx = dy * gelu'(pre) (bf16, in place allowed) and colsum[c] = sum_rows dx  (FFN: activation backward + bias gradient)
int se::launch_gelu_bwd_colsum(const uint16_t* dy, const uint16_t* pre, uint16_t* dx, int rows, int cols, float* colsum, hipStream_t st) {
  SE_REQUIRE(cols % 8 == 0, "gelu backward: cols must be a multiple of 8");
  { const int zrc_ = se::zero_async(colsum, sizeof(float) * cols, st); if (zrc_) return zrc_; }
  const int rpb = 128;
  dim3 grid((rows + rpb - 1) / rpb, (cols + 511) / 512);
  hipLaunchKernelGGL((se::colsum8_bf16_kernel<1>), grid, dim3(256), 0, st, dy, pre, dx, rows, cols, cols, rpb, colsum);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_gelu_bf16(const uint16_t* x, size_t n, uint16_t* y, void* stream) {
  SE_REQUIRE(x && y && n > 0 && n % 8 == 0, "se_gelu_bf16: n must be a positive multiple of 8");
  const size_t n8 = n / 8;
  hipLaunchKernelGGL(se::gelu_bf16_kernel, dim3((unsigned)std::min<size_t>((n8 + 255) / 256, 16384)), dim3(256), 0, se::as_stream(stream), x, n8, y);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_gelu_bwd_bf16(const uint16_t* dy, const uint16_t* x, size_t n, uint16_t* dx, void* stream) {
  SE_REQUIRE(dy && x && dx && n > 0 && n % 8 == 0, "se_gelu_bwd_bf16: n must be a positive multiple of 8");
  const size_t n8 = n / 8;
  hipLaunchKernelGGL(se::gelu_bwd_bf16_kernel, dim3((unsigned)std::min<size_t>((n8 + 255) / 256, 16384)), dim3(256), 0, se::as_stream(stream), dy, x, n8, dx);
  SE_LAUNCH_CHECK();
  return SE_OK;
}
