// optim.hip -- row E2's optimizer side on the device: gradient norms and the BertAdam update for ALL parameter tensors in
// two launches (the reference does ~15 small torch kernels per parameter tensor: runner.py:463-471 -> clip_grad_norm_,
// S3PRL BertAdam.step).  HBM-bound: 16 B read + 12 B written per parameter element in the update, 4 B read in the norm.
//
//   se_multi_sumsq_f32    sumsq[t] = sum g_t^2                       (one launch; fp64 atomics per 16 Ki-element chunk)
//   se_bertadam_step_f32  c_g = min(1, G / (sqrt(sum_t sumsq[t]) + 1e-6))      global clip (runner.py:464), G <= 0: off
//                         c_t = min(1, C / (c_g sqrt(sumsq[t]) + 1e-6))         BertAdam's per-parameter clip, C <= 0: off
//                         g' = c_g c_t g ; m = b1 m + (1-b1) g' ; v = b2 v + (1-b2) g'^2
//                         p -= lr_t (m / (sqrt(v) + e) + wd_t p)                no bias correction, decoupled decay
// Tensor tables travel in the kernel arguments (<= 64 tensors per launch), so nothing is staged in device memory and
// the call is asynchronous.
#include <math.h>
#include <algorithm>
#include "common.h"
#include "bf16.h"
#include "multi_copy.h"

namespace se {

constexpr int kOptSlots = 64;
constexpr uint32_t kOptChunk = 16384;      // elements per workgroup

struct OptSlot {
  float* p;
  const float* g;
  float* m;
  float* v;
  uint32_t n;
  float wd;
  uint32_t chunk0;      // first workgroup of this tensor inside the launch
  uint32_t tensor;      // index into sumsq
};
struct OptBatch {
  OptSlot s[kOptSlots];
  int count;
};

__device__ __forceinline__ int find_slot(const OptBatch& b, uint32_t blk) {
  int lo = 0;
  for (int i = 1; i < b.count; ++i)
    if (b.s[i].chunk0 <= blk) lo = i;
  return lo;
}

__global__ __launch_bounds__(256) void multi_sumsq_kernel(const OptBatch b, double* __restrict__ sumsq) {
  __shared__ float red[4];
  const int si = find_slot(b, blockIdx.x);
  const OptSlot& s = b.s[si];
  const uint32_t e0 = (blockIdx.x - s.chunk0) * kOptChunk, e1 = min(s.n, e0 + kOptChunk);
  float acc = 0.f;
  const float* g = s.g;
  if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
    uint32_t i = e0 + threadIdx.x * 4;
    for (; i + 3 < e1; i += 1024) {
      const float4 t = *reinterpret_cast<const float4*>(g + i);
      acc += (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
    }
    for (; i < e1; ++i) acc += g[i] * g[i];       // ragged tail of the last chunk (at most 3 elements on one thread)
  } else {
    for (uint32_t i = e0 + threadIdx.x; i < e1; i += 256) acc += g[i] * g[i];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&sumsq[s.tensor], (double)((red[0] + red[1]) + (red[2] + red[3])));
}

__global__ __launch_bounds__(256) void bertadam_kernel(const OptBatch b, const double* __restrict__ sumsq, int n_tensors, float lr_t, float b1,
                                                       float ob1, float b2, float ob2, float e, float max_grad_norm, float global_max_norm) {
  __shared__ float coef_s;
  const int si = find_slot(b, blockIdx.x);
  const OptSlot& s = b.s[si];
  if (threadIdx.x == 0) {
    float cg = 1.f;
    if (global_max_norm > 0.f) {
      double tot = 0.0;
      for (int t = 0; t < n_tensors; ++t) tot += sumsq[t];
      cg = fminf(1.f, global_max_norm / ((float)sqrt(tot) + 1e-6f));
    }
    float ct = 1.f;
    if (max_grad_norm > 0.f) ct = fminf(1.f, max_grad_norm / (cg * (float)sqrt(sumsq[s.tensor]) + 1e-6f));
    coef_s = cg * ct;
  }
  __syncthreads();
  const float coef = coef_s, wd = s.wd;
  const uint32_t e0 = (blockIdx.x - s.chunk0) * kOptChunk, e1 = min(s.n, e0 + kOptChunk);
  float* p = s.p;
  const float* g = s.g;
  float* m = s.m;
  float* v = s.v;
  const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                     reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  auto upd = [&](float& pp, float gg, float& mm, float& vv) {
    gg *= coef;
    mm = b1 * mm + ob1 * gg;
    vv = b2 * vv + ob2 * gg * gg;
    pp -= lr_t * (mm / (sqrtf(vv) + e) + wd * pp);
  };
  uint32_t i = e0 + (vec ? threadIdx.x * 4 : threadIdx.x);
  if (vec) {
    for (; i + 3 < e1; i += 1024) {
      float4 pp = *reinterpret_cast<float4*>(p + i), mm = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
      const float4 gg = *reinterpret_cast<const float4*>(g + i);
      upd(pp.x, gg.x, mm.x, vv.x); upd(pp.y, gg.y, mm.y, vv.y); upd(pp.z, gg.z, mm.z, vv.z); upd(pp.w, gg.w, mm.w, vv.w);
      *reinterpret_cast<float4*>(p + i) = pp;
      *reinterpret_cast<float4*>(m + i) = mm;
      *reinterpret_cast<float4*>(v + i) = vv;
    }
    for (; i < e1; ++i) upd(p[i], g[i], m[i], v[i]);
  } else {
    for (; i < e1; i += 256) upd(p[i], g[i], m[i], v[i]);
  }
}

// ---- many small device-to-device copies / fp32 -> bf16 casts in one launch (per-step plumbing of the training path: the refresh
//      of the encoder's bf16 weights, the split of the fused QKV gradients, the gather into the flat all-reduce buffer: ~230 launches
//      of 3-5 us each per step otherwise)
constexpr int kCopySlots = 96;
constexpr uint32_t kCopyChunk = 16384;     // elements per workgroup
struct CopySlot {
  void* dst;
  const float* src;
  uint32_t n;          // copy: elements; cast: rows * ld output elements
  uint32_t chunk0;
  uint32_t cols, ld;   // cast only: source row length, padded output row length
};
struct CopyBatch {
  CopySlot s[kCopySlots];
  int count;
};

__device__ __forceinline__ int find_copy_slot(const CopyBatch& b, uint32_t blk) {
  int lo = 0;
  for (int i = 1; i < b.count; ++i)
    if (b.s[i].chunk0 <= blk) lo = i;
  return lo;
}

__global__ __launch_bounds__(256) void multi_copy_kernel(const CopyBatch b) {
  const CopySlot& s = b.s[find_copy_slot(b, blockIdx.x)];
  const uint32_t e0 = (blockIdx.x - s.chunk0) * kCopyChunk, e1 = min(s.n, e0 + kCopyChunk);
  float* d = static_cast<float*>(s.dst);
  const float* g = s.src;
  if (((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(g)) & 15) == 0) {
    uint32_t i = e0 + threadIdx.x * 4;
    for (; i + 3 < e1; i += 1024) *reinterpret_cast<float4*>(d + i) = *reinterpret_cast<const float4*>(g + i);
    for (; i < e1; ++i) d[i] = g[i];              // ragged tail (at most 3 elements on one thread)
  } else {
    for (uint32_t i = e0 + threadIdx.x; i < e1; i += 256) d[i] = g[i];
  }
}

// out[r * ld + c] = c < cols ? bf16(src[r * cols + c]) : 0      (ld % 4 == 0)
__global__ __launch_bounds__(256) void multi_cast_pad_kernel(const CopyBatch b) {
  const CopySlot& s = b.s[find_copy_slot(b, blockIdx.x)];
  const uint32_t e0 = (blockIdx.x - s.chunk0) * kCopyChunk, e1 = min(s.n, e0 + kCopyChunk);
  uint16_t* d = static_cast<uint16_t*>(s.dst);
  const uint32_t cols = s.cols, ld = s.ld;
  const bool vec = (cols % 4 == 0) && ((reinterpret_cast<uintptr_t>(s.src) & 15) == 0);
  for (uint32_t i = e0 + threadIdx.x * 4; i < e1; i += 1024) {
    const uint32_t r = i / ld, c = i - r * ld;
    const float* xr = s.src + (size_t)r * cols;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
    if (vec && c + 3 < cols) {
      const float4 t = *reinterpret_cast<const float4*>(xr + c);
      v0 = t.x; v1 = t.y; v2 = t.z; v3 = t.w;
    } else {
      if (c < cols) v0 = xr[c];
      if (c + 1 < cols) v1 = xr[c + 1];
      if (c + 2 < cols) v2 = xr[c + 2];
      if (c + 3 < cols) v3 = xr[c + 3];
    }
    *reinterpret_cast<uint2*>(d + i) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
  }
}

int MultiCopy::add(void* dst, const float* src, size_t n, uint32_t cols, uint32_t ld) {
  if (n == 0) return SE_OK;
  if (n > 0xffffffffull) {
    set_error("multi copy: %llu elements (> 2^32 - 1)", (unsigned long long)n);
    return SE_ERR_INVALID;
  }
  if (b_->count == kCopySlots) {
    const int rc = flush();
    if (rc) return rc;
  }
  CopySlot& s = b_->s[b_->count++];
  s.dst = dst; s.src = src; s.n = (uint32_t)n; s.chunk0 = chunks_; s.cols = cols; s.ld = ld;
  chunks_ += ((uint32_t)n + kCopyChunk - 1) / kCopyChunk;
  return SE_OK;
}

int MultiCopy::flush() {
  if (b_->count == 0) return SE_OK;
  if (cast_) hipLaunchKernelGGL(multi_cast_pad_kernel, dim3(chunks_), dim3(256), 0, st_, *b_);
  else hipLaunchKernelGGL(multi_copy_kernel, dim3(chunks_), dim3(256), 0, st_, *b_);
  SE_LAUNCH_CHECK();
  b_->count = 0;
  chunks_ = 0;
  return SE_OK;
}

MultiCopy::MultiCopy(bool cast, hipStream_t st) : b_(new CopyBatch), chunks_(0), cast_(cast), st_(st) { b_->count = 0; }
MultiCopy::~MultiCopy() { delete b_; }

template <typename F>
int for_batches(float* const* params, const float* const* grads, float* const* m, float* const* v, const uint64_t* sizes,
                const float* weight_decay, int n_tensors, F&& launch) {
  int t = 0;
  while (t < n_tensors) {
    OptBatch b;
    b.count = 0;
    uint32_t chunks = 0;
    while (t < n_tensors && b.count < kOptSlots) {
      if (sizes[t] == 0) { ++t; continue; }
      if (sizes[t] > 0xffffffffull) {
        set_error("optimizer: tensor %d has %llu elements (> 2^32 - 1)", t, (unsigned long long)sizes[t]);
        return SE_ERR_INVALID;
      }
      OptSlot& s = b.s[b.count++];
      s.p = params ? params[t] : nullptr;
      s.g = grads[t];
      s.m = m ? m[t] : nullptr;
      s.v = v ? v[t] : nullptr;
      s.n = (uint32_t)sizes[t];
      s.wd = weight_decay ? weight_decay[t] : 0.f;
      s.chunk0 = chunks;
      s.tensor = (uint32_t)t;
      chunks += (s.n + kOptChunk - 1) / kOptChunk;
      ++t;
    }
    if (b.count == 0) break;
    const int rc = launch(b, chunks);
    if (rc) return rc;
  }
  return SE_OK;
}

}  // namespace se

extern "C" int se_multi_sumsq_f32(const float* const* grads, const uint64_t* sizes, int n_tensors, double* sumsq, void* stream) {
  SE_REQUIRE(grads && sizes && sumsq && n_tensors > 0, "se_multi_sumsq_f32: bad argument");
  hipStream_t st = se::as_stream(stream);
  { const int zrc_ = se::zero_async(sumsq, sizeof(double) * n_tensors, st); if (zrc_) return zrc_; }
  return se::for_batches(nullptr, grads, nullptr, nullptr, sizes, nullptr, n_tensors, [&](const se::OptBatch& b, uint32_t chunks) {
    hipLaunchKernelGGL(se::multi_sumsq_kernel, dim3(chunks), dim3(256), 0, st, b, sumsq);
    SE_LAUNCH_CHECK();
    return (int)SE_OK;
  });
}

extern "C" int se_bertadam_step_f32(float* const* params, const float* const* grads, float* const* m, float* const* v, const uint64_t* sizes,
                                    const float* weight_decay, int n_tensors, const double* sumsq, double lr_t, double b1, double b2, double e,
                                    double max_grad_norm, double global_max_norm, void* stream) {
  SE_REQUIRE(params && grads && m && v && sizes && n_tensors > 0, "se_bertadam_step_f32: bad argument");
  SE_REQUIRE(sumsq || (max_grad_norm <= 0.0 && global_max_norm <= 0.0), "se_bertadam_step_f32: clipping needs the gradient sums of squares");
  hipStream_t st = se::as_stream(stream);
  return se::for_batches(params, grads, m, v, sizes, weight_decay, n_tensors, [&](const se::OptBatch& b, uint32_t chunks) {
    hipLaunchKernelGGL(se::bertadam_kernel, dim3(chunks), dim3(256), 0, st, b, sumsq, n_tensors, (float)lr_t, (float)b1, (float)(1.0 - b1),
                       (float)b2, (float)(1.0 - b2), (float)e, (float)max_grad_norm, (float)global_max_norm);
    SE_LAUNCH_CHECK();
    return (int)SE_OK;
  });
}

extern "C" int se_multi_copy_f32(float* const* dsts, const float* const* srcs, const uint64_t* sizes, int n_tensors, void* stream) {
  SE_REQUIRE(dsts && srcs && sizes && n_tensors > 0, "se_multi_copy_f32: bad argument");
  se::MultiCopy mc(false, se::as_stream(stream));
  for (int t = 0; t < n_tensors; ++t) {
    if (sizes[t] == 0) continue;
    SE_REQUIRE(dsts[t] && srcs[t], "se_multi_copy_f32: null tensor %d", t);
    const int rc = mc.add(dsts[t], srcs[t], sizes[t]);
    if (rc) return rc;
  }
  return mc.flush();
}
