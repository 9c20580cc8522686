// encoder_train.hip -- training path of the TRANSFORMER encoder (rows B1-B3 under autograd: C4 `Mockingjay` fine-tune,
// E2; model.py:163-171, runner.py:453-471).  Forward keeps what the backward needs in a caller-owned `saved` buffer,
// backward walks the layers in reverse on the building blocks:
//   LayerNorm'            bwd.hip  (TF LayerNorm backward, dgamma / dbeta)
//   weight gradients      bwd.hip  (se_wgrad_bf16: operands transposed so M = B*T is the contiguous reduction dim, split-K)
//   input gradients       gemm*.hip (se_gemm_bf16 on a transposed bf16 copy of the weight; the residual branch's gradient
//                         rides the GEMM's fp32 residual input)
//   attention             mhsa_bwd.hip (flash backward from the stored log-sum-exp)
//   gelu'                 bwd.hip
// Mixed precision as the forward: bf16 GEMM operands, fp32 accumulation, fp32 residual-stream gradient, fp32 parameter
// gradients.  Dropout is not applied (hidden_dropout_prob / attention_probs_dropout_prob are treated as 0).
#include <stdlib.h>
#include <algorithm>
#include "common.h"
#include "encoder_impl.h"
#include "multi_copy.h"
#include "bf16.h"
#include "dropout.h"

extern "C" int se_gemm6_dual_gelu_launch(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, int M, int N, int K, uint16_t* out_pre,
                                         uint16_t* out_act, int ldc, void* stream);      // gemm6.hip
namespace se {

// Training-forward LayerNorm with the BERT dropout sites around it (one wave per row, H = 256 NV):
//   v = x (+ positional table) ; key_in: v = dropout(v) ; + residual ; [pre_out = v] ; y = LN(v) ; key_out: y = dropout(y)
// key_in is the dropout of a projection output before the residual add (attention-output / FFN-output dense), key_out the
// dropout after the input LayerNorm.  `residual` and `out_f32` may alias (row-local).
template <int NV>
__global__ __launch_bounds__(256) void ln_train_kernel(const float* __restrict__ x, const float* __restrict__ pe, int T, const float* residual,
                                                       const float* __restrict__ w, const float* __restrict__ b, int M, float eps,
                                                       float* __restrict__ pre_out, float* out_f32, uint16_t* __restrict__ out_bf16,
                                                       uint32_t key_in, uint32_t key_out, uint32_t thr16, float dscale) {
  constexpr int H = 256 * NV;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  float4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    v[i] = *reinterpret_cast<const float4*>(x + (size_t)row * H + c);
    if (pe) {
      const float4 p = *reinterpret_cast<const float4*>(pe + (size_t)(row % T) * H + c);
      v[i].x += p.x; v[i].y += p.y; v[i].z += p.z; v[i].w += p.w;
    }
    if (key_in) {
      const uint32_t pr = (uint32_t)row * (H / 2) + (uint32_t)(c >> 1);
      const uint32_t b0 = dropout_bits(key_in, pr), b1 = dropout_bits(key_in, pr + 1);
      v[i].x *= dropout_mul(b0, 0, thr16, dscale); v[i].y *= dropout_mul(b0, 1, thr16, dscale);
      v[i].z *= dropout_mul(b1, 0, thr16, dscale); v[i].w *= dropout_mul(b1, 1, thr16, dscale);
    }
    if (residual) {
      const float4 r = *reinterpret_cast<const float4*>(residual + (size_t)row * H + c);
      v[i].x += r.x; v[i].y += r.y; v[i].z += r.z; v[i].w += r.w;
    }
    if (pre_out) *reinterpret_cast<float4*>(pre_out + (size_t)row * H + c) = v[i];
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
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 64 + lane) * 4;
    const float4 ww = *reinterpret_cast<const float4*>(w + c), bb = *reinterpret_cast<const float4*>(b + c);
    float4 y;
    y.x = ww.x * (v[i].x * rstd) + bb.x; y.y = ww.y * (v[i].y * rstd) + bb.y;
    y.z = ww.z * (v[i].z * rstd) + bb.z; y.w = ww.w * (v[i].w * rstd) + bb.w;
    if (key_out) {
      const uint32_t pr = (uint32_t)row * (H / 2) + (uint32_t)(c >> 1);
      const uint32_t b0 = dropout_bits(key_out, pr), b1 = dropout_bits(key_out, pr + 1);
      y.x *= dropout_mul(b0, 0, thr16, dscale); y.y *= dropout_mul(b0, 1, thr16, dscale);
      y.z *= dropout_mul(b1, 0, thr16, dscale); y.w *= dropout_mul(b1, 1, thr16, dscale);
    }
    if (out_f32) *reinterpret_cast<float4*>(out_f32 + (size_t)row * H + c) = y;
    if (out_bf16) *reinterpret_cast<uint2*>(out_bf16 + (size_t)row * H + c) = make_uint2(pack_bf16x2(y.x, y.y), pack_bf16x2(y.z, y.w));
  }
}

}  // namespace se

namespace {

int launch_ln_train(const float* x, const float* pe, int T, const float* residual, const float* w, const float* b, int M, int H, float eps,
                    float* pre_out, float* out_f32, uint16_t* out_bf16, uint32_t key_in, uint32_t key_out, const se::DropoutCfg& d, hipStream_t st) {
  if (H == 768)
    hipLaunchKernelGGL((se::ln_train_kernel<3>), dim3((M + 3) / 4), dim3(256), 0, st, x, pe, T, residual, w, b, M, eps, pre_out, out_f32, out_bf16,
                       key_in, key_out, d.thr16, d.scale);
  else
    hipLaunchKernelGGL((se::ln_train_kernel<1>), dim3((M + 3) / 4), dim3(256), 0, st, x, pe, T, residual, w, b, M, eps, pre_out, out_f32, out_bf16,
                       key_in, key_out, d.thr16, d.scale);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

using se::al256;
constexpr int kSplits = 32;        // upper bound of the weight-gradient split count (workspace sizing)

struct SavedLayer {
  uint16_t *x0_bf, *qkv, *ctx, *x1_bf, *hpre, *h;
  float *lse, *pre1, *pre2;
};
struct Saved {
  uint16_t* xin;
  float* pre0;
  std::vector<SavedLayer> l;
  size_t total;
};

Saved carve_saved(const se_encoder* e, int B, int T, char* base) {
  const size_t M = (size_t)B * T, H = e->cfg.hidden, I = e->cfg.intermediate, heads = e->cfg.heads;
  Saved s;
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += al256(bytes); return p; };
  s.xin = (uint16_t*)take(M * se::kInPad * 2);
  s.pre0 = (float*)take(M * H * 4);
  for (int i = 0; i < e->cfg.layers; ++i) {
    SavedLayer y;
    y.x0_bf = (uint16_t*)take(M * H * 2);
    y.qkv = (uint16_t*)take(M * 3 * H * 2);
    y.lse = (float*)take((size_t)B * heads * T * 4);
    y.ctx = (uint16_t*)take(M * H * 2);
    y.pre1 = (float*)take(M * H * 4);
    y.x1_bf = (uint16_t*)take(M * H * 2);
    y.hpre = (uint16_t*)take(M * I * 2);
    y.h = (uint16_t*)take(M * I * 2);
    y.pre2 = (float*)take(M * H * 4);
    s.l.push_back(y);
  }
  s.total = off;
  return s;
}

struct TrainWs {
  float *fa, *fb;                    // fp32 (M, H): gradient of the residual stream / pre-LayerNorm gradient
  uint16_t *b1, *b2;                 // bf16 (M, H)
  uint16_t *bi;                      // bf16 (M, I)
  uint16_t *b3;                      // bf16 (M, 3H)
  float* dvec;                       // (B, heads, T)
  uint16_t* wt;                      // transposed weight copy (max 3H*H, I*H)
  float* partials;                   // split-K slabs
  float* gfused;                     // fused qkv weight gradient (3H, H) / padded input weight gradient (H, kInPad)
  float* bfused;                     // fused qkv bias gradient (3H)
  size_t total;
};

TrainWs carve_ws(const se_encoder* e, int B, int T, char* base) {
  const size_t M = (size_t)B * T, H = e->cfg.hidden, I = e->cfg.intermediate, heads = e->cfg.heads;
  const size_t wide = std::max(I, 3 * H);
  TrainWs w;
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += al256(bytes); return p; };
  w.fa = (float*)take(M * H * 4);
  w.fb = (float*)take(M * H * 4);
  w.b1 = (uint16_t*)take(M * H * 2);
  w.b2 = (uint16_t*)take(M * H * 2);
  w.bi = (uint16_t*)take(M * I * 2);
  w.b3 = (uint16_t*)take(M * 3 * H * 2);
  w.dvec = (float*)take((size_t)B * heads * T * 4);
  w.wt = (uint16_t*)take(wide * H * 2);
  w.partials = (float*)take((size_t)kSplits * wide * H * 4);
  w.gfused = (float*)take(std::max(3 * H * H, H * (size_t)se::kInPad) * 4);
  w.bfused = (float*)take(3 * H * 4);
  w.total = off;
  return w;
}

}  // namespace

#define SE_TRY(call)                 \
  do {                               \
    const int rc_ = (call);          \
    if (rc_) return rc_;             \
  } while (0)

// device fp32 master weights -> the encoder's bf16 / fp32 blob (after every optimizer step)
extern "C" int se_encoder_refresh_bf16(se_encoder* enc, const se_encoder_weights* w, void* stream) {
  SE_REQUIRE(enc && w, "se_encoder_refresh_bf16: null argument");
  const int H = enc->cfg.hidden, I = enc->cfg.intermediate, D = enc->cfg.input_dim, L = enc->cfg.layers;
  hipStream_t st = se::as_stream(stream);
  // two launches for the whole model (98 tensors): one batched cast of the matrices, one batched copy of the vectors
  se::MultiCopy cast(true, st), copy(false, st);
  auto cast_mat = [&](const float* src, int rows, int cols, int ld, uint16_t* dst) { return cast.add(dst, src, (size_t)rows * ld, cols, ld); };
  auto copy_vec = [&](const float* src, size_t n, float* dst) { return copy.add(dst, src, n); };
  SE_TRY(cast_mat(w->in_w, H, D, se::kInPad, enc->in_w));
  SE_TRY(copy_vec(w->in_b, H, enc->in_b));
  SE_TRY(copy_vec(w->in_ln_w, H, enc->in_ln_w));
  SE_TRY(copy_vec(w->in_ln_b, H, enc->in_ln_b));
  for (int i = 0; i < L; ++i) {
    se_encoder::Layer& y = enc->layers[i];
    SE_TRY(cast_mat(w->q_w[i], H, H, H, y.qkv_w));
    SE_TRY(cast_mat(w->k_w[i], H, H, H, y.qkv_w + (size_t)H * H));
    SE_TRY(cast_mat(w->v_w[i], H, H, H, y.qkv_w + (size_t)2 * H * H));
    SE_TRY(copy_vec(w->q_b[i], H, y.qkv_b));
    SE_TRY(copy_vec(w->k_b[i], H, y.qkv_b + H));
    SE_TRY(copy_vec(w->v_b[i], H, y.qkv_b + 2 * H));
    SE_TRY(cast_mat(w->ao_w[i], H, H, H, y.ao_w));
    SE_TRY(copy_vec(w->ao_b[i], H, y.ao_b));
    SE_TRY(copy_vec(w->aln_w[i], H, y.aln_w));
    SE_TRY(copy_vec(w->aln_b[i], H, y.aln_b));
    SE_TRY(cast_mat(w->ff1_w[i], I, H, H, y.ff1_w));
    SE_TRY(copy_vec(w->ff1_b[i], I, y.ff1_b));
    SE_TRY(cast_mat(w->ff2_w[i], H, I, I, y.ff2_w));
    SE_TRY(copy_vec(w->ff2_b[i], H, y.ff2_b));
    SE_TRY(copy_vec(w->oln_w[i], H, y.oln_w));
    SE_TRY(copy_vec(w->oln_b[i], H, y.oln_b));
  }
  SE_TRY(cast.flush());
  SE_TRY(copy.flush());
  for (int i = 0; i < L; ++i) {          // the inference copies of the QKV projections (pre-scaled queries) follow their masters
    se_encoder::Layer& y = enc->layers[i];
    SE_TRY(se::launch_qkv_inf(w->q_w[i], w->q_b[i], y.qkv_w, y.qkv_b, H, y.qkv_w_inf, y.qkv_b_inf, st));
  }
  return SE_OK;
}

extern "C" size_t se_encoder_saved_bytes(const se_encoder* enc, int B, int T) {
  if (!enc || B <= 0 || T <= 0) return 0;
  return carve_saved(enc, B, T, nullptr).total + 256;
}

extern "C" size_t se_encoder_train_workspace_bytes(const se_encoder* enc, int B, int T) {
  if (!enc || B <= 0 || T <= 0) return 0;
  return std::max(carve_ws(enc, B, T, nullptr).total, 2 * al256((size_t)B * T * enc->cfg.hidden * 4)) + 256;
}

static int check_train_shape(const se_encoder* enc, int B, int T, const char* who) {
  SE_REQUIRE(B > 0 && B <= 65535 && T > 0 && T <= se::kMaxPos, "%s: bad shape B=%d T=%d (T <= %d)", who, B, T, se::kMaxPos);
  SE_REQUIRE((size_t)B * T <= 0x7fffffff / 4, "%s: B*T too large", who);
  if ((enc->cfg.hidden != 768 && enc->cfg.hidden != 256) || enc->cfg.intermediate % 64 != 0) {
    se::set_error("%s: the training kernels are built for hidden_size 768 (and 256 for tests), got %d", who, enc->cfg.hidden);
    return SE_ERR_UNSUPPORTED;
  }
  return SE_OK;
}

extern "C" int se_encoder_fwd_train_bf16(const se_encoder* enc, const float* feats, const int32_t* lengths, int B, int T, float* hidden,
                                         void* saved, size_t saved_bytes, void* workspace, size_t workspace_bytes, float dropout_p,
                                         uint64_t seed, void* stream) {
  SE_REQUIRE(enc && feats && hidden && saved && workspace, "se_encoder_fwd_train_bf16: null argument");
  SE_TRY(check_train_shape(enc, B, T, "se_encoder_fwd_train_bf16"));
  SE_REQUIRE(saved_bytes >= se_encoder_saved_bytes(enc, B, T), "se_encoder_fwd_train_bf16: saved buffer too small");
  SE_REQUIRE(workspace_bytes >= 2 * al256((size_t)B * T * enc->cfg.hidden * 4), "se_encoder_fwd_train_bf16: workspace too small");
  SE_REQUIRE(dropout_p >= 0.f && dropout_p < 1.f, "se_encoder_fwd_train_bf16: dropout_p must be in [0, 1)");
  SE_REQUIRE(((uintptr_t)saved % 256 == 0) && ((uintptr_t)workspace % 256 == 0) && ((uintptr_t)hidden % 16 == 0),
             "se_encoder_fwd_train_bf16: buffers must be 256-B aligned");
  const int H = enc->cfg.hidden, I = enc->cfg.intermediate, D = enc->cfg.input_dim, L = enc->cfg.layers;
  const size_t Mz = (size_t)B * T;
  const int M = (int)Mz;
  hipStream_t st = se::as_stream(stream);
  Saved s = carve_saved(enc, B, T, reinterpret_cast<char*>(saved));
  float* x_f32 = reinterpret_cast<float*>(workspace);          // running residual stream (LayerNorm outputs)
  float* tmp = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + al256(Mz * H * 4));   // projection output before dropout
  const float eps = enc->cfg.ln_eps;
  const se::DropoutCfg dr = se::make_dropout(dropout_p, seed);
  const bool drop = dr.thr16 != 0;
  SE_REQUIRE(!drop || Mz * (size_t)H / 2 < 4294967296ull, "se_encoder_fwd_train_bf16: dropout pair index exceeds 32 bits");
  // B1
  SE_TRY(se::launch_cast_pad(feats, Mz, D, se::kInPad, s.xin, st));
  SE_TRY(se_gemm_bf16(s.xin, se::kInPad, enc->in_w, se::kInPad, enc->in_b, nullptr, M, H, se::kInPad, SE_ACT_IDENTITY, nullptr, s.pre0, H, stream));
  if (drop)     // dropout after the input LayerNorm
    SE_TRY(launch_ln_train(s.pre0, enc->pe, T, nullptr, enc->in_ln_w, enc->in_ln_b, M, H, eps, nullptr, x_f32, s.l[0].x0_bf, 0,
                           se::dropout_key(seed, se::dropout_site(L, 0)), dr, st));
  else
    SE_TRY(se::launch_layernorm(s.pre0, enc->pe, T, enc->in_ln_w, enc->in_ln_b, M, H, eps, x_f32, s.l[0].x0_bf, st));
  for (int i = 0; i < L; ++i) {
    const se_encoder::Layer& y = enc->layers[i];
    SavedLayer& a = s.l[i];
    const bool last = i == L - 1;
    // B2
    SE_TRY(se_gemm_bf16(a.x0_bf, H, y.qkv_w, H, y.qkv_b, nullptr, M, 3 * H, H, SE_ACT_IDENTITY, a.qkv, nullptr, 3 * H, stream));
    SE_TRY(se_mhsa_fwd_lse_bf16(a.qkv, lengths, B, T, enc->cfg.heads, a.ctx, a.lse, dropout_p, seed, se::dropout_site(i, 0), stream));
    if (drop) {   // projection -> dropout -> + residual -> LayerNorm: the dropout sits between the GEMM and the residual add
      SE_TRY(se_gemm_bf16(a.ctx, H, y.ao_w, H, y.ao_b, nullptr, M, H, H, SE_ACT_IDENTITY, nullptr, tmp, H, stream));
      SE_TRY(launch_ln_train(tmp, nullptr, 1, x_f32, y.aln_w, y.aln_b, M, H, eps, a.pre1, x_f32, a.x1_bf, se::dropout_key(seed, se::dropout_site(i, 1)),
                             0, dr, st));
    } else {
      SE_TRY(se_gemm_bf16(a.ctx, H, y.ao_w, H, y.ao_b, x_f32, M, H, H, SE_ACT_IDENTITY, nullptr, a.pre1, H, stream));
      SE_TRY(se::launch_layernorm(a.pre1, nullptr, 1, y.aln_w, y.aln_b, M, H, eps, x_f32, a.x1_bf, st));
    }
    // B3 (the pre-activation is kept for gelu')
    {
      // pre-activation (kept for the backward pass) and its GELU from ONE launch when the shape is the persistent kernel's (gemm6.hip)
      const int rc_dual = se_gemm6_dual_gelu_launch(a.x1_bf, H, y.ff1_w, H, y.ff1_b, M, I, H, a.hpre, a.h, I, stream);
      if (rc_dual < 0) return rc_dual;
      if (rc_dual > 0) {
        SE_TRY(se_gemm_bf16(a.x1_bf, H, y.ff1_w, H, y.ff1_b, nullptr, M, I, H, SE_ACT_IDENTITY, a.hpre, nullptr, I, stream));
        SE_TRY(se_gelu_bf16(a.hpre, Mz * I, a.h, stream));
      }
    }
    if (drop) {
      SE_TRY(se_gemm_bf16(a.h, I, y.ff2_w, I, y.ff2_b, nullptr, M, H, I, SE_ACT_IDENTITY, nullptr, tmp, H, stream));
      SE_TRY(launch_ln_train(tmp, nullptr, 1, x_f32, y.oln_w, y.oln_b, M, H, eps, a.pre2, last ? hidden : x_f32, last ? nullptr : s.l[i + 1].x0_bf,
                             se::dropout_key(seed, se::dropout_site(i, 2)), 0, dr, st));
    } else {
      SE_TRY(se_gemm_bf16(a.h, I, y.ff2_w, I, y.ff2_b, x_f32, M, H, I, SE_ACT_IDENTITY, nullptr, a.pre2, H, stream));
      SE_TRY(se::launch_layernorm(a.pre2, nullptr, 1, y.oln_w, y.oln_b, M, H, eps, last ? hidden : x_f32, last ? nullptr : s.l[i + 1].x0_bf, st));
    }
  }
  return SE_OK;
}

namespace {

// dW (N, K) = dY^T X from row-major bf16 dY (M, N) [ld ldy] and X (M, K) [ld ldx]; bias gradient by the caller.
// Split count: one 256 x 256 output tile per workgroup and one workgroup per CU (128 KiB of LDS), so the m range is split until
// tiles x splits just fills the 256 CUs; every split costs one fp32 slab (N K 4 bytes written + read back), hence the cap.
int weight_grad(const uint16_t* dY, int ldy, const uint16_t* X, int ldx, int M, int N, int K, float* dW, const TrainWs& w, void* stream) {
  const int tiles = ((N + 255) / 256) * ((K + 255) / 256);
  int splits = std::max(1, std::min(kSplits, 256 / tiles));
  while (splits > 1 && (size_t)splits * 64 > (size_t)M) --splits;
  return se_wgrad_tn_bf16(dY, ldy, X, ldx, M, N, K, splits, dW, 0, w.partials, (size_t)splits * N * K * sizeof(float), stream);
}

// dX (M, K) = dY (M, N) . W (N, K) [+ residual]   through the forward GEMM on W^T (K, N)
int input_grad(const uint16_t* dY, const uint16_t* W, int M, int N, int K, const float* residual, uint16_t* out_bf16, float* out_f32,
               const TrainWs& w, void* stream) {
  SE_TRY(se_transpose_bf16(W, N, K, K, w.wt, N, stream));
  return se_gemm_bf16(dY, N, w.wt, N, nullptr, residual, M, K, N, SE_ACT_IDENTITY, out_bf16, out_f32, K, stream);
}

}  // namespace

extern "C" int se_encoder_bwd_bf16(const se_encoder* enc, const int32_t* lengths, int B, int T, const float* d_hidden, const void* saved,
                                   size_t saved_bytes, const se_encoder_grads* g, void* workspace, size_t workspace_bytes, float dropout_p,
                                   uint64_t seed, void* stream) {
  return se_encoder_bwd_cb_bf16(enc, lengths, B, T, d_hidden, saved, saved_bytes, g, workspace, workspace_bytes, dropout_p, seed, nullptr, nullptr, stream);
}

// The same backward with a HOST callback after each layer's launches have been enqueued (layer L-1 first; -1 = the input stage): every
// parameter gradient of that layer is then ordered on `stream`, so a data-parallel caller can start that layer's gradient all-reduce
// (RCCL picks up the stream order) while the remaining layers' backward kernels run -- bucketed overlap instead of one collective at the end.
extern "C" int se_encoder_bwd_cb_bf16(const se_encoder* enc, const int32_t* lengths, int B, int T, const float* d_hidden, const void* saved,
                                      size_t saved_bytes, const se_encoder_grads* g, void* workspace, size_t workspace_bytes, float dropout_p,
                                      uint64_t seed, void (*layer_done)(int layer, void* user), void* user, void* stream) {
  SE_REQUIRE(enc && d_hidden && saved && g && workspace, "se_encoder_bwd_bf16: null argument");
  SE_TRY(check_train_shape(enc, B, T, "se_encoder_bwd_bf16"));
  SE_REQUIRE(saved_bytes >= se_encoder_saved_bytes(enc, B, T), "se_encoder_bwd_bf16: saved buffer too small");
  SE_REQUIRE(workspace_bytes >= se_encoder_train_workspace_bytes(enc, B, T), "se_encoder_bwd_bf16: workspace too small");
  SE_REQUIRE(((uintptr_t)saved % 256 == 0) && ((uintptr_t)workspace % 256 == 0), "se_encoder_bwd_bf16: buffers must be 256-B aligned");
  const int H = enc->cfg.hidden, I = enc->cfg.intermediate, D = enc->cfg.input_dim, L = enc->cfg.layers;
  const size_t Mz = (size_t)B * T;
  const int M = (int)Mz;
  hipStream_t st = se::as_stream(stream);
  const Saved s = carve_saved(enc, B, T, reinterpret_cast<char*>(const_cast<void*>(saved)));
  const TrainWs w = carve_ws(enc, B, T, reinterpret_cast<char*>(workspace));
  const float eps = enc->cfg.ln_eps;
  const se::DropoutCfg dr = se::make_dropout(dropout_p, seed);      // must be the forward's (p, seed): the masks are regenerated
  const bool drop = dr.thr16 != 0;
  auto dkey = [&](int layer, int which) { return drop ? se::dropout_key(seed, se::dropout_site(layer, which)) : 0u; };
  const float* gy = d_hidden;            // gradient wrt the current layer's output
  for (int i = L - 1; i >= 0; --i) {
    const se_encoder::Layer& y = enc->layers[i];
    const SavedLayer& a = s.l[i];
    // ---- output LayerNorm:  x2 = LN(pre2)
    //      (its dx column sums are the bias gradient of the linear that produced pre2)
    SE_TRY(se::launch_layernorm_bwd(a.pre2, nullptr, 1, gy, y.oln_w, M, H, eps, 0, w.fa, w.b1, g->oln_w[i], g->oln_b[i], g->ff2_b[i], 0, st,
                                    0, dkey(i, 2), dr.thr16, dr.scale));
    // ---- FFN output linear:  pre2 = h W2^T + b2 + x1
    SE_TRY(weight_grad(w.b1, H, a.h, I, M, H, I, g->ff2_w[i], w, stream));
    SE_TRY(input_grad(w.b1, y.ff2_w, M, H, I, nullptr, w.bi, nullptr, w, stream));                 // dh (M, I)
    SE_TRY(se::launch_gelu_bwd_colsum(w.bi, a.hpre, w.bi, M, I, g->ff1_b[i], st));                 // dhpre, and its column sums
    // ---- FFN input linear:  hpre = x1 W1^T + b1
    SE_TRY(weight_grad(w.bi, I, a.x1_bf, H, M, I, H, g->ff1_w[i], w, stream));
    SE_TRY(input_grad(w.bi, y.ff1_w, M, I, H, w.fa, nullptr, w.fb, w, stream));                    // dx1 = dhpre W1 + dpre2
    // ---- attention-output LayerNorm:  x1 = LN(pre1)
    SE_TRY(se::launch_layernorm_bwd(a.pre1, nullptr, 1, w.fb, y.aln_w, M, H, eps, 0, w.fa, w.b1, g->aln_w[i], g->aln_b[i], g->ao_b[i], 0, st,
                                    0, dkey(i, 1), dr.thr16, dr.scale));
    // ---- attention output linear:  pre1 = ctx Wo^T + bo + x0
    SE_TRY(weight_grad(w.b1, H, a.ctx, H, M, H, H, g->ao_w[i], w, stream));
    SE_TRY(input_grad(w.b1, y.ao_w, M, H, H, nullptr, w.b2, nullptr, w, stream));                  // dctx (M, H)
    // ---- attention core
    SE_TRY(se_mhsa_bwd_bf16(a.qkv, a.ctx, w.b2, a.lse, lengths, B, T, enc->cfg.heads, w.b3, w.dvec, dropout_p, seed, se::dropout_site(i, 0), stream));
    // ---- fused QKV linear:  qkv = x0 Wqkv^T + b
    SE_TRY(weight_grad(w.b3, 3 * H, a.x0_bf, H, M, 3 * H, H, w.gfused, w, stream));
    SE_TRY(se::launch_colsum_bf16(w.b3, M, 3 * H, 3 * H, w.bfused, st));
    {
      float* wdst[3] = {g->q_w[i], g->k_w[i], g->v_w[i]};
      float* bdst[3] = {g->q_b[i], g->k_b[i], g->v_b[i]};
      se::MultiCopy split(false, st);            // one launch instead of six copies
      for (int p = 0; p < 3; ++p) {
        SE_TRY(split.add(wdst[p], w.gfused + (size_t)p * H * H, (size_t)H * H));
        SE_TRY(split.add(bdst[p], w.bfused + (size_t)p * H, (size_t)H));
      }
      SE_TRY(split.flush());
    }
    SE_TRY(input_grad(w.b3, y.qkv_w, M, 3 * H, H, w.fa, nullptr, w.fb, w, stream));                // dx0 = dqkv Wqkv + dpre1
    gy = w.fb;
    if (layer_done) layer_done(i, user);
  }
  // ---- input stage:  x = LN(xin Win^T + b + PE)
  SE_TRY(se::launch_layernorm_bwd(s.pre0, enc->pe, T, gy, enc->in_ln_w, M, H, eps, 0, w.fa, w.b1, g->in_ln_w, g->in_ln_b, g->in_b, 0, st,
                                  dkey(L, 0), 0, dr.thr16, dr.scale));
  SE_TRY(weight_grad(w.b1, H, s.xin, se::kInPad, M, H, se::kInPad, w.gfused, w, stream));
  SE_HIP(hipMemcpy2DAsync(g->in_w, (size_t)D * 4, w.gfused, (size_t)se::kInPad * 4, (size_t)D * 4, H, hipMemcpyDeviceToDevice, st));
  if (layer_done) layer_done(-1, user);
  return SE_OK;
}
