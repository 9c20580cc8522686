// encoder_impl.h -- internals shared by encoder.hip (inference path), encoder_train.hip (training path) and bwd.hip.
#pragma once
#include <vector>
#include "common.h"

namespace se {
constexpr int kMaxPos = 5008;     // 50 s of 10 ms frames (MAX_POSITIONS_LEN = 16000*50 samples, runner.py:32)
constexpr int kInPad = 128;       // input feature dim padded to a multiple of the GEMM K tile

// host launchers of kernels that live in encoder.hip / bwd.hip
// gemm4.hip: x = LayerNorm(gelu(A . W^T + bias)), N = 768 only
int launch_gemm_pos_ln(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const float* table, int T, const float* ln_w,
                       const float* ln_b, float eps, int M, int N, int K, float* out_f32, uint16_t* out_bf16, uint8_t* out_lo, hipStream_t st);
// the same kernel on the 24-bit residual stream (bf16 hi = the x_bf16 tensor + int8 lo, tile-major)
int launch_gemm_res24_ln(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const uint16_t* res_hi, const uint8_t* res_lo,
                         const float* ln_w, const float* ln_b, float eps, int M, int N, int K, float* out_f32, uint16_t* out_bf16, uint8_t* out_lo,
                         hipStream_t st);
size_t gemm4_lo_bytes(int M);
// gemm8.hip: the K = 3072 projection on 256 x 384 tiles with the LayerNorm statistics exchanged between the two column halves; returns 1 when
// the call is not for it (the caller then uses launch_gemm_res24_ln).  `scratch`: gemm8_scratch_bytes() bytes whose trailing flag words are zero
#ifdef SE_AMD_EXPERIMENTS
size_t gemm8_scratch_bytes();
int launch_gemm8_res24_ln(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const uint16_t* res_hi, const uint8_t* res_lo,
                          const float* ln_w, const float* ln_b, float eps, int M, int N, int K, float* out_f32, uint16_t* out_bf16, uint8_t* out_lo,
                          void* scratch, hipStream_t st, int force = 0);
#else
inline size_t gemm8_scratch_bytes() { return 256; }      // the pair-exchange kernel (tools/experiments/kernels/gemm8.hip) is not part of the product library
#endif
int launch_gemm_gelu_ln(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const float* ln_w, const float* ln_b, float eps,
                        int M, int N, int K, float* out_f32, uint16_t* out_bf16, hipStream_t st);
int launch_layernorm(const float* x, const float* pe, int T, const float* w, const float* b, int M, int H, float eps,
                     float* out_f32, uint16_t* out_bf16, hipStream_t st);
int launch_cast_pad(const float* x, size_t rows, int cols, int ld_out, uint16_t* out, hipStream_t st);
int launch_layernorm_bwd(const float* x_in, const float* pe, int T, const float* dy, const float* w, int M, int H, float eps, int gelu_in,
                         float* dx, uint16_t* dx_bf16, float* dgamma, float* dbeta, float* dbias, int accumulate, hipStream_t st,
                         uint32_t key_dy = 0, uint32_t key_dx = 0, uint32_t thr16 = 0, float dscale = 1.f, int group_rows = 0);
int launch_gelu_bwd_colsum(const uint16_t* dy, const uint16_t* pre, uint16_t* dx, int rows, int cols, float* colsum, hipStream_t st);
int launch_colsum_bf16(const uint16_t* x, int rows, int cols, int ld, float* out, hipStream_t st);
inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
constexpr float kQScale = 0.125f * 1.44269504088896340736f;      // log2(e) / sqrt(64), folded into the inference copy of W_q, b_q
// rebuilds a layer's inference QKV copy from device fp32 query masters + the layer's bf16 K / V rows (encoder.hip)
int launch_qkv_inf(const float* q_w, const float* q_b, const uint16_t* qkv_w, const float* qkv_b, int H, uint16_t* qkv_w_inf, float* qkv_b_inf, hipStream_t st);
}  // namespace se

struct se_encoder {
  se_encoder_config cfg;
  void* blob;          // one device allocation
  size_t blob_bytes;
  // device views
  uint16_t* in_w;      // (H, kInPad) bf16
  float *in_b, *in_ln_w, *in_ln_b, *pe;
  struct Layer {
    uint16_t *qkv_w, *ao_w, *ff1_w, *ff2_w;
    // inference copy of the fused QKV projection: query rows and query bias pre-multiplied by log2(e) / sqrt(64) BEFORE the bf16 rounding
    // (se_mhsa_fwd_prescaled_bf16); the unscaled pair above stays what the training forward / backward use
    uint16_t* qkv_w_inf;
    float* qkv_b_inf;
    float *qkv_b, *ao_b, *aln_w, *aln_b, *ff1_b, *ff2_b, *oln_w, *oln_b;
  };
  std::vector<Layer> layers;
  uint16_t *sh_dense_w, *sh_out_w;
  float *sh_dense_b, *sh_ln_w, *sh_ln_b, *sh_out_b;
};
