/*
 * se_amd.h -- C ABI of the MI355X (gfx950) speech-enhancement hot path.
 *
 * Drop-in boundary for the per-utterance path of leo19941227/Speech-Enhancement-by-S3PRL
 * (STFT -> features -> [TERA/Mockingjay encoder] -> mask / spectrogram head -> mask (.) |X|^2 ->
 * iSTFT -> level normalisation, plus the masked log-L1 objective).  The reference has no native code
 * and no FFI: its interface for this path is the Python duck-typed surface of SURVEY.md section 8(b)
 * (S3PRL's OnlinePreprocessor / TRANSFORMER / TransformerSpecPredictionHead and the reference's
 * model.py / objective.py / utils.py / runner.py call sites).  Each entry point below cites the
 * reference call it replaces; the Python host (the .py files of speech-enhancement-by-s3prl_amd) mirrors those
 * classes and binds these symbols through ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain C: pointers + sizes only, no C++ / torch types.  Every data pointer is DEVICE memory
 *     owned by the caller (torch.Tensor.data_ptr() of a contiguous tensor), fp32 unless stated.
 *   - every launch takes `stream` = a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     calls are asynchronous and never synchronise.  No allocation happens inside launch calls;
 *     workspaces are caller-owned and sized by the *_workspace_bytes() queries.
 *   - return 0 on success, negative se_status on error; se_last_error() gives a thread-local message.
 *     Allocation failures carry the substring "CUDA out of memory" so the reference's skip-batch
 *     handler (runner.py:505,606) keeps working.
 *   - plans / encoders are immutable after creation and may be shared across streams.
 *     HIP is initialised lazily (first plan creation / launch), never at library load, so a spawned
 *     child may set HIP_VISIBLE_DEVICES first (sampler.py:145-153).
 */
#ifndef SE_AMD_H_
#define SE_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum se_status {
  SE_OK = 0,
  SE_ERR_INVALID = -1,      /* bad argument / shape */
  SE_ERR_UNSUPPORTED = -2,  /* geometry the gfx950 kernels are not specialised for */
  SE_ERR_HIP = -3,          /* a HIP runtime call failed */
  SE_ERR_OOM = -4,          /* device allocation failed ("CUDA out of memory ...") */
  SE_ERR_NO_DEVICE = -5
} se_status;

/* activation ids shared by the head kernels (model.py:12,24: nn.<activation>()) */
typedef enum se_act { SE_ACT_IDENTITY = 0, SE_ACT_RELU = 1, SE_ACT_SIGMOID = 2, SE_ACT_GELU = 3, SE_ACT_EXP = 4 } se_act;

const char* se_last_error(void);
const char* se_version(void);
/* 1 if a gfx950 device is visible, 0 otherwise (never throws; initialises HIP). */
int se_device_available(void);

/* ------------------------------------------------------------------------------------------------
 * Preprocessor plan: STFT geometry + device tables (Hann window, twiddles, sparse HTK mel bank).
 * Replaces OnlinePreprocessor.__init__ (S3PRL utility/preprocessor.py; constructed at
 * run_downstream.py:159 from config/pretrain_sample.yaml:39-48).
 * The kernels are specialised for n_fft = 400 (n_freq = 201), hop = 160, win <= 400; anything else
 * returns SE_ERR_UNSUPPORTED.
 * ---------------------------------------------------------------------------------------------- */
typedef struct se_plan se_plan;

typedef struct se_geometry {
  int sample_rate; /* 16000 */
  int win;         /* 400 samples (win_ms 25) */
  int hop;         /* 160 samples (hop_ms 10) */
  int n_freq;      /* 201 */
  int n_mels;      /* 40 in the shipped configs; <= 128 (the MFCC branch builds its own 128-filter bank) */
} se_geometry;

int se_plan_create(const se_geometry* geom, se_plan** out);
void se_plan_destroy(se_plan* plan);
/* copies the plan's host-side tables out (for tests): window[n_fft = 400] (win centred), mel_fb[n_freq*n_mels] row-major (k, m). */
int se_plan_tables(const se_plan* plan, float* window, float* mel_fb);

/* Number of STFT frames of a T-sample signal: T / hop + 1 (runner.py:455). */
int se_num_frames(const se_plan* plan, int n_samples);

/*
 * se_stft_f32 -- rows A1 + A2 (+ A3): OnlinePreprocessor.forward's `_stft` + `_magphase` (+ `_melscale`)
 * for ONE channel of wavs (runner.py:433,558; sampler.py:60,226-228).
 *   wavs     (B, C, T) fp32, contiguous
 *   power    (B, F, K) fp32 time-major  re^2+im^2            (may be NULL)
 *   phase    (B, F, K) fp32 time-major  atan2(im, re)        (may be NULL)
 *   complx   (B, F, K, 2) fp32 time-major (re, im)           (may be NULL)
 *   mel      (B, n_mels, F) fp32 FEATURE-major raw mel power (may be NULL) -- input of se_features_f32
 * F = T/hop + 1, K = n_freq.  reflect padding of n_fft/2 on both sides, periodic Hann, one-sided.
 */
int se_stft_f32(const se_plan* plan, const float* wavs, int B, int C, int T, int channel,
                float* power, float* phase, float* complx, float* mel, void* stream);
/* The same transform for TWO channels of the batch in one launch (the reference transforms the noisy and the clean channel of every batch,
   runner.py:433,558: feat_list entries with channel_inp / channel_tar); each output set as above, any pointer may be NULL. */
int se_stft2_f32(const se_plan* plan, const float* wavs, int B, int C, int T, int channel_a, float* power_a, float* phase_a, float* complx_a,
                 float* mel_a, int channel_b, float* power_b, float* phase_b, float* complx_b, float* mel_b, void* stream);

/*
 * se_stft_tphase_f32 -- rows A1 + A2 (+ A3) in the form every consumer INSIDE the path uses (runner.py:433,558 -> model -> runner.py:267):
 * the noisy channel's phase is only ever fed back into OnlinePreprocessor.istft, so the transform writes it as one 32-bit word per bin from
 * which (cos, sin) follow without transcendental functions -- t = tan(half the angle of (|re|, im)) = im / (|X| + |re|) in [-1, 1] as fp32,
 * bit 0 of the pattern = (re < 0); X == 0 -> 0 (the reference's atan2(0, 0) = 0) -- and `phase` = atan2 stays an on-demand output of
 * se_stft_f32.  One launch for one or two channels (channel_b < 0: one); every sample is loaded once.
 *   power_*   (B, F, K) fp32 time-major  re^2+im^2                 (may be NULL)
 *   tphase_*  (B, F, K) 32-bit words, time-major                   (may be NULL)
 *   mel_*     (B, n_mels, F) fp32 feature-major raw mel power      (may be NULL)
 */
int se_stft_tphase_f32(const se_plan* plan, const float* wavs, int B, int C, int T, int channel_a, float* power_a, unsigned* tphase_a, float* mel_a,
                       int channel_b, float* power_b, unsigned* tphase_b, float* mel_b, void* stream);

/*
 * se_features_f32 -- row A4: OnlinePreprocessor.forward's select_feat: log(x+eps), `delta` stacked
 * compute_deltas passes (5-tap, replicate), CMVN over time (unbiased std, +eps), transposed to
 * time-major.
 *   raw       feature-major (B, D, F) if raw_time_major == 0, else time-major (B, F, D)
 *   out       (B, F, D*(1+delta)) fp32 time-major
 *   workspace >= se_features_workspace_bytes(B, D, F, delta) bytes of device memory
 */
size_t se_features_workspace_bytes(int B, int D, int F, int delta);
int se_features_f32(const float* raw, int raw_time_major, int B, int D, int F,
                    int apply_log, int delta, int cmvn, float eps,
                    float* out, void* workspace, size_t workspace_bytes, void* stream);
/* The same with two side outputs for the encoder that consumes the features next (both optional): the rows as bf16 zero-padded to ld_pad columns
 * (B*F, ld_pad) -- the operand of se_encoder_fwd2_bf16's input projection -- and valid_count (B) int32 = the number of frames whose feature sum is
 * not zero (what se_valid_lengths_i32 computes: S3PRL's length rule).  valid_count is cleared inside. */
int se_features2_f32(const float* raw, int raw_time_major, int B, int D, int F,
                     int apply_log, int delta, int cmvn, float eps,
                     float* out, void* workspace, size_t workspace_bytes,
                     uint16_t* out_bf16_pad, int ld_pad, int32_t* valid_count, void* stream);

/* The same result in ONE pass over the raw plane (no feature-major intermediate): a workgroup owns 32 frames and recomputes log / deltas from a
 * (32 + 4 delta)-frame window; the CMVN statistics (cmvn != 0) come from a statistics-only launch in front.  workspace (cmvn != 0 only)
 * >= se_features3_workspace_bytes(B, D, delta).  colstats_out (B, D*(1+delta), 2) optional, cmvn == 0 only: per (utterance, output column) the pair
 * (mean over time, 1 / (unbiased std + colstats_eps)) of the rows written to `out` -- what LinearResidual's own CMVN (model.py:29-31) needs, handed to
 * se_head_linear_pre_f32 so that the head does not read the features a second time for it.  Side outputs as se_features2_f32. */
size_t se_features3_workspace_bytes(int B, int D, int delta);
size_t se_features3_colstats_workspace_bytes(int B, int D, int F, int delta);      /* workspace when colstats_out != NULL (8-byte aligned) */
int se_features3_f32(const float* raw, int raw_time_major, int B, int D, int F, int apply_log, int delta, int cmvn, float eps,
                     float* out, void* workspace, size_t workspace_bytes, uint16_t* out_bf16_pad, int ld_pad, int32_t* valid_count,
                     float* colstats_out, float colstats_eps, void* stream);

/*
 * se_istft_f32 -- row A6: OnlinePreprocessor.istft(linears, phases) (runner.py:267): mag = power^(1/linear_power),
 * (mag cos phi, mag sin phi) -> inverse 400-pt real DFT -> x Hann -> overlap-add -> / sum(w^2) -> trim n_fft/2.
 *   power, phase (B, F, K) time-major;   wav_out (B, wav_stride) with the first hop*(F-1) samples written and
 *   samples [hop*(F-1), wav_stride) zero-filled (the right-pad of runner.py:268).
 *   sumsq_out (B) optional: sum over n < lengths[b] of wav^2 (feeds se_dbnorm_f32); zeroed inside this call
 *   (a memset on `stream`).
 *   lengths   (B) int64 device pointer, required iff sumsq_out != NULL.
 */
int se_istft_f32(const se_plan* plan, const float* power, const float* phase, int B, int F,
                 float linear_power, float* wav_out, int wav_stride,
                 const int64_t* lengths, float* sumsq_out, void* stream);

/*
 * se_istft_tphase_f32 -- row A6 on (enhanced power, encoded phase of the noisy channel as se_stft_tphase_f32 writes it):
 * X' = sqrt(power) * (+-(1 - t^2), 2 t) / (1 + t^2), then as se_istft_f32 (linear_power = 2).  log_input != 0: `power` holds log_predicted
 * of a log-target head (model.py:108-124, predicted = exp(log_predicted)).  Outputs, zero fill and sumsq_out as se_istft_f32.
 * ref / ref_stride / ref_sumsq_out (optional): the reference waveform of the level normalisation (runner.py:570 passes wav_tar): its masked
 * square sum over n < min(lengths[b], hop (F-1)) is accumulated by the same launch (what se_masked_sumsq_f32 computes on its own).
 */
int se_istft_tphase_f32(const se_plan* plan, const float* power, const unsigned* tphase, int B, int F, int log_input,
                        float* wav_out, int wav_stride, const int64_t* lengths, float* sumsq_out,
                        const float* ref, int ref_stride, float* ref_sumsq_out, void* stream);

/*
 * se_masked_sumsq_f32 -- utils.py:26-29 numerator: sums[b] = sum_{n < lengths[b]} x[b,n]^2  (sums zeroed inside).
 */
int se_masked_sumsq_f32(const float* x, int B, int T, int x_stride, const int64_t* lengths, float* sums, void* stream);

/*
 * se_dbnorm_f32 -- row D2: masked_normalize_decibel (utils.py:31-46) given the masked square sums:
 *   scale[b] = sqrt( 10^(target_db[b]/10) / (wav_sumsq[b]/(len[b]+eps) + eps) ),  wav *= scale   (in place)
 *   target: if ref_sumsq != NULL: target_db[b] = 10 log10( ref_sumsq[b] / (len[b]+eps) )   (reference-audio form)
 *           else: fixed_db (e.g. -25)
 */
int se_dbnorm_f32(float* wav, int B, int T, int wav_stride, const int64_t* lengths,
                  const float* wav_sumsq, const float* ref_sumsq, float fixed_db, float eps, void* stream);

/*
 * se_length_masks_i64 -- row D1: Runner._get_length_masks (runner.py:216-220): masks[b,t] = t < lengths[b] (int64).
 */
int se_length_masks_i64(const int64_t* lengths, int B, int max_len, int64_t* masks, void* stream);

/*
 * se_head_linear_f32 -- rows C1 / C2: LinearResidual.forward (model.py:28-34) and Linear.forward (model.py:14-17).
 *   feats (B, F, D) time-major; W (N, D) row-major (nn.Linear.weight); bias (N)
 *   cmvn != 0: features are normalised over time (dim=1), unbiased std, `+eps` outside the sqrt, first.
 *   offset    (B, F, N) = act(feats W^T + bias)                      (may be NULL)
 *   predicted (B, F, N) = linears * offset  if linears != NULL (C1)  else = offset (C2)
 *   workspace >= se_head_workspace_bytes(B, F, D, N)
 *   fp32 arithmetic: both operands as three-term bf16 splits (x = x1 + x2 + x3, residuals exact), the six products of weight >= 2^-16 on the
 *   bf16 matrix instruction with fp32 accumulation (every product exact; dropped terms <= 2^-24): an fp32 dot product in another summation order.
 */
size_t se_head_workspace_bytes(int B, int F, int D, int N);
int se_head_linear_f32(const float* feats, const float* W, const float* bias, const float* linears,
                       int B, int F, int D, int N, int act, int cmvn, float eps,
                       float* predicted, float* offset, void* workspace, size_t workspace_bytes, void* stream);

/* The evaluate()-style pass (runner.py:556-575) calls the head once per batch with unchanged weights: se_head_split_weights_f32 writes the
 * three-term bf16 split se_head_linear_f32 builds internally on every call (W3: se_head_w3_bytes(N, D) bytes) once, and se_head_linear_pre_f32
 * is se_head_linear_f32 on that split and on ready-made column statistics `stats` (B, D, 2) = (mean, 1 / (unbiased std + eps)) per (utterance,
 * feature column) -- se_features3_f32's colstats_out -- or NULL for no CMVN.  Same kernel, same results. */
size_t se_head_w3_bytes(int N, int D);
/* stats (B, D, 2) = (mean over time, 1 / (unbiased std + eps)) of feats (B, F, D): the CMVN of model.py:29-31 as se_head_linear_pre_f32 takes it */
int se_head_colstats_f32(const float* feats, int B, int F, int D, float eps, float* stats, void* stream);
int se_head_split_weights_f32(const float* W, int N, int D, uint16_t* W3, void* stream);
int se_head_linear_pre_f32(const float* feats, const uint16_t* W3, const float* bias, const float* linears, const float* stats,
                           int B, int F, int D, int N, int act, float* predicted, float* offset, void* stream);

/* The evaluate()-style pass of a mask head scored by objective.SISDR (runner.py:556-575 with vcb.yaml / pseudo_noise.yaml): se_head_linear_pre_f32
 * AND se_sisdr_spec_loss_f32 of its `predicted` against linear_tar -- where the kernel allows (201 bins, Sigmoid, F >= 128, 16-B aligned planes) the
 * criterion's sums are taken from the products on their way out of the head's registers; otherwise the two entry points run one after the other.
 * lengths / len_div / loss_b / sums_out / loss_out as se_sisdr_spec_loss_f32; scratch: se_head_sisdr_scratch_doubles(B, F, N) doubles. */
size_t se_head_sisdr_scratch_doubles(int B, int F, int N);
int se_head_linear_sisdr_f32(const float* feats, const uint16_t* W3, const float* bias, const float* linears, const float* stats,
                             int B, int F, int D, int N, int act, float* predicted, float* offset,
                             const float* linear_tar, const int64_t* lengths, int len_div, float eps, double* scratch,
                             float* loss_b, double* sums_out, float* loss_out, void* stream);
/* the fold of that fused form's slab (two slots of three doubles per tile_rows-row workgroup): loss_b, {sum, B}, mean */
int se_sisdr_head_mean_f32(const double* slab, int B, int F, int tile_rows, float eps, float* loss_b, double* sums_out, float* loss_out, void* stream);

/*
 * se_head_linear_bwd_f32 -- autograd of C1/C2 wrt the head parameters (runner.py:459 loss.backward()):
 *   inputs: grad_predicted (B,F,N) = d loss / d predicted and / or grad_offset (B,F,N) = d loss / d offset (either may be NULL:
 *   L1 / SISDR score `predicted`, WSD scores the mask `offset`), linears (or NULL), offset (B,F,N);
 *   g_pre = (grad_predicted (.) linears + grad_offset) (.) act'(offset)
 *   computes gW (N, D) and gb (N) (accumulated over all frames of all utterances; zeroed inside).
 *   Needs the same feats / cmvn / eps as the forward (the normalised features are recomputed).
 */
int se_head_linear_bwd_f32(const float* feats, const float* linears, const float* offset, const float* grad_predicted,
                           const float* grad_offset, int B, int F, int D, int N, int act, int cmvn, float eps,
                           float* gW, float* gb, void* workspace, size_t workspace_bytes, void* stream);
/* gradient wrt the FEATURES of the same head (needed when it sits on a trainable stack: the Residual head's LSTM, model.py:62-91):
 *   dx (B,F,D) = CMVN'( g_pre . W ), g_pre as above; the product runs on the bf16 GEMM.  workspace: se_head_dx_workspace_bytes. */
size_t se_head_dx_workspace_bytes(int B, int F, int D, int N);
int se_head_linear_dx_f32(const float* feats, const float* linears, const float* offset, const float* grad_predicted,
                          const float* grad_offset, const float* W, int B, int F, int D, int N, int act, int cmvn, float eps,
                          float* dx, void* workspace, size_t workspace_bytes, void* stream);

/*
 * se_l1_masked_f32 -- row E1: L1.forward (objective.py:103-117) in un-normalised form:
 *   sums[0] = sum over valid frames (f < frame_lengths[b]) and bins of |log_pred - log(linear_tar + eps)|
 *   sums[1] = number of such elements.        loss = sums[0] / sums[1]  (global mean over the batch;
 *   under data parallelism both are all-reduced first).  sums is a DEVICE double[2], zeroed inside.
 *   grad (optional, may be NULL): sign(log_pred - log(tar+eps)) on valid elements, 0 elsewhere -- the caller
 *   scales by 1/sums[1] (global count).
 */
int se_l1_masked_f32(const float* log_pred, const float* linear_tar, const int64_t* frame_lengths,
                     int B, int F, int K, float eps, double* sums, float* grad, void* stream);
/* The same criterion as ONE launch with nothing in front or behind (objective.py:103-117 is one call): `lengths` are frame counts (len_div == 0) or
 * WAVEFORM lengths with frames = lengths / len_div + 1 (runner.py:455); `scratch3` is a persistent device buffer of se_l1_scratch_doubles(B) doubles
 * whose first word (the arrival ticket) is zero on entry and left zero on exit (the last workgroup to arrive folds the per-workgroup partial sums,
 * publishes and clears); sums_out double[2] = {sum, count}, loss_out float[1] = sum / count. */
size_t se_l1_scratch_doubles(int B);
int se_l1_masked_loss_f32(const float* log_pred, const float* linear_tar, const int64_t* lengths, int len_div, int B, int F, int K, float eps,
                          double* scratch3, double* sums_out, float* loss_out, float* grad, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Encoder: S3PRL TRANSFORMER (rows B1-B3) + TransformerSpecPredictionHead (row B4), bf16 MFMA.
 * Replaces TRANSFORMER.forward (model.py:164; runner.py:275,282) and
 * TransformerSpecPredictionHead.forward (model.py:120,165).
 * ---------------------------------------------------------------------------------------------- */
typedef struct se_encoder se_encoder;

typedef struct se_encoder_config {
  int input_dim;     /* D: 80 for mel/log/delta1 (pretrain_sample.yaml:54-59) */
  int hidden;        /* 768  */
  int layers;        /* 6 (sample config) / 3 (base) */
  int heads;         /* 12, head dim must be 64 */
  int intermediate;  /* 3072 */
  float ln_eps;      /* 1e-12 */
  int spec_out;      /* 0 = no spec head; else output dim of the head (201) */
  int fused_ln_min_rows; /* rows (B*T) from which the out-proj / FFN2 projections use the row-complete GEMM + LayerNorm kernel;
                            0 = default (16000: from about half a round of workgroups on the row-complete kernels win, re-measured in round 2).  Callers that keep two half
                            batches in flight on two streams set it lower: two half-size launches share the chip out of phase. */
} se_encoder_config;

/* HOST pointers to fp32 weights, nn.Linear layout (out, in); arrays of `layers` pointers per-layer. */
typedef struct se_encoder_weights {
  const float *in_w, *in_b, *in_ln_w, *in_ln_b;
  const float* const *q_w, * const *q_b, * const *k_w, * const *k_b, * const *v_w, * const *v_b;
  const float* const *ao_w, * const *ao_b, * const *aln_w, * const *aln_b;
  const float* const *ff1_w, * const *ff1_b, * const *ff2_w, * const *ff2_b, * const *oln_w, * const *oln_b;
  const float *sh_dense_w, *sh_dense_b, *sh_ln_w, *sh_ln_b, *sh_out_w, *sh_out_b; /* spec head (may be NULL) */
} se_encoder_weights;

int se_encoder_create(const se_encoder_config* cfg, const se_encoder_weights* w, se_encoder** out);
void se_encoder_destroy(se_encoder* enc);
size_t se_encoder_workspace_bytes(const se_encoder* enc, int B, int T);

/*
 * se_encoder_fwd_bf16 -- feats (B, T, D) fp32 -> hidden (B, T, H) fp32 (last layer, select_layer -1, eval mode).
 *   lengths (B) int32 device pointer: valid frames per utterance (keys >= length are masked, the
 *   (1-mask)*-10000 additive mask of S3PRL); NULL = all T valid.
 *   GEMM operands bf16, fp32 accumulate, fp32 residual stream, fp32 LayerNorm / softmax.
 */
int se_encoder_fwd_bf16(const se_encoder* enc, const float* feats, const int32_t* lengths, int B, int T,
                        float* hidden, void* workspace, size_t workspace_bytes, void* stream);
/* The same with the input features ALSO available as bf16 rows zero-padded to 128 columns (se_features2_f32's side output): the fp32 -> bf16
 * conversion pass in front of the input projection is skipped.  feats may then be NULL. */
int se_encoder_fwd2_bf16(const se_encoder* enc, const float* feats, const uint16_t* feats_bf16_pad, const int32_t* lengths, int B, int T,
                         float* hidden, void* workspace, size_t workspace_bytes, void* stream);

/*
 * se_spechead_fwd_bf16 -- rows B4 + C3: dense -> gelu -> LayerNorm -> output linear, then SpecHead.forward's
 * epilogue (model.py:119-126):  log_target != 0: predicted = act(exp(p)), log_predicted = p
 *                               else:            predicted = act(p), log_predicted = log(p + eps)
 *   hidden (B, T, H) fp32;  predicted, log_predicted (B, T, spec_out) fp32 (either may be NULL);
 *   raw (B, T, spec_out) optional: the un-activated linear output p.
 */
int se_spechead_fwd_bf16(const se_encoder* enc, const float* hidden, int B, int T, int log_target, int act, float eps,
                         float* predicted, float* log_predicted, float* raw,
                         void* workspace, size_t workspace_bytes, void* stream);
/* The same with the caller's word that `hidden` is exactly what the last se_encoder_fwd_bf16 on this `workspace` returned (same B, T, untouched since):
 * that call's final launch left the bf16 copy of the hidden states in the workspace, so the fp32 -> bf16 conversion pass is skipped (x_bf_valid != 0). */
int se_spechead_fwd2_bf16(const se_encoder* enc, const float* hidden, int B, int T, int log_target, int act, float eps,
                          float* predicted, float* log_predicted, float* raw,
                          void* workspace, size_t workspace_bytes, int x_bf_valid, void* stream);

/* frame validity: lengths[b] = #frames whose feature sum != 0 (S3PRL process_input_data). feats (B,T,D). */
int se_valid_lengths_i32(const float* feats, int B, int T, int D, int32_t* lengths, void* stream);

/* Building blocks, exported for tests / roofline measurement. All bf16 = uint16_t device arrays. */
/* C[M,N] = epilogue(A[M,K] . W[N,K]^T + bias);  A, W bf16 row-major (K contiguous); lda/ldw/ldc in elements.
 *   out_bf16 / out_f32 (either may be NULL); residual_f32 (M,N) optional, added before the store; act in se_act. */
int se_gemm_bf16(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias,
                 const float* residual_f32, int M, int N, int K, int act,
                 uint16_t* out_bf16, float* out_f32, int ldc, void* stream);
/* Row-complete fused form for N = 768 (attention-output and FFN2 projections, rows B2 / B3):
 *   x = LayerNorm(A[M,K] . W[768,K]^T + bias + residual_f32) * ln_w + ln_b  ->  out_f32 and / or out_bf16 (M, 768)
 * One workgroup owns 128 complete rows, so the LayerNorm runs on the accumulators (no fp32 round trip, no second launch).
 * Returns SE_ERR_UNSUPPORTED unless N == 768, K % 32 == 0, K >= 128. */
int se_gemm_res_ln_bf16(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const float* residual_f32,
                        const float* ln_w, const float* ln_b, float eps, int M, int N, int K,
                        float* out_f32, uint16_t* out_bf16, void* stream);
/* qkv (B*T, 3H) bf16 = [Q | K | V] per row, heads of 64 columns; ctx (B*T, H) bf16. */
int se_mhsa_fwd_bf16(const uint16_t* qkv, const int32_t* lengths, int B, int T, int heads, uint16_t* ctx, void* stream);
/* The same attention on PRE-SCALED queries: the Q block of `qkv` already carries log2(e) / sqrt(64) (the encoder's inference copy of the
   fused QKV weights folds it into the query rows before their bf16 rounding), i.e. ctx = softmax_base2(Q' K^T + pad-mask) V: no
   per-score multiply in front of the exponential, and the running reference starts at 0 (csrc/mhsa.hip, PRE = 1).  Inference only. */
int se_mhsa_fwd_prescaled_bf16(const uint16_t* qkv, const int32_t* lengths, int B, int T, int heads, uint16_t* ctx, void* stream);
/* Test / measurement surface: the two kernels behind se_mhsa_fwd_prescaled_bf16 by number -- variant 0 = csrc/mhsa.hip (4-wave workgroups, the default
   below 768 workgroups), 10 = csrc/mhsa8.hip (8-wave workgroups on one LDS-DMA staged tile, the default from there on); bit-identical results.  Other
   numbers are the parked experiments of tools/experiments/kernels/ (developer builds only): SE_ERR_UNSUPPORTED in the product library. */
int se_mhsa_fwd_prescaled_variant_bf16(const uint16_t* qkv, const int32_t* lengths, int B, int T, int heads, uint16_t* ctx, int variant, void* stream);
/* Row-complete projection on the encoder's 24-bit residual stream (bf16 hi rows + int8 lo bytes, tile-major: csrc/gemm4.hip), exported for
 * tests / measurement: x = LayerNorm(A . W^T + bias + residual), out as fp32 rows or as (bf16, lo).  variant 0 = the encoder's dispatch,
 * 7 = 128 x 768 tiles (8 = the parked 256 x 384 pair-exchange experiment: SE_ERR_UNSUPPORTED in the product library).
 * scratch: se_gemm_res24_scratch_bytes() zeroed bytes; lo buffers: se_gemm_res24_lo_bytes(M) bytes. */
size_t se_gemm_res24_scratch_bytes(void);
size_t se_gemm_res24_lo_bytes(int M);
int se_gemm_res24_ln_bf16(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const uint16_t* res_hi, const uint8_t* res_lo,
                          const float* ln_w, const float* ln_b, float eps, int M, int N, int K, float* out_f32, uint16_t* out_bf16, uint8_t* out_lo,
                          int variant, void* scratch, void* stream);

/* y = LN(x) * w + b over the last dim H (TF style, eps inside sqrt); x fp32 (M,H); outputs fp32 and/or bf16. */
int se_layernorm_f32(const float* x, const float* w, const float* b, int M, int H, float eps,
                     float* out_f32, uint16_t* out_bf16, void* stream);
int se_cast_f32_bf16(const float* x, size_t n, uint16_t* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Exact-fp32 building blocks of the encoder's parity mode (TRANSFORMER / TransformerSpecPredictionHead with precision = 'fp32':
 * rows B1-B4 at north_star's 1e-4 tolerance against the reference's fp32 PyTorch path; the bf16 kernels above stay the bench default).
 * se_gemm_f32: C = act(alpha * A . W^T + bias) [+ residual[row % res_mod]] with fp32 operands on the fp32 matrix instruction (a k-ordered
 *   fmaf chain), batched over batch_outer x batch_inner problems with free element strides; W is (N, K) row-major (nn.Linear weight, the
 *   K operand of Q K^T) or, with w_kmajor = 1, (K, N) row-major (the V operand of P V).  lda / ldw / ldc in elements.
 * se_softmax_rows_f32: in-place row softmax of scores (B, heads, T, T) with the reference's additive -10000 on keys >= lengths[b]. */
int se_gemm_f32(const float* A, long lda, const float* W, long ldw, int w_kmajor, const float* bias, const float* residual, int res_mod,
                int M, int N, int K, int act, float alpha, float* C, long ldc, int batch_outer, int batch_inner, long strideA_outer,
                long strideA_inner, long strideW_outer, long strideW_inner, long strideC_outer, long strideC_inner, void* stream);
int se_softmax_rows_f32(float* scores, const int32_t* lengths, int B, int heads, int T, void* stream);
/* Three-term bf16 split of an fp32 matrix (rows, cols) [row stride ld] for the "bf16x3" parity mode: x = x1 + x2 + r, x1 = bf16(x),
 * x2 = bf16(x - x1).  out (rows, 3 Kp) bf16, Kp >= cols (zero padded), Kp % 8 == 0: which = 0 -> [x1 | x1 | x2] (activations),
 * which = 1 -> [w1 | w2 | w1] (nn.Linear weights), so that se_gemm_bf16 on the two outputs with K = 3 Kp sums x1 w1 + x1 w2 + x2 w1 in its fp32
 * accumulators: fp32-operand products to 3 . 2^-18 relative on the bf16 matrix pipe (runner.py:556-575 at the 1e-4 tolerance, 3 x the bf16 cost). */
int se_split3_bf16(const float* x, long ld, int rows, int cols, int Kp, int which, uint16_t* out, void* stream);

/*
 * Producers of the three-term operand (round 4): the bf16x3 mode's projections, LayerNorms and attention core hand their fp32 result to the next
 * projection already split -- [y1 | y1 | y2], row stride 3 Kp, se_split3_bf16's activation layout -- instead of writing fp32 rows that a separate
 * se_split3_bf16 pass reads back (14 % of that mode's pass).  Same S3PRL rows as the fp32 mode: nn.Linear (+ gelu) of the encoder layers (B2 / B3),
 * their LayerNorms, the attention core behind model.py:164.  Unlike se_split3_bf16 these producers write NO pad columns: Kp must EQUAL the slice width
 * (N, H, heads * 64 respectively; SE_ERR_INVALID otherwise), so that the next projection's depth 3 Kp never sums over unwritten columns.
 *   se_gemm_x3out_bf16      out3 = split(act(A . W^T + bias)); A (M, K) / W (N, K) three-term operands (K = 3 x the layer's depth); act identity / GELU (erf)
 *   se_layernorm_x3_f32     out_f32 (may be NULL) = LayerNorm(x), out3 = split(LayerNorm(x)); H = 256, 512, 768, 1024
 *   se_mhsa_fwd_x3_split_f32  se_mhsa_fwd_x3_f32 with the context written as its split
 *   se_gemm_res_ln_x3_bf16  the row-complete projection + residual + LayerNorm launch of the bf16 path (se_gemm_res_ln_bf16, N = 768) on three-term
 *                           operands: out_f32 = LayerNorm(A . W^T + bias + residual_f32) (the next residual), out3 (M, 3 x 768) = its split --
 *                           the attention-output and FFN-output dense + LayerNorm of model.py's encoder layers in one launch
 */
int se_gemm_x3out_bf16(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, int M, int N, int K, int act,
                       uint16_t* out3, int Kp, void* stream);
int se_layernorm_x3_f32(const float* x, const float* w, const float* b, int M, int H, float eps, float* out_f32, uint16_t* out3, int Kp,
                        void* stream);
int se_mhsa_fwd_x3_split_f32(const float* qkv, const int32_t* lengths, int B, int T, int heads, uint16_t* ctx3, int Kp, void* stream);
int se_gemm_res_ln_x3_bf16(const uint16_t* A, int lda, const uint16_t* W, int ldw, const float* bias, const float* residual_f32, const float* ln_w,
                           const float* ln_b, float eps, int M, int N, int K, float* out_f32, uint16_t* out3, void* stream);
/* The attention core of the same mode (csrc/mhsa_x3.hip): ctx (B*T, H) fp32 = softmax(Q K^T / 8 + pad mask) V from the fp32 fused projection
 * qkv (B*T, 3H) = [Q | K | V], flash style on the bf16 matrix pipe with two-term splits of Q, K, V and P (three products each); replaces the
 * materialised scores of the fp32 mode (se_gemm_f32 batched + se_softmax_rows_f32).  heads of 64; buffers 16-B aligned. */
int se_mhsa_fwd_x3_f32(const float* qkv, const int32_t* lengths, int B, int T, int heads, float* ctx, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Backward building blocks (row E2 beyond the linear heads: autograd through TransformerSpecPredictionHead /
 * SpecHead, model.py:94-126, runner.py:459).  Mixed precision as the forward: bf16 GEMM operands, fp32 sums.
 * The input gradient dX = dY . W is se_gemm_bf16 on a transposed bf16 weight copy (se_transpose_*).
 * ---------------------------------------------------------------------------------------------- */
/* out[c][r] = in[r][c], r < rows; out rows are ld_out long, zero filled for rows <= r < ld_out */
int se_transpose_bf16(const uint16_t* in, int rows, int cols, int ld_in, uint16_t* out, int ld_out, void* stream);
int se_transpose_f32_bf16(const float* in, int rows, int cols, int ld_in, uint16_t* out, int ld_out, void* stream);
/* weight gradient dW[N,K] (+)= dY^T . X from the TRANSPOSED operands dYt (N, Mp), Xt (K, Mp) bf16 (reduction dim contiguous):
 * split-K over Mp on the forward GEMM kernel + slab reduce.  Mp % splits == 0, (Mp / splits) % 64 == 0 and >= 128;
 * workspace >= splits * N * K * 4 bytes. */
int se_wgrad_bf16(const uint16_t* dYt, const uint16_t* Xt, int Mp, int N, int K, int splits, float* dW, int accumulate,
                  void* workspace, size_t workspace_bytes, void* stream);
/* same result straight from the ROW-MAJOR operands (no transposes): dW[N,K] (+)= dY[M,N]^T . X[M,K]; the fragments are taken
 * from row-major LDS panels with the hardware-transposed LDS read.  N, K, ldy, ldx multiples of 8; workspace >= splits*N*K floats. */
int se_wgrad_tn_bf16(const uint16_t* dY, int ldy, const uint16_t* X, int ldx, int M, int N, int K, int splits, float* dW,
                     int accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* per-group weight gradients (no reduce): slabs[g] (N, K) = dY[g R .. (g+1) R)^T . X[same rows], R = rows_per_slab -- with R = frames
 * per utterance these are the PER-UTTERANCE gradients active sampling scores with (sampler.py:59-111) from one launch. */
int se_wgrad_tn_slabs_bf16(const uint16_t* dY, int ldy, const uint16_t* X, int ldx, int M, int N, int K, int rows_per_slab,
                           float* slabs, void* stream);
/* bias gradient: out[c] (+)= sum_r x[r][c] */
int se_colsum_f32(const float* x, int rows, int cols, int ld, float* out, int accumulate, void* stream);
/* out[g][c] = sum over the `rows` rows of group g: per-utterance bias gradients of the active-sampling scoring (sampler.py:59-110) in one
   launch; x is fp32 or (is_bf16) bf16, (groups * rows, ld). */
int se_colsum_groups(const void* x, int is_bf16, int groups, int rows, int cols, int ld, float* out, void* stream);
/* LayerNorm backward (TF style).  x_in = LayerNorm input (gelu_in: its pre-GELU value, i.e. y = LN(gelu(x_in)), and
 * the returned gradient is wrt x_in).  dx / dx_bf16 (M, H); dgamma, dbeta (H) accumulated by atomics.  H = 768. */
int se_layernorm_bwd_f32(const float* x_in, const float* dy, const float* w, int M, int H, float eps, int gelu_in,
                         float* dx, uint16_t* dx_bf16, float* dgamma, float* dbeta, int accumulate, void* stream);
/* The same over `groups` blocks of `rows` rows with one (dgamma, dbeta) row per block: every utterance's LayerNorm parameter gradient
   of the active-sampling scoring in one launch (sampler.py:84-104 runs one backward per utterance).  dgamma / dbeta: (groups, H). */
int se_layernorm_bwd_groups_f32(const float* x_in, const float* dy, const float* w, int groups, int rows, int H, float eps, int gelu_in,
                                float* dx, uint16_t* dx_bf16, float* dgamma, float* dbeta, void* stream);
/* y = LayerNorm(gelu(pre)) (spec-head transform, training path keeps `pre`).  H = 768. */
int se_gelu_layernorm_f32(const float* pre, const float* w, const float* b, int M, int H, float eps,
                          float* out_f32, uint16_t* out_bf16, void* stream);
/* SpecHead.forward's epilogue (model.py:121-125) on a raw linear output p (n elements) and its backward wrt p.
   log_target 0 / 1: SpecHead's two branches; 2: LSTM.forward's (model.py:56-58) log_predicted = act(p), predicted = exp(log_predicted). */
int se_spec_epilogue_f32(const float* p, size_t n, int log_target, int act, float eps, float* predicted, float* log_predicted, void* stream);
int se_spec_epilogue_bwd_f32(const float* p, const float* d_pred, const float* d_logp, int M, int N, int ldp, int log_target,
                             int act, float eps, float* dp_f32, uint16_t* dp_bf16, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Training path of the encoder (rows B1-B3 under autograd; C4 `Mockingjay` fine-tune and E2: model.py:163-171,
 * runner.py:453-471).  Replaces what torch autograd records / replays through TRANSFORMER.forward.
 * hidden_size must be 768 (256 is also built, for small-size parity tests).
 * ---------------------------------------------------------------------------------------------- */
/* flash MHSA forward that also stores the per-(utterance, head, query) log-sum-exp (log2 domain), (B, heads, T) fp32.
 * dropout_p > 0 drops the attention probabilities (attention_probs_dropout_prob, pretrain_sample.yaml:10) with the
 * counter-based mask of (seed, site) -- csrc/dropout.h; the backward regenerates it from the same three values. */
int se_mhsa_fwd_lse_bf16(const uint16_t* qkv, const int32_t* lengths, int B, int T, int heads, uint16_t* ctx, float* lse,
                         float dropout_p, uint64_t seed, uint32_t site, void* stream);
/* flash MHSA backward: d_ctx (B*T, H) bf16 -> dqkv (B*T, 3H) bf16 = [dQ | dK | dV]; dvec (B, heads, T) fp32 scratch */
int se_mhsa_bwd_bf16(const uint16_t* qkv, const uint16_t* ctx, const uint16_t* d_ctx, const float* lse, const int32_t* lengths,
                     int B, int T, int heads, uint16_t* dqkv, float* dvec, float dropout_p, uint64_t seed, uint32_t site, void* stream);
/* y = gelu(x) and dx = dy * gelu'(x), bf16 arrays of n elements (n % 8 == 0) */
int se_gelu_bf16(const uint16_t* x, size_t n, uint16_t* y, void* stream);
int se_gelu_bwd_bf16(const uint16_t* dy, const uint16_t* x, size_t n, uint16_t* dx, void* stream);

/* DEVICE pointers to the fp32 gradients, same members / layouts as se_encoder_weights (without the spec head) */
typedef struct se_encoder_grads {
  float *in_w, *in_b, *in_ln_w, *in_ln_b;
  float* const *q_w, * const *q_b, * const *k_w, * const *k_b, * const *v_w, * const *v_b;
  float* const *ao_w, * const *ao_b, * const *aln_w, * const *aln_b;
  float* const *ff1_w, * const *ff1_b, * const *ff2_w, * const *ff2_b, * const *oln_w, * const *oln_b;
} se_encoder_grads;

/* re-pack the encoder's bf16 operand copies from DEVICE fp32 master weights (after each optimizer step; the spec-head
 * members of `w` are ignored) */
int se_encoder_refresh_bf16(se_encoder* enc, const se_encoder_weights* w, void* stream);
size_t se_encoder_saved_bytes(const se_encoder* enc, int B, int T);
size_t se_encoder_train_workspace_bytes(const se_encoder* enc, int B, int T);
/* forward that keeps the activations the backward needs in `saved` (caller-owned, se_encoder_saved_bytes).
 * dropout_p > 0: training mode with BERT's dropout sites (after the input LayerNorm, attention probabilities, after the
 * attention-output and FFN-output dense; hidden_dropout_prob = attention_probs_dropout_prob = dropout_p) and counter-based
 * masks from `seed` (csrc/dropout.h) -- nothing is stored, the backward regenerates them from the same (dropout_p, seed). */
int se_encoder_fwd_train_bf16(const se_encoder* enc, const float* feats, const int32_t* lengths, int B, int T, float* hidden,
                              void* saved, size_t saved_bytes, void* workspace, size_t workspace_bytes, float dropout_p, uint64_t seed,
                              void* stream);
/* backward: d_hidden (B, T, H) fp32 -> every parameter gradient (overwritten, not accumulated) */
int se_encoder_bwd_bf16(const se_encoder* enc, const int32_t* lengths, int B, int T, const float* d_hidden, const void* saved,
                        size_t saved_bytes, const se_encoder_grads* grads, void* workspace, size_t workspace_bytes, float dropout_p,
                        uint64_t seed, void* stream);
/* The same backward with a host callback after each layer's launches are enqueued (layer L-1 first, -1 = input stage): that layer's
   parameter gradients are then ordered on `stream`, so a data-parallel caller can start the layer's gradient all-reduce while the
   remaining layers' kernels run (dist.py: bucketed overlap).  layer_done may be NULL. */
int se_encoder_bwd_cb_bf16(const se_encoder* enc, const int32_t* lengths, int B, int T, const float* d_hidden, const void* saved,
                           size_t saved_bytes, const se_encoder_grads* g, void* workspace, size_t workspace_bytes, float dropout_p,
                           uint64_t seed, void (*layer_done)(int layer, void* user), void* user, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optimizer side of row E2 (runner.py:463-471): gradient norms and BertAdam for all parameter tensors in two launches.
 * The pointer / size arrays are HOST arrays of n_tensors entries (device pointers inside); they are consumed during the
 * call (copied into kernel arguments), the launches themselves are asynchronous.
 * ---------------------------------------------------------------------------------------------- */
/* sumsq[t] = sum of squares of gradient tensor t  (device double[n_tensors], overwritten) */
int se_multi_sumsq_f32(const float* const* grads, const uint64_t* sizes, int n_tensors, double* sumsq, void* stream);
/* BertAdam (S3PRL downstream.solver.get_optimizer, runner.py:110-113,470): Adam without bias correction, decoupled weight
 * decay (per tensor), per-tensor gradient clip at max_grad_norm, preceded by the runner's global clip_grad_norm_ at
 * global_max_norm (runner.py:464; <= 0 disables either).  sumsq from se_multi_sumsq_f32 on the same gradients. */
int se_bertadam_step_f32(float* const* params, const float* const* grads, float* const* m, float* const* v, const uint64_t* sizes,
                         const float* weight_decay, int n_tensors, const double* sumsq, double lr_t, double b1, double b2, double e,
                         double max_grad_norm, double global_max_norm, void* stream);
/* dsts[t][0 .. sizes[t]) = srcs[t][...] for all tensors in one launch per 96 tensors: the gather of the per-parameter gradients into the
 * flat all-reduce buffer (the build's data-parallel step; runner.py:459-471 has no counterpart, it is single-process) */
int se_multi_copy_f32(float* const* dsts, const float* const* srcs, const uint64_t* sizes, int n_tensors, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The stages either side of the hot path, batched on the device.
 * ---------------------------------------------------------------------------------------------- */
/* OnlineDataset.__getitem__'s arithmetic (dataset.py:141-161): per utterance b, speech (len_s[b] samples of row b) and
 * noise (len_n[b] samples of row b starting at off_n[b] -- `half_noise`, dataset.py:147-152; off_n may be NULL) are
 * level-normalised (normalize != 0: normalize_wav_decibel, dataset.py:106-111), the noise is tiled / cut to the speech
 * length and mixed at snr_db[b] (add_noise, dataset.py:54-74, eps as OnlineDataset.eps), and written as
 * wavs (B, 3, T_out) = (noisy, clean, scaled noise), zero padded past len_s[b] (collate_fn, dataset.py:169-179).
 * sums: device double[3 B] scratch.  lengths of the batch are len_s. */
int se_mix_f32(const float* speech, int ld_s, const int64_t* len_s, const float* noise, int ld_n, const int64_t* len_n,
               const int64_t* off_n, const float* snr_db, int B, int T_out, int normalize, float target_level_db, float eps,
               float* wavs, double* sums, void* stream);
/* evaluation.sisdr_eval (evaluation.py:5-10) for each utterance over its first lengths[b] samples (runner.py:597-603);
 * src, tar (B, ld) fp32; sums: device double[3 B] scratch; sisdr (B) fp32 in dB. */
int se_sisdr_f32(const float* src, const float* tar, int ld, const int64_t* lengths, int B, float eps, double* sums, float* sisdr,
                 void* stream);

/* ------------------------------------------------------------------------------------------------
 * Recurrent core of the (Bi)LSTM downstream heads (model.py:37-91, nn.LSTM hidden 256).
 * The input projection and all gradient GEMMs run on se_gemm_bf16 / se_wgrad_tn_bf16; these two kernels are the sequential
 * part: one 512-thread workgroup per (utterance, direction) with W_hh resident in registers + LDS as MFMA fragments for all T steps.
 * ---------------------------------------------------------------------------------------------- */
/* w_hh [ndir][1024][256] bf16 = nn.LSTM's weight_hh_l* (row-major, gate order i, f, g, o); xproj [ndir][B][T][1024] fp32 = x W_ih^T + b_ih
 * + b_hh; direction 1 runs t = T-1 .. 0.  Outputs: h_out (B, T, ndir*256) bf16 (directions concatenated, the next
 * layer's input), gates_out [ndir][B][T][1024] fp32 post-activation gates and c_out [ndir][B][T][256] fp32 (kept for the backward). */
int se_lstm_fwd_bf16(const uint16_t* w_hh, const float* xproj, int B, int T, int ndir, uint16_t* h_out, float* gates_out, float* c_out,
                     void* stream);
/* BPTT: w_hh_t [ndir][256][1024] bf16 = W_hh^T (row-major); dh_out (B, T, ld_dh) fp32 =
 * gradient wrt h_out (columns dir*256 ..); dgates_out [ndir][B][T][1024] bf16 = gradient wrt the pre-activation gates. */
int se_lstm_bwd_bf16(const uint16_t* w_hh_t, const float* gates, const float* c_saved, const float* dh_out, int ld_dh, int B, int T,
                     int ndir, uint16_t* dgates_out, void* stream);
/* column sums of a bf16 matrix (bias gradients): out[c] = sum_r x[r][c]; cols and ld multiples of 8 */
int se_colsum_bf16(const uint16_t* x, int rows, int cols, int ld, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Spectrogram-domain criteria besides L1: what the mask heads are trained with.
 * frame_lengths (B) int64 = valid frames per utterance (the stft_length_masks of runner.py:216-220 as counts).
 * ---------------------------------------------------------------------------------------------- */
/* objective.SISDR (objective.py:81-100): loss_b[b] per utterance (the criterion is their mean over the batch); grad (B,F,N),
 * optional, = grad_scale * d loss_b / d predicted (pass 1/B, or 1/global_B under data parallelism). scratch: device double[5 B]. */
int se_sisdr_spec_f32(const float* predicted, const float* linear_tar, const int64_t* frame_lengths, int B, int F, int N, float eps,
                      float grad_scale, double* scratch, float* loss_b, float* grad, void* stream);

/* The same criterion for the evaluate()-style pass (no gradient) as two launches: `lengths` are frame counts (len_div == 0) or WAVEFORM lengths
 * (frames = lengths / len_div + 1, runner.py:455), partial sums in `scratch` (se_sisdr_spec_loss_scratch_doubles(B, F, N) doubles, no clearing),
 * loss_b (B), sums_out = {sum_b loss_b, B}, loss_out = their ratio: objective.py:81-100 incl. its mean over the utterances.  Fixed summation order. */
size_t se_sisdr_spec_loss_scratch_doubles(int B, int F, int N);
int se_sisdr_spec_loss_f32(const float* predicted, const float* linear_tar, const int64_t* lengths, int len_div, int B, int F, int N, float eps,
                           double* scratch, float* loss_b, double* sums_out, float* loss_out, void* stream);

/* objective.WSD (objective.py:119-153) in two steps, so that a data-parallel caller can all-reduce(MAX) the batch-wide energy
 * maximum in between: energy (B*F) = sum_n linear_tar, energy_max (1 float) = its maximum over every frame of the batch; then
 * sums[0..1] = (sum_b speech_b, sum_b noise_b) (device double[3]) and grad (optional) = grad_scale * d(alpha sums[0] +
 * (1 - alpha) sums[1]) / d offset. */
int se_wsd_energy_f32(const float* linear_tar, int B, int F, int N, float* energy, float* energy_max, void* stream);
int se_wsd_f32(const float* linear_inp, const float* offset, const float* linear_tar, const int64_t* frame_lengths, const float* energy,
               const float* energy_max, int B, int F, int N, float alpha, float db_interval, float eps, float grad_scale, double* sums,
               float* grad, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Optional in-library timing for bench.py's roofline leg: HIP events recorded on the launch stream around
 * every kernel of a family while enabled.  kind: 0 = bf16 GEMM (work = 2MNK flop), 1 = MHSA (4 B h T^2 64 flop),
 * 2 = STFT, 3 = iSTFT (work = algorithmic bytes), 4 = LayerNorm, 5 = head, 6 = MHSA backward (14 B h T^2 64 flop).  se_prof_read synchronises on the recorded events.
 * ---------------------------------------------------------------------------------------------- */
int se_prof_enable(int on);
int se_prof_reset(void);
int se_prof_read(int kind, double* total_ms, double* total_work, long long* launches);

#ifdef __cplusplus
}
#endif
#endif /* SE_AMD_H_ */
