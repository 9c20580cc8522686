"""GPU: size-independent properties of the hot path AT BASELINE.json's FULL SIZES, where the CPU oracle would take minutes: configs[1] (B = 32 utterances
of 10 s, T' = 1001, 6 x 768 x 12 x 3072) and configs[3] / [0] (vcb.yaml: (B, 2, T) batches of 256).  Each property holds for the reference's
arithmetic whatever the size (runner.py:556-575, model.py:20-34, utils.py:26-46), so it needs no oracle run: utterance independence (every stage
is per utterance: a batch must give each utterance what it gets alone), STFT -> iSTFT reconstruction, homogeneity of the power spectrogram,
rows of a softmax summing to one, zero loss of identical inputs, invariance of the decoded waveform to the scale of the predicted magnitudes."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _relmax(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / b.abs().max()).item()


@pytest.fixture(scope='module')
def P(gpu):
    from speech_enhancement_by_s3prl_amd import pipeline
    return pipeline.build_preprocessor(pipeline.make_config(), gpu, upstream='baseline',
                                       downstream_feat={'feat_type': 'linear', 'log': False, 'delta': 0, 'cmvn': False})       # vcb.yaml:10-14


def test_stft_istft_roundtrip_at_vcb_batch(P, gpu):
    """256 x 2 channels x 160 000 samples: istft(stft(x)) == x on every sample with full window coverage (torch.stft / istft, center = True),
    through the encoded-phase path the pipelines use"""
    from speech_enhancement_by_s3prl_amd import synth
    lengths, wavs = synth.fast_batch(256, 160000, seed=11, device=gpu)
    wavs = wavs[:, :2].contiguous()
    feats_up, feats_down, lin_inp, ph_inp, lin_tar, ph_tar = P(wavs)
    assert lin_inp.shape == (256, 1001, 201)
    rec = P.istft(lin_inp, ph_inp)
    n = min(rec.shape[-1], 160000)
    x = wavs[:, 0, :n]
    assert _relmax(rec[:, 400:n - 400], x[:, 400:n - 400]) < 2e-5
    # homogeneity: the power spectrogram of 3 x is 9 x the power spectrogram
    lin3 = P(wavs * 3.0)[2]
    assert _relmax(lin3, lin_inp * 9.0) < 1e-5
    # Parseval-type check on the analysis alone: the frame energies follow the windowed signal's energy (hann^2 overlap-add constant 1.5 at hop = win / 2.5)
    assert torch.isfinite(lin_inp).all() and (lin_inp >= 0).all()


def test_head_pass_gives_every_utterance_what_it_gets_alone(P, gpu):
    """configs[3]: the evaluate()-style pass on a batch of 256 (B, 2, T) against the same utterances in batches of 1 and 12 (vcb.yaml:3):
    STFT, CMVN, the fp32 mask GEMM, iSTFT and the dB normalisation are all per utterance"""
    from speech_enhancement_by_s3prl_amd import pipeline, synth
    from speech_enhancement_by_s3prl_amd.heads import LinearResidual
    torch.manual_seed(0)
    head = LinearResidual(input_size=201, output_size=201, cmvn=True).to(gpu)
    step = pipeline.HeadEnhanceStep(P, head)
    lengths, wavs = synth.fast_batch(256, 160000, seed=12, device=gpu)
    wavs = wavs[:, :2].contiguous()
    lengths = lengths.clone()
    lengths[5], lengths[200] = 100000, 31234                       # ragged, zero padded as collate_fn does (dataset.py:169-179)
    wavs[5, :, 100000:] = 0.0
    wavs[200, :, 31234:] = 0.0
    wav_all, pred_all, tar_all, _ = step(wavs, lengths, max_len=160000)
    assert wav_all.shape[0] == 256 and torch.isfinite(wav_all).all()
    for i in (0, 5, 200, 255):
        w1, p1, t1, _ = step(wavs[i:i + 1], lengths[i:i + 1], max_len=160000)
        assert _relmax(p1[0], pred_all[i]) < 1e-6                  # same kernels, same per-row summation order
        assert _relmax(w1[0], wav_all[i]) < 1e-5
    w12, p12, _, _ = step(wavs[192:204], lengths[192:204], max_len=160000)
    assert _relmax(p12[8], pred_all[200]) < 1e-6 and _relmax(w12[8], wav_all[200]) < 1e-5


def test_decoded_waveform_is_invariant_to_the_scale_of_the_prediction(P, gpu):
    """masked_normalize_decibel (utils.py:26-46) rescales the decoded waveform to the target's level: istft is linear in the magnitude, so
    predicted power x 4 (magnitude x 2) must give the same normalised waveform -- at the full vcb batch"""
    from speech_enhancement_by_s3prl_amd import decode, synth
    lengths, wavs = synth.fast_batch(256, 160000, seed=13, device=gpu)
    wavs = wavs[:, :2].contiguous()
    _, _, lin_inp, ph_inp, _, _ = P(wavs)
    a = decode.decode_wav(P, lin_inp, ph_inp, lengths, wavs[:, 1], max_len=160000)
    b = decode.decode_wav(P, lin_inp * 4.0, ph_inp, lengths, wavs[:, 1], max_len=160000)
    assert _relmax(b, a) < 1e-5


def test_encoder_pass_gives_every_utterance_what_it_gets_alone(gpu):
    """configs[1] at the bench shape (B = 32, T' = 1001: the persistent 256 x 256 GEMMs, the row-complete GEMM + LayerNorm with the 24-bit
    residual stream, the pre-scaled speculative attention) against the same utterances alone (1 001 rows: the small-M kernels, the fp32 residual
    stream): two different kernel sets computing the same per-utterance function, within the bf16 bound of the pipeline tests"""
    from speech_enhancement_by_s3prl_amd import pipeline, synth
    cfg = pipeline.make_config()
    ckpt = pipeline.synthetic_checkpoint(cfg, seed=0)
    up = pipeline.build_upstream(ckpt, gpu)
    pre = pipeline.build_preprocessor(cfg, gpu)
    lengths, wavs = synth.fast_batch(32, 160000, seed=14, device=gpu)
    lengths = lengths.clone()
    lengths[3] = 70000
    wavs[3, :, 70000:] = 0.0
    with torch.no_grad():
        hid = up(pre(wavs)[0])
        assert hid.shape == (32, 1001, 768) and torch.isfinite(hid).all()
        for i in (0, 3, 31):
            h1 = up(pre(wavs[i:i + 1])[0])
            n = int(lengths[i]) // 160 + 1
            rel = ((h1[0, :n] - hid[i, :n]).double().norm() / hid[i, :n].double().norm()).item()
            assert rel < 6e-3, (i, rel)


def test_attention_rows_sum_to_one_at_the_bench_shape(gpu):
    """B = 32, T = 1001, 12 heads: with V = 1 the context is the row sum of the probabilities = 1 for every query, whatever the scores and the
    lengths (bf16 output: 2^-8); both the inference (pre-scaled, speculative) and the training (log-sum-exp) forward"""
    from speech_enhancement_by_s3prl_amd import _lib as L
    lib = L.load()
    B, T, heads = 32, 1001, 12
    H = 64 * heads
    torch.manual_seed(15)
    qkv = torch.randn(B * T, 3 * H, device=gpu)
    qkv[:, 2 * H:] = 1.0
    qkv = qkv.bfloat16()
    lengths = torch.randint(1, T + 1, (B,), device=gpu, dtype=torch.int32)
    lengths[0] = T
    ctx = torch.empty(B * T, H, device=gpu, dtype=torch.bfloat16)
    L.check(lib.se_mhsa_fwd_prescaled_bf16(L.ptr(qkv), L.ptr(lengths), B, T, heads, L.ptr(ctx), L.stream()), 'mhsa')
    assert (ctx.float() - 1.0).abs().max().item() <= 2.0 ** -7
    lse = torch.empty(B, heads, T, device=gpu)
    L.check(lib.se_mhsa_fwd_lse_bf16(L.ptr(qkv), L.ptr(lengths), B, T, heads, L.ptr(ctx), L.ptr(lse), 0.0, 0, 0, L.stream()), 'mhsa_lse')
    assert (ctx.float() - 1.0).abs().max().item() <= 2.0 ** -7 and torch.isfinite(lse).all()


def test_losses_of_identical_inputs_at_full_size(gpu):
    """L1 (objective.py:103-117) of log(x + eps) against x is exactly 0 and SI-SDR (objective.py:81-100) of a signal with itself is at its
    eps-limited maximum, for 256 x 1001 x 201 spectrograms with ragged lengths"""
    from speech_enhancement_by_s3prl_amd.objective import L1, SISDR
    torch.manual_seed(16)
    B, F, N = 256, 1001, 201
    tar = torch.rand(B, F, N, device=gpu) + 0.05
    lens = torch.randint(1, F + 1, (B,), device=gpu)
    crit = L1()
    log_tar = torch.log(tar.cpu() + crit.eps).to(gpu)       # objective.py:116 with the criterion's own eps (1e-10), libm log on the CPU
    loss, _ = crit(log_predicted=log_tar, linear_tar=tar, stft_lengths=lens)
    assert abs(loss.item()) < 2e-7
    s, _ = SISDR()(predicted=tar, linear_tar=tar, stft_lengths=lens)
    assert s.item() < -60.0            # -SI-SDR in dB: a perfect estimate is limited only by eps
