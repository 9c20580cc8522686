"""GPU: size-independent properties of the hot path AT BASELINE.json's FULL SIZES, where the CPU oracle would take minutes: configs[1] (B = 32 utterances
of 10 s, T' = 1001, 6 x 768 x 12 x 3072) and configs[3] / [0] (vcb.yaml: (B, 2, T) batches of 256).  Each property holds for the reference's
arithmetic whatever the size (runner.py:556-575, model.py:20-34, utils.py:26-46), so it needs no oracle run: utterance independence (every stage
is per utterance: a batch must give each utterance what it gets alone), STFT -> iSTFT reconstruction, homogeneity of the power spectrogram,
rows of a softmax summing to one, zero loss of identical inputs, invariance of the decoded waveform to the scale of the predicted magnitudes."""
import pytest
import torch

from conftest import bounded

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


def _mockingjay_grads(model, crit, feats, tar, lens):
    for p in model.parameters():
        p.grad = None
    pred, res = model(features=feats)
    loss, _ = crit(log_predicted=res['log_predicted'], linear_tar=tar, stft_lengths=lens)
    loss.backward()
    return float(loss.detach()), {n: p.grad.detach().double().clone() for n, p in model.named_parameters()}


def test_finetune_step_full_size_equals_the_sum_of_its_utterances(gpu):
    """configs[2]'s training step at its per-GPU shape (B = 32, L = 6, T' = 1001: 32 032 rows, the sizes at which the row-complete N = 768 GEMMs, the
    persistent FFN1 dual store and the staggered weight-gradient kernel switch on -- runner.py:431-471) checked through a size-independent property:
    the loss is the GLOBAL masked mean (objective.py:113-116), so with S_b / C_b the L1 sum / element count of utterance b
        loss = sum_b S_b / sum_b C_b        and        grad(loss) = sum_b (C_b / C) grad(loss_b),   loss_b = S_b / C_b,
    where every loss_b and its gradient comes from the SAME model run on that utterance alone -- 1 001 rows, i.e. through the small-M kernels and the
    unfused paths.  Dropout is off here (its masks are keyed by element index, so utterance b alone would draw other masks); the sign gradient of L1
    flips where bf16 rounding moves log_predicted across the target, hence the cosine / norm form of the gradient bound (as the per-utterance
    gradient scoring test)."""
    from speech_enhancement_by_s3prl_amd import pipeline
    from speech_enhancement_by_s3prl_amd.objective import L1
    cfg = pipeline.make_config(layers=6)
    cfg['transformer'].update(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    ckpt = pipeline.synthetic_checkpoint(cfg, seed=11)
    model = pipeline.build_mockingjay(ckpt, gpu)
    model.train()
    B, T = 32, 1001
    g = torch.Generator().manual_seed(77)
    feats = torch.randn(B, T, 80, generator=g).to(gpu)
    tar = (torch.rand(B, T, 201, generator=g) + 0.05).to(gpu)
    lens = torch.randint(500, T + 1, (B,), generator=g)
    lens[0] = T
    feats = feats * (torch.arange(T)[None, :, None] < lens[:, None, None]).to(gpu)      # S3PRL derives the valid frames from all-zero feature rows
    lens = lens.to(gpu)
    crit = L1()
    loss_full, g_full = _mockingjay_grads(model, crit, feats, tar, lens)
    counts = lens.double() * 201
    C = counts.sum().item()
    acc = {n: torch.zeros_like(v) for n, v in g_full.items()}
    s_sum = 0.0
    for b in range(B):
        lb, gb = _mockingjay_grads(model, crit, feats[b:b + 1], tar[b:b + 1], lens[b:b + 1])
        s_sum += lb * counts[b].item()
        for n in acc:
            acc[n] += gb[n] * (counts[b].item() / C)
    assert abs(loss_full - s_sum / C) < 2e-3 * abs(loss_full), (loss_full, s_sum / C)
    worst_cos, worst_norm = 1.0, 0.0
    for n in acc:
        if n.endswith('attention.self.key.bias'):
            continue                  # its true gradient is ZERO (a constant added to every key's score leaves the softmax unchanged): both sides are rounding noise
        a, r = g_full[n].flatten(), acc[n].flatten()
        cos = float(torch.dot(a, r) / (a.norm() * r.norm() + 1e-300))
        ratio = float(a.norm() / (r.norm() + 1e-300))
        worst_cos, worst_norm = min(worst_cos, cos), max(worst_norm, abs(ratio - 1.0))
        assert cos > 0.985 and abs(ratio - 1.0) < 0.05, (n, cos, ratio)
    bounded('finetune_fullsize 1 - min cosine(grad, sum of per-utterance grads)', 1.0 - worst_cos, 2e-3)
    bounded('finetune_fullsize max |norm ratio - 1|', worst_norm, 5e-3)


def test_finetune_step_dispatch_switches_agree():
    """the same full-size step (train mode, dropout 0.1, identical seeded masks) with the large-M dispatch of round 4 ON (default) and OFF
    (SE_AMD_GEMM7_PLAIN=0: 256 x 256 tiles instead of the row-complete N = 768 kernel; SE_AMD_GEMM6_DUAL=0: FFN1 pre-activation and GELU as two
    launches; SE_AMD_WGRAD_STAG=0: the unstaggered weight-gradient ring): loss, gradient norm and a seeded sample of every parameter's gradient agree
    to the bf16 bound.  Two subprocesses (the switches are read once per process)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(env_extra):
        env = dict(os.environ, **env_extra)
        r = subprocess.run([sys.executable, os.path.join(root, 'tests', '_finetune_fullsize_worker.py')], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith('RESULT ')][-1]
        return json.loads(line[7:])

    on = run({})
    off = run({'SE_AMD_GEMM7_PLAIN': '0', 'SE_AMD_GEMM6_DUAL': '0', 'SE_AMD_WGRAD_STAG': '0'})
    assert abs(on['loss'] - off['loss']) < 2e-3 * abs(off['loss']), (on['loss'], off['loss'])
    assert abs(on['gnorm'] - off['gnorm']) < 3e-2 * off['gnorm'], (on['gnorm'], off['gnorm'])
    import torch as _t
    a = _t.tensor([v for n in sorted(on['samples']) for v in on['samples'][n]], dtype=_t.float64)
    b = _t.tensor([v for n in sorted(off['samples']) for v in off['samples'][n]], dtype=_t.float64)
    cos = float(_t.dot(a, b) / (a.norm() * b.norm()))
    assert cos > 0.99, cos


def test_scored_head_pass_at_vcb_batch_mel_features(gpu):
    """configs[0] / [3] with pseudo_noise.yaml's features (mel / log / delta-2, 120 dims) and the SISDR criterion at B = 256: the round-5 route (one-pass
    feature launch handing over its column statistics, branch-free head taking the criterion's sums) against (a) the criterion recomputed from the
    returned planes by its own launches, (b) the same utterances in a batch of 3 (every stage is per utterance; the loss is a mean over utterances),
    (c) the defining property of the statistics: the head's CMVN input has zero mean and unit unbiased std per (utterance, column)"""
    from speech_enhancement_by_s3prl_amd import pipeline, synth
    from speech_enhancement_by_s3prl_amd.heads import LinearResidual
    from speech_enhancement_by_s3prl_amd.objective import SISDR, sisdr_loss_inference
    pre = pipeline.build_preprocessor(pipeline.make_config(), gpu, upstream='baseline')
    torch.manual_seed(1)
    head = LinearResidual(input_size=120, output_size=201, cmvn=True).to(gpu)
    step = pipeline.HeadEnhanceStep(pre, head, criterion=SISDR())
    lengths, wavs = synth.fast_batch(256, 160000, seed=13, device=gpu)
    wavs = wavs[:, :2].contiguous()
    lengths = lengths.clone()
    lengths[7], lengths[130] = 90000, 12345
    wavs[7, :, 90000:] = 0.0
    wavs[130, :, 12345:] = 0.0
    wav_all, pred_all, tar_all, loss = step(wavs, lengths, max_len=160000)
    assert getattr(pred_all, '_se_sisdr', None) is not None            # the criterion's sums came out of the head's launch
    # (a)
    l2, lb = sisdr_loss_inference(pred_all, tar_all, lengths, 160, 1e-10)
    bounded('scored_head_pass fused vs own-launch criterion', abs(loss.item() - l2.item()) / abs(l2.item()), 1e-5)
    # (b)
    idx = [7, 130, 255]
    w3, p3, t3, l3 = step(wavs[idx].contiguous(), lengths[idx].contiguous(), max_len=160000)
    bounded('scored_head_pass predicted, batch of 3 vs 256', _relmax(p3, pred_all[idx]), 2e-5)
    bounded('scored_head_pass waveform, batch of 3 vs 256', _relmax(w3, wav_all[idx]), 2e-5)
    bounded('scored_head_pass loss of 3 = mean of their loss_b', abs(l3.item() - lb[idx].double().mean().item()) / abs(l3.item()), 1e-5)
    # (c)
    feats = pre(wavs[:8].contiguous())[1]
    cst = feats._se_colstats[0]
    z = (feats.double() - cst[..., 0].double()[:, None, :]) * cst[..., 1].double()[:, None, :]
    assert z.mean(dim=1).abs().max().item() < 1e-4
    assert (z.std(dim=1, unbiased=True) - 1.0).abs().max().item() < 1e-3
