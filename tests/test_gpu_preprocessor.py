"""GPU parity: rows A1-A6 (STFT / power+phase / mel / select_feat / iSTFT) -- HIP path vs the CPU oracle,
called through the C ABI (ctypes) via the OnlinePreprocessor drop-in.

Tolerances (north_star: 1e-4 relative on magnitudes): power and waveform are compared norm-wise per
utterance, max|a-b| <= 1e-4 * max|b| (a per-bin relative bound is meaningless for bins 60 dB below the
peak even between two fp32 FFT libraries); phase is compared through the complex value it reconstructs."""
import numpy as np
import pytest
import torch

import oracle
from oracle import preprocessor as opre

pytestmark = pytest.mark.gpu

GEOM = opre.Geometry()
FEAT_LIST = None


def _feat_list(P):
    return [
        {'feat_type': 'mel', 'channel': 0, 'log': True, 'delta': 1, 'cmvn': True},      # pretrain_sample.yaml:54-59
        {'feat_type': 'mel', 'channel': 0, 'log': True, 'delta': 2, 'cmvn': False},     # pseudo_noise.yaml:11-15
        P.get_feat_config('linear', 0), P.get_feat_config('phase', 0),
        P.get_feat_config('linear', 1), P.get_feat_config('phase', 1),
    ]


def _relmax(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).abs().flatten(1).max(dim=1).values / b.abs().flatten(1).max(dim=1).values).max().item()


@pytest.fixture(scope='module')
def P(gpu):
    from speech_enhancement_by_s3prl_amd.preprocessor import OnlinePreprocessor
    p = OnlinePreprocessor(sample_rate=16000, win_ms=25, hop_ms=10, n_freq=201, n_mels=40, n_mfcc=13,
                           roots=['x'], max_time=10000, target_level=-25, noise_proportion=0.5, snrs=[3, 6])
    p.feat_list = _feat_list(p)
    return p.to(gpu)


@pytest.mark.parametrize('T', [16000, 160000, 8123, 401])
def test_stft_features_vs_oracle(P, gpu, T):
    from speech_enhancement_by_s3prl_amd import synth
    torch.manual_seed(T)
    if T == 160000:
        _, wavs = synth.synth_batch(2, T)
    else:
        wavs = torch.randn(3, 3, T) * 0.1
    feats = P(wavs.to(gpu))
    ref = opre.forward(wavs, _feat_list(P), GEOM)
    F = T // 160 + 1
    assert [tuple(f.shape) for f in feats] == [tuple(r.shape) for r in ref]
    assert feats[2].shape == (wavs.shape[0], F, 201)
    # power (A1+A2)
    assert _relmax(feats[2], ref[2]) < 1e-4
    assert _relmax(feats[4], ref[4]) < 1e-4
    # phase through the complex value: sqrt(P) e^{i phi}
    for lin, ph, rlin, rph in ((feats[2], feats[3], ref[2], ref[3]), (feats[4], feats[5], ref[4], ref[5])):
        z = torch.polar(lin.cpu().double().sqrt(), ph.cpu().double())
        rz = torch.polar(rlin.double().sqrt(), rph.double())
        assert ((z - rz).abs().flatten(1).max(dim=1).values / rz.abs().flatten(1).max(dim=1).values).max().item() < 1e-4
    # log-mel + deltas (A3+A4): absolute tolerance on log values (log of a sum of >= 2 power bins)
    assert (feats[1].cpu() - ref[1]).abs().max().item() < 2e-3
    # CMVN'd features: unit-variance scale
    assert (feats[0].cpu() - ref[0]).abs().max().item() < 5e-3
    assert torch.isfinite(feats[0]).all()


@pytest.mark.parametrize('T,B', [(16000, 2), (160000, 1), (8123, 3)])
def test_mfcc_vs_oracle(P, gpu, T, B):
    """row A5 (`feat_type: mfcc`, pretrain_sample.yaml:50-53): own 128-filter mel bank -> log(. + 1e-6) -> orthonormal DCT-II, then the
    generic delta / CMVN stages (39 = 13 x 3 dims, pretrain_sample.yaml:2).  Oracle: oracle/preprocessor.py `mfcc` -- a restatement
    of torchaudio-0.6 MFCC(log_mels=True) as S3PRL is recalled to build it: PARITY UNPINNED (S3PRL / torchaudio absent)."""
    torch.manual_seed(T)
    wavs = torch.randn(B, 2, T) * 0.1
    fl = [P.get_feat_config('mfcc', 1), P.get_feat_config('mfcc', 0, delta=2, cmvn=True), P.get_feat_config('linear', 0)]
    feats = P(wavs.to(gpu), fl)
    ref = opre.forward(wavs, fl, GEOM)
    F = T // 160 + 1
    assert feats[0].shape == (B, F, 13) and feats[1].shape == (B, F, 39)
    assert [tuple(f.shape) for f in feats] == [tuple(r.shape) for r in ref]
    # raw cepstra: sums of 128 log-mel values (each within ~1e-3 absolute of the oracle's, cf. the log-mel bound) times |dct| <= 0.125
    scale = ref[0].abs().max().item()
    assert (feats[0].cpu() - ref[0]).abs().max().item() < 2e-4 * scale
    assert (feats[1].cpu() - ref[1]).abs().max().item() < 5e-3                  # CMVN'd: unit-variance scale
    assert _relmax(feats[2], ref[2]) < 1e-4


def test_mfcc_zero_arg_dims(gpu):
    """the reference's dimension probe (run_downstream.py:163) on an mfcc list: 39 = 13 x (1 + delta 2)"""
    from speech_enhancement_by_s3prl_amd.preprocessor import OnlinePreprocessor
    p = OnlinePreprocessor(sample_rate=16000, win_ms=25, hop_ms=10, n_freq=201, n_mels=40, n_mfcc=13,
                           feat_list=[OnlinePreprocessor.get_feat_config('mfcc', 0, delta=2, cmvn=True)])
    f, = p()
    assert f.shape[-1] == 39 and f.shape[-2] == 101


def test_stft_vs_float64_dft(P, gpu):
    """second opinion: independent float64 DFT-matrix STFT (oracle/dft64.py)"""
    torch.manual_seed(5)
    wav = torch.randn(1, 1, 4000) * 0.3
    lin, = P(wav.to(gpu), [P.get_feat_config('linear', 0)])
    spec = oracle.dft64.stft64(wav[0, 0].numpy())
    ref = torch.from_numpy(np.abs(spec) ** 2).T
    assert _relmax(lin.cpu()[0:1], ref[None]) < 1e-5


def test_pure_tone_known_answer(P, gpu):
    """first-principles: a tone at bin k concentrates power at k (Hann main lobe k-1..k+1)."""
    k = 37
    t = torch.arange(16000, dtype=torch.float64)
    wav = torch.cos(2 * np.pi * k * t / 400.0).float().view(1, 1, -1)
    lin, = P(wav.to(gpu), [P.get_feat_config('linear', 0)])
    mid = lin[0, 10:90].cpu()
    assert (mid.argmax(dim=-1) == k).all()
    # periodic Hann: |X[k]| = N/4 * 2 = sum(w)/2 = 100 -> power 1e4
    assert torch.allclose(mid[:, k], torch.full_like(mid[:, k], 1.0e4), rtol=1e-4)
    outside = torch.cat([mid[:, :k - 2], mid[:, k + 3:]], dim=1)
    assert outside.max().item() < 1e-3


@pytest.mark.parametrize('T', [16000, 160000, 4640 * 3, 480])
def test_istft_roundtrip_and_oracle(P, gpu, T):
    torch.manual_seed(T + 1)
    wavs = torch.randn(2, 1, T) * 0.1
    lin, ph = P(wavs.to(gpu), [P.get_feat_config('linear', 0), P.get_feat_config('phase', 0)])
    wav = P.istft(lin, ph)
    n_out = 160 * (T // 160)
    assert wav.shape == (2, n_out)
    assert _relmax(wav.cpu(), wavs[:, 0, :n_out]) < 1e-4            # iSTFT(STFT(x)) = x
    # vs oracle on a MODIFIED spectrum (mask), the enhancement use case
    mask = torch.rand(2, lin.shape[1], 201)
    ref = opre.istft(lin.cpu() * mask, ph.cpu(), GEOM)
    got = P.istft(lin * mask.to(gpu), ph)
    assert _relmax(got.cpu(), ref) < 1e-4


def test_zero_arg_dim_discovery(P):
    """run_downstream.py:163-164: preprocessor() on the internal pseudo wav returns the six features."""
    feats = P()
    assert [f.shape[-1] for f in feats] == [80, 120, 201, 201, 201, 201]
    assert feats[0].shape[:2] == (1, 101)


def test_stft_attr_surface(P, gpu):
    """sampler.py:226-228: _stft(wav2d, window=) -> (..., K, F, 2), _magphase -> (power, phase)."""
    wav = torch.randn(2, 3200, device=gpu)
    c = P._stft(wav, window=P._window)
    assert c.shape == (2, 201, 21, 2)
    lin, ph = P._magphase(c)
    ref = opre.magphase(opre.stft(wav.cpu(), GEOM))
    assert _relmax(lin.cpu().transpose(1, 2), ref[0].transpose(1, 2)) < 1e-4


# ---- round 3: the encoded-phase form of the STFT / iSTFT (stft2.hip, istft2.hip) and the on-demand phase of the boundary ----------------------

@pytest.mark.parametrize('T,B,C', [(160000, 2, 3), (160000, 3, 2), (16000, 5, 2), (8123, 3, 3), (401, 2, 2), (4800, 33, 2), (12346, 2, 3)])
def test_stft_phasor_vs_oracle(P, gpu, T, B, C):
    """se_stft_tphase_f32 through the drop-in: power, encoded phase (decoded and compared through the complex spectrum it reconstructs),
    raw mel via the log-mel features; chunk sizes / persistent loop exercised by batch sizes above and below the resident workgroup count."""
    from speech_enhancement_by_s3prl_amd.preprocessor import LazyPhase
    torch.manual_seed(T + B)
    wavs = torch.randn(B, C, T) * 0.1
    wavs[0, 0, : T // 3] = 0.0                     # a silent stretch: X == 0 -> phasor (1, 0) (the reference's atan2(0, 0) = 0)
    P.channel_inp, P.channel_tar = 0, 1
    feats = P(wavs.to(gpu))
    ref = opre.forward(wavs, _feat_list(P), GEOM)
    assert type(feats[3]) is LazyPhase and type(feats[5]) is LazyPhase
    assert feats[3]._tphase is not None and feats[5]._tphase is None          # the encoded phase is kept for the input channel only
    assert feats[3].shape == ref[3].shape and feats[3].device == feats[2].device and feats[3].dtype == torch.float32
    assert _relmax(feats[2], ref[2]) < 1e-4 and _relmax(feats[4], ref[4]) < 1e-4
    # decode as istft2.hip does: t = the word as fp32, bit 0 = (cos < 0): (cos, sin) = (+-(1 - t^2), 2 t) / (1 + t^2)
    word = feats[3]._tphase.cpu()
    t = word.view(torch.float32).double()
    sgn = 1.0 - 2.0 * (word & 1).double()
    ph = torch.stack([sgn * (1 - t * t) / (1 + t * t), 2 * t / (1 + t * t)], dim=-1)
    assert t.abs().max().item() <= 1.0 + 1e-6
    z = torch.view_as_complex(ph.contiguous()) * feats[2].cpu().double().sqrt()
    rz = torch.polar(ref[2].double().sqrt(), ref[3].double())
    assert ((z - rz).abs().flatten(1).max(dim=1).values / rz.abs().flatten(1).max(dim=1).values).max().item() < 1e-4
    silent = ref[2][0] == 0
    if silent.any():
        assert (word[0][silent] == 0).all()                                    # X == 0 -> (cos, sin) = (1, 0)
    assert (feats[1].cpu() - ref[1]).abs().max().item() < 2e-3
    assert (feats[0].cpu() - ref[0]).abs().max().item() < 5e-3
    # on-demand phase == what the eager atan2 path returns, bit for bit, for both channels
    P.lazy_phase = False
    try:
        eager = P(wavs.to(gpu))
    finally:
        P.lazy_phase = True
    assert type(eager[3]) is torch.Tensor
    assert torch.equal(feats[3].materialize(), eager[3]) and torch.equal(feats[5] + 0, eager[5])
    for a, b in zip(feats[:3] + [feats[4]], eager[:3] + [eager[4]]):
        assert _relmax(a, b) < 1e-5


def test_lazy_phase_behaves_like_a_tensor(P, gpu, tmp_path):
    """everything a caller may do with the boundary's `phase` output works and yields the atan2 phase: arithmetic, indexing, .cpu(),
    torch.stack, torch.save / load, repr; istft() on a materialised or modified phase takes the sin / cos kernel."""
    from speech_enhancement_by_s3prl_amd.preprocessor import LazyPhase
    torch.manual_seed(3)
    wavs = (torch.randn(2, 2, 16000) * 0.1).to(gpu)
    P.channel_inp, P.channel_tar = 0, 1
    f = P(wavs)
    P.lazy_phase = False
    try:
        e = P(wavs)
    finally:
        P.lazy_phase = True
    ph = f[3]
    assert type(ph) is LazyPhase and ph._value is None
    assert ph.shape == e[3].shape and ph.is_cuda and ph.dim() == 3 and ph.numel() == e[3].numel()
    assert ph._value is None                                   # metadata queries do not materialise
    wav_fast = P.istft(f[2], ph)                               # se_istft_tphase_f32
    assert ph._value is None
    assert torch.equal(ph[1, 5], e[3][1, 5]) and ph._value is not None
    assert torch.equal(ph.cpu(), e[3].cpu()) and torch.equal((ph * 2.0), e[3] * 2.0)
    assert torch.equal(torch.stack([ph, ph])[1], e[3]) and 'cuda' in repr(ph)
    torch.save({'phase': ph}, tmp_path / 'p.pt')
    assert torch.equal(torch.load(tmp_path / 'p.pt', weights_only=True)['phase'].to(gpu), e[3])
    wav_slow = P.istft(f[2], ph)                               # materialised by now: the power / phase kernel
    wav_ref = P.istft(f[2], e[3])
    assert torch.equal(wav_slow, wav_ref)
    assert ((wav_fast - wav_ref).abs().max() / wav_ref.abs().max()).item() < 2e-6
    assert ((wav_fast[:, :15840] - wavs[:, 0, :15840]).abs().max() / wavs[:, 0].abs().max()).item() < 1e-5     # round trip


@pytest.mark.parametrize('T,B', [(160000, 2), (160000, 33), (16000, 3), (8123, 2), (480, 1), (2560, 3)])
def test_istft_phasor_vs_oracle(P, gpu, T, B):
    """se_istft_tphase_f32 (enhanced power x the noisy channel's encoded phase) against torch.istft on (sqrt(power), oracle phase), with
    the fused masked square sum; ragged lengths; a padded output row."""
    from oracle import decode as odec
    torch.manual_seed(T * 7 + B)
    wavs = torch.randn(B, 2, T) * 0.1
    lengths = torch.randint(T // 2, T + 1, (B,))
    lengths[0] = T                                              # collate_fn pads to the longest utterance (dataset.py:169-179)
    P.channel_inp, P.channel_tar = 0, 1
    feats = P(wavs.to(gpu))
    ref = opre.forward(wavs, _feat_list(P), GEOM)
    pred = ref[2] * (0.25 + torch.rand_like(ref[2]))            # an "enhanced" power: random mask on the noisy power
    n_out = 160 * (T // 160)
    wav, sumsq = P.istft_with_sumsq(pred.to(gpu), feats[3], lengths=lengths.to(gpu), out_len=T)
    assert feats[3]._value is None and wav.shape == (B, max(T, n_out))
    rwav = opre.istft(pred, ref[3], GEOM)
    assert rwav.shape[1] == n_out
    assert ((wav[:, :n_out].cpu() - rwav).abs().flatten(1).max(dim=1).values / rwav.abs().flatten(1).max(dim=1).values).max().item() < 1e-4
    assert (wav[:, n_out:] == 0).all()
    mask = (torch.arange(n_out)[None] < lengths[:, None]).float()
    rs = (rwav.double().pow(2) * mask).sum(dim=1)
    assert ((sumsq.cpu().double() - rs).abs() / rs).max().item() < 1e-4
    # and the whole decode (D2) through the phasor path equals the oracle's
    dec = __import__('speech_enhancement_by_s3prl_amd.decode', fromlist=['decode_wav']).decode_wav(P, pred.to(gpu), feats[3], lengths.to(gpu), wavs[:, 1].to(gpu))
    rdec = odec.decode_wav(pred, ref[3], lengths, GEOM, wavs[:, 1])
    assert ((dec.cpu() - rdec).abs().max() / rdec.abs().max()).item() < 1e-4


def test_identical_feature_requests_are_computed_once(P, gpu):
    """`--upstream baseline` (run_downstream.py:134-135, 150-151): the reference's feature list then holds the baseline feature twice; the
    preprocessor computes it once and returns it for both entries (as two raw 'linear' requests already were one tensor).  Values equal
    the separately requested feature."""
    torch.manual_seed(5)
    wavs = torch.randn(2, 2, 16000, device=gpu) * 0.1
    base = {'feat_type': 'mel', 'channel': 0, 'log': True, 'delta': 2, 'cmvn': False}       # pseudo_noise.yaml:11-15
    twice = P(wavs, feat_list=[dict(base), dict(base), P.get_feat_config('linear', 0), P.get_feat_config('linear', 1)])
    once = P(wavs, feat_list=[dict(base)])
    assert twice[0] is twice[1]
    assert torch.equal(twice[0], once[0])
    assert twice[0].shape == (2, 101, 120)
    other = P(wavs, feat_list=[dict(base), dict(base, delta=1)])          # different requests stay different
    assert other[0] is not other[1] and other[1].shape == (2, 101, 80)


def test_unused_downstream_feature_costs_nothing(P, gpu):
    """the standard six-feature list (run_downstream.py:150-157) on the device: the first (upstream) feature is computed, a second derived
    feature -- feats_for_downstream, which the upstream + SpecHead pipelines never read -- is a LazyTensor with the right shape that holds no
    value until something uses it, and then equals the eagerly computed feature bit for bit"""
    from speech_enhancement_by_s3prl_amd.preprocessor import LazyTensor
    torch.manual_seed(8)
    wavs = torch.randn(2, 2, 16000, device=gpu) * 0.1
    feats = P(wavs)                                    # fixture list: mel(delta 1, cmvn), mel(delta 2), linear / phase of both channels
    assert type(feats[0]) is torch.Tensor and feats[0].shape == (2, 101, 80)
    assert type(feats[1]) is LazyTensor and feats[1]._value is None and feats[1].shape == (2, 101, 120)
    P.lazy_features = False
    try:
        eager = P(wavs)
    finally:
        P.lazy_features = True
    assert type(eager[1]) is torch.Tensor
    assert feats[1]._value is None                     # still nothing computed
    assert torch.equal(feats[1] * 1.0, eager[1])       # first use materialises it
    assert feats[1]._value is not None and torch.equal(feats[0], eager[0])


def test_lazy_phase_is_tied_to_the_input_buffer(P, gpu):
    """the lazy phase recomputes atan2 from the caller's waveform buffer: an in-place refill between preprocessor(wavs) and the first read of the
    phase must be an ERROR (it would be the phase of another batch next to this call's `linear` planes), not a silent wrong answer; a phase
    read BEFORE the refill, the istft() fast path (which never touches the waveforms again) and the eager mode are unaffected"""
    torch.manual_seed(11)
    wavs = (torch.randn(2, 2, 16000) * 0.1).to(gpu)
    nxt = (torch.randn(2, 2, 16000) * 0.1).to(gpu)
    P.channel_inp, P.channel_tar = 0, 1
    f = P(wavs)
    early = f[5] + 0                                   # read before the refill: fine
    wav_fast = P.istft(f[2], f[3])                     # the encoded phase word: independent of the buffer
    wavs.copy_(nxt)
    with pytest.raises(RuntimeError, match='modified in place'):
        f[3].materialize()
    with pytest.raises(RuntimeError, match='modified in place'):
        f[3] + 0
    assert torch.equal(P.istft(f[2], f[3]), wav_fast)
    assert torch.isfinite(early).all()
    P.lazy_phase = False
    try:
        e = P(wavs)
        wavs.zero_()
        assert torch.isfinite(e[3]).all() and e[3].abs().max().item() > 1.0      # eager phase: a plain tensor, self-contained
    finally:
        P.lazy_phase = True


def test_preprocessor_under_inference_mode(P, gpu):
    """tensors created under torch.inference_mode() track no version counter (reading `_version` raises): the fused STFT path must work there too
    (it raised for every caller that evaluates under inference_mode) -- the lazy phase is tied to a private copy of the batch instead,
    so an in-place refill of the caller's buffer cannot change it"""
    torch.manual_seed(12)
    P.channel_inp, P.channel_tar = 0, 1
    host = torch.randn(2, 2, 16000) * 0.1
    ref = P(host.to(gpu))
    ref_phase = ref[3] + 0
    with torch.inference_mode():
        wavs = host.to(gpu)
        assert wavs.is_inference()
        f = P(wavs)
        wavs.zero_()                                   # refill after the call: the lazy phase must still be this batch's
        ph = f[3] + 0
        wav = P.istft(f[2], f[3])
    assert torch.equal(f[2], ref[2])
    assert torch.equal(ph, ref_phase)
    assert torch.equal(wav, P.istft(ref[2], ref[3]))
