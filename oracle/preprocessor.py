"""Oracle rows A1-A6: S3PRL ``OnlinePreprocessor`` restated on PyTorch-CPU (test infrastructure only).

The original lives in the un-vendored S3PRL package (``utility/preprocessor.py``); the reference uses it at
run_downstream.py:153-164, runner.py:267,433,558 and sampler.py:60,226-228.  What the call sites force:
201 bins / hop 160 / F = len//hop + 1 => center=True (runner.py:455); ``linear`` is POWER
(sampler.py:229 takes sqrt of _magphase, objective.py:89); outputs are time-major (B, T', D)
(model.py:30, objective.py:110-111).  Mel / delta / CMVN follow torchaudio 0.6 (MelScale HTK, no norm;
compute_deltas win 5 replicate) as recalled -- PARITY UNPINNED vs original S3PRL for those (A3, A4).
The MFCC branch (A5) restates torchaudio-0.6 `transforms.MFCC(sample_rate, n_mfcc, log_mels=True, melkwargs=win_args)` as S3PRL
is recalled to construct it: its OWN mel spectrogram on the same STFT geometry with torchaudio's default 128 filters, log(mel + 1e-6),
orthonormal DCT-II -- PARITY UNPINNED (neither S3PRL nor torchaudio is in the container; no shipped config selects `mfcc`).
"""
import math

import torch
import torch.nn.functional as F


class Geometry:
    """STFT geometry from config/pretrain_sample.yaml:39-48 (``online`` section)."""

    def __init__(self, sample_rate=16000, win_ms=25, hop_ms=10, n_freq=201, n_mels=40, n_mfcc=13, eps=1e-10, **_):
        self.sample_rate = sample_rate
        self.win = round(win_ms * sample_rate / 1000)
        self.hop = round(hop_ms * sample_rate / 1000)
        self.n_fft = (n_freq - 1) * 2
        self.n_freq = n_freq
        self.n_mels = n_mels
        self.n_mfcc = n_mfcc
        self.eps = eps


def hann_window(geom, dtype=torch.float32):
    # S3PRL registers torch.hann_window(win) (periodic) as `_window` (used at sampler.py:226)
    return torch.hann_window(geom.win, dtype=dtype)


def stft(wav2d, geom, window=None):
    """A1: torch.stft(center=True, reflect, onesided, not normalized). wav2d (N, T) -> complex (N, K, F)."""
    window = hann_window(geom, wav2d.dtype) if window is None else window
    return torch.stft(wav2d, n_fft=geom.n_fft, hop_length=geom.hop, win_length=geom.win, window=window,
                      center=True, pad_mode='reflect', normalized=False, onesided=True, return_complex=True)


def magphase(complx):
    """A2: torchaudio.functional.magphase(power=2): power = re^2+im^2, phase = atan2(im, re)."""
    power = complx.real.pow(2) + complx.imag.pow(2)
    phase = torch.atan2(complx.imag, complx.real)
    return power, phase


def mel_filterbank(geom, dtype=torch.float32):
    """A3: torchaudio-0.6 create_fb_matrix(n_freqs, f_min=0, f_max=sr//2, n_mels, sr): HTK mel, no area norm.
    Returns fb (n_freq, n_mels).  Built in float64 then cast, so the HIP plan and the oracle share one table."""
    n_freqs, n_mels, sr = geom.n_freq, geom.n_mels, geom.sample_rate
    f_min, f_max = 0.0, float(sr // 2)
    all_freqs = torch.linspace(0, sr // 2, n_freqs, dtype=torch.float64)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2, dtype=torch.float64)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)          # (n_freqs, n_mels+2)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.clamp(torch.min(down, up), min=0.0)
    return fb.to(dtype)


def melscale(power, geom):
    """power (..., K, F) -> mel (..., n_mels, F)   (MelScale: fb^T . power)."""
    fb = mel_filterbank(geom, power.dtype)
    return torch.matmul(power.transpose(-1, -2), fb).transpose(-1, -2)


MFCC_N_MELS = 128          # torchaudio MelSpectrogram default, not the preprocessor's n_mels
MFCC_LOG_OFFSET = 1e-6     # torchaudio MFCC(log_mels=True)


def dct_matrix(n_mfcc, n_mels, dtype=torch.float32):
    """torchaudio-0.6 functional.create_dct(n_mfcc, n_mels, norm='ortho'): (n_mels, n_mfcc), DCT-II,
    dct[n, k] = cos(pi / n_mels * (n + 0.5) * k) * sqrt(2 / n_mels), column 0 additionally * 1 / sqrt(2)."""
    n = torch.arange(n_mels, dtype=torch.float64)
    k = torch.arange(n_mfcc, dtype=torch.float64).unsqueeze(1)
    dct = torch.cos(math.pi / n_mels * (n + 0.5) * k)            # (n_mfcc, n_mels)
    dct[0] *= 1.0 / math.sqrt(2.0)
    dct *= math.sqrt(2.0 / n_mels)
    return dct.t().contiguous().to(dtype)


def mfcc(power, geom, n_mfcc=13):
    """A5: power (..., K, F) -> mfcc (..., n_mfcc, F): 128-filter HTK mel of the power spectrogram, log(. + 1e-6), DCT-II (ortho)."""
    g128 = Geometry.__new__(Geometry)
    g128.__dict__.update(geom.__dict__)
    g128.n_mels = MFCC_N_MELS
    mel = melscale(power, g128)
    logmel = (mel + MFCC_LOG_OFFSET).log()
    return torch.matmul(logmel.transpose(-1, -2), dct_matrix(n_mfcc, MFCC_N_MELS, power.dtype)).transpose(-1, -2)


def compute_deltas(x, win_length=5):
    """A4: torchaudio.functional.compute_deltas along the last (time) dim, replicate padding.
    kernel = [-2,-1,0,1,2] / 10."""
    n = (win_length - 1) // 2
    denom = n * (n + 1) * (2 * n + 1) / 3
    shape = x.shape
    flat = x.reshape(1, -1, shape[-1])
    flat = F.pad(flat, (n, n), mode='replicate')
    kernel = torch.arange(-n, n + 1, dtype=x.dtype).repeat(flat.shape[1], 1, 1)
    out = F.conv1d(flat, kernel, groups=flat.shape[1]) / denom
    return out.reshape(shape)


def get_feat_config(feat_type, channel=0, log=False, delta=0, cmvn=False):
    # OnlinePreprocessor.get_feat_config, used at run_downstream.py:153-156, runner.py:50
    assert feat_type in ('complx', 'linear', 'phase', 'mel', 'mfcc')
    return {'feat_type': feat_type, 'channel': channel, 'log': log, 'delta': delta, 'cmvn': cmvn}


def select_feat(raw, eps, log=False, delta=0, cmvn=False):
    """A4: raw (B, D, F) feature-major -> (B, D*(1+delta), F): log(x+eps), stacked deltas-of-deltas,
    CMVN over time with the UNBIASED std and `+eps` outside the sqrt."""
    if bool(log):
        raw = (raw + eps).log()
    feats = [raw.contiguous()]
    for _ in range(int(delta)):
        feats.append(compute_deltas(feats[-1]))
    feats = torch.cat(feats, dim=-2)
    if bool(cmvn):
        feats = (feats - feats.mean(dim=-1, keepdim=True)) / (feats.std(dim=-1, keepdim=True) + eps)
    return feats


def forward(wavs, feat_list, geom):
    """OnlinePreprocessor.forward: wavs (B, C, T) -> list of time-major (B, T', D) features.
    The MFCC branch (A5) is computed-and-discarded in S3PRL on every call; here it is only produced if requested."""
    shape = wavs.shape
    complx = stft(wavs.reshape(-1, shape[-1]), geom)
    complx = complx.reshape(shape[:-1] + complx.shape[-2:])      # (B, C, K, F)
    linear, phase = magphase(complx)
    outs = []
    for args in feat_list:
        ft = args['feat_type']
        ch = int(args.get('channel', 0))
        if ft == 'linear':
            raw = linear[:, ch]
        elif ft == 'phase':
            raw = phase[:, ch]
        elif ft == 'mel':
            raw = melscale(linear[:, ch], geom)
        elif ft == 'mfcc':
            raw = mfcc(linear[:, ch], geom, getattr(geom, 'n_mfcc', 13))
        elif ft == 'complx':
            c = complx[:, ch]
            raw = torch.stack([c.real, c.imag], dim=-1).transpose(-1, -2).reshape(c.shape[0], -1, c.shape[-1])
        else:
            raise NotImplementedError(ft)
        feat = select_feat(raw, geom.eps, args.get('log', False), args.get('delta', 0), args.get('cmvn', False))
        outs.append(feat.transpose(-1, -2).contiguous())
    return outs


def istft(linears, phases, geom, linear_power=2):
    """A6: OnlinePreprocessor.istft(linears, phases) (runner.py:267): time-major (B, T', K) power + phase
    -> wav (B, hop*(T'-1)).  mag = linear^(1/2); complex = mag*(cos, sin); torch.istft with the same window."""
    lin = linears.transpose(-1, -2)
    ph = phases.transpose(-1, -2)
    mag = lin.pow(1.0 / linear_power)
    complx = torch.complex(mag * ph.cos(), mag * ph.sin())
    window = hann_window(geom, linears.dtype)
    return torch.istft(complx, n_fft=geom.n_fft, hop_length=geom.hop, win_length=geom.win, window=window,
                       center=True, normalized=False, onesided=True)
