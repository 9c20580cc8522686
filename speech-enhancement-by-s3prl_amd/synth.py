"""Synthetic input generator: no audio exists offline, so benchmarks and parity tests
use seeded harmonic "speech" + white noise mixed by the reference's formula.  Restates the INPUT CONTRACT of
dataset.py: add_noise (dataset.py:54-74), normalize_wav_decibel (dataset.py:106-111), the 3-channel stack
(noisy, clean, scaled_noise) (dataset.py:161) and collate_fn's (lengths (B,) int64, wavs (B, C, T) fp32)
(dataset.py:169-179).  Host-side torch (CPU DataLoader territory in the reference), not a kernel."""
import math

import torch
from torch.nn.utils.rnn import pad_sequence

SNRS = [-8, -6, -4, -2, 0, 2, 4, 6, 8]          # pseudo_noise.yaml:73


def normalize_wav_decibel(audio, target_level=-25):
    rms = audio.pow(2).mean().pow(0.5)
    scalar = (10 ** (target_level / 20)) / (rms + 1e-10)
    return audio * scalar


def add_noise(speech, noise, snrs, eps=1e-10):
    """speech, noise: (1, T); snrs: (1,)  -- the reference only works at batch 1 (its (B,)x(B,1) broadcast)."""
    if speech.size(-1) >= noise.size(-1):
        times = speech.size(-1) // noise.size(-1)
        remainder = speech.size(-1) % noise.size(-1)
        noise_expanded = noise.unsqueeze(-2).expand(-1, times, -1).reshape(speech.size(0), -1)
        noise = torch.cat([noise_expanded, noise[:, :remainder]], dim=-1)
    else:
        noise = noise[:, :speech.size(-1)]
    snr_exp = 10.0 ** (snrs / 10.0)
    speech_power = speech.pow(2).sum(dim=-1, keepdim=True)
    noise_power = noise.pow(2).sum(dim=-1, keepdim=True)
    scalar = (speech_power / (snr_exp * noise_power + eps)).pow(0.5)
    scaled_noise = scalar * noise
    return speech + scaled_noise, scaled_noise


def collate_fn(samples):
    """samples: list of (T_i, C) -> lengths (B,) int64, wavs (B, C, T_max)."""
    lengths = torch.LongTensor([len(s) for s in samples])
    wavs = pad_sequence(samples, batch_first=True).transpose(-1, -2).contiguous()
    return lengths, wavs


def synth_utterance(i, n_samples=160000, sample_rate=16000, target_level=-25):
    """Utterance i: 5 harmonics of f0 in U[100,300] Hz with a 4 Hz raised-cosine envelope + 0.01 N(0,1),
    white noise, both at -25 dBFS, SNR from SNRS, mixed with add_noise. Returns (T, 3) = (noisy, clean, noise)."""
    g = torch.Generator().manual_seed(1337 + i)          # 1337 = the reference's default seed (run_downstream.py:62)
    f0 = 100.0 + 200.0 * torch.rand(1, generator=g).item()
    t = torch.arange(n_samples, dtype=torch.float64) / sample_rate
    speech = torch.zeros(n_samples, dtype=torch.float64)
    for h in range(1, 6):
        speech += torch.sin(2 * math.pi * f0 * h * t + 0.3 * h) / h
    env = 0.5 - 0.5 * torch.cos(2 * math.pi * 4.0 * t)
    speech = (speech * env).float() + 0.01 * torch.randn(n_samples, generator=g)
    speech = normalize_wav_decibel(speech, target_level)
    noise = normalize_wav_decibel(torch.randn(n_samples, generator=g), target_level)
    snr = SNRS[int(torch.randint(len(SNRS), (1,), generator=g).item())]
    noisy, scaled = add_noise(speech.unsqueeze(0), noise.unsqueeze(0), torch.ones(1) * snr)
    return torch.stack([noisy.squeeze(0), speech, scaled.squeeze(0)], dim=-1)


def synth_batch(batch, n_samples=160000, first=0, ragged=False):
    """(lengths (B,), wavs (B, 3, T)). ragged: lengths in U[n/2, n] (seeded), zero padded as collate_fn does."""
    samples = []
    for i in range(first, first + batch):
        s = synth_utterance(i, n_samples)
        if ragged:
            g = torch.Generator().manual_seed(4242 + i)
            n = int(torch.randint(n_samples // 2, n_samples + 1, (1,), generator=g).item())
            s = s[:n]
        samples.append(s)
    return collate_fn(samples)


def fast_batch(batch, n_samples=160000, seed=0, device='cpu'):
    """Cheap large-batch generator for throughput runs (same statistics, generated on `device`)."""
    g = torch.Generator(device=device).manual_seed(seed)
    t = torch.arange(n_samples, device=device, dtype=torch.float32) / 16000.0
    f0 = 100.0 + 200.0 * torch.rand(batch, 1, device=device, generator=g)
    speech = torch.zeros(batch, n_samples, device=device)
    for h in range(1, 6):
        speech += torch.sin(2 * math.pi * f0 * h * t + 0.3 * h) / h
    speech = speech * (0.5 - 0.5 * torch.cos(2 * math.pi * 4.0 * t)) + 0.01 * torch.randn(batch, n_samples, device=device, generator=g)
    lvl = 10 ** (-25 / 20)
    speech = speech * (lvl / (speech.pow(2).mean(-1, keepdim=True).sqrt() + 1e-10))
    noise = torch.randn(batch, n_samples, device=device, generator=g)
    noise = noise * (lvl / (noise.pow(2).mean(-1, keepdim=True).sqrt() + 1e-10))
    snr = torch.tensor(SNRS, device=device, dtype=torch.float32)[torch.randint(len(SNRS), (batch,), device=device, generator=g)]
    scalar = (speech.pow(2).sum(-1, keepdim=True) / (10.0 ** (snr[:, None] / 10.0) * noise.pow(2).sum(-1, keepdim=True) + 1e-10)).sqrt()
    scaled = scalar * noise
    wavs = torch.stack([speech + scaled, speech, scaled], dim=1).contiguous()
    lengths = torch.full((batch,), n_samples, dtype=torch.int64, device=device)
    return lengths, wavs
