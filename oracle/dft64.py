"""Independent float64 DFT-matrix STFT / iSTFT (numpy) -- a second opinion on torch.stft / torch.istft
for oracle rows A1/A6 (test infrastructure only).  No FFT library is used: the transform is the
definition X[k] = sum_n w[n] x[n] exp(-2 pi i k n / N)."""
import numpy as np


def hann_periodic(n):
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def stft64(wav, n_fft=400, hop=160):
    """wav (T,) -> complex128 (n_fft//2+1, T//hop+1); center=True, reflect padding, periodic Hann."""
    wav = np.asarray(wav, dtype=np.float64)
    pad = n_fft // 2
    x = np.pad(wav, (pad, pad), mode='reflect')
    n_frames = len(wav) // hop + 1
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    frames = x[idx] * hann_periodic(n_fft)[None, :]
    k = np.arange(n_fft // 2 + 1)
    basis = np.exp(-2j * np.pi * np.outer(np.arange(n_fft), k) / n_fft)
    return (frames @ basis).T


def istft64(spec, n_fft=400, hop=160):
    """complex (K, F) -> wav (hop*(F-1),): inverse DFT by definition, window, overlap-add, / sum w^2, trim n_fft/2."""
    spec = np.asarray(spec, dtype=np.complex128)
    K, F = spec.shape
    full = np.concatenate([spec, np.conj(spec[-2:0:-1])], axis=0)            # Hermitian extension (N, F)
    n = np.arange(n_fft)
    basis = np.exp(2j * np.pi * np.outer(n, n) / n_fft) / n_fft
    frames = (basis @ full).real.T                                            # (F, N)
    w = hann_periodic(n_fft)
    total = hop * (F - 1) + n_fft
    out = np.zeros(total)
    env = np.zeros(total)
    for f in range(F):
        out[f * hop:f * hop + n_fft] += frames[f] * w
        env[f * hop:f * hop + n_fft] += w * w
    pad = n_fft // 2
    return out[pad:total - pad] / env[pad:total - pad]
