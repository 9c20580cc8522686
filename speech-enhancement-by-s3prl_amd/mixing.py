"""On-GPU mixing: the arithmetic of OnlineDataset.__getitem__ + collate_fn
(dataset.py:54-74,106-111,141-179) for a whole batch in two launches of libse_amd.so (se_mix_f32), replacing the
reference's 12 CPU DataLoader workers once the utterances are decoded.  No CPU fallback (synth.py holds the host-side
restatement the tests compare against)."""
import torch

from . import _lib


def mix_batch(speech, speech_lengths, noise, noise_lengths, snrs, target_level=-25, eps=1e-8, normalize=True, noise_offsets=None,
              max_len=None):
    """speech (B, Ts), noise (B, Tn) fp32 device tensors (rows right-padded), lengths (B,) int64, snrs (B,) in dB.
    Returns (lengths (B,) int64, wavs (B, 3, T)) = collate_fn's contract with channels (noisy, clean, scaled noise).
    `noise_offsets` selects a sub-range of each noise row (half_noise: 'front' = offset 0 / length n//2, 'end' = offset
    n//2 / length n - n//2, dataset.py:147-152)."""
    if not speech.is_cuda:
        raise _lib.SEError('mix_batch runs on MI355X only (no CPU fallback)')
    lib = _lib.load()
    dev = speech.device
    speech, noise = speech.contiguous().float(), noise.contiguous().float()
    B = speech.shape[0]
    ls = speech_lengths.to(dev, torch.int64).contiguous()
    ln = noise_lengths.to(dev, torch.int64).contiguous()
    off = None if noise_offsets is None else noise_offsets.to(dev, torch.int64).contiguous()
    snr = snrs.to(dev, torch.float32).contiguous()
    T = int(max_len) if max_len is not None else int(speech.shape[1])
    wavs = torch.empty(B, 3, T, device=dev, dtype=torch.float32)
    sums = torch.empty(3 * B, device=dev, dtype=torch.float64)
    with torch.cuda.device(dev):
        _lib.check(lib.se_mix_f32(_lib.ptr(speech), speech.shape[1], _lib.ptr(ls), _lib.ptr(noise), noise.shape[1], _lib.ptr(ln), _lib.ptr(off),
                                  _lib.ptr(snr), B, T, int(bool(normalize)), float(target_level), float(eps), _lib.ptr(wavs), _lib.ptr(sums),
                                  _lib.stream()), 'se_mix_f32')
    return ls, wavs
