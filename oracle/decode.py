"""Oracle rows D1-D2 (test infrastructure only); PINNED against the reference's utils.py / runner.py
via tests/golden/reference_golden.npz."""
import torch

from . import preprocessor as _pre


def get_length_masks(lengths, max_len=None):
    """D1: runner.py:216-220  arange[:max(lengths)] < lengths[:,None]  as int64."""
    max_len = int(lengths.max().item()) if max_len is None else max_len
    return (torch.arange(max_len)[None, :] < lengths[:, None]).long()


def masked_mean(batch, length_masks, keepdim=False, eps=1e-8):
    """utils.py:26-29"""
    return (batch * length_masks).sum(dim=-1, keepdim=keepdim) / (length_masks.sum(dim=-1, keepdim=keepdim) + eps)


def masked_normalize_decibel(audio, target, length_masks, eps=1e-8):
    """D2: utils.py:31-46.  target: a number (fixed dB) or reference audio (B, T) whose masked power sets the level."""
    if isinstance(target, (int, float)):
        target = torch.ones(len(audio), dtype=audio.dtype) * target
    elif isinstance(target, torch.Tensor) and target.dim() > 1:
        target = 10.0 * torch.log10(masked_mean(target.pow(2), length_masks, keepdim=False))
    assert target.dim() == 1
    scalar_square = (10.0 ** (target.unsqueeze(-1) / 10.0)) / (masked_mean(audio.pow(2), length_masks, keepdim=True) + eps)
    return audio * scalar_square.pow(0.5)


def decode_wav(linear, phase, lengths, geom, target_level=-25):
    """D2: runner.py:266-270  istft -> right-pad zeros to max(lengths) -> masked dB-normalise.
    NB evaluate() passes wav_tar as the 4th positional arg, i.e. normalises to the CLEAN wav's level (runner.py:570)."""
    wav = _pre.istft(linear, phase, geom)
    pad = int(max(lengths)) - wav.size(1)
    wav = torch.cat([wav, wav.new_zeros(wav.size(0), pad)], dim=1)
    return masked_normalize_decibel(wav, target_level, get_length_masks(lengths))
