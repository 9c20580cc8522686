"""Rows D1-D2: length masks, masked dB normalisation and Runner._decode_wav (runner.py:216-220,266-270;
utils.py:26-46) on the gfx950 kernels.  The iSTFT launch already accumulates the masked sum of squares,
so decode_wav is: se_istft_f32 (+ se_masked_sumsq_f32 of the reference wav) + se_dbnorm_f32."""
import torch

from . import _lib


def get_length_masks(lengths, max_len=None):
    """D1: (B,) int64 device tensor -> (B, max_len) int64 0/1 mask. `max_len` defaults to lengths.max()
    (one host sync, as runner.py:218); pass it explicitly to stay asynchronous."""
    lib = _lib.load()
    lengths = lengths.to(torch.int64).contiguous()
    if max_len is None:
        max_len = int(lengths.max().item())
    masks = torch.empty(lengths.shape[0], max_len, device=lengths.device, dtype=torch.int64)
    _lib.check(lib.se_length_masks_i64(_lib.ptr(lengths), lengths.shape[0], max_len, _lib.ptr(masks), _lib.stream()),
               'se_length_masks_i64')
    return masks


def masked_sumsq(x, lengths):
    lib = _lib.load()
    x = x.contiguous().float()
    B, T = x.shape
    lengths = lengths.to(device=x.device, dtype=torch.int64).contiguous()
    sums = torch.empty(B, device=x.device, dtype=torch.float32)
    _lib.check(lib.se_masked_sumsq_f32(_lib.ptr(x), B, T, T, _lib.ptr(lengths), _lib.ptr(sums), _lib.stream()), 'se_masked_sumsq_f32')
    return sums


def masked_normalize_decibel(audio, target, lengths, eps=1e-8, audio_sumsq=None, inplace=False, ref_sumsq=None):
    """utils.py:31-46 with the mask given as lengths (the reference builds it from lengths, runner.py:269).
    target: number (fixed dB) or reference audio (B, T)."""
    lib = _lib.load()
    audio = audio.contiguous().float()
    if not inplace:
        audio = audio.clone()
    B, T = audio.shape
    lengths = lengths.to(device=audio.device, dtype=torch.int64).contiguous()
    if audio_sumsq is None:
        audio_sumsq = masked_sumsq(audio, lengths)
    fixed = 0.0
    if isinstance(target, (int, float)):
        fixed, ref_sumsq = float(target), None
    elif isinstance(target, torch.Tensor) and target.dim() > 1:
        if ref_sumsq is None:          # else: already summed by the iSTFT launch (preprocessor._istft_tphase)
            ref = target
            if ref.dtype != torch.float32 or ref.stride(1) != 1:
                ref = ref.contiguous().float()
            if ref.shape[1] < T:
                raise _lib.SEError('reference audio shorter than the audio to normalise')
            if not ref.is_cuda:
                raise _lib.SEError('reference audio must live on the GPU (no CPU fallback)')
            ref_sumsq = torch.empty(B, device=audio.device, dtype=torch.float32)
            # rows may be strided (wavs[:, channel_tar] of a (B, C, T) batch is read in place, runner.py:561,570)
            _lib.check(lib.se_masked_sumsq_f32(ref.data_ptr(), B, T, ref.stride(0), _lib.ptr(lengths), _lib.ptr(ref_sumsq),
                                               _lib.stream()), 'se_masked_sumsq_f32')
    else:
        raise NotImplementedError('per-utterance dB tensor targets are unused by the reference')
    _lib.check(lib.se_dbnorm_f32(_lib.ptr(audio), B, T, T, _lib.ptr(lengths), _lib.ptr(audio_sumsq), _lib.ptr(ref_sumsq),
                                 fixed, float(eps), _lib.stream()), 'se_dbnorm_f32')
    return audio


def decode_wav(preprocessor, linear, phase, lengths, target_level=-25, max_len=None):
    """D2: Runner._decode_wav (runner.py:266-270): istft -> right-pad to max(lengths) -> masked dB-normalise.
    `max_len` = max(lengths) if known on the host (avoids the sync of runner.py:268)."""
    if max_len is None:
        max_len = int(lengths.max().item())
    ref = target_level if isinstance(target_level, torch.Tensor) and target_level.dim() == 2 else None
    if ref is not None:
        wav, sumsq, ref_sumsq = preprocessor.istft_with_sumsq(linear, phase, lengths=lengths, out_len=max_len, ref=ref)
        return masked_normalize_decibel(wav, target_level, lengths, audio_sumsq=sumsq, inplace=True, ref_sumsq=ref_sumsq)
    wav, sumsq = preprocessor.istft_with_sumsq(linear, phase, lengths=lengths, out_len=max_len)
    return masked_normalize_decibel(wav, target_level, lengths, audio_sumsq=sumsq, inplace=True)
