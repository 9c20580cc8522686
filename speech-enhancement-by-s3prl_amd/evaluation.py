"""Metric scoring on the device: SI-SDR of every utterance of a batch without moving the
waveforms to the host.  Mirrors evaluation.sisdr_eval (evaluation.py:5-10) and the per-utterance trimming of
runner.py:597-603.  PESQ / STOI stay third-party CPU metrics (absent offline)."""
import torch

from . import _lib


def sisdr_batch(wav_pred, wav_tar, lengths=None, eps=1e-10):
    """wav_pred, wav_tar (B, T) fp32 device tensors; lengths (B,) -> SI-SDR (B,) in dB, each over its first lengths[b] samples."""
    if not wav_pred.is_cuda:
        raise _lib.SEError('sisdr_batch runs on MI355X only (no CPU fallback)')
    lib = _lib.load()
    dev = wav_pred.device
    a, b = wav_pred.contiguous().float(), wav_tar.contiguous().float()
    assert a.shape == b.shape and a.dim() == 2
    B, T = a.shape
    ln = None if lengths is None else lengths.to(dev, torch.int64).contiguous()
    sums = torch.empty(3 * B, device=dev, dtype=torch.float64)
    out = torch.empty(B, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _lib.check(lib.se_sisdr_f32(_lib.ptr(a), _lib.ptr(b), T, _lib.ptr(ln), B, float(eps), _lib.ptr(sums), _lib.ptr(out), _lib.stream()),
                   'se_sisdr_f32')
    return out


def sisdr_eval(src, tar, sr=16000, eps=1e-10):
    """Drop-in for evaluation.sisdr_eval (one utterance, returns a float)."""
    return float(sisdr_batch(src.reshape(1, -1), tar.reshape(1, -1), None, eps)[0])
