"""Row E1: the L1 objective of the reference (objective.py:103-117) on the gfx950 kernel, with autograd.

Criterion contract kept: forward(**all_runner_locals, **model_results) -> (loss, dict); arguments are bound
by local-variable name and the rest swallowed by **kwargs (runner.py:458,575).  Under data parallelism the
un-normalised sum and the element count are all-reduced BEFORE dividing (a global masked mean, not a mean of
per-rank means); see dist.py."""
import torch
import torch.nn as nn

from . import _lib


class _L1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, log_predicted, linear_tar, frame_lengths, eps, reduce_fn):
        lib = _lib.load()
        lp = log_predicted.contiguous().float()
        tar = linear_tar.contiguous().float()
        B, F, K = lp.shape
        lens = frame_lengths.to(device=lp.device, dtype=torch.int64).contiguous()
        sums = torch.empty(2, device=lp.device, dtype=torch.float64)
        need_grad = log_predicted.requires_grad
        grad = torch.empty_like(lp) if need_grad else None
        _lib.check(lib.se_l1_masked_f32(_lib.ptr(lp), _lib.ptr(tar), _lib.ptr(lens), B, F, K, float(eps), _lib.ptr(sums),
                                        _lib.ptr(grad), _lib.stream()), 'se_l1_masked_f32')
        if reduce_fn is not None:
            sums = reduce_fn(sums)          # all-reduce(sum) of (sum |.|, count) across ranks
        ctx.save_for_backward(grad if grad is not None else torch.empty(0), sums)
        return (sums[0] / sums[1]).float()

    @staticmethod
    def backward(ctx, g):
        sign, sums = ctx.saved_tensors
        return sign * (g / sums[1].float()), None, None, None, None


class L1(nn.Module):
    """objective.py:103-117.  `stft_length_masks` (B, T') is what the runner passes; the kernel takes the
    equivalent per-utterance frame counts, so either `stft_lengths` (preferred, no extra pass) or the mask."""

    def __init__(self, eps=1e-10, **kwargs):
        super().__init__()
        self.eps = eps
        self.reduce_fn = None   # set by dist.DataParallelStep for the global masked mean

    def forward(self, log_predicted, linear_tar, stft_length_masks=None, stft_lengths=None, **kwargs):
        if stft_lengths is None:
            stft_lengths = stft_length_masks.sum(dim=-1)
        loss = _L1Fn.apply(log_predicted, linear_tar, stft_lengths, self.eps, self.reduce_fn)
        return loss, {}
