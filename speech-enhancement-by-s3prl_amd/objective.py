"""Row E1: the L1 objective of the reference (objective.py:103-117) on the gfx950 kernel, with autograd.

Criterion contract kept: forward(**all_runner_locals, **model_results) -> (loss, dict); arguments are bound
by local-variable name and the rest swallowed by **kwargs (runner.py:458,575).  Under data parallelism the
un-normalised sum and the element count are all-reduced BEFORE dividing (a global masked mean, not a mean of
per-rank means); see dist.py."""
import torch
import torch.nn as nn

from . import _lib


_L1_SCRATCH = {}      # (device, stream) -> persistent scratch of se_l1_masked_loss_f32: {ticket, -, per-workgroup partial sums}; the ticket is self-cleaning


def _l1_scratch(dev, B):
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream)      # one per stream: calls on two streams may overlap
    n = int(_lib.load().se_l1_scratch_doubles(int(B)))
    t = _L1_SCRATCH.get(key)
    if t is None or t.numel() < n:
        t = _L1_SCRATCH[key] = torch.zeros(max(n, 1024), device=dev, dtype=torch.float64)
    return t


class _L1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, log_predicted, linear_tar, lengths, len_div, eps, reduce_fn):
        """lengths: frame counts (len_div == 0) or waveform lengths (frames = lengths // len_div + 1, runner.py:455)"""
        lib = _lib.load()
        lp = log_predicted.contiguous().float()
        tar = linear_tar.contiguous().float()
        B, F, K = lp.shape
        lens = lengths.to(device=lp.device, dtype=torch.int64).contiguous()
        sums = torch.empty(2, device=lp.device, dtype=torch.float64)
        need_grad = log_predicted.requires_grad
        grad = torch.empty_like(lp) if need_grad else None
        if reduce_fn is None:
            # one launch: sums, count AND the loss come out of the kernel (last-arriving workgroup), no zeroing launch, no division kernels
            loss = torch.empty((), device=lp.device, dtype=torch.float32)
            rc = lib.se_l1_masked_loss_f32(_lib.ptr(lp), _lib.ptr(tar), _lib.ptr(lens), int(len_div), B, F, K, float(eps), _lib.ptr(_l1_scratch(lp.device, B)),
                                           _lib.ptr(sums), _lib.ptr(loss), _lib.ptr(grad), _lib.stream())
            if rc != 0:
                # the arrival ticket is reset by the launch's LAST workgroup: after a launch that did not run to completion it may be anything, and a
                # stale ticket would leave `loss` unwritten on every later call -- start the next call from a fresh zeroed scratch
                _L1_SCRATCH.clear()
            _lib.check(rc, 'se_l1_masked_loss_f32')
            ctx.save_for_backward(grad if grad is not None else torch.empty(0), sums)
            return loss
        if len_div:
            lens = lens // int(len_div) + 1
        _lib.check(lib.se_l1_masked_f32(_lib.ptr(lp), _lib.ptr(tar), _lib.ptr(lens), B, F, K, float(eps), _lib.ptr(sums),
                                        _lib.ptr(grad), _lib.stream()), 'se_l1_masked_f32')
        if reduce_fn is not None:
            sums = reduce_fn(sums)          # all-reduce(sum) of (sum |.|, count) across ranks
        ctx.save_for_backward(grad if grad is not None else torch.empty(0), sums)
        return (sums[0] / sums[1]).float()

    @staticmethod
    def backward(ctx, g):
        sign, sums = ctx.saved_tensors
        return sign * (g / sums[1].float()), None, None, None, None, None


class L1(nn.Module):
    """objective.py:103-117.  `stft_length_masks` (B, T') is what the runner passes; the kernel takes the equivalent per-utterance frame
    counts, so either `stft_lengths` (no extra pass), the mask, or -- cheapest, nothing to compute in front -- the waveform `wav_lengths`
    with `hop` (frames = wav_lengths // hop + 1, runner.py:455)."""

    def __init__(self, eps=1e-10, **kwargs):
        super().__init__()
        self.eps = eps
        self.reduce_fn = None   # set by dist.DataParallelStep for the global masked mean

    def forward(self, log_predicted, linear_tar, stft_length_masks=None, stft_lengths=None, wav_lengths=None, hop=None, **kwargs):
        if wav_lengths is not None and hop:
            loss = _L1Fn.apply(log_predicted, linear_tar, wav_lengths, int(hop), self.eps, self.reduce_fn)
            return loss, {}
        if stft_lengths is None:
            stft_lengths = stft_length_masks.sum(dim=-1)
        loss = _L1Fn.apply(log_predicted, linear_tar, stft_lengths, 0, self.eps, self.reduce_fn)
        return loss, {}


def _frame_lengths(stft_length_masks, stft_lengths, device):
    if stft_lengths is None:
        stft_lengths = stft_length_masks.sum(dim=-1)
    return stft_lengths.to(device=device, dtype=torch.int64).contiguous()


def sisdr_loss_inference(predicted, linear_tar, lengths, len_div, eps):
    """objective.py:81-100 without autograd (evaluate(), runner.py:575): two launches -- partial sums, then loss_b and their mean -- with the
    frame counts derived inside (`lengths` = waveform lengths when len_div = hop, runner.py:455).  Returns (loss, loss_b)."""
    lib = _lib.load()
    p, t = predicted.contiguous().float(), linear_tar.contiguous().float()
    B, F, N = p.shape
    lens = lengths.to(device=p.device, dtype=torch.int64).contiguous()
    scratch = torch.empty(int(lib.se_sisdr_spec_loss_scratch_doubles(B, F, N)) + 2, device=p.device, dtype=torch.float64)
    loss_b = torch.empty(B, device=p.device, dtype=torch.float32)
    loss = torch.empty((), device=p.device, dtype=torch.float32)
    _lib.check(lib.se_sisdr_spec_loss_f32(_lib.ptr(p), _lib.ptr(t), _lib.ptr(lens), int(len_div), B, F, N, float(eps), scratch[2:].data_ptr(), _lib.ptr(loss_b),
                                          scratch.data_ptr(), _lib.ptr(loss), _lib.stream()), 'se_sisdr_spec_loss_f32')
    return loss, loss_b


class _SISDRFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, predicted, linear_tar, lens, eps, reduce_fn):
        lib = _lib.load()
        p, t = predicted.contiguous().float(), linear_tar.contiguous().float()
        B, F, N = p.shape
        scratch = torch.empty(5 * B, device=p.device, dtype=torch.float64)
        loss_b = torch.empty(B, device=p.device, dtype=torch.float32)
        grad = torch.empty_like(p) if predicted.requires_grad else None
        _lib.check(lib.se_sisdr_spec_f32(_lib.ptr(p), _lib.ptr(t), _lib.ptr(lens), B, F, N, float(eps), 1.0, _lib.ptr(scratch), _lib.ptr(loss_b),
                                         _lib.ptr(grad), _lib.stream()), 'se_sisdr_spec_f32')
        sums = torch.stack([loss_b.double().sum(), torch.full((), float(B), device=p.device, dtype=torch.float64)])      # a fill kernel: capturable in a hipGraph (torch.tensor() is a host copy)
        if reduce_fn is not None:
            sums = reduce_fn(sums)          # (sum of per-utterance losses, utterance count) across ranks: a global mean over utterances
        ctx.save_for_backward(grad if grad is not None else torch.empty(0), sums)
        return (sums[0] / sums[1]).float()

    @staticmethod
    def backward(ctx, g):
        grad, sums = ctx.saved_tensors
        return grad * (g / sums[1].float()), None, None, None, None


class SISDR(nn.Module):
    """objective.py:81-100: scale-invariant SDR between sqrt-power spectrograms, mean over the utterances of the batch."""

    def __init__(self, eps=1e-10, **kwargs):
        super().__init__()
        self.eps = eps
        self.reduce_fn = None

    def forward(self, predicted, linear_tar, stft_length_masks=None, stft_lengths=None, wav_lengths=None, hop=None, **kwargs):
        no_grad = not (torch.is_grad_enabled() and predicted.requires_grad)
        if no_grad and self.reduce_fn is None and predicted.is_cuda:
            # evaluate(): nothing differentiates the criterion and no other rank contributes -- loss and mean from two launches
            side = getattr(predicted, '_se_sisdr', None)        # heads.LinearResidual.enhance: the sums came out of the head's own launch
            if side is not None and side[1] == predicted._version and side[2] is linear_tar and side[3] == float(self.eps):
                return side[0], {}
            if wav_lengths is not None and hop:
                return sisdr_loss_inference(predicted, linear_tar, wav_lengths, int(hop), self.eps)[0], {}
            return sisdr_loss_inference(predicted, linear_tar, _frame_lengths(stft_length_masks, stft_lengths, predicted.device), 0, self.eps)[0], {}
        if stft_lengths is None and stft_length_masks is None and wav_lengths is not None and hop:
            stft_lengths = wav_lengths // int(hop) + 1
        lens = _frame_lengths(stft_length_masks, stft_lengths, predicted.device)
        return _SISDRFn.apply(predicted, linear_tar, lens, self.eps, self.reduce_fn), {}


class _WSDFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, linear_inp, offset, linear_tar, lens, alpha, db_interval, eps, reduce_fn, max_reduce_fn):
        lib = _lib.load()
        inp, off, tar = linear_inp.contiguous().float(), offset.contiguous().float(), linear_tar.contiguous().float()
        B, F, N = off.shape
        dev = off.device
        energy = torch.empty(B * F, device=dev, dtype=torch.float32)
        emax = torch.empty(1, device=dev, dtype=torch.float32)
        _lib.check(lib.se_wsd_energy_f32(_lib.ptr(tar), B, F, N, _lib.ptr(energy), _lib.ptr(emax), _lib.stream()), 'se_wsd_energy_f32')
        if max_reduce_fn is not None:
            emax = max_reduce_fn(emax)      # the reference's threshold is relative to the loudest frame of the WHOLE batch
        sums = torch.empty(3, device=dev, dtype=torch.float64)
        grad = torch.empty_like(off) if offset.requires_grad else None
        _lib.check(lib.se_wsd_f32(_lib.ptr(inp), _lib.ptr(off), _lib.ptr(tar), _lib.ptr(lens), _lib.ptr(energy), _lib.ptr(emax), B, F, N, float(alpha),
                                  float(db_interval), float(eps), 1.0, _lib.ptr(sums), _lib.ptr(grad), _lib.stream()), 'se_wsd_f32')
        tot = torch.stack([alpha * sums[0] + (1.0 - alpha) * sums[1], torch.full((), float(B), device=dev, dtype=torch.float64)])
        if reduce_fn is not None:
            tot = reduce_fn(tot)
        ctx.save_for_backward(grad if grad is not None else torch.empty(0), tot)
        return (tot[0] / tot[1]).float()

    @staticmethod
    def backward(ctx, g):
        grad, tot = ctx.saved_tensors
        return None, grad * (g / tot[1].float()), None, None, None, None, None, None, None


class WSD(nn.Module):
    """objective.py:119-153: weighted speech-distortion loss on the mask (`offset`), without the TensorBoard logger."""

    def __init__(self, alpha=0.5, db_interval=30, eps=1e-10, **kwargs):
        super().__init__()
        self.alpha, self.db_interval, self.eps = alpha, db_interval, eps
        self.reduce_fn = None
        self.max_reduce_fn = None

    def forward(self, linear_inp, offset, linear_tar, stft_length_masks=None, stft_lengths=None, **kwargs):
        lens = _frame_lengths(stft_length_masks, stft_lengths, offset.device)
        loss = _WSDFn.apply(linear_inp, offset, linear_tar, lens, self.alpha, self.db_interval, self.eps, self.reduce_fn, self.max_reduce_fn)
        return loss, {}
