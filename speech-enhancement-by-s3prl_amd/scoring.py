"""Per-utterance gradient scoring for active sampling (sampler.py:59-116).

The reference's `scoring` runs the downstream model once and then B sequential `loss_b.backward(retain_graph=True)` passes,
collecting each utterance's flattened parameter gradient; `matching` turns them into cosine scores against the mean query
gradient.  Here ONE backward sweep produces all B gradients: the row-wise parts of the backward (epilogue, LayerNorm, GELU)
are per-frame anyway, the LayerNorm parameter gradients come from one grouped launch (grid.y = utterance), and the weight gradients come out of the TN weight-gradient kernel as its per-split slabs with the
split boundaries on utterance boundaries (se_wgrad_tn_slabs_bf16) -- no reduce, no retain_graph, no B-fold replay.
Implemented for the downstream models whose `log_predicted` the reference's `scoring` reads (sampler.py:72): `SpecHead` (model.py:94-126,
per_sample_gradients) and `LSTM` (model.py:37-59 -- the head run_active.sh:11 actually scores; per_sample_gradients_lstm, with the
`active_layerid` filter of sampler.py:91-100).  `LinearResidual` / `Residual` return `offset`, not `log_predicted`: the reference's own
`scoring` raises KeyError on them, so there is nothing to mirror."""
import re

import torch

from . import _lib
from . import spechead_train as st


def _colsum_groups(x, groups, rows):
    """(groups * rows, C) fp32 -> (groups, C): per-utterance bias gradients, one grouped launch."""
    lib = _lib.load()
    C = x.shape[1]
    out = torch.empty(groups, C, device=x.device, dtype=torch.float32)
    _lib.check(lib.se_colsum_groups(_lib.ptr(x), 0, groups, rows, C, C, _lib.ptr(out), _lib.stream()), 'se_colsum_groups')
    return out


def _l1_row_gradients(log_predicted, linear_tar, stft_lengths, eps):
    """d L1_b / d log_predicted, row block b: masked sign / (frames_b * N)  (objective.py:103-117 on ONE utterance, as sampler.py:84-86 slices it)"""
    lib = _lib.load()
    B, T, N = log_predicted.shape
    dev = log_predicted.device
    lens = stft_lengths.to(dev, torch.int64).contiguous()
    sums = torch.empty(2, device=dev, dtype=torch.float64)
    sign = torch.empty(B * T, N, device=dev, dtype=torch.float32)
    _lib.check(lib.se_l1_masked_f32(_lib.ptr(log_predicted.contiguous().float()), _lib.ptr(linear_tar.contiguous().float()), _lib.ptr(lens), B, T, N, float(eps),
                                    _lib.ptr(sums), _lib.ptr(sign), _lib.stream()), 'se_l1_masked_f32')
    return (sign.view(B, T * N) / (lens.clamp(min=1).float() * N)[:, None]).view(B, T, N)


def per_sample_gradients_lstm(head, features, linear_tar, stft_lengths, active_layerid=None, eps=1e-10):
    """head: lstm.LSTM; features (B, T, D); linear_tar (B, T, N); stft_lengths (B,).  Returns grads (B, P): row b = the flattened gradient of
    utterance b's masked log-L1 wrt the head's parameters in named_parameters() order; with `active_layerid` only the nn.LSTM parameters of
    that layer, selected by the reference's own pattern (sampler.py:91-100: re.search('lstm.*l(\\d+)', key))."""
    if not features.is_cuda:
        raise _lib.SEError('per_sample_gradients_lstm runs on MI355X only (no CPU fallback)')
    B = features.shape[0]
    by_name = head.per_utterance_gradients(features, lambda logp: _l1_row_gradients(logp, linear_tar, stft_lengths, eps))
    cols = []
    for key, _ in head.named_parameters():
        if active_layerid is None:
            cols.append(by_name[key].reshape(B, -1))
        else:
            pattern = re.search(r'lstm.*l(\d+)', key)
            if pattern is not None and int(pattern.group().split('_')[-1][1:]) == active_layerid:
                cols.append(by_name[key].reshape(B, -1))
    return torch.cat(cols, dim=1)


def per_sample_gradients(head, features, linear_tar, stft_lengths, eps=1e-10):
    """head: heads.SpecHead; features (B, T, 768); linear_tar (B, T, N); stft_lengths (B,).
    Returns grads (B, P) fp32, P = number of head parameters, columns in `head.named_parameters()` order -- row b is
    d L1_b / d params with L1_b the masked log-L1 of utterance b alone (sampler.py:84-104)."""
    if not features.is_cuda:
        raise _lib.SEError('per_sample_gradients runs on MI355X only (no CPU fallback)')
    lib = _lib.load()
    sh = head.spechead
    dev = features.device
    B, T, H = features.shape
    N = sh.output.out_features
    NP = max(128, (N + 63) // 64 * 64)
    M = B * T
    ln_eps = float(sh.LayerNorm.variance_epsilon)
    # ---- forward of the head, keeping what the backward needs (as SpecHeadTrainFn.forward)
    x16 = st.cast_bf16(features.reshape(M, H))
    wd16, wo16 = st.cast_bf16(sh.dense.weight.detach()), st.cast_bf16(sh.output.weight.detach())
    pre = st._gemm(x16, wd16, sh.dense.bias.detach().contiguous().float(), M, H, H)
    xn16 = torch.empty(M, H, device=dev, dtype=torch.bfloat16)
    lnw = sh.LayerNorm.weight.detach().contiguous().float()
    _lib.check(lib.se_gelu_layernorm_f32(_lib.ptr(pre), _lib.ptr(lnw), _lib.ptr(sh.LayerNorm.bias.detach().contiguous().float()), M, H, ln_eps,
                                         None, _lib.ptr(xn16), _lib.stream()), 'se_gelu_layernorm_f32')
    p = st._gemm(xn16, wo16, sh.output.bias.detach().contiguous().float(), M, N, H)
    log_target = bool(head.log)
    if log_target:
        logp = p
    else:
        logp = torch.empty_like(p)
        _lib.check(lib.se_spec_epilogue_f32(_lib.ptr(p), M * N, 0, 0, float(head.eps), None, _lib.ptr(logp), _lib.stream()), 'se_spec_epilogue_f32')
    # ---- d L1_b / d log_predicted: masked sign / (frames_b * N)   (objective.py:103-117 on one utterance)
    lens = stft_lengths.to(dev, torch.int64).contiguous()
    sums = torch.empty(2, device=dev, dtype=torch.float64)
    sign = torch.empty(M, N, device=dev, dtype=torch.float32)
    tar = linear_tar.contiguous().float()
    _lib.check(lib.se_l1_masked_f32(_lib.ptr(logp), _lib.ptr(tar), _lib.ptr(lens), B, T, N, float(eps), _lib.ptr(sums), _lib.ptr(sign),
                                    _lib.stream()), 'se_l1_masked_f32')
    d_logp = (sign.view(B, T * N) / (lens.clamp(min=1).float() * N)[:, None]).view(M, N)
    # ---- row-wise backward, whole batch at once
    dp32 = torch.empty(M, N, device=dev, dtype=torch.float32)
    dp16 = torch.empty(M, NP, device=dev, dtype=torch.bfloat16)
    _lib.check(lib.se_spec_epilogue_bwd_f32(_lib.ptr(p), None, _lib.ptr(d_logp), M, N, NP, int(log_target), _lib.SE_ACT['Identity'], float(head.eps),
                                            _lib.ptr(dp32), _lib.ptr(dp16), _lib.stream()), 'se_spec_epilogue_bwd_f32')
    wot = st.transpose_f32_bf16(sh.output.weight.detach().contiguous().float(), NP)
    dxn = st._gemm(dp16, wot, None, M, H, NP)
    dpre32 = torch.empty(M, H, device=dev, dtype=torch.float32)
    dpre16 = torch.empty(M, H, device=dev, dtype=torch.bfloat16)
    d_lnw = torch.empty(B, H, device=dev, dtype=torch.float32)
    d_lnb = torch.empty(B, H, device=dev, dtype=torch.float32)
    # LayerNorm parameter gradients reduce over an utterance's rows: grid.y = utterance, one launch (row-wise outputs land in place)
    _lib.check(lib.se_layernorm_bwd_groups_f32(_lib.ptr(pre), _lib.ptr(dxn), _lib.ptr(lnw), B, T, H, ln_eps, 1, _lib.ptr(dpre32), _lib.ptr(dpre16),
                                               _lib.ptr(d_lnw), _lib.ptr(d_lnb), _lib.stream()), 'se_layernorm_bwd_groups_f32')
    # ---- per-utterance weight gradients: the TN kernel's slabs, split on utterance boundaries
    d_wo = torch.empty(B, NP, H, device=dev, dtype=torch.float32)
    _lib.check(lib.se_wgrad_tn_slabs_bf16(_lib.ptr(dp16), NP, _lib.ptr(xn16), H, M, NP, H, T, _lib.ptr(d_wo), _lib.stream()), 'se_wgrad_tn_slabs_bf16')
    d_wd = torch.empty(B, H, H, device=dev, dtype=torch.float32)
    _lib.check(lib.se_wgrad_tn_slabs_bf16(_lib.ptr(dpre16), H, _lib.ptr(x16), H, M, H, H, T, _lib.ptr(d_wd), _lib.stream()), 'se_wgrad_tn_slabs_bf16')
    d_bo = _colsum_groups(dp32, B, T)
    d_bd = _colsum_groups(dpre32, B, T)
    by_name = {'spechead.dense.weight': d_wd.reshape(B, -1), 'spechead.dense.bias': d_bd, 'spechead.LayerNorm.weight': d_lnw,
               'spechead.LayerNorm.bias': d_lnb, 'spechead.output.weight': d_wo[:, :N].reshape(B, -1), 'spechead.output.bias': d_bo}
    return torch.cat([by_name[n] for n, _ in head.named_parameters()], dim=1)


def matching(query_scores, key_scores, eps=1e-12):
    """sampler.py:113-116: cosine of every key gradient with the mean normalised query gradient."""
    q = query_scores / (query_scores.pow(2).sum(dim=-1, keepdim=True).pow(0.5) + eps)
    k = key_scores / (key_scores.pow(2).sum(dim=-1, keepdim=True).pow(0.5) + eps)
    return torch.mm(k, q.mean(dim=0).unsqueeze(1)).reshape(-1)


def thresholding(match_scores):
    """sampler.py:119-120"""
    return match_scores > 0
