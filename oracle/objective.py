"""Oracle row E1 + metrics (test infrastructure only); PINNED against the reference's objective.py /
evaluation.py via tests/golden/reference_golden.npz."""
import torch


def l1(log_predicted, linear_tar, stft_length_masks, eps=1e-10):
    """E1: objective.py:103-117  mean |log_predicted - log(linear_tar+eps)| over elements of valid frames
    (ONE global mean over the whole batch)."""
    m = stft_length_masks.unsqueeze(-1).bool()
    src = log_predicted.masked_select(m)
    tar = linear_tar.masked_select(m)
    return (src - (tar + eps).log()).abs().mean()


def l1_sums(log_predicted, linear_tar, stft_length_masks, eps=1e-10):
    """Un-normalised form used under data parallelism: (sum |.|, element count)."""
    m = stft_length_masks.unsqueeze(-1).to(log_predicted.dtype)
    d = (log_predicted - (linear_tar + eps).log()).abs() * m
    return d.sum(), m.sum() * log_predicted.size(-1)


def sisdr_objective(predicted, linear_tar, stft_length_masks, eps=1e-10):
    """objective.py:81-100"""
    src = torch.relu(predicted).pow(0.5) * stft_length_masks.unsqueeze(-1)
    tar = torch.relu(linear_tar).pow(0.5) * stft_length_masks.unsqueeze(-1)
    src = src.flatten(start_dim=1)
    tar = tar.flatten(start_dim=1)
    alpha = torch.sum(src * tar, dim=1) / (torch.sum(tar * tar, dim=1) + eps)
    ay = alpha.unsqueeze(1) * tar
    norm = torch.sum((ay - src) * (ay - src), dim=1) + eps
    loss = -10 * torch.log10(torch.sum(ay * ay, dim=1) / norm + eps)
    return loss.mean()


def wsd_objective(linear_inp, offset, linear_tar, stft_length_masks, alpha=0.5, db_interval=30, eps=1e-10):
    """objective.py:119-153 (without the TensorBoard logger)"""
    S, G = linear_tar, offset
    N = torch.clamp(linear_inp - linear_tar, min=0.0)
    energy = S.sum(dim=-1, keepdim=True)
    db_thres = 10 * torch.log10(energy.max() + eps) - db_interval
    voice_mask = ((10 * torch.log10(energy + eps)) > db_thres).to(S.dtype)
    m = stft_length_masks.unsqueeze(-1).to(S.dtype)
    speech_loss = ((S - G * S) * voice_mask * m).pow(2).sum(-1).sum(-1).mean()
    noise_loss = (G * N * m).pow(2).sum(-1).sum(-1).mean()
    return alpha * speech_loss + (1 - alpha) * noise_loss


def sisdr_eval(src, tar, eps=1e-10):
    """evaluation.py:5-10"""
    alpha = (src * tar).sum() / ((tar * tar).sum() + eps)
    ay = alpha * tar
    norm = ((ay - src) * (ay - src)).sum() + eps
    return (10 * ((ay * ay).sum() / norm + eps).log10()).item()


def matching(query_scores, key_scores, eps=1e-12):
    """sampler.py:113-116"""
    q = query_scores / (query_scores.pow(2).sum(dim=-1, keepdim=True).pow(0.5) + eps)
    k = key_scores / (key_scores.pow(2).sum(dim=-1, keepdim=True).pow(0.5) + eps)
    return torch.mm(k, q.mean(dim=0).unsqueeze(1)).reshape(-1)
