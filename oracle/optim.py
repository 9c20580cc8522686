"""Oracle row E2: gradient clipping + BertAdam step (test infrastructure only).
BertAdam (pytorch-pretrained-BERT, wrapped by S3PRL downstream.solver.get_optimizer, called at
runner.py:110-113): Adam WITHOUT bias correction, decoupled weight decay 0.01 on everything except
bias / LayerNorm params, per-parameter grad clip at max_grad_norm=1.0, e=1e-6, linear warm-up then
linear decay.  PARITY UNPINNED vs original S3PRL (source absent)."""
import torch


def warmup_linear(x, warmup=0.002):
    if x < warmup:
        return x / warmup
    return max((x - 1.0) / (warmup - 1.0), 0.0)


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (runner.py:464): global L2 norm, scale by max_norm/(norm+1e-6) if > 1."""
    total = torch.sqrt(sum(g.pow(2).sum() for g in grads))
    coef = max_norm / (total + 1e-6)
    if coef < 1:
        grads = [g * coef for g in grads]
    return grads, total


def bert_adam_step(p, g, m, v, step, lr, warmup, t_total, weight_decay, b1=0.9, b2=0.999, e=1e-6, max_grad_norm=1.0):
    """One BertAdam update of one parameter; returns (p, m, v)."""
    if max_grad_norm > 0:
        n = g.norm()
        c = max_grad_norm / (n + 1e-6)
        if c < 1:
            g = g * c
    m = m * b1 + (1 - b1) * g
    v = v * b2 + (1 - b2) * g * g
    update = m / (v.sqrt() + e)
    if weight_decay > 0.0:
        update = update + weight_decay * p
    lr_scheduled = lr * warmup_linear(step / t_total, warmup) if t_total != -1 else lr
    return p - lr_scheduled * update, m, v
