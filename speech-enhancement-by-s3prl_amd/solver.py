"""`get_optimizer` -- drop-in for S3PRL downstream.solver.get_optimizer (called at runner.py:110-113): BertAdam
(pytorch-pretrained-BERT): Adam WITHOUT bias correction, decoupled weight decay 0.01 on everything except bias /
LayerNorm parameters, per-parameter gradient clipping at 1.0, e = 1e-6, linear warm-up then linear decay.
Row E2; parity unpinned vs the original S3PRL (source absent offline).  Host-side torch (a 24 k .. 4 M parameter
update; not on the roofline path)."""
import torch
from torch.optim import Optimizer


def warmup_linear(x, warmup=0.002):
    if x < warmup:
        return x / warmup
    return max((x - 1.0) / (warmup - 1.0), 0.0)


class BertAdam(Optimizer):
    def __init__(self, params, lr, warmup=-1, t_total=-1, b1=0.9, b2=0.999, e=1e-6, weight_decay=0.01, max_grad_norm=1.0):
        defaults = dict(lr=lr, warmup=warmup, t_total=t_total, b1=b1, b2=b2, e=e, weight_decay=weight_decay,
                        max_grad_norm=max_grad_norm)
        super().__init__(params, defaults)

    def get_lr(self):
        lr = []
        for group in self.param_groups:
            for p in group['params']:
                state = self.state[p]
                if len(state) == 0:
                    return [0]
                if group['t_total'] != -1:
                    lr.append(group['lr'] * warmup_linear(state['step'] / group['t_total'], group['warmup']))
                else:
                    lr.append(group['lr'])
        return lr

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is None:
                    continue
                grad = p.grad
                state = self.state[p]
                if len(state) == 0:
                    state['step'] = 0
                    state['next_m'] = torch.zeros_like(p)
                    state['next_v'] = torch.zeros_like(p)
                next_m, next_v = state['next_m'], state['next_v']
                if group['max_grad_norm'] > 0:
                    torch.nn.utils.clip_grad_norm_(p, group['max_grad_norm'])
                next_m.mul_(group['b1']).add_(grad, alpha=1 - group['b1'])
                next_v.mul_(group['b2']).addcmul_(grad, grad, value=1 - group['b2'])
                update = next_m / (next_v.sqrt() + group['e'])
                if group['weight_decay'] > 0.0:
                    update = update + group['weight_decay'] * p
                if group['t_total'] != -1:
                    lr_scheduled = group['lr'] * warmup_linear(state['step'] / group['t_total'], group['warmup'])
                else:
                    lr_scheduled = group['lr']
                p.add_(update, alpha=-lr_scheduled)
                state['step'] += 1
        return loss


def get_optimizer(params, lr, warmup_proportion, training_steps):
    """params = list(model.named_parameters()) (runner.py:110)."""
    no_decay = ['bias', 'LayerNorm.bias', 'LayerNorm.weight']
    grouped = [
        {'params': [p for n, p in params if not any(nd in n for nd in no_decay)], 'weight_decay': 0.01},
        {'params': [p for n, p in params if any(nd in n for nd in no_decay)], 'weight_decay': 0.0},
    ]
    return BertAdam(grouped, lr=lr, warmup=warmup_proportion, t_total=training_steps)
