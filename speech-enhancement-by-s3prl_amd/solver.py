"""`get_optimizer` -- drop-in for S3PRL downstream.solver.get_optimizer (called at runner.py:110-113): BertAdam
(pytorch-pretrained-BERT): Adam WITHOUT bias correction, decoupled weight decay 0.01 on everything except bias /
LayerNorm parameters, per-parameter gradient clipping at 1.0, e = 1e-6, linear warm-up then linear decay.
Row E2; parity unpinned vs the original S3PRL (source absent offline).

Parameters on the GPU are updated by two HIP launches for the whole model (se_multi_sumsq_f32 + se_bertadam_step_f32:
per-tensor norms, then clip + Adam + decay fused) instead of ~15 torch kernels per parameter tensor -- for the 43 M
parameter Mockingjay fine-tune that is the difference between ~2 and ~1700 launches a step.  The per-tensor torch arithmetic below is
the restated algorithm for parameter sets the fused launch does not take (heterogeneous groups, non-contiguous gradients) and what the
fused kernel is tested against; it runs where the parameters live.  There is no CPU product path: the downstream modules themselves
raise on host tensors, so an optimizer over host parameters is only ever reached by the world-size-2 gloo test of the reduction logic."""
import ctypes

import torch
from torch.optim import Optimizer

from . import _lib


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

    # ---- fused device path -------------------------------------------------------------------------------------------
    def _fused_table(self, grads=None):
        """ctypes tables over every parameter that has a gradient (state is created on first use)."""
        ps, gs, wds, steps = [], [], [], []
        for group in self.param_groups:
            for p in group['params']:
                g = grads.get(p) if grads is not None else p.grad
                if g is None:
                    continue
                state = self.state[p]
                if len(state) == 0:
                    state['step'] = 0
                    state['next_m'] = torch.zeros_like(p)
                    state['next_v'] = torch.zeros_like(p)
                if not (p.is_contiguous() and g.is_contiguous() and p.dtype == torch.float32 and g.dtype == torch.float32 and g.is_cuda):
                    return None
                ps.append((p, g, state, group))
        if not ps:
            return None
        g0 = ps[0][3]
        for _, _, st, gr in ps:
            if st['step'] != ps[0][2]['step'] or any(gr[k] != g0[k] for k in ('lr', 'warmup', 't_total', 'b1', 'b2', 'e', 'max_grad_norm')):
                return None                       # heterogeneous groups: keep the per-tensor torch path
        n = len(ps)
        VP, U64, F32 = ctypes.c_void_p * n, ctypes.c_uint64 * n, ctypes.c_float * n
        tab = {'n': n, 'items': ps,
               'p': VP(*[p.data_ptr() for p, _, _, _ in ps]), 'g': VP(*[g.data_ptr() for _, g, _, _ in ps]),
               'm': VP(*[st['next_m'].data_ptr() for _, _, st, _ in ps]), 'v': VP(*[st['next_v'].data_ptr() for _, _, st, _ in ps]),
               'sizes': U64(*[p.numel() for p, _, _, _ in ps]), 'wd': F32(*[float(gr['weight_decay']) for _, _, _, gr in ps])}
        return tab

    def grad_sumsq(self, tab):
        """Per-tensor sums of squares of the gradients in `tab` (device fp64 tensor); one launch."""
        lib = _lib.load()
        dev = tab['items'][0][0].device
        sumsq = torch.empty(tab['n'], device=dev, dtype=torch.float64)
        with torch.cuda.device(dev):
            _lib.check(lib.se_multi_sumsq_f32(tab['g'], tab['sizes'], tab['n'], _lib.ptr(sumsq), _lib.stream()), 'se_multi_sumsq_f32')
        return sumsq

    @torch.no_grad()
    def step_fused(self, tab, sumsq, global_max_norm=-1.0):
        """One BertAdam update of every tensor in `tab` (one launch per 64 tensors).  `global_max_norm` > 0 folds the
        runner's clip_grad_norm_ (runner.py:464) into the same pass -- the gradients themselves are left unscaled."""
        lib = _lib.load()
        p0, _, st0, gr = tab['items'][0]
        if gr['t_total'] != -1:
            lr_t = gr['lr'] * warmup_linear(st0['step'] / gr['t_total'], gr['warmup'])
        else:
            lr_t = gr['lr']
        with torch.cuda.device(p0.device):
            _lib.check(lib.se_bertadam_step_f32(tab['p'], tab['g'], tab['m'], tab['v'], tab['sizes'], tab['wd'], tab['n'], _lib.ptr(sumsq),
                                                float(lr_t), float(gr['b1']), float(gr['b2']), float(gr['e']), float(gr['max_grad_norm']),
                                                float(global_max_norm), _lib.stream()), 'se_bertadam_step_f32')
        for p, _, st, _ in tab['items']:
            st['step'] += 1
            torch.autograd.graph.increment_version(p)          # written outside autograd's sight: views / engines key on it

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if all(p.is_cuda for group in self.param_groups for p in group['params']):
            tab = self._fused_table()
            if tab is not None:
                self.step_fused(tab, self.grad_sumsq(tab))
                return loss
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
