"""Downstream heads -- drop-in mirror of the reference's model.py (rows C1-C4), backed by libse_amd.so.

Same class names, constructor keywords (every CLI arg is swallowed by **kwargs, run_downstream.py:208-210),
forward(features=, linears=, **kw) -> (predicted (B, T', 201), dict) contract and state_dict keys
(`linear.weight`, `linear.bias`) as model.py, so reference checkpoints ('Downstream', runner.py:131) load.
"""
import torch
import torch.nn as nn

from . import _lib


class _HeadLinearFn(torch.autograd.Function):
    """offset = act(norm(x) W^T + b); predicted = linears * offset.  Forward: se_head_linear_f32,
    backward wrt (W, b): se_head_linear_bwd_f32.  Gradients wrt features / linears are not needed by the
    reference (only downstream parameters are optimised, runner.py:110-115) and are not produced."""

    @staticmethod
    def forward(ctx, feats, linears, weight, bias, act, cmvn, eps):
        lib = _lib.load()
        B, F, D = feats.shape
        N = weight.shape[0]
        feats = feats.contiguous().float()
        lin = None if linears is None else linears.contiguous().float()
        w, b = weight.contiguous().float(), bias.contiguous().float()
        predicted = torch.empty(B, F, N, device=feats.device, dtype=torch.float32)
        offset = torch.empty(B, F, N, device=feats.device, dtype=torch.float32)
        nbytes = lib.se_head_workspace_bytes(B, F, D, N)
        ws = torch.empty(nbytes, device=feats.device, dtype=torch.uint8)
        _lib.check(lib.se_head_linear_f32(_lib.ptr(feats), _lib.ptr(w), _lib.ptr(b), _lib.ptr(lin), B, F, D, N, act,
                                          int(cmvn), float(eps), _lib.ptr(predicted), _lib.ptr(offset), _lib.ptr(ws), nbytes,
                                          _lib.stream()), 'se_head_linear_f32')
        ctx.save_for_backward(feats, lin if lin is not None else torch.empty(0, device=feats.device), offset)
        ctx.meta = (act, int(cmvn), float(eps), lin is not None, weight.shape)
        ctx.mark_non_differentiable(offset)
        return predicted, offset

    @staticmethod
    def backward(ctx, grad_predicted, _grad_offset):
        lib = _lib.load()
        feats, lin, offset = ctx.saved_tensors
        act, cmvn, eps, has_lin, wshape = ctx.meta
        B, F, D = feats.shape
        N = wshape[0]
        gp = grad_predicted.contiguous().float()
        gW = torch.empty(N, D, device=feats.device, dtype=torch.float32)
        gb = torch.empty(N, device=feats.device, dtype=torch.float32)
        nbytes = lib.se_head_workspace_bytes(B, F, D, N)
        ws = torch.empty(nbytes, device=feats.device, dtype=torch.uint8)
        _lib.check(lib.se_head_linear_bwd_f32(_lib.ptr(feats), _lib.ptr(lin) if has_lin else None, _lib.ptr(offset), _lib.ptr(gp),
                                              B, F, D, N, act, cmvn, eps, _lib.ptr(gW), _lib.ptr(gb), _lib.ptr(ws), nbytes,
                                              _lib.stream()), 'se_head_linear_bwd_f32')
        return None, None, gW, gb, None, None, None


def _act_id(name):
    if name not in _lib.SE_ACT:
        raise NotImplementedError(f'activation nn.{name} is not supported by the HIP head kernel '
                                  f'(supported: {sorted(_lib.SE_ACT)})')
    return _lib.SE_ACT[name]


class Linear(nn.Module):
    """model.py:8-17"""

    def __init__(self, input_dim, output_dim, activation='ReLU', **kwargs):
        super().__init__()
        self.linear = nn.Linear(input_dim, output_dim)
        self.activation = activation
        self._act = _act_id(activation)

    def forward(self, features, **kwargs):
        predicted, _ = _HeadLinearFn.apply(features, None, self.linear.weight, self.linear.bias, self._act, False, 0.0)
        return predicted, {}


class LinearResidual(nn.Module):
    """model.py:20-34"""

    def __init__(self, input_size=201, output_size=201, activation='Sigmoid', cmvn=True, eps=1e-6, **kwargs):
        super().__init__()
        self.linear = nn.Linear(input_size, output_size)
        self.activation = activation
        self._act = _act_id(activation)
        self.cmvn = cmvn
        self.eps = eps

    def forward(self, features, linears, **kwargs):
        predicted, offset = _HeadLinearFn.apply(features, linears, self.linear.weight, self.linear.bias, self._act,
                                                self.cmvn, self.eps)
        return predicted, {'offset': offset}
