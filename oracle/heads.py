"""Oracle rows C1-C4: the reference's downstream heads restated functionally (test infrastructure only).
C1/C2 are PINNED against the reference's own model.py via tests/golden/reference_golden.npz."""
import torch
import torch.nn.functional as F

from . import encoder as _enc

_ACT = {'ReLU': torch.relu, 'Sigmoid': torch.sigmoid, 'Identity': lambda x: x, 'Tanh': torch.tanh}


def linear_head(features, weight, bias, activation='ReLU'):
    """C2: model.py:14-17  act(Linear(x)) -> (predicted, {})."""
    return _ACT[activation](F.linear(features, weight, bias)), {}


def linear_residual(features, linears, weight, bias, activation='Sigmoid', cmvn=True, eps=1e-6):
    """C1: model.py:28-34  optional CMVN over time (dim=1, unbiased std, +eps outside), Linear, act,
    predicted = linears * offset  (mask x noisy POWER)."""
    if cmvn:
        features = (features - features.mean(dim=1, keepdim=True)) / (features.std(dim=1, keepdim=True) + eps)
    offset = _ACT[activation](F.linear(features, weight, bias))
    return linears * offset, {'offset': offset}


def spec_head(hidden, head, cfg, log=True, activation='ReLU', eps=1e-6):
    """C3: model.py:119-126  B4 then exp (if the pre-training target was log-scale) / log, then act."""
    predicted, _ = _enc.spec_head_forward(hidden, head, cfg)
    if log:
        predicted, log_predicted = predicted.exp(), predicted
    else:
        log_predicted = (predicted + eps).log()
    return _ACT[activation](predicted), {'log_predicted': log_predicted}


def mockingjay(features, sd, head, cfg, log=True, activation='ReLU', eps=1e-6, lengths=None):
    """C4: model.py:163-171  encoder (B1-B3) then C3; ignores `linears`."""
    hidden = _enc.encoder_forward(features, sd, cfg, lengths=lengths)
    return spec_head(hidden, head, cfg, log=log, activation=activation, eps=eps)
