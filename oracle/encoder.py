"""Oracle rows B1-B4: S3PRL ``TRANSFORMER`` (BERT-style encoder) and ``TransformerSpecPredictionHead``
restated functionally on PyTorch-CPU (test infrastructure only).  PARITY UNPINNED vs original S3PRL
(the source is absent; see oracle/__init__.py).  Anchors in the reference: options dict model.py:132-141,
architecture hyper-parameters config/pretrain_sample.yaml:1-22, spec head returns a 2-tuple and has
`.output` (model.py:103,120), checkpoint keys 'Transformer' / 'SpecHead' (model.py:101,153).

Weights are a flat dict using S3PRL's state_dict key names, e.g.
  input_representations.spec_transform.{weight,bias}, input_representations.LayerNorm.{weight,bias},
  encoder.layer.{i}.attention.self.{query,key,value}.{weight,bias},
  encoder.layer.{i}.attention.output.{dense,LayerNorm}.{weight,bias},
  encoder.layer.{i}.intermediate.dense.{weight,bias}, encoder.layer.{i}.output.{dense,LayerNorm}.{weight,bias}
and for the head: dense.*, LayerNorm.*, output.*
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


class Config:
    """TransformerConfig(config_dict) restated: reads config['transformer'] (pretrain_sample.yaml:1-22)."""

    def __init__(self, config):
        t = config['transformer'] if 'transformer' in config else config
        self.downsample_rate = int(t.get('downsample_rate', 1))
        self.hidden_size = int(t.get('hidden_size', 768))
        self.num_hidden_layers = int(t.get('num_hidden_layers', 6))
        self.num_attention_heads = int(t.get('num_attention_heads', 12))
        self.intermediate_size = int(t.get('intermediate_size', 3072))
        self.hidden_act = t.get('hidden_act', 'gelu')
        self.layer_norm_eps = float(t.get('layer_norm_eps', 1e-12))
        self.initializer_range = float(t.get('initializer_range', 0.02))


def gelu(x):
    # BERT gelu: x * 0.5 * (1 + erf(x / sqrt(2)))
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def layer_norm(x, w, b, eps):
    # TransformerLayerNorm: TF style, epsilon inside the square root, biased variance
    u = x.mean(-1, keepdim=True)
    s = (x - u).pow(2).mean(-1, keepdim=True)
    return w * ((x - u) / torch.sqrt(s + eps)) + b


def position_encoding(seq_len, hidden_size, dtype=torch.float32):
    """Sinusoid table: angle(pos, j) = pos / 10000^(2*(j//2)/hidden); even j -> sin, odd j -> cos."""
    pos = np.arange(seq_len, dtype=np.float64)[:, None]
    j = np.arange(hidden_size, dtype=np.float64)[None, :]
    table = pos / np.power(10000.0, 2.0 * np.floor(j / 2.0) / hidden_size)
    table[:, 0::2] = np.sin(table[:, 0::2])
    table[:, 1::2] = np.cos(table[:, 1::2])
    return torch.from_numpy(table).to(dtype)


def valid_lengths(feats):
    """process_input_data: number of frames whose feature sum is non-zero; frames past it are masked."""
    return (feats.sum(dim=-1) != 0).sum(dim=-1)


def init_weights(cfg, inp_dim, seed=0, spec_out=None, dtype=torch.float32):
    """Seeded random weights at the real sizes (no checkpoints exist offline): N(0, initializer_range) for
    matrices, zeros for biases, LayerNorm weight 1 -- BERT's init_Transformer_weights."""
    g = torch.Generator().manual_seed(seed)
    H, I = cfg.hidden_size, cfg.intermediate_size
    sd = {}

    def lin(name, o, i):
        sd[name + '.weight'] = (torch.randn(o, i, generator=g) * cfg.initializer_range).to(dtype)
        sd[name + '.bias'] = (torch.randn(o, generator=g) * cfg.initializer_range).to(dtype)

    def ln(name):
        sd[name + '.weight'] = (1.0 + 0.1 * torch.randn(H, generator=g)).to(dtype)
        sd[name + '.bias'] = (0.1 * torch.randn(H, generator=g)).to(dtype)

    lin('input_representations.spec_transform', H, inp_dim * cfg.downsample_rate)
    ln('input_representations.LayerNorm')
    for i in range(cfg.num_hidden_layers):
        p = f'encoder.layer.{i}.'
        for n in ('query', 'key', 'value'):
            lin(p + 'attention.self.' + n, H, H)
        lin(p + 'attention.output.dense', H, H)
        ln(p + 'attention.output.LayerNorm')
        lin(p + 'intermediate.dense', I, H)
        lin(p + 'output.dense', H, I)
        ln(p + 'output.LayerNorm')
    head = None
    if spec_out is not None:
        head = {}
        head['dense.weight'] = (torch.randn(H, H, generator=g) * cfg.initializer_range).to(dtype)
        head['dense.bias'] = (torch.randn(H, generator=g) * cfg.initializer_range).to(dtype)
        head['LayerNorm.weight'] = (1.0 + 0.1 * torch.randn(H, generator=g)).to(dtype)
        head['LayerNorm.bias'] = (0.1 * torch.randn(H, generator=g)).to(dtype)
        head['output.weight'] = (torch.randn(spec_out * cfg.downsample_rate, H, generator=g) * cfg.initializer_range).to(dtype)
        head['output.bias'] = (torch.randn(spec_out * cfg.downsample_rate, generator=g) * cfg.initializer_range).to(dtype)
    return sd, head


def _lowbias32(x):
    """the 32-bit mixer of csrc/dropout.h on int64 tensors holding uint32 values"""
    M = 0xffffffff
    x = x & M
    x = x ^ (x >> 16)
    x = (x * 0x7feb352d) & M
    x = x ^ (x >> 15)
    x = (x * 0x846ca68b) & M
    x = x ^ (x >> 16)
    return x


def _mix24(x):
    """the per-pair mixer of csrc/dropout.h (lowbias32 structure with 24-bit multiplies) on int64 tensors holding uint32 values"""
    M = 0xffffffff
    x = x & M
    x = x ^ (x >> 16)
    x = ((x & 0xffffff) * 0x7feb35) & M
    x = x ^ (x >> 15)
    x = ((x & 0xffffff) * 0x46ca6b) & M
    x = x ^ (x >> 16)
    return x


def dropout_key(seed, site):
    M = 0xffffffff
    v = ((seed & M) * 0x9E3779B9 + ((seed >> 32) & M) * 0x85EBCA6B + site) & M
    k = int(_lowbias32(torch.tensor([v], dtype=torch.int64))[0])
    return k if k else 1


def keep_mask(seed, site, rows, cols, p, pairs_per_row=None):
    """Dropout keep-mask of a (rows, cols) site, bit for bit what csrc/dropout.h generates: one 32-bit hash per element
    pair (pair index = row * pairs_per_row + col // 2, pairs_per_row = ceil(cols / 2)), 16 bits per element, keep iff
    bits >= round(p * 65536).  Returns a bool tensor (rows, cols)."""
    ppr = (cols + 1) // 2 if pairs_per_row is None else pairs_per_row
    thr = int(p * 65536.0 + 0.5)
    key = dropout_key(seed, site)
    r = torch.arange(rows, dtype=torch.int64)[:, None]
    c = torch.arange(cols, dtype=torch.int64)[None, :]
    pair = (r * ppr + (c >> 1)) & 0xffffffff
    bits = _mix24(pair ^ key)
    v = torch.where((c & 1) == 1, bits >> 16, bits & 0xffff)
    return v >= thr


def dropout_site(layer, which):
    return 4 * layer + which


def encoder_forward(feats, sd, cfg, lengths=None, all_layers=False, dropout_p=0.0, seed=0):
    """B1-B3: feats (B, T, D) fp32 -> last hidden (B, T, H)  (select_layer -1).  dropout_p = 0: eval mode.
    dropout_p > 0: training mode with the BERT dropout sites (after the input LayerNorm, attention probabilities, after the
    attention-output and FFN-output dense) and the counter-based masks of keep_mask(seed, site, ...).
    Attention mask: additive (1-mask)*-10000 on key positions >= valid length."""
    B, T, _ = feats.shape
    H, nh = cfg.hidden_size, cfg.num_attention_heads
    dh = H // nh
    if lengths is None:
        lengths = valid_lengths(feats)
    key_mask = (torch.arange(T)[None, :] < lengths[:, None]).to(feats.dtype)       # (B, T)
    ext = (1.0 - key_mask)[:, None, None, :] * -10000.0
    x = F.linear(feats, sd['input_representations.spec_transform.weight'], sd['input_representations.spec_transform.bias'])
    x = x + position_encoding(T, H, feats.dtype)[None]
    x = layer_norm(x, sd['input_representations.LayerNorm.weight'], sd['input_representations.LayerNorm.bias'], cfg.layer_norm_eps)
    drop = dropout_p > 0
    scale = 1.0 / (1.0 - dropout_p) if drop else 1.0

    def hidden_drop(t, site):          # t (B, T, H): element index = (b T + t) H + col
        m = keep_mask(seed, site, B * T, H, dropout_p).reshape(B, T, H)
        return t * m.to(t.dtype) * scale
    if drop:
        x = hidden_drop(x, dropout_site(cfg.num_hidden_layers, 0))
    outs = []
    for i in range(cfg.num_hidden_layers):
        p = f'encoder.layer.{i}.'
        q = F.linear(x, sd[p + 'attention.self.query.weight'], sd[p + 'attention.self.query.bias'])
        k = F.linear(x, sd[p + 'attention.self.key.weight'], sd[p + 'attention.self.key.bias'])
        v = F.linear(x, sd[p + 'attention.self.value.weight'], sd[p + 'attention.self.value.bias'])
        q = q.view(B, T, nh, dh).permute(0, 2, 1, 3)
        k = k.view(B, T, nh, dh).permute(0, 2, 1, 3)
        v = v.view(B, T, nh, dh).permute(0, 2, 1, 3)
        scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh) + ext
        probs = torch.softmax(scores, dim=-1)
        if drop:                       # row id = (b heads + head) T + q, pair = row_id * ceil(T/2) + key // 2
            m = keep_mask(seed, dropout_site(i, 0), B * nh * T, T, dropout_p).reshape(B, nh, T, T)
            probs = probs * m.to(probs.dtype) * scale
        ctx = torch.matmul(probs, v).permute(0, 2, 1, 3).reshape(B, T, H)
        a = F.linear(ctx, sd[p + 'attention.output.dense.weight'], sd[p + 'attention.output.dense.bias'])
        if drop:
            a = hidden_drop(a, dropout_site(i, 1))
        x = layer_norm(a + x, sd[p + 'attention.output.LayerNorm.weight'], sd[p + 'attention.output.LayerNorm.bias'], cfg.layer_norm_eps)
        h = gelu(F.linear(x, sd[p + 'intermediate.dense.weight'], sd[p + 'intermediate.dense.bias']))
        o = F.linear(h, sd[p + 'output.dense.weight'], sd[p + 'output.dense.bias'])
        if drop:
            o = hidden_drop(o, dropout_site(i, 2))
        x = layer_norm(o + x, sd[p + 'output.LayerNorm.weight'], sd[p + 'output.LayerNorm.bias'], cfg.layer_norm_eps)
        outs.append(x)
    return outs if all_layers else x


def spec_head_forward(hidden, head, cfg):
    """B4: TransformerSpecPredictionHead: dense -> act -> LayerNorm -> output; returns (pred, hidden)."""
    h = F.linear(hidden, head['dense.weight'], head['dense.bias'])
    h = gelu(h)
    h = layer_norm(h, head['LayerNorm.weight'], head['LayerNorm.bias'], cfg.layer_norm_eps)
    return F.linear(h, head['output.weight'], head['output.bias']), h
