"""Downstream heads -- drop-in mirror of the reference's model.py (rows C1-C4), backed by libse_amd.so.

Same class names, constructor keywords (every CLI arg is swallowed by **kwargs, run_downstream.py:208-210),
forward(features=, linears=, **kw) -> (predicted (B, T', 201), dict) contract and state_dict keys
(`linear.weight`, `linear.bias`) as model.py, so reference checkpoints ('Downstream', runner.py:131) load.
"""
import torch
import torch.nn as nn

from . import _lib
from .checkpoint import load_checkpoint
from .preprocessor import OnlinePreprocessor
from .transformer import TRANSFORMER, TransformerConfig, TransformerSpecPredictionHead


class _HeadLinearFn(torch.autograd.Function):
    """offset = act(norm(x) W^T + b); predicted = linears * offset.  Forward: se_head_linear_f32,
    backward wrt (W, b): se_head_linear_bwd_f32, from the gradient of `predicted` (L1 / SISDR) and / or of the mask `offset`
    itself (WSD).  Gradients wrt features / linears are not needed by the reference (only downstream parameters are
    optimised, runner.py:110-115) and are not produced."""

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
        ctx.save_for_backward(feats, lin if lin is not None else torch.empty(0, device=feats.device), offset, w)
        ctx.meta = (act, int(cmvn), float(eps), lin is not None, weight.shape)
        return predicted, offset

    @staticmethod
    def backward(ctx, grad_predicted, grad_offset):
        lib = _lib.load()
        feats, lin, offset, w = ctx.saved_tensors
        act, cmvn, eps, has_lin, wshape = ctx.meta
        B, F, D = feats.shape
        N = wshape[0]
        gp = None if grad_predicted is None else grad_predicted.contiguous().float()
        go = None if grad_offset is None else grad_offset.contiguous().float()
        gW = torch.empty(N, D, device=feats.device, dtype=torch.float32)
        gb = torch.empty(N, device=feats.device, dtype=torch.float32)
        nbytes = lib.se_head_workspace_bytes(B, F, D, N)
        ws = torch.empty(nbytes, device=feats.device, dtype=torch.uint8)
        _lib.check(lib.se_head_linear_bwd_f32(_lib.ptr(feats), _lib.ptr(lin) if has_lin else None, _lib.ptr(offset), _lib.ptr(gp), _lib.ptr(go),
                                              B, F, D, N, act, cmvn, eps, _lib.ptr(gW), _lib.ptr(gb), _lib.ptr(ws), nbytes,
                                              _lib.stream()), 'se_head_linear_bwd_f32')
        d_feats = None
        if ctx.needs_input_grad[0]:      # the head sits on a trainable stack (Residual: LSTM below): dx = CMVN'(g_pre W)
            d_feats = torch.empty_like(feats)
            nb = lib.se_head_dx_workspace_bytes(B, F, D, N)
            ws2 = torch.empty(nb, device=feats.device, dtype=torch.uint8)
            _lib.check(lib.se_head_linear_dx_f32(_lib.ptr(feats), _lib.ptr(lin) if has_lin else None, _lib.ptr(offset), _lib.ptr(gp), _lib.ptr(go),
                                                 _lib.ptr(w), B, F, D, N, act, cmvn, eps, _lib.ptr(d_feats), _lib.ptr(ws2), nb, _lib.stream()),
                       'se_head_linear_dx_f32')
        return d_feats, None, gW, gb, None, None, None


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
        needs_grad = torch.is_grad_enabled() and (features.requires_grad or any(p.requires_grad for p in self.parameters()))
        if not needs_grad and features.is_cuda and getattr(self, 'lazy_offset', True):
            # evaluate(): `offset` is only read by the WSD criterion (objective.py:127); every other consumer drops it, so it is a LazyTensor that
            # re-runs the kernel with the mask output when (and only when) something reads it -- the launch then stores one plane instead of two
            return self._forward_lazy(features, linears)
        predicted, offset = _HeadLinearFn.apply(features, linears, self.linear.weight, self.linear.bias, self._act,
                                                self.cmvn, self.eps)
        return predicted, {'offset': offset}

    def _w3(self, dev):
        """the three-term bf16 split of the weights (se_head_split_weights_f32), rebuilt when the parameter changed (in-place update, new storage, device)"""
        import weakref
        w = self.linear.weight
        key = (w.data_ptr(), w._version, str(dev))
        c = getattr(self, '_w3_cache', None)
        # the entry belongs to one Parameter OBJECT (a new parameter may reuse a freed one's address with the same version count)
        if c is None or c[0] != key or c[2]() is not w:
            lib = _lib.load()
            N, D = w.shape
            wf = w.detach().contiguous().float()
            w3 = torch.empty(int(lib.se_head_w3_bytes(N, D)), device=dev, dtype=torch.uint8)
            _lib.check(lib.se_head_split_weights_f32(_lib.ptr(wf), N, D, _lib.ptr(w3), _lib.stream()), 'se_head_split_weights_f32')
            c = self._w3_cache = (key, w3, weakref.ref(w))
        return c[1]

    def _forward_lazy(self, features, linears, sisdr=None):
        """sisdr = (linear_tar, lengths, len_div, eps): evaluate() scores `predicted` with objective.SISDR right after this call (runner.py:575) -- the
        criterion's sums then come out of this launch and the finished loss rides on `predicted` as `_se_sisdr` (objective.SISDR.forward picks it up when it
        is handed this very tensor, unmodified, with this very linear_tar and eps)"""
        from .preprocessor import LazyTensor
        lib = _lib.load()
        B, F, D = features.shape
        N = self.linear.weight.shape[0]
        feats = features.contiguous().float()
        lin = linears.contiguous().float()
        act, cmvn, eps = self._act, int(self.cmvn), float(self.eps)
        # column statistics: handed over by the feature launch (preprocessor.head_stats_eps; every feature row was in LDS there) when they belong to
        # exactly this tensor and this eps -- otherwise one pass over the features here
        stats = None
        if cmvn:
            side = getattr(features, '_se_colstats', None)
            if side is not None and side[1] == eps and side[2] == features._version and feats.data_ptr() == features.data_ptr():
                stats = side[0]
            else:
                stats = torch.empty(B, D, 2, device=feats.device, dtype=torch.float32)
                _lib.check(lib.se_head_colstats_f32(_lib.ptr(feats), B, F, D, eps, _lib.ptr(stats), _lib.stream()), 'se_head_colstats_f32')
        w3 = self._w3(feats.device)
        bias = self.linear.bias.detach().contiguous().float()         # no copy for an fp32 parameter
        wkey = (self.linear.weight.data_ptr(), self.linear.weight._version, self.linear.bias._version)

        def run(want_offset):
            if (self.linear.weight.data_ptr(), self.linear.weight._version, self.linear.bias._version) != wkey:
                raise RuntimeError('LinearResidual: the parameters changed after forward() returned and before its lazy `offset` was read; read it first '
                                   '(or set head.lazy_offset = False)')
            out = torch.empty(B, F, N, device=feats.device, dtype=torch.float32)
            if want_offset:      # the mask itself: the same launch without the noisy-power product
                _lib.check(lib.se_head_linear_pre_f32(_lib.ptr(feats), _lib.ptr(w3), _lib.ptr(bias), None, _lib.ptr(stats), B, F, D, N, act, _lib.ptr(out), None,
                                                      _lib.stream()), 'se_head_linear_pre_f32')
            else:
                _lib.check(lib.se_head_linear_pre_f32(_lib.ptr(feats), _lib.ptr(w3), _lib.ptr(bias), _lib.ptr(lin), _lib.ptr(stats), B, F, D, N, act, _lib.ptr(out), None,
                                                      _lib.stream()), 'se_head_linear_pre_f32')
            return out
        if sisdr is not None:
            tar, lengths, len_div, ceps = sisdr
            tarc = tar.contiguous().float()
            lens = lengths.to(device=feats.device, dtype=torch.int64).contiguous()
            predicted = torch.empty(B, F, N, device=feats.device, dtype=torch.float32)
            scratch = torch.empty(int(lib.se_head_sisdr_scratch_doubles(B, F, N)) + 2, device=feats.device, dtype=torch.float64)
            loss_b = torch.empty(B, device=feats.device, dtype=torch.float32)
            loss = torch.empty((), device=feats.device, dtype=torch.float32)
            _lib.check(lib.se_head_linear_sisdr_f32(_lib.ptr(feats), _lib.ptr(w3), _lib.ptr(bias), _lib.ptr(lin), _lib.ptr(stats), B, F, D, N, act, _lib.ptr(predicted),
                                                    None, _lib.ptr(tarc), _lib.ptr(lens), int(len_div), float(ceps), scratch[2:].data_ptr(), _lib.ptr(loss_b),
                                                    scratch.data_ptr(), _lib.ptr(loss), _lib.stream()), 'se_head_linear_sisdr_f32')
            predicted._se_sisdr = (loss, predicted._version, tar, float(ceps))
        else:
            predicted = run(False)
        return predicted, {'offset': LazyTensor((B, F, N), feats.device, lambda: run(True))}

    def enhance_scored(self, features, linears, linear_tar, lengths, len_div, eps):
        """forward() for evaluate() when objective.SISDR(eps) scores the result next: same (predicted, {'offset': lazy}) -- see _forward_lazy"""
        # the hand-over to the criterion is tied to `predicted`'s version counter: tensors made under torch.inference_mode() have none -> the plain call
        if torch.is_grad_enabled() or torch.is_inference_mode_enabled() or not features.is_cuda or not getattr(self, 'lazy_offset', True):
            return self.forward(features, linears)
        return self._forward_lazy(features, linears, sisdr=(linear_tar, lengths, len_div, eps))


class SpecHead(nn.Module):
    """model.py:94-126: TransformerSpecPredictionHead from an upstream checkpoint + exp / log / activation epilogue,
    fused into one se_spechead_fwd_bf16 call."""

    def __init__(self, output_size, ckpt, activation='ReLU', random_init=False, eps=1e-6, **kwargs):
        super().__init__()
        assert ckpt != ''
        ckpt = load_checkpoint(ckpt) if isinstance(ckpt, str) else ckpt
        trans_config = TransformerConfig(ckpt['Settings']['Config'])
        trans_spechead = TransformerSpecPredictionHead(trans_config, output_size)
        trans_spechead.load_state_dict(ckpt['SpecHead'])
        assert trans_spechead.output.out_features == output_size
        self.spechead = trans_spechead
        self.eps = eps
        target_config = ckpt['Settings']['Config']['online']['target']
        self.log = False if 'log' not in target_config else target_config['log']
        self.activation = activation
        _act_id(activation)
        if random_init:
            for param in self.parameters():
                if param.dim() >= 2:
                    nn.init.xavier_uniform_(param.data)
                else:
                    nn.init.constant_(param.data, 0)

    def forward(self, features, **kwargs):
        sh = self.spechead
        if sh._needs_grad():
            # fine-tuning the head on (frozen) upstream features: forward keeps intermediates, backward on the HIP kernels
            from .spechead_train import SpecHeadTrainFn
            predicted, log_predicted = SpecHeadTrainFn.apply(features, sh.dense.weight, sh.dense.bias, sh.LayerNorm.weight, sh.LayerNorm.bias,
                                                             sh.output.weight, sh.output.bias, sh.LayerNorm.variance_epsilon, self.log,
                                                             _act_id(self.activation), self.eps)
            return predicted, {'log_predicted': log_predicted}
        predicted, log_predicted = sh._engine.spechead(sh, None, features, mode='full', log_target=self.log, act=self.activation, eps=self.eps)
        return predicted, {'log_predicted': log_predicted}


class Mockingjay(nn.Module):
    """model.py:129-171: the whole upstream encoder + spec head as the downstream model (ignores `linears`).
    With gradients enabled both parts run their training paths (activations kept, backward on the HIP kernels)."""

    def __init__(self, dckpt, activation='ReLU', eps=1e-6, **kwargs):
        super().__init__()
        options = {'ckpt_file': dckpt, 'load_pretrain': 'True', 'no_grad': 'False', 'dropout': 'default', 'spec_aug': 'False',
                   'spec_aug_prev': 'True', 'weighted_sum': 'False', 'select_layer': -1, 'permute_input': 'False'}
        ckpt = load_checkpoint(dckpt)
        pretrain_config = ckpt['Settings']['Config']
        online = pretrain_config['online']
        # feature dims without running the (GPU-only) preprocessor: D = base * (1 + delta)
        def feat_dim(cfg):
            base = {'mel': online.get('n_mels', 40), 'linear': online.get('n_freq', 201), 'phase': online.get('n_freq', 201),
                    'complx': 2 * online.get('n_freq', 201)}[cfg['feat_type']]
            return base * (1 + int(cfg.get('delta', 0)))
        inp_dim, tar_dim = feat_dim(online['input']), feat_dim(online['target'])
        self.mockingjay = TRANSFORMER(options, inp_dim)
        trans_config = TransformerConfig(pretrain_config)
        trans_spechead = TransformerSpecPredictionHead(trans_config, tar_dim)
        trans_spechead.load_state_dict(ckpt['SpecHead'])
        assert trans_spechead.output.out_features == tar_dim
        self.spechead = trans_spechead
        self.eps = eps
        target_config = online['target']
        self.log = False if 'log' not in target_config else target_config['log']
        self.activation = activation
        _act_id(activation)

    def forward(self, features, **kwargs):
        features = self.mockingjay(features)
        sh = self.spechead
        if sh._needs_grad() or features.requires_grad:
            from .spechead_train import SpecHeadTrainFn
            predicted, log_predicted = SpecHeadTrainFn.apply(features, sh.dense.weight, sh.dense.bias, sh.LayerNorm.weight, sh.LayerNorm.bias,
                                                             sh.output.weight, sh.output.bias, sh.LayerNorm.variance_epsilon, self.log,
                                                             _act_id(self.activation), self.eps)
            return predicted, {'log_predicted': log_predicted}
        predicted, log_predicted = sh._engine.spechead(sh, None, features, mode='full', log_target=self.log, act=self.activation, eps=self.eps)
        return predicted, {'log_predicted': log_predicted}
