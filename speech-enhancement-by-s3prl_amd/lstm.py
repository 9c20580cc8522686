"""(Bi)LSTM downstream heads on the MI355X kernels: `LSTM` (model.py:37-59) -- the head the
reference's scripts train (run_active.sh `--downstream LSTM`, pseudo_noise.yaml:50-53: hidden 256, 3 layers, bidirectional).

Parameter names / shapes are nn.LSTM's (`lstm.weight_ih_l0`, `lstm.weight_hh_l0_reverse`, ...) and `scaling_layer.0.*`, so the
reference's `--dckpt` checkpoints load unchanged.  The kernels are built for 256 hidden units per direction (W_hh of one direction
resident in one CU's registers + LDS as MFMA fragments); a smaller hidden_size -- the reference's class default is 201, model.py:38 -- runs
on the same kernels with every weight zero-padded to 256 units per gate, which is EXACT: a padded unit has zero pre-activations,
so c = 0.5 c + 0.5 tanh(0) stays 0, h = 0.5 tanh(0) = 0, and its gradients vanish with its (zero) outgoing weights.
hidden_size > 256 is not built.  Forward and backward run on libse_amd.so: input projections, the output linear and every gradient
GEMM on the bf16 GEMM / TN weight-gradient kernels, the recurrence on se_lstm_fwd_bf16 / se_lstm_bwd_bf16.  No CPU fallback."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from . import spechead_train as st
from .heads import _HeadLinearFn, _act_id

_H = 256


def _pad8k(d):          # GEMM reduction dims are multiples of 64
    return (d + 63) // 64 * 64


def _pad_gates(w, hs):
    """(4 hs, ...) gate-blocked rows (i, f, g, o) -> (4 * 256, ...): gate g's rows at [256 g, 256 g + hs), zeros elsewhere."""
    if hs == _H:
        return w
    out = w.new_zeros((4 * _H,) + tuple(w.shape[1:]))
    for g in range(4):
        out[g * _H: g * _H + hs] = w[g * hs: (g + 1) * hs]
    return out


def _unpad_gates(g, hs):
    """inverse of _pad_gates on dim -2 (weights: (..., 1024, K)) or -1 (biases: (..., 1024)) chosen by `g.shape`"""
    if hs == _H:
        return g
    return torch.cat([g[..., k * _H: k * _H + hs, :] for k in range(4)], dim=-2)


def _unpad_gate_vec(g, hs):
    if hs == _H:
        return g
    return torch.cat([g[..., k * _H: k * _H + hs] for k in range(4)], dim=-1)


def _pad_dirs(w, hs, ndir):
    """columns of a layer > 0 input weight: (.., ndir * hs) -> (.., ndir * 256), direction d's block at [256 d, 256 d + hs)"""
    if hs == _H:
        return w
    out = w.new_zeros(tuple(w.shape[:-1]) + (ndir * _H,))
    for d in range(ndir):
        out[..., d * _H: d * _H + hs] = w[..., d * hs: (d + 1) * hs]
    return out


def _unpad_dirs(g, hs, ndir):
    if hs == _H:
        return g
    return torch.cat([g[..., d * _H: d * _H + hs] for d in range(ndir)], dim=-1)


def _pack_hh_fwd(w_hh):
    """(1024, 256) fp32 -> bf16 row-major: the forward kernel reads its MFMA A-fragments straight from nn.LSTM's layout."""
    return w_hh.detach().to(torch.bfloat16).contiguous()


def _pack_hh_bwd(w_hh):
    """(1024, 256) fp32 -> (256, 1024) bf16 = W_hh^T, the backward kernel's A operand."""
    return w_hh.detach().to(torch.bfloat16).t().contiguous()


class _LSTMFn(torch.autograd.Function):
    """nn.LSTM(batch_first=True) forward / backward.  apply(x, num_layers, ndir, *weights) with weights in nn.LSTM's
    parameter order (per layer: w_ih, w_hh, b_ih, b_hh [, the same four for the reverse direction])."""

    @staticmethod
    def forward(ctx, x, num_layers, ndir, *weights):
        lib = _lib.load()
        if not x.is_cuda:
            raise _lib.SEError('LSTM heads run on MI355X only (no CPU fallback)')
        B, T, D = x.shape
        M = B * T
        dev = x.device
        inp = x.reshape(M, D).float()
        saved = []
        Dp = _pad8k(D)
        inp16 = F.pad(inp, (0, Dp - D)).to(torch.bfloat16).contiguous()          # layer-0 operand, zero padded to the GEMM K granule
        for l in range(num_layers):
            ws = weights[4 * ndir * l: 4 * ndir * (l + 1)]
            K = inp16.shape[1]
            xproj = torch.empty(ndir, M, 4 * _H, device=dev, dtype=torch.float32)
            wp = torch.empty(ndir, 4 * _H, _H, device=dev, dtype=torch.bfloat16)
            for d in range(ndir):
                w_ih, w_hh, b_ih, b_hh = ws[4 * d: 4 * d + 4]
                hs = w_hh.shape[1]
                if hs > _H or w_hh.shape[0] != 4 * hs:
                    raise NotImplementedError(f'the LSTM kernels are built for hidden_size <= {_H} (pseudo_noise.yaml:50-58), got {tuple(w_hh.shape)}')
                wi = w_ih.detach().float()
                if l > 0:
                    wi = _pad_dirs(wi, hs, ndir)
                w16 = F.pad(_pad_gates(wi, hs), (0, K - wi.shape[1])).to(torch.bfloat16).contiguous()
                bias = _pad_gates((b_ih.detach() + b_hh.detach()).float(), hs).contiguous()
                w_hh = F.pad(_pad_gates(w_hh.detach().float(), hs), (0, _H - hs))
                _lib.check(lib.se_gemm_bf16(_lib.ptr(inp16), K, _lib.ptr(w16), K, _lib.ptr(bias), None, M, 4 * _H, K, 0, None,
                                            _lib.ptr(xproj) + d * M * 4 * _H * 4, 4 * _H, _lib.stream()), 'se_gemm_bf16')
                wp[d] = _pack_hh_fwd(w_hh)
            h16 = torch.empty(M, ndir * _H, device=dev, dtype=torch.bfloat16)
            gates = torch.empty(ndir, M, 4 * _H, device=dev, dtype=torch.float32)
            cst = torch.empty(ndir, M, _H, device=dev, dtype=torch.float32)
            _lib.check(lib.se_lstm_fwd_bf16(_lib.ptr(wp), _lib.ptr(xproj), B, T, ndir, _lib.ptr(h16), _lib.ptr(gates), _lib.ptr(cst), _lib.stream()),
                       'se_lstm_fwd_bf16')
            saved.append((inp16, h16, gates, cst))
            inp16 = h16
        ctx.saved = saved
        ctx.meta = (B, T, D, num_layers, ndir)
        ctx.weights = weights
        # (B, T, ndir * 256) in the PADDED layout (hidden_size < 256: columns [256 d + hs, 256 (d + 1)) are exactly zero); the heads
        # pad the scaling layer's weight columns to match (_pad_dirs), so no 201-wide GEMM operand ever exists
        return inp16.float().view(B, T, ndir * _H)

    @staticmethod
    def backward(ctx, d_out):
        grads = _lstm_backward(ctx.saved, ctx.meta, ctx.weights, d_out, per_utterance=False)
        ctx.saved = None
        return (None, None, None) + tuple(grads)


def _wgrad(dY16, X16, N, K, groups, rows):
    """dW = dY^T X over all rows (groups == 0) or per group of `rows` rows (per-utterance slabs of the TN kernel: (groups, N, K))."""
    if not groups:
        return st.wgrad_tn(dY16, X16, N, K)
    lib = _lib.load()
    slabs = torch.empty(groups, N, K, device=dY16.device, dtype=torch.float32)
    _lib.check(lib.se_wgrad_tn_slabs_bf16(_lib.ptr(dY16), dY16.shape[1], _lib.ptr(X16), X16.shape[1], dY16.shape[0], N, K, rows, _lib.ptr(slabs), _lib.stream()),
               'se_wgrad_tn_slabs_bf16')
    return slabs


def _colsum(x, cols, groups, rows):
    lib = _lib.load()
    is16 = x.dtype == torch.bfloat16
    if not groups:
        out = torch.empty(cols, device=x.device, dtype=torch.float32)
        if is16:
            _lib.check(lib.se_colsum_bf16(_lib.ptr(x), x.shape[0], cols, x.shape[1], _lib.ptr(out), _lib.stream()), 'se_colsum_bf16')
        else:
            _lib.check(lib.se_colsum_f32(_lib.ptr(x), x.shape[0], cols, x.shape[1], _lib.ptr(out), 0, _lib.stream()), 'se_colsum_f32')
        return out
    out = torch.empty(groups, cols, device=x.device, dtype=torch.float32)
    _lib.check(lib.se_colsum_groups(_lib.ptr(x), int(is16), groups, rows, cols, x.shape[1], _lib.ptr(out), _lib.stream()), 'se_colsum_groups')
    return out


def _lstm_backward(saved, meta, weights, d_out, per_utterance=False):
    """BPTT of the stacked (Bi)LSTM.  per_utterance: every reduction over the B T rows (weight gradients, biases) is taken per utterance
    instead -- the recurrence is per utterance anyway, so ONE sweep yields all B parameter gradients (active-sampling scoring,
    sampler.py:59-110, replaces B sequential backward passes); gradients then carry a leading B dimension."""
    lib = _lib.load()
    B, T, D, num_layers, ndir = meta
    M = B * T
    G = B if per_utterance else 0
    dev = d_out.device
    hs = weights[1].shape[1]                     # hidden units of the module (<= 256: the kernels run on zero-padded weights)
    dh = d_out.reshape(M, ndir * _H).contiguous().float()
    grads = [None] * len(weights)
    for l in range(num_layers - 1, -1, -1):
        inp16, h16, gates, cst = saved[l]
        ws = weights[4 * ndir * l: 4 * ndir * (l + 1)]
        K = inp16.shape[1]
        wq = torch.stack([_pack_hh_bwd(F.pad(_pad_gates(ws[4 * d + 1].detach().float(), hs), (0, _H - hs))) for d in range(ndir)]).contiguous()
        dg = torch.empty(ndir, M, 4 * _H, device=dev, dtype=torch.bfloat16)
        _lib.check(lib.se_lstm_bwd_bf16(_lib.ptr(wq), _lib.ptr(gates), _lib.ptr(cst), _lib.ptr(dh), ndir * _H, B, T, ndir, _lib.ptr(dg),
                                        _lib.stream()), 'se_lstm_bwd_bf16')
        hv = h16.view(B, T, ndir * _H)
        dx = None
        for d in range(ndir):
            w_ih, w_hh, b_ih, b_hh = ws[4 * d: 4 * d + 4]
            dgd = dg[d]                                                      # (M, 1024) bf16
            # h of the previous step (in this direction's order), zero at the sequence start
            hprev = torch.zeros(B, T, _H, device=dev, dtype=torch.bfloat16)
            if d == 0:
                hprev[:, 1:] = hv[:, :-1, :_H]
            else:
                hprev[:, :-1] = hv[:, 1:, _H:]
            g_ih = _wgrad(dgd, inp16, 4 * _H, K, G, T)
            g_ih = (_unpad_dirs(g_ih, hs, ndir) if l > 0 else g_ih[..., :w_ih.shape[1]])
            g_ih = _unpad_gates(g_ih, hs).contiguous()
            g_hh = _unpad_gates(_wgrad(dgd, hprev.view(M, _H), 4 * _H, _H, G, T)[..., :hs], hs).contiguous()
            g_b = _unpad_gate_vec(_colsum(dgd, 4 * _H, G, T), hs).contiguous()
            base = 4 * ndir * l + 4 * d
            grads[base], grads[base + 1], grads[base + 2], grads[base + 3] = g_ih, g_hh, g_b, g_b.clone()
            if l > 0:     # dx = sum over directions of dgates . W_ih ; the second direction rides the GEMM's residual input
                Din = ndir * _H
                wt = st.transpose_f32_bf16(_pad_gates(_pad_dirs(w_ih.detach().float(), hs, ndir), hs).contiguous(), 4 * _H)      # (Din, 1024) = W_ih^T
                nxt = torch.empty(M, Din, device=dev, dtype=torch.float32)
                _lib.check(lib.se_gemm_bf16(_lib.ptr(dgd), 4 * _H, _lib.ptr(wt), 4 * _H, None, _lib.ptr(dx), M, Din, 4 * _H, 0, None, _lib.ptr(nxt),
                                            Din, _lib.stream()), 'se_gemm_bf16')
                dx = nxt
        dh = dx
    return grads


class _DenseLogExpFn(torch.autograd.Function):
    """scaling_layer (Linear + activation) + LSTM.forward's epilogue (model.py:56-58): log_predicted = act(x W^T + b),
    predicted = exp(log_predicted); bf16 GEMM, HIP backward incl. the gradient wrt x."""

    @staticmethod
    def forward(ctx, x, w, b, act=0):
        lib = _lib.load()
        lead, K = x.shape[:-1], x.shape[-1]
        N = w.shape[0]
        x16 = st.cast_bf16(x.reshape(-1, K))
        M = x16.shape[0]
        p = st._gemm(x16, st.cast_bf16(w.detach()), b.detach().contiguous().float(), M, N, K)
        pred = torch.empty(M, N, device=x.device, dtype=torch.float32)
        if act == _lib.SE_ACT['Identity']:
            logp = p
            _lib.check(lib.se_spec_epilogue_f32(_lib.ptr(p), M * N, 1, act, 0.0, _lib.ptr(pred), None, _lib.stream()), 'se_spec_epilogue_f32')
        else:       # mode 2: log_predicted = act(p), predicted = exp(log_predicted)
            logp = torch.empty_like(p)
            _lib.check(lib.se_spec_epilogue_f32(_lib.ptr(p), M * N, 2, act, 0.0, _lib.ptr(pred), _lib.ptr(logp), _lib.stream()), 'se_spec_epilogue_f32')
        ctx.save_for_backward(x16, p, w)
        ctx.lead, ctx.act = lead, act
        return pred.view(*lead, N), logp.view(*lead, N)

    @staticmethod
    def backward(ctx, d_pred, d_logp):
        x16, p, w = ctx.saved_tensors
        d_x, g_w, g_b = _dense_logexp_backward(x16, p, w, d_pred, d_logp, 0, 0, ctx.act)
        return d_x.view(*ctx.lead, x16.shape[1]), g_w, g_b, None


def _dense_logexp_backward(x16, p, w, d_pred, d_logp, groups, rows, act=0):
    """backward of log_predicted = act(x W^T + b), predicted = exp(log_predicted); groups > 0: (W, b) gradients per group of `rows` rows."""
    lib = _lib.load()
    M, K = x16.shape
    N = p.shape[1]
    NP = max(128, (N + 63) // 64 * 64)
    dev = p.device
    dp32 = torch.empty(M, N, device=dev, dtype=torch.float32)
    dp16 = torch.empty(M, NP, device=dev, dtype=torch.bfloat16)
    gp = None if d_pred is None else d_pred.reshape(M, N).contiguous().float()
    gl = None if d_logp is None else d_logp.reshape(M, N).contiguous().float()
    _lib.check(lib.se_spec_epilogue_bwd_f32(_lib.ptr(p), _lib.ptr(gp), _lib.ptr(gl), M, N, NP, 1 if act == _lib.SE_ACT['Identity'] else 2, act, 0.0,
                                            _lib.ptr(dp32), _lib.ptr(dp16), _lib.stream()), 'se_spec_epilogue_bwd_f32')
    g_w = _wgrad(dp16, x16, NP, K, groups, rows)[..., :N, :].contiguous()
    g_b = _colsum(dp32, N, groups, rows)
    wt = st.transpose_f32_bf16(w.detach().float().contiguous(), NP)             # (K, NP) = W^T zero padded
    d_x = st._gemm(dp16, wt, None, M, K, NP)
    return d_x, g_w, g_b


class LSTM(nn.Module):
    """model.py:37-59"""

    def __init__(self, input_size=201, output_size=201, hidden_size=201, num_layers=3, bidirectional=False, activation='Identity', **kwargs):
        super().__init__()
        if hidden_size > _H:
            raise NotImplementedError(f'the MI355X LSTM kernels are built for hidden_size <= {_H} (config/pseudo_noise.yaml:50-58), got {hidden_size}')
        self._act = _act_id(activation)
        if self._act not in (_lib.SE_ACT['Identity'], _lib.SE_ACT['ReLU'], _lib.SE_ACT['Sigmoid']):
            raise NotImplementedError(f"LSTM head: activation '{activation}' is not built (Identity / ReLU / Sigmoid)")
        self.lstm = nn.LSTM(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers, batch_first=True, bidirectional=bidirectional)
        self.scaling_layer = nn.Sequential(nn.Linear(max(1, int(bidirectional) * 2) * hidden_size, output_size), eval(f'nn.{activation}()'))
        self.bidirectional = bidirectional
        self.num_layers = num_layers
        self.init_weights()

    def init_weights(self):
        for name, param in self.named_parameters():
            if 'weight_ih' in name or 'scaling_layer.0.weight' in name:
                nn.init.xavier_uniform_(param.data)
            elif 'weight_hh' in name:
                nn.init.orthogonal_(param.data)
            elif 'bias' in name:
                nn.init.constant_(param.data, 0)

    def _flat_weights(self):
        ws = []
        for l in range(self.num_layers):
            for suffix in ([''] + (['_reverse'] if self.bidirectional else [])):
                ws += [getattr(self.lstm, f'{n}_l{l}{suffix}') for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]
        return ws

    def forward(self, features, **kwargs):
        h = _LSTMFn.apply(features, self.num_layers, 2 if self.bidirectional else 1, *self._flat_weights())
        ndir = 2 if self.bidirectional else 1
        w = _pad_dirs(self.scaling_layer[0].weight, self.lstm.hidden_size, ndir)          # differentiable: the gradient comes back un-padded
        predicted, log_predicted = _DenseLogExpFn.apply(h, w, self.scaling_layer[0].bias, self._act)
        return predicted, {'log_predicted': log_predicted}

    def per_utterance_gradients(self, features, d_log_predicted_fn):
        """One forward + ONE backward sweep -> {parameter name: (B, *shape) fp32}: the gradient of each utterance's own loss (the
        reference replays B sequential `loss_b.backward(retain_graph=True)` passes, sampler.py:84-109).
        `d_log_predicted_fn(log_predicted (B, T, N)) -> (B, T, N)` returns, row block b, d loss_b / d log_predicted[b]."""
        import types
        ndir = 2 if self.bidirectional else 1
        weights = self._flat_weights()
        B, T, _ = features.shape
        with torch.no_grad():
            c1, c2 = types.SimpleNamespace(), types.SimpleNamespace(save_for_backward=lambda *t: setattr(c2, 'saved_tensors', t))
            h = _LSTMFn.forward(c1, features, self.num_layers, ndir, *weights)
            w, b = self.scaling_layer[0].weight, self.scaling_layer[0].bias
            hs = self.lstm.hidden_size
            wp = _pad_dirs(w, hs, ndir)
            _, logp = _DenseLogExpFn.forward(c2, h, wp, b, self._act)
            x16, p, _w = c2.saved_tensors
            d_h, g_w, g_b = _dense_logexp_backward(x16, p, wp, None, d_log_predicted_fn(logp), B, T, self._act)
            g_w = _unpad_dirs(g_w, hs, ndir).contiguous()
            grads = _lstm_backward(c1.saved, c1.meta, weights, d_h, per_utterance=True)
        by_param = {id(pw): g for pw, g in zip(weights, grads)}
        by_param[id(w)], by_param[id(b)] = g_w, g_b
        return {n: by_param[id(pm)] for n, pm in self.named_parameters()}


class Residual(nn.Module):
    """model.py:62-91: LSTM -> (CMVN over time) -> Linear + activation = mask `offset`; predicted = linears * offset.
    The mask stage is the LinearResidual kernel (exact fp32) with its gradient wrt the LSTM output switched on."""

    def __init__(self, input_size=201, output_size=201, hidden_size=201, num_layers=3, bidirectional=False, activation='Sigmoid', cmvn=False,
                 eps=1e-6, **kwargs):
        super().__init__()
        if hidden_size > _H:
            raise NotImplementedError(f'the MI355X LSTM kernels are built for hidden_size <= {_H} (config/pseudo_noise.yaml:54-59), got {hidden_size}')
        self.lstm = nn.LSTM(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers, batch_first=True, bidirectional=bidirectional)
        self.scaling_layer = nn.Sequential(nn.Linear(max(1, int(bidirectional) * 2) * hidden_size, output_size), eval(f'nn.{activation}()'))
        self._act = _act_id(activation)
        self.bidirectional, self.num_layers, self.cmvn, self.eps = bidirectional, num_layers, cmvn, eps
        LSTM.init_weights(self)

    _flat_weights = LSTM._flat_weights

    def forward(self, features, linears, **kwargs):
        h = _LSTMFn.apply(features, self.num_layers, 2 if self.bidirectional else 1, *self._flat_weights())
        # padded columns of h are exactly zero and stay zero under the CMVN ((0 - 0) / (0 + eps)); the weight gets matching zero columns
        w = _pad_dirs(self.scaling_layer[0].weight, self.lstm.hidden_size, 2 if self.bidirectional else 1)
        predicted, offset = _HeadLinearFn.apply(h, linears, w, self.scaling_layer[0].bias, self._act, self.cmvn, self.eps)
        return predicted, {'offset': offset}
