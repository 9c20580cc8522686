"""Training path of the spectrogram-prediction head (rows B4 + C3 with autograd, row E2): forward keeps what the
backward needs, backward runs on the HIP building blocks of csrc/bwd.hip through the C ABI.

    x (M, 768) fp32 --cast--> x_bf --GEMM(Wd)+bd--> pre (fp32) --gelu+LayerNorm--> xn_bf --GEMM(Wo)+bo--> p (M, N) fp32
    predicted, log_predicted = SpecHead epilogue(p)                                   (model.py:121-125)

backward (mixed precision as the forward: bf16 GEMM operands, fp32 sums):
    dp   = epilogue'(p; d_pred, d_logp)              fp32 (M, N) + bf16 (M, 256) zero padded
    dWo  = dp^T xn   (se_wgrad_tn_bf16: row-major operands, split over M)   dbo = colsum(dp)
    dxn  = dp Wo     (se_gemm_bf16 on the transposed weight)
    dpre = LayerNorm'(gelu(pre)) . gelu'(pre)        (+ dgamma, dbeta)
    dWd  = dpre^T x  (wgrad)                                            dbd = colsum(dpre)
    dx   = dpre Wd   (only when the input requires a gradient: Mockingjay, where the encoder below is trained too)
"""
import torch

from . import _lib

_SPLITS = 8


def _mp(M, splits=_SPLITS):
    mc = (M + splits - 1) // splits
    mc = max(128, (mc + 63) // 64 * 64)
    return mc * splits


def _gemm(A16, W16, bias, M, N, K, act=0, out_f32=True, residual=None):
    lib = _lib.load()
    o32 = torch.empty(M, N, device=A16.device, dtype=torch.float32) if out_f32 else None
    o16 = None if out_f32 else torch.empty(M, N, device=A16.device, dtype=torch.bfloat16)
    _lib.check(lib.se_gemm_bf16(_lib.ptr(A16), A16.shape[1], _lib.ptr(W16), W16.shape[1], _lib.ptr(bias), _lib.ptr(residual), M, N, K, act,
                                _lib.ptr(o16), _lib.ptr(o32), N, _lib.stream()), 'se_gemm_bf16')
    return o32 if out_f32 else o16


def cast_bf16(x):
    lib = _lib.load()
    x = x.contiguous().float()
    out = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    n = x.numel()
    assert n % 4 == 0
    _lib.check(lib.se_cast_f32_bf16(_lib.ptr(x), n, _lib.ptr(out), _lib.stream()), 'se_cast_f32_bf16')
    return out


def transpose_bf16(x16, ld_out, cols=None):
    """(rows, cols) bf16 (the first `cols` columns of a contiguous (rows, ld) tensor) -> (cols, ld_out) bf16, zero padded."""
    lib = _lib.load()
    rows = x16.shape[0]
    cols = x16.shape[1] if cols is None else cols
    out = torch.empty(cols, ld_out, device=x16.device, dtype=torch.bfloat16)
    _lib.check(lib.se_transpose_bf16(_lib.ptr(x16), rows, cols, x16.stride(0), _lib.ptr(out), ld_out, _lib.stream()), 'se_transpose_bf16')
    return out


def transpose_f32_bf16(x, ld_out):
    lib = _lib.load()
    rows, cols = x.shape
    out = torch.empty(cols, ld_out, device=x.device, dtype=torch.bfloat16)
    _lib.check(lib.se_transpose_f32_bf16(_lib.ptr(x), rows, cols, x.stride(0), _lib.ptr(out), ld_out, _lib.stream()), 'se_transpose_f32_bf16')
    return out


def wgrad(dYt16, Xt16, N, K, splits=_SPLITS):
    """dW (N, K) fp32 = dY^T X from transposed bf16 operands dYt (N, Mp), Xt (K, Mp)."""
    lib = _lib.load()
    Mp = dYt16.shape[1]
    dW = torch.empty(N, K, device=dYt16.device, dtype=torch.float32)
    ws = torch.empty(splits * N * K, device=dYt16.device, dtype=torch.float32)
    _lib.check(lib.se_wgrad_bf16(_lib.ptr(dYt16), _lib.ptr(Xt16), Mp, N, K, splits, _lib.ptr(dW), 0, _lib.ptr(ws), ws.numel() * 4,
                                 _lib.stream()), 'se_wgrad_bf16')
    return dW


def wgrad_tn(dY16, X16, N, K, splits=8):
    """dW (N, K) fp32 = dY^T X straight from the row-major bf16 operands dY (M, ldy >= N), X (M, ldx >= K)."""
    lib = _lib.load()
    M = dY16.shape[0]
    while splits > 1 and splits * 64 > M:
        splits //= 2
    dW = torch.empty(N, K, device=dY16.device, dtype=torch.float32)
    ws = torch.empty(splits * N * K, device=dY16.device, dtype=torch.float32)
    _lib.check(lib.se_wgrad_tn_bf16(_lib.ptr(dY16), dY16.shape[1], _lib.ptr(X16), X16.shape[1], M, N, K, splits, _lib.ptr(dW), 0,
                                    _lib.ptr(ws), ws.numel() * 4, _lib.stream()), 'se_wgrad_tn_bf16')
    return dW


def colsum(x, cols=None):
    lib = _lib.load()
    rows, ld = x.shape
    cols = ld if cols is None else cols
    out = torch.empty(cols, device=x.device, dtype=torch.float32)
    _lib.check(lib.se_colsum_f32(_lib.ptr(x), rows, cols, ld, _lib.ptr(out), 0, _lib.stream()), 'se_colsum_f32')
    return out


class SpecHeadTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hidden, dense_w, dense_b, ln_w, ln_b, out_w, out_b, ln_eps, log_target, act, eps):
        lib = _lib.load()
        if not hidden.is_cuda:
            raise _lib.SEError('the spec head trains on MI355X only (no CPU fallback)')
        lead = hidden.shape[:-1]
        H = hidden.shape[-1]
        N = out_w.shape[0]
        if H not in (768, 256):
            raise NotImplementedError('the spec-head training kernels are built for hidden_size 768; for inference at other sizes call the head under '
                                      'torch.no_grad() (as Runner.evaluate does, runner.py:547)')
        x = hidden.reshape(-1, H)
        M = x.shape[0]
        x16 = cast_bf16(x)
        wd16, wo16 = cast_bf16(dense_w), cast_bf16(out_w)
        pre = _gemm(x16, wd16, dense_b.contiguous().float(), M, H, H)
        xn16 = torch.empty(M, H, device=x.device, dtype=torch.bfloat16)
        _lib.check(lib.se_gelu_layernorm_f32(_lib.ptr(pre), _lib.ptr(ln_w.contiguous().float()), _lib.ptr(ln_b.contiguous().float()), M, H,
                                             float(ln_eps), None, _lib.ptr(xn16), _lib.stream()), 'se_gelu_layernorm_f32')
        p = _gemm(xn16, wo16, out_b.contiguous().float(), M, N, H)
        pred = torch.empty(M, N, device=x.device, dtype=torch.float32)
        logp = torch.empty(M, N, device=x.device, dtype=torch.float32)
        _lib.check(lib.se_spec_epilogue_f32(_lib.ptr(p), M * N, int(bool(log_target)), act, float(eps), _lib.ptr(pred), _lib.ptr(logp),
                                            _lib.stream()), 'se_spec_epilogue_f32')
        ctx.save_for_backward(x16, pre, xn16, p, ln_w, out_w, dense_w)
        ctx.meta = (float(ln_eps), int(bool(log_target)), act, float(eps), lead)
        return pred.reshape(*lead, N), logp.reshape(*lead, N)

    @staticmethod
    def backward(ctx, d_pred, d_logp):
        lib = _lib.load()
        x16, pre, xn16, p, ln_w, out_w, dense_w = ctx.saved_tensors
        ln_eps, log_target, act, eps, lead = ctx.meta
        M, H = pre.shape
        N = p.shape[1]
        NP = (N + 63) // 64 * 64
        NP = max(NP, 128)
        dev = pre.device
        dpred = None if d_pred is None else d_pred.reshape(M, N).contiguous().float()
        dlogp = None if d_logp is None else d_logp.reshape(M, N).contiguous().float()
        dp32 = torch.empty(M, N, device=dev, dtype=torch.float32)
        dp16 = torch.empty(M, NP, device=dev, dtype=torch.bfloat16)
        _lib.check(lib.se_spec_epilogue_bwd_f32(_lib.ptr(p), _lib.ptr(dpred), _lib.ptr(dlogp), M, N, NP, log_target, act, eps, _lib.ptr(dp32),
                                                _lib.ptr(dp16), _lib.stream()), 'se_spec_epilogue_bwd_f32')
        Mp = _mp(M)
        # output linear: dWo = dp^T xn, dbo = colsum(dp), dxn = dp Wo
        dpt = transpose_bf16(dp16, Mp, cols=N)                # (N, Mp)
        xnt = transpose_bf16(xn16, Mp)                        # (H, Mp)
        d_out_w = wgrad(dpt, xnt, N, H)
        d_out_b = colsum(dp32)
        wot = transpose_f32_bf16(out_w.contiguous().float(), NP)        # (H, NP) = Wo^T zero padded along N
        dxn = _gemm(dp16, wot, None, M, H, NP)
        # gelu + LayerNorm backward
        dpre32 = torch.empty(M, H, device=dev, dtype=torch.float32)
        dpre16 = torch.empty(M, H, device=dev, dtype=torch.bfloat16)
        d_ln_w = torch.empty(H, device=dev, dtype=torch.float32)
        d_ln_b = torch.empty(H, device=dev, dtype=torch.float32)
        _lib.check(lib.se_layernorm_bwd_f32(_lib.ptr(pre), _lib.ptr(dxn), _lib.ptr(ln_w.contiguous().float()), M, H, ln_eps, 1, _lib.ptr(dpre32),
                                            _lib.ptr(dpre16), _lib.ptr(d_ln_w), _lib.ptr(d_ln_b), 0, _lib.stream()), 'se_layernorm_bwd_f32')
        # dense linear: dWd = dpre^T x, dbd = colsum(dpre)
        d_dense_w = wgrad_tn(dpre16, x16, H, H)
        d_dense_b = colsum(dpre32)
        d_hidden = None
        if ctx.needs_input_grad[0]:
            wdt = transpose_f32_bf16(dense_w.contiguous().float(), H)       # (H_in, H_out) = Wd^T
            d_hidden = _gemm(dpre16, wdt, None, M, H, H).reshape(*lead, H)
        return d_hidden, d_dense_w, d_dense_b, d_ln_w, d_ln_b, d_out_w, d_out_b, None, None, None, None
