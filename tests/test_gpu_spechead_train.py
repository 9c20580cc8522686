"""GPU parity of the backward building blocks and of SpecHead training (row E2 for C3): HIP path through the C ABI vs
PyTorch autograd on the CPU oracle arithmetic (fp64)."""
import pytest
import torch

from oracle import encoder as oenc
from oracle import heads as oheads
from oracle import objective as oobj

pytestmark = pytest.mark.gpu


def _L():
    from speech_enhancement_by_s3prl_amd import _lib
    return _lib


def test_transpose_and_wgrad(gpu):
    from speech_enhancement_by_s3prl_amd import spechead_train as st
    torch.manual_seed(0)
    M, N, K = 1001, 201, 768
    dY = torch.randn(M, N, device=gpu).bfloat16()
    X = torch.randn(M, K, device=gpu).bfloat16()
    Mp = st._mp(M)
    dYt, Xt = st.transpose_bf16(dY, Mp), st.transpose_bf16(X, Mp)
    assert dYt.shape == (N, Mp) and torch.equal(dYt[:, :M], dY.T) and torch.count_nonzero(dYt[:, M:]) == 0
    dW = st.wgrad(dYt, Xt, N, K)
    ref = dY.double().T @ X.double()
    assert (dW.double() - ref).abs().max().item() < 1e-4 * ref.abs().max().item() + 1e-3
    cs = st.colsum(dY.float())
    assert torch.allclose(cs.double(), dY.double().sum(0), atol=1e-3)


@pytest.mark.parametrize('gelu_in', [0, 1])
def test_layernorm_backward(gpu, gelu_in):
    L = _L()
    lib = L.load()
    torch.manual_seed(1)
    M, H = 515, 768
    x = torch.randn(M, H, device=gpu) * 2
    dy = torch.randn(M, H, device=gpu)
    w = 1 + 0.1 * torch.randn(H, device=gpu)
    dx = torch.empty(M, H, device=gpu)
    dg, db = torch.empty(H, device=gpu), torch.empty(H, device=gpu)
    L.check(lib.se_layernorm_bwd_f32(L.ptr(x), L.ptr(dy), L.ptr(w), M, H, 1e-12, gelu_in, L.ptr(dx), None, L.ptr(dg), L.ptr(db), 0, L.stream()), 'ln_bwd')
    xd = x.double().cpu().requires_grad_(True)
    wd = w.double().cpu().requires_grad_(True)
    bd = torch.zeros(H, dtype=torch.float64, requires_grad=True)
    inp = oenc.gelu(xd) if gelu_in else xd
    y = oenc.layer_norm(inp, wd, bd, 1e-12)
    (y * dy.double().cpu()).sum().backward()
    assert (dx.double().cpu() - xd.grad).abs().max().item() < 2e-5 * xd.grad.abs().max().item() + 1e-5
    assert (dg.double().cpu() - wd.grad).abs().max().item() < 1e-4 * wd.grad.abs().max().item()
    assert (db.double().cpu() - bd.grad).abs().max().item() < 1e-4 * bd.grad.abs().max().item()


def test_spechead_training_gradients_vs_autograd(gpu):
    """SpecHead as the trainable downstream model on frozen features: loss = L1(log_predicted, target)."""
    from speech_enhancement_by_s3prl_amd import pipeline
    from speech_enhancement_by_s3prl_amd.heads import SpecHead
    from speech_enhancement_by_s3prl_amd.objective import L1
    cfg = pipeline.make_config(layers=1)
    ckpt = pipeline.synthetic_checkpoint(cfg, seed=5)
    head = SpecHead(201, ckpt).to(gpu)
    torch.manual_seed(2)
    B, T = 2, 300
    feats = torch.randn(B, T, 768)
    tar = torch.rand(B, T, 201) + 0.05
    lens = torch.tensor([300, 177])
    pred, res = head(features=feats.to(gpu))
    assert pred.requires_grad and res['log_predicted'].requires_grad
    loss, _ = L1()(log_predicted=res['log_predicted'], linear_tar=tar.to(gpu), stft_lengths=lens.to(gpu))
    loss.backward()
    # oracle: fp64 autograd on the same arithmetic
    sd = {k: v.double().clone().requires_grad_(True) for k, v in ckpt['SpecHead'].items()}
    ocfg = oenc.Config(cfg)
    rpred, rres = oheads.spec_head(feats.double(), sd, ocfg, log=True)
    masks = (torch.arange(T)[None] < lens[:, None]).long()
    rloss = oobj.l1(rres['log_predicted'], tar.double(), masks)
    rloss.backward()
    assert abs(loss.item() - rloss.item()) < 2e-2 * abs(rloss.item())
    got = {'dense.weight': head.spechead.dense.weight.grad, 'dense.bias': head.spechead.dense.bias.grad,
           'LayerNorm.weight': head.spechead.LayerNorm.weight.grad, 'LayerNorm.bias': head.spechead.LayerNorm.bias.grad,
           'output.weight': head.spechead.output.weight.grad, 'output.bias': head.spechead.output.bias.grad}
    for k, g in got.items():
        r = sd[k].grad
        rel = ((g.double().cpu() - r).norm() / r.norm()).item()
        # bf16 GEMM operands vs an fp64 reference, and L1's sign() gradient flips wherever the two forwards straddle the target
        assert rel < 1e-1, (k, rel)


def test_spechead_gradients_linear_functional(gpu):
    """Same chain under a loss that is linear in the outputs (no sign() flips), against fp64 autograd that rounds the GEMM
    operands to bf16 at the same places as the HIP forward: what is left is the backward's own bf16 operand rounding."""
    from speech_enhancement_by_s3prl_amd import pipeline
    from speech_enhancement_by_s3prl_amd.heads import SpecHead
    cfg = pipeline.make_config(layers=1)
    ckpt = pipeline.synthetic_checkpoint(cfg, seed=7)
    head = SpecHead(201, ckpt).to(gpu)
    torch.manual_seed(4)
    B, T = 2, 333
    feats = torch.randn(B, T, 768)
    G1, G2 = torch.randn(B, T, 201), torch.randn(B, T, 201)
    pred, res = head(features=feats.to(gpu))
    ((pred * G1.to(gpu)).sum() + (res['log_predicted'] * G2.to(gpu)).sum()).backward()

    class RoundBF16(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.float().bfloat16().double()

        @staticmethod
        def backward(ctx, g):
            return g
    r = RoundBF16.apply
    sd = {k: v.double().clone().requires_grad_(True) for k, v in ckpt['SpecHead'].items()}
    pre = r(feats.double()) @ r(sd['dense.weight']).T + sd['dense.bias']
    xn = oenc.layer_norm(oenc.gelu(pre), sd['LayerNorm.weight'], sd['LayerNorm.bias'], oenc.Config(cfg).layer_norm_eps)
    p = r(xn) @ r(sd['output.weight']).T + sd['output.bias']
    rpred, rlogp = p.exp(), p                    # log target, identity activation (model.py:121-125)
    assert (pred.double().cpu() - rpred).abs().max().item() < 2e-3 * rpred.abs().max().item()
    ((rpred * G1.double()).sum() + (rlogp * G2.double()).sum()).backward()
    got = {'dense.weight': head.spechead.dense.weight.grad, 'dense.bias': head.spechead.dense.bias.grad,
           'LayerNorm.weight': head.spechead.LayerNorm.weight.grad, 'LayerNorm.bias': head.spechead.LayerNorm.bias.grad,
           'output.weight': head.spechead.output.weight.grad, 'output.bias': head.spechead.output.bias.grad}
    for k, g in got.items():
        ref = sd[k].grad
        rel = ((g.double().cpu() - ref).norm() / ref.norm()).item()
        assert rel < 1e-2, (k, rel)


def test_spechead_train_step_decreases_loss(gpu):
    """a few BertAdam steps on the HIP gradients reduce the L1 loss (runner.py:453-471 with the SpecHead downstream)."""
    from speech_enhancement_by_s3prl_amd import pipeline
    from speech_enhancement_by_s3prl_amd.heads import SpecHead
    from speech_enhancement_by_s3prl_amd.objective import L1
    from speech_enhancement_by_s3prl_amd.solver import get_optimizer
    cfg = pipeline.make_config(layers=1)
    head = SpecHead(201, pipeline.synthetic_checkpoint(cfg, seed=6)).to(gpu)
    opt = get_optimizer(list(head.named_parameters()), lr=1e-3, warmup_proportion=0.07, training_steps=100)
    torch.manual_seed(3)
    feats = torch.randn(2, 200, 768, device=gpu)
    tar = torch.rand(2, 200, 201, device=gpu) + 0.05
    lens = torch.tensor([200, 150], device=gpu)
    crit = L1()
    losses = []
    for _ in range(8):
        pred, res = head(features=feats)
        loss, _ = crit(log_predicted=res['log_predicted'], linear_tar=tar, stft_lengths=lens)
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(list(head.parameters()), 1.0)
        assert torch.isfinite(gn)
        opt.step()
        opt.zero_grad()
        losses.append(loss.item())
    assert losses[-1] < losses[0]


@pytest.mark.parametrize('M,N,K,splits', [(5003, 768, 3072, 8), (4100, 2304, 768, 8), (1001, 768, 128, 4), (333, 256, 512, 1), (70, 8, 8, 3)])
def test_wgrad_tn_vs_fp64(gpu, M, N, K, splits):
    """weight gradient straight from the row-major operands (tr-read fragments, zero source for the ragged m tail)."""
    L = _L()
    lib = L.load()
    torch.manual_seed(M)
    ldy, ldx = N + 8, K                      # a padded leading dimension on one side
    dY = torch.randn(M, ldy, device=gpu).bfloat16()
    X = torch.randn(M, ldx, device=gpu).bfloat16()
    dW = torch.full((N, K), float('nan'), device=gpu)
    ws = torch.empty(splits * N * K, device=gpu)
    L.check(lib.se_wgrad_tn_bf16(L.ptr(dY), ldy, L.ptr(X), ldx, M, N, K, splits, L.ptr(dW), 0, L.ptr(ws), ws.numel() * 4, L.stream()), 'wgrad_tn')
    ref = dY[:, :N].double().T @ X.double()
    assert (dW.double() - ref).abs().max().item() < 2e-5 * ref.abs().max().item() + 1e-4
    # accumulate form
    L.check(lib.se_wgrad_tn_bf16(L.ptr(dY), ldy, L.ptr(X), ldx, M, N, K, splits, L.ptr(dW), 1, L.ptr(ws), ws.numel() * 4, L.stream()), 'wgrad_tn')
    assert (dW.double() - 2 * ref).abs().max().item() < 4e-5 * ref.abs().max().item() + 2e-4
