"""GPU parity of the LinearResidual training step pieces (runner.py:453-471): head backward through the HIP
kernels vs the reference's own autograd (golden chain gradient) and the oracle at full size."""
import pytest
import torch

from oracle import heads as oheads
from oracle import objective as oobj

pytestmark = pytest.mark.gpu


def T(x, dev=None):
    t = torch.from_numpy(x)
    return t.to(dev) if dev is not None else t


def test_chain_gradient_golden(gpu, golden):
    """loss = L1(log(LinearResidual(feats, linears) + 1e-10), tar); d loss / d (W, b) from the reference's autograd."""
    from speech_enhancement_by_s3prl_amd.decode import get_length_masks
    from speech_enhancement_by_s3prl_amd.heads import LinearResidual
    from speech_enhancement_by_s3prl_amd.objective import L1
    m = LinearResidual(input_size=120, output_size=201, cmvn=True).to(gpu)
    m.load_state_dict({'linear.weight': T(golden['c1_weight']), 'linear.bias': T(golden['c1_bias'])})
    pred, _ = m(features=T(golden['c1_feats'], gpu), linears=T(golden['c1_linears'], gpu))
    logp = (pred + 1e-10).log()
    masks = get_length_masks(T(golden['d1_lengths'], gpu))
    loss, _ = L1()(log_predicted=logp, linear_tar=T(golden['e1_linear_tar'], gpu), stft_length_masks=masks)
    loss.backward()
    assert abs(loss.item() - float(golden['chain_loss'])) < 1e-5 * abs(float(golden['chain_loss']))
    gw, gb = T(golden['chain_gw']), T(golden['chain_gb'])
    assert (m.linear.weight.grad.cpu() - gw).abs().max().item() < 1e-4 * gw.abs().max().item()
    assert (m.linear.bias.grad.cpu() - gb).abs().max().item() < 1e-4 * gb.abs().max().item()


@pytest.mark.parametrize('B,F,D,N,act', [(2, 1001, 120, 201, 'Sigmoid'), (1, 300, 768, 201, 'ReLU'), (3, 50, 40, 33, 'Identity')])
def test_head_backward_vs_autograd(gpu, B, F, D, N, act):
    from speech_enhancement_by_s3prl_amd.heads import LinearResidual
    torch.manual_seed(F)
    feats, lin, g = torch.randn(B, F, D), torch.rand(B, F, N) + 0.1, torch.randn(B, F, N)
    m = LinearResidual(input_size=D, output_size=N, activation=act, cmvn=True)
    w = m.linear.weight.detach().clone().double().requires_grad_(True)
    b = m.linear.bias.detach().clone().double().requires_grad_(True)
    rp, _ = oheads.linear_residual(feats.double(), lin.double(), w, b, activation=act)
    (rp * g.double()).sum().backward()
    m = m.to(gpu)
    p, _ = m(features=feats.to(gpu), linears=lin.to(gpu))
    (p * g.to(gpu)).sum().backward()
    assert (m.linear.weight.grad.cpu().double() - w.grad).abs().max().item() < 1e-4 * w.grad.abs().max().item()
    assert (m.linear.bias.grad.cpu().double() - b.grad).abs().max().item() < 1e-4 * b.grad.abs().max().item()
