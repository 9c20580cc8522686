"""GPU parity of the (Bi)LSTM downstream heads (model.py:37-91): (a) against OUTPUTS OF THE REFERENCE'S OWN model.LSTM /
model.Residual (tests/golden/reference_golden_heads.npz: forward + every parameter gradient, hidden 201 / 256, both directions, CMVN on / off),
(b) at the sizes no fixture holds (T up to 1001) against the class bodies restated on torch.nn.LSTM in fp64 (tests/ref_heads.py, themselves pinned
to the same fixture by tests/test_oracle_golden.py).  bf16 recurrent / projection weights and bf16 h operands on the HIP side: bounds are relative
L2 per tensor, stated at the asserts."""
import copy

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    if b.norm().item() == 0.0:          # e.g. dW_hh at T = 1 (h_0 = 0): the HIP result must be exactly zero too
        return a.norm().item()
    return ((a - b).norm() / b.norm()).item()


from ref_heads import RefLSTM, RefResidual      # noqa: E402  (pinned to the reference's own classes by tests/test_oracle_golden.py)
import ref_heads as RH      # noqa: E402


@pytest.mark.parametrize('B,T,D,layers,bidir', [(3, 50, 120, 3, True), (2, 37, 120, 2, False), (2, 1001, 120, 1, True),
                                                  (2, 1, 120, 1, True), (1, 2, 120, 2, True), (2, 3, 120, 1, False), (1, 4, 120, 1, True)])   # T < 5: the DMA ring's special cases
def test_lstm_head_forward_backward_vs_torch(gpu, B, T, D, layers, bidir):
    from speech_enhancement_by_s3prl_amd.lstm import LSTM
    torch.manual_seed(B * 1000 + T)
    head = LSTM(input_size=D, output_size=201, hidden_size=256, num_layers=layers, bidirectional=bidir)
    with torch.no_grad():                      # non-zero biases, recurrent weights at a realistic scale
        for n, p in head.named_parameters():
            if 'bias' in n:
                p.normal_(0, 0.05)
            if 'scaling_layer.0.weight' in n:
                p.mul_(0.3)
    ref = RefLSTM(D, 201, 256, layers, bidir).double()
    ref.load_state_dict({k: v.double() for k, v in head.state_dict().items()})       # same parameter names: checkpoints interchange
    head = head.to(gpu)
    feats = torch.randn(B, T, D)
    G1, G2 = torch.randn(B, T, 201) * 0.1, torch.randn(B, T, 201)
    pred, res = head(features=feats.to(gpu))
    logp = res['log_predicted']
    ((pred * G1.to(gpu)).sum() + (logp * G2.to(gpu)).sum()).backward()
    rpred, rlogp = ref(feats.double())
    ((rpred * G1.double()).sum() + (rlogp * G2.double()).sum()).backward()
    assert rel_l2(logp, rlogp) < 1.5e-2
    assert rel_l2(pred, rpred) < 2.5e-2
    refp = dict(ref.named_parameters())
    for n, p in head.named_parameters():
        assert p.grad is not None, n
        r = rel_l2(p.grad, refp[n].grad)
        assert r < 4e-2, (n, r)


@pytest.mark.parametrize('hidden,bidir,act', [(201, False, 'Identity'), (201, True, 'ReLU'), (64, True, 'Sigmoid'), (256, False, 'ReLU')])
def test_lstm_head_other_widths_and_activations(gpu, hidden, bidir, act):
    """the reference's class defaults (model.py:38: hidden_size 201, any nn activation on the scaling layer): hidden sizes below the
    kernels' 256 run on zero-padded weights (exact: a padded unit stays at h = c = 0), log_predicted = act(linear), predicted = exp(.)"""
    from speech_enhancement_by_s3prl_amd.lstm import LSTM
    torch.manual_seed(hidden)
    B, T, D, layers = 2, 41, 201, 2
    head = LSTM(input_size=D, output_size=201, hidden_size=hidden, num_layers=layers, bidirectional=bidir, activation=act)
    with torch.no_grad():
        for n, p in head.named_parameters():
            if 'bias' in n:
                p.normal_(0, 0.05)
            if 'scaling_layer.0.weight' in n:
                p.mul_(0.3)
    ref = RefLSTM(D, 201, hidden, layers, bidir, act).double()
    ref.load_state_dict({k: v.double() for k, v in head.state_dict().items()})
    head = head.to(gpu)
    feats = torch.randn(B, T, D)
    G1, G2 = torch.randn(B, T, 201) * 0.1, torch.randn(B, T, 201)
    pred, res = head(features=feats.to(gpu))
    logp = res['log_predicted']
    assert logp.shape == (B, T, 201)
    ((pred * G1.to(gpu)).sum() + (logp * G2.to(gpu)).sum()).backward()
    rpred, rlogp = ref(feats.double())
    ((rpred * G1.double()).sum() + (rlogp * G2.double()).sum()).backward()
    assert rel_l2(logp, rlogp) < 1.5e-2
    assert rel_l2(pred, rpred) < 2.5e-2
    refp = dict(ref.named_parameters())
    for n, p in head.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape, n
        r = rel_l2(p.grad, refp[n].grad)
        assert r < 4e-2, (n, r)


def test_lstm_head_trains(gpu):
    from speech_enhancement_by_s3prl_amd.lstm import LSTM
    from speech_enhancement_by_s3prl_amd.objective import L1
    from speech_enhancement_by_s3prl_amd.solver import get_optimizer
    torch.manual_seed(0)
    head = LSTM(input_size=120, output_size=201, hidden_size=256, num_layers=3, bidirectional=True).to(gpu)
    opt = get_optimizer(list(head.named_parameters()), lr=1e-3, warmup_proportion=0.07, training_steps=100)
    feats = torch.randn(2, 120, 120, device=gpu)
    tar = torch.rand(2, 120, 201, device=gpu) + 0.05
    lens = torch.tensor([120, 77], device=gpu)
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
    assert losses[-1] < losses[0], losses


@pytest.mark.parametrize('bidir,cmvn,hidden', [(False, True, 256), (True, False, 256), (True, True, 201)])
def test_residual_head_vs_torch(gpu, bidir, cmvn, hidden):
    """LSTM -> CMVN over time -> Linear + Sigmoid mask -> mask (.) noisy power, trained through `predicted` AND the mask
    (SISDR + WSD style gradients), against the reference's class body on torch.nn.LSTM in fp64."""
    from speech_enhancement_by_s3prl_amd.lstm import Residual
    torch.manual_seed(5)
    B, T, D = 3, 60, 120
    head = Residual(input_size=D, output_size=201, hidden_size=hidden, num_layers=2, bidirectional=bidir, activation='Sigmoid', cmvn=cmvn)
    with torch.no_grad():
        for n, p in head.named_parameters():
            if 'bias' in n:
                p.normal_(0, 0.05)
    ref = RefResidual(D, 201, hidden, 2, bidir, cmvn).double()
    ref.load_state_dict({k: v.double() for k, v in head.state_dict().items()})
    head = head.to(gpu)
    feats, lin = torch.randn(B, T, D), torch.rand(B, T, 201) + 0.1
    G1, G2 = torch.randn(B, T, 201), torch.randn(B, T, 201)
    pred, res = head(features=feats.to(gpu), linears=lin.to(gpu))
    ((pred * G1.to(gpu)).sum() + (res['offset'] * G2.to(gpu)).sum()).backward()
    rpred, roff = ref(feats.double(), lin.double())
    ((rpred * G1.double()).sum() + (roff * G2.double()).sum()).backward()
    assert rel_l2(res['offset'], roff) < 1e-2
    assert rel_l2(pred, rpred) < 1e-2
    refp = dict(ref.named_parameters())
    for n, p in head.named_parameters():
        assert p.grad is not None, n
        r = rel_l2(p.grad, refp[n].grad)
        assert r < 4e-2, (n, r)


def _T(x, dev=None):
    import numpy as np
    t = torch.from_numpy(np.asarray(x))
    return t.to(dev) if dev is not None else t


@pytest.mark.parametrize('ci', range(len(RH.HEAD_CASES)))
def test_lstm_and_residual_heads_vs_reference_golden(gpu, golden_heads, ci):
    """the HIP heads against what the REFERENCE's classes returned for the same seeded parameters and inputs (make_golden.py: heads_fixture)"""
    from speech_enhancement_by_s3prl_amd.lstm import LSTM, Residual
    G = golden_heads
    tag, hidden, bidir, cmvn, layers = RH.HEAD_CASES[ci]
    feats, G1, G2 = _T(G[f'lstm_{tag}_feats'], gpu), _T(G[f'lstm_{tag}_G1'], gpu), _T(G[f'lstm_{tag}_G2'], gpu)
    head = RH.seeded.fill_params(LSTM(input_size=RH.HEAD_D, output_size=RH.HEAD_K, hidden_size=hidden, num_layers=layers, bidirectional=bidir), 100 + ci).to(gpu)
    pred, res = head(features=feats)
    ((pred * G1).sum() + (res['log_predicted'] * G2).sum()).backward()
    assert rel_l2(res['log_predicted'], _T(G[f'lstm_{tag}_log_predicted'])) < 1.5e-2
    assert rel_l2(pred, _T(G[f'lstm_{tag}_predicted'])) < 2.5e-2
    RH.check_grad_samples(G, f'lstm_{tag}', [p.grad for p in head.parameters()], 1000 * ci, 4e-2, 'LSTM')
    rhead = RH.seeded.fill_params(Residual(input_size=RH.HEAD_D, output_size=RH.HEAD_K, hidden_size=hidden, num_layers=layers, bidirectional=bidir,
                                           activation='Sigmoid', cmvn=cmvn), 200 + ci).to(gpu)
    pred, res = rhead(features=feats, linears=_T(G[f'res_{tag}_linears'], gpu))
    ((pred * G1).sum() + (res['offset'] * G2).sum()).backward()
    assert rel_l2(res['offset'], _T(G[f'res_{tag}_offset'])) < 1e-2
    assert rel_l2(pred, _T(G[f'res_{tag}_predicted'])) < 1e-2
    RH.check_grad_samples(G, f'res_{tag}', [p.grad for p in rhead.parameters()], 1000 * ci + 500, 4e-2, 'Residual')
