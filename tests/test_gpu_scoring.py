"""GPU parity of per-utterance gradient scoring (sampler.py:59-116): one batched backward sweep with
per-utterance weight-gradient slabs vs the reference's procedure -- B sequential per-utterance backward passes -- run with
fp64 autograd on the CPU oracle.  The forward runs on bf16 GEMM operands and the L1 gradient is a sign, so the bound is on
the direction (cosine) and norm of each utterance's gradient and on the resulting matching scores, not element-wise."""
import pytest
import torch

from oracle import encoder as oenc
from oracle import heads as oheads
from oracle import objective as oobj

pytestmark = pytest.mark.gpu


def test_wgrad_slabs_per_group(gpu):
    from speech_enhancement_by_s3prl_amd import _lib as L
    lib = L.load()
    torch.manual_seed(0)
    G, R, N, K = 5, 101, 256, 768
    dY = torch.randn(G * R, N, device=gpu).bfloat16()
    X = torch.randn(G * R, K, device=gpu).bfloat16()
    slabs = torch.full((G, N, K), float('nan'), device=gpu)
    L.check(lib.se_wgrad_tn_slabs_bf16(L.ptr(dY), N, L.ptr(X), K, G * R, N, K, R, L.ptr(slabs), L.stream()), 'slabs')
    for g in range(G):
        ref = dY[g * R:(g + 1) * R].double().T @ X[g * R:(g + 1) * R].double()
        assert (slabs[g].double() - ref).abs().max().item() < 2e-5 * ref.abs().max().item() + 1e-4


def test_per_sample_gradients_vs_sequential_autograd(gpu):
    from speech_enhancement_by_s3prl_amd import pipeline, scoring
    from speech_enhancement_by_s3prl_amd.heads import SpecHead
    cfg = pipeline.make_config(layers=1)
    ckpt = pipeline.synthetic_checkpoint(cfg, seed=12)
    head = SpecHead(201, ckpt).to(gpu)
    torch.manual_seed(7)
    B, T = 4, 80
    feats = torch.randn(B, T, 768)
    tar = torch.rand(B, T, 201) + 0.05
    lens = torch.tensor([80, 51, 17, 64])
    got = scoring.per_sample_gradients(head, feats.to(gpu), tar.to(gpu), lens.to(gpu)).double().cpu()
    names = [n for n, _ in head.named_parameters()]
    assert got.shape == (B, sum(p.numel() for p in head.parameters()))
    # the reference's procedure (sampler.py:84-109): one loss per utterance, one backward each, gradients flattened in
    # named_parameters order
    ocfg = oenc.Config(cfg)
    sd = {k: v.double().clone().requires_grad_(True) for k, v in ckpt['SpecHead'].items()}
    _, res = oheads.spec_head(feats.double(), sd, ocfg, log=True)
    masks = (torch.arange(T)[None] < lens[:, None]).long()
    ref = []
    for b in range(B):
        loss = oobj.l1(res['log_predicted'][b:b + 1], tar[b:b + 1].double(), masks[b:b + 1])
        grads = torch.autograd.grad(loss, [sd[n.replace('spechead.', '')] for n in names], retain_graph=True)
        ref.append(torch.cat([g.reshape(-1) for g in grads]))
    ref = torch.stack(ref)
    for b in range(B):
        cos = torch.nn.functional.cosine_similarity(got[b], ref[b], dim=0).item()
        ratio = (got[b].norm() / ref[b].norm()).item()
        assert cos > 0.99, (b, cos)
        assert abs(ratio - 1) < 0.05, (b, ratio)
    # matching scores (sampler.py:113-116): keys = the 4 utterances, queries = the first two
    m_got = scoring.matching(got[:2].float(), got.float())
    m_ref = scoring.matching(ref[:2].float(), ref.float())
    assert (m_got - m_ref).abs().max().item() < 0.02
    assert torch.equal(scoring.thresholding(m_got), scoring.thresholding(m_ref))


@pytest.mark.parametrize('active_layerid', [None, 1])
def test_lstm_per_sample_gradients_vs_sequential_autograd(gpu, active_layerid):
    """The head the reference's active runs score (run_active.sh:11 `--downstream LSTM`; pseudo_noise.yaml:50-53): ONE sweep with
    per-utterance slabs / grouped sums vs the reference's procedure (sampler.py:84-109: one `loss_b.backward(retain_graph=True)` per
    utterance, gradients concatenated in named_parameters order, `active_layerid` filtering nn.LSTM parameters by the reference's own
    pattern), run on torch.nn.LSTM in fp64."""
    import re
    from speech_enhancement_by_s3prl_amd import scoring
    from speech_enhancement_by_s3prl_amd.lstm import LSTM
    torch.manual_seed(5)
    B, T, D, N = 3, 70, 120, 201
    head = LSTM(input_size=D, output_size=N, hidden_size=256, num_layers=2, bidirectional=True).to(gpu)
    feats = torch.randn(B, T, D)
    tar = torch.rand(B, T, N) + 0.05
    lens = torch.tensor([70, 41, 9])
    got = scoring.per_sample_gradients_lstm(head, feats.to(gpu), tar.to(gpu), lens.to(gpu), active_layerid=active_layerid).double().cpu()
    # reference procedure in fp64
    ref_lstm = torch.nn.LSTM(input_size=D, hidden_size=256, num_layers=2, batch_first=True, bidirectional=True).double()
    ref_lin = torch.nn.Linear(512, N).double()
    sd = {k: v.detach().double().cpu() for k, v in head.state_dict().items()}
    ref_lstm.load_state_dict({k[len('lstm.'):]: v for k, v in sd.items() if k.startswith('lstm.')})
    ref_lin.load_state_dict({'weight': sd['scaling_layer.0.weight'], 'bias': sd['scaling_layer.0.bias']})
    named = [('lstm.' + n, p) for n, p in ref_lstm.named_parameters()] + [('scaling_layer.0.weight', ref_lin.weight), ('scaling_layer.0.bias', ref_lin.bias)]
    assert [n for n, _ in named] == [n for n, _ in head.named_parameters()]
    out, _ = ref_lstm(feats.double())
    log_predicted = ref_lin(out)
    masks = (torch.arange(T)[None] < lens[:, None]).long()
    ref = []
    for b in range(B):
        loss = oobj.l1(log_predicted[b:b + 1], tar[b:b + 1].double(), masks[b:b + 1])
        grads = torch.autograd.grad(loss, [p for _, p in named], retain_graph=True)
        grad = []
        for (key, _), g in zip(named, grads):
            if active_layerid is None:
                grad.append(g.reshape(-1))
            else:
                pattern = re.search(r'lstm.*l(\d+)', key)
                if pattern is not None and int(pattern.group().split('_')[-1][1:]) == active_layerid:
                    grad.append(g.reshape(-1))
        ref.append(torch.cat(grad))
    ref = torch.stack(ref)
    assert got.shape == ref.shape
    for b in range(B):
        cos = torch.nn.functional.cosine_similarity(got[b], ref[b], dim=0).item()
        ratio = (got[b].norm() / ref[b].norm()).item()
        assert cos > 0.99, (b, cos)
        assert abs(ratio - 1) < 0.05, (b, ratio)
    m_got = scoring.matching(got[:2].float(), got.float())
    m_ref = scoring.matching(ref[:2].float(), ref.float())
    assert (m_got - m_ref).abs().max().item() < 0.03
    assert torch.equal(scoring.thresholding(m_got), scoring.thresholding(m_ref))


def test_colsum_groups(gpu):
    from speech_enhancement_by_s3prl_amd import _lib as L
    lib = L.load()
    torch.manual_seed(1)
    G, R, C = 5, 37, 201
    x = torch.randn(G * R, C, device=gpu)
    out = torch.empty(G, C, device=gpu)
    L.check(lib.se_colsum_groups(L.ptr(x), 0, G, R, C, C, L.ptr(out), L.stream()), 'colsum_groups')
    assert (out.double() - x.double().view(G, R, C).sum(1)).abs().max().item() < 1e-4
    xb = torch.randn(G * R, 1024, device=gpu).bfloat16()
    out = torch.empty(G, 1024, device=gpu)
    L.check(lib.se_colsum_groups(L.ptr(xb), 1, G, R, 1024, 1024, L.ptr(out), L.stream()), 'colsum_groups')
    assert (out.double() - xb.double().view(G, R, 1024).sum(1)).abs().max().item() < 1e-4


@pytest.mark.parametrize('lid', [None, 1])
def test_lstm_scoring_vs_reference_golden(gpu, golden_heads, lid):
    """one batched sweep against what the REFERENCE's own sampler.scoring + sampler.matching returned (B sequential backward passes through its
    model.LSTM + objective.L1; tests/golden/make_golden.py: heads_fixture) for the same seeded parameters and fixture features"""
    import numpy as np
    import ref_heads as RH
    from speech_enhancement_by_s3prl_amd import scoring
    from speech_enhancement_by_s3prl_amd.lstm import LSTM
    G = golden_heads
    t = 'all' if lid is None else f'l{lid}'
    T_ = lambda x: torch.from_numpy(np.asarray(x))      # noqa: E731
    feats, tar, lengths = T_(G['score_feats']), T_(G['score_linear_tar']), T_(G['score_lengths'])
    head = RH.seeded.fill_params(LSTM(input_size=RH.HEAD_D, output_size=RH.HEAD_K, hidden_size=256, num_layers=2, bidirectional=True), 300).to(gpu)
    frames = (lengths // 160 + 1).to(gpu)                           # runner.py:455
    got = scoring.per_sample_gradients_lstm(head, feats.to(gpu), tar.to(gpu), frames, active_layerid=lid).double().cpu()
    assert got.shape == (3, int(G[f'score_{t}_numel']))
    idx = RH.seeded.sample_index(got.shape[1], RH.seeded.SCORE_SAMPLES, 4242)
    ref_s, ref_n = T_(G[f'score_{t}_samp']).double(), T_(G[f'score_{t}_norms']).double()
    for b in range(3):
        cos = torch.nn.functional.cosine_similarity(got[b, idx], ref_s[b], dim=0).item()
        ratio = (got[b].norm() / ref_n[b]).item()
        assert cos > 0.99, (b, cos)
        assert abs(ratio - 1) < 0.05, (b, ratio)
    gram = got @ got.t()
    rg = T_(G[f'score_{t}_gram']).double()
    d = rg.diag().sqrt()
    assert ((gram / (got.norm(dim=1)[:, None] * got.norm(dim=1)[None])) - rg / (d[:, None] * d[None])).abs().max().item() < 0.03      # pairwise cosines
    m_got = scoring.matching(got[:2].float(), got.float())
    assert (m_got - T_(G[f'score_{t}_match'])).abs().max().item() < 0.03
    assert torch.equal(scoring.thresholding(m_got), T_(G[f'score_{t}_keep']))
