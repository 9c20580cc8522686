"""GPU parity: rows C1/C2 (heads), D1/D2 (length masks, dB normalise, decode_wav), E1 (L1) -- the HIP path
through the C ABI vs (a) the golden vectors produced by the REFERENCE's own code
(tests/golden/reference_golden.npz) and (b) the CPU oracle at larger sizes."""
import pytest
import torch

from oracle import decode as odec
from oracle import heads as oheads
from oracle import objective as oobj
from oracle import preprocessor as opre

pytestmark = pytest.mark.gpu


def T(x, dev=None):
    t = torch.from_numpy(x)
    return t.to(dev) if dev is not None else t


def test_linear_residual_golden(gpu, golden):
    from speech_enhancement_by_s3prl_amd.heads import LinearResidual
    for tag, cmvn in (('c1', True), ('c1n', False)):
        m = LinearResidual(input_size=120, output_size=201, cmvn=cmvn).to(gpu)
        m.load_state_dict({'linear.weight': T(golden[f'{tag}_weight']), 'linear.bias': T(golden[f'{tag}_bias'])})
        pred, res = m(features=T(golden['c1_feats'], gpu), linears=T(golden['c1_linears'], gpu))
        # fp32 end to end; the reference GEMM sums in a different order -> 1e-5 absolute on O(1) values
        assert torch.allclose(res['offset'].cpu(), T(golden[f'{tag}_offset']), atol=2e-5, rtol=1e-5)
        assert torch.allclose(pred.cpu(), T(golden[f'{tag}_predicted']), atol=5e-5, rtol=1e-4)


def test_linear_golden(gpu, golden):
    from speech_enhancement_by_s3prl_amd.heads import Linear
    m = Linear(120, 201, activation='ReLU').to(gpu)
    m.load_state_dict({'linear.weight': T(golden['c2_weight']), 'linear.bias': T(golden['c2_bias'])})
    pred, res = m(features=T(golden['c1_feats'], gpu))
    assert res == {}
    assert torch.allclose(pred.cpu(), T(golden['c2_predicted']), atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize('B,F,D,N', [(2, 1001, 120, 201), (3, 77, 768, 201), (1, 130, 40, 64)])
def test_linear_residual_vs_oracle(gpu, B, F, D, N):
    from speech_enhancement_by_s3prl_amd.heads import LinearResidual
    torch.manual_seed(B * F)
    feats = torch.randn(B, F, D) * 3 + 1
    lin = torch.rand(B, F, N) * 5
    m = LinearResidual(input_size=D, output_size=N, cmvn=True)
    ref_pred, ref_res = oheads.linear_residual(feats, lin, m.linear.weight.detach(), m.linear.bias.detach())
    m = m.to(gpu)
    pred, res = m(features=feats.to(gpu), linears=lin.to(gpu))
    assert torch.allclose(res['offset'].cpu(), ref_res['offset'], atol=3e-5, rtol=1e-4)
    assert ((pred.cpu() - ref_pred).abs().max() / ref_pred.abs().max()).item() < 1e-4


def test_linear_residual_lazy_offset(gpu, golden):
    """evaluate() (no gradients): `offset` (model.py:34) comes back as a LazyTensor that costs nothing until something reads it, and then holds
    exactly what the eager path returns; `predicted` is the eager value bit for bit"""
    from speech_enhancement_by_s3prl_amd.heads import LinearResidual
    from speech_enhancement_by_s3prl_amd.preprocessor import LazyTensor
    m = LinearResidual(input_size=120, output_size=201, cmvn=True).to(gpu)
    m.load_state_dict({'linear.weight': T(golden['c1_weight']), 'linear.bias': T(golden['c1_bias'])})
    feats, lin = T(golden['c1_feats'], gpu), T(golden['c1_linears'], gpu)
    pred_e, res_e = m(features=feats, linears=lin)                 # gradients enabled: the eager autograd path
    with torch.no_grad():
        pred, res = m(features=feats, linears=lin)
    off = res['offset']
    assert type(off) is LazyTensor and off._value is None
    assert off.shape == res_e['offset'].shape and off.dtype == torch.float32 and off.device == pred.device
    assert torch.equal(pred, pred_e.detach())

    def ignores(**kwargs):          # a criterion that takes **kwargs and drops the mask (objective.py:103, 81): nothing is computed
        return None
    ignores(**res)
    assert off._value is None
    assert torch.equal(off + 0.0, res_e['offset'].detach())        # first real use materialises it
    assert off._value is not None
    assert torch.allclose(off.cpu(), T(golden['c1_offset']), atol=2e-5, rtol=1e-5)
    m.lazy_offset = False
    with torch.no_grad():
        _, res2 = m(features=feats, linears=lin)
    assert type(res2['offset']) is torch.Tensor


def test_length_masks_golden(gpu, golden):
    from speech_enhancement_by_s3prl_amd.decode import get_length_masks
    m = get_length_masks(T(golden['d1_lengths'], gpu))
    assert m.dtype == torch.int64
    assert torch.equal(m.cpu(), T(golden['d1_masks']))        # integer work: bit exact


def test_masked_normalize_decibel_golden(gpu, golden):
    from speech_enhancement_by_s3prl_amd.decode import masked_normalize_decibel
    wav, ref, lens = T(golden['d2_wav'], gpu), T(golden['d2_ref'], gpu), T(golden['d2_lengths'], gpu)
    fixed = masked_normalize_decibel(wav, -25, lens)
    toref = masked_normalize_decibel(wav, ref, lens)
    assert torch.allclose(fixed.cpu(), T(golden['d2_fixed']), rtol=1e-5, atol=1e-7)
    assert torch.allclose(toref.cpu(), T(golden['d2_toref']), rtol=1e-5, atol=1e-7)


def test_l1_golden_value_and_grad(gpu, golden):
    from speech_enhancement_by_s3prl_amd.decode import get_length_masks
    from speech_enhancement_by_s3prl_amd.objective import L1
    lp = T(golden['e1_log_predicted'], gpu).requires_grad_(True)
    masks = get_length_masks(T(golden['d1_lengths'], gpu))
    loss, extra = L1()(log_predicted=lp, linear_tar=T(golden['e1_linear_tar'], gpu), stft_length_masks=masks,
                       predicted=None, lengths=None, some_runner_local=1)      # swallowed by **kwargs (runner.py:458)
    assert extra == {}
    assert abs(loss.item() - float(golden['e1_loss'])) < 1e-5 * abs(float(golden['e1_loss']))
    loss.backward()
    assert torch.allclose(lp.grad.cpu(), T(golden['e1_grad']), rtol=1e-5, atol=1e-9)


def test_decode_wav_vs_oracle_ragged(gpu):
    """runner.py:266-270 + evaluate()'s 'normalise to the clean wav' quirk (runner.py:570), ragged lengths."""
    from speech_enhancement_by_s3prl_amd import synth
    from speech_enhancement_by_s3prl_amd.decode import decode_wav
    from speech_enhancement_by_s3prl_amd.preprocessor import OnlinePreprocessor
    geom = opre.Geometry()
    lengths, wavs = synth.synth_batch(3, 32000, ragged=True)
    P = OnlinePreprocessor().to(gpu)
    fl = [P.get_feat_config('linear', 0), P.get_feat_config('phase', 0)]
    lin, ph = P(wavs.to(gpu), fl)
    mask = torch.rand(3, lin.shape[1], 201)
    rlin, rph = opre.forward(wavs, fl, geom)
    for target, rtarget in ((-25, -25), (wavs[:, 1].to(gpu), wavs[:, 1])):
        got = decode_wav(P, lin * mask.to(gpu), ph, lengths.to(gpu), target)
        ref = odec.decode_wav(rlin * mask, rph, lengths, geom, rtarget)
        assert got.shape == ref.shape
        assert ((got.cpu() - ref).abs().max() / ref.abs().max()).item() < 1e-4


@pytest.mark.parametrize('B,F', [(3, 50), (32, 1001), (70, 17)])
def test_l1_one_launch_form_equals_the_two_step_form(gpu, B, F):
    """se_l1_masked_loss_f32 (frame counts derived from waveform lengths inside the kernel, sums + loss published by the last workgroup, scratch
    left zero) against the sums / divide form and the oracle; called repeatedly (the scratch is self-cleaning), with gradients."""
    from oracle import decode as odec2
    from oracle import objective as oobj
    from speech_enhancement_by_s3prl_amd.objective import L1
    torch.manual_seed(B * 1000 + F)
    K = 201
    lp = torch.randn(B, F, K)
    tar = torch.rand(B, F, K) + 1e-3
    wav_len = torch.randint(160, 160 * (F - 1) + 1, (B,))
    wav_len[0] = 160 * (F - 1)
    frames = wav_len // 160 + 1
    ref = oobj.l1(lp, tar, odec2.get_length_masks(frames, F))
    crit = L1()
    for rep in range(3):
        a = lp.to(gpu).requires_grad_(True)
        loss, _ = crit(log_predicted=a, linear_tar=tar.to(gpu), wav_lengths=wav_len.to(gpu), hop=160)
        loss.backward()
        b = lp.to(gpu).requires_grad_(True)
        crit.reduce_fn = lambda t: t                      # forces the sums / divide form
        loss2, _ = crit(log_predicted=b, linear_tar=tar.to(gpu), stft_lengths=frames.to(gpu))
        crit.reduce_fn = None
        loss2.backward()
        assert abs(loss.item() - ref.item()) < 1e-5 * abs(ref.item()), (rep, loss.item(), ref.item())
        assert abs(loss.item() - loss2.item()) < 1e-6 * abs(ref.item())
        assert torch.equal(a.grad, b.grad)
    from speech_enhancement_by_s3prl_amd import objective as prod_obj
    for t in prod_obj._L1_SCRATCH.values():
        assert t[0].item() == 0                           # the arrival ticket is self-cleaning


def test_lazy_results_reach_kernels_as_values(gpu, golden):
    """a LazyTensor handed to a HIP kernel (here: LinearResidual's lazy `offset` into the WSD criterion, objective.py:119-153, under no_grad as in
    evaluate()) arrives as its value: `.contiguous()` / `.float()` on the wrapper are no-ops, so _lib.ptr() is where it materialises"""
    from speech_enhancement_by_s3prl_amd.heads import LinearResidual
    from speech_enhancement_by_s3prl_amd.objective import WSD
    from speech_enhancement_by_s3prl_amd.preprocessor import LazyTensor
    m = LinearResidual(input_size=120, output_size=201, cmvn=True).to(gpu)
    m.load_state_dict({'linear.weight': T(golden['c1_weight']), 'linear.bias': T(golden['c1_bias'])})
    feats, lin = T(golden['c1_feats'], gpu), T(golden['c1_linears'], gpu)
    tar = lin * 0.5 + 0.01
    lens = torch.full((feats.shape[0],), feats.shape[1], device=gpu)
    crit = WSD()
    with torch.no_grad():
        pred, res = m(features=feats, linears=lin)
        assert type(res['offset']) is LazyTensor
        lazy_loss, _ = crit(predicted=pred, linear_inp=lin, linear_tar=tar, stft_lengths=lens, **res)
    pred_e, res_e = m(features=feats, linears=lin)
    eager_loss, _ = crit(predicted=pred_e.detach(), linear_inp=lin, linear_tar=tar, stft_lengths=lens, offset=res_e['offset'].detach())
    assert torch.equal(lazy_loss, eager_loss.detach())
