"""Generate golden fixtures from the reference's OWN arithmetic (run in the build container only).

The reference (/root/reference) is imported here, never on the GPU box. Third-party names that
are absent offline (S3PRL packages, torchaudio, librosa, asteroid, pesq, pystoi, tensorboardX,
joblib extras, matplotlib) are replaced by empty stub modules so that the reference's own
modules load; only functions whose arithmetic lives entirely in the reference's files are
called.  Inputs are seeded synthetic tensors; inputs + outputs are written to
tests/golden/reference_golden.npz (small: B=3, T=16, 201 bins).

Rows pinned: C1 LinearResidual, C2 Linear, D1 length masks,
D2 masked_mean / masked_normalize_decibel, E1 L1, SISDR and WSD objectives (values and gradients), add_noise,
OnlineDataset.normalize_wav_decibel, OnlineDataset.collate_fn, sampler.matching,
evaluation.sisdr_eval.

Usage:  python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'reference_golden.npz')


class _Anything:
    """Attribute sink: any attribute access / call returns another sink."""
    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        return _Anything

    def __call__(self, *a, **k):
        return _Anything()


def _stub(name, **attrs):
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    mod.__getattr__ = lambda attr: _Anything  # PEP 562
    sys.modules[name] = mod
    return mod


def install_stubs():
    import scipy
    import scipy.signal.windows
    if not hasattr(scipy, 'hanning'):
        scipy.hanning = scipy.signal.windows.hann  # utils.py:13 uses the removed alias
    for name in ['utility', 'utility.preprocessor', 'transformer', 'transformer.nn_transformer',
                 'transformer.model', 'downstream', 'downstream.model', 'downstream.solver',
                 'torchaudio', 'librosa', 'librosa.util', 'asteroid', 'asteroid.losses',
                 'asteroid.losses.sdr', 'asteroid.losses.stoi', 'asteroid.losses.pmsqe',
                 'pesq', 'pystoi', 'tensorboardX', 'matplotlib', 'matplotlib.pyplot', 'ipdb']:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                _stub(name)


def main():
    install_stubs()
    sys.path.insert(0, REF)
    import model as ref_model
    import objective as ref_objective
    import utils as ref_utils
    import dataset as ref_dataset
    import sampler as ref_sampler
    import evaluation as ref_eval

    g = torch.Generator().manual_seed(20201004)
    B, T, D, K = 3, 16, 120, 201
    out = {}

    feats = torch.randn(B, T, D, generator=g)
    linears = torch.rand(B, T, K, generator=g) * 3.0 + 1e-3
    linear_tar = torch.rand(B, T, K, generator=g) * 2.0 + 1e-4
    stft_lengths = torch.tensor([16, 11, 1])
    asc = torch.arange(1000)

    # ---- C1 LinearResidual (model.py:20-34)
    torch.manual_seed(7)
    lr = ref_model.LinearResidual(input_size=D, output_size=K, cmvn=True)
    pred, res = lr(features=feats, linears=linears)
    out.update(c1_feats=feats, c1_linears=linears, c1_weight=lr.linear.weight.detach(),
               c1_bias=lr.linear.bias.detach(), c1_predicted=pred.detach(), c1_offset=res['offset'].detach())
    lr2 = ref_model.LinearResidual(input_size=D, output_size=K, cmvn=False)
    pred2, res2 = lr2(features=feats, linears=linears)
    out.update(c1n_weight=lr2.linear.weight.detach(), c1n_bias=lr2.linear.bias.detach(),
               c1n_predicted=pred2.detach(), c1n_offset=res2['offset'].detach())

    # ---- C2 Linear (model.py:8-17)
    lin = ref_model.Linear(D, K, activation='ReLU')
    pred3, _ = lin(features=feats)
    out.update(c2_weight=lin.linear.weight.detach(), c2_bias=lin.linear.bias.detach(), c2_predicted=pred3.detach())

    # ---- D1 length masks (sampler.py:35-39 == runner.py:216-220)
    masks = ref_sampler.get_length_masks(stft_lengths, asc)
    out.update(d1_lengths=stft_lengths, d1_masks=masks)

    # ---- D2 masked_mean / masked_normalize_decibel (utils.py:26-46)
    wav_len = torch.tensor([2000, 1250, 1999])
    wmask = ref_sampler.get_length_masks(wav_len, torch.arange(10000))
    wav = torch.randn(B, 2000, generator=g) * 0.05
    refwav = torch.randn(B, 2000, generator=g) * 0.2
    out.update(d2_wav=wav, d2_ref=refwav, d2_lengths=wav_len,
               d2_mean=ref_utils.masked_mean(wav.pow(2), wmask),
               d2_fixed=ref_utils.masked_normalize_decibel(wav, -25, wmask),
               d2_toref=ref_utils.masked_normalize_decibel(wav, refwav, wmask))

    # ---- E1 L1 (objective.py:103-117) value and gradient
    logp = torch.randn(B, T, K, generator=g).requires_grad_(True)
    l1 = ref_objective.L1()
    loss, _ = l1(log_predicted=logp, linear_tar=linear_tar, stft_length_masks=masks)
    loss.backward()
    out.update(e1_log_predicted=logp.detach(), e1_linear_tar=linear_tar, e1_loss=loss.detach(), e1_grad=logp.grad)

    # ---- SISDR objective (objective.py:81-100)
    sis, _ = ref_objective.SISDR()(predicted=linears, linear_tar=linear_tar, stft_length_masks=masks)
    out.update(sisdr_obj=sis.detach())

    # ---- full C1 -> E1 chain gradient wrt head params (runner.py:453-459)
    lr.zero_grad()
    pred, res = lr(features=feats, linears=linears)
    logpred = (pred + 1e-10).log()  # LinearResidual returns no log_predicted; runner users pair it with log-free losses;
    loss2, _ = l1(log_predicted=logpred, linear_tar=linear_tar, stft_length_masks=masks)
    loss2.backward()
    out.update(chain_loss=loss2.detach(), chain_gw=lr.linear.weight.grad.clone(), chain_gb=lr.linear.bias.grad.clone())

    # ---- add_noise (dataset.py:54-74), normalize_wav_decibel (dataset.py:106-111), collate_fn (dataset.py:169-179)
    # NB: the reference only ever calls add_noise at batch 1 (dataset.py:157, sampler.py:51); with batch > 1 its
    # (B,) x (B,1) broadcast raises, so the fixture uses the same call shape.
    speech = torch.randn(1, 1000, generator=g) * 0.1
    noise_short = torch.randn(1, 300, generator=g)
    noise_long = torch.randn(1, 1500, generator=g)
    snrs = torch.ones(1) * -8.0
    noisy_s, scaled_s = ref_dataset.add_noise(speech, noise_short, snrs)
    noisy_l, scaled_l = ref_dataset.add_noise(speech, noise_long, snrs)
    out.update(an_speech=speech, an_noise_short=noise_short, an_noise_long=noise_long, an_snrs=snrs,
               an_noisy_s=noisy_s, an_scaled_s=scaled_s, an_noisy_l=noisy_l, an_scaled_l=scaled_l)
    ds = ref_dataset.OnlineDataset.__new__(ref_dataset.OnlineDataset)
    ds.target_level = -25
    out.update(nwd_in=speech[0], nwd_out=ds.normalize_wav_decibel(speech[0]))
    samples = [torch.randn(n, 3, generator=g) for n in (70, 100, 45)]
    lengths, wavs = ds.collate_fn(samples)
    out.update(col_s0=samples[0], col_s1=samples[1], col_s2=samples[2], col_lengths=lengths, col_wavs=wavs)

    # ---- matching (sampler.py:113-116), sisdr_eval (evaluation.py:5-10)
    q = torch.randn(5, 64, generator=g)
    k = torch.randn(7, 64, generator=g)
    out.update(match_q=q, match_k=k, match_scores=ref_sampler.matching(q, k))
    a = torch.randn(3000, generator=g)
    b = a + 0.3 * torch.randn(3000, generator=g)
    out.update(se_src=b, se_tar=a, se_val=torch.tensor(ref_eval.sisdr_eval(b, a)))

    # ---- appended (new generator draws only AFTER everything above, so the older entries keep their values):
    # SISDR gradient wrt predicted and WSD value / gradient wrt offset (objective.py:81-100, 119-153)
    pred_leaf = (linears.detach().clone() * (torch.rand(B, T, K, generator=g) + 0.5)).requires_grad_(True)
    sis2, _ = ref_objective.SISDR()(predicted=pred_leaf, linear_tar=linear_tar, stft_length_masks=masks)
    sis2.backward()
    out.update(sis_pred=pred_leaf.detach(), sis_loss=sis2.detach(), sis_grad=pred_leaf.grad.clone())
    off_leaf = torch.rand(B, T, K, generator=g).requires_grad_(True)
    lin_inp = linear_tar + 0.3 * torch.randn(B, T, K, generator=g).abs() * torch.rand(B, T, K, generator=g).round()   # noisy >= clean on ~half the bins
    wsd, _ = ref_objective.WSD()(linear_inp=lin_inp, offset=off_leaf, linear_tar=linear_tar, stft_length_masks=masks)
    wsd.backward()
    out.update(wsd_inp=lin_inp, wsd_offset=off_leaf.detach(), wsd_loss=wsd.detach(), wsd_grad=off_leaf.grad.clone())

    np.savez_compressed(OUT, **{k: (v.detach().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()})
    print('wrote', OUT, {k: tuple(np.asarray(v).shape) for k, v in out.items()})


if __name__ == '__main__':
    main()
