"""GPU parity of the round-5 fusions of the configs[3] / configs[0] pass (vcb.yaml / pseudo_noise.yaml, runner.py:556-575 with a feature-input
mask head): the one-pass feature launch (se_features3_f32) against the two-launch form and the CPU oracle, the column statistics it hands to the
head, the head on a ready-made weight split + statistics (se_head_linear_pre_f32) against se_head_linear_f32, the two-launch SISDR criterion
against the reference-golden values, and the assembled HeadEnhanceStep against the un-fused composition of the same calls."""
import pytest
import torch

from oracle import preprocessor as opre

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(a)


def _features2(lib, L, raw, tm, log, delta, cmvn, eps, pad=0):
    B, D, F = (raw.shape[0], raw.shape[2], raw.shape[1]) if tm else raw.shape
    Dout = D * (1 + delta)
    out = torch.empty(B, F, Dout, device=raw.device)
    n = lib.se_features_workspace_bytes(B, D, F, delta)
    ws = torch.empty(n, device=raw.device, dtype=torch.uint8)
    xin = torch.zeros(B * F, pad, device=raw.device, dtype=torch.bfloat16) if pad else None
    valid = torch.empty(B, device=raw.device, dtype=torch.int32) if pad else None
    L.check(lib.se_features2_f32(L.ptr(raw), int(tm), B, D, F, int(log), delta, int(cmvn), eps, L.ptr(out), L.ptr(ws), n, L.ptr(xin), pad, L.ptr(valid),
                                 L.stream()), 'se_features2_f32')
    return out, xin, valid


def _features3(lib, L, raw, tm, log, delta, cmvn, eps, pad=0, colstats_eps=None):
    B, D, F = (raw.shape[0], raw.shape[2], raw.shape[1]) if tm else raw.shape
    Dout = D * (1 + delta)
    out = torch.empty(B, F, Dout, device=raw.device)
    n = max(lib.se_features3_workspace_bytes(B, D, delta), lib.se_features3_colstats_workspace_bytes(B, D, F, delta))
    ws = torch.empty(n, device=raw.device, dtype=torch.uint8)
    xin = torch.zeros(B * F, pad, device=raw.device, dtype=torch.bfloat16) if pad else None
    valid = torch.empty(B, device=raw.device, dtype=torch.int32) if pad else None
    cst = torch.empty(B, Dout, 2, device=raw.device) if colstats_eps is not None else None
    L.check(lib.se_features3_f32(L.ptr(raw), int(tm), B, D, F, int(log), delta, int(cmvn), eps, L.ptr(out), L.ptr(ws), n, L.ptr(xin), pad, L.ptr(valid),
                                 L.ptr(cst), float(colstats_eps or 0.0), L.stream()), 'se_features3_f32')
    return out, xin, valid, cst


@pytest.mark.parametrize('B,D,F,tm,log,delta,cmvn', [
    (3, 40, 1001, False, True, 2, False),     # pseudo_noise.yaml baseline: mel / log / delta 2
    (2, 40, 1001, False, True, 1, True),      # pretrain_sample.yaml input: mel / log / delta 1 / cmvn
    (2, 40, 37, False, True, 3, True),        # one ragged tile beside a full one, the deepest halo
    (5, 40, 2, False, True, 2, False),        # two frames: every tap is a replicate
    (1, 40, 5, False, False, 2, True),
    (2, 201, 130, True, True, 0, True),       # time-major raw plane (linear / log / cmvn)
    (2, 201, 64, True, False, 1, False),
    (9, 13, 33, True, False, 2, True),
])
def test_one_pass_features_equal_the_two_launch_form(gpu, B, D, F, tm, log, delta, cmvn):
    from speech_enhancement_by_s3prl_amd import _lib as L
    lib = L.load()
    torch.manual_seed(B * 1000 + F)
    raw = (torch.rand(B, F, D) if tm else torch.rand(B, D, F)).mul(4).add(1e-3).to(gpu)
    if B > 2:
        raw[1] = 0.0 if not log else raw[1]          # an all-zero utterance (valid-frame rule) where log is not taken
    pad = 128 if D * (1 + delta) <= 128 else 0
    o2, x2, v2 = _features2(lib, L, raw, tm, log, delta, cmvn, 1e-10, pad)
    o3, x3, v3, _ = _features3(lib, L, raw, tm, log, delta, cmvn, 1e-10, pad)
    # same expressions on the same inputs: bit-identical unless the compiler contracts one of the two delta expressions differently (then 1 ulp)
    assert torch.allclose(o3, o2, rtol=0, atol=2e-6 * max(1.0, o2.abs().max().item()))
    if pad:
        assert (x3.float() - x2.float()).abs().max().item() <= 2e-2 * max(1.0, o2.abs().max().item()) * 2 ** -7
        assert torch.equal(v3, v2)


@pytest.mark.parametrize('B,F,delta', [(2, 1001, 2), (3, 130, 1), (1, 40, 0)])
def test_one_pass_features_vs_oracle(gpu, B, F, delta):
    from speech_enhancement_by_s3prl_amd import _lib as L
    lib = L.load()
    torch.manual_seed(F)
    mel = torch.rand(B, 40, F).mul(3).add(1e-4)
    for cmvn in (False, True):
        ref = opre.select_feat(mel, 1e-10, log=True, delta=delta, cmvn=cmvn).transpose(1, 2)      # (B, D', F) -> time-major
        out = _features3(lib, L, mel.to(gpu), False, True, delta, cmvn, 1e-10)[0]
        assert (out.cpu() - ref).abs().max().item() < 1e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize('B,D,F,delta', [(3, 40, 1001, 2), (2, 40, 33, 1), (2, 201, 77, 0)])
def test_feature_launch_hands_the_head_its_column_statistics(gpu, B, D, F, delta):
    from speech_enhancement_by_s3prl_amd import _lib as L
    lib = L.load()
    torch.manual_seed(D + F)
    raw = torch.rand(B, D, F).mul(5).add(1e-3).to(gpu)
    out, _, _, cst = _features3(lib, L, raw, False, True, delta, False, 1e-10, colstats_eps=1e-6)
    ref = torch.empty_like(cst)
    L.check(lib.se_head_colstats_f32(L.ptr(out), B, F, D * (1 + delta), 1e-6, L.ptr(ref), L.stream()), 'se_head_colstats_f32')
    o = out.double()
    mean, std = o.mean(dim=1), o.std(dim=1, unbiased=True)
    assert torch.allclose(cst[..., 0].double(), mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(cst[..., 1].double(), 1.0 / (std + 1e-6), rtol=2e-5)
    assert torch.allclose(cst, ref, rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize('B,F,D,N,cmvn,act', [
    (2, 1001, 120, 201, True, 2),      # the branch-free kernel (201 bins, Sigmoid, F >= 128): pseudo_noise.yaml's features
    (3, 1001, 201, 201, True, 2),      # vcb.yaml's linear features: D % 4 != 0, the last 16-B piece of a row is shifted back into the tensor
    (5, 130, 120, 201, True, 2),       # every 128-row tile straddles two utterances
    (2, 128, 13, 201, False, 1),       # ReLU, no CMVN, one K chunk
    (1, 300, 36, 193, True, 2),        # 193 bins: the smallest width with seven column tiles; two K chunks + a partial third
    (3, 77, 201, 201, True, 2),        # F < 128: the general kernel
    (1, 130, 40, 64, False, 2),        # two column tiles: the general kernel
])
def test_head_on_ready_made_split_and_statistics_is_bit_identical(gpu, B, F, D, N, cmvn, act):
    from speech_enhancement_by_s3prl_amd import _lib as L
    lib = L.load()
    torch.manual_seed(B * F)
    feats = (torch.randn(B, F, D) * 3 + 1).to(gpu)
    lin = (torch.rand(B, F, N) * 5).to(gpu)
    W, bias = (torch.randn(N, D) * 0.1).to(gpu), torch.randn(N).to(gpu)
    n = lib.se_head_workspace_bytes(B, F, D, N)
    ws = torch.empty(n, device=gpu, dtype=torch.uint8)
    p0, o0 = torch.empty(B, F, N, device=gpu), torch.empty(B, F, N, device=gpu)
    L.check(lib.se_head_linear_f32(L.ptr(feats), L.ptr(W), L.ptr(bias), L.ptr(lin), B, F, D, N, act, int(cmvn), 1e-6, L.ptr(p0), L.ptr(o0), L.ptr(ws), n, L.stream()),
            'se_head_linear_f32')
    w3 = torch.empty(lib.se_head_w3_bytes(N, D), device=gpu, dtype=torch.uint8)
    L.check(lib.se_head_split_weights_f32(L.ptr(W), N, D, L.ptr(w3), L.stream()), 'se_head_split_weights_f32')
    stats = None
    if cmvn:
        stats = torch.empty(B, D, 2, device=gpu)
        L.check(lib.se_head_colstats_f32(L.ptr(feats), B, F, D, 1e-6, L.ptr(stats), L.stream()), 'se_head_colstats_f32')
    p1, o1 = torch.empty(B, F, N, device=gpu), torch.empty(B, F, N, device=gpu)
    L.check(lib.se_head_linear_pre_f32(L.ptr(feats), L.ptr(w3), L.ptr(bias), L.ptr(lin), L.ptr(stats), B, F, D, N, act, L.ptr(p1), L.ptr(o1), L.stream()),
            'se_head_linear_pre_f32')
    # same splits, same products in the same order, same activation instructions: the two kernels agree to the bit
    assert torch.equal(p1, p0) and torch.equal(o1, o0)
    # mask only / product only (the lazy `offset` of evaluate())
    p2 = torch.empty(B, F, N, device=gpu)
    L.check(lib.se_head_linear_pre_f32(L.ptr(feats), L.ptr(w3), L.ptr(bias), None, L.ptr(stats), B, F, D, N, act, L.ptr(p2), None, L.stream()), 'se_head_linear_pre_f32')
    assert torch.equal(p2, o0)


@pytest.mark.parametrize('B,F,D,N', [(4, 1001, 120, 201), (5, 130, 120, 201), (3, 200, 201, 201), (2, 77, 120, 201)])
def test_head_with_the_criterion_sums_taken_in_its_epilogue(gpu, B, F, D, N):
    """se_head_linear_sisdr_f32 = se_head_linear_pre_f32 + se_sisdr_spec_loss_f32: ragged lengths (one utterance shorter than a tile, one empty
    after the first frame, one full), tiles that straddle two utterances; (2, 77, ...) takes the un-fused route inside."""
    from speech_enhancement_by_s3prl_amd import _lib as L
    lib = L.load()
    torch.manual_seed(B + F)
    feats = (torch.randn(B, F, D) * 2).to(gpu)
    lin, tar = (torch.rand(B, F, N) * 5).to(gpu), (torch.rand(B, F, N) * 5).to(gpu)
    W, bias = (torch.randn(N, D) * 0.1).to(gpu), torch.randn(N).to(gpu)
    frames = torch.tensor([F, 1, max(F // 3, 1), F - 1, 100][:B], dtype=torch.int64)
    wav_len = ((frames - 1) * 160 + 31).to(gpu)
    w3 = torch.empty(lib.se_head_w3_bytes(N, D), device=gpu, dtype=torch.uint8)
    L.check(lib.se_head_split_weights_f32(L.ptr(W), N, D, L.ptr(w3), L.stream()), 'se_head_split_weights_f32')
    stats = torch.empty(B, D, 2, device=gpu)
    L.check(lib.se_head_colstats_f32(L.ptr(feats), B, F, D, 1e-6, L.ptr(stats), L.stream()), 'se_head_colstats_f32')
    p0 = torch.empty(B, F, N, device=gpu)
    L.check(lib.se_head_linear_pre_f32(L.ptr(feats), L.ptr(w3), L.ptr(bias), L.ptr(lin), L.ptr(stats), B, F, D, N, 2, L.ptr(p0), None, L.stream()), 'pre')
    sc0 = torch.empty(lib.se_sisdr_spec_loss_scratch_doubles(B, F, N), device=gpu, dtype=torch.float64)
    lb0, s0, l0 = torch.empty(B, device=gpu), torch.empty(2, device=gpu, dtype=torch.float64), torch.empty((), device=gpu)
    L.check(lib.se_sisdr_spec_loss_f32(L.ptr(p0), L.ptr(tar), L.ptr(wav_len), 160, B, F, N, 1e-10, L.ptr(sc0), L.ptr(lb0), L.ptr(s0), L.ptr(l0), L.stream()), 'loss')
    p1 = torch.empty(B, F, N, device=gpu)
    sc1 = torch.empty(lib.se_head_sisdr_scratch_doubles(B, F, N), device=gpu, dtype=torch.float64)
    lb1, s1, l1 = torch.empty(B, device=gpu), torch.empty(2, device=gpu, dtype=torch.float64), torch.empty((), device=gpu)
    L.check(lib.se_head_linear_sisdr_f32(L.ptr(feats), L.ptr(w3), L.ptr(bias), L.ptr(lin), L.ptr(stats), B, F, D, N, 2, L.ptr(p1), None, L.ptr(tar), L.ptr(wav_len),
                                         160, 1e-10, L.ptr(sc1), L.ptr(lb1), L.ptr(s1), L.ptr(l1), L.stream()), 'fused')
    assert torch.equal(p1, p0)
    # the fused form sums a pass's 28 elements per lane in fp32 with hardware square roots: 1e-6 relative on the sums
    assert torch.allclose(lb1, lb0, rtol=1e-5, atol=1e-4)
    assert abs(l1.item() - l0.item()) <= 1e-5 * abs(l0.item()) + 1e-5 and s1[1].item() == B
    # and against the definition (objective.py:81-100) in fp64
    src, ref = p0.double().clamp(min=0).sqrt(), tar.double().clamp(min=0).sqrt()
    m = (torch.arange(F, device=gpu)[None, :] < frames.to(gpu)[:, None]).double()[:, :, None]
    src, ref = (src * m).flatten(1), (ref * m).flatten(1)
    a = (src * ref).sum(1, keepdim=True) / ((ref * ref).sum(1, keepdim=True) + 1e-10)
    want = -10 * torch.log10((a * ref).pow(2).sum(1) / ((a * ref - src).pow(2).sum(1) + 1e-10) + 1e-10)
    assert torch.allclose(lb1.double(), want, rtol=1e-5, atol=1e-4)


def test_sisdr_inference_criterion_golden_and_equal_to_the_autograd_form(gpu, golden):
    from speech_enhancement_by_s3prl_amd.objective import SISDR, sisdr_loss_inference
    crit = SISDR()
    masks, tar = T(golden['d1_masks']).to(gpu), T(golden['e1_linear_tar']).to(gpu)
    pred = T(golden['c1_linears']).to(gpu)
    with torch.no_grad():
        loss, _ = crit(predicted=pred, linear_tar=tar, stft_length_masks=masks)           # the two-launch inference form
    assert abs(loss.item() - float(golden['sisdr_obj'])) < 1e-4 * abs(float(golden['sisdr_obj']))
    lg, _ = crit(predicted=pred.clone().requires_grad_(True), linear_tar=tar, stft_length_masks=masks)      # sums + final + torch tail
    assert abs(loss.item() - lg.item()) < 1e-6 * abs(lg.item())
    # waveform lengths + hop (runner.py:455) give the same frame counts as the masks
    frames = masks.sum(dim=-1)
    wav_len = (frames - 1) * 160 + 7
    l2, lb = sisdr_loss_inference(pred, tar, wav_len, 160, crit.eps)
    assert torch.equal(l2, loss)
    assert abs(lb.double().mean().item() - loss.item()) < 1e-6 * abs(loss.item())


@pytest.mark.parametrize('feat', ['mel120', 'linear201'])
def test_head_enhance_step_equals_its_unfused_composition(gpu, feat):
    """HeadEnhanceStep (one-pass features, statistics handed over, cached weight split, two-launch criterion) against the same objects called the
    round-4 way: two-launch features, the head computing its own statistics, the autograd-form criterion on `lengths // hop + 1`."""
    from speech_enhancement_by_s3prl_amd import decode, pipeline, synth
    from speech_enhancement_by_s3prl_amd.heads import LinearResidual
    from speech_enhancement_by_s3prl_amd.objective import SISDR, _SISDRFn
    cfg = pipeline.make_config()
    down = {'mel120': None, 'linear201': {'feat_type': 'linear', 'log': False, 'delta': 0, 'cmvn': False}}[feat]
    pre = pipeline.build_preprocessor(cfg, gpu, downstream_feat=down, upstream='baseline')
    torch.manual_seed(5)
    head = LinearResidual(input_size=120 if feat == 'mel120' else 201, output_size=201, cmvn=True).to(gpu)
    step = pipeline.HeadEnhanceStep(pre, head, criterion=SISDR())
    lengths, wavs = synth.synth_batch(3, 48000, ragged=True)
    wavs, lengths = wavs[:, :2].contiguous().to(gpu), lengths.to(gpu)
    wav_pred, predicted, lin_tar, loss = step(wavs, lengths)
    f = pre(wavs)
    if feat == 'mel120':
        assert getattr(f[1], '_se_colstats', None) is not None       # the hand-off happened
    # the un-fused composition
    pre2 = pipeline.build_preprocessor(cfg, gpu, downstream_feat=down, upstream='baseline')
    pre2.one_pass_features = False
    f2 = pre2(wavs)
    assert getattr(f2[1], '_se_colstats', None) is None
    with torch.no_grad():
        p2, _ = head(features=f2[1], linears=f2[2])
    w2 = decode.decode_wav(pre2, p2, f2[3], lengths, wavs[:, 1])
    l2 = _SISDRFn.apply(p2, f2[4], lengths // 160 + 1, 1e-10, None)
    assert torch.allclose(predicted, p2, rtol=2e-5, atol=2e-5 * p2.abs().max().item())
    assert torch.allclose(wav_pred, w2, rtol=0, atol=2e-5 * w2.abs().max().item())
    assert abs(loss.item() - l2.item()) < 1e-5 * abs(l2.item())


def test_head_weight_split_follows_the_parameter(gpu):
    """the cached three-term split is rebuilt after an in-place update (optimizer step) and after load_state_dict"""
    from speech_enhancement_by_s3prl_amd.heads import LinearResidual
    torch.manual_seed(3)
    head = LinearResidual(input_size=40, output_size=64, cmvn=True).to(gpu)
    feats, lin = torch.randn(2, 50, 40, device=gpu), torch.rand(2, 50, 64, device=gpu)
    with torch.no_grad():
        p0, _ = head(features=feats, linears=lin)
        head.linear.weight.mul_(0.5)
        p1, _ = head(features=feats, linears=lin)
    fresh = LinearResidual(input_size=40, output_size=64, cmvn=True).to(gpu)
    fresh.load_state_dict(head.state_dict())
    with torch.no_grad():
        p2, _ = fresh(features=feats, linears=lin)
    assert not torch.equal(p0, p1)
    assert torch.equal(p1, p2)


def test_head_enhance_step_under_inference_mode_and_with_other_criteria(gpu):
    """tensors made under torch.inference_mode() track no version counter: every hand-over that is tied to one (statistics, criterion sums) steps
    aside and the plain calls run; a criterion other than SISDR never sees the fused route"""
    from speech_enhancement_by_s3prl_amd import pipeline, synth
    from speech_enhancement_by_s3prl_amd.heads import LinearResidual
    from speech_enhancement_by_s3prl_amd.objective import SISDR, WSD
    cfg = pipeline.make_config()
    pre = pipeline.build_preprocessor(cfg, gpu, upstream='baseline')
    torch.manual_seed(9)
    head = LinearResidual(input_size=120, output_size=201, cmvn=True).to(gpu)
    lengths, wavs = synth.synth_batch(2, 32000, ragged=True)
    wavs, lengths = wavs[:, :2].contiguous().to(gpu), lengths.to(gpu)
    ref = pipeline.HeadEnhanceStep(pre, head, criterion=SISDR())(wavs, lengths)
    with torch.inference_mode():
        got = pipeline.HeadEnhanceStep(pre, head, criterion=SISDR())(wavs, lengths)
    assert torch.allclose(got[0], ref[0], rtol=0, atol=2e-5 * ref[0].abs().max().item())
    assert torch.allclose(got[1], ref[1], rtol=2e-5, atol=2e-5 * ref[1].abs().max().item())
    assert abs(got[3].item() - ref[3].item()) < 1e-5 * abs(ref[3].item())
    wsd = pipeline.HeadEnhanceStep(pre, head, criterion=WSD())(wavs, lengths)          # reads the lazy `offset`
    assert torch.isfinite(wsd[3]) and torch.equal(wsd[1], ref[1])
