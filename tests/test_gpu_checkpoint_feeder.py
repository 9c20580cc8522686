"""GPU: (a) the reference's checkpoint wire format with the product's modules -- Runner.save_model's dict {'Downstream', 'Optimizer',
'Global_step', 'Settings'} (runner.py:129-151), Runner.load_model (runner.py:122-126), the resume route of run_downstream.py:94-106 and the
`SmallModel` key-strip of run_downstream.py:213-214 -- must round-trip: save after some steps, reload into FRESH objects, and the next
training step of the resumed run equals the uninterrupted run bit for bit.  (b) the pinned double-buffered host -> HBM feeder in front of
the path (runner.py:431-432) delivers exactly the host batches, in order, on the caller's stream."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _train_steps(model, opt, pre, batches, gpu, criterion):
    from speech_enhancement_by_s3prl_amd import decode
    losses = []
    for lengths, wavs in batches:
        wavs, lengths = wavs.to(gpu), lengths.to(gpu)
        with torch.no_grad():
            feats_up, feats_down, linear_inp, phase_inp, linear_tar, phase_tar = pre(wavs)
        predicted, model_results = model(features=feats_down, linears=linear_inp)
        stft_lengths = lengths // pre._win_args['hop_length'] + 1
        loss, _ = criterion(predicted=predicted, linear_inp=linear_inp, linear_tar=linear_tar, stft_lengths=stft_lengths,
                            stft_length_masks=decode.get_length_masks(stft_lengths), **model_results)
        loss.backward()
        grad_norm = torch.nn.utils.clip_grad_norm_(list(model.parameters()), 1.0)       # runner.py:463-464
        assert torch.isfinite(grad_norm)
        opt.step()
        opt.zero_grad()
        losses.append(loss.item())
    return losses


@pytest.mark.parametrize('kind', ['LinearResidual', 'LSTM'])
def test_checkpoint_roundtrip_resumes_identically(gpu, tmp_path, kind):
    from speech_enhancement_by_s3prl_amd import pipeline, synth
    from speech_enhancement_by_s3prl_amd.checkpoint import load_checkpoint, reference_paras
    from speech_enhancement_by_s3prl_amd.heads import LinearResidual
    from speech_enhancement_by_s3prl_amd.lstm import LSTM
    from speech_enhancement_by_s3prl_amd.objective import L1, SISDR
    from speech_enhancement_by_s3prl_amd.solver import get_optimizer
    cfg = pipeline.make_config(layers=1, hidden=256, heads=4, intermediate=512)
    pre = pipeline.build_preprocessor(cfg, gpu)

    def make():
        torch.manual_seed(21)
        if kind == 'LinearResidual':
            m = LinearResidual(input_size=120, output_size=201, cmvn=True).to(gpu)
            crit = SISDR().to(gpu)                  # LinearResidual returns `offset`, not log_predicted (objective.py:81-100)
        else:
            m = LSTM(input_size=120, output_size=201, hidden_size=256, num_layers=1, bidirectional=True).to(gpu)
            crit = L1().to(gpu)
        o = get_optimizer(params=list(m.named_parameters()), lr=2e-4, warmup_proportion=0.07, training_steps=100)      # runner.py:110-113
        return m.train(), o, crit

    batches = [synth.synth_batch(2, 8000, first=10 * i, ragged=True) for i in range(5)]
    model, opt, crit = make()
    _train_steps(model, opt, pre, batches[:3], gpu, crit)
    global_step = 4
    # ---- Runner.save_model (runner.py:129-151)
    all_states = {'Downstream': model.state_dict(), 'Optimizer': opt.state_dict(), 'Global_step': global_step,
                  'Settings': {'Config': {'runner': {'learning_rate': '2e-4'}, 'model': {kind: {}}}, 'Paras': reference_paras(downstream=kind)}}
    path = str(tmp_path / f'states-{global_step}.ckpt')
    torch.save(all_states, path)
    # the uninterrupted run goes on
    cont = _train_steps(model, opt, pre, batches[3:], gpu, crit)
    # ---- resume (run_downstream.py:94-106 picks the newest states-*.ckpt; Runner.load_model, runner.py:122-126): FRESH objects
    ckpt = torch.load(path, map_location='cpu')     # the reference's own call (runner.py:123): works because the product registers the Namespace class
    assert ckpt['Settings']['Paras'].downstream == kind
    ckpt = load_checkpoint(path)                    # the product's loader: weights-only + argparse.Namespace allow-listed, nothing is executed
    assert ckpt['Settings']['Paras'].downstream == kind       # run_downstream.py:206
    assert sorted(ckpt) == ['Downstream', 'Global_step', 'Optimizer', 'Settings'] and ckpt['Global_step'] == global_step
    model2, opt2, crit2 = make()
    with torch.no_grad():                           # make sure the reload, not the seed, provides the parameters
        for p in model2.parameters():
            p.add_(1.0)
    model2.load_state_dict(ckpt['Downstream'])
    opt2.load_state_dict(ckpt['Optimizer'])
    st = opt2.state_dict()['state']
    assert len(st) == len(list(model2.parameters())) and all(v['step'] == 3 for v in st.values())
    resumed = _train_steps(model2, opt2, pre, batches[3:], gpu, crit2)
    assert resumed == cont                          # the same losses, bit for bit
    for (n, a), (_, b) in zip(model.named_parameters(), model2.named_parameters()):
        assert torch.equal(a, b), n
    # ---- the SmallModel key-strip of run_downstream.get_downstream_model (run_downstream.py:213-214)
    dckpt = {'SmallModel': {'model.' + k: v for k, v in model.state_dict().items()}}
    state_dict = {'.'.join(key.split('.')[1:]): value for key, value in dckpt['SmallModel'].items()}
    model3, _, _ = make()
    model3.load_state_dict(state_dict)
    for (n, a), (_, b) in zip(model.named_parameters(), model3.named_parameters()):
        assert torch.equal(a, b), n


def test_host_batch_feeder_delivers_batches_in_order(gpu):
    from speech_enhancement_by_s3prl_amd import pipeline, synth
    from speech_enhancement_by_s3prl_amd.feeder import HostBatchFeeder
    cfg = pipeline.make_config(layers=1, hidden=256, heads=4, intermediate=512)
    pre = pipeline.build_preprocessor(cfg, gpu)
    batches = [synth.synth_batch(3, 16000, first=7 * i) for i in range(5)]
    seen = 0
    for (dl, dw), (hl, hw) in zip(HostBatchFeeder(batches, gpu), batches):
        assert dw.is_cuda and dl.is_cuda
        lin = pre(dw, [pre.get_feat_config('linear', 0)])[0]         # consume on the caller's stream while the next copy is in flight
        ref = pre(hw.to(gpu), [pre.get_feat_config('linear', 0)])[0]
        assert torch.equal(dw.cpu(), hw) and torch.equal(dl.cpu(), hl) and torch.equal(lin, ref)
        seen += 1
    assert seen == len(batches)
    # only the channels the pass reads cross PCIe: channels 0, 1 equal the host batch, the shape stays (B, C, T)
    for (dl, dw), (hl, hw) in zip(HostBatchFeeder(batches, gpu, channels=2), batches):
        assert dw.shape == hw.shape
        assert torch.equal(dw[:, :2].cpu(), hw[:, :2]) and torch.equal(dl.cpu(), hl)
        lin = pre(dw, [pre.get_feat_config('linear', 1)])[0]
        assert torch.equal(lin, pre(hw.to(gpu), [pre.get_feat_config('linear', 1)])[0])
    # fewer batches than slots, and an empty source
    assert sum(1 for _ in HostBatchFeeder(batches[:1], gpu)) == 1
    assert sum(1 for _ in HostBatchFeeder([], gpu)) == 0
