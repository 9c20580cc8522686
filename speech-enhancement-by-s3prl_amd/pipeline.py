"""The per-batch hot path assembled the way Runner.evaluate() / Runner.train() run it (runner.py:431-471,556-575),
plus seeded random weights at the real sizes (no checkpoints exist offline).  Used by bench.py, the smoke test
and the parity tests; `run_downstream.py`-style drivers compose the same objects themselves."""
import copy

import torch

from . import decode
from .checkpoint import reference_paras
from .heads import SpecHead
from .objective import L1
from .preprocessor import OnlinePreprocessor
from .transformer import TRANSFORMER

# config/pretrain_sample.yaml restated (the only architecture config the reference ships)
PRETRAIN_SAMPLE = {
    'transformer': {'input_dim': 80, 'downsample_rate': 1, 'hidden_size': 768, 'num_hidden_layers': 6,
                    'num_attention_heads': 12, 'intermediate_size': 3072, 'hidden_act': 'gelu',
                    'hidden_dropout_prob': 0.1, 'attention_probs_dropout_prob': 0.1, 'initializer_range': 0.02,
                    'layer_norm_eps': '1e-12', 'share_layer': False},
    'online': {'sample_rate': 16000, 'max_time': 10000, 'target_level': -25, 'noise_proportion': 0.5, 'snrs': [3, 6],
               'win_ms': 25, 'hop_ms': 10, 'n_freq': 201, 'n_mels': 40, 'n_mfcc': 13,
               'input': {'feat_type': 'mel', 'channel': 0, 'log': True, 'delta': 1, 'cmvn': True},
               'target': {'feat_type': 'linear', 'channel': 1, 'log': True, 'delta': 0, 'cmvn': False}},
}
# config/pseudo_noise.yaml preprocessor.baseline
BASELINE_FEAT = {'feat_type': 'mel', 'log': True, 'delta': 2, 'cmvn': False}


def make_config(layers=6, hidden=768, heads=12, intermediate=3072):
    cfg = copy.deepcopy(PRETRAIN_SAMPLE)
    cfg['transformer'].update(num_hidden_layers=layers, hidden_size=hidden, num_attention_heads=heads, intermediate_size=intermediate)
    return cfg


def random_upstream_states(config, inp_dim=80, spec_out=201, seed=0):
    """Seeded state dicts (S3PRL key names) at the sizes of `config`: N(0, initializer_range) matrices and biases,
    LayerNorm weight 1 + 0.1 N(0,1), bias 0.1 N(0,1).  Returns (transformer_state, spechead_state)."""
    t = config['transformer']
    H, I, L, r = t['hidden_size'], t['intermediate_size'], t['num_hidden_layers'], float(t['initializer_range'])
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def lin(d, name, o, i):
        d[name + '.weight'] = torch.randn(o, i, generator=g) * r
        d[name + '.bias'] = torch.randn(o, generator=g) * r

    def ln(d, name):
        d[name + '.weight'] = 1.0 + 0.1 * torch.randn(H, generator=g)
        d[name + '.bias'] = 0.1 * torch.randn(H, generator=g)

    lin(sd, 'input_representations.spec_transform', H, inp_dim)
    ln(sd, 'input_representations.LayerNorm')
    for i in range(L):
        p = f'encoder.layer.{i}.'
        for n in ('query', 'key', 'value'):
            lin(sd, p + 'attention.self.' + n, H, H)
        lin(sd, p + 'attention.output.dense', H, H)
        ln(sd, p + 'attention.output.LayerNorm')
        lin(sd, p + 'intermediate.dense', I, H)
        lin(sd, p + 'output.dense', H, I)
        ln(sd, p + 'output.LayerNorm')
    head = {}
    lin(head, 'dense', H, H)
    ln(head, 'LayerNorm')
    lin(head, 'output', spec_out, H)
    return sd, head


def synthetic_checkpoint(config=None, seed=0, inp_dim=80, spec_out=201):
    """An in-memory checkpoint with the S3PRL layout the reference reads (model.py:98-108,144-153)."""
    config = config or make_config()
    sd, head = random_upstream_states(config, inp_dim, spec_out, seed)
    # Settings.Paras is the argparse.Namespace of the writing run in every file S3PRL / the reference saves (runner.py:136)
    return {'Settings': {'Config': config, 'Paras': reference_paras(downstream='SpecHead')}, 'Transformer': sd, 'SpecHead': head}


def build_upstream(ckpt, device):
    """TRANSFORMER + SpecHead from a checkpoint dict, as run_downstream.get_upstream_model (run_downstream.py:167-185)."""
    options = {'ckpt_file': '', 'load_pretrain': 'False', 'no_grad': 'True', 'dropout': 'default', 'spec_aug': 'False',
               'spec_aug_prev': 'True', 'weighted_sum': 'False', 'select_layer': -1, 'permute_input': 'False'}
    inp_dim = ckpt['Transformer']['input_representations.spec_transform.weight'].shape[1]
    out_dim = ckpt['SpecHead']['output.weight'].shape[0]
    upstream = TRANSFORMER(options, inp_dim, config=ckpt['Settings']['Config'])
    upstream.model.load_state_dict(ckpt['Transformer'])
    setattr(upstream, 'SpecHead', SpecHead(out_dim, ckpt))
    return upstream.to(device).eval()


def build_preprocessor(config, device, channel_inp=0, channel_tar=1, downstream_feat=None, upstream='transformer'):
    """run_downstream.get_preprocessor (run_downstream.py:123-164): the six-feature list.  upstream = 'transformer': the upstream feature is the
    pre-training input (run_downstream.py:132-133); 'baseline' (no upstream model, `dummy_upstream`): it is the baseline feature itself (:134-135),
    i.e. the same request as the downstream feature -- which the preprocessor then computes once."""
    online = config['online']
    down = dict(downstream_feat or BASELINE_FEAT, channel=channel_inp)
    up = dict(online['input'], channel=channel_inp) if upstream == 'transformer' else dict(down)
    P = OnlinePreprocessor
    feat_list = [up, down, P.get_feat_config('linear', channel_inp), P.get_feat_config('phase', channel_inp),
                 P.get_feat_config('linear', channel_tar), P.get_feat_config('phase', channel_tar)]
    pre = OnlinePreprocessor(**online, feat_list=feat_list)
    pre.channel_inp, pre.channel_tar = channel_inp, channel_tar
    pre.encoder_side = upstream == 'transformer'      # no upstream encoder: the first feature needs no bf16 operand copy / valid-frame counts
    return pre.to(device)


class UpstreamEnhanceStep:
    """One evaluate()-style pass (runner.py:556-575) with the upstream + SpecHead as the enhancer
    (the _pseudo_clean path, runner.py:273-277): wavs -> 6 features -> encoder -> spec head -> decode_wav
    (iSTFT with the noisy phase, normalised to the clean wav's level) -> L1 loss.

    streams = 2: the batch is processed as two half batches on two HIP streams.  Every stage is per-utterance (the loss is
    recombined from the halves' (sum, count) pairs), and the big kernels are one-round launches whose workgroups all reach their
    HBM-bound epilogue together -- two half-size launches out of phase keep both the matrix pipes and HBM busier: 4.84 vs
    5.07 ms for the batch of 32 (tools/two_streams.py).  The halves use the row-complete GEMM + LayerNorm kernel although
    each is below the single-stream threshold."""

    def __init__(self, preprocessor, upstream, criterion=None, streams=1):
        self.pre, self.up = preprocessor, upstream
        self.criterion = criterion or L1()
        self.streams = int(streams)
        self._side = None
        if self.streams > 1:
            self.up._engine.fused_ln_min_rows = 8192

    def _one(self, wavs, lengths, max_len, want_sums=False):
        feats_up, feats_down, lin_inp, ph_inp, lin_tar, ph_tar = self.pre(wavs)
        hidden = self.up(feats_up)
        predicted, res = self.up.SpecHead(hidden)
        wav_tar = wavs[:, self.pre.channel_tar, :]
        wav_pred = decode.decode_wav(self.pre, predicted, ph_inp, lengths, wav_tar, max_len=max_len)
        hop = self.pre._win_args['hop_length']
        if isinstance(self.criterion, L1):       # frame counts derived inside the kernel (runner.py:455): no element-wise launches in front of it
            lens_kw = {'wav_lengths': lengths, 'hop': hop}
        else:
            lens_kw = {'stft_lengths': lengths // hop + 1}
        if want_sums:
            sums = []
            self.criterion.reduce_fn = lambda t: (sums.append(t) or t)
            try:
                self.criterion(log_predicted=res['log_predicted'], linear_tar=lin_tar, **lens_kw)
            finally:
                self.criterion.reduce_fn = None
            return wav_pred, sums[0], predicted
        loss, _ = self.criterion(log_predicted=res['log_predicted'], linear_tar=lin_tar, **lens_kw)
        return wav_pred, loss, predicted

    @torch.no_grad()
    def __call__(self, wavs, lengths, max_len=None):
        B = wavs.shape[0]
        if self.streams <= 1 or B < 2 * self.streams:
            return self._one(wavs, lengths, max_len)
        if max_len is None:
            max_len = int(lengths.max().item())
        dev = wavs.device
        if self._side is None:
            self._side = [torch.cuda.Stream(device=dev) for _ in range(self.streams)]
        cur = torch.cuda.current_stream(dev)
        bounds = [B * i // self.streams for i in range(self.streams + 1)]
        parts = []
        for i, st in enumerate(self._side):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                parts.append(self._one(wavs[bounds[i]:bounds[i + 1]], lengths[bounds[i]:bounds[i + 1]], max_len, want_sums=True))
        for st in self._side:
            cur.wait_stream(st)
        for p in parts:                      # produced on the side streams, consumed on the caller's stream from here on
            for t in p:
                t.record_stream(cur)
        wav_pred = torch.cat([p[0] for p in parts], dim=0)
        predicted = torch.cat([p[2] for p in parts], dim=0)
        sums = torch.stack([p[1] for p in parts]).sum(dim=0)          # (sum |.|, count): the global masked mean of objective.py:113-116
        return wav_pred, (sums[0] / sums[1]).float(), predicted


class HeadEnhanceStep:
    """evaluate()-style pass for a feature-input head (LinearResidual / Linear), config 1 / 4 (runner.py:556-575):
    wavs (B, C >= 2, T) -> features -> mask head -> mask (.) noisy power -> decode_wav [-> criterion].
    Returns (wav_pred, predicted, linear_tar, loss); loss is None without a criterion."""

    fuse_criterion = True        # False: the criterion as its own launches behind the head (A/B)

    def __init__(self, preprocessor, head, criterion=None):
        self.pre, self.head, self.criterion = preprocessor, head, criterion
        if getattr(head, 'cmvn', False) and hasattr(head, 'eps'):
            # the head normalises its input over time (model.py:29-31): the feature launch has every row in LDS and hands the statistics over
            self.pre.head_stats_eps = float(head.eps)

    @torch.no_grad()
    def __call__(self, wavs, lengths, max_len=None):
        from .objective import SISDR
        feats_up, feats_down, lin_inp, ph_inp, lin_tar, ph_tar = self.pre(wavs)
        hop = self.pre._win_args['hop_length']
        if (self.fuse_criterion and isinstance(self.criterion, SISDR) and self.criterion.reduce_fn is None and hasattr(self.head, 'enhance_scored') and
                not lin_tar.is_inference()):
            # the criterion's sums come out of the head's own launch (the finished loss rides on `predicted`; objective.SISDR.forward below takes it)
            predicted, res = self.head.enhance_scored(feats_down, lin_inp, lin_tar, lengths, hop, self.criterion.eps)
        else:
            predicted, res = self.head(features=feats_down, linears=lin_inp)
        wav_tar = wavs[:, self.pre.channel_tar, :]
        wav_pred = decode.decode_wav(self.pre, predicted, ph_inp, lengths, wav_tar, max_len=max_len)
        loss = None
        if self.criterion is not None:
            if isinstance(self.criterion, (L1, SISDR)):      # frame counts derived inside the kernel (runner.py:455): no element-wise launches in front
                lens_kw = {'wav_lengths': lengths, 'hop': hop}
            else:
                lens_kw = {'stft_lengths': lengths // hop + 1}
            loss, _ = self.criterion(predicted=predicted, linear_inp=lin_inp, linear_tar=lin_tar, **lens_kw, **res)
        return wav_pred, predicted, lin_tar, loss


class MockingjayFinetuneStep:
    """One training step of config 3 (runner.py:431-471 with `--downstream Mockingjay`): wavs -> features -> Mockingjay
    (encoder + spec head, training paths) -> global-mean masked log-L1 -> backward on the HIP kernels -> ONE flat-buffer
    gradient all-reduce across ranks (RCCL) -> clip_grad_norm_(1.0) / NaN skip -> BertAdam."""

    def __init__(self, preprocessor, model, optimizer, grad_clip=1.0):
        from .dist import DataParallelTrainStep
        self.pre, self.model = preprocessor, model
        self.criterion = L1()
        self.dp = DataParallelTrainStep(model, self.criterion, optimizer, grad_clip=grad_clip)

    def __call__(self, wavs, lengths):
        with torch.no_grad():
            feats_up, feats_down, lin_inp, ph_inp, lin_tar, ph_tar = self.pre(wavs)
        predicted, res = self.model(features=feats_up, linears=lin_inp)
        stft_lengths = lengths // self.pre._win_args['hop_length'] + 1
        loss, _ = self.criterion(log_predicted=res['log_predicted'], linear_tar=lin_tar, stft_lengths=stft_lengths)
        grad_norm, skipped = self.dp.step(loss)
        return loss.detach(), grad_norm, skipped


def build_mockingjay(ckpt, device, tmp_dir=None):
    """Mockingjay(dckpt) from an in-memory checkpoint dict (the class loads from a path, model.py:148)."""
    import os
    import tempfile
    from .heads import Mockingjay
    d = tmp_dir or tempfile.mkdtemp(prefix='se_amd_ckpt_')
    path = os.path.join(d, 'states.ckpt')
    torch.save(ckpt, path)
    try:
        model = Mockingjay(path)
    finally:
        os.remove(path)
        if tmp_dir is None:
            os.rmdir(d)
    return model.to(device).train()


class GraphedStep:
    """Replays a fixed-shape step as ONE hipGraph launch (torch.cuda.CUDAGraph == hipGraph on ROCm): the evaluate()-style pass is
    ~100 kernel launches, which at serving batch sizes (1-8 utterances) is launch-bound, not GPU-bound.  Every kernel of the
    library launches on torch's current stream and allocates nothing itself, so the whole pass is capturable; inputs are
    copied into static buffers, outputs are the captured tensors (valid until the next replay).

        step = GraphedStep(pipeline.UpstreamEnhanceStep(pre, upstream), wavs, lengths, max_len)
        wav_pred, loss, predicted = step(wavs, lengths)
    """

    def __init__(self, fn, wavs, lengths, *static_args, warmup=3):
        self.fn, self.static_args = fn, static_args
        self.wavs = wavs.clone()
        self.lengths = lengths.clone()
        side = torch.cuda.Stream(device=wavs.device)
        side.wait_stream(torch.cuda.current_stream(wavs.device))
        with torch.cuda.stream(side):          # first calls create plans / set kernel attributes / warm the allocator: not capturable
            for _ in range(warmup):
                fn(self.wavs, self.lengths, *static_args)
        torch.cuda.current_stream(wavs.device).wait_stream(side)
        torch.cuda.synchronize(wavs.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = fn(self.wavs, self.lengths, *static_args)

    def __call__(self, wavs, lengths):
        self.wavs.copy_(wavs, non_blocking=True)
        self.lengths.copy_(lengths, non_blocking=True)
        self.graph.replay()
        return self.out


class HeadFinetuneStep:
    """One training step of a feature-input downstream head (config 5: `--downstream LSTM --from_rawfeature`, runner.py:431-471):
    wavs -> downstream features (mel / log / delta-2, 120 dims) -> head -> criterion -> backward -> flat-buffer gradient
    all-reduce -> clip / NaN skip -> BertAdam."""

    def __init__(self, preprocessor, head, optimizer, criterion=None, grad_clip=1.0):
        from .dist import DataParallelTrainStep
        self.pre, self.head = preprocessor, head
        self.criterion = criterion or L1()
        self.dp = DataParallelTrainStep(head, self.criterion, optimizer, grad_clip=grad_clip)

    def __call__(self, wavs, lengths):
        with torch.no_grad():
            feats_up, feats_down, lin_inp, ph_inp, lin_tar, ph_tar = self.pre(wavs)
        predicted, res = self.head(features=feats_down, linears=lin_inp)
        stft_lengths = lengths // self.pre._win_args['hop_length'] + 1
        loss, _ = self.criterion(predicted=predicted, linear_inp=lin_inp, linear_tar=lin_tar, stft_lengths=stft_lengths, **res)
        grad_norm, skipped = self.dp.step(loss)
        return loss.detach(), grad_norm, skipped
