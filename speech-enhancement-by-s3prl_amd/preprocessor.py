"""``OnlinePreprocessor`` -- drop-in for S3PRL ``utility.preprocessor.OnlinePreprocessor`` backed by the
gfx950 kernels of libse_amd.so (rows A1-A6 of SURVEY.md section 8a).

Surface kept (cited uses in the reference):
  ctor(**config['online'], feat_list=None) tolerating extra keys   run_downstream.py:159, model.py:145
  get_feat_config(feat_type, channel, log, delta, cmvn)            run_downstream.py:153-156, runner.py:50
  __call__(wavs=None, feat_list=None) -> [ (B, T', D) ... ]        runner.py:433,558; sampler.py:60; model.py:146
  istft(linears, phases) -> (B, T)                                 runner.py:267
  _win_args['hop_length'], _sample_rate, _window                   runner.py:48,455; sampler.py:226
  _stft(wav2d, window=) / _magphase(complx)                        sampler.py:226-228
  .to() / .cpu() / deepcopy / pickle (spawned sampler child)       runner.py:65,70,232

The compute path is HIP only.  The reference calls the module while it (and its input) still live on the host --
the zero-argument dimension probe (run_downstream.py:163,182-183; model.py:145-146), the `.cpu()` copy used for
logging (runner.py:51,65) and the sampler child's default device (sampler.py:145-151).  Those calls are served by
staging the host tensor on the current HIP device, running the SAME kernels there and handing the results back on the
caller's device: still the HIP path, not a CPU implementation.  Without a gfx950 device every call raises (SEError).
Construction and pickling work without a device; plans are created lazily on first use so a spawned child can pick
its GPU first (sampler.py:145-153).
"""
import torch
import torch.nn as nn

from . import _lib

N_SAMPLED_PSEUDO_WAV = 2


class LazyTensor(torch.Tensor):
    """An fp32 result of the boundary that its usual consumers never read, produced ON DEMAND: shape / dtype / device are the real tensor's, and
    any use -- arithmetic, indexing, .cpu(), printing, torch.save -- goes through __torch_dispatch__, which first materialises the value with the
    kernel that makes it and then runs the requested op on it.  Merely carrying the object around (returning it, passing it as a keyword
    argument that the callee ignores) costs nothing.  Users: the `phase` features (LazyPhase below) and LinearResidual's `offset` result."""

    @staticmethod
    def __new__(cls, shape, device, materialize, *args, **kwargs):
        r = torch.Tensor._make_wrapper_subclass(cls, tuple(shape), dtype=torch.float32, device=device, requires_grad=False)
        r._materialize_fn, r._value = materialize, None
        return r

    def materialize(self):
        if self._value is None:
            self._value = self._materialize_fn()
            self._materialize_fn = None
        return self._value

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        from torch.utils._pytree import tree_map

        def unwrap(t):
            return t.materialize() if isinstance(t, LazyTensor) else t
        return func(*tree_map(unwrap, args), **tree_map(unwrap, kwargs or {}))

    def __reduce_ex__(self, protocol):          # pickling / torch.save / deepcopy: as the plain tensor
        return self.materialize().__reduce_ex__(protocol)


def Fp_(raw, time_major):
    """frame count of a raw feature plane: (B, F, D) time-major or (B, D, F) feature-major"""
    return raw.shape[1] if time_major else raw.shape[2]


class LazyPhase(LazyTensor):
    """The `phase` feature of the boundary (run_downstream.py:150-157: `get_feat_config('phase', channel)`), produced ON DEMAND.

    Every consumer of the noisy phase inside the reference's pipelines is `preprocessor.istft(linears, phases)` (runner.py:267), and nothing
    ever reads the clean channel's phase (runner.py:433,558 unpack it and drop it).  So the hot path neither computes atan2 in the STFT nor
    sin / cos in the iSTFT: the STFT kernel writes the phase as one word per bin from which (cos, sin) follow rationally (`_tphase`, int32
    (..., F, K): fp32 t = tan(half angle) with the sign of cos in bit 0, stft2.hip), `istft` recognises this object and runs
    se_istft_tphase_f32 on it.  Any OTHER use -- arithmetic, indexing, .cpu(), printing, torch.save -- goes through __torch_dispatch__,
    which first materialises the real (…, F, K) fp32 phase with the atan2 kernel (se_stft_f32 on the retained waveform: bit-identical to
    what the eager path returns) and then runs the requested op on it.  Shape / dtype / device are the real tensor's."""

    @staticmethod
    def __new__(cls, shape, device, materialize, tphase=None):
        r = LazyTensor.__new__(cls, shape, device, materialize)
        r._tphase = tphase
        return r


class OnlinePreprocessor(nn.Module):
    _FEAT_TYPES = ('complx', 'linear', 'phase', 'mel', 'mfcc')

    def __init__(self, sample_rate=16000, win_ms=25, hop_ms=10, n_freq=201, n_mels=40, n_mfcc=13, feat_list=None,
                 eps=1e-10, **kwargs):
        super().__init__()
        self._sample_rate = sample_rate
        self._win_ms, self._hop_ms = win_ms, hop_ms
        self._n_freq, self._n_mels, self._n_mfcc = n_freq, n_mels, n_mfcc
        win = round(win_ms * sample_rate / 1000)
        hop = round(hop_ms * sample_rate / 1000)
        n_fft = (n_freq - 1) * 2
        self._win_args = {'n_fft': n_fft, 'hop_length': hop, 'win_length': win}
        self.register_buffer('_window', torch.hann_window(win))
        self.feat_list = feat_list
        g = torch.Generator().manual_seed(0)
        self.register_buffer('_pseudo_wavs', torch.randn(N_SAMPLED_PSEUDO_WAV, sample_rate, generator=g))
        self.eps = eps
        self._plans = {}          # device index -> se_plan* (not pickled / deep-copied)

    # ---- plan management -------------------------------------------------------------------------
    def __getstate__(self):
        state = self.__dict__.copy()
        state['_plans'] = {}
        state.pop('_tables', None)
        return state

    def __deepcopy__(self, memo):
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = {} if k in ('_plans', '_tables') else copy.deepcopy(v, memo)
        return new

    def __del__(self):
        try:
            lib = _lib._lib
            if lib is not None:
                for p in self._plans.values():
                    lib.se_plan_destroy(p)
        except Exception:
            pass

    def _plan(self, device, n_mels=None):
        idx = device.index if device.index is not None else torch.cuda.current_device()
        n_mels = self._n_mels if n_mels is None else n_mels
        key = idx if n_mels == self._n_mels else (idx, n_mels)       # the MFCC branch has its own 128-filter plan
        plan = self._plans.get(key)
        if plan is None:
            lib = _lib.load()
            geom = _lib.Geometry(self._sample_rate, self._win_args['win_length'], self._win_args['hop_length'],
                                 self._n_freq, n_mels)
            out = _lib.c_void_p()
            with torch.cuda.device(idx):
                _lib.check(lib.se_plan_create(geom, out), 'se_plan_create')
            plan = self._plans[key] = out.value
        return plan

    # ---- row A5: MFCC (torchaudio-0.6 transforms.MFCC(log_mels=True, melkwargs=win_args) as S3PRL builds it; parity unpinned) ----
    _MFCC_N_MELS = 128          # torchaudio's MelSpectrogram default: the MFCC transform has its OWN mel bank
    _MFCC_LOG_OFFSET = 1e-6

    def _dct(self, dev):
        """(n_mfcc, 128) fp32 = the transposed orthonormal DCT-II matrix of torchaudio's create_dct, built in float64"""
        import math
        key = ('dct', dev.index)
        d = self.__dict__.setdefault('_tables', {}).get(key)
        if d is None:
            n = torch.arange(self._MFCC_N_MELS, dtype=torch.float64)
            k = torch.arange(self._n_mfcc, dtype=torch.float64).unsqueeze(1)
            d = torch.cos(math.pi / self._MFCC_N_MELS * (n + 0.5) * k)
            d[0] *= 1.0 / math.sqrt(2.0)
            d = (d * math.sqrt(2.0 / self._MFCC_N_MELS)).float().contiguous().to(dev)
            self.__dict__['_tables'][key] = d
        return d

    def _mfcc_channel(self, wavs, channel):
        """wavs (B, C, T) -> raw MFCC (B, F, n_mfcc) time-major: STFT + 128-filter sparse mel in the STFT kernel (no power / phase
        planes are written), log(mel + 1e-6) + transpose in the feature kernel, DCT-II as an exact-fp32 GEMM (se_gemm_f32)."""
        lib = _lib.load()
        B, C, T = wavs.shape
        F = T // self._win_args['hop_length'] + 1
        dev = wavs.device
        NM = self._MFCC_N_MELS
        mel = torch.empty(B, NM, F, device=dev, dtype=torch.float32)
        _lib.check(lib.se_stft_f32(self._plan(dev, NM), _lib.ptr(wavs), B, C, T, channel, None, None, None, _lib.ptr(mel), _lib.stream()), 'se_stft_f32')
        logmel = torch.empty(B, F, NM, device=dev, dtype=torch.float32)
        nbytes = lib.se_features_workspace_bytes(B, NM, F, 0)
        ws = torch.empty(max(nbytes, 1), device=dev, dtype=torch.uint8)
        _lib.check(lib.se_features_f32(_lib.ptr(mel), 0, B, NM, F, 1, 0, 0, float(self._MFCC_LOG_OFFSET), _lib.ptr(logmel), _lib.ptr(ws), nbytes,
                                       _lib.stream()), 'se_features_f32')
        out = torch.empty(B, F, self._n_mfcc, device=dev, dtype=torch.float32)
        _lib.check(lib.se_gemm_f32(_lib.ptr(logmel), NM, _lib.ptr(self._dct(dev)), NM, 0, None, None, 0, B * F, self._n_mfcc, NM, 0, 1.0,
                                   _lib.ptr(out), self._n_mfcc, 1, 1, 0, 0, 0, 0, 0, 0, _lib.stream()), 'se_gemm_f32')
        return out

    # ---- S3PRL surface ---------------------------------------------------------------------------
    @classmethod
    def get_feat_config(cls, feat_type, channel=0, log=False, delta=0, cmvn=False):
        assert feat_type in cls._FEAT_TYPES
        assert type(channel) is int and type(log) is bool and type(delta) is int and type(cmvn) is bool
        return {'feat_type': feat_type, 'channel': channel, 'log': log, 'delta': delta, 'cmvn': cmvn}

    def _check_list(self, feat_list):
        feat_list = self.feat_list if feat_list is None else feat_list
        assert type(feat_list) is list
        return feat_list

    @staticmethod
    def _stage(t):
        """(tensor on a HIP device, device the caller expects results on).  A host tensor is copied to the current HIP
        device (the reference's CPU-resident call sites, see the module docstring); there is no CPU implementation, so
        without a device this raises."""
        if t.is_cuda:
            return t, t.device
        if not torch.cuda.is_available():
            raise _lib.SEError('OnlinePreprocessor runs on MI355X only: no HIP device is visible and the path has no CPU '
                               'fallback (host inputs are staged on the current HIP device, they are never computed on the host)')
        return t.to(torch.device('cuda', torch.cuda.current_device())), t.device

    def _stft_channel(self, wavs, channel, need):
        """One se_stft_f32 launch for `channel`; `need` is a set of {'linear','phase','complx','mel'}."""
        lib = _lib.load()
        B, C, T = wavs.shape
        F = T // self._win_args['hop_length'] + 1
        K = self._n_freq
        dev = wavs.device
        out = {}
        if 'linear' in need:
            out['linear'] = torch.empty(B, F, K, device=dev, dtype=torch.float32)
        if 'phase' in need:
            out['phase'] = torch.empty(B, F, K, device=dev, dtype=torch.float32)
        if 'complx' in need:
            out['complx'] = torch.empty(B, F, 2 * K, device=dev, dtype=torch.float32)
        if 'mel' in need:
            out['mel'] = torch.empty(B, self._n_mels, F, device=dev, dtype=torch.float32)
        _lib.check(lib.se_stft_f32(self._plan(dev), _lib.ptr(wavs), B, C, T, channel, _lib.ptr(out.get('linear')),
                                   _lib.ptr(out.get('phase')), _lib.ptr(out.get('complx')), _lib.ptr(out.get('mel')),
                                   _lib.stream()), 'se_stft_f32')
        return out

    def _alloc_planes(self, B, F, need, dev):
        K = self._n_freq
        out = {}
        if 'linear' in need:
            out['linear'] = torch.empty(B, F, K, device=dev, dtype=torch.float32)
        if 'phase' in need:
            out['phase'] = torch.empty(B, F, K, device=dev, dtype=torch.float32)
        if 'complx' in need:
            out['complx'] = torch.empty(B, F, 2 * K, device=dev, dtype=torch.float32)
        if 'mel' in need:
            out['mel'] = torch.empty(B, self._n_mels, F, device=dev, dtype=torch.float32)
        return out

    def _stft_two_channels(self, wavs, need):
        """One se_stft2_f32 launch for the two channels of `need` = {channel: kinds}."""
        lib = _lib.load()
        B, C, T = wavs.shape
        F = T // self._win_args['hop_length'] + 1
        (ca, ka), (cb, kb) = sorted(need.items())
        oa, ob = self._alloc_planes(B, F, ka, wavs.device), self._alloc_planes(B, F, kb, wavs.device)
        g = lambda o, k: _lib.ptr(o.get(k))       # noqa: E731
        _lib.check(lib.se_stft2_f32(self._plan(wavs.device), _lib.ptr(wavs), B, C, T, ca, g(oa, 'linear'), g(oa, 'phase'), g(oa, 'complx'), g(oa, 'mel'),
                                    cb, g(ob, 'linear'), g(ob, 'phase'), g(ob, 'complx'), g(ob, 'mel'), _lib.stream()), 'se_stft2_f32')
        return {ca: oa, cb: ob}

    # ---- the encoded-phase form of the STFT (stft2.hip): what the hot path runs ------------------------------------------------------
    lazy_phase = True          # False: always return eager atan2 phase planes (the round-2 behaviour)

    def _tphase_path_ok(self, need, wavs3, home):
        """phase planes can be LazyPhase objects when the caller lives on the device (host-resident callers get plain tensors), at most two
        channels are transformed and nobody asked for `complx`."""
        if not self.lazy_phase or home != wavs3.device or not need or len(need) > 2:
            return False
        if not any('phase' in k for k in need.values()):
            return False
        return all(k <= {'linear', 'phase', 'mel'} for k in need.values())

    def _stft_tphase(self, wavs3, need, lead):
        """One se_stft_tphase_f32 launch for the (one or two) channels of `need`.  The encoded phase plane is kept for the INPUT channel only
        (`channel_inp`, set by run_downstream.py:161 / pipeline.build_preprocessor; default: the lowest channel that asks for a phase):
        it is what istft() consumes; every other channel's phase is never read by the reference and costs nothing until somebody does."""
        lib = _lib.load()
        B, C, T = wavs3.shape
        F = T // self._win_args['hop_length'] + 1
        K = self._n_freq
        dev = wavs3.device
        with_phase = sorted(ch for ch, k in need.items() if 'phase' in k)
        ch_inp = getattr(self, 'channel_inp', None)
        if ch_inp not in with_phase:
            ch_inp = with_phase[0]
        chans = sorted(need)
        outs = {}
        for ch in chans:
            o = {}
            if 'linear' in need[ch]:
                o['linear'] = torch.empty(B, F, K, device=dev, dtype=torch.float32)
            if 'mel' in need[ch]:
                o['mel'] = torch.empty(B, self._n_mels, F, device=dev, dtype=torch.float32)
            if ch == ch_inp and 'phase' in need[ch]:
                o['_tphase'] = torch.empty(B, F, K, device=dev, dtype=torch.int32)
            outs[ch] = o
        g = lambda o, k: _lib.ptr(o.get(k))       # noqa: E731
        a = chans[0]
        b = chans[1] if len(chans) > 1 else -1
        ob = outs[b] if b >= 0 else {}
        _lib.check(lib.se_stft_tphase_f32(self._plan(dev), _lib.ptr(wavs3), B, C, T, a, g(outs[a], 'linear'), g(outs[a], '_tphase'), g(outs[a], 'mel'),
                                          b, g(ob, 'linear'), g(ob, '_tphase'), g(ob, 'mel'), _lib.stream()), 'se_stft_tphase_f32')
        # The true atan2 phase is recomputed FROM THE WAVEFORMS on first use, and `wavs3` is the caller's own tensor whenever that was already
        # contiguous fp32 on the device: the lazy phase is tied to that buffer's contents.  A caller that refills the buffer in place (a static graph
        # input, wavs.copy_(next_batch)) before reading the phase would get the phase of ANOTHER batch next to this call's `linear` planes: the
        # version counter recorded here turns that into an error; `lazy_phase = False` restores the eager, self-contained result.
        # Tensors created under torch.inference_mode() track no version counter (reading `_version` raises): the lazy phase is then tied to a private
        # clone of the batch instead -- one device copy, only on that path.
        if wavs3.is_inference():
            wavs3 = wavs3.clone()
            ver0 = None
        else:
            ver0 = wavs3._version
        for ch in chans:
            if 'phase' in need[ch]:
                def materialize(ch=ch):
                    if ver0 is not None and wavs3._version != ver0:
                        raise RuntimeError('LazyPhase: the waveform batch this phase belongs to was modified in place after preprocessor(wavs) returned '
                                           '(version %d -> %d); read the phase before re-using the input buffer, or set preprocessor.lazy_phase = False'
                                           % (ver0, wavs3._version))
                    return self._stft_channel(wavs3, ch, {'phase'})['phase'].reshape(*lead, F, K)
                ph = outs[ch].get('_tphase')
                outs[ch]['phase'] = LazyPhase((*lead, F, K), dev, materialize, None if ph is None else ph.reshape(*lead, F, K))
        return outs

    ENCODER_IN_PAD = 128      # csrc/encoder_impl.h kInPad: the encoder's input projection takes bf16 rows zero-padded to this many columns

    def _select(self, raw, raw_time_major, log, delta, cmvn, encoder_side=False):
        """se_features_f32: raw (B, D, F) feature-major or (B, F, D) time-major -> (B, F, D*(1+delta)).
        encoder_side: the same launches also write what the upstream encoder needs from these features -- the rows as zero-padded bf16 (the
        operand of its input projection) and S3PRL's valid-frame counts -- and the pair rides on the returned tensor as `_se_side` (consumed by
        transformer._Engine.encode when it is handed this very tensor, unmodified: run_downstream.py's feats_for_upstream)."""
        lib = _lib.load()
        if raw_time_major:
            B, F, D = raw.shape
        else:
            B, D, F = raw.shape
        Dout = D * (1 + delta)
        out = torch.empty(B, F, Dout, device=raw.device, dtype=torch.float32)
        side = encoder_side and Dout <= self.ENCODER_IN_PAD
        xin = torch.empty(B * F, self.ENCODER_IN_PAD, device=raw.device, dtype=torch.bfloat16) if side else None
        valid = torch.empty(B, device=raw.device, dtype=torch.int32) if side else None
        if not self.one_pass_features:      # A/B: the round-1 two-launch form through a feature-major intermediate
            nbytes = lib.se_features_workspace_bytes(B, D, F, delta)
            ws = torch.empty(nbytes, device=raw.device, dtype=torch.uint8)
            _lib.check(lib.se_features2_f32(_lib.ptr(raw), int(raw_time_major), B, D, F, int(bool(log)), int(delta),
                                            int(bool(cmvn)), float(self.eps), _lib.ptr(out), _lib.ptr(ws), nbytes,
                                            _lib.ptr(xin), self.ENCODER_IN_PAD if side else 0, _lib.ptr(valid), _lib.stream()), 'se_features2_f32')
        else:
            # LinearResidual's own CMVN (model.py:29-31) needs the column statistics of exactly these rows: the owner of the head says so
            # (head_stats_eps, pipeline.HeadEnhanceStep) and they ride on the returned tensor as `_se_colstats`
            hs_eps = None if (cmvn or encoder_side) else self.head_stats_eps
            nbytes = (lib.se_features3_workspace_bytes(B, D, delta) if cmvn else
                      lib.se_features3_colstats_workspace_bytes(B, D, F, delta) if hs_eps is not None else 0)
            ws = torch.empty(nbytes, device=raw.device, dtype=torch.uint8) if nbytes else None
            colstats = torch.empty(B, Dout, 2, device=raw.device, dtype=torch.float32) if hs_eps is not None else None
            _lib.check(lib.se_features3_f32(_lib.ptr(raw), int(raw_time_major), B, D, F, int(bool(log)), int(delta), int(bool(cmvn)), float(self.eps),
                                            _lib.ptr(out), _lib.ptr(ws), nbytes, _lib.ptr(xin), self.ENCODER_IN_PAD if side else 0, _lib.ptr(valid),
                                            _lib.ptr(colstats), float(hs_eps) if hs_eps is not None else 0.0, _lib.stream()), 'se_features3_f32')
            if colstats is not None:
                out._se_colstats = (colstats, float(hs_eps))
        if side:
            out._se_side = (xin, valid)
        return out

    one_pass_features = True       # False: se_features2_f32 (two launches, feature-major intermediate)
    head_stats_eps = None          # eps of a LinearResidual(cmvn=True) fed by these features: its column statistics come out of the feature launch

    def forward(self, wavs=None, feat_list=None):
        # wavs: (batch_size, channel, max_len); returns [(batch, max_feat_len, feat_dim), ...]
        feat_list = self._check_list(feat_list)
        if wavs is None:
            max_channel_id = max(int(a['channel']) if 'channel' in a else 0 for a in feat_list)
            wavs = self._pseudo_wavs[0].view(1, 1, -1).repeat(1, max_channel_id + 1, 1)
        assert wavs.dim() >= 3
        wavs, home = self._stage(wavs)
        wavs = wavs.contiguous().float()
        lead = wavs.shape[:-2]
        wavs3 = wavs.reshape(-1, wavs.shape[-2], wavs.shape[-1])

        # which raw planes does each channel need?  (the reference transforms every channel every step and
        # also computes an MFCC it discards -- row A5; here only requested channels / planes are produced)
        need, mfcc_channels = {}, set()
        for a in feat_list:
            ft = a['feat_type']
            if ft == 'mfcc':      # its own transform (128-filter mel bank): _mfcc_channel
                mfcc_channels.add(int(a.get('channel', 0)))
                continue
            need.setdefault(int(a.get('channel', 0)), set()).add(ft)
        lazy = self._tphase_path_ok(need, wavs3, home)
        if lazy:
            planes = self._stft_tphase(wavs3, need, lead)
        elif len(need) == 2:       # the reference's standard list (noisy + clean channel): both transforms in ONE launch
            planes = self._stft_two_channels(wavs3, need)
        else:
            planes = {ch: self._stft_channel(wavs3, ch, kinds) for ch, kinds in need.items()}
        for ch in mfcc_channels:
            planes.setdefault(ch, {})['mfcc'] = self._mfcc_channel(wavs3, ch)

        feats = []
        done = {}          # identical requests are computed once: with `--upstream baseline` the reference's list asks for the baseline feature
                           # twice (run_downstream.py:132-135, 150-151); both entries are then the same tensor, as two raw 'linear' requests already are
        for a in feat_list:
            ft, ch = a['feat_type'], int(a.get('channel', 0))
            log, delta, cmvn = bool(a.get('log', False)), int(a.get('delta', 0)), bool(a.get('cmvn', False))
            key = (ft, ch, log, delta, cmvn)
            if key in done:
                feats.append(done[key])
                continue
            raw = planes[ch][ft]                 # mel: (B, D, F) feature-major; everything else time-major
            if type(raw) is LazyPhase:
                feats.append(raw)                # (*lead, F, K) already; materialises itself on any use but istft()
                continue
            # feats_for_upstream (run_downstream.py:150): what the TRANSFORMER is fed -- unless the owner said there is none (`encoder_side = False`)
            first = len(feats) == 0 and home == wavs3.device and getattr(self, 'encoder_side', True)
            time_major = ft != 'mel'
            if (ft == 'mel' or log or delta or cmvn) and lazy and len(feats) > 0 and getattr(self, 'lazy_features', True):
                # a derived feature that is not the first of the list (run_downstream.py:150: the first is what the upstream is fed): the pipelines
                # that run an upstream never read feats_for_downstream (runner.py:273-284, 556-575 with --from_waveform / the SpecHead path), so it
                # is produced when -- and only when -- something reads it (LazyTensor); shape / dtype / device are known without computing it
                Bp, Dp, Fp = (raw.shape[0], raw.shape[2], raw.shape[1]) if time_major else raw.shape
                feat = LazyTensor((*lead, Fp, Dp * (1 + delta)), home,
                                  lambda raw=raw, tm=time_major, a=(log, delta, cmvn): self._select(raw, tm, *a).reshape(*lead, Fp_(raw, tm), -1))
                done[key] = feat
                feats.append(feat)
                continue
            if ft == 'mel':
                feat = self._select(raw, False, log, delta, cmvn, encoder_side=first)
            elif log or delta or cmvn:
                feat = self._select(raw, True, log, delta, cmvn, encoder_side=first)
            else:
                feat = raw
            side = getattr(feat, '_se_side', None)
            cst = getattr(feat, '_se_colstats', None)
            feat = feat.reshape(*lead, *feat.shape[-2:]).to(home)
            if side is not None:
                if not feat.is_inference():                       # inference tensors track no version: no hand-off, the encoder recomputes its operand
                    feat._se_side = side + (feat._version,)       # (bf16 rows, valid-frame counts, version the pair belongs to)
            if cst is not None and feat.is_cuda and not feat.is_inference():
                feat._se_colstats = cst + (feat._version,)        # (statistics, eps, version of the tensor they describe)
            done[key] = feat
            feats.append(feat)
        return feats

    def istft(self, linears=None, phases=None, linear_power=2, complxs=None):
        # linears, phases: (*, max_feat_len, n_freq) -> (*, max_wav_len)
        if complxs is not None:
            raise NotImplementedError('istft(complxs=) is unused by the reference (runner.py:267 passes linears, phases)')
        assert linears is not None and phases is not None
        wav, _ = self.istft_with_sumsq(linears, phases, linear_power=linear_power)
        return wav

    def istft_with_sumsq(self, linears, phases, linear_power=2, lengths=None, out_len=None, ref=None):
        """se_istft_f32; with `lengths` also returns the masked sum of squares (fused, for the dB normalisation).  With `ref` (the reference
        waveform of that normalisation) returns (wav, sumsq, ref_sumsq or None): the encoded-phase kernel sums it in the same launch."""
        linears, home = self._stage(linears)
        if (type(phases) is LazyPhase and phases._tphase is not None and phases._value is None and float(linear_power) == 2.0 and
                phases._tphase.device == linears.device and home == linears.device):
            return self._istft_tphase(linears, phases._tphase, lengths, out_len, ref=ref)
        phases = phases.to(linears.device)
        lib = _lib.load()
        lead = linears.shape[:-2]
        F, K = linears.shape[-2:]
        lin = linears.contiguous().float().reshape(-1, F, K)
        ph = phases.contiguous().float().reshape(-1, F, K)
        B = lin.shape[0]
        n_out = self._win_args['hop_length'] * (F - 1)
        stride = n_out if out_len is None else max(int(out_len), n_out)
        wav = torch.empty(B, stride, device=lin.device, dtype=torch.float32)
        sumsq = None
        if lengths is not None:
            lengths = lengths.to(device=lin.device, dtype=torch.int64).contiguous()
            sumsq = torch.empty(B, device=lin.device, dtype=torch.float32)
        _lib.check(lib.se_istft_f32(self._plan(lin.device), _lib.ptr(lin), _lib.ptr(ph), B, F, float(linear_power),
                                    _lib.ptr(wav), stride, _lib.ptr(lengths), _lib.ptr(sumsq), _lib.stream()), 'se_istft_f32')
        if ref is not None:
            return wav.reshape(*lead, stride).to(home), (None if sumsq is None else sumsq.to(home)), None
        return wav.reshape(*lead, stride).to(home), (None if sumsq is None else sumsq.to(home))

    def _istft_tphase(self, linears, tphase, lengths, out_len, log_input=False, ref=None):
        """se_istft_tphase_f32: X' = sqrt(linears) * (cos, sin)(tphase) -> waveform (+ the masked square sum when `lengths` is given; + the masked
        square sum of `ref` (B, >= out_len), the reference waveform of the level normalisation, in the same launch).  Returns (wav, sumsq[, ref_sumsq])."""
        lib = _lib.load()
        lead = linears.shape[:-2]
        F, K = linears.shape[-2:]
        lin = linears.contiguous().float().reshape(-1, F, K)
        ph = tphase.reshape(-1, F, K)
        assert ph.shape[0] == lin.shape[0] and ph.is_contiguous()
        B = lin.shape[0]
        n_out = self._win_args['hop_length'] * (F - 1)
        stride = n_out if out_len is None else max(int(out_len), n_out)
        wav = torch.empty(B, stride, device=lin.device, dtype=torch.float32)
        sums = None
        use_ref = (ref is not None and lengths is not None and ref.dim() == 2 and ref.shape[0] == B and ref.dtype == torch.float32 and ref.is_cuda and
                   ref.stride(1) == 1 and ref.shape[1] >= stride)
        if lengths is not None:
            lengths = lengths.to(device=lin.device, dtype=torch.int64).contiguous()
            sums = torch.empty(2 if use_ref else 1, B, device=lin.device, dtype=torch.float32)       # [wav, ref] side by side: one clearing launch
        _lib.check(lib.se_istft_tphase_f32(self._plan(lin.device), _lib.ptr(lin), _lib.ptr(ph), B, F, int(bool(log_input)), _lib.ptr(wav), stride,
                                           _lib.ptr(lengths), None if sums is None else sums[0].data_ptr(),
                                           ref.data_ptr() if use_ref else None, int(ref.stride(0)) if use_ref else 0,
                                           sums[1].data_ptr() if use_ref else None, _lib.stream()), 'se_istft_tphase_f32')
        wav = wav.reshape(*lead, stride)
        if ref is None:
            return wav, (None if sums is None else sums[0])
        return wav, (None if sums is None else sums[0]), (sums[1] if use_ref else None)

    # ---- attributes used by sampler.hist_scoring (sampler.py:226-228) -----------------------------------
    def _stft(self, wav2d, window=None):
        """(N, T) -> (N, K, F, 2) real view, the torch<=1.6 torch.stft layout S3PRL exposed."""
        wav2d, home = self._stage(wav2d)
        out = self._stft_channel(wav2d.contiguous().float().unsqueeze(1), 0, {'complx'})['complx']   # (N, F, 2K)
        N, F, _ = out.shape
        return out.view(N, F, self._n_freq, 2).permute(0, 2, 1, 3).contiguous().to(home)

    @staticmethod
    def _magphase(complx, power=2.0):
        """torchaudio.functional.magphase(power=2) on the (…, 2) real view: thin torch elementwise (not on the hot path)."""
        mag = complx.pow(2).sum(-1).pow(power / 2.0)
        phase = torch.atan2(complx[..., 1], complx[..., 0])
        return mag, phase
