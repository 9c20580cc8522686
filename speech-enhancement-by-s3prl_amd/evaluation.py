"""Metric scoring on the device: SI-SDR of every utterance of a batch without moving the
waveforms to the host.  Mirrors evaluation.sisdr_eval (evaluation.py:5-10) and the per-utterance trimming of
runner.py:597-603.  PESQ / STOI stay third-party CPU metrics (absent offline)."""
import torch

from . import _lib


def sisdr_batch(wav_pred, wav_tar, lengths=None, eps=1e-10):
    """wav_pred, wav_tar (B, T) fp32 device tensors; lengths (B,) -> SI-SDR (B,) in dB, each over its first lengths[b] samples."""
    if not wav_pred.is_cuda:
        raise _lib.SEError('sisdr_batch runs on MI355X only (no CPU fallback)')
    lib = _lib.load()
    dev = wav_pred.device
    a, b = wav_pred.contiguous().float(), wav_tar.contiguous().float()
    assert a.shape == b.shape and a.dim() == 2
    B, T = a.shape
    ln = None if lengths is None else lengths.to(dev, torch.int64).contiguous()
    sums = torch.empty(3 * B, device=dev, dtype=torch.float64)
    out = torch.empty(B, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _lib.check(lib.se_sisdr_f32(_lib.ptr(a), _lib.ptr(b), T, _lib.ptr(ln), B, float(eps), _lib.ptr(sums), _lib.ptr(out), _lib.stream()),
                   'se_sisdr_f32')
    return out


def sisdr_eval(src, tar, sr=16000, eps=1e-10):
    """Drop-in for evaluation.sisdr_eval (one utterance, returns a float)."""
    return float(sisdr_batch(src.reshape(1, -1), tar.reshape(1, -1), None, eps)[0])


class MetricStage:
    """The scoring stage behind the path (runner.py:586-603, evaluation.py:12-35) with the GPU kept busy.

    The reference, per batch: `wav_predicted.detach().cpu()` / `wav_tar.detach().cpu()` (synchronous D2H on the compute stream), then
    joblib workers run every metric on every utterance (`metric_fn(predicted[:length], target[:length])`) while the GPU idles, then the
    next batch starts.  Here `submit()` returns at once: the two waveform tensors cross PCIe into pinned slots on a copy stream (ordered
    behind the kernels that produced them by an event, not by a host sync), and the host metrics of batch i run in a worker pool while
    the GPU already enhances batch i + 1.  Metrics given as the string 'sisdr' never leave the device (`sisdr_batch`: three fp64 sums
    per utterance).  Aggregation is the reference's: per batch the mean over utterances of each metric, summed over batches, divided by
    the number of batches (`scores_sum / n_sample`, runner.py:603-617).

    `metrics`: list of 'sisdr' or callables `fn(src_1d_cpu_tensor, tar_1d_cpu_tensor) -> float` (the signature of evaluation.py's
    `pesq_nb_eval`, `stoi_eval`, ...; PESQ / STOI themselves are third-party CPU packages that are absent offline -- any callable of that
    shape plugs in).  `n_jobs` = worker threads (the metrics are numpy / C code that releases the GIL; `args.n_jobs`, runner.py:596)."""

    def __init__(self, metrics, device, n_jobs=4, depth=2, eps=1e-10):
        import concurrent.futures
        self.metrics = list(metrics)
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise _lib.SEError('MetricStage scores the output of the MI355X path: device must be a HIP device')
        self.eps = eps
        self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=max(1, int(n_jobs)))
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.depth = max(2, int(depth))
        self._slots = [None] * self.depth            # (pinned pred, pinned tar, copied event, futures using the slot)
        self._batches = []                           # per batch: list over metrics of (device tensor | list of futures)
        self._n = 0

    def _slot(self, i, shape):
        s = self._slots[i]
        if s is None or s[0].shape != shape:
            s = [torch.empty(shape, dtype=torch.float32).pin_memory(), torch.empty(shape, dtype=torch.float32).pin_memory(), torch.cuda.Event(), []]
            self._slots[i] = s
        return s

    def submit(self, wav_predicted, wav_tar, lengths):
        """wav_predicted, wav_tar (B, T) device tensors, lengths (B,) (device or host).  Returns immediately."""
        if not wav_predicted.is_cuda:
            raise _lib.SEError('MetricStage.submit takes the device tensors of the enhance step (no CPU fallback)')
        B = wav_predicted.shape[0]
        host_fns = [(k, fn) for k, fn in enumerate(self.metrics) if not isinstance(fn, str)]
        per_metric = [None] * len(self.metrics)
        for k, fn in enumerate(self.metrics):
            if isinstance(fn, str):
                if fn != 'sisdr':
                    raise ValueError(f"MetricStage: unknown device metric '{fn}' (only 'sisdr')")
                per_metric[k] = sisdr_batch(wav_predicted, wav_tar, lengths, self.eps)          # stays on the device until average()
        if host_fns:
            s = self._slot(self._n % self.depth, tuple(wav_predicted.shape))
            pp, pt, copied, users = s
            for f in users:                          # the slot's previous batch has been scored (host buffers are about to be overwritten)
                f.result()
            users.clear()
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(self.device))
            self.copy_stream.wait_event(done)        # D2H behind the producing kernels, off the compute stream
            with torch.cuda.stream(self.copy_stream):
                pp.copy_(wav_predicted.detach().float(), non_blocking=True)
                pt.copy_(wav_tar.detach().float(), non_blocking=True)
                copied.record(self.copy_stream)
            lens = [int(v) for v in (lengths.tolist() if torch.is_tensor(lengths) else lengths)]

            def score(fn, b, ev=copied, pp=pp, pt=pt, lens=lens):
                ev.synchronize()                     # a worker waits for the copy; the submitting thread never does
                n = lens[b]
                return float(fn(pp[b, :n], pt[b, :n]))
            for k, fn in host_fns:
                futs = [self.pool.submit(score, fn, b) for b in range(B)]
                users.extend(futs)
                per_metric[k] = futs
        self._batches.append(per_metric)
        self._n += 1

    def average(self):
        """(n_metrics,) fp32 tensor: sum over batches of the per-batch utterance means / number of batches (runner.py:603-617).  Blocks
        until every submitted batch has been scored."""
        total = torch.zeros(len(self.metrics), dtype=torch.float64)
        for per_metric in self._batches:
            for k, v in enumerate(per_metric):
                if torch.is_tensor(v):
                    total[k] += float(v.double().mean())
                else:
                    total[k] += sum(f.result() for f in v) / max(1, len(v))
        return (total / max(1, len(self._batches))).float()

    def close(self):
        self.pool.shutdown(wait=True)
