"""Data-parallel glue: one process per GPU, `torch.distributed` ("nccl" == RCCL over xGMI on
ROCm; "gloo" in the CPU tests).  The reference has no distributed code at all; this is what the build adds.

Inference (configs 2, 4): utterances shard i = r (mod world); NO collective on the data path.
Training (configs 3, 5): replicate the downstream parameters; every rank computes the UN-normalised L1 sum and its
masked element count; ONE all-reduce(sum) of (sum, count) gives the global masked mean (objective.py:113-116 is a
global mean over the whole batch -- a mean of per-rank means would break parity on ragged batches); the local
gradients are therefore already scaled by 1/global_count, and ONE all-reduce(sum) over a single flat fp32 buffer
of all downstream gradients yields the exact single-process gradient.  Clipping and the NaN/Inf skip decision
(runner.py:463-470) are then taken on the all-reduced gradient, identically on every rank.
Message sizes: 24 321 floats (LinearResidual 120->201) ... 43.3 M floats (Mockingjay L=6): one flat buffer, one
collective per step -- small messages are latency-bound on the xGMI mesh, large ones want one big ring / direct
reduce-scatter instead of many small buckets."""
import math

import os

import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def shard_indices(n_items, rank=None, world=None):
    """Utterance sharding for inference: item i goes to rank i % world."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    return list(range(rank, n_items, world))


def _host_staged(t):
    """gloo moves HOST memory; handed a device tensor, ProcessGroupGloo stages it through pinned memory on worker threads and streams of its own --
    the path on which the four-rank one-device rehearsal hung with two or more collectives queued (profiles/r04_gloo_4rank/).  This module never
    enters that path: on `backend == 'gloo'` every device tensor is staged HERE (device -> host, collective on the host tensor -- gloo's native mode,
    the one tests/test_dist_gloo.py exercises at world 2 and 4 -- host -> device).  RCCL ('nccl') takes device tensors as they are."""
    return t.is_cuda and dist.get_backend() == 'gloo'


def all_reduce_(t, op=None):
    """in-place all-reduce(sum) of `t` on the default group, through the host on gloo + device tensors"""
    op = dist.ReduceOp.SUM if op is None else op
    if _host_staged(t):
        h = t.detach().cpu()                 # synchronises with the current stream: every kernel that wrote `t` is done
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)
    return t


def all_reduce_sums(sums):
    """reduce_fn for objective.L1: (sum |.|, count) -> global sums (in place, returns the tensor)."""
    if is_distributed():
        all_reduce_(sums)
    return sums


@torch.no_grad()
def broadcast_parameters(module, src=0):
    """Replicas must START equal: rank `src`'s parameters and buffers go to every rank through one flat buffer per dtype
    (identical seeds make this a no-op in value, but nothing else guarantees it -- e.g. a checkpoint loaded on rank 0 only)."""
    if not is_distributed():
        return
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
    by_type = {}
    for t in tensors:
        by_type.setdefault((t.dtype, t.device), []).append(t)
    for group in by_type.values():
        flat = torch.cat([t.reshape(-1) for t in group])
        if _host_staged(flat):
            h = flat.cpu()
            dist.broadcast(h, src=src)
            flat.copy_(h)
        else:
            dist.broadcast(flat, src=src)
        off = 0
        for t in group:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()
    for p in module.parameters():
        torch.autograd.graph.increment_version(p)      # engines key their bf16 operand copies on the version counter


class FlatGradAllReducer:
    """All-reduces (sum) the gradients of `params` through ONE flat fp32 buffer (allocated once)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        p0 = self.params[0]
        self.flat = torch.zeros(self.numel, device=p0.device, dtype=torch.float32)
        self.views, off = [], 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def _gather(self, skip=()):
        """per-parameter gradients -> the flat buffer (`skip`: parameter indices whose views were written in place by the backward).
        Records which parameters HAD a gradient this step (`self.active`): the optimizer must skip the others entirely -- BertAdam's
        `if p.grad is None: continue` -- instead of decaying them against a zero gradient.  On the GPU: ONE batched copy launch (se_multi_copy_f32) instead of a torch
        copy kernel per tensor (~100 launches per step for the Mockingjay encoder)."""
        self.active = [(i in skip) or (p.grad is not None) for i, p in enumerate(self.params)]
        if not self.flat.is_cuda:
            for i, (p, v) in enumerate(zip(self.params, self.views)):
                if i in skip:
                    continue
                if p.grad is None:
                    v.zero_()
                else:
                    v.copy_(p.grad)
            return
        import ctypes
        from . import _lib
        src, dst, sizes, keep = [], [], [], []
        for i, (p, v) in enumerate(zip(self.params, self.views)):
            if i in skip:
                continue
            g = p.grad
            if g is None:
                v.zero_()
                continue
            if g.data_ptr() == v.data_ptr():
                continue                                   # already written in place
            if g.dtype != torch.float32 or not g.is_contiguous() or g.device != self.flat.device:
                v.copy_(g)
                continue
            src.append(g.data_ptr())
            dst.append(v.data_ptr())
            sizes.append(g.numel())
            keep.append(g)
        n = len(src)
        if n:
            lib = _lib.load()
            with torch.cuda.device(self.flat.device):
                _lib.check(lib.se_multi_copy_f32((ctypes.c_void_p * n)(*dst), (ctypes.c_void_p * n)(*src), (ctypes.c_uint64 * n)(*sizes), n,
                                                 _lib.stream()), 'se_multi_copy_f32')

    def reduce(self, copy_back=True, sink=None):
        """copy_back=False leaves the reduced gradients in the flat buffer's views only (the fused optimizer reads them there).
        sink: a BucketedGradSink whose buckets were already written in place and all-reduced during the backward: only the rest of the
        buffer (parameters outside the encoder trunk) is gathered and reduced here."""
        done = sink.done if sink is not None else set()
        self._gather(skip=done)
        if is_distributed():
            if not done:
                all_reduce_(self.flat)
            else:
                rest = [i for i in range(len(self.params)) if i not in done]
                lo = None
                for i in rest + [None]:          # contiguous runs of the remaining parameters
                    if lo is not None and (i is None or i != prev + 1):
                        a = sink.offsets[lo]
                        b = sink.offsets[prev] + self.params[prev].numel()
                        all_reduce_(self.flat[a:b])
                        lo = None
                    if i is not None and lo is None:
                        lo = i
                    prev = i
                sink.wait()
        if copy_back:
            for p, v, a in zip(self.params, self.views, self.active):
                if not a:
                    continue                 # no gradient on any rank: stays None (the optimizer skips it, as the reference's does)
                if p.grad is None:
                    p.grad = v.clone()
                else:
                    p.grad.copy_(v)
        return self.flat


class BucketedGradSink:
    """Gradient sink for the encoder's HIP backward (transformer._EncoderTrainFn): the backward kernels write each parameter's gradient
    straight into its view of the reducer's flat buffer, and after every encoder layer (host callback of se_encoder_bwd_cb_bf16, last layer
    first) that layer's contiguous slice of the buffer is all-reduced asynchronously: the collective is ordered behind the layer's kernels
    on the stream and runs under the backward of the layers still to come, instead of one 173 MB all-reduce after the last of them
    (Mockingjay, 6 layers: 6 x 28 MB + a tail).  Parameters outside the encoder trunk are reduced by FlatGradAllReducer.reduce() as before."""

    def __init__(self, reducer):
        self.reducer = reducer
        self.by_param = {id(p): (i, v) for i, (p, v) in enumerate(zip(reducer.params, reducer.views))}
        offs, o = [], 0
        for p in reducer.params:
            offs.append(o)
            o += p.numel()
        self.offsets = offs
        self.handles = []
        self.done = set()
        self.buckets = self.collectives = 0          # of the last step (tests read them)
        self.launch_stream = None                    # the stream the backward kernels are enqueued on (set by the engine around its C call)
        self._staged = []                            # gloo + device tensors: (lo, hi, device -> host copy done) of every bucket of this step
        self._host = self._stage_stream = None

    def begin(self):
        self.handles, self.done, self._staged = [], set(), []
        self.buckets = self.collectives = 0

    def view(self, p):
        e = self.by_param.get(id(p))
        return None if e is None else e[1]

    def _stream(self):
        """where the bucket's kernels were launched: the stream the engine handed to se_encoder_bwd_cb_bf16 (the callback runs on the
        autograd worker thread, whose current stream need not be that one), else the calling thread's current stream"""
        return self.launch_stream if self.launch_stream is not None else torch.cuda.current_stream(self.reducer.flat.device)

    def bucket_done(self, params):
        idx = sorted(self.by_param[id(p)][0] for p in params if id(p) in self.by_param)
        if not idx:
            return
        self.done.update(idx)
        self.buckets += 1
        if not is_distributed():
            return
        flat = self.reducer.flat
        # contiguous runs of parameter indices = contiguous slices of the flat buffer
        run = [idx[0]]
        for i in idx[1:] + [None]:
            if i is not None and i == run[-1] + 1:
                run.append(i)
                continue
            lo = self.offsets[run[0]]
            hi = self.offsets[run[-1]] + self.reducer.params[run[-1]].numel()
            if _host_staged(flat):
                # gloo rehearsal on device tensors: the bucket leaves for a pinned host buffer NOW, on a side stream behind an event recorded on
                # the launch stream (the copy runs under the backward of the layers still to come); the host all-reduces of ALL buckets are issued
                # together in wait(), where gloo sees host tensors only.  No cap on the buckets in flight any more.
                if self._host is None:
                    self._host = torch.empty(flat.numel(), dtype=flat.dtype, pin_memory=True)
                    self._stage_stream = torch.cuda.Stream(device=flat.device)
                ev = torch.cuda.Event()
                ev.record(self._stream())
                self._stage_stream.wait_event(ev)
                with torch.cuda.stream(self._stage_stream):
                    self._host[lo:hi].copy_(flat[lo:hi], non_blocking=True)
                    done_ev = torch.cuda.Event()
                    done_ev.record(self._stage_stream)
                self._staged.append((lo, hi, done_ev))
            elif flat.is_cuda:
                # RCCL: ProcessGroupNCCL orders the collective behind an event it records on the CURRENT stream at the call -- issue it with the
                # launch stream current, so that event covers every kernel of the bucket whichever thread the callback fires on
                with torch.cuda.stream(self._stream()):
                    self.handles.append(dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
            else:
                self.handles.append(dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
            if os.environ.get('SE_DP_BUCKET_SYNC') == '1' and self.handles:
                self.handles.pop().wait()
            self.collectives += 1
            if i is not None:
                run = [i]

    def wait(self):
        if self._staged:
            flat = self.reducer.flat
            handles = []
            for lo, hi, done_ev in self._staged:          # same order on every rank (the backward's layer order)
                done_ev.synchronize()
                handles.append(dist.all_reduce(self._host[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
            for h in handles:
                h.wait()
            cur = torch.cuda.current_stream(flat.device)
            with torch.cuda.stream(self._stage_stream):
                for lo, hi, _ in self._staged:
                    flat[lo:hi].copy_(self._host[lo:hi], non_blocking=True)
                back = torch.cuda.Event()
                back.record(self._stage_stream)
            cur.wait_event(back)                          # whoever reads the reduced gradients next does so on the current stream
            self._staged = []
        for h in self.handles:
            h.wait()
        self.handles = []


class DataParallelTrainStep:
    """The reference's training step (runner.py:453-471) under data parallelism.
    `forward_fn(batch) -> (log_predicted, linear_tar, stft_lengths)`; criterion = objective.L1."""

    def __init__(self, model, criterion, optimizer, grad_clip=1.0):
        self.model, self.criterion, self.optimizer, self.grad_clip = model, criterion, optimizer, grad_clip
        self.criterion.reduce_fn = all_reduce_sums
        self.reducer = FlatGradAllReducer(list(model.parameters()))
        broadcast_parameters(model)
        # encoder trunks with the HIP backward (Mockingjay): their gradients go straight into the flat buffer, bucket by bucket
        self.sink = None
        engines = []
        if self.reducer.flat.is_cuda:
            engines = [m._engine for m in model.modules() if hasattr(m, '_engine') and hasattr(m, 'model') and hasattr(m._engine, 'encode_train')]
            if engines:
                self.sink = BucketedGradSink(self.reducer)
        self._engines = engines if self.sink is not None else []

    def _attach(self, on):
        """The sink is registered on the engines only while step() runs its backward: a plain loss.backward() on the same model --
        outside this class -- fills .grad as usual."""
        for e in self._engines:
            e.grad_sink = self.sink if on else None

    def step(self, loss):
        """loss already computed with the global-mean criterion; backward, all-reduce, clip, (maybe) step."""
        if self.sink is not None:
            self.sink.begin()
            self._attach(True)
        try:
            loss.backward()
        finally:
            if self.sink is not None:
                self._attach(False)
        fused = getattr(self.optimizer, 'step_fused', None) is not None and self.reducer.flat.is_cuda
        if fused:
            # device path: gradients stay in the flat (all-reduced) buffer; norms in one launch, global clip + per-tensor clip +
            # BertAdam in one more.  The skip decision reads the reduced norm, identically on every rank.
            self.reducer.reduce(copy_back=False, sink=self.sink)
            act = self.reducer.active          # the same on every rank (same graph); a gradient-free parameter is not touched (no decay)
            tab = self.optimizer._fused_table(grads={p: v for p, v, a in zip(self.reducer.params, self.reducer.views, act) if a})
            if tab is not None:
                sumsq = self.optimizer.grad_sumsq(tab)
                gn = float(sumsq.sum().sqrt())
                skipped = math.isnan(gn) or math.isinf(gn)
                if not skipped:
                    self.optimizer.step_fused(tab, sumsq, global_max_norm=self.grad_clip)
                self.optimizer.zero_grad()
                return gn, skipped
            for p, v, a in zip(self.reducer.params, self.reducer.views, act):
                if a:
                    p.grad = v.clone() if p.grad is None else p.grad.copy_(v)
        else:
            self.reducer.reduce(sink=self.sink)
        grad_norm = torch.nn.utils.clip_grad_norm_(self.reducer.params, self.grad_clip)
        gn = float(grad_norm)
        skipped = math.isnan(gn) or math.isinf(gn)          # identical on every rank: taken on the reduced gradient
        if not skipped:
            self.optimizer.step()
        self.optimizer.zero_grad()
        return gn, skipped
