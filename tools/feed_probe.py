#!/usr/bin/env python
"""Where does the host-fed loop lose time?  (developer probe behind DESIGN.md section 6, PCIe-inclusive rate)
    python tools/feed_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speech_enhancement_by_s3prl_amd import pipeline, synth

dev = torch.device('cuda', 0)
cfg = pipeline.make_config(layers=6)
ckpt = pipeline.synthetic_checkpoint(cfg, seed=0)
up = pipeline.build_upstream(ckpt, dev)
pre = pipeline.build_preprocessor(cfg, dev)
step = pipeline.UpstreamEnhanceStep(pre, up)
lengths, wavs = synth.fast_batch(32, 160000, seed=1, device=dev)
hl, hw = lengths.cpu().pin_memory(), wavs.cpu().pin_memory()
K = 20


def timed(name, fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{name:60s} {1e3*(t2-t0)/K:7.3f} ms per iteration  (host-side issue {1e3*(t1-t0)/K:7.3f} ms)', flush=True)


timed('step on resident tensors', lambda: step(wavs, lengths, 160000))
side = torch.cuda.Stream(device=dev)
dw, dl = torch.empty_like(wavs), torch.empty_like(lengths)


def copy_only():
    with torch.cuda.stream(side):
        dw.copy_(hw, non_blocking=True)
timed('61 MB pinned H2D on a side stream, nothing else', copy_only)


def copy_main():
    dw.copy_(hw, non_blocking=True)
timed('61 MB pinned H2D on the compute stream, nothing else', copy_main)


def serial():
    dw.copy_(hw, non_blocking=True)
    dl.copy_(hl, non_blocking=True)
    step(dw, dl, 160000)
timed('copy then step, same stream (the reference order)', serial)

ev = [torch.cuda.Event(), torch.cuda.Event()]
done = [torch.cuda.Event(), torch.cuda.Event()]
bufs = [(torch.empty_like(wavs), torch.empty_like(lengths)) for _ in range(2)]
state = {'i': 0}
for e in done:
    e.record()


def overlapped():            # copy i on the side stream while step i-1 runs; the copy only waits for step i-2 (its buffer's last reader)
    i = state['i']
    cur = torch.cuda.current_stream()
    w, l = bufs[i & 1]
    with torch.cuda.stream(side):
        side.wait_event(done[i & 1])
        w.copy_(hw, non_blocking=True)
        l.copy_(hl, non_blocking=True)
        ev[i & 1].record(side)
    cur.wait_event(ev[i & 1])
    step(w, l, 160000)
    done[i & 1].record(cur)
    state['i'] += 1
timed('copy on side stream (waits for step i-2 only) + step i, no host sync', overlapped)


def pipelined():             # software-pipelined by hand: issue copy i+1 BEFORE launching step i
    i = state['i']
    cur = torch.cuda.current_stream()
    w, l = bufs[i & 1]
    wn, ln = bufs[(i + 1) & 1]
    with torch.cuda.stream(side):
        side.wait_event(done[(i + 1) & 1])
        wn.copy_(hw, non_blocking=True)
        ln.copy_(hl, non_blocking=True)
        ev[(i + 1) & 1].record(side)
    cur.wait_event(ev[i & 1])
    step(w, l, 160000)
    done[i & 1].record(cur)
    state['i'] += 1
state['i'] = 0
with torch.cuda.stream(side):
    bufs[0][0].copy_(hw, non_blocking=True)
    bufs[0][1].copy_(hl, non_blocking=True)
    ev[0].record(side)
timed('copy i+1 issued before step i (side stream), no host sync', pipelined)

from speech_enhancement_by_s3prl_amd.feeder import HostBatchFeeder
t0 = time.perf_counter()
for l, w in HostBatchFeeder([(hl, hw)] * K, dev):
    step(w, l, 160000)
torch.cuda.synchronize()
print(f'{"HostBatchFeeder loop":60s} {1e3*(time.perf_counter()-t0)/K:7.3f} ms per iteration', flush=True)
