"""Developer experiment: the evaluate()-style pass on two half batches issued to two HIP streams (separate engines / workspaces),
against one full batch on one stream.  Are the big one-round kernels (gemm4: 251 workgroups) better fed when two of them share
the chip out of phase?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_enhancement_by_s3prl_amd import pipeline, synth  # noqa: E402

dev = torch.device('cuda:0')
cfg = pipeline.make_config(layers=6)
ckpt = pipeline.synthetic_checkpoint(cfg, seed=0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
nstreams = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = 20
lengths, wavs = synth.fast_batch(B, 160000, seed=1000, device=dev)


def make():
    up = pipeline.build_upstream(ckpt, dev)
    pre = pipeline.build_preprocessor(cfg, dev)
    return pipeline.UpstreamEnhanceStep(pre, up)


full = make()
for _ in range(3):
    full(wavs, lengths, 160000)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    full(wavs, lengths, 160000)
torch.cuda.synchronize()
t_full = (time.perf_counter() - t0) / steps
parts = [make() for _ in range(nstreams)]
streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
h = B // nstreams
chunks = [(wavs[i * h:(i + 1) * h].contiguous(), lengths[i * h:(i + 1) * h].contiguous()) for i in range(nstreams)]
for _ in range(3):
    for st, p, (w, l) in zip(streams, parts, chunks):
        with torch.cuda.stream(st):
            p(w, l, 160000)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    for st, p, (w, l) in zip(streams, parts, chunks):
        with torch.cuda.stream(st):
            p(w, l, 160000)
torch.cuda.synchronize()
t_split = (time.perf_counter() - t0) / steps
print(f'B={B}: one stream {t_full * 1e3:.3f} ms / step ({B / t_full:.0f} utt/s);  {nstreams} streams x {h} utterances {t_split * 1e3:.3f} ms / step ({B / t_split:.0f} utt/s)')
