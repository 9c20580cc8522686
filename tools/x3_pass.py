"""developer tool: the enhance pass in a parity mode (TRANSFORMER.set_precision), for rocprofv3 --kernel-trace --stats:
    rocprofv3 --kernel-trace --stats ... -- python3 tools/x3_pass.py [bf16x3|fp32] [batch] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from speech_enhancement_by_s3prl_amd import pipeline, synth  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'bf16x3'
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device('cuda', 0)
cfg = pipeline.make_config()
ckpt = pipeline.synthetic_checkpoint(cfg, seed=0)
up = pipeline.build_upstream(ckpt, dev).set_precision(mode)
pre = pipeline.build_preprocessor(cfg, dev)
step = pipeline.UpstreamEnhanceStep(pre, up)
lengths, wavs = synth.fast_batch(batch, 160000, seed=1, device=dev)
for _ in range(2):
    step(wavs, lengths, 160000)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step(wavs, lengths, 160000)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f'{mode} batch {batch}: {1e3 * dt:.2f} ms per pass, {batch / dt:.0f} utt/s', flush=True)
