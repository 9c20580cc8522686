import sys, torch
sys.path.insert(0, '.')
from speech_enhancement_by_s3prl_amd import _lib, pipeline, synth
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
cfg = pipeline.make_config(layers=6)
ckpt = pipeline.synthetic_checkpoint(cfg, seed=0)
up = pipeline.build_upstream(ckpt, dev); pre = pipeline.build_preprocessor(cfg, dev)
step = pipeline.UpstreamEnhanceStep(pre, up)
lengths, wavs = synth.fast_batch(32, 160000, seed=1, device=dev)
for _ in range(3): step(wavs, lengths, 160000)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(wavs, lengths, 160000); torch.cuda.synchronize()
ev = prof.events()
for e in ev:
    n = e.name
    if ('Memcpy' in n or 'copy_' == n or n == 'aten::copy_' or 'aten::clone' in n or 'aten::contiguous' in n or 'aten::to' == n or 'aten::_to_copy' in n or 'aten::fill_' in n or 'aten::zero_' in n) and e.cpu_time_total > 0:
        st = [s for s in (e.stack or []) if 'speech' in s or 'bench' in s][:2]
        print(n, [str(s) for s in e.input_shapes][:2] if e.input_shapes else '', st)
