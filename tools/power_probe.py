"""Developer tool (GPU box): socket power and clocks while one kernel runs back to back (rocm-smi sampled from a thread): is the launch under the
power cap?  python tools/power_probe.py [qkv|ffn1|mhsa|pass]   (SE_AMD_LIB selects a library, e.g. the -DSE6_ABL builds of tools/clk_probe_gemm6.py)"""
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_enhancement_by_s3prl_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device('cuda:0')
what = sys.argv[1] if len(sys.argv) > 1 else 'qkv'
M = 32 * 1001
if what in ('qkv', 'ffn1'):
    N, K, act = (2304, 768, 0) if what == 'qkv' else (3072, 768, 3)
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    run = lambda: L.check(lib.se_gemm_bf16(L.ptr(A), K, L.ptr(W), K, L.ptr(bias), None, M, N, K, act, L.ptr(out), None, N, L.stream()), 'gemm')      # noqa: E731
elif what == 'mhsa':
    qkv = torch.randn(M, 2304, device=dev).bfloat16()
    ctx = torch.empty(M, 768, device=dev, dtype=torch.bfloat16)
    run = lambda: L.check(lib.se_mhsa_fwd_prescaled_bf16(L.ptr(qkv), None, 32, 1001, 12, L.ptr(ctx), L.stream()), 'mhsa')      # noqa: E731
else:
    from speech_enhancement_by_s3prl_amd import pipeline, synth
    cfg = pipeline.make_config()
    ckpt = pipeline.synthetic_checkpoint(cfg, seed=0)
    step = pipeline.UpstreamEnhanceStep(pipeline.build_preprocessor(cfg, dev), pipeline.build_upstream(ckpt, dev))
    lengths, wavs = synth.fast_batch(32, 160000, seed=1, device=dev)
    run = lambda: step(wavs, lengths, 160000)      # noqa: E731

samples, stop = [], False


def sampler():
    while not stop:
        try:
            r = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--showmaxpower'], capture_output=True, text=True, timeout=10).stdout
        except Exception as e:      # noqa: BLE001
            r = repr(e)
        samples.append(r)
        time.sleep(0.3)


for _ in range(5):
    run()
torch.cuda.synchronize()
th = threading.Thread(target=sampler)
th.start()
t0 = time.time()
n = 0
while time.time() - t0 < 6.0:
    for _ in range(20):
        run()
    torch.cuda.synchronize()
    n += 20
dt = time.time() - t0
stop = True
th.join()
pw = [float(x) for s in samples for x in re.findall(r'Power \(W\):\s*([\d.]+)', s)]
cap = [float(x) for s in samples for x in re.findall(r'Max Graphics Package Power \(W\):\s*([\d.]+)', s)]
sclk = [int(x) for s in samples for x in re.findall(r'sclk clock level: \d+: \((\d+)Mhz\)', s)]
print(f'{what}: {dt / n * 1e6:.1f} us per launch over {dt:.1f} s | power W: {pw} | cap W: {sorted(set(cap))} | sclk MHz: {sclk}')
if not pw:
    print('raw sample:', samples[-1][:1500] if samples else None)
