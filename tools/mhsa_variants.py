"""Developer tool (GPU box): the inference attention forward variants side by side at the bench shape (B = 32, T = 1001, 12 heads).
    [SE_AMD_LIB=...] python tools/mhsa_variants.py 0 9 16"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_enhancement_by_s3prl_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device('cuda:0')
B, T, heads = int(os.environ.get('MHSA_B', 32)), int(os.environ.get('MHSA_T', 1001)), 12
torch.manual_seed(0)
q = torch.randn(B * T, 3 * 768, device=dev)
q[:, :768] *= 1.4426950408889634 / 8.0
q = q.bfloat16()
ctx = torch.empty(B * T, 768, device=dev, dtype=torch.bfloat16)
variants = [int(v) for v in sys.argv[1:]] or [0, 8, 9, 16]
res = {v: [] for v in variants}
for rnd in range(5):
    for v in variants:
        for _ in range(3):
            L.check(lib.se_mhsa_fwd_prescaled_variant_bf16(L.ptr(q), None, B, T, heads, L.ptr(ctx), v, L.stream()), 'mhsa')
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            L.check(lib.se_mhsa_fwd_prescaled_variant_bf16(L.ptr(q), None, B, T, heads, L.ptr(ctx), v, L.stream()), 'mhsa')
        b.record()
        torch.cuda.synchronize()
        res[v].append(a.elapsed_time(b) / 20 * 1e3)
tag = os.path.basename(os.environ.get('SE_AMD_LIB', 'libse_amd.so'))
print(f'{tag:22s} B={B:3d} T={T} ' + '   '.join(f'v{v}: {sorted(r)[len(r) // 2]:6.1f} us' for v, r in res.items()), flush=True)
