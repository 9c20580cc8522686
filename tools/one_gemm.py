"""Runs one GEMM shape a few times (for rocprofv3 --pmc passes): python tools/one_gemm.py N K out(bf16|f32) res(0|1) act"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_enhancement_by_s3prl_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device('cuda:0')
N, K = int(sys.argv[1]), int(sys.argv[2])
out, res, act = sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
M = 32 * 1001
A = torch.randn(M, K, device=dev).bfloat16()
W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
bias = torch.randn(N, device=dev)
resid = torch.randn(M, N, device=dev) if res else None
o16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if out == 'bf16' else None
o32 = torch.empty(M, N, device=dev) if out == 'f32' else None
for _ in range(5):
    L.check(lib.se_gemm_bf16(L.ptr(A), K, L.ptr(W), K, L.ptr(bias), L.ptr(resid), M, N, K, act, L.ptr(o16), L.ptr(o32), N, L.stream()), 'gemm')
torch.cuda.synchronize()
