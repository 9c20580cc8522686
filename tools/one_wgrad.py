"""Runs one TN weight-gradient shape a few times (for rocprofv3 --pmc passes): python tools/one_wgrad.py N K splits"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_enhancement_by_s3prl_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device('cuda:0')
N, K, splits = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
M = 32 * 1001
dY = torch.randn(M, N, device=dev).bfloat16()
X = torch.randn(M, K, device=dev).bfloat16()
dW = torch.empty(N, K, device=dev)
ws = torch.empty(splits * N * K, device=dev)
for _ in range(5):
    L.check(lib.se_wgrad_tn_bf16(L.ptr(dY), N, L.ptr(X), K, M, N, K, splits, L.ptr(dW), 0, L.ptr(ws), ws.numel() * 4, L.stream()), 'wgrad_tn')
torch.cuda.synchronize()
