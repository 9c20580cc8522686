"""Developer tool: per-phase s_memtime stamps of the TN weight-gradient main loop (SE_AMD_WGRAD_STAMPS=1)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['SE_AMD_WGRAD_STAMPS'] = '1'
from speech_enhancement_by_s3prl_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device('cuda:0')
N, K, splits = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
M = 32 * 1001
dY = torch.randn(M, N, device=dev).bfloat16()
X = torch.randn(M, K, device=dev).bfloat16()
dW = torch.empty(N, K, device=dev)
ws = torch.zeros(splits * N * K + 8 * 8 * 256 * 2, device=dev)
for _ in range(3):
    ws[splits * N * K:].zero_()
    L.check(lib.se_wgrad_tn_bf16(L.ptr(dY), N, L.ptr(X), K, M, N, K, splits, L.ptr(dW), 0, L.ptr(ws), ws.numel() * 4, L.stream()), 'wgrad_tn')
torch.cuda.synchronize()
b = ws[splits * N * K:].view(torch.int64).cpu().view(8, 8, 256)
names = ['top->wait', 'wait', 'barrier', 'issue', 'compute']
for wg in (0, 5):
    for wave in (0, 7):
        s = b[wg, wave]
        n = int((s != 0).sum())
        print(f'wg {wg} wave {wave}: {n} stamps; prologue {int(s[1]) - int(s[0])}')
        for t in range(4, 20):
            seg = [int(s[1 + 4 * t + i]) for i in range(5)]
            prev = int(s[1 + 4 * t - 1]) if t > 0 else int(s[1])
            d = [seg[0] - int(s[4 * t])] + [seg[i] - seg[i - 1] for i in range(1, 4)] + [int(s[1 + 4 * (t + 1)]) - seg[3]]
            print(f'  t={t:2d} ' + ' '.join(f'{nm}={v:5d}' for nm, v in zip(names[1:], d[1:])) + f'  stage={int(s[1 + 4 * (t + 1)]) - seg[0]}')
