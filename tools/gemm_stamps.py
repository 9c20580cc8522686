"""Developer tool: per-phase s_memtime stamps of the ping-pong GEMM main loop (SE_AMD_GEMM_DBG=17)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['SE_AMD_GEMM_DBG'] = '16'
from speech_enhancement_by_s3prl_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device('cuda:0')
M, N, K = 32 * 1001, 2304, 768
A = torch.randn(M, K, device=dev).bfloat16()
W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
bias = torch.randn(N, device=dev)
o32 = torch.empty(M, N, device=dev, dtype=torch.float32)
# the stamp buffer rides in the `residual` argument: it must be large enough for the (garbage) residual reads too
buf = torch.zeros(M * N // 2 + 8 * 8 * 256, device=dev, dtype=torch.int64)
for _ in range(3):
    buf.zero_()
    L.check(lib.se_gemm_bf16(L.ptr(A), K, L.ptr(W), K, L.ptr(bias), buf.data_ptr(), M, N, K, 0, None, L.ptr(o32), N, L.stream()), 'gemm')
torch.cuda.synchronize()
b = buf[:8 * 8 * 256].cpu().view(8, 8, 256)
names = ['reads0', 'B0', 'mfma0', 'B1', 'reads1', 'issue', 'dmawait', "B0'", 'mfma1', "B1'"]
NS = len(names)
for wg in (0, 3):
    for wave in (0, 4):
        s = b[wg, wave]
        n = int((s != 0).sum())
        t0 = int(s[0])
        print(f'wg {wg} wave {wave}: {n} stamps; start->loop {int(s[1]) - t0} cycles')
        base = 2
        for t in range(12):
            seg = s[base + NS * t: base + NS * (t + 1)]
            if int(seg[-1]) == 0:
                break
            prev = int(s[base + NS * t - 1]) if t > 0 else int(s[1])
            d = [int(seg[0]) - prev] + [int(seg[i]) - int(seg[i - 1]) for i in range(1, NS)]
            print(f'  t={t:2d} ' + ' '.join(f'{nm}={v:5d}' for nm, v in zip(names, d)) + f'  total={int(seg[-1]) - prev}')
