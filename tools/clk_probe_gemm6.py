"""Developer tool (GPU box): the in-kernel clock of the persistent 256 x 256 GEMM (gemm6p) with and without its epilogue's stores -- is the launch
power-limited (the clock rises when work is taken away) or stalled (it does not)?  Needs probe builds (see tools/clk_probe.py), e.g. with
-DSE6_ABL=8 (no epilogue), 16 (no store instructions), 32 (stores to L2-resident rows); SE_AMD_LIB selects the library."""
import os
import sys

os.environ.setdefault('SE_AMD_GEMM6Q', '0')
sys.argv = [sys.argv[0]]
import runpy  # noqa: E402

src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'clk_probe.py')).read()
head, tail = src.split("B, T, heads, H = 32, 1001, 12, 768")
body = '''B, T, heads, H = 32, 1001, 12, 768
M = B * T
torch.manual_seed(0)
x = torch.randn(M, 3072, device=dev).bfloat16()
for (N, K, act, lab) in ((2304, 768, 0, 'QKV projection'), (3072, 768, 3, 'FFN1 + GELU')):
    w = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    xa = x[:, :K].contiguous()
    run(f'{lab} (gemm6 persistent) M={M} N={N} K={K}', 'clkprobe_gemm6q' if os.environ['SE_AMD_GEMM6Q'] != '0' else 'clkprobe_gemm6',
        lambda: L.check(lib.se_gemm_bf16(L.ptr(xa), K, L.ptr(w), K, L.ptr(bias), None, M, N, K, act, L.ptr(out), None, N, L.stream()), 'gemm'), 2.0 * M * N * K)
'''
exec(compile(head + body, 'clk_probe_gemm6', 'exec'))
