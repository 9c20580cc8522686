"""Developer tool (GPU box): the clock the chip HOLDS inside the dominant kernels at the bench shape (csrc/clkprobe.h).  Needs the probe build:
    SE_AMD_BUILD_TAG=clk SE_AMD_EXTRA_DEFINES=-DSE_AMD_CLKPROBE python speech-enhancement-by-s3prl_amd/build.py
    SE_AMD_LIB=speech-enhancement-by-s3prl_amd/libse_amd.clk.so python tools/clk_probe.py
Each kernel is launched back to back on random data for ~1.5 s first (the clock settles under load), then the probes of the last launch are read:
in-kernel clock = delta(s_memtime) / delta(s_memrealtime) x 100 MHz, median over workgroups; `cycles` = median workgroup lifetime in shader cycles.
"""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_enhancement_by_s3prl_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device('cuda:0')
SLOTS = 8192
LAST_DETAIL = ''


def read(name):
    fn = getattr(lib, 'se_dev_' + name)
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p]
    buf = np.zeros(4 * SLOTS, dtype=np.uint64)
    assert fn(buf.ctypes.data) == 0
    r = buf.reshape(SLOTS, 4).astype(np.float64)
    r = r[r[:, 0] > 0]
    dt, dr = r[:, 2] - r[:, 0], r[:, 3] - r[:, 1]
    ok = dr > 50                                   # >= 0.5 us of lifetime
    clk = dt[ok] / dr[ok] * 0.1                    # GHz
    span_us = (r[:, 3].max() - r[:, 1].min()) / 100.0
    life = dr[ok] / 100.0                          # us
    start = (r[ok, 1] - r[:, 1].min()) / 100.0
    global LAST_DETAIL
    LAST_DETAIL = (f'      workgroup lifetime us: mean {life.mean():.1f}  p5 {np.percentile(life, 5):.1f}  p50 {np.percentile(life, 50):.1f}  p95 {np.percentile(life, 95):.1f}  max {life.max():.1f}'
                   f' | sum of lifetimes / (CU slots x span): {life.sum() / span_us:.0f} workgroups resident on average'
                   f' | starts: p25 {np.percentile(start, 25):.1f} p50 {np.percentile(start, 50):.1f} p75 {np.percentile(start, 75):.1f} max {start.max():.1f} us')
    return np.median(clk), np.percentile(clk, 5), np.percentile(clk, 95), np.median(dt[ok]), int(ok.sum()), span_us


def run(label, name, launch, flops):
    launch()
    torch.cuda.synchronize()
    read(name)                       # reading clears the probe buffer: what is read below is this kernel's last launch only
    t0 = time.time()
    n = 0
    while time.time() - t0 < 1.5:
        for _ in range(50):
            launch()
        torch.cuda.synchronize()
        n += 50
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        launch()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 50 * 1e3
    clk, lo, hi, cyc, nwg, span = read(name)
    tf = flops / us * 1e-6
    print(f'{label:58s} {us:7.1f} us  {tf:7.1f} TF/s = {tf / 2500:.3f} of 2.5 PF | in-kernel clock {clk:.3f} GHz (5-95 %: {lo:.3f}-{hi:.3f}), '
          f'workgroup lifetime {cyc:9.0f} cycles, {nwg} workgroups, first start -> last end {span:.1f} us | '
          f'of the matrix peak AT that clock: {tf / (2500 * clk / 2.4):.3f}', flush=True)
    print(LAST_DETAIL, flush=True)


B, T, heads, H = 32, 1001, 12, 768
M = B * T
torch.manual_seed(0)
qkv = torch.randn(M, 3 * H, device=dev)
qkv[:, :H] *= 1.4426950408889634 / 8.0
qkv = qkv.bfloat16()
ctx = torch.empty(M, H, device=dev, dtype=torch.bfloat16)
run('attention forward (mhsaN<8,4,1>) B=32 T=1001', 'clkprobe_mhsa',
    lambda: L.check(lib.se_mhsa_fwd_prescaled_variant_bf16(L.ptr(qkv), None, B, T, heads, L.ptr(ctx), 10, L.stream()), 'mhsa'), 4.0 * B * heads * T * T * 64)

x = torch.randn(M, 3072, device=dev).bfloat16()
for (N, K, act, lab) in ((2304, 768, 0, 'QKV projection'), (3072, 768, 3, 'FFN1 + GELU'), (3072, 768, 0, 'FFN1 shape WITHOUT the GELU'), (2304, 768, 3, 'QKV shape WITH a GELU')):
    w = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    xa = x[:, :K].contiguous()
    run(f'{lab} (gemm6 persistent) M={M} N={N} K={K}', 'clkprobe_gemm6',
        lambda: L.check(lib.se_gemm_bf16(L.ptr(xa), K, L.ptr(w), K, L.ptr(bias), None, M, N, K, act, L.ptr(out), None, N, L.stream()), 'gemm'), 2.0 * M * N * K)

# the row-complete projections on the 24-bit residual stream (gemm7_res_ln_kernel, 128 x 768 tiles, one per CU)
N = 768
for K, lab in ((768, 'attention-output projection + residual + LN'), (3072, 'FFN2 + residual + LN')):
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias, res = torch.randn(N, device=dev), torch.randn(M, N, device=dev).bfloat16()
    nlo = lib.se_gemm_res24_lo_bytes(M)
    rlo = torch.zeros(nlo, device=dev, dtype=torch.uint8)
    lw, lb = torch.ones(N, device=dev), torch.zeros(N, device=dev)
    o16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    olo = torch.empty(nlo, device=dev, dtype=torch.uint8)
    scratch = torch.zeros(lib.se_gemm_res24_scratch_bytes(), device=dev, dtype=torch.uint8)
    run(f'{lab} (gemm7) M={M} K={K}', 'clkprobe_gemm7',
        lambda: L.check(lib.se_gemm_res24_ln_bf16(L.ptr(A), K, L.ptr(W), K, L.ptr(bias), L.ptr(res), L.ptr(rlo), L.ptr(lw), L.ptr(lb), 1e-12, M, N, K, None,
                                                  L.ptr(o16), L.ptr(olo), 7, L.ptr(scratch), L.stream()), 'res24'), 2.0 * M * N * K)

# the weight-gradient kernel (wgrad_tn_kernel: dW = dY^T X on row-major operands, split over the 32 032 rows; the slab reduce rides along)
for (N, K, lab) in ((3072, 768, 'FFN1'), (768, 3072, 'FFN2'), (2304, 768, 'QKV')):
    dY = torch.randn(M, N, device=dev).bfloat16()
    X = torch.randn(M, K, device=dev).bfloat16()
    dW = torch.empty(N, K, device=dev)
    splits = max(1, min(32, 256 // (((N + 255) // 256) * ((K + 255) // 256))))
    ws2 = torch.empty(splits * N * K, device=dev)
    run(f'weight gradient {lab} (wgrad_tn, {splits} row splits) N={N} K={K}', 'clkprobe_wgrad',
        lambda: L.check(lib.se_wgrad_tn_bf16(L.ptr(dY), N, L.ptr(X), K, M, N, K, splits, L.ptr(dW), 0, L.ptr(ws2), ws2.numel() * 4, L.stream()), 'wtn'), 2.0 * M * N * K)
