"""Micro-benchmarks of the individual kernels on the bench shapes (developer tool; run on the GPU box).
    python tools/bench_kernels.py gemm|gemmln|mhsa|stft|wgrad|all
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_enhancement_by_s3prl_amd import _lib  # noqa: E402

L = _lib
lib = _lib.load()
dev = torch.device('cuda:0')


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def bench_gemm():
    M = 32 * 1001
    for (N, K, act, res) in [(2304, 768, 0, False), (768, 768, 0, True), (3072, 768, 3, False), (768, 3072, 0, True), (768, 128, 0, False), (201, 768, 0, False)]:
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        bias = torch.randn(N, device=dev)
        resid = torch.randn(M, N, device=dev) if res else None
        o16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16) if not res else None
        o32 = torch.empty(M, N, device=dev) if res or N == 201 else None
        if N == 201:
            o16 = None

        def run():
            L.check(lib.se_gemm_bf16(L.ptr(A), K, L.ptr(W), K, L.ptr(bias), L.ptr(resid), M, N, K, act, L.ptr(o16), L.ptr(o32), N, L.stream()), 'gemm')
        ms = timeit(run)
        # spot check
        ref = (A[:64].float() @ W.float().T + bias)
        if act == 3:
            ref = torch.nn.functional.gelu(ref)
        if res:
            ref = ref + resid[:64]
        got = (o32 if o32 is not None else o16.float())[:64]
        err = (got - ref).abs().max().item() / ref.abs().max().item()
        print(f'gemm M={M} N={N} K={K} act={act} res={res}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TF/s  relerr {err:.1e}', flush=True)


def bench_gemm_square():
    """the guide's reference shapes (cdna_hip_programming.md section 5: 256^2 8-phase template, 1320-1340 TF at 4096^3 and ~1470 at 8192^3
    on uniform random operands) through this library's kernel, to separate main-loop quality from the K = 768 fill / epilogue share"""
    for n in (4096, 8192):
        A = (torch.rand(n, n, device=dev) * 2 - 1).bfloat16()
        W = (torch.rand(n, n, device=dev) * 2 - 1).bfloat16()
        o16 = torch.empty(n, n, device=dev, dtype=torch.bfloat16)

        def run():
            L.check(lib.se_gemm_bf16(L.ptr(A), n, L.ptr(W), n, None, None, n, n, n, 0, L.ptr(o16), None, n, L.stream()), 'gemm')
        ms = timeit(run)
        print(f'gemm {n}^3 uniform[-1,1): {ms*1e3:8.1f} us  {2.0*n*n*n/ms/1e9:8.1f} TF/s', flush=True)


def bench_gemm_qkv():
    M, N, K = 32 * 1001, 2304, 768
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev)
    o16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)

    def run():
        L.check(lib.se_gemm_bf16(L.ptr(A), K, L.ptr(W), K, L.ptr(bias), None, M, N, K, 0, L.ptr(o16), None, N, L.stream()), 'gemm')
    ms = timeit(run)
    print(f'gemm QKV M={M} N={N} K={K} [SE_AMD_GEMM3_SCHED={os.environ.get("SE_AMD_GEMM3_SCHED")}]: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TF/s', flush=True)


def bench_gemm_ksweep():
    """time(K) at the QKV / FFN1 output shapes: the intercept is the launch's fixed cost (fill + epilogue + output write per tile x rounds)"""
    M = 32 * 1001
    for N, act in ((2304, 0), (3072, 3)):
        for K in (128, 256, 384, 768, 1536, 3072):
            A = torch.randn(M, K, device=dev).bfloat16()
            W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
            bias = torch.randn(N, device=dev)
            o16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)

            def run():
                L.check(lib.se_gemm_bf16(L.ptr(A), K, L.ptr(W), K, L.ptr(bias), None, M, N, K, act, L.ptr(o16), None, N, L.stream()), 'gemm')
            ms = timeit(run)
            print(f'ksweep N={N} act={act} K={K}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TF/s', flush=True)


def bench_wgrad():
    """weight gradient dW = dY^T X: transposes + split-K forward GEMM (old) vs the TN kernel on row-major operands."""
    M = 32 * 1001
    Mp = 32256
    for (N, K) in [(768, 3072), (3072, 768), (2304, 768), (768, 768), (768, 128)]:
        dY = torch.randn(M, N, device=dev).bfloat16()
        X = torch.randn(M, K, device=dev).bfloat16()
        dW = torch.empty(N, K, device=dev)
        ws = torch.empty(8 * N * K, device=dev)
        ta = torch.empty(N, Mp, device=dev, dtype=torch.bfloat16)
        tb = torch.empty(K, Mp, device=dev, dtype=torch.bfloat16)

        def old():
            L.check(lib.se_transpose_bf16(L.ptr(dY), M, N, N, L.ptr(ta), Mp, L.stream()), 't')
            L.check(lib.se_transpose_bf16(L.ptr(X), M, K, K, L.ptr(tb), Mp, L.stream()), 't')
            L.check(lib.se_wgrad_bf16(L.ptr(ta), L.ptr(tb), Mp, N, K, 8, L.ptr(dW), 0, L.ptr(ws), ws.numel() * 4, L.stream()), 'w')
        ms_old = timeit(old)
        ref = dW.clone()
        line = f'wgrad N={N} K={K}: transposes + split-K gemm2 {ms_old*1e3:8.1f} us ({2.0*M*N*K/ms_old/1e9:6.1f} TF/s)'
        for splits in sorted({max(1, min(32, 256 // (((N + 255) // 256) * ((K + 255) // 256)))), 8, 16}):
            ws2 = torch.empty(splits * N * K, device=dev)

            def tn():
                L.check(lib.se_wgrad_tn_bf16(L.ptr(dY), N, L.ptr(X), K, M, N, K, splits, L.ptr(dW), 0, L.ptr(ws2), ws2.numel() * 4, L.stream()), 'wtn')
            ms = timeit(tn)
            err = (dW - ref).abs().max().item() / ref.abs().max().item()
            line += f' | TN s={splits}: {ms*1e3:7.1f} us ({2.0*M*N*K/ms/1e9:6.1f} TF/s, diff {err:.0e})'
        print(line, flush=True)


def bench_lstm():
    """the reference's LSTM head at its configured size (pseudo_noise.yaml:50-53), B = 32 x 10 s: forward and forward + backward"""
    from speech_enhancement_by_s3prl_amd.lstm import LSTM
    head = LSTM(input_size=120, output_size=201, hidden_size=256, num_layers=3, bidirectional=True).to(dev)
    feats = torch.randn(32, 1001, 120, device=dev)
    G = torch.randn(32, 1001, 201, device=dev)

    def fwd():
        with torch.no_grad():
            head(features=feats)

    def fwd_bwd():
        pred, res = head(features=feats)
        (res['log_predicted'] * G).sum().backward()
        head.zero_grad()
    print(f'LSTM head 3 x BiLSTM-256, B=32 T=1001: forward {timeit(fwd, iters=5, warm=2):8.2f} ms   forward+backward {timeit(fwd_bwd, iters=5, warm=2):8.2f} ms', flush=True)


def bench_gemm_ln():
    M, N = 32 * 1001, 768
    for K in (768, 3072):
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        bias, res = torch.randn(N, device=dev), torch.randn(M, N, device=dev)
        lw, lb = torch.ones(N, device=dev), torch.zeros(N, device=dev)
        o32 = torch.empty(M, N, device=dev)
        o16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)

        def run():
            L.check(lib.se_gemm_res_ln_bf16(L.ptr(A), K, L.ptr(W), K, L.ptr(bias), L.ptr(res), L.ptr(lw), L.ptr(lb), 1e-12, M, N, K,
                                            L.ptr(o32), L.ptr(o16), L.stream()), 'gemm_ln')
        ms = timeit(run)
        print(f'gemm+res+LN M={M} N={N} K={K}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TF/s', flush=True)


def bench_res24():
    """FFN-output / attention-output projections on the 24-bit stream: the 128 x 768 row-complete tiles (variant 7)"""
    M, N = 32 * 1001, 768
    for K in (768, 3072):
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        bias, res = torch.randn(N, device=dev), torch.randn(M, N, device=dev).bfloat16()
        nlo = lib.se_gemm_res24_lo_bytes(M)
        rlo = torch.zeros(nlo, device=dev, dtype=torch.uint8)
        lw, lb = torch.ones(N, device=dev), torch.zeros(N, device=dev)
        o16 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        olo = torch.empty(nlo, device=dev, dtype=torch.uint8)
        scratch = torch.zeros(lib.se_gemm_res24_scratch_bytes(), device=dev, dtype=torch.uint8)
        for variant in (7,):      # 8 = the parked 256 x 384 pair-exchange experiment (developer builds)
            def run():
                L.check(lib.se_gemm_res24_ln_bf16(L.ptr(A), K, L.ptr(W), K, L.ptr(bias), L.ptr(res), L.ptr(rlo), L.ptr(lw), L.ptr(lb), 1e-12, M, N, K, None,
                                                  L.ptr(o16), L.ptr(olo), variant, L.ptr(scratch), L.stream()), 'res24')
            ms = timeit(run)
            print(f'gemm+res24+LN variant {variant} M={M} K={K}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TF/s', flush=True)


def bench_mhsa():
    B, T, heads = 32, 1001, 12
    qkv = (torch.randn(B * T, 3 * 768, device=dev)).bfloat16()
    ctx = torch.empty(B * T, 768, device=dev, dtype=torch.bfloat16)

    def run():
        L.check(lib.se_mhsa_fwd_bf16(L.ptr(qkv), None, B, T, heads, L.ptr(ctx), L.stream()), 'mhsa')
    ms = timeit(run)
    print(f'mhsa B={B} T={T}: {ms*1e3:8.1f} us  {4.0*B*heads*T*T*64/ms/1e9:8.1f} TF/s', flush=True)
    # the pre-scaled two-tile-pipeline kernel on the same scores (Q block multiplied by log2(e) / 8 before its bf16 rounding)
    q2 = qkv.float()
    q2[:, :768] *= 1.4426950408889634 / 8.0
    q2 = q2.bfloat16()
    ctx2 = torch.empty_like(ctx)

    def run2():
        L.check(lib.se_mhsa_fwd_prescaled_bf16(L.ptr(q2), None, B, T, heads, L.ptr(ctx2), L.stream()), 'mhsa2')
    ms = timeit(run2)
    d = (ctx2.float() - ctx.float()).abs().max().item() / ctx.float().abs().max().item()
    print(f'mhsa prescaled B={B} T={T}: {ms*1e3:8.1f} us  {4.0*B*heads*T*T*64/ms/1e9:8.1f} TF/s   max diff vs the unscaled kernel {d:.2e}', flush=True)
    # the selectable variants side by side, interleaved (0 = mhsa.hip, 8 = mhsa8.hip: 8-wave alternating segments)
    ctx3 = torch.empty_like(ctx)
    runs = {f'variant {v}': (lambda v=v: L.check(lib.se_mhsa_fwd_prescaled_variant_bf16(L.ptr(q2), None, B, T, heads, L.ptr(ctx3), v, L.stream()), 'mhsa-v'))
            for v in (0, 10)}      # the product's two kernels (developer builds: add the parked variants' numbers)
    for k, (mn, md) in interleaved(runs, rounds=5, iters=20).items():
        print(f'mhsa prescaled {k}: {mn*1e3:8.1f} us (min) {md*1e3:8.1f} us (median)   {4.0*B*heads*T*T*64/md/1e9:8.1f} TF/s', flush=True)
    runs['variant 10']()
    print(f'mhsa variant 10 vs variant 0: bit-identical {bool((ctx3.view(torch.int16) == ctx2.view(torch.int16)).all().item())}', flush=True)


def bench_mhsa_peaked():
    """what the exact fallback of the speculative softmax costs: the same launch with the scores (log2 domain, std 1.44 at gain 1) multiplied by `gain`;
    a wave leaves the speculative path for good the first time a 64-key partial row sum leaves [2^-60, 2^60)"""
    B, T, heads = 32, 1001, 12
    base = torch.randn(B * T, 3 * 768, device=dev)
    ctx = torch.empty(B * T, 768, device=dev, dtype=torch.bfloat16)
    for gain in (1.0, 4.0, 8.0, 12.0, 16.0, 32.0):
        q = base.clone()
        q[:, :768] *= gain * 1.4426950408889634 / 8.0
        q = q.bfloat16()

        def run():
            L.check(lib.se_mhsa_fwd_prescaled_bf16(L.ptr(q), None, B, T, heads, L.ptr(ctx), L.stream()), 'mhsa')
        ms = timeit(run)
        # share of (query row, key tile) partial sums outside the speculation window, from the scores themselves (one head of one utterance)
        qh = q[:T, :64].float(); kh = q[:T, 768:832].float()
        sc = qh @ kh.t()
        pad = (-T) % 64
        e = torch.exp2(torch.nn.functional.pad(sc, (0, pad), value=-1e30).double()).view(T, -1, 64).sum(-1)
        bad = (e >= 2.0 ** 60) | ((e < 2.0 ** -60) & (torch.arange(e.shape[1], device=dev) == 0))
        rows = bad.any(1).float().mean().item()
        waves = bad.any(1).float().view(-1)[: (T // 32) * 32].view(-1, 32).amax(1).mean().item()
        print(f'mhsa peaked gain {gain:5.1f} [SE_AMD_MHSA_SPEC={os.environ.get("SE_AMD_MHSA_SPEC")}]: {ms*1e3:8.1f} us   rows that fall back {rows:6.3f}   waves that fall back {waves:6.3f}   finite {bool(torch.isfinite(ctx.float()).all())}', flush=True)


def bench_mhsa_train():
    """training-mode attention (dropout 0.1): the in-kernel hash against no dropout (the bit-matrix form is parked: tools/experiments/kernels/dropmask.hip (generated once, read three times)"""
    B, T, heads, p = 32, 1001, 12, 0.1
    qkv = torch.randn(B * T, 3 * 768, device=dev).bfloat16()
    d_o = torch.randn(B * T, 768, device=dev).bfloat16()
    ctx = torch.empty(B * T, 768, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(B, heads, T, device=dev)
    dqkv = torch.empty_like(qkv)
    dvec = torch.empty_like(lse)
    seed, site = 1234, 5
    runs = {
        'fwd, hashed': lambda: L.check(lib.se_mhsa_fwd_lse_bf16(L.ptr(qkv), None, B, T, heads, L.ptr(ctx), L.ptr(lse), p, seed, site, L.stream()), 'f'),
        'fwd, no dropout': lambda: L.check(lib.se_mhsa_fwd_lse_bf16(L.ptr(qkv), None, B, T, heads, L.ptr(ctx), L.ptr(lse), 0.0, seed, site, L.stream()), 'f'),
        'bwd, hashed': lambda: L.check(lib.se_mhsa_bwd_bf16(L.ptr(qkv), L.ptr(ctx), L.ptr(d_o), L.ptr(lse), None, B, T, heads, L.ptr(dqkv), L.ptr(dvec),
                                                            p, seed, site, L.stream()), 'b'),
        'bwd, no dropout': lambda: L.check(lib.se_mhsa_bwd_bf16(L.ptr(qkv), L.ptr(ctx), L.ptr(d_o), L.ptr(lse), None, B, T, heads, L.ptr(dqkv), L.ptr(dvec),
                                                                0.0, seed, site, L.stream()), 'b'),
    }
    for k, (mn, md) in interleaved(runs, rounds=3, iters=10).items():
        print(f'mhsa train B={B} T={T} p={p}: {k:18s} {mn*1e3:8.1f} us (min) {md*1e3:8.1f} us (median)', flush=True)


def interleaved(variants, rounds=7, iters=20):
    """cdna_hip_programming.md rule 24: N variants x M rounds interleaved in ONE process; prints min and median per variant"""
    import statistics
    times = {k: [] for k in variants}
    for fn in variants.values():
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for k, fn in variants.items():
            times[k].append(timeit(fn, iters=iters, warm=1))
    return {k: (min(v), statistics.median(v)) for k, v in times.items()}


def bench_stft():
    """STFT / iSTFT stand-alone, old (power + atan2 phase; sqrt / sin / cos) and new (power + unit phasor) forms, one channel and
    the bench pass's two-channel launch, interleaved.  GB/s = SURVEY 8d's algorithmic bytes (2 249 608 B per utterance-channel) / time."""
    from speech_enhancement_by_s3prl_amd.preprocessor import OnlinePreprocessor
    from speech_enhancement_by_s3prl_amd import pipeline
    P = OnlinePreprocessor().to(dev)
    P6 = pipeline.build_preprocessor(pipeline.make_config(), dev)
    byt = 2249608
    for B in (32, 256):
        wavs = torch.randn(B, 3, 160000, device=dev) * 0.1
        fl = [P.get_feat_config('linear', 0), P.get_feat_config('phase', 0)]
        need = {0: {'linear', 'phase', 'mel'}, 1: {'linear', 'phase'}}
        P.lazy_phase = False
        lin, ph = P(wavs, fl)
        P.lazy_phase = True
        lin2, ph2 = P(wavs, fl)

        def old1():
            P.lazy_phase = False
            P(wavs, fl)

        def new1():
            P.lazy_phase = True
            P(wavs, fl)
        res = interleaved({'stft  atan2  1ch': old1, 'stft  phasor 1ch': new1,
                           'stft  atan2  2ch': lambda: P6._stft_two_channels(wavs, need), 'stft  phasor 2ch': lambda: P6._stft_tphase(wavs, need, (B,)),
                           'istft atan2': lambda: P.istft(lin, ph), 'istft phasor': lambda: P.istft(lin2, ph2)})
        for k, (mn, med) in res.items():
            n = 2 * B if '2ch' in k else B
            print(f'{k} B={B}: min {mn*1e3:7.1f} us ({n*byt/mn/1e6:7.1f} GB/s)  median {med*1e3:7.1f} us ({n*byt/med/1e6:7.1f} GB/s)', flush=True)


def bench_hbm():
    """the box's achievable HBM rate for roofline.json's 'confirmed' column: device-to-device copy (read + write) and fill of 2 GiB"""
    n = 1 << 29
    a = torch.empty(n, device=dev, dtype=torch.float32)
    b = torch.empty(n, device=dev, dtype=torch.float32)
    ms = timeit(lambda: b.copy_(a), iters=10)
    print(f'hbm copy 2 GiB -> 2 GiB: {ms*1e3:8.1f} us  {2*4*n/ms/1e6:8.1f} GB/s (read + write)', flush=True)
    ms = timeit(lambda: b.zero_(), iters=10)
    print(f'hbm fill 2 GiB: {ms*1e3:8.1f} us  {4*n/ms/1e6:8.1f} GB/s (write)', flush=True)
    h = torch.empty(1 << 26, dtype=torch.float32).pin_memory()      # 256 MiB pinned
    d = torch.empty(1 << 26, device=dev, dtype=torch.float32)
    ms = timeit(lambda: d.copy_(h, non_blocking=True), iters=5)
    print(f'pinned H2D 256 MiB: {ms*1e3:8.1f} us  {4*(1<<26)/ms/1e6:8.1f} GB/s', flush=True)
    ms = timeit(lambda: h.copy_(d, non_blocking=True), iters=5)
    print(f'pinned D2H 256 MiB: {ms*1e3:8.1f} us  {4*(1<<26)/ms/1e6:8.1f} GB/s', flush=True)
    ms = timeit(lambda: a.sum(), iters=10)
    print(f'hbm read 2 GiB (sum): {ms*1e3:8.1f} us  {4*n/ms/1e6:8.1f} GB/s (read)', flush=True)


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if what in ('gemm', 'all'):
        bench_gemm()
    if what in ('ksweep',):
        bench_gemm_ksweep()
    if what in ('square',):
        bench_gemm_square()
    if what in ('qkv',):
        bench_gemm_qkv()
    if what in ('gemmln', 'all'):
        bench_gemm_ln()
    if what in ('res24', 'all'):
        bench_res24()
    if what in ('mhsa', 'all'):
        bench_mhsa()
    if what in ('mhsa_train',):
        bench_mhsa_train()
    if what in ('mhsa_peaked',):
        bench_mhsa_peaked()
    if what in ('stft', 'all'):
        bench_stft()
    if what in ('wgrad',):
        bench_wgrad()
    if what in ('lstm',):
        bench_lstm()
    if what in ('hbm',):
        bench_hbm()
