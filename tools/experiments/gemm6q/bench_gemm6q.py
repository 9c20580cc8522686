"""A/B of the persistent 256 x 256 GEMMs on the bench shapes (developer tool; run on the GPU box): gemm6p (round 2, through se_gemm_bf16 with
SE_AMD_GEMM6Q=0) against gemm6q (round 5, called directly) in its padded-output and exact-output forms.  SE_AMD_LIB selects a tagged library."""
import ctypes
import os
import sys

os.environ['SE_AMD_GEMM6Q'] = '0'
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_enhancement_by_s3prl_amd import _lib as L  # noqa: E402

lib = L.load()
dev = torch.device('cuda:0')
fn = lib.se_gemm6q_launch
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
               ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]


def timeit(f, iters=20, warm=3):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


M = 32 * 1001
Mp = (M + 255) // 256 * 256
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for (N, K, act) in [(2304, 768, 0), (3072, 768, 3)]:
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev)
    out = torch.empty(Mp, N, device=dev, dtype=torch.bfloat16)

    def old():
        L.check(lib.se_gemm_bf16(L.ptr(A), K, L.ptr(W), K, L.ptr(bias), None, M, N, K, act, L.ptr(out), None, N, L.stream()), 'gemm')

    def new_pad():
        assert fn(L.ptr(A), K, L.ptr(W), K, L.ptr(bias), M, N, K, act, L.ptr(out), N, Mp, 0, L.stream()) == 0

    def new_exact():
        assert fn(L.ptr(A), K, L.ptr(W), K, L.ptr(bias), M, N, K, act, L.ptr(out), N, M, 0, L.stream()) == 0

    res = {'gemm6p': [], 'gemm6q padded': [], 'gemm6q exact': []}
    for _ in range(rounds):
        res['gemm6p'].append(timeit(old))
        res['gemm6q padded'].append(timeit(new_pad))
        res['gemm6q exact'].append(timeit(new_exact))
    print(f'N={N} K={K} act={act}: ' + ' | '.join(f'{k} ' + '/'.join(f'{v:.1f}' for v in vs) for k, vs in res.items()), flush=True)
