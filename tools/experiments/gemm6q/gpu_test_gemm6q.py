"""GPU parity of the persistent 256 x 256 GEMM with the pipelined epilogue (csrc/gemm6q.hip: se_gemm6q_launch, the kernel behind se_gemm_bf16 for the
encoder's QKV and FFN1 projections at bench size) -- called DIRECTLY, so that no dispatch switch can route around it: against the fp64 product of
the same bf16 operands, with guard rows behind the output (a ragged last row tile sends its missing row groups to a dump and masks the cut one), for
identity and GELU, K = 256 (no steady-state K-tiles at all), 768 and 3072, and M with M % 16 == 0 as well as != 0."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _launch(A, W, bias, act, out, M, rows_alloc=None):
    """rows_alloc: how many rows of `out` the kernel may write (default M: the masked / dump form; >= ceil(M / 256) * 256: whole tiles are written)"""
    from speech_enhancement_by_s3prl_amd import _lib as L
    lib = L.load()
    fn = lib.se_gemm6q_launch
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                   ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    N, K = W.shape
    return fn(L.ptr(A), K, L.ptr(W), K, L.ptr(bias), M, N, K, act, L.ptr(out), N, M if rows_alloc is None else rows_alloc, 0, L.stream())


@pytest.mark.parametrize('M,N,K,act', [
    (32032, 2304, 768, 0),        # the bench's QKV projection (32032 = 125 x 256 + 32: two whole 16-row groups in the last row tile)
    (32032, 3072, 768, 3),        # the bench's FFN1 projection + GELU
    (8229, 2304, 768, 0),         # 8229 = 32 x 256 + 37: a cut 16-row group (EXEC-masked store) and groups that go to the dump
    (7681, 2304, 256, 3),         # four K-tiles per output tile: first / last K-tile pairs only, one row in the last row tile
    (5400, 3072, 3072, 0),        # long K loop
    (7936, 2304, 768, 3),         # M a multiple of 256: no ragged tile
])
@pytest.mark.parametrize('padded', [False, True])
def test_gemm6q_vs_fp64(gpu, M, N, K, act, padded):
    torch.manual_seed(M + N + K + act)
    A = torch.randn(M, K, device=gpu).bfloat16()
    W = (torch.randn(N, K, device=gpu) * 0.05).bfloat16()
    bias = torch.randn(N, device=gpu)
    guard = 300
    out = torch.full((M + guard, N), 777.0, device=gpu, dtype=torch.bfloat16)
    Mp = (M + 255) // 256 * 256
    rc = _launch(A, W, bias, act, out, M, Mp if padded else None)
    assert rc == 0, rc
    torch.cuda.synchronize()
    # the padded form may write the rest of its last row tile, nothing behind it; the masked form writes no row past M
    assert torch.all(out[(Mp if padded else M):] == 777.0), 'rows past the allowed range were written'
    worst = 0.0
    for lo in range(0, M, 4096):      # fp64 reference in row blocks (memory)
        hi = min(M, lo + 4096)
        ref = A[lo:hi].double() @ W.double().T + bias.double()
        if act == 3:
            ref = torch.nn.functional.gelu(ref)
        err = (out[lo:hi].double() - ref).abs().max().item() / ref.abs().max().item()
        worst = max(worst, err)
    assert worst < 5e-3, worst          # bf16 rounding of the output: 2^-9 of the value


def test_gemm6q_exact_integers(gpu):
    """A with one 1 per row (a permutation-like gather of W's columns), W small integers: every output is exact in bf16, so any row / column / lane-pair
    slip of the pipelined epilogue (quadrant order, v_permlane16_swap, dump / mask selection) shows as a wrong integer."""
    M, N, K = 7713, 2304, 768
    idx = (torch.arange(M, device=gpu) * 7 + 3) % K
    A = torch.zeros(M, K, device=gpu)
    A[torch.arange(M, device=gpu), idx] = 1.0
    W = ((torch.arange(N * K, device=gpu).reshape(N, K) * 31 + torch.arange(N, device=gpu)[:, None] * 17) % 127 - 63).float()
    bias = ((torch.arange(N, device=gpu) % 5) - 2).float()
    out = torch.zeros(M + 64, N, device=gpu, dtype=torch.bfloat16)
    rc = _launch(A.bfloat16(), W.bfloat16(), bias, 0, out, M)
    assert rc == 0, rc
    ref = W.T[idx] + bias          # (M, N): row m = column idx[m] of W
    assert torch.equal(out[:M].float(), ref)
    assert torch.count_nonzero(out[M:]) == 0


def test_gemm6q_declines_what_it_cannot_do(gpu):
    """too few tiles, an odd number of K-tiles, N not a multiple of 256: returns 1 (the dispatcher then uses the other kernels), launches nothing"""
    for (M, N, K) in [(1001, 2304, 768), (32032, 2304, 704), (32032, 2000, 768)]:
        A = torch.zeros(M, K, device=gpu, dtype=torch.bfloat16)
        W = torch.zeros(N, K, device=gpu, dtype=torch.bfloat16)
        out = torch.full((M, N), 5.0, device=gpu, dtype=torch.bfloat16)
        assert _launch(A, W, torch.zeros(N, device=gpu), 0, out, M) == 1
        torch.cuda.synchronize()
        assert torch.all(out == 5.0)
