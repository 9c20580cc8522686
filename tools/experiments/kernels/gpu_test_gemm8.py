"""GPU test of the parked 256 x 384 pair-exchange kernel (gemm8.hip): needs a developer build (SE_AMD_BUILD_EXPERIMENTS=kernels).
Moved out of tests/test_gpu_encoder_blocks.py in round 5 with the kernel; `gemm` / `_lib` helpers as in that file."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from speech_enhancement_by_s3prl_amd import _lib
    return _lib


@pytest.mark.parametrize('M,K', [(32032, 3072), (602, 768), (1001, 3072), (257, 128)])
def test_gemm_res24_pair_exchange_vs_single_tile(gpu, M, K):
    """The 256 x 384 tile kernel whose two column halves exchange LayerNorm statistics across workgroups (gemm8) against the 128 x 768
    row-complete kernel (gemm7) on the same 24-bit-stream inputs, and both against fp64: both output forms (fp32 rows; bf16 + lo bytes),
    ragged row counts (602 = 2 x 256 + 90, 257: one row into the second tile), the bench shape, repeated launches (flags self-reset)."""
    L = _lib()
    lib = L.load()
    torch.manual_seed(M + K)
    N = 768
    A = torch.randn(M, K, device=gpu).bfloat16()
    W = (torch.randn(N, K, device=gpu) * 0.03).bfloat16()
    bias = torch.randn(N, device=gpu) * 0.1
    res = torch.randn(M, N, device=gpu).bfloat16()
    nlo = lib.se_gemm_res24_lo_bytes(M)
    res_lo = torch.zeros(nlo, device=gpu, dtype=torch.uint8)            # low bytes 0: the residual is exactly the bf16 tensor
    lw, lb = torch.randn(N, device=gpu), torch.randn(N, device=gpu)
    scratch = torch.zeros(lib.se_gemm_res24_scratch_bytes(), device=gpu, dtype=torch.uint8)
    ref = torch.nn.functional.layer_norm(A.double() @ W.double().T + bias.double() + res.double(), (N,), lw.double(), lb.double(), 1e-12)

    def run(variant, fp32_out):
        o32 = torch.full((M, N), float('nan'), device=gpu) if fp32_out else None
        o16 = None if fp32_out else torch.zeros(M, N, device=gpu, dtype=torch.bfloat16)
        olo = None if fp32_out else torch.zeros(nlo, device=gpu, dtype=torch.uint8)
        L.check(lib.se_gemm_res24_ln_bf16(L.ptr(A), K, L.ptr(W), K, L.ptr(bias), L.ptr(res), L.ptr(res_lo), L.ptr(lw), L.ptr(lb), 1e-12, M, N, K,
                                          L.ptr(o32), L.ptr(o16), L.ptr(olo), variant, L.ptr(scratch), L.stream()), 'se_gemm_res24_ln_bf16')
        return o32, o16, olo

    for rep in range(3):                                                 # repeated launches on the same scratch: the pair flags reset themselves
        o8, _, _ = run(8, True)
    o7, _, _ = run(7, True)
    scale = ref.abs().max().item()
    assert (o8.double() - ref).abs().max().item() < 2e-5 * scale + 2e-4
    assert (o7.double() - ref).abs().max().item() < 2e-5 * scale + 2e-4
    assert (o8 - o7).abs().max().item() < 1e-4 * scale                  # same products; the statistics are combined in a different order
    _, b8, l8 = run(8, False)
    _, b7, l7 = run(7, False)
    assert (b8.double() - ref).abs().max().item() < 2 ** -8 * scale
    agree = (b8 == b7).float().mean().item()
    assert agree > 0.999, agree                                          # bf16 roundings of values a few 1e-6 apart
    assert torch.equal(scratch[-(128 * 2 * 4 + 256):], torch.zeros(128 * 2 * 4 + 256, device=gpu, dtype=torch.uint8))   # flags + error word left clear
    # the low bytes (same tile-major positions in both kernels) differ by at most one step where the values are a few 1e-6 apart
    d = (l8.view(torch.int8).int() - l7.view(torch.int8).int()).abs()
    assert (d <= 1).float().mean().item() > 0.99


