"""GPU test of the parked bit-matrix dropout mask (dropmask.hip + the masked attention entries): needs a developer build
(SE_AMD_BUILD_EXPERIMENTS=kernels).  Moved out of tests/test_gpu_encoder_train.py in round 5 with the kernels."""
@pytest.mark.parametrize('B,T,heads,drop', [(2, 301, 3, 0.1), (1, 1001, 2, 0.1), (1, 64, 1, 0.5), (1, 513, 1, 0.25)])
def test_dropout_bit_matrices_equal_the_hashed_mask(gpu, B, T, heads, drop):
    """csrc/dropmask.hip against the oracle's keep_mask (the restatement of csrc/dropout.h the hashed kernels are tested with): BIT-EXACT, in both
    layouts -- query-major word pairs (even keys, odd keys) per 64-key tile, key-major words of 32 queries"""
    import numpy as np
    L = _L()
    lib = L.load()
    seed, site = 0x0fedcba987654321, 3
    nr, nc = lib.se_mhsa_dropmask_bytes(B, T, heads, 0) // 4, lib.se_mhsa_dropmask_bytes(B, T, heads, 1) // 4
    mr = torch.zeros(nr, device=gpu, dtype=torch.int32)
    mc = torch.zeros(nc, device=gpu, dtype=torch.int32)
    L.check(lib.se_mhsa_dropmask(B, T, heads, drop, seed, site, L.ptr(mr), L.ptr(mc), L.stream()), 'dropmask')
    keep = oenc.keep_mask(seed, site, B * heads * T, T, drop).reshape(B * heads, T, T).numpy().astype(bool)
    Wr, Tc, Wc = 4 * ((T + 127) // 128), 128 * ((T + 127) // 128), 8 * ((T + 255) // 256)
    R = mr.cpu().numpy().view(np.uint32).reshape(B * heads, T, Wr)
    C = mc.cpu().numpy().view(np.uint32).reshape(B * heads, Tc, Wc)
    # query-major: key k of row q = bit (k % 64) // 2 of word 2 (k // 64) + (k & 1)
    k = np.arange(T)
    got_r = (R[:, :, 2 * (k // 64) + (k & 1)] >> ((k % 64) // 2).astype(np.uint32)) & 1
    assert np.array_equal(got_r.astype(bool), keep)
    # key-major: query q of key k = bit q % 32 of word q // 32
    q = np.arange(T)
    got_c = (C[:, :T, :][:, :, q // 32] >> (q % 32).astype(np.uint32)) & 1          # [bh, key, query]
    assert np.array_equal(got_c.astype(bool).transpose(0, 2, 1), keep)
