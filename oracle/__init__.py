"""CPU oracle for the speech-enhancement hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and
only as the checker / the reported CPU baseline.  Nothing under ``speech-enhancement-by-s3prl_amd/`` imports
it; the product path fails loudly if the HIP extension is missing.

What it is: a plain PyTorch-CPU fp32 (optionally fp64) restatement of the algorithm of the reference
path, each function citing the reference file:line it follows.

Pinning status (see DESIGN.md "Oracle"):
  * rows C1, C2, D1, D2, E1, the SISDR objective, add_noise, normalize_wav_decibel, collate_fn,
    matching and sisdr_eval are PINNED against outputs of the reference's own code, generated in the
    build container by tests/golden/make_golden.py and committed as tests/golden/reference_golden.npz.
  * rows A1, A2, A6 (STFT / power+phase / iSTFT) are pinned against torch.stft / torch.istft (the entry
    points S3PRL wraps) and an independent float64 DFT-matrix implementation (oracle/dft64.py).
  * rows A3, A4 (mel / delta / CMVN), B1-B4 (S3PRL TRANSFORMER + spec head), C3/C4 and E2 (BertAdam) live
    in the un-vendored, un-pinned S3PRL dependency (README.md:12-13 of the reference is a bare git clone)
    that is absent from /root/reference and from the container: for those rows the oracle restates the
    published S3PRL / torchaudio-0.6 / pytorch-pretrained-BERT algorithm and is
    **PARITY UNPINNED vs original S3PRL**; it is anchored on the reference's call sites (shapes,
    conventions) and on first-principles known answers (tests/test_oracle_*.py).
"""
from . import preprocessor, encoder, heads, decode, objective, optim, dft64  # noqa: F401
