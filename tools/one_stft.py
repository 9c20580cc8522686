"""Runs STFT + iSTFT a few times at batch B (for rocprofv3 --pmc passes): python tools/one_stft.py B"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_enhancement_by_s3prl_amd.preprocessor import OnlinePreprocessor  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device('cuda:0')
P = OnlinePreprocessor().to(dev)
wavs = torch.randn(B, 2, 160000, device=dev) * 0.1
fl = [P.get_feat_config('linear', 0), P.get_feat_config('phase', 0)]
for _ in range(4):
    lin, ph = P(wavs, fl)
    wav = P.istft(lin, ph)
torch.cuda.synchronize()
