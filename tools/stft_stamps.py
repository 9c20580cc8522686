import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['SE_AMD_STFT_DBG'] = '1'
from speech_enhancement_by_s3prl_amd import _lib as L
from speech_enhancement_by_s3prl_amd.preprocessor import OnlinePreprocessor
lib = L.load(); dev = torch.device('cuda:0')
P = OnlinePreprocessor().to(dev)
B = 256
wavs = torch.randn(B, 2, 160000, device=dev) * 0.1
F = 1001
power = torch.empty(B, F, 201, device=dev); phase = torch.empty_like(power)
dbg = torch.zeros(B * F * 201, device=dev, dtype=torch.int64)
for _ in range(3):
    dbg.zero_()
    L.check(lib.se_stft_f32(P._plan(dev), L.ptr(wavs), B, 2, 160000, 0, L.ptr(power), L.ptr(phase), dbg.data_ptr(), None, L.stream()), 'stft')
torch.cuda.synchronize()
d = dbg[:8 * 16].cpu().view(8, 16)
names = ['fill', 'passA', 'passB', 'post']
for b in range(8):
    s = [int(v) for v in d[b] if int(v) != 0]
    print(f'b={b}: ' + ' '.join(f'{n}={s[i+1]-s[i]}' for i, n in enumerate(names) if i + 1 < len(s)) + f' total={s[-1]-s[0]}')
