"""developer tool: time the encoded-phase STFT (stft2.hip) with parts of the chunk body switched off (SE_AMD_STFT_ABLATE bit mask; results are wrong,
only the time matters): python tools/stft_ablate.py [B]"""
import os, sys, subprocess
B = sys.argv[1] if len(sys.argv) > 1 else '256'
code = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from speech_enhancement_by_s3prl_amd import pipeline
from tools.bench_kernels import timeit, dev
B = int(sys.argv[1])
P6 = pipeline.build_preprocessor(pipeline.make_config(), dev)
wavs = torch.randn(B, 3, 160000, device=dev) * 0.1
for name, need in (('1ch lin+ph', {0: {'linear', 'phase'}}), ('2ch bench ', {0: {'linear', 'phase', 'mel'}, 1: {'linear', 'phase'}})):
    ts = [timeit(lambda: P6._stft_tphase(wavs, need, (B,)), iters=20, warm=3) for _ in range(5)]
    print(f'  {name}: min {min(ts)*1e3:7.1f} us', flush=True)
'''
for mask, what in ((0, 'full'), (1, '- pass A'), (2, '- pass B'), (3, '- pass A, B'), (4, '- post'), (8, '- global stores'), (16, '- loads'), (32, '- fill'), (64, '- mel'),
                   (7, '- A, B, post'), (31 + 32 + 64, 'barriers + loop only'), (8 + 16, '- loads, stores')):
    print(f'ablate {mask:3d} ({what}):', flush=True)
    env = dict(os.environ, SE_AMD_STFT_ABLATE=str(mask))
    subprocess.run([sys.executable, '-c', code, B], env=env, stderr=subprocess.DEVNULL)
