"""Developer tool (no GPU): flags compiler-inserted `s_waitcnt vmcnt(0)` inside LOOPS of kernels that issue LDS-DMA (global_load_lds).
Through __builtin_amdgcn_global_load_lds the compiler orders every later LDS read behind the DMA with vmcnt(0), which silently turns a counted
multi-stage ring into a single stage in flight (wgrad.hip shipped that for three rounds).  Waits written by the source (inline asm) are not flagged.
    python tools/isa_wait_lint.py [file.hip ...]      (default: every csrc/*.hip that mentions global_load_lds)
"""
import glob
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, 'speech-enhancement-by-s3prl_amd', 'csrc')
files = sys.argv[1:] or sorted(f for f in glob.glob(os.path.join(csrc, '*.hip')) if 'global_load_lds' in open(f).read())
bad = 0
for f in files:
    out = '/tmp/lint_' + os.path.basename(f) + '.s'
    extra = ['-fno-slp-vectorize'] if re.match(r'(mhsa|stft|istft)', os.path.basename(f)) else []
    subprocess.run(['hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-fno-gpu-rdc', '-I', os.path.join(root, 'include'), '-I', csrc] + extra +
                   ['-S', '--cuda-device-only', '-o', out, f], check=True, stderr=subprocess.DEVNULL)
    kern, in_asm, depth_lines = None, False, []
    has_dma = {}
    findings = {}
    for ln in open(out):
        m = re.match(r'^(_Z\S+):', ln)
        if m:
            kern = m.group(1)
        if kern is None:
            continue
        if '#ASMSTART' in ln:
            in_asm = True
        if '#ASMEND' in ln:
            in_asm = False
        if 'global_load_lds' in ln:
            has_dma[kern] = True
        if re.match(r'^\.LBB', ln):
            in_loop = 'in Loop' in ln or 'Loop Header' in ln
            findings.setdefault(kern, {'loop': False})['loop'] = in_loop
        if 's_waitcnt' in ln and 'vmcnt(0)' in ln and not in_asm and findings.get(kern, {}).get('loop'):
            findings[kern].setdefault('hits', 0)
            findings[kern]['hits'] += 1
        if 's_endpgm' in ln:
            kern_done = kern
    for k, v in findings.items():
        if has_dma.get(k) and v.get('hits'):
            bad += 1
            print(f'{os.path.basename(f)}: {k[:90]}: {v["hits"]} compiler-inserted vmcnt(0) inside loops of an LDS-DMA kernel')
print(f'{bad} kernel(s) flagged in {len(files)} file(s)')
