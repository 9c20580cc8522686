"""CPU (hipcc cross-compiles gfx950 without a GPU): properties of the generated code that no numerical test can see.

Round 4 found that the weight-gradient kernel had carried a compiler-inserted `s_waitcnt vmcnt(0)` in front of every step's first LDS fragment read
since round 2 -- the LDS-DMA went through __builtin_amdgcn_global_load_lds, for which the compiler orders every later LDS read behind the DMA -- so its
"three tiles in flight" ring had never had more than one.  Results were right, the kernel was 10 % slow.  This test keeps that from coming back."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'speech-enhancement-by-s3prl_amd', 'csrc')


def _isa(src, tmp_path):
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('hipcc not available')
    out = str(tmp_path / (os.path.basename(src) + '.s'))
    subprocess.run([hipcc, '-O3', '-std=c++17', '--offload-arch=gfx950', '-fno-gpu-rdc', '-I', os.path.join(ROOT, 'include'), '-I', CSRC, '-S',
                    '--cuda-device-only', '-o', out, src], check=True, stderr=subprocess.DEVNULL)
    return open(out).read().split('\n')


def _kernel(lines, mangled_prefix):
    start = next(i for i, l in enumerate(lines) if l.startswith(mangled_prefix) and l.rstrip().endswith(':') or (l.startswith(mangled_prefix) and ':' in l))
    end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
    return lines[start:end + 1]


def _compiler_vmcnt0_next_to_mfma(body):
    """compiler-inserted (not inline-asm) `s_waitcnt vmcnt(0)` inside a loop block whose neighbourhood issues MFMAs and no compiler-visible global load"""
    hits, in_asm, in_loop = [], False, False
    for i, l in enumerate(body):
        if '#ASMSTART' in l:
            in_asm = True
        if '#ASMEND' in l:
            in_asm = False
        if re.match(r'^\.LBB', l):
            in_loop = 'Loop' in l
        if 's_waitcnt' in l and 'vmcnt(0)' in l and not in_asm and in_loop:
            ctx = body[max(0, i - 40):i + 40]
            n_mfma = sum('v_mfma' in c for c in ctx)
            n_load = sum(('global_load_dword' in c and 'lds' not in c) or 'scratch_load' in c for c in ctx)
            if n_mfma > 0 and n_load == 0:
                hits.append((i, l.strip()))
    return hits


@pytest.mark.parametrize('stag', [0, 1])
def test_wgrad_ring_is_not_drained_by_the_compiler(tmp_path, stag):
    lines = _isa(os.path.join(CSRC, 'wgrad.hip'), tmp_path)
    body = _kernel(lines, f'_ZN2se15wgrad_tn_kernelILi{stag}EE')
    assert sum('v_mfma' in l for l in body) >= 16 and sum('global_load_lds' in l for l in body) >= 4        # the right kernel, LDS-DMA by inline asm
    hits = _compiler_vmcnt0_next_to_mfma(body)
    assert not hits, f'compiler-inserted vmcnt(0) inside the MFMA loop of wgrad_tn_kernel<{stag}>: {hits[:3]}'
    assert not any('scratch_' in l for l in body), 'wgrad_tn_kernel spills'


@pytest.mark.parametrize('src,prefix,flags', [
    ('gemm6.hip', '_ZN2se18gemm6p_bf16_kernelILi0ELi2ELi0EE', []),          # persistent 256 x 256 GEMM (QKV)
    ('gemm6.hip', '_ZN2se18gemm6p_bf16_kernelILi3ELi2ELi0EE', []),          # ... with the GELU epilogue (FFN1)
    ('gemm4.hip', '_ZN2se19gemm7_res_ln_kernelILi0ELi1ELi1ELi1ELi1ELi0EE', []),   # row-complete GEMM + residual + LayerNorm on the 24-bit stream
    ('mhsa8.hip', '_ZN2se16mhsaN_fwd_kernelILi8ELi4ELi1EE', ['-fno-slp-vectorize']),   # the inference attention forward
])
def test_hot_lds_dma_loops_are_not_drained_by_the_compiler(tmp_path, src, prefix, flags):
    """the other LDS-DMA kernels of the inference pass: same property (their DMA is inline asm, their waits are counted by hand)"""
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('hipcc not available')
    out = str(tmp_path / (src + '.s'))
    subprocess.run([hipcc, '-O3', '-std=c++17', '--offload-arch=gfx950', '-fno-gpu-rdc', '-I', os.path.join(ROOT, 'include'), '-I', CSRC] + flags +
                   ['-S', '--cuda-device-only', '-o', out, os.path.join(CSRC, src)], check=True, stderr=subprocess.DEVNULL)
    body = _kernel(open(out).read().split('\n'), prefix)
    assert sum('v_mfma' in l for l in body) >= 16 and sum('global_load_lds' in l for l in body) >= 2
    hits = _compiler_vmcnt0_next_to_mfma(body)
    assert not hits, f'{prefix}: compiler-inserted vmcnt(0) inside an MFMA loop: {hits[:3]}'
