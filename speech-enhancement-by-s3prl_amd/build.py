"""Builds libse_amd.so (all HIP kernels + the C ABI of include/se_amd.h) for gfx950 with hipcc.

In-tree build: the .so lands next to this file, travels to the GPU box with the snapshot and is
git-ignored.  No torch / pybind involvement: the library is plain C ABI, bound via ctypes (_lib.py).

    python speech-enhancement-by-s3prl_amd/build.py [--force] [--jobs N]
"""
import argparse
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
# SE_AMD_BUILD_TAG=<tag>: a developer build (stamps, ablation defines) lands in libse_amd.<tag>.so / build_<tag>/ beside the product's library
_TAG = os.environ.get('SE_AMD_BUILD_TAG')
OBJ = os.path.join(HERE, 'build' + ('_' + _TAG if _TAG else ''))
LIB = os.path.join(HERE, 'libse_amd' + ('.' + _TAG if _TAG else '') + '.so')
ARCH = 'gfx950'
FLAGS = ['-O3', '-std=c++17', '-fPIC', f'--offload-arch={ARCH}', '-fno-gpu-rdc', '-Wall', '-Wno-unused-function',
         '-Wno-comment', '-I' + os.path.join(HERE, '..', 'include'), '-I' + CSRC]
if os.environ.get('SE_AMD_BUILD_STAMPS') == '1':       # developer build: in-kernel s_memtime stamps (tools/*_stamps.py); never for measurements
    FLAGS.append('-DSE_AMD_STAMPS')
if os.environ.get('SE_AMD_BUILD_EXPERIMENTS'):           # developer builds with the parked kernels of tools/experiments/<dir>/: their call sites are compiled in
    FLAGS.append('-DSE_AMD_EXPERIMENTS')
if os.environ.get('SE_AMD_EXTRA_DEFINES'):              # developer A/B builds, e.g. -DSE_AMD_OLD_CODEC
    FLAGS.extend(os.environ['SE_AMD_EXTRA_DEFINES'].split())
# per-file extra flags.  The flash attention forward and the STFT / iSTFT are VALU-issue bound and v_pk_*_f32 (what the SLP vectoriser
# makes of adjacent fp32 adds / multiplies) costs more issue time there than the two plain instructions it replaces: MHSA 147 -> 139 us,
# STFT 2.57 -> 2.78 TB/s.  Applied to every file it is a small net loss (GEMM epilogues, element-wise passes), hence per file.
FILE_FLAGS = {'mhsa.hip': ['-fno-slp-vectorize'], 'mhsa2.hip': ['-fno-slp-vectorize'], 'mhsa3.hip': ['-fno-slp-vectorize'], 'mhsa8.hip': ['-fno-slp-vectorize'], 'mhsa9.hip': ['-fno-slp-vectorize'], 'mhsa_pipe.hip': ['-fno-slp-vectorize'], 'stft.hip': ['-fno-slp-vectorize'], 'stft_small.hip': ['-fno-slp-vectorize'], 'istft.hip': ['-fno-slp-vectorize'],
              'stft2.hip': ['-fno-slp-vectorize'], 'istft2.hip': ['-fno-slp-vectorize']}


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError('hipcc not found')


def _digest(paths):
    h = hashlib.sha256()
    for p in sorted(paths):
        h.update(os.path.basename(p).encode())      # not the absolute path: the tree is copied to another root on the GPU box
        with open(p, 'rb') as f:
            h.update(f.read())
    h.update(' '.join(FLAGS).replace(HERE, '.').encode())
    h.update(repr(sorted(FILE_FLAGS.items())).encode())
    return h.hexdigest()


def sources():
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))
    # SE_AMD_BUILD_EXPERIMENTS=<dir>[,<dir>]: developer builds that add the kernels parked under tools/experiments/<dir>/ (not part of the product library)
    for exp in filter(None, os.environ.get('SE_AMD_BUILD_EXPERIMENTS', '').split(',')):
        d = os.path.join(HERE, '..', 'tools', 'experiments', exp)
        srcs += sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith('.hip'))
    return srcs


def build(force=False, jobs=None, verbose=True):
    srcs = sources()
    hdrs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h'))
    hdrs.append(os.path.join(HERE, '..', 'include', 'se_amd.h'))
    stamp = os.path.join(OBJ, 'stamp')
    dig = _digest(srcs + hdrs)
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        if verbose:
            print('[build] libse_amd.so up to date')
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    cc = _hipcc()
    # a failed rebuild must not leave the PREVIOUS library behind as if it were this tree's (it travels to the GPU box with the snapshot, and a
    # `build.py | tail` in a shell line hides the failure: round 4 measured two "A/B"s on a stale library that way): the stamp goes first, and
    # the library with it if a compile or the link fails -- loading then fails loudly
    for stale in (stamp,):
        if os.path.exists(stale):
            os.remove(stale)

    def _fail(msg):
        if os.path.exists(LIB):
            os.remove(LIB)
        raise RuntimeError(msg)

    def compile_one(src):
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + '.o')
        cmd = [cc] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + ['-c', src, '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            _fail('hipcc failed: ' + ' '.join(cmd) + '\n' + r.stdout + r.stderr)
        return obj, r.stderr

    jobs = jobs or min(len(srcs), os.cpu_count() or 4)
    with concurrent.futures.ThreadPoolExecutor(jobs) as ex:
        results = list(ex.map(compile_one, srcs))
    for obj, err in results:
        if verbose and err.strip():
            print(err.strip())
    objs = [o for o, _ in results]
    cmd = [cc, '-shared', '-fPIC', f'--offload-arch={ARCH}', '-o', LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        _fail('link failed: ' + ' '.join(cmd) + '\n' + r.stdout + r.stderr)
    with open(stamp, 'w') as f:
        f.write(dig)
    if verbose:
        print(f'[build] built {LIB} from {len(srcs)} sources')
    return LIB


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--force', action='store_true')
    ap.add_argument('--jobs', type=int, default=None)
    a = ap.parse_args()
    try:
        build(force=a.force, jobs=a.jobs)
    except RuntimeError as e:
        print(e, file=sys.stderr)
        sys.exit(1)
