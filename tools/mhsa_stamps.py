"""Developer tool (GPU box): where one wave of the inference attention forward spends its cycles, per phase of the 64-key tile, at 1 / 2 / 3 waves
per SIMD.  Needs the stamp build:  SE_AMD_BUILD_TAG=stamps SE_AMD_BUILD_STAMPS=1 python speech-enhancement-by-s3prl_amd/build.py
    SE_AMD_LIB=speech-enhancement-by-s3prl_amd/libse_amd.stamps.so python tools/mhsa_stamps.py
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speech_enhancement_by_s3prl_amd import _lib as L  # noqa: E402

lib = L.load()
fn = lib.se_mhsa_fwd_stamps_bf16
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 2 + [ctypes.c_int, ctypes.c_void_p]
dev = torch.device('cuda:0')
B, T, heads = 32, 1001, 12
q = torch.randn(B * T, 3 * 768, device=dev)
q[:, :768] *= 1.4426950408889634 / 8.0
q = q.bfloat16()
ctx = torch.empty(B * T, 768, device=dev, dtype=torch.bfloat16)
nwg = ((T + 127) // 128) * heads * B
names = ['QK^T MFMAs issued', 'first exp issued', 'last pack issued', 'PV MFMAs issued', 'next tile -> LDS', 'barrier', 'prologue', 'kernel']
for occ in (1, 2, 3):
    st = torch.zeros(nwg * 4 * 8, device=dev, dtype=torch.int64)
    for _ in range(3):
        L.check(fn(L.ptr(q), None, B, T, heads, L.ptr(ctx), L.ptr(st), occ, L.stream()), 'stamps')
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        L.check(fn(L.ptr(q), None, B, T, heads, L.ptr(ctx), L.ptr(st), occ, L.stream()), 'stamps')
    b.record()
    torch.cuda.synchronize()
    s = st.view(nwg, 4, 8).double()
    full = s[((torch.arange(nwg, device=dev) >> 3) % 8) != 7]     # workgroups whose query tile is full (XCD remap of mhsa.hip: query tile = (linear id >> 3) % 8; the 8th is ragged)
    ntile = 16.0
    per = full[:, :, :6].mean((0, 1)) / ntile
    tot = per.sum().item()
    print(f'--- {occ} wave(s) per SIMD: launch {a.elapsed_time(b) / 10 * 1e3:.1f} us (stamp build); one wave, cycles per 64-key tile (mean over {full.shape[0] * 4} waves): total {tot:.0f}')
    for i in range(6):
        print(f'    -> {names[i]:20s} {per[i].item():7.0f}  ({100 * per[i].item() / tot:4.1f} %)')
    print(f'    prologue {full[:, :, 6].mean().item():.0f}   kernel {full[:, :, 7].mean().item():.0f}   (16 tiles x total = {16 * tot:.0f})')
    for w in range(4):
        pw = full[:, w, :6].mean(0) / ntile
        print(f'    wave {w}: ' + ' '.join(f'{v:6.0f}' for v in pw.tolist()))


# ---- the 8-wave alternating kernel (mhsa8.hip): cycles per segment and per barrier wait, waves 0-3 (one segment ahead) and 4-7 apart
fn8 = lib.se_mhsa8_fwd_stamps_bf16
fn8.restype = ctypes.c_int
fn8.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3
nwg8 = ((T + 255) // 256) * heads * B
st = torch.zeros(nwg8 * 8 * 10, device=dev, dtype=torch.int64)
for _ in range(3):
    L.check(fn8(L.ptr(q), None, B, T, heads, L.ptr(ctx), L.ptr(st), L.stream()), 'stamps8')
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    L.check(fn8(L.ptr(q), None, B, T, heads, L.ptr(ctx), L.ptr(st), L.stream()), 'stamps8')
b.record()
torch.cuda.synchronize()
s = st.view(nwg8, 8, 10).double()
seg = ['C1 (QK^T)', 'barrier', 'L1 (V reads, softmax, DMA)', 'barrier', 'C2 (PV)', 'barrier', 'L2 (K reads)', 'barrier']
print(f'--- mhsa8: launch {a.elapsed_time(b) / 10 * 1e3:.1f} us (stamp build); cycles per 64-key tile, mean over workgroups')
for grp, sl in (('waves 0-3', slice(0, 4)), ('waves 4-7', slice(4, 8))):
    per = s[:, sl, :8].mean((0, 1)) / 16.0
    print(f'  {grp}: total {per.sum().item():.0f}; loop {s[:, sl, 8].mean().item():.0f} cycles')
    for i in range(8):
        print(f'      {seg[i]:28s} {per[i].item():7.0f}')
