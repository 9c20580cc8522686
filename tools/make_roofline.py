#!/usr/bin/env python
"""roofline.json: nominal peaks, what the box confirms, and every hot kernel's achieved fraction, from one evidence
set gathered by tools/final_evidence.sh.
    python tools/make_roofline.py r02c          # reads profiles/r02c_*, writes roofline.json at the repo root"""
import csv
import json
import os
import re
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
P = lambda name: os.path.join(root, 'profiles', f'{tag}_{name}')      # noqa: E731

bench = json.loads(open(P('bench.json')).read().strip().splitlines()[-1])
txt = open(P('bench_kernels.txt')).read()
stats = {r['Name']: r for r in csv.DictReader(open(P('bench_kernel_stats.csv')))}


def grab(pattern, cast=float):
    m = re.search(pattern, txt)
    return cast(m.group(1)) if m else None


def avg_us(substr):
    subs = (substr, 'se::stft_small_kernel') if substr == 'se::stft_kernel' else (substr,)      # the STFT has two builds since round 3
    tot = calls = 0
    for name, r in stats.items():
        if any(x in name for x in subs):
            tot += float(r['TotalDurationNs'])
            calls += int(r['Calls'])
    return tot / calls / 1e3 if calls else None


M, H = 32 * 1001, 768
flop = {'qkv': 2.0 * M * 2304 * H, 'ffn1': 2.0 * M * 3072 * H, 'mhsa': 4.0 * 32 * 12 * 1001 * 1001 * 64}
out = {
    'device': 'AMD Instinct MI355X (gfx950), one GPU of a pool box; clocks not pinned',
    'evidence': f'profiles/{tag}_* (tools/final_evidence.sh {tag})',
    'peaks_nominal': {'hbm_GBps': 8000.0, 'mfma_bf16_dense_TFLOPs': 2500.0, 'mfma_fp32_TFLOPs': 157.3,
                      'source': '/opt/skills/guides/MI355X_MICROARCH.md; BASELINE.md section 3'},
    'confirmed_on_box': {
        'hbm_copy_read_plus_write_GBps': grab(r'hbm copy[^\n]*?([\d.]+) GB/s'),
        'hbm_fill_GBps': grab(r'hbm fill[^\n]*?([\d.]+) GB/s'),
        'pinned_h2d_GBps': grab(r'pinned H2D[^\n]*?([\d.]+) GB/s'),
        'bf16_gemm_8192cubed_TFLOPs': grab(r'gemm 8192\^3[^\n]*?([\d.]+) TF/s'),
        'bf16_gemm_4096cubed_TFLOPs': grab(r'gemm 4096\^3[^\n]*?([\d.]+) TF/s'),
        'note': 'the core clock sits at ~1.9-2.0 GHz under full MFMA load (2.4 GHz nominal): ~2.1 PFLOP/s is the 100 %-matrix-pipe rate; '
                'the square GEMMs are this library\'s 256 x 256 x 64 kernel (csrc/gemm6.hip)',
    },
    'algorithmic_work': {
        'stft_bytes_per_utterance': '640 000 read (one 16 kHz 10 s channel) + 4 x 1001 x 201 per plane written; round 3 writes power + encoded phase for the noisy '
                                    'channel and power only for the clean one (the clean phase has no consumer): 2 x 640 000 + 3 x 804 804 = 3 694 412 per utterance',
        'istft_bytes_per_utterance': 2249608,
        'encoder_gflop_per_layer_per_utterance': 17.25, 'mhsa_gflop_per_layer_per_utterance': 3.08,
        'qkv_gflop_per_launch_B32': flop['qkv'] / 1e9, 'ffn1_gflop_per_launch_B32': flop['ffn1'] / 1e9, 'mhsa_gflop_per_launch_B32': flop['mhsa'] / 1e9,
    },
    'bench_line': {k: bench[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'dtype')},
    'dominant_kernel': bench.get('roofline'),
    'other_kernels': bench.get('roofline_other_kernels'),
    'per_kernel_rocprof': {},
}
for key, sub, f in (('gemm6p_qkv', 'gemm6p_bf16_kernel<0', flop['qkv']), ('gemm6p_ffn1_gelu', 'gemm6p_bf16_kernel<3', flop['ffn1']),
                    ('mhsa_fwd_prescaled', 'mhsaN_fwd_kernel<8, 4, 1', flop['mhsa'])):
    us = avg_us(sub)
    if us:
        out['per_kernel_rocprof'][key] = {'avg_us': round(us, 1), 'TFLOPs': round(f / us / 1e6, 1), 'frac_of_2500': round(f / us / 1e6 / 2500.0, 3)}
for key, sub, byts in (('stft_two_channels', 'se::stft_kernel', 32 * 3694412.0), ('istft', 'se::istft_kernel', 32 * 2249608.0)):
    us = avg_us(sub)
    if us:
        d = {'avg_us': round(us, 1)}
        if byts:
            d.update(GBps=round(byts / us / 1e3, 1), frac_of_8000=round(byts / us / 1e3 / 8000.0, 3))
        out['per_kernel_rocprof'][key] = d
for k in ('gemm', 'gemmln', 'mhsa', 'stft'):
    try:
        d = json.load(open(P(f'pmc_sq_{k}.json')))
    except OSError:
        continue
    for name, v in d.items():
        if 'se::' not in name or not v.get('GRBM_GUI_ACTIVE'):
            continue
        simd = v['GRBM_GUI_ACTIVE'] * 128.0            # SQ counters cover one XCD: 32 CUs x 4 SIMDs
        out.setdefault('sq_counters', {})[name] = {
            'matrix_pipe_busy': round(v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / simd, 3),
            'vector_issue_busy': round(4.0 * v.get('SQ_ACTIVE_INST_VALU', 0.0) / simd, 3),
            'lds_bank_conflict_frac': round(v.get('lds_bank_conflict_frac', 0.0), 3)}
# ---- the attention forward against its own floors.  Round 4 CORRECTION: the round-3 entry priced ds_read_b128 at 128 B/clk/CU and
# ds_read_b64_tr_b16 at 64 and concluded "feed floor 0.57 of the matrix rate, >= 0.5 of nominal out of reach".  tools/micro/lds_rate.hip
# (profiles/r04_micro_lds_rate.txt) measures 256 B/clk/CU for b128 and 247-250 for tr_b16 (MI355X_MICROARCH.md's LDS table), the unit's 8 + 16
# fragment reads take 64 LDS-array cycles per wave = 256 per CU and unit against 512 matrix cycles, and interleaved with the 16 MFMAs they cost
# 542 (1 wave per SIMD) / 526 (2) cycles per unit: the LDS feed is NOT the floor (0.95-0.97 of the matrix rate).
mh = (bench.get('roofline_other_kernels') or {}).get('mhsa_fwd_kernel') or {}


def _micro(name, pattern, cast=float):
    try:
        t = open(os.path.join(root, 'profiles', name)).read()
    except OSError:
        return None
    m = re.search(pattern, t)
    return cast(m.group(1)) if m else None


skel = {f'{w}_per_simd': _micro('r04_micro_attn_skel.txt', r'QK\^T \| softmax \| PV in program order.*?%d/SIMD:\s+(\d+)' % w) for w in (1, 2, 3, 4)}
skel_p = {f'{w}_per_simd': _micro('r04_micro_attn_skel.txt', r'QK\^T of the next unit issued before.*?%d/SIMD:\s+(\d+)' % w) for w in (1, 2, 3)}
clock_ratio = 1.93 / 2.4                                   # in-kernel clock of the attention forward (profiles/r04_clk_probe.txt: 1.89-1.98 GHz) / the clock of the nominal 2.5 PFLOP/s
best_skel = min(v for v in list(skel.values()) + list(skel_p.values()) if v) if any(skel.values()) else None
out['mhsa_fwd_floor'] = {
    'unit': 'one wave, 32 queries x 64 keys, d = 64: 16 v_mfma_f32_32x32x16_bf16 = 512 matrix clocks on its SIMD',
    'correction_round4': 'the round-3 LDS-feed floor (0.57 of the matrix rate, ">= 0.5 out of reach") was wrong: it assumed 128 / 64 B/clk/CU for '
                         'ds_read_b128 / ds_read_b64_tr_b16; measured 256 / 247-250 (profiles/r04_micro_lds_rate.txt)',
    'lds_feed': {
        'ds_read_b128_B_per_clk_per_CU': _micro('r04_micro_lds_rate.txt', r'ds_read_b128\s+continuous\s+1 waves/SIMD.*?=\s+([\d.]+) B/clk'),
        'ds_read_b64_tr_b16_B_per_clk_per_CU': _micro('r04_micro_lds_rate.txt', r'ds_read_b64_tr_b16\s+continuous\s+1 waves/SIMD.*?=\s+([\d.]+) B/clk'),
        'ds_write_b128_B_per_clk_per_CU': _micro('r04_micro_lds_rate.txt', r'ds_write_b128\s+continuous\s+1 waves/SIMD.*?=\s+([\d.]+) B/clk'),
        'unit_reads_interleaved_with_mfma_cycles_per_unit': {'1_per_simd': _micro('r04_micro_lds_rate.txt', r'reads interleaved with the MFMAs\s+1 waves/SIMD:\s+([\d.]+)'),
                                                             '2_per_simd': _micro('r04_micro_lds_rate.txt', r'reads interleaved with the MFMAs\s+2 waves/SIMD:\s+([\d.]+)')},
        'lds_array_cycles_per_unit_per_CU': 256, 'feed_floor_frac_of_matrix_rate': 0.95,
    },
    'compute_skeleton_cycles_per_unit_at_the_simd': {
        'what': 'tools/micro/attn_skel.hip: the 16 MFMAs + 32 v_exp + 32 adds + 16 packs of a unit with their true dependencies and NO memory traffic',
        'program_order_QK_softmax_PV': skel, 'next_tile_QK_before_this_softmax': skel_p,
        'floor_frac_of_matrix_rate': round(512.0 / best_skel, 3) if best_skel else None,
        'floor_frac_of_nominal': round(512.0 / best_skel * clock_ratio, 3) if best_skel else None,
        'note': 'what bounds head dim 64 is the vector side of the softmax next to the matrix pipe, not LDS bandwidth: a single wave issues a vector instruction '
                'every ~5 cycles (8.6 for v_exp), a SIMD with 3-4 waves one every ~2 (profiles/r03_micro_pk_rate.txt), and next to a busy matrix pipe ~9.5 / 12.8 '
                '(profiles/r03_micro_coissue.txt).  At the 1.89-1.98 GHz the chip holds in this kernel (profiles/r04_clk_probe.txt) >= 0.5 of the 2.4-GHz peak means <= 780-820 '
                'cycles per unit over the WHOLE launch (prologues, tails, idle slots included): the kernel averages ~890 inside a workgroup lifetime and ~1 070 over the launch; '
                'the register-only skeleton needs 600-780.',
    },
    'stamps': 'profiles/r04_mhsa_stamps.txt: one wave of mhsa.hip alone on its SIMD spends 2 205 cycles per tile (stamped build) for 512 matrix + ~430 softmax-issue '
              'cycles: K fragment reads in front of QK^T ~300 exposed, matrix results in front of the first exponential ~250, V^T reads inside PV ~160, '
              'staging + barrier ~230',
    'achieved_frac': mh.get('frac'),
    'achieved_cycles_per_unit_at_1.93GHz': round(mh['avg_launch_ms'] * 1e-3 * 1.93e9 / 192.0) if mh.get('avg_launch_ms') else None,
    'in_kernel_clock_GHz': {'attention forward': '1.89-1.98', 'QKV projection': '1.80-2.00', 'FFN1 + GELU': '1.62-1.78', 'FFN2 / attention output (row-complete)': '1.81-1.83 / 2.02-2.12',
                            'weight gradient': '2.20-2.26', 'source': 'profiles/r04_clk_probe.txt (csrc/clkprobe.h, tools/clk_probe.py)'},
    'measured_restructurings_round4': {
        'mhsa8.hip: 8 waves, SIMD partners alternating matrix / load segments, 4 barriers per tile (the guide\'s two-waves-per-SIMD layout)': '156-172 us: the load segment '
        'carries the whole softmax of ONE wave (~1 200 stamped cycles) while its partner has 256 cycles of MFMA (profiles/r04_mhsa_stamps.txt)',
        'mhsaN<16>: 16 waves free-running on one LDS-DMA staged tile (1 piece per wave and tile), one barrier per tile': '122-125 us',
        'mhsaN<8>: the same with 8 waves, two workgroups per CU': '115 us',
        'mhsaN<8, stagger>: + half of each SIMD\'s waves take the tile barrier mid-tile (DEFAULT from 768 workgroups on)': '107-113 us vs 118-127 us for mhsa.hip on the same boxes',
        'mhsa9.hip: 8 waves, next tile\'s QK^T in front of this tile\'s softmax, K fragments prefetched (256 registers, one workgroup per CU)': '163 us',
        'ablation of mhsaN<16> (profiles/r04_mhsaN_ablation.txt)': 'staging -9, barrier -10, LDS fragment reads -13, exponentials -11 of 124 us; 87 us with all four removed',
        'source': 'profiles/r04_mhsa_variants.txt, profiles/r04_mhsa_bsweep.txt, DESIGN.md section 5d',
    },
}
json.dump(out, open(os.path.join(root, 'roofline.json'), 'w'), indent=1)
print('wrote roofline.json')
