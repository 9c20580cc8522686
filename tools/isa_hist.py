"""Developer tool: static instruction histogram of one kernel of a gfx950 .s file (tools/isa.sh <file>.hip), by what the instruction is FOR.
    python tools/isa_hist.py /tmp/stft.s stft_kernel [more kernel-name substrings ...]
Static counts (the code as emitted, loops counted once): the STFT / iSTFT kernels are almost fully unrolled -- their only loops are the pass-A item
loop, the boundary-frame fill, the write-out and the mel taps -- so static ~ per-thread dynamic for an interior workgroup."""
import collections
import re
import sys

CATS = [
    ('matrix', r'^v_mfma|^v_smfma'),
    ('transcendental (rcp/rsq/sqrt/exp/log/sin/cos)', r'^v_(rcp|rsq|sqrt|exp|log|sin|cos)_'),
    ('fp32 fma/mul/add/sub/mac (the arithmetic)', r'^v_(fma|fmac|mul|add|sub|subrev|mac|mad|pk_fma|pk_mul|pk_add)_(f32|f16|legacy_f32)'),
    ('fp min/max/med/cmp/cndmask/sign tricks', r'^v_(min|max|med3|cmp|cmpx|cndmask|bfi|and|or|xor|not)_|^v_cmp'),
    ('conversions / packs', r'^v_cvt|^v_perm|^v_pack'),
    ('integer + address arithmetic', r'^v_(add|sub|subrev|mul|mad|lshl|lshr|ashr|lshlrev|lshrrev|ashrrev|add3|lshl_add|add_lshl|mul_lo|mul_hi|mul_u32|bfe|mbcnt|readfirstlane|readlane|writelane)_?(u|i|co|nc|b)?'),
    ('moves', r'^v_mov|^v_accvgpr|^v_swap|^v_nop'),
    ('LDS read', r'^ds_read|^ds_load'),
    ('LDS write', r'^ds_write|^ds_store'),
    ('LDS other (bpermute, atomics)', r'^ds_'),
    ('global / buffer load', r'^(global|buffer|flat|scratch)_load'),
    ('global / buffer store + atomics', r'^(global|buffer|flat|scratch)_(store|atomic)'),
    ('s_waitcnt', r'^s_waitcnt'),
    ('s_barrier', r'^s_barrier'),
    ('branches', r'^s_cbranch|^s_branch'),
    ('scalar ALU / moves / compares', r'^s_'),
]


def kernel_lines(path, key):
    out, on = [], False
    for ln in open(path):
        t = ln.strip()
        if not on:
            if re.match(r'^_Z\w*%s\w*:' % re.escape(key), t):
                on = True
            continue
        if t.startswith('s_endpgm'):
            break
        t = t.split(';')[0].strip()
        if not t or t.startswith('.') or t.endswith(':'):
            continue
        out.append(t.split()[0])
    return out


def main():
    path = sys.argv[1]
    for key in sys.argv[2:]:
        ins = kernel_lines(path, key)
        hist = collections.Counter()
        other = collections.Counter()
        for op in ins:
            for name, pat in CATS:
                if re.match(pat, op):
                    hist[name] += 1
                    break
            else:
                hist['other'] += 1
                other[op] += 1
        vec = sum(v for k, v in hist.items() if k.split()[0] in ('matrix', 'transcendental', 'fp32', 'fp', 'conversions', 'integer', 'moves'))
        print(f'== {key}: {len(ins)} instructions, {vec} of them vector-ALU')
        for name, _ in CATS + [('other', '')]:
            if hist[name]:
                print(f'   {hist[name]:6d}  {100.0 * hist[name] / len(ins):5.1f} %   {name}')
        if other:
            print('   other:', dict(other.most_common(8)))


if __name__ == '__main__':
    main()
