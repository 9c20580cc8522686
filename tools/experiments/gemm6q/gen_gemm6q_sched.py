"""Generates csrc/gemm6q_sched.h: where the pieces of the pipelined epilogue of gemm6q.hip sit in the last K-tile of an output tile (tag L) and in
the first K-tile of the next one (tag N), and the hand-counted waits that follow from it.

A wave's 128 x 64 output is four 64 x 32 quadrants (mq, nq) that a K-tile's phases visit as (0,0) (0,1) (1,1) (1,0).  In the last K-tile quadrant
q is final after its phase's MFMAs and is not written again before the same phase of the next tile's first K-tile: 13 sections lie between -- the
read section (R) and the MFMA section (M) of the phases that follow.  A quadrant's pieces are
    ACT b (b = 0..7): the activation of one MFMA tile (row tile i = b >> 1, column tile j = b & 1), in place   [GELU kernels only]
    GH i  (i = 0..3): bf16 conversion of one 16-row tile pair + ONE store of 16 rows x 64 B
and run over the quadrant's first seven sections as  R: ACT 0 1 | M: ACT 2 3 | R: ACT 4 5 | M: ACT 6 7 | R: GH 0 1 | M: GH 2 | R: GH 3.
An R section takes its pieces after its LDS-DMA issue (they run while the fragment reads are in flight and beside the SIMD partner's MFMAs); an M
section has eight hook points, one after every second MFMA, and spreads its pieces over them.

Waits (vmcnt retires in order; a K-tile's wait sits at the end of its phase-4 read section and must cover everything up to B_0 of the next K-tile,
which is issued in phase 1's read section BEFORE that section's pieces):  vmcnt(6 + stores issued since then).
"""
import os

PH_ORDER = [(0, 0), (0, 1), (1, 1), (1, 0)]          # quadrant of phase 1..4
# global sections: (tag, phase, kind)
SECTIONS = []
for tag in 'LN':
    for ph in range(1, 5):
        SECTIONS.append((tag, ph, 'R'))
        SECTIONS.append((tag, ph, 'M'))
QUAD_PLAN = [['ACT 0', 'ACT 1'], ['ACT 2', 'ACT 3'], ['ACT 4', 'ACT 5'], ['ACT 6', 'ACT 7'], ['GH 0', 'GH 1'], ['GH 2'], ['GH 3']]
# whole-line variant (SE6Q_WL=1): GW i = one 16-row group of BOTH column halves (needs quadrants (mq, 0) and (mq, 1) final): conversion, lane ^ 8
# exchange, two stores of 8 rows x 128 B.  Everything in READ sections (vector work between a wave's own MFMAs delays them: profiles/r05_gemm6q_steps.txt).
WL_PLAN = {
    ('L', 2, 'R'): ['ACT 0 0 0', 'ACT 0 0 1', 'ACT 0 0 2', 'ACT 0 0 3', 'ACT 0 0 4', 'ACT 0 0 5', 'ACT 0 0 6', 'ACT 0 0 7'],
    ('L', 3, 'R'): ['ACT 0 1 0', 'ACT 0 1 1', 'ACT 0 1 2', 'ACT 0 1 3', 'ACT 0 1 4', 'ACT 0 1 5', 'ACT 0 1 6', 'ACT 0 1 7', 'GW 0 0'],
    ('L', 4, 'R'): ['ACT 1 1 0', 'ACT 1 1 1', 'ACT 1 1 2', 'ACT 1 1 3', 'ACT 1 1 4', 'ACT 1 1 5', 'ACT 1 1 6', 'ACT 1 1 7', 'GW 0 1', 'GW 0 2'],
    ('N', 1, 'R'): ['ACT 1 0 0', 'ACT 1 0 1', 'ACT 1 0 2', 'ACT 1 0 3', 'ACT 1 0 4', 'ACT 1 0 5', 'ACT 1 0 6', 'ACT 1 0 7', 'GW 0 3', 'GW 1 0'],
    ('N', 2, 'R'): ['GW 1 1', 'GW 1 2'],
    ('N', 3, 'R'): ['GW 1 3'],
}


def build():
    sec = {s: [] for s in SECTIONS}
    for ph, (mq, nq) in enumerate(PH_ORDER, start=1):
        start = SECTIONS.index(('L', ph, 'M')) + 1           # first section after the phase that finishes the quadrant
        last = SECTIONS.index(('N', ph, 'R'))                 # the quadrant is rewritten by ('N', ph, 'M')
        assert last - start + 1 >= len(QUAD_PLAN)
        for k, pieces in enumerate(QUAD_PLAN):
            for p in pieces:
                kind, idx = p.split()
                sec[SECTIONS[start + k]].append((kind, mq, nq, int(idx)))
    return sec


def build_wl():
    sec = {s: [] for s in SECTIONS}
    final_after = {q: SECTIONS.index(('L', ph, 'M')) for ph, q in enumerate(PH_ORDER, start=1)}
    rewritten_at = {q: SECTIONS.index(('N', ph, 'M')) for ph, q in enumerate(PH_ORDER, start=1)}
    for s, pieces in WL_PLAN.items():
        at = SECTIONS.index(s)
        for p in pieces:
            f = p.split()
            if f[0] == 'ACT':
                mq, nq, b = int(f[1]), int(f[2]), int(f[3])
                assert final_after[(mq, nq)] < at < rewritten_at[(mq, nq)], p
                sec[s].append(('ACT', mq, nq, b))
            else:
                mq, i = int(f[1]), int(f[2])
                for nq in (0, 1):
                    assert final_after[(mq, nq)] < at < rewritten_at[(mq, nq)], p
                    # the activation of both halves' tiles of this row group must be in front of it
                    for b in (2 * i, 2 * i + 1):
                        done = [SECTIONS.index(t) for t, ps in sec.items() if ('ACT', mq, nq, b) in ps]
                        assert done and (done[0] < at or (done[0] == at)), (p, nq, b)
                sec[s].append(('GW', mq, 0, i))
    return sec


def piece_src(p):
    kind, mq, nq, idx = p
    if kind == 'GW':
        return f'SEQ_GW({mq}, {idx})'
    return f'SEQ_ACT({mq}, {nq}, {idx}, {idx + 1})' if kind == 'ACT' else f'SEQ_GH({mq}, {nq}, {idx})'


def emit(sec, out, nstores):
    for tag in 'PLN':
        for ph in range(1, 5):
            pieces = [] if tag == 'P' else sec[(tag, ph, 'R')]
            out.append(f'#define SEQ_{tag}_P{ph}R ' + ' SEQ_SB '.join(piece_src(p) for p in pieces))
            pieces = [] if tag == 'P' else sec[(tag, ph, 'M')]
            n = len(pieces)
            pts = [[] for _ in range(8)]
            for i, p in enumerate(pieces):
                pts[(i * 8) // max(n, 1)].append(p)
            for k in range(8):
                out.append(f'#define SEQ_{tag}_P{ph}M{k} ' + ' SEQ_SB '.join(piece_src(p) for p in pts[k]))

    def stores(s):
        return sum(nstores[p[0]] for p in sec[s])
    wl = sum(stores(s) for s in SECTIONS[SECTIONS.index(('L', 1, 'R')):SECTIONS.index(('L', 4, 'R')) + 1])
    wn = sum(stores(s) for s in SECTIONS[SECTIONS.index(('N', 1, 'R')):SECTIONS.index(('N', 4, 'R')) + 1])
    assert stores(('N', 4, 'M')) == 0, 'pieces behind the N wait would have to be counted into the next K-tile\'s wait'
    assert sum(stores(s) for s in SECTIONS) == 16
    out += ['', f'#define SEQ_WAIT_L {6 + wl}      // 6 + the {wl} stores issued between B_0 of the next tile\'s K-tile 0 and the wait',
            f'#define SEQ_WAIT_N {6 + wn}     // 6 + the {wn} stores issued between B_0 of K-tile 1 and the wait', '']
    return 6 + wl, 6 + wn


def main():
    out = ['// generated by tools/gen_gemm6q_sched.py -- do not edit.  Placement of the epilogue pieces of gemm6q.hip (tag L: last K-tile of an output tile,',
           '// tag N: first K-tile of the next one, tag P: every other K-tile) and the store counts behind its hand-counted waits.', '#pragma once', '',
           '#if SE6Q_WL       // whole-line stores (8 rows x 128 B), every piece in a read section']
    sec_wl = build_wl()
    w_wl = emit(sec_wl, out, {'ACT': 0, 'GW': 2, 'GH': 1})
    out.append('#else              // 16 rows x 64 B stores, pieces in read and MFMA sections')
    sec = build()
    w = emit(sec, out, {'ACT': 0, 'GW': 2, 'GH': 1})
    out.append('#endif')
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'gemm6q_sched.h')
    with open(path, 'w') as f:
        f.write('\n'.join(out) + '\n')
    for name, sc in (('whole-line', sec_wl), ('half-line', sec)):
        print(name)
        for s in SECTIONS:
            print(' ', s, [f'{p[0]}{p[1]}{p[2]}.{p[3]}' for p in sc[s]])
    print('waits whole-line', w_wl, 'half-line', w)


if __name__ == '__main__':
    main()
