#!/usr/bin/env python
"""Generator of pose2room_amd/csrc/gcn3_sched.inc: the STATIC work schedule of the third-generation graph-conv
kernel (csrc/stgcn_gcn3.hip) for the P2RNet skeleton.

Why generate code.  The second-generation kernel (stgcn_gcn2.hip) walks a per-wave work stream at run time: records
in SGPRs, a slot dispatch chain, scalar loads one record ahead.  Ablations on MI355X (tools/dev_gcn2_exp.py, round 3)
show the MFMAs themselves run at the pipe's peak rate (doubling them adds exactly the roofline time) while
~0.32 ms of a 1.35 ms launch is that per-record control flow, which neither overlaps with the other wave's MFMAs nor
can be scheduled by the compiler.  The adjacency PATTERN (which joints feed which, per plane) is a property of the
skeleton, fixed for the life of the model -- only the coefficient VALUES (A * edge_importance) change.  So the
schedule is resolved here, at build time: per wave a straight-line sequence of steps whose LDS offsets, accumulator
slots and coefficient-table indices are immediates.

Output: for FORM 0 (column lists: forward) and FORM 1 (row lists: data gradient), per wave w
    #define G3_BODY_<form>_<w>   G3_FIRST(...) G3_VISIT(...) G3_STEP(...) ... G3_LAST(...)
plus the slot -> joint table, the per-form signature the host checks against the run-time tables, and counts.

    python tools/gen_gcn_sched.py            # rewrites pose2room_amd/csrc/gcn3_sched.inc
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NW, SLOTS, CHUNK, PIECES_PER_WAVE = 8, 7, 6, 7


def fnv1a(values):
    h = 0xcbf29ce484222325
    for v in values:
        for b in int(v & 0xffffffff).to_bytes(4, 'little'):
            h ^= b
            h = (h * 0x100000001b3) & 0xffffffffffffffff
    return h


def pattern_signature(nbr, gidx, Lk):
    """64-bit FNV-1a over (K, V, per plane: list length, per joint: real length and (source joint, table index) of
    every entry) -- identical to gcn_tables.pattern_signature (the host side of the check)."""
    from pose2room_amd.p2rnet import gcn_tables
    return gcn_tables.pattern_signature(nbr, gidx, Lk)


def schedule(nbr, gidx, Lk):
    """Per wave: (slot joints, list of visits); a visit = (plane, [(slot, [(joint, table index), ...]), ...]) with the
    entries of a list longer than CHUNK split over consecutive steps of the same slot."""
    from pose2room_amd.p2rnet import gcn_tables
    gidx = np.asarray(gidx)
    nbr = np.asarray(nbr)
    V = gidx.shape[1]
    K = len(Lk)
    lofs = np.concatenate([[0], np.cumsum(Lk)])
    length = np.zeros((K, V), dtype=np.int64)
    for k in range(K):
        length[k] = (gidx[lofs[k]:lofs[k + 1]] >= 0).sum(0)
    steps_per_joint = np.ceil(length / float(CHUNK)).astype(np.int64).sum(0)
    owner = gcn_tables.deal_runs(steps_per_joint, NW, SLOTS)
    waves = []
    for w in range(NW):
        visits = []
        for k in range(K):
            steps = []
            for slot, j in enumerate(owner[w]):
                if j < 0 or length[k, j] == 0:
                    continue
                ent = [(int(nbr[lofs[k] + e, j]), int((lofs[k] + e) * V + j)) for e in range(int(length[k, j]))]
                for c in range(0, len(ent), CHUNK):
                    steps.append((slot, ent[c:c + CHUNK]))
            if steps:
                visits.append((k, steps))
        waves.append((owner[w], visits))
    return waves


def emit_form(form, nbr, gidx, Lk, out):
    waves = schedule(nbr, gidx, Lk)
    V = np.asarray(gidx).shape[1]
    n_steps = [sum(len(s) for _, s in v) for _, v in waves]
    n_units = sum(n_steps)
    out.append(f'// ---- FORM {form}: {"column lists (forward)" if form == 0 else "row lists (data gradient)"}: '
               f'{n_units} steps per 16-frame tile and channel phase; per wave {n_steps}')
    out.append(f'#define G3_SIGNATURE_{form} 0x{pattern_signature(nbr, gidx, Lk):016x}ull')
    out.append(f'#define G3_STEPS_{form} {n_units}')
    sj = ', '.join('{' + ', '.join(str(j) for j in o) + '}' for o, _ in waves)
    out.append(f'#define G3_SLOT_JOINTS_{form} {{{sj}}}')
    out.append(f'#define G3_PLANE0_{form} {{' + ', '.join(str(v[0][0]) for _, v in waves) + '}')
    for w, (owner, visits) in enumerate(waves):
        flat = [(k, slot, ent) for k, steps in visits for slot, ent in steps]
        lines = []
        nvis = len(visits)
        # DMA pieces of the next slice are spread over the first visits; A operands ping-pong between two register
        # sets (visit v uses set v & 1 and prefetches the planes of visit v + 1 into the other one)
        first = flat[0]
        lines.append('G3_FIRST(%d, %s)' % (len(first[2]), fmt_entries(first[2])))
        u = 0
        for vi, (k, steps) in enumerate(visits):
            wrap = vi + 1 >= nvis
            nk = visits[0][0] if wrap else visits[vi + 1][0]
            piece = vi if vi < PIECES_PER_WAVE else -1
            lines.append('G3_VISIT(%d, %d, %d, %d, %d)' % (vi & 1, k, nk, int(wrap), piece))
            for slot, ent in steps:
                if u + 1 < len(flat):
                    nxt = flat[u + 1][2]
                    lines.append('G3_STEP(%d, %d, %d, %s)' % (vi & 1, slot, len(nxt), fmt_entries(nxt)))
                else:
                    lines.append('G3_LAST(%d, %d)' % (vi & 1, slot))
                u += 1
        # pieces that did not find a visit (fewer visits than pieces) are issued by the kernel after the body
        lines.append('G3_END(%d, %d, %d)' % (nvis & 1, min(nvis, PIECES_PER_WAVE), visits[0][0]))
        out.append(f'#define G3_BODY_{form}_{w} \\')
        out.append(' \\\n'.join('  ' + l for l in lines))
        out.append('')
    return n_units


def emit_coef_grad(nbr, gidx, Lk, out):
    """Adjacency-gradient kernel (gcn3_dcoef_kernel): the row-list schedule with every step carrying its OWN entries
    (the step's MFMAs produce Y_k of one (plane, joint) unit, reduced at once against the gathered rows)."""
    waves = schedule(nbr, gidx, Lk)
    out.append('// ---- adjacency gradient (row lists): D3_VISIT(set, plane, next, wrap, piece), '
               'D3_STEP(set, slot, ne, o0,c0, .. o5,c5), D3_CONT(ne, o0,c0, ..), D3_END(parity, pieces)')
    for w, (owner, visits) in enumerate(waves):
        lines = []
        nvis = len(visits)
        for vi, (k, steps) in enumerate(visits):
            wrap = vi + 1 >= nvis
            nk = visits[0][0] if wrap else visits[vi + 1][0]
            piece = vi if vi < PIECES_PER_WAVE else -1
            lines.append('D3_VISIT(%d, %d, %d, %d, %d)' % (vi & 1, k, nk, int(wrap), piece))
            prev = None
            for slot, ent in steps:
                if slot == prev:      # rest of a list longer than six entries: same product, no new MFMAs
                    lines.append('D3_CONT(%d, %s)' % (len(ent), fmt_entries(ent)))
                else:
                    lines.append('D3_STEP(%d, %d, %d, %s)' % (vi & 1, slot, len(ent), fmt_entries(ent)))
                prev = slot
        lines.append('D3_END(%d, %d)' % (nvis & 1, min(nvis, PIECES_PER_WAVE)))
        out.append(f'#define D3_BODY_{w} \\')
        out.append(' \\\n'.join('  ' + l for l in lines))
        out.append('')


def fmt_entries(ent):
    """Six (LDS byte offset of the source joint, coefficient-table index) pairs, padded with (0, -1)."""
    pad = list(ent) + [(0, -1)] * (CHUNK - len(ent))
    return ', '.join('%d, %d' % (4 * j, ci) for j, ci in pad)


def main():
    from pose2room_amd.p2rnet import gcn_tables
    from pose2room_amd.p2rnet.modules.stgcn_layers import Graph
    A = Graph().A
    K, V = A.shape[0], A.shape[1]
    out = ['// GENERATED by tools/gen_gcn_sched.py from the P2RNet skeleton (stgcn_layers.Graph: %d planes, %d joints, %d '
           'non-zeros) -- do not edit.' % (K, V, int((A != 0).sum())),
           '// Static work schedule of csrc/stgcn_gcn3.hip; macro vocabulary:',
           '//   G3_FIRST(ne, o0,c0, .. o5,c5)        build the B operand of the first step (ne entries: LDS byte offset of',
           '//                                         the source joint, index into the coefficient table; padding 0,-1)',
           '//   G3_VISIT(set, plane, next, wrap, piece)  plane switch: steps use A-operand set `set`; prefetch plane `next`',
           '//                                         (wrap: of the next phase) into the other set; issue DMA piece (-1: none)',
           '//   G3_STEP(set, slot, ne, o0,c0, ..)    gather the NEXT step\'s entries, 16 MFMAs into accumulator `slot`,',
           '//                                         combine the next step\'s B operand',
           '//   G3_LAST(set, slot)                   final step of the phase',
           '//   G3_END(parity, pieces, plane0)       parity of the visit count, DMA pieces already issued, first plane',
           f'#define G3_K {K}', f'#define G3_V {V}', '']
    for form, tr in ((0, False), (1, True)):
        nbr, gidx, Lk = gcn_tables.build(A, transpose=tr)
        emit_form(form, nbr, gidx, Lk, out)
        if form == 1:
            emit_coef_grad(nbr, gidx, Lk, out)
    path = os.path.join(ROOT, 'pose2room_amd', 'csrc', 'gcn3_sched.inc')
    with open(path, 'w') as f:
        f.write('\n'.join(out) + '\n')
    print('wrote', path, sum(len(l) for l in out), 'bytes')


if __name__ == '__main__':
    main()
