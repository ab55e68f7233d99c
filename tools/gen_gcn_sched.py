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


DW_SETS = ((1, 5), (0, 9), (3, 2, 4, 6), (7, 8, 10))      # plane sets of the weight-gradient kernel (see emit_weight_grad)
DW_MAXNE = 12


def emit_weight_grad(nbr, gidx, Lk, out):
    """Weight-gradient kernel (gcn3_dw_kernel).  dW_k^T[ci][c] = sum over columns of X[ci][col] * V_k[c][col] with
    V_k(v) = sum_j a_j dZ(w_j): one MFMA k-step = the 4 frames of a tile at ONE joint v, so the row list of a
    (plane, joint) unit is wave-uniform and empty units are skipped exactly.  A wave owns a SET of planes and one half
    of the 64 columns c (two 16-column n-tiles, built as packed pairs): 8 waves = 4 sets x 2 halves.  The sets are
    balanced by live units -- (52+49, 53+46, 52+17+14+12, 49+12+13) for the P2RNet skeleton -- and dealt to the waves
    so that the two waves of a SIMD (w, w + 4) carry 175 / 194 / 175 / 194 units.
    Macros: W3_A(set, joint) loads the A operands (X at the joint, 4 m-tiles) of the joint whose steps come next into
    register set `set`; W3_FIRST(ne, o.., c..) builds the first B pair; W3_STEP(aset, plane_slot, ne, o.., c..) =
    gathers of the NEXT step, 8 MFMAs of this one, combine of the next; W3_LAST(aset, plane_slot)."""
    gidx = np.asarray(gidx)
    nbr = np.asarray(nbr)
    V = gidx.shape[1]
    K = len(Lk)
    lofs = np.concatenate([[0], np.cumsum(Lk)])
    length = np.zeros((K, V), dtype=np.int64)
    for k in range(K):
        length[k] = (gidx[lofs[k]:lofs[k + 1]] >= 0).sum(0)
    assert sorted(k for s in DW_SETS for k in s) == list(range(K)), 'DW_SETS must partition the planes'
    assert int(length.max()) <= DW_MAXNE
    loads = [int(sum((length[k] > 0).sum() for k in s)) for s in DW_SETS]
    # wave -> (set, half): waves w and w + 4 share a SIMD; heavy sets are paired with light ones
    order = sorted(range(4), key=lambda i: -loads[i])               # heaviest .. lightest
    pairs = [(order[0], order[3]), (order[3], order[0]), (order[1], order[2]), (order[2], order[1])]
    wave_set = [None] * NW
    for i, (a, b) in enumerate(pairs):
        wave_set[i] = (a, 0)
        wave_set[i + 4] = (b, 1)
    out.append('// ---- weight gradient (row lists): plane sets %s, live units %s; wave -> (set, column half): %s'
               % (DW_SETS, loads, wave_set))
    out.append('#define W3_WAVE_SET {' + ', '.join(str(s) for s, _ in wave_set) + '}')
    out.append('#define W3_WAVE_HALF {' + ', '.join(str(h) for _, h in wave_set) + '}')
    out.append('#define W3_SET_PLANES {' + ', '.join('{' + ', '.join(str(k) for k in (list(s) + [-1] * 4)[:4]) + '}' for s in DW_SETS) + '}')
    out.append('#define W3_MAXPL 4')
    out.append('#define W3_LIGHT_SET %d' % order[3])
    for si, planes in enumerate(DW_SETS):
        steps = []                                     # (joint, plane slot, entries)
        for v in range(V):
            for slot, k in enumerate(planes):
                L = int(length[k, v])
                if L:
                    steps.append((v, slot, [(int(nbr[lofs[k] + e, v]), int((lofs[k] + e) * V + v)) for e in range(L)]))
        lines = []
        joints = []
        for v, _, _ in steps:
            if not joints or joints[-1] != v:
                joints.append(v)
        aset = {v: i & 1 for i, v in enumerate(joints)}
        lines.append('W3_A(%d, %d)' % (aset[joints[0]], joints[0]))
        lines.append('W3_FIRST(%d, %s)' % (len(steps[0][2]), fmt_entries_n(steps[0][2], DW_MAXNE)))
        for u, (v, slot, ent) in enumerate(steps):
            nxt = steps[u + 1] if u + 1 < len(steps) else None
            if nxt is not None and nxt[0] != v:        # the A operands of the next joint land under this step's MFMAs
                lines.append('W3_A(%d, %d)' % (aset[nxt[0]], nxt[0]))
            if nxt is not None:
                lines.append('W3_STEP(%d, %d, %d, %s)' % (aset[v], slot, len(nxt[2]), fmt_entries_n(nxt[2], DW_MAXNE)))
            else:
                lines.append('W3_LAST(%d, %d)' % (aset[v], slot))
        out.append(f'#define W3_BODY_{si} \\')
        out.append(' \\\n'.join('  ' + l for l in lines))
        out.append('')


def fmt_entries_n(ent, n):
    pad = list(ent) + [(0, -1)] * (n - len(ent))
    return ', '.join('%d, %d' % (4 * j, ci) for j, ci in pad)


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
            emit_weight_grad(nbr, gidx, Lk, out)
    path = os.path.join(ROOT, 'pose2room_amd', 'csrc', 'gcn3_sched.inc')
    with open(path, 'w') as f:
        f.write('\n'.join(out) + '\n')
    print('wrote', path, sum(len(l) for l in out), 'bytes')


if __name__ == '__main__':
    main()
