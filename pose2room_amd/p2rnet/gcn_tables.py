"""Sparse neighbour tables of the skeleton adjacency for the fused graph-conv kernels.

For plane k of A (K,V,V) and output column w, the kernel needs the list of source
joints v with A[k,v,w] != 0 (column lists, forward) -- and for the data gradient the
row lists (for source row v, the w with A[k,v,w] != 0).  Lists are padded per plane to
the longest list of that plane; padded slots point at joint 0 with coefficient 0.

  nbr   uint8 [sum_k L_k, V]   joint index of the j-th neighbour
  gidx  int64 [sum_k L_k, V]   flat index into (K*V*V) of that entry, -1 when padded
  Lk    list[int]              list length per plane
`coefficients(Aeff, gidx)` gathers the current A*importance values (differentiable).
"""
import numpy as np
import torch


def build(A, transpose=False):
    A = np.asarray(A)
    K, V, _ = A.shape
    nz = A != 0
    Lk, nbr_rows, gidx_rows = [], [], []
    for k in range(K):
        lists = []
        for w in range(V):
            src = np.nonzero(nz[k, w, :])[0] if transpose else np.nonzero(nz[k, :, w])[0]
            lists.append(src)
        L = max(1, max(len(s) for s in lists))
        Lk.append(L)
        nb = np.zeros((L, V), dtype=np.uint8)
        gi = -np.ones((L, V), dtype=np.int64)
        for w, src in enumerate(lists):
            for j, v in enumerate(src):
                nb[j, w] = v
                # transpose: output column is the source row v_out = w, neighbour = destination joint
                gi[j, w] = (k * V + w) * V + v if transpose else (k * V + v) * V + w
        nbr_rows.append(nb)
        gidx_rows.append(gi)
    return torch.from_numpy(np.concatenate(nbr_rows)), torch.from_numpy(np.concatenate(gidx_rows)), Lk


def coefficients(Aeff, gidx):
    """Aeff (K,V,V) -> coef [sum L_k, V] with zeros at padded slots (differentiable)."""
    flat = Aeff.reshape(-1)
    safe = gidx.clamp(min=0)
    # index_select (backward: one index_add_) instead of advanced indexing (backward: a sort-based index_put_)
    vals = flat.index_select(0, safe.reshape(-1)).view_as(gidx)
    return torch.where(gidx >= 0, vals, torch.zeros((), dtype=Aeff.dtype, device=Aeff.device))


STREAM_UMAX, STREAM_REC, STREAM_HDR = 80, 12, 16     # csrc/stgcn_gcn2.hip: G2_UMAX, G2_REC, G2_HDR


class StreamBudgetError(ValueError):
    """The adjacency pattern does not fit the static work stream of the second-generation kernels."""


def deal_runs(cost, n_waves, slots):
    """Joints -> (wave, slot) such that a wave's slots 0-3 hold one run of CONSECUTIVE joints and its slots 4-6 a
    second one (the kernel stores the values of a run with one 16- or 12-byte store per lane).

    The joints are cut into 2 * n_waves consecutive runs of three or four joints; every ordering of the run lengths
    is tried, the runs are paired into waves (a four-run with a three-run) and the waves onto SIMDs (waves i and
    i + 4 share one) heaviest with lightest; the split with the lightest busiest SIMD wins, ties broken by the
    lightest busiest wave.  Returns per wave the list of 7 slot joints (-1 = unused slot)."""
    import itertools
    V = len(cost)
    if not (n_waves == 8 and slots == 7 and 3 * 16 <= V <= 4 * 8 + 3 * 8):
        raise StreamBudgetError('joint count %d outside 48..56' % V)
    n4 = V - 3 * 16                                   # runs of four (the other 16 - n4 runs have three joints)
    best = None
    for pos in itertools.combinations(range(16), n4):
        lens = [4 if i in pos else 3 for i in range(16)]
        if sum(1 for l in lens if l == 4) > 8:        # a wave has one 4-slot group only
            continue
        start, runs = 0, []
        for l in lens:
            runs.append((int(sum(cost[start:start + l])), start, l))
            start += l
        # slot group A takes the runs of four plus the heaviest runs of three, group B the rest
        fours = sorted([r for r in runs if r[2] == 4])
        threes = sorted([r for r in runs if r[2] == 3])
        ga = sorted(fours + threes[len(threes) - (8 - len(fours)):])
        gb = sorted(threes[:len(threes) - (8 - len(fours))])
        waves = sorted(((ga[i][0] + gb[7 - i][0], ga[i], gb[7 - i]) for i in range(8)), key=lambda t: t[0])
        simd = max(waves[i][0] + waves[7 - i][0] for i in range(4))
        key = (simd, waves[7][0])
        if best is None or key < best[0]:
            best = (key, waves)
    waves = best[1]
    owner = [None] * n_waves
    for i in range(4):
        for widx, (_, ra, rb) in ((i, waves[7 - i]), (i + 4, waves[i])):
            sl = [-1] * 7
            for j in range(ra[2]):
                sl[j] = ra[1] + j
            for j in range(rb[2]):
                sl[4 + j] = rb[1] + j
            owner[widx] = sl
    return owner


def build_stream(nbr, gidx, Lk, n_waves=8, slots=7, joint_stride=1):
    """Static per-wave work stream of the second-generation graph-conv kernel (csrc/stgcn_gcn2.hip).

    A unit is (plane k, output joint w) with a non-empty neighbour list; a wave walks its units plane by plane.
    One 12-int record per pass (lists longer than six entries take two passes):
      [0]     slot | first-of-visit << 3 | (at most 2 entries) << 4 | plane << 8 | plane of the wave's next visit << 12
              (15: wrap)
      [1..3]  six 16-bit byte offsets  joint * joint_stride * 4  of the source joints
      [4..9]  flat index into the coefficient table [ltot * V] per entry, -1 for padding (the kernel puts the
              current coefficient there)
    followed by a 16-int header: [0] records, [1] first plane, [2..8] joint of slot i or -1, [9] visits.
    A visit is one pass over the slots in ascending order for one plane; long lists continue in an extra visit.
    Returns (int32 [n_waves, 80 * 12 + 16], busiest SIMD's units, all units).
    """
    gidx = np.asarray(gidx)
    nbr = np.asarray(nbr)
    V = gidx.shape[1]
    K = len(Lk)
    if not (K < 15 and slots <= 7 and V * joint_stride * 4 < 65536):
        raise StreamBudgetError('K = %d, V = %d' % (K, V))
    lofs = np.concatenate([[0], np.cumsum(Lk)])
    length = np.zeros((K, V), dtype=np.int64)           # real list length per (plane, joint)
    for k in range(K):
        length[k] = (gidx[lofs[k]:lofs[k + 1]] >= 0).sum(0)
    cost = (length > 0).sum(0)
    rec_cost = np.ceil(length / 6.0).astype(np.int64).sum(0)      # records per joint (lists > 6 entries: several)
    owner = deal_runs(rec_cost, n_waves, slots)
    load = np.array([sum(int(rec_cost[w]) for w in o if w >= 0) for o in owner])
    simd_load = load.reshape(n_waves // 4, 4).sum(0)
    out = np.zeros((n_waves, STREAM_UMAX * STREAM_REC + STREAM_HDR), dtype=np.int32)
    for i in range(n_waves):
        recs = []                                         # (plane, slot, first-of-visit, [(joint, table index)])
        visits = 0
        for k in range(K):
            chunks = {}                                   # slot -> list of <= 6-entry chunks
            for slot, w in enumerate(owner[i]):
                if w < 0:
                    continue
                L = int(length[k, w])
                ent = []
                for j in range(L):
                    row = int(lofs[k]) + j
                    assert gidx[row, w] >= 0
                    ent.append((int(nbr[row, w]), row * V + w))       # (source joint, coefficient table index)
                if L:
                    chunks[slot] = [ent[c:c + 6] for c in range(0, L, 6)]
            depth = max((len(c) for c in chunks.values()), default=0)
            for pass_i in range(depth):                   # a visit = one pass over the slots (ascending) of plane k
                first = True
                for slot in sorted(chunks):
                    if pass_i < len(chunks[slot]):
                        recs.append((k, slot, first, chunks[slot][pass_i]))
                        first = False
                visits += 1
        if len(recs) + 2 > STREAM_UMAX:
            raise StreamBudgetError('%d records for one wave (budget %d)' % (len(recs), STREAM_UMAX - 2))
        for u, (k, slot, first, ch) in enumerate(recs):
            later = [kk for (kk, _, f, _) in recs[u + 1:] if f]
            nk = later[0] if later else 15
            base = u * STREAM_REC
            out[i, base] = slot | (int(first) << 3) | (int(len(ch) <= 2) << 4) | (k << 8) | (nk << 12)
            out[i, base + 4:base + 10] = -1
            offs = [0] * 6
            for j, (v, flat) in enumerate(ch):
                out[i, base + 4 + j] = flat
                offs[j] = v * joint_stride * 4
            for q in range(3):
                word = offs[2 * q] | (offs[2 * q + 1] << 16)
                out[i, base + 1 + q] = word - (1 << 32) if word >= (1 << 31) else word
        hdr = STREAM_UMAX * STREAM_REC
        out[i, hdr] = len(recs)
        out[i, hdr + 1] = recs[0][0] if recs else 0
        out[i, hdr + 9] = visits
        for slot, w in enumerate(owner[i]):
            out[i, hdr + 2 + slot] = w
    return out, owner, int(simd_load.max()), int(load.sum())
