"""Generate the golden fixtures under tests/golden/ (run in the BUILD container only).

    python tests/golden/make_golden.py [g1 g2 g6 g3 g4 g5 ...]

The reference Python (/root/reference, read-only) is imported here -- and only
here -- to produce input/output vectors; it never travels to the GPU box.  The
fixtures are data: seeded inputs (or the seeds to rebuild them through
tests/cases.py) and the reference's outputs.

  G1  nn_distance: the reference's own demo inputs (net_utils/nn_distance.py:63-94,
      np.random.seed(0)) and the three loss call shapes, all modes, with the
      autograd gradients of a fixed scalarisation.
  G2  nms_2d_faster / nms_3d_faster / nms_3d_faster_samecls pick lists on seeded boxes.
  G6  the nine _ext ops: outputs of the CPU restatement (oracle/p2r_oracle.c) on
      seeded random and adversarial clouds.  The reference has no CPU path and
      no tests for these, so G6 pins the *restatement* against regressions; it
      is not reference-generated ("parity unpinned").
  G6b furthest_point_sampling / gather_points against the reference's OWN pure-torch
      `farthest_point_sample` / `index_points` (net_utils/libs.py:152-190; random start
      patched to index 0) on the tie-free, origin-free clouds: reference-generated
      vectors for a1 / a2 (the CUDA kernel adds the |p|^2 <= 1e-3 skip and the block
      tie rule, which these clouds do not exercise; margins recorded).  Also: group_points forward
      from `index_points` with a 3-D index, three_nn(x, x) index sets from `vn_dgcnn_util.knn`
      (vn_dgcnn_util.py:4-10).
  G3-G5 are produced by tests/golden/make_model_golden.py (full model paths).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from tests import cases  # noqa: E402
from oracle import cpu_ext  # noqa: E402


def _ref_path():
    sys.path.insert(0, "/root/reference")


def g1():
    _ref_path()
    from net_utils.nn_distance import nn_distance
    out = {}
    # (a) the reference demo's own inputs
    np.random.seed(0)
    pc1 = np.random.random((1, 5, 3)).astype(np.float32)
    pc2 = np.random.random((1, 6, 3)).astype(np.float32)
    sets = {"demo": (torch.from_numpy(pc1), torch.from_numpy(pc2))}
    # (b) the loss call shapes (models/loss.py:105,127-131,64)
    g = torch.Generator().manual_seed(11)
    sets["vote"] = (torch.randn(40, 3, 3, generator=g), torch.randn(40, 53, 3, generator=g))
    sets["assign"] = (torch.randn(1, 128, 3, generator=g), torch.randn(1, 4, 3, generator=g))
    c = torch.randn(3, 10, 3, generator=g)
    c[:, 6:] = 0.0  # zero-padded GT rows take part (loss.py:64)
    sets["center"] = (torch.randn(3, 128, 3, generator=g), c)
    for name, (a, q) in sets.items():
        out[f"{name}_pc1"] = a.numpy()
        out[f"{name}_pc2"] = q.numpy()
        gg = torch.Generator().manual_seed(5)
        w1 = torch.randn(a.shape[0], a.shape[1], generator=gg)
        w2 = torch.randn(q.shape[0], q.shape[1], generator=gg)
        out[f"{name}_w1"] = w1.numpy()
        out[f"{name}_w2"] = w2.numpy()
        for mode, kw in {"l2": {}, "l1smooth": {"l1smooth": True}, "l1": {"l1": True}}.items():
            ar = a.clone().requires_grad_(True)
            qr = q.clone().requires_grad_(True)
            d1, i1, d2, i2 = nn_distance(ar, qr, **kw)
            ((d1 * w1).sum() + (d2 * w2).sum()).backward()
            for k, v in dict(dist1=d1, idx1=i1, dist2=d2, idx2=i2, grad1=ar.grad, grad2=qr.grad).items():
                out[f"{name}_{mode}_{k}"] = v.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "g1_nn_distance.npz"), **out)
    print("g1", len(out), "arrays")


def g2():
    _ref_path()
    from net_utils.nms import nms_2d_faster, nms_3d_faster, nms_3d_faster_samecls
    out = {}
    for K in (1, 2, 16, 128, 300):
        boxes = cases.random_boxes(K, seed=K)
        out[f"boxes_{K}"] = boxes
        for thr in (0.10, 0.25):
            for old in (False, True):
                tag = f"{K}_{int(thr * 100)}_{int(old)}"
                out[f"pick_{tag}"] = np.asarray(nms_3d_faster(boxes[:, :7], thr, old), dtype=np.int32)
                out[f"pickcls_{tag}"] = np.asarray(nms_3d_faster_samecls(boxes, thr, old), dtype=np.int32)
                # nms_2d_faster (nms.py:7-39) on the (x, z) extents + score of the same boxes -- what
                # ap_helper.py:198-214 builds when use_3d_nms is false
                out[f"pick2d_{tag}"] = np.asarray(nms_2d_faster(cases.boxes_xz(boxes), thr, old), dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "g2_nms.npz"), **out)
    print("g2", len(out), "arrays")


# ---- decision margins of the CUDA-only `_ext` kernels (SURVEY 8c hazard (i)) -------------------------------------
# The oracle and the HIP kernels evaluate (dx*dx + dy*dy) + dz*dz in source order without contraction; nvcc's default
# -fmad=true may contract it to fma(dz, dz, fma(dy, dy, dx*dx)) or fma(dz, dz, fma(dx, dx, dy*dy)).  For every
# random-cloud fixture the three arithmetic variants are replayed in numpy: the fixture records the smallest
# relative gap at any decision (in float32 ulps) and whether both contracted variants reproduce the very same
# indices.  Lattice / duplicate clouds have exact ties by construction and are excluded (their distances are exact
# in all three variants).
def _f32(x):
    return np.asarray(x, dtype=np.float32)


def _sqdist(p, q, variant):
    """p (..., 3), q (..., 3) float32 -> squared distance float32 in the given arithmetic variant."""
    dx, dy, dz = (_f32(p[..., i] - q[..., i]) for i in range(3))
    if variant == 'plain':
        return _f32(_f32(_f32(dx * dx) + _f32(dy * dy)) + _f32(dz * dz))
    fma = lambda a, b, c: _f32(a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64))
    if variant == 'fma_xy':
        return fma(dz, dz, fma(dy, dy, _f32(dx * dx)))
    return fma(dz, dz, fma(dx, dx, _f32(dy * dy)))


def _ulps(rel):
    return float(rel / 2.0 ** -23)


def ball_margin(new_xyz, xyz, radius, nsample):
    new_xyz, xyz = new_xyz.numpy(), xyz.numpy()
    r2 = np.float32(np.float32(radius) * np.float32(radius))
    res = {}
    for variant in ('plain', 'fma_xy', 'fma_yx'):
        d2 = _sqdist(new_xyz[:, :, None, :], xyz[:, None, :, :], variant)           # (B, M, N)
        hit = d2 < r2
        B, M, N = hit.shape
        idx = np.zeros((B, M, nsample), dtype=np.int32)
        for b in range(B):
            for j in range(M):
                h = np.nonzero(hit[b, j])[0][:nsample]
                if len(h):
                    idx[b, j] = h[0]
                    idx[b, j, :len(h)] = h
        res[variant] = idx
        if variant == 'plain':
            gap = _ulps(np.min(np.abs(d2.astype(np.float64) - float(r2)) / float(r2)))
    return res, gap


def fps_margin(xyz, m):
    xyz = xyz.numpy()
    res, gap = {}, np.inf
    for variant in ('plain', 'fma_xy', 'fma_yx'):
        B, N, _ = xyz.shape
        out = np.zeros((B, m), dtype=np.int32)
        zero = np.zeros_like(xyz)
        mag = _sqdist(xyz, zero, variant)            # x*x + y*y + z*z, the same contraction pattern
        for b in range(B):
            temp = np.full(N, 1e10, dtype=np.float32)
            old = 0
            for j in range(1, m):
                d = _sqdist(xyz[b], xyz[b, old][None], variant)
                d2 = np.minimum(d, temp)
                temp = d2
                cand = np.where(mag[b] <= 1e-3, np.float32(-1.0), d2)
                old = int(np.argmax(cand))
                out[b, j] = old
                if variant == 'plain':
                    top = np.partition(cand, -2)[-2:]
                    if top[1] > 0:
                        gap = min(gap, _ulps((float(top[1]) - float(top[0])) / float(top[1])))
        res[variant] = out
    return res, gap


def nn3_margin(unknown, known):
    unknown, known = unknown.numpy(), known.numpy()
    res = {}
    for variant in ('plain', 'fma_xy', 'fma_yx'):
        d2 = _sqdist(unknown[:, :, None, :], known[:, None, :, :], variant)          # (B, n, m)
        order = np.argsort(d2, axis=2, kind='stable')
        res[variant] = order[:, :, :3].astype(np.int32)
        if variant == 'plain' and d2.shape[2] > 3:
            s = np.sort(d2, axis=2).astype(np.float64)
            gap = _ulps(np.min((s[:, :, 3] - s[:, :, 2]) / np.maximum(s[:, :, 3], 1e-30)))
    return res, gap


RANDOM_KINDS = ('uniform', 'walk')


def g6():
    E = cpu_ext.OracleExt
    out = {}
    for (b, n, m, kind, seed) in cases.FPS_CASES:
        if n > 6000:
            continue  # keep the fixture small; large clouds are checked live against the oracle
        xyz = cases.cloud(b, n, seed, kind)
        key = f"fps_{b}_{n}_{m}_{kind}_{seed}"
        out[key] = E.furthest_point_sampling(xyz, m).numpy()
        if kind in RANDOM_KINDS and n > 1:
            res, gap = fps_margin(xyz, m)
            assert np.array_equal(res['plain'], out[key]), key       # the numpy replay is the oracle's algorithm
            out['margin_' + key] = np.array([gap, float(np.array_equal(res['fma_xy'], out[key]) and
                                                        np.array_equal(res['fma_yx'], out[key]))])
    for (b, n, m, radius, nsample, kind, seed) in cases.BALL_CASES:
        xyz = cases.cloud(b, n, seed, kind)
        new_xyz = cases.centres_from(xyz, m, seed)
        key = f"ball_{b}_{n}_{m}_{nsample}_{kind}_{seed}"
        out[key] = E.ball_query(new_xyz, xyz, radius, nsample).numpy()
        if kind in RANDOM_KINDS and n > 1:
            res, gap = ball_margin(new_xyz, xyz, radius, nsample)
            assert np.array_equal(res['plain'], out[key]), key
            out['margin_' + key] = np.array([gap, float(np.array_equal(res['fma_xy'], out[key]) and
                                                        np.array_equal(res['fma_yx'], out[key]))])
    for (b, n, m, kind, seed) in [(2, 512, 128, "uniform", 1), (2, 100, 2, "uniform", 2), (2, 300, 64, "lattice", 4)]:
        unknown = cases.cloud(b, n, seed, kind)
        known = cases.cloud(b, m, seed + 50, kind)
        d, i = E.three_nn(unknown, known)
        out[f"nn3_{b}_{n}_{m}_{kind}_{seed}_dist2"] = d.numpy()
        out[f"nn3_{b}_{n}_{m}_{kind}_{seed}_idx"] = i.numpy()
        if kind in RANDOM_KINDS and m > 3:
            res, gap = nn3_margin(unknown, known)
            assert np.array_equal(res['plain'], i.numpy()), (b, n, m, kind)
            out[f"margin_nn3_{b}_{n}_{m}_{kind}_{seed}"] = np.array([gap, float(np.array_equal(res['fma_xy'], i.numpy()) and
                                                                                np.array_equal(res['fma_yx'], i.numpy()))])
    np.savez_compressed(os.path.join(HERE, "g6_ext_ops.npz"), **out)
    print("g6", len(out), "arrays")


def g6b():
    """a1 / a2 pinned by reference-held code.  `libs.farthest_point_sample` is the textbook FPS: start at a random
    point (patched to 0 = the CUDA kernel's start), keep min squared distance, take torch.max's index.  It has neither
    the CUDA kernel's skip of points with |p|^2 <= 1e-3 nor its block-shaped tie rule, so the cases are the random
    clouds without such points, and every pick's winner must lead the runner-up by > 4 float32 ulps (recorded), which
    also makes the result independent of the summation order of dx^2 + dy^2 + dz^2."""
    _ref_path()
    from net_utils import libs
    out = {}
    real_randint = torch.randint
    torch.randint = lambda lo, hi, size, **kw: torch.zeros(size, dtype=kw.get('dtype', torch.long))
    try:
        for (b, n, m, kind, seed) in cases.FPS_CASES:
            if kind not in RANDOM_KINDS or n < 2 or n > 20000:
                continue
            xyz = cases.cloud(b, n, seed, kind)
            if bool(((xyz ** 2).sum(-1) <= 2e-3).any()):
                continue
            _, gap = fps_margin(xyz, m)
            if not gap > 4.0:
                continue
            key = f"{b}_{n}_{m}_{kind}_{seed}"
            cent = libs.farthest_point_sample(xyz, m)                       # (b, m) int64
            out["fps_" + key] = cent.numpy().astype(np.int32)
            out["gap_" + key] = np.array(gap)
            # gather: reference index_points on (B, N, C) rows == gather_points on (B, C, N) columns
            g = torch.Generator().manual_seed(seed)
            feats = torch.randn(b, 5, n, generator=g)
            out["gather_" + key] = libs.index_points(feats.transpose(1, 2).contiguous(), cent).transpose(1, 2).contiguous().numpy()
    finally:
        torch.randint = real_randint
    # a4 forward / a17 three_nn indices from two more pieces of reference-held torch code:
    #   group_points(points (B,C,N), idx (B,P,S)) == libs.index_points on (B,N,C) rows with a 3-D index (libs.py:175-190);
    #   three_nn(x, x) indices == vn_dgcnn_util.knn(x, 3) (vn_dgcnn_util.py:4-10: -|a|^2 + 2ab - |b|^2 and topk), for
    #   clouds whose 3rd / 4th neighbour gap is > 64 ulps of the LARGEST squared norm involved (the expanded form cancels
    #   at that scale), compared as index SETS per point (topk orders its ties with the point itself differently).
    from net_utils import vn_dgcnn_util
    for (b, n, p_, s_, c, seed) in [(2, 512, 128, 16, 7, 1), (1, 100, 9, 5, 3, 2), (3, 64, 64, 4, 1, 3)]:
        g = torch.Generator().manual_seed(100 + seed)
        pts = torch.randn(b, c, n, generator=g)
        idx = torch.randint(0, n, (b, p_, s_), generator=g, dtype=torch.int64)
        key = f"{b}_{n}_{p_}_{s_}_{c}_{seed}"
        out["group_" + key] = libs.index_points(pts.transpose(1, 2).contiguous(), idx).permute(0, 3, 1, 2).contiguous().numpy()
    for (b, n, kind, seed) in [(2, 64, "uniform", 1), (3, 48, "uniform", 2), (2, 100, "uniform", 4), (1, 32, "uniform", 5), (4, 24, "uniform", 6)]:
        xyz = cases.cloud(b, n, seed, kind)
        d2 = ((xyz[:, :, None, :].double() - xyz[:, None, :, :].double()) ** 2).sum(-1)
        srt = d2.sort(-1)[0]
        gap = ((srt[..., 3] - srt[..., 2]) / (xyz.double() ** 2).sum(-1).max()).min().item() / 2.0 ** -23
        if not gap > 64.0:
            continue
        key = f"{b}_{n}_{kind}_{seed}"
        out["knn3_" + key] = vn_dgcnn_util.knn(xyz.transpose(1, 2).contiguous(), 3).sort(-1)[0].numpy().astype(np.int32)
        out["knn3gap_" + key] = np.array(gap)
    assert len([k for k in out if k.startswith('knn3_')]) >= 2, sorted(out)
    assert len([k for k in out if k.startswith('fps_')]) >= 6, sorted(out)
    np.savez_compressed(os.path.join(HERE, "g6b_fps_ref.npz"), **out)
    print("g6b", sorted(k for k in out if k.startswith('gap_')), [float(out[k]) for k in sorted(out) if k.startswith('gap_')])


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g6", "g6b"]
    for w in which:
        globals()[w]()
