"""Generate the golden fixtures under tests/golden/ (run in the BUILD container only).

    python tests/golden/make_golden.py [g1 g2 g6 g3 g4 g5 ...]

The reference Python (/root/reference, read-only) is imported here -- and only
here -- to produce input/output vectors; it never travels to the GPU box.  The
fixtures are data: seeded inputs (or the seeds to rebuild them through
tests/cases.py) and the reference's outputs.

  G1  nn_distance: the reference's own demo inputs (net_utils/nn_distance.py:63-94,
      np.random.seed(0)) and the three loss call shapes, all modes, with the
      autograd gradients of a fixed scalarisation.
  G2  nms_3d_faster / nms_3d_faster_samecls pick lists on seeded boxes.
  G6  the nine _ext ops: outputs of the CPU restatement (oracle/p2r_oracle.c) on
      seeded random and adversarial clouds.  The reference has no CPU path and
      no tests for these, so G6 pins the *restatement* against regressions; it
      is not reference-generated ("parity unpinned").
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
    from net_utils.nms import nms_3d_faster, nms_3d_faster_samecls
    out = {}
    for K in (1, 2, 16, 128, 300):
        boxes = cases.random_boxes(K, seed=K)
        out[f"boxes_{K}"] = boxes
        for thr in (0.10, 0.25):
            for old in (False, True):
                tag = f"{K}_{int(thr * 100)}_{int(old)}"
                out[f"pick_{tag}"] = np.asarray(nms_3d_faster(boxes[:, :7], thr, old), dtype=np.int32)
                out[f"pickcls_{tag}"] = np.asarray(nms_3d_faster_samecls(boxes, thr, old), dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "g2_nms.npz"), **out)
    print("g2", len(out), "arrays")


def g6():
    E = cpu_ext.OracleExt
    out = {}
    for (b, n, m, kind, seed) in cases.FPS_CASES:
        if n > 6000:
            continue  # keep the fixture small; large clouds are checked live against the oracle
        xyz = cases.cloud(b, n, seed, kind)
        out[f"fps_{b}_{n}_{m}_{kind}_{seed}"] = E.furthest_point_sampling(xyz, m).numpy()
    for (b, n, m, radius, nsample, kind, seed) in cases.BALL_CASES:
        xyz = cases.cloud(b, n, seed, kind)
        new_xyz = cases.centres_from(xyz, m, seed)
        out[f"ball_{b}_{n}_{m}_{nsample}_{kind}_{seed}"] = E.ball_query(new_xyz, xyz, radius, nsample).numpy()
    for (b, n, m, kind, seed) in [(2, 512, 128, "uniform", 1), (2, 100, 2, "uniform", 2), (2, 300, 64, "lattice", 4)]:
        unknown = cases.cloud(b, n, seed, kind)
        known = cases.cloud(b, m, seed + 50, kind)
        d, i = E.three_nn(unknown, known)
        out[f"nn3_{b}_{n}_{m}_{kind}_{seed}_dist2"] = d.numpy()
        out[f"nn3_{b}_{n}_{m}_{kind}_{seed}_idx"] = i.numpy()
    np.savez_compressed(os.path.join(HERE, "g6_ext_ops.npz"), **out)
    print("g6", len(out), "arrays")


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g6"]
    for w in which:
        globals()[w]()
