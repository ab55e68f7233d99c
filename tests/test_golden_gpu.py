"""GPU: HIP path against the committed golden vectors (reference-generated G1/G2,
oracle-generated G6) -- no oracle call involved."""
import os

import numpy as np
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def test_g1_nn_distance(dev):
    from pose2room_amd.net_utils.nn_distance import nn_distance
    z = np.load(os.path.join(G, "g1_nn_distance.npz"))
    for name in ("demo", "vote", "assign", "center"):
        w1 = torch.from_numpy(z[f"{name}_w1"]).to(dev); w2 = torch.from_numpy(z[f"{name}_w2"]).to(dev)
        for mode, kw in {"l2": {}, "l1smooth": {"l1smooth": True}, "l1": {"l1": True}}.items():
            a = torch.from_numpy(z[f"{name}_pc1"]).to(dev).requires_grad_(True)
            q = torch.from_numpy(z[f"{name}_pc2"]).to(dev).requires_grad_(True)
            d1, i1, d2, i2 = nn_distance(a, q, **kw)
            # bit-exact forward (tolerance stated by the north star is 1e-4; we hold 0)
            assert np.array_equal(d1.detach().cpu().numpy(), z[f"{name}_{mode}_dist1"])
            assert np.array_equal(i1.cpu().numpy(), z[f"{name}_{mode}_idx1"])
            assert np.array_equal(d2.detach().cpu().numpy(), z[f"{name}_{mode}_dist2"])
            assert np.array_equal(i2.cpu().numpy(), z[f"{name}_{mode}_idx2"])
            ((d1 * w1).sum() + (d2 * w2).sum()).backward()
            np.testing.assert_allclose(a.grad.cpu().numpy(), z[f"{name}_{mode}_grad1"], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(q.grad.cpu().numpy(), z[f"{name}_{mode}_grad2"], rtol=1e-5, atol=1e-6)


def test_g2_nms(dev):
    from pose2room_amd.net_utils import nms
    z = np.load(os.path.join(G, "g2_nms.npz"))
    for K in (1, 2, 16, 128, 300):
        boxes = z[f"boxes_{K}"]
        for thr in (0.10, 0.25):
            for old in (False, True):
                tag = f"{K}_{int(thr * 100)}_{int(old)}"
                assert nms.nms_3d_faster(boxes[:, :7], thr, old) == z[f"pick_{tag}"].tolist()
                assert nms.nms_3d_faster_samecls(boxes, thr, old) == z[f"pickcls_{tag}"].tolist()
                assert nms.nms_2d_faster(cases.boxes_xz(boxes), thr, old) == z[f"pick2d_{tag}"].tolist()


def test_g6_ext_ops(dev):
    from pose2room_amd.pointnet2_ops import _ext
    z = np.load(os.path.join(G, "g6_ext_ops.npz"))
    for (b, n, m, kind, seed) in cases.FPS_CASES:
        key = f"fps_{b}_{n}_{m}_{kind}_{seed}"
        if key in z:
            got = _ext.furthest_point_sampling(cases.cloud(b, n, seed, kind).to(dev), m)
            assert np.array_equal(got.cpu().numpy(), z[key]), key
    for (b, n, m, radius, nsample, kind, seed) in cases.BALL_CASES:
        xyz = cases.cloud(b, n, seed, kind)
        got = _ext.ball_query(cases.centres_from(xyz, m, seed).to(dev), xyz.to(dev), radius, nsample)
        assert np.array_equal(got.cpu().numpy(), z[f"ball_{b}_{n}_{m}_{nsample}_{kind}_{seed}"])
    for (b, n, m, kind, seed) in [(2, 512, 128, "uniform", 1), (2, 100, 2, "uniform", 2), (2, 300, 64, "lattice", 4)]:
        d, i = _ext.three_nn(cases.cloud(b, n, seed, kind).to(dev), cases.cloud(b, m, seed + 50, kind).to(dev))
        assert np.array_equal(d.cpu().numpy(), z[f"nn3_{b}_{n}_{m}_{kind}_{seed}_dist2"])
        assert np.array_equal(i.cpu().numpy(), z[f"nn3_{b}_{n}_{m}_{kind}_{seed}_idx"])


def test_g6b_fps_and_gather_vs_reference_torch_code(dev):
    """HIP furthest_point_sampling / gather_points against the vectors of the reference's own pure-torch FPS /
    index_points (net_utils/libs.py:152-190; tests/golden/make_golden.py g6b) -- no oracle involved."""
    from pose2room_amd.pointnet2_ops import _ext
    from tests.test_oracle_golden import _g6b_cases
    z = np.load(os.path.join(G, "g6b_fps_ref.npz"))
    seen = 0
    for key, b, n, m, kind, seed in _g6b_cases(z):
        xyz = cases.cloud(b, n, seed, kind).to(dev)
        idx = _ext.furthest_point_sampling(xyz, m)
        assert np.array_equal(idx.cpu().numpy(), z["fps_" + key]), key
        feats = torch.randn(b, 5, n, generator=torch.Generator().manual_seed(seed)).to(dev)
        assert np.array_equal(_ext.gather_points(feats, idx).cpu().numpy(), z["gather_" + key]), key
        seen += 1
    assert seen >= 6


def test_g6b_group_points_and_three_nn_vs_reference_torch_code(dev):
    """HIP group_points / three_nn against vectors of the reference's `index_points` / `knn` (no oracle involved)."""
    from pose2room_amd.pointnet2_ops import _ext
    from tests.test_oracle_golden import _g6b_group_cases, _g6b_knn_cases
    z = np.load(os.path.join(G, "g6b_fps_ref.npz"))
    n_g = n_k = 0
    for key, pts, idx in _g6b_group_cases(z):
        assert np.array_equal(_ext.group_points(pts.to(dev), idx.to(dev)).cpu().numpy(), z["group_" + key]), key
        n_g += 1
    for key, xyz in _g6b_knn_cases(z):
        _, i3 = _ext.three_nn(xyz.to(dev), xyz.to(dev))
        assert np.array_equal(np.sort(i3.cpu().numpy(), -1), z["knn3_" + key]), key
        n_k += 1
    assert n_g >= 3 and n_k >= 2
