"""CPU: the oracle against the committed golden vectors.

G1/G2 were produced by the imported reference Python (tests/golden/make_golden.py),
so these tests pin the oracle's nn_distance / NMS to the reference.  G6 pins the
restatement of the nine _ext kernels against regressions, and the property tests
below check it against independent brute-force definitions."""
import os

import numpy as np
import pytest
import torch

from tests import cases

G = os.path.join(os.path.dirname(__file__), "golden")


def test_g1_nn_distance_matches_reference(oracle):
    z = np.load(os.path.join(G, "g1_nn_distance.npz"))
    for name in ("demo", "vote", "assign", "center"):
        a = torch.from_numpy(z[f"{name}_pc1"]); q = torch.from_numpy(z[f"{name}_pc2"])
        w1 = torch.from_numpy(z[f"{name}_w1"]); w2 = torch.from_numpy(z[f"{name}_w2"])
        for mode, kw in {"l2": {}, "l1smooth": {"l1smooth": True}, "l1": {"l1": True}}.items():
            d1, i1, d2, i2 = oracle.nn_distance(a, q, **kw)
            assert np.array_equal(d1.numpy(), z[f"{name}_{mode}_dist1"])
            assert np.array_equal(i1.numpy(), z[f"{name}_{mode}_idx1"])
            assert np.array_equal(d2.numpy(), z[f"{name}_{mode}_dist2"])
            assert np.array_equal(i2.numpy(), z[f"{name}_{mode}_idx2"])
            ga, gq = oracle.nn_distance_grad(a, q, i1, i2, w1, w2, **kw)
            np.testing.assert_allclose(ga.numpy(), z[f"{name}_{mode}_grad1"], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(gq.numpy(), z[f"{name}_{mode}_grad2"], rtol=1e-5, atol=1e-6)


def test_g1_demo_known_answer(oracle):
    """The reference demo's check (nn_distance.py:63-94): torch result == NumPy double loop."""
    z = np.load(os.path.join(G, "g1_nn_distance.npz"))
    a, q = z["demo_pc1"][0].astype(np.float64), z["demo_pc2"][0].astype(np.float64)
    dist = ((a[:, None, :] - q[None, :, :]) ** 2).sum(-1)
    d1, i1, d2, i2 = oracle.nn_distance(torch.from_numpy(z["demo_pc1"]), torch.from_numpy(z["demo_pc2"]))
    np.testing.assert_allclose(d1[0].numpy(), dist.min(1), rtol=1e-6)
    assert np.array_equal(i1[0].numpy(), dist.argmin(1))
    np.testing.assert_allclose(d2[0].numpy(), dist.min(0), rtol=1e-6)
    assert np.array_equal(i2[0].numpy(), dist.argmin(0))


def test_g2_nms_matches_reference(oracle):
    z = np.load(os.path.join(G, "g2_nms.npz"))
    for K in (1, 2, 16, 128, 300):
        boxes = z[f"boxes_{K}"]
        assert np.array_equal(boxes, cases.random_boxes(K, seed=K))  # builder is stable
        for thr in (0.10, 0.25):
            for old in (False, True):
                tag = f"{K}_{int(thr * 100)}_{int(old)}"
                assert oracle.nms_3d(boxes[:, :7], thr, old) == z[f"pick_{tag}"].tolist()
                assert oracle.nms_3d(boxes, thr, old, True) == z[f"pickcls_{tag}"].tolist()
                assert oracle.nms_2d(cases.boxes_xz(boxes), thr, old) == z[f"pick2d_{tag}"].tolist()


def test_g5b_2d_nms_masks_match_reference(oracle):
    """The reference's `use_3d_nms: False` keep masks (ap_helper.py:198-214) from its own recorded corners and scores:
    the 2-D oracle (nms.py:7-39) reproduces them at every threshold / overlap definition of the fixture."""
    z2 = np.load(os.path.join(G, "g5b_nms2d.npz"))
    for tag, B in (("g3u", 1), ("g3f", 2)):
        corners, prob = z2[f"{tag}_pred_corners_3d"], z2[f"{tag}_obj_prob"]
        K = corners.shape[1]
        for iou in (0.25, 0.7, 0.9, 0.97):
            for old in (False, True):
                want = z2[f"{tag}_pred_mask_2d_{int(round(iou * 100))}_{int(old)}"]
                for i in range(B):
                    b2 = np.zeros((K, 5))
                    b2[:, 0], b2[:, 2] = corners[i, :, :, 0].min(1), corners[i, :, :, 0].max(1)
                    b2[:, 1], b2[:, 3] = corners[i, :, :, 2].min(1), corners[i, :, :, 2].max(1)
                    b2[:, 4] = prob[i]
                    mask = np.zeros(K, dtype=np.uint8)
                    mask[oracle.nms_2d(b2, iou, old)] = 1
                    assert np.array_equal(mask, want[i]), (tag, iou, old, i)


def test_g6_ext_ops_stable(oracle):
    z = np.load(os.path.join(G, "g6_ext_ops.npz"))
    E = oracle.OracleExt
    for (b, n, m, kind, seed) in cases.FPS_CASES:
        key = f"fps_{b}_{n}_{m}_{kind}_{seed}"
        if key in z:
            assert np.array_equal(E.furthest_point_sampling(cases.cloud(b, n, seed, kind), m).numpy(), z[key])
    for (b, n, m, radius, nsample, kind, seed) in cases.BALL_CASES:
        xyz = cases.cloud(b, n, seed, kind)
        got = E.ball_query(cases.centres_from(xyz, m, seed), xyz, radius, nsample).numpy()
        assert np.array_equal(got, z[f"ball_{b}_{n}_{m}_{nsample}_{kind}_{seed}"])


def _g6b_cases(z):
    for key in sorted(k[4:] for k in z.files if k.startswith("fps_")):
        b, n, m, kind, seed = key.split("_")
        yield key, int(b), int(n), int(m), kind, int(seed)


def test_g6b_fps_and_gather_match_the_references_torch_code(oracle):
    """a1 / a2 against vectors produced by the REFERENCE's own pure-torch `farthest_point_sample` / `index_points`
    (net_utils/libs.py:152-190, start index patched to 0) on tie-free, origin-free clouds (every pick leads the
    runner-up by > 4 ulps, recorded): the restatement of the CUDA kernel must pick the very same points."""
    z = np.load(os.path.join(G, "g6b_fps_ref.npz"))
    E = oracle.OracleExt
    seen = 0
    for key, b, n, m, kind, seed in _g6b_cases(z):
        assert float(z["gap_" + key]) > 4.0
        xyz = cases.cloud(b, n, seed, kind)
        idx = E.furthest_point_sampling(xyz, m)
        assert np.array_equal(idx.numpy(), z["fps_" + key]), key
        feats = torch.randn(b, 5, n, generator=torch.Generator().manual_seed(seed))
        assert np.array_equal(E.gather_points(feats, idx).numpy(), z["gather_" + key]), key
        seen += 1
    assert seen >= 6


def _g6b_group_cases(z):
    for key in sorted(k[6:] for k in z.files if k.startswith("group_")):
        b, n, p_, s_, c, seed = (int(v) for v in key.split("_"))
        g = torch.Generator().manual_seed(100 + seed)
        pts = torch.randn(b, c, n, generator=g)
        idx = torch.randint(0, n, (b, p_, s_), generator=g, dtype=torch.int64)
        yield key, pts, idx.int()


def _g6b_knn_cases(z):
    for key in sorted(k[5:] for k in z.files if k.startswith("knn3_")):
        b, n, kind, seed = key.split("_")
        yield key, cases.cloud(int(b), int(n), int(seed), kind)


def test_g6b_group_points_and_three_nn_match_the_references_torch_code(oracle):
    """a4 forward against `libs.index_points` with a 3-D index (libs.py:175-190); three_nn(x, x) index sets against
    `vn_dgcnn_util.knn(x, 3)` (vn_dgcnn_util.py:4-10) on clouds with a recorded 3rd / 4th neighbour gap."""
    z = np.load(os.path.join(G, "g6b_fps_ref.npz"))
    E = oracle.OracleExt
    n_g = n_k = 0
    for key, pts, idx in _g6b_group_cases(z):
        assert np.array_equal(E.group_points(pts, idx).numpy(), z["group_" + key]), key
        n_g += 1
    for key, xyz in _g6b_knn_cases(z):
        assert float(z["knn3gap_" + key]) > 64.0
        _, i3 = E.three_nn(xyz, xyz)
        assert np.array_equal(np.sort(i3.numpy(), -1), z["knn3_" + key]), key
        n_k += 1
    assert n_g >= 3 and n_k >= 2


# ---- independent definitions (property tests of the restatement) -------------

def _fps_bruteforce(xyz, m):
    """FPS from first principles for clouds without ties and without skipped points."""
    n = xyz.shape[0]
    d = np.full(n, 1e10, dtype=np.float32)
    out = [0]
    for _ in range(1, m):
        p = xyz[out[-1]]
        diff = (xyz - p).astype(np.float32)
        dist = (diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2]
        d = np.minimum(d, dist.astype(np.float32))
        out.append(int(np.argmax(d)))
    return out


def test_fps_against_bruteforce(oracle):
    xyz = cases.cloud(2, 300, 77) + 5.0  # away from the |p|^2 <= 1e-3 skip ball, no exact ties
    got = oracle.OracleExt.furthest_point_sampling(xyz, 64).numpy()
    for b in range(2):
        assert got[b].tolist() == _fps_bruteforce(xyz[b].numpy(), 64)


def test_fps_tie_rule_is_bit_reversed_thread_order(oracle):
    """n=512 -> block 512, one point per thread; among exactly tied candidates the
    reference's tree picks the smallest bit-reversed thread id."""
    n = 512
    xyz = torch.zeros(1, n, 3)
    xyz[0, :, 0] = 10.0          # everything sits on one point ...
    xyz[0, 3] = torch.tensor([11.0, 0, 0])    # ... except two points equally far from point 0
    xyz[0, 258] = torch.tensor([9.0, 0, 0])
    got = oracle.OracleExt.furthest_point_sampling(xyz, 2).numpy()[0]
    assert got.tolist() == [0, 258]   # bitrev9(258)=129 < bitrev9(3)=384


def test_fps_skips_points_near_origin(oracle):
    xyz = cases.cloud(1, 64, 5) + 3.0
    xyz[0, 10] = torch.tensor([0.01, 0.01, 0.01])   # |p|^2 = 3e-4 <= 1e-3: never selected
    got = oracle.OracleExt.furthest_point_sampling(xyz, 64).numpy()[0]
    assert 10 not in got[1:].tolist()
    allz = torch.zeros(1, 16, 3)
    assert oracle.OracleExt.furthest_point_sampling(allz, 5).numpy().tolist() == [[0] * 5]


def test_ball_query_against_definition(oracle):
    xyz = cases.cloud(2, 200, 9)
    new_xyz = cases.centres_from(xyz, 30, 9)
    r, ns = 0.7, 8
    got = oracle.OracleExt.ball_query(new_xyz, xyz, r, ns).numpy()
    for b in range(2):
        for j in range(30):
            d = xyz[b] - new_xyz[b, j]
            # new - p vs p - new differ only in sign before squaring
            d2 = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]).numpy()
            hits = np.nonzero(d2 < np.float32(r) * np.float32(r))[0][:ns].tolist()
            want = (hits + [hits[0]] * (ns - len(hits))) if hits else [0] * ns
            assert got[b, j].tolist() == want


def test_three_nn_against_sort(oracle):
    unknown = cases.cloud(1, 50, 1); known = cases.cloud(1, 40, 2)
    d2, idx = oracle.OracleExt.three_nn(unknown, known)
    full = ((unknown[0, :, None, :] - known[0, None, :, :]) ** 2).sum(-1)
    want = torch.topk(full, 3, dim=1, largest=False)
    assert torch.equal(idx[0].long(), want.indices)
    torch.testing.assert_close(d2[0], want.values, rtol=1e-6, atol=1e-7)
    d2, idx = oracle.OracleExt.three_nn(unknown, known[:, :2].contiguous())
    assert torch.isinf(d2[..., 2]).all() and (idx[..., 2] == 0).all()   # m < 3


def test_group_gather_roundtrip(oracle):
    g = torch.Generator().manual_seed(0)
    pts = torch.randn(2, 6, 50, generator=g)
    idx = torch.randint(0, 50, (2, 9, 4), generator=g, dtype=torch.int32)
    out = oracle.OracleExt.group_points(pts, idx)
    want = torch.gather(pts.unsqueeze(2).expand(2, 6, 9, 50), 3, idx.long().unsqueeze(1).expand(2, 6, 9, 4))
    assert torch.equal(out, want)
    go = torch.randn(2, 6, 9, 4, generator=g)
    grad = oracle.OracleExt.group_points_grad(go, idx, 50)
    ref = torch.zeros(2, 6, 50).scatter_add_(2, idx.long().view(2, 1, 36).expand(2, 6, 36), go.view(2, 6, 36))
    torch.testing.assert_close(grad, ref, rtol=1e-5, atol=1e-6)


def test_opt_n_threads(oracle):
    f = oracle.lib().p2r_oracle_opt_n_threads
    assert [f(w) for w in (1, 2, 3, 7, 8, 100, 511, 512, 513, 54272)] == [1, 2, 2, 4, 8, 64, 256, 512, 512, 512]


def test_g6_decision_margins():
    """SURVEY 8c hazard (i): the CUDA-only reference kernels may be built with FMA contraction (nvcc -fmad=true), the
    oracle and the HIP kernels are not.  For every random-cloud fixture, make_golden.py replayed the algorithm in the
    three arithmetic variants (source order, and both contractions of (dx*dx + dy*dy) + dz*dz): the smallest relative
    gap at any decision (d2 vs r2 in ball_query, winner vs runner-up in FPS, 3rd vs 4th neighbour in three_nn) is
    recorded in float32 ulps, together with whether the contracted variants give the very same indices."""
    z = np.load(os.path.join(G, "g6_ext_ops.npz"))
    keys = [k for k in z.files if k.startswith('margin_')]
    assert len(keys) >= 12 and any('fps' in k for k in keys) and any('ball' in k for k in keys) and any('nn3' in k for k in keys)
    for k in keys:
        gap_ulps, same_under_fma = z[k]
        assert gap_ulps > 4.0, (k, gap_ulps)           # no decision within rounding distance of flipping
        assert same_under_fma == 1.0, k                 # and the FMA-contracted replays reproduce the fixture exactly
