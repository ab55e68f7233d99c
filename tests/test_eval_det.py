"""Detection metric against G7, the outputs of the reference's net_utils/box_util.py,
net_utils/eval_det.py and ap_helper.APCalculator recorded by tests/golden/make_eval_golden.py."""
import os

import numpy as np
import pytest
import torch

from pose2room_amd.net_utils import box_util, eval_det
from pose2room_amd.net_utils.ap_helper import APCalculator

G7 = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g7_eval_det.npz'))
DEVICES = ['cpu', pytest.param('cuda', marks=pytest.mark.gpu)]


def _maps():
    n_scan = int(G7['n_scan'])
    pred_all = {i: [] for i in range(n_scan)}
    gt_all = {i: [] for i in range(n_scan)}
    for row, c in zip(G7['det_rows'], G7['det_corners']):
        pred_all[int(row[0])].append((int(row[1]), c, float(row[2])))
    for row, c in zip(G7['gt_rows'], G7['gt_corners']):
        gt_all[int(row[0])].append((int(row[1]), c))
    return pred_all, gt_all, n_scan


@pytest.mark.parametrize('device', DEVICES)
def test_box3d_iou_pairs_match_reference(device):
    iou, iou2d = box_util.box3d_iou_pairs(torch.as_tensor(G7['iou_a'], device=device), torch.as_tensor(G7['iou_b'], device=device))
    np.testing.assert_allclose(iou.cpu().numpy(), G7['iou_3d'], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(iou2d.cpu().numpy(), G7['iou_2d'], rtol=1e-9, atol=1e-12)
    assert (G7['iou_3d'] > 0.3).sum() > 30 and (G7['iou_3d'] == 0).sum() > 30      # the set spans both regimes


def test_box3d_iou_scalar_and_matrix_forms():
    a, b = G7['iou_a'][:7], G7['iou_b'][:5]
    m = box_util.box3d_iou_matrix(a, b).numpy()
    assert m.shape == (7, 5)
    for i in (0, 3, 6):
        for j in (0, 4):
            i3, i2 = box_util.box3d_iou(a[i], b[j])
            assert i3 == pytest.approx(m[i, j], abs=1e-15)
            assert eval_det.get_iou_obb(a[i], b[j]) == pytest.approx(i3, abs=1e-15)
    assert box_util.box3d_iou_matrix(a[:0], b).shape == (0, 5)


@pytest.mark.parametrize('device', DEVICES)
@pytest.mark.parametrize('thr', [0.25, 0.5])
def test_eval_det_matches_reference(device, thr):
    pred_all, gt_all, _ = _maps()
    rec, prec, ap = eval_det.eval_det_multiprocessing_wo_mesh(pred_all, gt_all, ovthresh=thr, device=device)
    tag = 'thr%02d' % int(thr * 100)
    assert sorted(ap.keys()) == list(G7[tag + '_classes'])
    for k in ap:
        np.testing.assert_allclose(ap[k], G7['%s_ap_%d' % (tag, k)], rtol=1e-12, equal_nan=True)
        np.testing.assert_allclose(np.asarray(rec[k], dtype=np.float64), G7['%s_rec_%d' % (tag, k)], rtol=1e-12, equal_nan=True)
        np.testing.assert_allclose(np.asarray(prec[k], dtype=np.float64), G7['%s_prec_%d' % (tag, k)], rtol=1e-12, equal_nan=True)


@pytest.mark.parametrize('thr', [0.25, 0.5])
def test_ap_calculator_matches_reference(thr):
    pred_all, gt_all, n_scan = _maps()
    calc = APCalculator(thr, None, False)
    half = n_scan // 2
    calc.step([pred_all[i] for i in range(half)], [gt_all[i] for i in range(half)])
    calc.step([pred_all[i] for i in range(half, n_scan)], [gt_all[i] for i in range(half, n_scan)])
    m = calc.compute_metrics()
    tag = 'thr%02d' % int(thr * 100)
    assert list(m.keys()) == list(G7[tag + '_metric_keys'])
    np.testing.assert_allclose(np.array([float(v) for v in m.values()]), G7[tag + '_metric_vals'], rtol=1e-12, equal_nan=True)
    calc.reset()
    assert calc.scan_cnt == 0 and not calc.pred_map_cls
    with pytest.raises(NotImplementedError):
        APCalculator(thr, None, True)


def test_per_pair_iou_callback_path():
    """A user-supplied get_iou_func goes through the per-pair loop (eval_det.py:303-309) and must agree."""
    pred_all, gt_all, _ = _maps()
    f = lambda a, b: box_util.box3d_iou(a, b)[0]       # noqa: E731
    r0 = eval_det.eval_det_multiprocessing_wo_mesh(pred_all, gt_all, ovthresh=0.25)
    r1 = eval_det.eval_det_multiprocessing_wo_mesh(pred_all, gt_all, ovthresh=0.25, get_iou_func=f)
    for k in r0[2]:
        np.testing.assert_allclose(r0[2][k], r1[2][k], rtol=1e-12, equal_nan=True)


def test_voc_ap_forms():
    rec = np.array([0.1, 0.2, 0.2, 0.5, 0.9])
    prec = np.array([1.0, 1.0, 0.66, 0.75, 0.6])
    assert eval_det.voc_ap(rec, prec) == pytest.approx(0.1 * 1.0 + 0.1 * 1.0 + 0.3 * 0.75 + 0.4 * 0.6)
    assert eval_det.voc_ap(rec, prec, use_07_metric=True) == pytest.approx((3 * 1.0 + 3 * 0.75 + 4 * 0.6) / 11.)


def test_lazy_prediction_lists_equal_the_reference_lists():
    """ap_helper.PredMapCls (what assembly_pred_map_cls returns per sample) must BE the reference's list -- same length,
    order (class-major, then proposal index), tuples -- and the array fast path of the AP evaluation must give the very
    numbers the tuple-by-tuple path gives (ap_helper.py:294-350, eval_det.py:424-473)."""
    from pose2room_amd.net_utils import ap_helper, eval_det
    from pose2room_amd.p2rnet.config import DatasetConfig
    rng = np.random.default_rng(3)
    B, K, C = 3, 40, 22
    corners = rng.normal(size=(B, K, 8, 3))
    probs = rng.uniform(size=(B, K, C)).astype(np.float32)
    obj = rng.uniform(size=(B, K)).astype(np.float32)
    mask = (rng.uniform(size=(B, K)) > 0.4).astype(np.uint8)
    cls = rng.integers(0, C, (B, K))
    for per_class in (True, False):
        cfg = {'conf_thresh': 0.05, 'per_class_proposal': per_class, 'dataset_config': DatasetConfig()}
        parsed = {'pred_corners_3d': corners, 'sem_cls_probs': probs, 'obj_prob': obj, 'pred_sem_cls': cls}
        lazy = ap_helper.assembly_pred_map_cls({'pred_mask': mask}, parsed, cfg)['batch_pred_map_cls']
        # the reference's comprehensions, literally
        want = []
        for i in range(B):
            if per_class:
                cur = []
                for ii in range(C):
                    cur += [(ii, corners[i, j], probs[i, j, ii] * obj[i, j]) for j in range(K)
                            if mask[i, j] == 1 and obj[i, j] > 0.05]
            else:
                cur = [(cls[i, j].item(), corners[i, j], obj[i, j]) for j in range(K) if mask[i, j] == 1 and obj[i, j] > 0.05]
            want.append(cur)
        for a, b in zip(lazy, want):
            assert len(a) == len(b)
            for (c1, b1, s1), (c2, b2, s2) in zip(a, b):
                assert c1 == c2 and np.array_equal(b1, b2) and s1 == s2
            assert len(a) == 0 or (a[-1][0] == b[-1][0] and a[len(a) // 2][2] == b[len(b) // 2][2])
        # AP: array path == tuple path
        gts = {i: [(int(rng.integers(0, C)), corners[i, j] + 0.01) for j in range(5)] for i in range(B)}
        r1 = eval_det.eval_det_multiprocessing_wo_mesh({i: lazy[i] for i in range(B)}, gts, 0.25)
        r2 = eval_det.eval_det_multiprocessing_wo_mesh({i: want[i] for i in range(B)}, gts, 0.25)
        assert set(r1[2]) == set(r2[2])
        for k in r2[2]:
            assert np.array_equal(np.asarray(r1[2][k]), np.asarray(r2[2][k]), equal_nan=True), k
            assert np.array_equal(np.asarray(r1[0][k]), np.asarray(r2[0][k]), equal_nan=True)


def test_assembly_gt_map_cls_from_tensors():
    import torch
    from pose2room_amd.net_utils import ap_helper
    labels = torch.tensor([[3, 5, 0], [1, 0, 0]])
    corners = np.arange(2 * 3 * 24, dtype=np.float64).reshape(2, 3, 8, 3)
    mask = np.array([[1, 1, 0], [1, 0, 0]])
    out = ap_helper.assembly_gt_map_cls({'sem_cls_label': labels, 'gt_corners_3d': corners, 'box_label_mask': mask})
    assert [[c for c, _ in s] for s in out] == [[3, 5], [1]]
    assert np.array_equal(out[0][1][1], corners[0, 1]) and all(isinstance(c, int) for s in out for c, _ in s)
