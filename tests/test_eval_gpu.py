"""GPU: evaluation path (prediction parsing + HIP NMS) against golden vectors captured
from the imported reference's `P2RNet.generate` (G3/G5)."""
import os

import numpy as np
import pytest
import torch

from tests.test_model_cpu import build, G

pytestmark = pytest.mark.gpu


def _endpoints_from_golden(z, tag, dev):
    ep = {k: torch.from_numpy(z[f'{tag}_{k}']).to(dev)
          for k in ['center', 'size', 'heading', 'objectness_scores', 'sem_cls_scores']}
    return ep


@pytest.mark.parametrize("tag,B,T", [('g3u', 1, 768), ('g3f', 2, 512)])
def test_parse_predictions_from_reference_endpoints(dev, tag, B, T):
    """Same network outputs in -> same parsed predictions / NMS mask / per-class lists out."""
    from pose2room_amd.net_utils import ap_helper
    from pose2room_amd.p2rnet import P2RConfig, default_config
    from pose2room_amd.p2rnet.synthetic import make_batch
    z = np.load(G)
    cfg = P2RConfig(default_config('test', data={'num_frames': T}, test={'remove_far_box': False}), device=dev)
    data = make_batch(B, T, seed=100 + T, device=dev)
    ep = _endpoints_from_golden(z, tag, dev)
    eval_dict, parsed = ap_helper.parse_predictions(ep, data, cfg.eval_config)
    assert eval_dict['pred_mask'].dtype == np.uint8
    assert np.array_equal(eval_dict['pred_mask'], z[f'{tag}_pred_mask'])
    np.testing.assert_allclose(parsed['pred_corners_3d'], z[f'{tag}_pred_corners_3d'], rtol=1e-5, atol=1e-6)  # device exp/atan2/cos differ from libm by an ulp
    assert parsed['pred_corners_3d'].dtype == np.float64
    np.testing.assert_allclose(parsed['obj_prob'], z[f'{tag}_obj_prob'], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(parsed['sem_cls_probs'], z[f'{tag}_sem_cls_probs'], rtol=1e-6, atol=1e-7)
    assert np.array_equal(parsed['pred_sem_cls'], z[f'{tag}_pred_sem_cls'])
    eval_dict = ap_helper.assembly_pred_map_cls(eval_dict, parsed, cfg.eval_config)
    gts = ap_helper.assembly_gt_map_cls(ap_helper.parse_groundtruths(data, cfg.eval_config))
    for i in range(B):
        lst = eval_dict['batch_pred_map_cls'][i]
        want = z[f'{tag}_map_cls_{i}']
        assert len(lst) == want.shape[0]
        assert [c for c, _, _ in lst] == want[:, 0].astype(int).tolist()      # class-major ordering
        np.testing.assert_allclose([s for _, _, s in lst], want[:, 1], rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(np.array([b for _, b, _ in lst]).reshape(-1, 8, 3), z[f'{tag}_map_box_{i}'],
                                   rtol=1e-5, atol=1e-6)
        assert [c for c, _ in gts[i]] == z[f'{tag}_gt_cls_{i}'].tolist()
        np.testing.assert_allclose(np.array([b for _, b in gts[i]]).reshape(-1, 8, 3), z[f'{tag}_gt_box_{i}'],
                                   rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("tag,B,T", [('g3u', 1, 768), ('g3f', 2, 512)])
def test_parse_predictions_2d_nms_branch(dev, tag, B, T):
    """`use_3d_nms: False` (ap_helper.py:198-214 -> nms_2d_faster on the (x, z) extents), against the masks the
    reference's parse_predictions gives on its own end points (G5b, tests/golden/make_nms2d_golden.py):
    (1) through our parse_predictions at the configured threshold, both overlap definitions;
    (2) at thresholds where the suppression is selective (57-252 of 128-256 boxes survive) a keep mask is a function of
        IoU comparisons whose margins are below the one-ulp differences between device and libm exp / atan2 / cos in
        the corners -- there the 2-D boxes are built from the reference's RECORDED corners and scores exactly as
        ap_helper.py:201-207 builds them, and `nms_2d_faster` must return the reference's mask bit for bit."""
    from pose2room_amd.net_utils import ap_helper, nms
    from pose2room_amd.p2rnet import P2RConfig, default_config
    from pose2room_amd.p2rnet.synthetic import make_batch
    z = np.load(G)
    z2 = np.load(os.path.join(os.path.dirname(G), 'g5b_nms2d.npz'))
    data = make_batch(B, T, seed=100 + T, device=dev)
    ep = _endpoints_from_golden(z, tag, dev)
    ep['objectness_scores'] = torch.from_numpy(z2[f'{tag}_objectness_scores']).to(dev)     # de-tied scores, see the script
    for old in (False, True):
        cfg = P2RConfig(default_config('test', data={'num_frames': T},
                                       test={'remove_far_box': False, 'use_3d_nms': False, 'use_old_type_nms': old,
                                             'nms_iou': 0.25}), device=dev)
        assert cfg.eval_config['use_3d_nms'] is False
        eval_dict, _ = ap_helper.parse_predictions(ep, data, cfg.eval_config)
        want = z2[f'{tag}_pred_mask_2d_25_{int(old)}']
        assert eval_dict['pred_mask'].dtype == np.uint8
        assert np.array_equal(eval_dict['pred_mask'], want), (tag, old, int(eval_dict['pred_mask'].sum()), int(want.sum()))
    corners, prob = z2[f'{tag}_pred_corners_3d'], z2[f'{tag}_obj_prob']
    K = corners.shape[1]
    for iou in (0.25, 0.7, 0.9, 0.97):
        for old in (False, True):
            want = z2[f'{tag}_pred_mask_2d_{int(round(iou * 100))}_{int(old)}']
            for i in range(B):
                b2 = np.zeros((K, 5))
                b2[:, 0], b2[:, 2] = corners[i, :, :, 0].min(1), corners[i, :, :, 0].max(1)
                b2[:, 1], b2[:, 3] = corners[i, :, :, 2].min(1), corners[i, :, :, 2].max(1)
                b2[:, 4] = prob[i]
                mask = np.zeros(K, dtype=np.uint8)
                mask[nms.nms_2d_faster(b2, iou, old)] = 1
                assert np.array_equal(mask, want[i]), (tag, iou, old, i, int(mask.sum()), int(want[i].sum()))


# ---- decision-level comparison of end-to-end keep masks -------------------------------------------------------------
# `north_star`: keep masks bit-exact.  From the reference's end points they are (test above).  End to end the inputs of
# the parse are OUR network's fp32 outputs, equal to the reference's to ~1e-5, and the mask is a function of DECISIONS
# on them (ap_helper.py:171-232, nms.py:41-77): which boxes pass the far-box filter, which of two boxes scores higher,
# whether a pair overlaps by more than the threshold.  The check is therefore made on the decisions:
#   (1) every decision of our run that differs from the reference run's must sit within MARGIN of its threshold IN THE
#       REFERENCE'S OWN NUMBERS (score gap, |IoU - thr|, distance of the nearest hip position to the enlarged box face);
#   (2) our mask is exactly what the reference's algorithm (the oracle's nms_3d, pinned bit for bit to
#       net_utils/nms.py by G2) makes of OUR boxes, and the reference's recorded mask is exactly what it makes of the
#       reference's boxes -- so masks can differ only through decisions that (1) has attributed to fp32 noise;
#   (3) when no decision differs, the masks are bit-equal.
MARGIN = 1e-4


def _aabb_iou(corners):
    mins, maxs = corners.min(2), corners.max(2)                                    # (B,K,3)
    vol = np.prod(maxs - mins, -1)
    lo = np.maximum(mins[:, :, None, :], mins[:, None, :, :])
    hi = np.minimum(maxs[:, :, None, :], maxs[:, None, :, :])
    inter = np.prod(np.maximum(0.0, hi - lo), -1)
    return inter / (vol[:, :, None] + vol[:, None, :] - inter), np.concatenate([mins, maxs], -1)


def _replay_mask(oracle, corners, obj_prob, thr, nonempty):
    """the reference's mask construction (ap_helper.py:216-232) on the given boxes, NMS by the oracle"""
    _, aabb = _aabb_iou(corners)
    B, K = obj_prob.shape
    mask = np.zeros((B, K), dtype=np.uint8)
    for i in range(B):
        boxes = np.concatenate([aabb[i], obj_prob[i].astype(np.float64)[:, None]], 1)
        inds = np.nonzero(nonempty[i])[0]
        pick = oracle.nms_3d(np.ascontiguousarray(boxes[inds]), thr, False)
        mask[i, inds[pick]] = 1
    return mask


def _far_box_margin(cfg, data, center, size_log, heading_sc):
    """(nonempty (B,K) bool, margin (B,K)): the far-box filter's decision and how far it is from flipping (metres /
    size units), float64 replay of ap_helper.py:171-196 in its closed form."""
    from pose2room_amd.net_utils import ap_helper
    center = torch.as_tensor(center).double()
    size = torch.exp(torch.as_tensor(size_log)).double()
    hd = torch.as_tensor(heading_sc).double()
    heading = torch.atan2(hd[..., 0], hd[..., 1])
    hips = data['input_joints'][:, :, cfg.dataset_config.origin_joint_id, 0:3].cpu().double()
    R = ap_helper.head2rot_t(heading)
    half = size / 2. + cfg.dataset_config.contact_dist_thresh
    local = torch.einsum('bktd,bkid->bkti', hips[:, None] - center[:, :, None], R)
    surf = (local.abs() - half[:, :, None, :]).max(-1).values.min(-1).values           # <= 0: some hip inside
    size_bad = ((size < 0.01) | (size > 10)).any(-1)
    size_m = torch.minimum((size - 0.01).abs(), (size - 10).abs()).min(-1).values
    return ((surf <= 0) & ~size_bad).numpy(), torch.minimum(surf.abs(), size_m).numpy()


def _check_keep_masks(oracle, what, thr, ours, ref, got_mask, ref_mask, ours_nonempty=None, ref_nonempty=None,
                      ref_nonempty_margin=None):
    """ours / ref = (corners (B,K,8,3) f64, obj_prob (B,K))."""
    (c_o, p_o), (c_r, p_r) = ours, ref
    B, K = p_r.shape
    ones = np.ones((B, K), dtype=bool)
    ne_o = ones if ours_nonempty is None else ours_nonempty
    ne_r = ones if ref_nonempty is None else ref_nonempty
    # (2) both masks are the reference algorithm's output on their own boxes
    assert np.array_equal(_replay_mask(oracle, c_r, p_r, thr, ne_r), ref_mask), f'{what}: replay of the reference run'
    assert np.array_equal(_replay_mask(oracle, c_o, p_o, thr, ne_o), got_mask), f'{what}: our mask vs nms.py on our boxes'
    # (1) decisions
    iou_o, _ = _aabb_iou(c_o)
    iou_r, _ = _aabb_iou(c_r)
    d_over = (iou_o > thr) != (iou_r > thr)
    gap_r = p_r.astype(np.float64)[:, :, None] - p_r.astype(np.float64)[:, None, :]
    gap_o = p_o.astype(np.float64)[:, :, None] - p_o.astype(np.float64)[:, None, :]
    d_order = (np.sign(gap_o) != np.sign(gap_r)) & ~np.eye(K, dtype=bool)[None]
    assert (np.abs(iou_r - thr)[d_over] < MARGIN).all(), f'{what}: an overlap decision flipped outside the noise margin'
    assert (np.abs(gap_r)[d_order] < MARGIN).all(), f'{what}: a score order flipped outside the noise margin'
    n_ne = 0
    if ref_nonempty_margin is not None:
        d_ne = ne_o != ne_r
        n_ne = int(d_ne.sum())
        assert (ref_nonempty_margin[d_ne] < MARGIN).all(), f'{what}: a far-box decision flipped outside the noise margin'
    n_dec = int(d_over.sum()) // 2 + int(d_order.sum()) // 2 + n_ne
    # (3)
    if n_dec == 0:
        assert np.array_equal(got_mask, ref_mask), f'{what}: same decisions, other mask'
    print(f'{what}: {int((got_mask != ref_mask).sum())} of {ref_mask.size} keep entries differ; decisions that differ from the '
          f'reference run: {int(d_over.sum()) // 2} overlap, {int(d_order.sum()) // 2} order, {n_ne} far-box '
          f'(all within {MARGIN:g} of their threshold in the reference run)')


def test_generate_end_to_end(dev, oracle, mathmode):
    """Whole `generate` on the GPU (network + parsing + NMS) against the reference run."""
    from pose2room_amd.p2rnet.synthetic import make_batch
    z = np.load(G)
    net, cfg = build('test', 512, device=dev, remove_far_box=False)
    net = net.to(dev).eval()
    data = make_batch(2, 512, seed=612, device=dev)
    with torch.no_grad():
        ep, eval_dict, parsed = net.generate(data, eval=True)
    np.testing.assert_allclose(parsed['pred_corners_3d'], z['g3f_pred_corners_3d'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(parsed['obj_prob'], z['g3f_obj_prob'], rtol=1e-4, atol=1e-5)
    _check_keep_masks(oracle, 'generate (g3f)', cfg.eval_config['nms_iou'],
                      (parsed['pred_corners_3d'], parsed['obj_prob']), (z['g3f_pred_corners_3d'], z['g3f_obj_prob']),
                      eval_dict['pred_mask'], z['g3f_pred_mask'])
    assert len(eval_dict['batch_gt_map_cls']) == 2


def test_far_box_filter_matches_delaunay_reference(dev, oracle, mathmode):
    """remove_far_box=True: the closed-form point-in-box test vs the reference's
    Delaunay hull test (G5, captured with the same weights and inputs)."""
    from pose2room_amd.p2rnet.synthetic import make_batch
    z = np.load(G)
    if 'g5far_pred_mask' not in z.files:
        pytest.skip("reference asserted on this fixture")
    net, cfg = build('test', 512, device=dev, remove_far_box=True)
    net = net.to(dev).eval()
    data = make_batch(2, 512, seed=612, device=dev)
    with torch.no_grad():
        ep, eval_dict, parsed = net.generate(data, eval=False)
    # far-box decisions of both runs from their own end points (the reference's: same weights as the g3f run)
    ne_r, margin_r = _far_box_margin(cfg, data, z['g3f_center'], z['g3f_size'], z['g3f_heading'])
    ne_o, _ = _far_box_margin(cfg, data, ep['center'].detach().cpu(), ep['size'].detach().cpu(), ep['heading'].detach().cpu())
    _check_keep_masks(oracle, 'generate with far-box filter', cfg.eval_config['nms_iou'],
                      (parsed['pred_corners_3d'], parsed['obj_prob']), (z['g3f_pred_corners_3d'], z['g3f_obj_prob']),
                      eval_dict['pred_mask'], z['g5far_pred_mask'], ne_o, ne_r, margin_r)


def test_test_loop_metrics(dev, mathmode):
    """test_epoch-style loop: Tester.test_step over batches -> loss meters + APCalculator per IoU threshold.
    The metric of the pipeline's own lists must equal the metric of the same lists evaluated pair by pair."""
    from pose2room_amd.net_utils import eval_det, box_util
    from pose2room_amd.p2rnet import testing
    from pose2room_amd.p2rnet.training import ModuleWrapper
    from pose2room_amd.p2rnet.synthetic import make_batch
    net, cfg = build('test', 256, device=dev, remove_far_box=True)
    tester = testing.Tester(cfg, ModuleWrapper(net.to(dev)), dev)
    batches = [make_batch(2, 256, seed=900 + i, device=dev) for i in range(2)]
    logged = []
    cfg.log_string = logged.append
    out = testing.test(cfg, tester, batches, ap_device=dev)
    assert set(out['loss']) >= {'total', 'vote_loss', 'center_loss', 'obj_acc'}
    assert len(out['metrics']) == len(cfg.config['test']['ap_iou_thresholds'])
    for m in out['metrics']:
        assert 'mAP' in m and 'AR' in m and 0.0 <= m['AR'] <= 1.0
    assert any(s.startswith('eval mAP') for s in logged)
    # same lists, per-pair IoU callback on the CPU
    with torch.no_grad():
        _, calcs = testing.test_func(cfg, tester, batches, ap_device='cpu')
    f = lambda a, b: box_util.box3d_iou(a, b)[0]       # noqa: E731
    for calc, m in zip(calcs, out['metrics']):
        _, _, ap = eval_det.eval_det_multiprocessing_wo_mesh(calc.pred_map_cls, calc.gt_map_cls, calc.ap_iou_thresh,
                                                             get_iou_func=f)
        vals = [v for v in ap.values() if not np.isnan(v)]
        assert m['mAP'] == pytest.approx(np.mean(vals), abs=1e-9)
