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


def test_generate_end_to_end(dev):
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
    # keep masks agree except where a score / IoU sits within the fp32 noise of the network
    assert (eval_dict['pred_mask'] != z['g3f_pred_mask']).mean() <= 0.01
    assert len(eval_dict['batch_gt_map_cls']) == 2


def test_far_box_filter_matches_delaunay_reference(dev):
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
    assert (eval_dict['pred_mask'] != z['g5far_pred_mask']).mean() <= 0.01


def test_test_loop_metrics(dev):
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
