"""G5b: the reference's `parse_predictions` with `use_3d_nms: False` (ap_helper.py:198-214 -> nms_2d_faster) on the
reference's own end points of the G3 cases (read back from g345_model.npz).  BUILD container only.

    python tests/golden/make_nms2d_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import make_model_golden as mm  # noqa: E402


def main():
    from net_utils.ap_helper import parse_predictions
    z = np.load(os.path.join(HERE, 'g345_model.npz'))
    out = {}
    for tag, B, T in (('g3u', 1, 768), ('g3f', 2, 512)):
        ep = {k: torch.from_numpy(z[f'{tag}_{k}']) for k in ['center', 'size', 'heading', 'objectness_scores', 'sem_cls_scores']}
        # the random-weight network gives several proposals the very same objectness probability, and np.argsort
        # (nms.py:15, quicksort) orders equal scores in no specified way: a ramp on the positive logit makes the scores
        # distinct, so that the pick order is a property of the algorithm
        K = ep['objectness_scores'].shape[1]
        ep['objectness_scores'] = ep['objectness_scores'].clone()
        ep['objectness_scores'][:, :, 1] += torch.linspace(0.0, 0.5, K).unsqueeze(0)
        out[f'{tag}_objectness_scores'] = ep['objectness_scores'].numpy()
        data = mm.make_batch(B, T, seed=100 + T)
        # the random-weight network's proposals overlap heavily: at a low nms_iou a handful of boxes per sample survive,
        # so the fixture also holds the masks at thresholds where the suppression is selective
        for iou in (0.25, 0.7, 0.9, 0.97):
            for old in (False, True):
                _, cfg = mm.build_ref(_METHODS, 'test', T, remove_far_box=False, use_3d_nms=False, use_old_type_nms=old,
                                      nms_iou=iou)
                eval_dict, parsed = parse_predictions(ep, data, cfg.eval_config)
                out[f'{tag}_pred_mask_2d_{int(round(iou * 100))}_{int(old)}'] = eval_dict['pred_mask']
                print(tag, iou, old, int(eval_dict['pred_mask'].sum()), 'kept of', eval_dict['pred_mask'].size)
        for i in range(B):
            assert np.unique(parsed['obj_prob'][i]).size == K, 'tied scores'
        out[f'{tag}_obj_prob'] = parsed['obj_prob']
        out[f'{tag}_pred_corners_3d'] = parsed['pred_corners_3d']
    np.savez_compressed(os.path.join(HERE, 'g5b_nms2d.npz'), **out)


if __name__ == '__main__':
    _METHODS = mm.import_reference()
    main()
