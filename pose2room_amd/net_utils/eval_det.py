"""VOC-style detection AP over oriented 3D boxes.

Mirror of the mesh-free path of the reference's net_utils/eval_det.py: `voc_ap` (:93-123),
`eval_det_cls_wo_mesh` (:259-343) and `eval_det_multiprocessing_wo_mesh` (:424-473), same inputs
({img_id: [(classname, corners (8,3), score)]} / {img_id: [(classname, corners)]}), same outputs
(rec, prec, ap dicts keyed by class), same matching rule (detections in `np.argsort(-score)` order, each
takes the ground truth of highest IoU -- first one on ties -- and is a true positive if that IoU >
ovthresh and the ground truth is still free).

The reference evaluates one `box3d_iou` Python call per (detection, ground truth) pair inside a
`multiprocessing.Pool(10)` over classes.  Here all pairs of one class are clipped in a single batched
tensor pass (`box_util.box3d_iou_matrix`), so no process pool is needed; only the greedy matching
walks the sorted detections.  The mesh variants (:133-257, :355-422, voxelised-mesh IoU) belong to a
branch P2RNet never enables (`evaluate_mesh=False`, test_epoch.py:22) and are not provided.
"""
import numpy as np
import torch

from .box_util import box3d_iou_pairs


def voc_ap(rec, prec, use_07_metric=False):
    """AP from a recall/precision curve (eval_det.py:93-123): 11-point VOC07 or area under the
    monotone precision envelope."""
    if use_07_metric:
        ap = 0.
        for t in np.arange(0., 1.1, 0.1):
            reached = rec >= t
            ap = ap + (np.max(prec[reached]) if np.sum(reached) != 0 else 0) / 11.
        return ap
    mrec = np.concatenate(([0.], rec, [1.]))
    mpre = np.concatenate(([0.], prec, [0.]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    step = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[step + 1] - mrec[step]) * mpre[step + 1])


def get_iou_obb(bb1, bb2):
    """eval_det.py:86-88."""
    return float(box3d_iou_pairs(np.asarray(bb1, dtype=np.float64)[None], np.asarray(bb2, dtype=np.float64)[None])[0][0])


def _pair_ious(det_boxes, det_img, gt_boxes_by_img, device):
    """IoU of every detection with every ground truth of its own image, one batched pass.
    Returns {detection index: np.ndarray (n_gt_of_that_image,)}."""
    d_idx, pairs_d, pairs_g, spans = [], [], [], {}
    for d, img in enumerate(det_img):
        g = gt_boxes_by_img[img]
        if len(g) == 0:
            continue
        spans[d] = (len(pairs_d), len(g))
        for j in range(len(g)):
            pairs_d.append(det_boxes[d])
            pairs_g.append(g[j])
    if not pairs_d:
        return {}
    c1 = torch.as_tensor(np.asarray(pairs_d, dtype=np.float64), device=device)
    c2 = torch.as_tensor(np.asarray(pairs_g, dtype=np.float64), device=device)
    iou = box3d_iou_pairs(c1, c2)[0].cpu().numpy()
    return {d: iou[o:o + n] for d, (o, n) in spans.items()}


def eval_det_cls_wo_mesh(pred, gt, ovthresh=0.25, use_07_metric=False, get_iou_func=None, device='cpu'):
    """One class (eval_det.py:259-343).  pred {img_id: [(corners, score)]}, gt {img_id: [corners]} ->
    (rec (nd,), prec (nd,), ap).  `get_iou_func` other than None / get_iou_obb falls back to one call
    per pair, as the reference does."""
    recs, npos = {}, 0
    for img_id in gt.keys():
        boxes = np.array(gt[img_id])
        recs[img_id] = {'bbox': boxes, 'det': [False] * len(boxes)}
        npos += len(boxes)
    for img_id in pred.keys():
        if img_id not in gt:
            recs[img_id] = {'bbox': np.array([]), 'det': []}

    image_ids, confidence, BB = [], [], []
    for img_id in pred.keys():
        entry = pred[img_id]
        if isinstance(entry, tuple):            # (corners (n,8,3), scores (n,)) arrays of ap_helper.PredMapCls
            image_ids += [img_id] * len(entry[1])
            confidence.append(np.asarray(entry[1]))
            BB.append(np.asarray(entry[0]).reshape(-1, 8, 3))
        else:
            for box, score in entry:
                image_ids.append(img_id)
                confidence.append(np.asarray([score]))
                BB.append(np.asarray(box)[None])
    confidence = np.concatenate(confidence) if confidence else np.array([])
    BB = np.concatenate(BB) if BB else np.array([])
    order = np.argsort(-confidence)
    BB = BB[order, ...]
    image_ids = [image_ids[x] for x in order]

    nd = len(image_ids)
    batched = get_iou_func is None or get_iou_func is get_iou_obb
    if batched:
        ious = _pair_ious(BB.astype(float) if nd else BB, image_ids,
                          {k: v['bbox'].astype(float) if v['bbox'].size else [] for k, v in recs.items()}, device)
    tp, fp = np.zeros(nd), np.zeros(nd)
    for d in range(nd):
        R = recs[image_ids[d]]
        ovmax, jmax = -np.inf, -1
        if R['bbox'].size > 0:
            if batched:
                ov = ious[d]
            else:
                ov = np.array([get_iou_func(BB[d].astype(float), R['bbox'][j].astype(float)) for j in range(R['bbox'].shape[0])])
            # the reference keeps the first j whose IoU is strictly above the running maximum (NaN never wins)
            ok = ~np.isnan(ov)
            if ok.any():
                jmax = int(np.argmax(np.where(ok, ov, -np.inf)))
                ovmax = ov[jmax]
        if ovmax > ovthresh and not R['det'][jmax]:
            tp[d] = 1.
            R['det'][jmax] = 1
        else:
            fp[d] = 1.

    fp, tp = np.cumsum(fp), np.cumsum(tp)
    with np.errstate(divide='ignore', invalid='ignore'):
        rec = tp / float(npos)
    prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
    return rec, prec, voc_ap(rec, prec, use_07_metric)


def eval_det_multiprocessing_wo_mesh(pred_all, gt_all, ovthresh=0.25, use_07_metric=False, get_iou_func=None,
                                     device='cpu'):
    """All classes (eval_det.py:424-473); the name is kept for the call sites, the work is batched
    tensor math instead of a process pool.  A class that is predicted anywhere also gets (empty) ground
    truth lists for those images, as in the reference (:443-446), so it appears in the result."""
    pred, gt = {}, {}
    for img_id in pred_all.keys():
        entry = pred_all[img_id]
        if hasattr(entry, 'class_arrays'):      # ap_helper.PredMapCls: the same lists, read as arrays
            for classname, boxes, scores in entry.class_arrays():
                if len(scores) == 0:
                    continue
                pred.setdefault(classname, {})[img_id] = (boxes, scores)
                gt.setdefault(classname, {}).setdefault(img_id, [])
            continue
        for classname, bbox, score in entry:
            pred.setdefault(classname, {}).setdefault(img_id, [])
            gt.setdefault(classname, {}).setdefault(img_id, [])
            pred[classname][img_id].append((bbox, score))
    for img_id in gt_all.keys():
        for classname, bbox in gt_all[img_id]:
            gt.setdefault(classname, {}).setdefault(img_id, [])
            gt[classname][img_id].append(bbox)

    rec, prec, ap = {}, {}, {}
    for classname in gt.keys():
        if classname in pred:
            rec[classname], prec[classname], ap[classname] = eval_det_cls_wo_mesh(
                pred[classname], gt[classname], ovthresh, use_07_metric, get_iou_func, device)
        else:
            rec[classname] = prec[classname] = ap[classname] = 0
    return rec, prec, ap
