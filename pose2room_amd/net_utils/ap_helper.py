"""Prediction parsing for evaluation, on the device.

Mirror of the reference's net_utils/ap_helper.py:133-255 (`parse_predictions`),
:257-292 (`parse_groundtruths`), :294-350 (`assembly_pred_map_cls`, mesh-free
branches) and :402-428 (`assembly_gt_map_cls`): same signatures, same returned dict
keys, shapes and dtypes (NumPy arrays, as the reference returns), same ordering of
`batch_pred_map_cls` (class-major, then proposal index).

What changes is where the work happens.  The reference copies the predictions to the
host and then runs Python loops: `get_3d_box` per proposal (B*K calls), a SciPy
Delaunay point-in-hull test per box for the far-box filter, a NumPy min/max per box
for the AABBs and a NumPy NMS per sample.  Here everything up to and including the
NMS keep mask is tensor math on the GPU in float64 -- oriented boxes from
(size, heading, centre) in closed form (utils/pc_utils.py:22-27,50-67,
utils/tools.py:33-51), the far-box test as the closed-form point-in-oriented-box
predicate the Delaunay test implements, AABBs by min/max over corners -- and one
batched launch of the HIP NMS kernel (nms.py:41-119).  Only the final results cross
PCIe, once.
"""
import numpy as np
import torch

from . import nms as nms_hip

# corner order of utils/tools.py:33-51 get_box_corners: signs of (v0, v1, v2)
_CORNER_SIGNS = [(-1, -1, -1), (+1, -1, -1), (+1, +1, -1), (-1, +1, -1),
                 (-1, -1, +1), (+1, -1, +1), (+1, +1, +1), (-1, +1, +1)]


class PredMapCls(object):
    """One sample's `batch_pred_map_cls` entry -- the reference's list of (class, corners (8,3), score) tuples
    (ap_helper.py:294-350: class-major, then proposal index) -- held as the arrays it is made of.  It has the list's
    length, iteration and indexing (tuples are made on demand; `==` is object identity -- the reference's lists hold
    arrays and have no usable equality either), so everything written against
    the reference's lists works; `eval_det.eval_det_multiprocessing_wo_mesh` reads the arrays directly
    (`class_arrays`) instead of walking ~2,800 tuples per sample.  Building the tuples eagerly was 60 % of the
    evaluation's wall time."""
    __slots__ = ('corners', 'scores', 'classes', 'num_class')

    def __init__(self, corners, scores, classes=None, num_class=None):
        """per-class proposals: corners (n,8,3), scores (n, num_class), classes None;
        otherwise: corners (n,8,3), scores (n,), classes (n,) int."""
        self.corners, self.scores, self.classes = corners, scores, classes
        self.num_class = num_class if classes is None else None

    def __len__(self):
        n = self.corners.shape[0]
        return n * self.num_class if self.classes is None else n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        if self.classes is None:
            ii, j = divmod(i, self.corners.shape[0])
            return (ii, self.corners[j], self.scores[j, ii])
        return (int(self.classes[i]), self.corners[i], self.scores[i])

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def class_arrays(self):
        """-> iterable of (class, corners (n,8,3), scores (n,)) in the list's order"""
        if self.classes is None:
            for ii in range(self.num_class):
                yield ii, self.corners, self.scores[:, ii]
        else:
            for c in dict.fromkeys(int(v) for v in self.classes):       # first-appearance order
                sel = np.nonzero(self.classes == c)[0]
                yield c, self.corners[sel], self.scores[sel]


def softmax(x):
    """NumPy softmax over the last axis (net_utils/libs.py:75-80)."""
    probs = np.exp(x - np.max(x, axis=len(x.shape) - 1, keepdims=True))
    probs /= np.sum(probs, axis=len(x.shape) - 1, keepdims=True)
    return probs


def head2rot_t(heading):
    """heading (...,) f64 -> rotation matrices (...,3,3) (utils/pc_utils.py:50-67)."""
    c, s = torch.cos(heading), torch.sin(heading)
    R = torch.zeros(heading.shape + (3, 3), dtype=heading.dtype, device=heading.device)
    R[..., 0, 0] = c
    R[..., 0, 2] = -s
    R[..., 1, 1] = 1
    R[..., 2, 0] = s
    R[..., 2, 2] = c
    return R


def boxes_to_corners(size, heading, center):
    """size (...,3) f32, heading (...) f64, center (...,3) f32 -> corners (...,8,3) f64.
    get_3d_box (utils/pc_utils.py:22-27): vectors = diag(size / 2) . R, corners = centre
    +- v0 +- v1 +- v2.  `size / 2` is evaluated in float32 like NumPy does there."""
    half = (size / 2.).to(torch.float64)
    vectors = half.unsqueeze(-1) * head2rot_t(heading)            # (...,3,3): row i = half_i * R[i]
    signs = torch.tensor(_CORNER_SIGNS, dtype=torch.float64, device=size.device)   # (8,3)
    c = center.to(torch.float64).unsqueeze(-2)                    # (...,1,3)
    # reference order: ((centre +- v0) +- v1) +- v2
    out = c + signs[:, 0, None] * vectors[..., None, 0, :]
    out = out + signs[:, 1, None] * vectors[..., None, 1, :]
    out = out + signs[:, 2, None] * vectors[..., None, 2, :]
    return out


def _far_box_mask(pred_size, pred_heading, pred_center, hips, contact_dist):
    """nonempty mask (B,K): box size sane and at least one hip position inside the box
    enlarged by `contact_dist` on every half-extent (ap_helper.py:183-196; the reference
    tests membership in the convex hull of the enlarged box's corners)."""
    ok_size = ~((pred_size < 0.01).any(-1) | (pred_size > 10).any(-1))
    R = head2rot_t(pred_heading)                                                      # (B,K,3,3)
    half = (pred_size / 2. + contact_dist).to(torch.float64)                            # (B,K,3)
    rel = hips.to(torch.float64)[:, None, :, :] - pred_center.to(torch.float64)[:, :, None, :]   # (B,K,T,3)
    local = torch.einsum('bktd,bkid->bkti', rel, R)                                     # coords along the box axes
    inside = (local.abs() <= half[:, :, None, :]).all(-1).any(-1)
    return ok_size & inside


def _to_host(tensors):
    """NumPy copies of device tensors with ONE synchronisation: all copies are issued into pinned staging tensors
    (PyTorch's caching host allocator) on the current stream, then the stream is waited for once -- `.cpu()` per
    tensor is a synchronisation each."""
    if not tensors or not tensors[0].is_cuda:
        return [t.cpu().numpy() for t in tensors]
    outs = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in tensors]
    for o, t in zip(outs, tensors):
        o.copy_(t, non_blocking=True)
    torch.cuda.current_stream(tensors[0].device).synchronize()
    return [o.numpy() for o in outs]


def parse_predictions(est_data, gt_data, config_dict, return_device=False):
    """est_data: end_points of `P2RNet.generate`; gt_data: batch (needs 'input_joints' when
    remove_far_box).  Returns (eval_dict{'pred_mask' (B,K) uint8}, parsed{'pred_corners_3d'
    (B,K,8,3) f64, 'sem_cls_probs' (B,K,C) f32, 'obj_prob' (B,K) f32, 'pred_sem_cls' (B,K) i64})."""
    dataset_config = config_dict['dataset_config']
    pred_center = est_data['center'].detach()
    pred_size = torch.exp(est_data['size']).detach()
    pred_sin_cos = est_data['heading'].detach()
    pred_heading = torch.atan2(pred_sin_cos[..., 0], pred_sin_cos[..., 1])              # f64

    sem_scores = est_data['sem_cls_scores'].detach()
    if config_dict['sample_cls']:
        sem_cls_probs_t = torch.softmax(sem_scores, dim=-1)
        pred_sem_cls_t = torch.distributions.Categorical(sem_cls_probs_t).sample()
    else:
        pred_sem_cls_t = torch.argmax(sem_scores, -1)
        sem_cls_probs_t = None
    obj_logits = est_data['objectness_scores'].detach()
    obj_shift = obj_logits - obj_logits.max(-1, keepdim=True).values                   # libs.softmax, on device
    obj_e = torch.exp(obj_shift)
    obj_prob_t = (obj_e / obj_e.sum(-1, keepdim=True))[:, :, 1]                          # (B,K) f32

    corners = boxes_to_corners(pred_size, pred_heading, pred_center)                    # (B,K,8,3) f64
    bsize, K = pred_center.shape[0], pred_center.shape[1]

    nonempty = torch.ones((bsize, K), dtype=torch.bool, device=pred_center.device)
    if config_dict['remove_far_box']:
        hips = gt_data['input_joints'][:, :, dataset_config.origin_joint_id, 0:3]
        nonempty = _far_box_mask(pred_size, pred_heading, pred_center, hips.to(pred_center.device),
                                 dataset_config.contact_dist_thresh)

    mins = corners.min(dim=2).values
    maxs = corners.max(dim=2).values
    score = obj_prob_t.to(torch.float64).unsqueeze(-1)
    if not config_dict['use_3d_nms']:
        # 2D (x,z) NMS (nms_2d_faster, ap_helper.py:198-214) == 3D NMS on boxes with a unit y extent, bit for bit
        # (nms.boxes_2d_as_3d; pinned by G2's 2-D pick lists and the `use_3d_nms: False` fixture of G7)
        boxes = nms_hip.boxes_2d_as_3d(torch.cat([mins[..., 0:1], mins[..., 2:3], maxs[..., 0:1], maxs[..., 2:3], score], -1))
        same_cls = False
    elif not config_dict['cls_nms']:
        boxes = torch.cat([mins, maxs, score], -1)
        same_cls = False
    else:
        boxes = torch.cat([mins, maxs, score, pred_sem_cls_t.to(torch.float64).unsqueeze(-1)], -1)
        same_cls = True
    keep, _, npick = nms_hip.nms_3d_batched(boxes.contiguous(), config_dict['nms_iou'],
                                            config_dict['use_old_type_nms'], same_cls, valid=nonempty,
                                            return_pick=True)
    assert bool((npick > 0).all()), "NMS kept no box for some sample (reference: assert len(pick) > 0)"

    if return_device:
        return {'pred_mask': keep}, {'pred_corners_3d': corners, 'obj_prob': obj_prob_t,
                                     'pred_sem_cls': pred_sem_cls_t, 'sem_cls_scores': sem_scores}
    keep_h, corners_h, sem_h, obj_h, cls_h = _to_host(
        [keep, corners, sem_scores if sem_cls_probs_t is None else sem_cls_probs_t, obj_prob_t, pred_sem_cls_t])
    sem_cls_probs = softmax(sem_h) if sem_cls_probs_t is None else sem_h
    return ({'pred_mask': keep_h},
            {'pred_corners_3d': corners_h, 'sem_cls_probs': sem_cls_probs, 'obj_prob': obj_h, 'pred_sem_cls': cls_h})


def parse_groundtruths(gt_data, config_dict):
    """GT boxes -> corners (ap_helper.py:257-292); masked-out slots stay zero.
    Returns HOST arrays: {'sem_cls_label' (B,G) int64 ndarray, 'gt_corners_3d' (B,G,8,3) float64 ndarray,
    'box_label_mask' (B,G) ndarray} -- the reference keeps 'sem_cls_label' as the input tensor and calls `.item()` per
    box downstream; here the three results cross PCIe in one transfer and `assembly_gt_map_cls` accepts either."""
    gt_center = gt_data['center_label'][:, :, 0:3].detach()
    gt_size = torch.exp(gt_data['size']).detach()
    gt_heading = torch.atan2(gt_data['heading'][..., 0], gt_data['heading'][..., 1]).detach()
    mask = gt_data['box_label_mask'].detach()
    corners = boxes_to_corners(gt_size, gt_heading.to(torch.float64), gt_center)
    corners = corners * (mask != 0).to(torch.float64)[:, :, None, None]
    labels_h, corners_h, mask_h = _to_host([gt_data['sem_cls_label'].detach(), corners, mask])
    return {'sem_cls_label': labels_h, 'gt_corners_3d': corners_h, 'box_label_mask': mask_h}


def assembly_pred_map_cls(eval_dict, parsed_predictions, config_dict, mesh_outputs=None, voxel_size=0.047):
    """Per sample list of (class, corners (8,3), confidence) (ap_helper.py:294-350)."""
    assert mesh_outputs is None, "mesh evaluation is outside the hot path"
    pred_corners_3d = parsed_predictions['pred_corners_3d']
    sem_cls_probs = parsed_predictions['sem_cls_probs']
    obj_prob = parsed_predictions['obj_prob']
    pred_mask = eval_dict['pred_mask']
    pred_sem_cls = parsed_predictions['pred_sem_cls']
    bsize, n_prop = pred_sem_cls.shape
    out = []
    conf = config_dict['conf_thresh']
    for i in range(bsize):
        # the reference's nested comprehensions (class-major, then proposal index) as a lazy sequence over the kept
        # proposals' arrays; the scores are one array product per sample
        keep = np.nonzero((pred_mask[i] == 1) & (obj_prob[i] > conf))[0]
        boxes = pred_corners_3d[i][keep]
        if config_dict['per_class_proposal']:
            scores = sem_cls_probs[i][keep] * obj_prob[i][keep, None]          # (n_keep, num_class)
            out.append(PredMapCls(boxes, scores, num_class=config_dict['dataset_config'].num_class))
        else:
            out.append(PredMapCls(boxes, obj_prob[i][keep], classes=np.asarray(pred_sem_cls[i][keep])))
    eval_dict['batch_pred_map_cls'] = out
    return eval_dict


def assembly_gt_map_cls(parsed_gts, mesh_outputs=None, voxel_size=0.047):
    """Per sample list of (class, corners (8,3)) (ap_helper.py:402-428)."""
    assert mesh_outputs is None, "mesh evaluation is outside the hot path"
    sem_cls_label = parsed_gts['sem_cls_label']
    if torch.is_tensor(sem_cls_label):          # one transfer instead of an `.item()` synchronisation per box
        sem_cls_label = sem_cls_label.detach().cpu().numpy()
    corners = parsed_gts['gt_corners_3d']
    mask = parsed_gts['box_label_mask']
    labels = sem_cls_label.tolist()
    return [[(labels[i][j], corners[i, j]) for j in np.nonzero(mask[i] == 1)[0]]
            for i in range(sem_cls_label.shape[0])]


class APCalculator(object):
    """Accumulates per-scan predictions / ground truths and computes AP, mAP, recall and AR
    (net_utils/ap_helper.py:24-128, mesh-free branch; the class P2RNet's test loop instantiates with
    `evaluate_mesh=False`, test_epoch.py:22).  Same constructor, `step`, `compute_metrics`, `reset`
    and result keys ('<cls> Average Precision', 'mAP', '<cls> Recall', 'AR')."""

    def __init__(self, ap_iou_thresh=0.25, class2type_map=None, evaluate_mesh=False, device='cpu'):
        if evaluate_mesh:
            raise NotImplementedError("mesh AP (voxelised-mesh IoU, ap_helper.py:86-128) is outside the P2RNet path")
        self.ap_iou_thresh = ap_iou_thresh
        self.class2type_map = class2type_map
        self.evaluate_mesh = evaluate_mesh
        self.device = device
        self.reset()

    def step(self, batch_pred_map_cls, batch_gt_map_cls):
        """batch_pred_map_cls [[(cls, corners (8,3), score), ...], ...]; batch_gt_map_cls [[(cls, corners), ...], ...]."""
        bsize = len(batch_pred_map_cls)
        assert bsize == len(batch_gt_map_cls)
        for i in range(bsize):
            self.gt_map_cls[self.scan_cnt] = batch_gt_map_cls[i]
            self.pred_map_cls[self.scan_cnt] = batch_pred_map_cls[i]
            self.scan_cnt += 1

    def compute_metrics(self):
        from .eval_det import eval_det_multiprocessing_wo_mesh
        rec, _, ap = eval_det_multiprocessing_wo_mesh(self.pred_map_cls, self.gt_map_cls, ovthresh=self.ap_iou_thresh,
                                                      device=self.device)
        name = (lambda k: self.class2type_map[k]) if self.class2type_map else str
        ret = {}
        for key in sorted(ap.keys()):
            ret['%s Average Precision' % name(key)] = ap[key]
        ret['mAP'] = np.mean([v for v in ap.values() if not np.isnan(v)])
        recalls = []
        for key in sorted(ap.keys()):
            try:
                last = rec[key][-1]
            except (TypeError, IndexError):      # class without predictions (rec == 0) or without detections
                last = 0
            ret['%s Recall' % name(key)] = last
            recalls.append(last)
        ret['AR'] = np.mean([v for v in recalls if not np.isnan(v)])
        return ret

    def reset(self):
        self.gt_map_cls = {}
        self.pred_map_cls = {}
        self.scan_cnt = 0
