"""Detection loss (mirror of the reference's models/loss.py:8-189).

total = 10*vote + 5*objectness + 10*center + 10*size + 10*heading + sem_cls
(loss.py:167).  On GPU tensors the whole loss is one fused HIP op (csrc/det_loss.hip:
forward = per-sample partial sums + a combine launch, backward = one launch) instead of
~350 torch micro-kernels; `BoxNetDetectionLoss.composed` is the same computation as a
composition of torch ops around `nn_distance` -- what CPU tensors run (the oracle-backed
tests) and what the fused op is tested against.

Padded ground-truth rows: the reference compacts per sample (`per_gt_center[per_mask > 0]`,
loss.py:127-131) in a Python loop; here they are pushed out to a far sentinel so the minimum
runs over the valid rows only.  `object_assignment` therefore indexes the UNcompacted GT rows,
which equals the reference's index whenever `box_label_mask` is a prefix of ones -- the form
the reference loader produces (dataloader.py:113-121) and `synthetic.make_batch` keeps.
"""
import ctypes

import torch
from torch import nn
from torch.autograd import Function

from .. import _lib

from ..net_utils.nn_distance import nn_distance, huber_loss
from .registers import LOSSES

FAR_THRESHOLD = 0.6
NEAR_THRESHOLD = 0.3
GT_VOTE_FACTOR = 3                      # GT votes per joint
OBJECTNESS_CLS_WEIGHTS = [0.1, 0.9]     # larger weight on positive objectness
_FAR_AWAY = 1.0e18                      # sentinel coordinate of a padded GT row (squared: 1e36 < f32 max)

criterion_sem_cls = nn.CrossEntropyLoss(reduction='none')


class BaseLoss(object):
    def __init__(self, weight=1, device=0, cfg=None):
        self.weight = weight
        self.device = device
        self.origin_joint_id = cfg.dataset_config.origin_joint_id


@LOSSES.register_module
class Null(BaseLoss):
    """For modules whose loss is computed elsewhere."""
    def __call__(self, loss):
        return self.weight * torch.mean(loss)


@LOSSES.register_module
class BoxNetDetectionLoss(BaseLoss):
    def __init__(self, weight, device, cfg=None):
        super().__init__(weight, device, cfg)
        self.objectness_criterion = nn.CrossEntropyLoss(
            torch.Tensor(OBJECTNESS_CLS_WEIGHTS).to(self.device), reduction='none')

    # loss.py:42-88
    def compute_box_and_sem_cls_loss(self, est_data, gt_data, meta_data, config):
        object_assignment = meta_data['object_assignment']
        objectness_label = meta_data['objectness_label'].float()
        n_pos = torch.sum(objectness_label) + 1e-6
        box_label_mask = gt_data['box_label_mask']

        # centre: chamfer against ALL max_gt slots, zero padding included (only dist2 is masked)
        dist1, _, dist2, _ = nn_distance(est_data['center'], gt_data['center_label'])
        centroid_reg_loss1 = torch.sum(dist1 * objectness_label) / n_pos
        centroid_reg_loss2 = torch.sum(dist2 * box_label_mask) / (torch.sum(box_label_mask) + 1e-6)
        center_loss = (centroid_reg_loss1 + centroid_reg_loss2) / 2.

        gt_size = torch.gather(gt_data['size'], 1, object_assignment.unsqueeze(-1).repeat(1, 1, 3))
        size_loss = torch.mean(huber_loss(est_data['size'] - gt_size, delta=1.0), -1)
        size_loss = torch.sum(size_loss * objectness_label) / n_pos

        gt_heading = torch.gather(gt_data['heading'], 1, object_assignment.unsqueeze(-1).repeat(1, 1, 2))
        heading_loss = torch.mean(huber_loss(est_data['heading'] - gt_heading, delta=1.0), -1)
        heading_loss = torch.sum(heading_loss * objectness_label) / n_pos

        gt_cls_label = torch.gather(gt_data['sem_cls_label'], 1, object_assignment)
        sem_cls_loss = criterion_sem_cls(est_data['sem_cls_scores'].transpose(2, 1), gt_cls_label)
        sem_cls_loss = torch.sum(sem_cls_loss * objectness_label) / n_pos
        return center_loss, size_loss, heading_loss, sem_cls_loss

    # loss.py:90-115
    def compute_vote_loss(self, est_data, gt_data):
        j0 = self.origin_joint_id
        batch_size, num_seed, num_joints = est_data['seed_skeleton'].shape[:3]
        vote_xyz = est_data['vote_xyz']
        seed_inds = est_data['seed_inds'].long()

        seed_gt_votes_mask = torch.gather(gt_data['vote_label_mask'][..., j0], 1, seed_inds)
        seed_gt_votes = torch.gather(gt_data['vote_label'][:, :, j0], 1,
                                     seed_inds.view(batch_size, num_seed, 1).repeat(1, 1, 3 * GT_VOTE_FACTOR))
        seed_gt_votes = seed_gt_votes.view(batch_size, num_seed, GT_VOTE_FACTOR, 3)
        # slice, not the reference's list index `[:, :, [j0]]`: a list index uploads an index tensor from the host,
        # which synchronises the stream in the middle of the step
        seed_gt_votes = est_data['seed_skeleton'][:, :, j0:j0 + 1] + seed_gt_votes

        # which of the 3 GT votes is closest to any joint of the seed skeleton
        _, _, dist2, ind2 = nn_distance(seed_gt_votes.view(batch_size * num_seed, GT_VOTE_FACTOR, 3),
                                        est_data['seed_skeleton'].reshape(batch_size * num_seed, num_joints, 3))
        vote_argmin = torch.gather(ind2, dim=1, index=dist2.argmin(-1).unsqueeze(-1)).view(batch_size, num_seed, 1)
        seed_gt_votes = torch.gather(seed_gt_votes, 2, vote_argmin.unsqueeze(-1).repeat(1, 1, 1, 3)).squeeze(2)

        vote_loss = torch.mean(huber_loss(vote_xyz - seed_gt_votes, delta=1.0), -1)
        mask = seed_gt_votes_mask.float()
        return torch.sum(vote_loss * mask) / (torch.sum(mask) + 1e-6)

    # loss.py:117-150
    def compute_correspondence(self, est_data, gt_data):
        aggregated_vote_xyz = est_data['aggregated_vote_xyz']
        gt_center = gt_data['center_label'][:, :, 0:3]
        valid = gt_data['box_label_mask'] > 0
        # batched form of the reference's per-sample nn_distance over per_gt_center[per_mask > 0]
        masked_center = torch.where(valid.unsqueeze(-1), gt_center, torch.full_like(gt_center, _FAR_AWAY))
        dist1, object_assignment, _, _ = nn_distance(aggregated_vote_xyz.detach(), masked_center)

        B, K = aggregated_vote_xyz.shape[0], aggregated_vote_xyz.shape[1]
        euclidean_dist1 = torch.sqrt(dist1 + 1e-6)
        # the reference fills host-created zero tensors through boolean-mask assignment (loss.py:141-145): three
        # device->host synchronisations (mask -> nonzero) and two pageable uploads in the middle of the step; the
        # same values as pure device expressions keep the step asynchronous
        near = euclidean_dist1 < NEAR_THRESHOLD
        objectness_label = near.long()
        objectness_mask = (near | (euclidean_dist1 > FAR_THRESHOLD)).float()

        objectness_loss = self.objectness_criterion(est_data['objectness_scores'].transpose(2, 1), objectness_label)
        objectness_loss = torch.sum(objectness_loss * objectness_mask) / (torch.sum(objectness_mask) + 1e-6)
        return object_assignment, objectness_loss, objectness_label, objectness_mask

    # loss.py:152-189
    def __call__(self, est_data, gt_data, dataset_config):
        if est_data['vote_xyz'].is_cuda:
            # the HIP ops read raw pointers: bring every tensor to the dtype they expect (no-op for the tensors the
            # loader and the network produce), then route on what the fused kernel can hold
            est_data, gt_data = _canonical_dtypes(est_data, gt_data)
            if fused_supported(est_data, gt_data):
                return fused_detection_loss(est_data, gt_data, self.origin_joint_id)
        return self.composed(est_data, gt_data, dataset_config)

    def composed(self, est_data, gt_data, dataset_config):
        vote_loss = self.compute_vote_loss(est_data, gt_data)
        object_assignment, objectness_loss, objectness_label, objectness_mask = \
            self.compute_correspondence(est_data, gt_data)
        meta_data = {'object_assignment': object_assignment, 'objectness_label': objectness_label}
        center_loss, size_loss, heading_loss, sem_cls_loss = \
            self.compute_box_and_sem_cls_loss(est_data, gt_data, meta_data, dataset_config)
        loss = 10 * vote_loss + 5 * objectness_loss + 10 * center_loss + 10 * size_loss + \
            10 * heading_loss + sem_cls_loss

        total = objectness_label.shape[0] * objectness_label.shape[1]
        pos_ratio = torch.sum(objectness_label.float()) / float(total)
        neg_ratio = torch.sum(objectness_mask.float()) / float(total) - pos_ratio
        obj_pred_val = torch.argmax(est_data['objectness_scores'], 2)
        obj_acc = torch.sum((obj_pred_val == objectness_label.long()).float() * objectness_mask) / (
            torch.sum(objectness_mask) + 1e-6)
        return {'total': loss, 'vote_loss': vote_loss, 'objectness_loss': objectness_loss,
                'center_loss': center_loss, 'size_loss': size_loss, 'heading_loss': heading_loss,
                'sem_cls_loss': sem_cls_loss, 'pos_ratio': pos_ratio, 'neg_ratio': neg_ratio,
                'obj_acc': obj_acc}


_F32_EST = ('vote_xyz', 'objectness_scores', 'center', 'size', 'sem_cls_scores', 'seed_skeleton',
            'aggregated_vote_xyz')
_F32_GT = ('vote_label', 'center_label', 'box_label_mask', 'size', 'heading')
_MAX_GT, _DL_T, _DL_NPART = 32, 256, 12        # csrc/det_loss.hip: DL_MAXG, DL_T, DL_NPART


def _canonical_dtypes(est_data, gt_data):
    """Shallow copies of the two dicts with the loss inputs in the dtypes of the reference pipeline: f32 floats, the
    f64 heading head, int64 masks / labels.  Casts are differentiable (`.to`) and skipped when already right."""
    est, gt = dict(est_data), dict(gt_data)
    for d, names, dt in ((est, _F32_EST, torch.float32), (gt, _F32_GT, torch.float32), (est, ('heading',), torch.float64),
                         (gt, ('vote_label_mask', 'sem_cls_label'), torch.int64)):
        for n in names:
            if torch.is_tensor(d.get(n)) and d[n].dtype != dt:
                d[n] = d[n].to(dt)
    return est, gt


def fused_supported(est_data, gt_data):
    """True when the fused op (csrc/det_loss.hip) takes these tensors as they are: everything on one GPU, the dtypes
    the kernel reads through raw pointers (f32 everywhere except the f64 heading head and the int64 masks / labels /
    seed indices' integer type), at most 32 ground-truth slots and a per-sample working set inside its 64 KB of LDS.
    Anything else (autocast outputs, a loader emitting float64 boxes, more ground-truth slots) takes `composed`."""
    v = est_data['vote_xyz']
    if not v.is_cuda:
        return False
    dev = v.device
    for d, names, dt in ((est_data, _F32_EST, torch.float32), (gt_data, _F32_GT, torch.float32),
                         (est_data, ('heading',), torch.float64),
                         (gt_data, ('vote_label_mask', 'sem_cls_label'), torch.int64)):
        for n in names:
            t = d[n]
            if not torch.is_tensor(t) or t.dtype != dt or t.device != dev:
                return False
    if est_data['seed_inds'].device != dev or est_data['seed_inds'].dtype not in (torch.int32, torch.int64):
        return False
    G, K = gt_data['center_label'].shape[1], est_data['center'].shape[1]
    lds = _DL_T * _DL_NPART * 4 + _DL_T * 8 + G * K * 4 + 4 * G * 4 + G * 4 + 3 * K * 4
    return 0 < G <= _MAX_GT and lds <= 64 * 1024


_NAMES32 = ('vote_loss', 'objectness_loss', 'center_loss', 'size_loss', 'sem_cls_loss', 'pos_ratio', 'neg_ratio',
            'obj_acc')


_TERM_WEIGHTS = {}


def _term_weights(device):
    """[6] f64: weights of (vote, objectness, center, size, heading, sem_cls) in `total`
    (models/loss.py:167 of the reference: 10, 5, 10, 10, 10, 1)."""
    key = str(device)
    if key not in _TERM_WEIGHTS:
        _TERM_WEIGHTS[key] = torch.tensor([10.0, 5.0, 10.0, 10.0, 10.0, 1.0], dtype=torch.float64, device=device)
    return _TERM_WEIGHTS[key]


class _DetectionLoss(Function):
    """Inputs with gradient: vote_xyz, objectness_scores, center, size, heading (f64), sem_cls_scores.
    Outputs: the ten loss-dict entries in the reference's dtypes (`heading_loss`, `total` f64, the rest f32)."""

    @staticmethod
    def forward(ctx, vote_xyz, obj_scores, center, size, heading, sem_scores, seed_skeleton, seed_inds, agg_xyz,
                gt, j0):
        dev = vote_xyz.device
        vote_xyz, obj_scores, center, size, heading, sem_scores = (
            t.contiguous() for t in (vote_xyz, obj_scores, center, size, heading, sem_scores))
        seed_skeleton, agg_xyz = seed_skeleton.contiguous(), agg_xyz.contiguous()
        seed_inds = seed_inds.long().contiguous()
        if heading.dtype != torch.float64:
            raise RuntimeError("det_loss: heading must be float64 (the mixture head emits f64)")
        B, S, J = seed_skeleton.shape[:3]
        K, NC = center.shape[1], sem_scores.shape[2]
        T, G = gt['vote_label'].shape[1], gt['center_label'].shape[1]
        f32 = dict(dtype=torch.float32, device=dev)
        part, part64 = torch.empty((B, 12), **f32), torch.empty((B,), dtype=torch.float64, device=dev)
        out32, out64 = torch.empty((12,), **f32), torch.empty((2,), dtype=torch.float64, device=dev)
        g_vote = torch.empty((B, S, 3), **f32)
        g_obj, g_c1, g_c2, g_size = (torch.empty((B, K, d), **f32) for d in (2, 3, 3, 3))
        g_head = torch.empty((B, K, 2), dtype=torch.float64, device=dev)
        g_sem = torch.empty((B, K, NC), **f32)
        gts = [gt[k].contiguous() for k in ('vote_label', 'vote_label_mask', 'center_label', 'box_label_mask', 'size',
                                            'heading', 'sem_cls_label')]
        if gts[1].dtype != torch.int64 or gts[6].dtype != torch.int64:
            raise RuntimeError("det_loss: vote_label_mask and sem_cls_label must be int64")
        c = ctypes.c_float
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().p2r_det_loss_forward(
                B, S, J, T, K, G, NC, int(j0), c(NEAR_THRESHOLD), c(FAR_THRESHOLD), c(OBJECTNESS_CLS_WEIGHTS[0]),
                c(OBJECTNESS_CLS_WEIGHTS[1]), _lib.ptr(seed_skeleton), _lib.ptr(vote_xyz), _lib.ptr(seed_inds),
                _lib.ptr(gts[0]), _lib.ptr(gts[1]), _lib.ptr(agg_xyz), _lib.ptr(center), _lib.ptr(size),
                _lib.ptr(heading), _lib.ptr(obj_scores), _lib.ptr(sem_scores), _lib.ptr(gts[2]), _lib.ptr(gts[3]),
                _lib.ptr(gts[4]), _lib.ptr(gts[5]), _lib.ptr(gts[6]), _lib.ptr(part), _lib.ptr(part64),
                _lib.ptr(out32), _lib.ptr(out64), _lib.ptr(g_vote), _lib.ptr(g_obj), _lib.ptr(g_c1), _lib.ptr(g_c2),
                _lib.ptr(g_size), _lib.ptr(g_head), _lib.ptr(g_sem), _lib.current_stream(dev)), "det_loss_forward")
        ctx.save_for_backward(out32, g_vote, g_obj, g_c1, g_c2, g_size, g_head, g_sem)
        ctx.set_materialize_grads(False)      # unused loss terms arrive as None, not as zero tensors
        ctx.dims = (B, S, K, NC)
        outs = tuple(out32[i] for i in range(8)) + (out64[0], out64[1])
        ctx.mark_non_differentiable(outs[5], outs[6], outs[7])
        return outs

    @staticmethod
    def backward(ctx, d_vote, d_obj, d_center, d_size, d_sem, _pr, _nr, _acc, d_head, d_total):
        out32, g_vote, g_obj, g_c1, g_c2, g_size, g_head, g_sem = ctx.saved_tensors
        B, S, K, NC = ctx.dims
        dev = out32.device
        terms = [(d_vote, 10.0), (d_obj, 5.0), (d_center, 10.0), (d_size, 10.0), (d_head, 10.0), (d_sem, 1.0)]
        if d_total is not None and all(g is None for g, _ in terms):
            # the training step: only `total` is back-propagated -> one launch for the six term coefficients
            coef = (d_total.double() * _term_weights(dev)).contiguous()
        else:
            zero = torch.zeros((), dtype=torch.float64, device=dev)
            dt = d_total.double() if d_total is not None else zero
            coef = torch.stack([dt * w + (g.double() if g is not None else zero) for g, w in terms]).contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        o_vote = torch.empty((B, S, 3), **f32)
        o_obj, o_center, o_size = (torch.empty((B, K, d), **f32) for d in (2, 3, 3))
        o_head = torch.empty((B, K, 2), dtype=torch.float64, device=dev)
        o_sem = torch.empty((B, K, NC), **f32)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().p2r_det_loss_backward(
                B, S, K, NC, _lib.ptr(coef), _lib.ptr(out32), _lib.ptr(g_vote), _lib.ptr(g_obj), _lib.ptr(g_c1),
                _lib.ptr(g_c2), _lib.ptr(g_size), _lib.ptr(g_head), _lib.ptr(g_sem), _lib.ptr(o_vote), _lib.ptr(o_obj),
                _lib.ptr(o_center), _lib.ptr(o_size), _lib.ptr(o_head), _lib.ptr(o_sem), _lib.current_stream(dev)),
                "det_loss_backward")
        return o_vote, o_obj, o_center, o_size, o_head, o_sem, None, None, None, None, None


def fused_detection_loss(est_data, gt_data, origin_joint_id):
    """The loss dict of BoxNetDetectionLoss.__call__ from the fused op (GPU tensors)."""
    outs = _DetectionLoss.apply(est_data['vote_xyz'], est_data['objectness_scores'], est_data['center'],
                                est_data['size'], est_data['heading'], est_data['sem_cls_scores'],
                                est_data['seed_skeleton'].detach(), est_data['seed_inds'],
                                est_data['aggregated_vote_xyz'].detach(), gt_data, origin_joint_id)
    d = dict(zip(_NAMES32, outs[:8]))
    return {'total': outs[9], 'vote_loss': d['vote_loss'], 'objectness_loss': d['objectness_loss'],
            'center_loss': d['center_loss'], 'size_loss': d['size_loss'], 'heading_loss': outs[8],
            'sem_cls_loss': d['sem_cls_loss'], 'pos_ratio': d['pos_ratio'], 'neg_ratio': d['neg_ratio'],
            'obj_acc': d['obj_acc']}
