"""Autograd front end of the pointnet2 ops (host-side mirror of the reference's
external/pointnet2_ops_lib/pointnet2_ops/pointnet2_utils.py).

Same public names, argument order and autograd contract as the reference:

  furthest_point_sample(xyz, npoint)            pointnet2_utils.py:34-65
  gather_operation(features, idx)               :68-101
  three_nn(unknown, known)                      :104-136
  three_interpolate(features, idx, weight)      :139-191
  grouping_operation(features, idx)             :194-240
  ball_query(radius, nsample, xyz, new_xyz)     :243-276   (note the argument order)
  QueryAndGroup / GroupAll                      :279-411

Index-producing ops are non-differentiable; gather / group / interpolate save
(idx, features) and call the matching `_grad` entry point with a contiguous
incoming gradient.  `_ext` is the C-ABI backed module `pose2room_amd.pointnet2_ops._ext`;
there is no JIT-compile or CPU fallback.
"""
import torch
import torch.nn as nn
from torch.autograd import Function

from . import _ext


class FurthestPointSampling(Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        """xyz (B,N,3) f32, npoint int -> (B,npoint) int32 indices."""
        inds = _ext.furthest_point_sampling(xyz, npoint)
        ctx.mark_non_differentiable(inds)
        return inds

    @staticmethod
    def backward(ctx, grad_out):
        return ()


furthest_point_sample = FurthestPointSampling.apply


class GatherOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        """features (B,C,N), idx (B,npoint) int32 -> (B,C,npoint)."""
        ctx.save_for_backward(idx, features)
        return _ext.gather_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        idx, features = ctx.saved_tensors
        grad_features = _ext.gather_points_grad(grad_out.contiguous(), idx, features.size(2))
        return grad_features, None


gather_operation = GatherOperation.apply


class ThreeNN(Function):
    @staticmethod
    def forward(ctx, unknown, known):
        """unknown (B,n,3), known (B,m,3) -> (dist (B,n,3) L2, idx (B,n,3) int32)."""
        dist2, idx = _ext.three_nn(unknown, known)
        dist = torch.sqrt(dist2)
        ctx.mark_non_differentiable(dist, idx)
        return dist, idx

    @staticmethod
    def backward(ctx, grad_dist, grad_idx):
        return ()


three_nn = ThreeNN.apply


class ThreeInterpolate(Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        """features (B,c,m), idx (B,n,3) int32, weight (B,n,3) -> (B,c,n)."""
        ctx.save_for_backward(idx, weight, features)
        return _ext.three_interpolate(features, idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight, features = ctx.saved_tensors
        grad_features = _ext.three_interpolate_grad(grad_out.contiguous(), idx, weight,
                                                    features.size(2))
        return grad_features, torch.zeros_like(idx), torch.zeros_like(weight)


three_interpolate = ThreeInterpolate.apply


class GroupingOperation(Function):
    @staticmethod
    def forward(ctx, features, idx):
        """features (B,C,N), idx (B,npoint,nsample) int32 -> (B,C,npoint,nsample)."""
        ctx.save_for_backward(idx, features)
        return _ext.group_points(features, idx)

    @staticmethod
    def backward(ctx, grad_out):
        idx, features = ctx.saved_tensors
        grad_features = _ext.group_points_grad(grad_out.contiguous(), idx, features.size(2))
        return grad_features, torch.zeros_like(idx)


grouping_operation = GroupingOperation.apply


class BallQuery(Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        """xyz (B,N,3), new_xyz (B,npoint,3) -> (B,npoint,nsample) int32."""
        idx = _ext.ball_query(new_xyz, xyz, radius, nsample)
        ctx.mark_non_differentiable(idx)
        return idx

    @staticmethod
    def backward(ctx, grad_out):
        return ()


ball_query = BallQuery.apply


def _resample_unique(idx, nsample):
    """Per ball: its distinct neighbour indices first (ascending), the remaining slots re-drawn uniformly from
    them -- what the reference's `sample_uniformly` double loop does (pointnet2_utils.py:321-330), for all balls at
    once.  Returns (idx int32 (B,P,S), number of distinct neighbours (B,P))."""
    srt, _ = torch.sort(idx.long(), dim=2)
    first = torch.ones_like(srt, dtype=torch.bool)
    first[..., 1:] = srt[..., 1:] != srt[..., :-1]
    count = first.sum(dim=2)                                              # (B,P) distinct neighbours
    # stable partition: distinct values keep their ascending order in slots [0, count)
    rank = torch.where(first, first.long().cumsum(dim=2) - 1, torch.full_like(srt, nsample))
    order = torch.argsort(rank, dim=2, stable=True)
    uniq = torch.gather(srt, 2, order)
    draw = (torch.rand(idx.shape, device=idx.device) * count.unsqueeze(-1)).long().clamp_(max=nsample - 1)
    slot = torch.arange(nsample, device=idx.device).view(1, 1, -1)
    pick = torch.where(slot < count.unsqueeze(-1), slot.expand_as(draw), draw)
    return torch.gather(uniq, 2, pick).to(idx.dtype), count


class QueryAndGroup(nn.Module):
    """Neighbourhoods of `new_xyz` inside `xyz`: ball query, then the neighbours' offsets from their centre
    (divided by the radius when `normalize_xyz`) and / or their features, stacked along the channel axis.
    Same constructor, argument order and return conventions as pointnet2_utils.py:279-361."""

    def __init__(self, radius, nsample, use_xyz=True, ret_grouped_xyz=False, normalize_xyz=False,
                 sample_uniformly=False, ret_unique_cnt=False):
        super().__init__()
        if ret_unique_cnt and not sample_uniformly:
            raise AssertionError("ret_unique_cnt needs sample_uniformly")
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz
        self.normalize_xyz = normalize_xyz
        self.sample_uniformly = sample_uniformly
        self.ret_unique_cnt = ret_unique_cnt

    def _offsets(self, xyz, new_xyz, idx):
        """(B,3,P,S): neighbour minus centre.  The subtraction / scaling happen in place on the freshly grouped tensor
        (an op output, never a view of an input), as in the reference."""
        local = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
        local -= new_xyz.transpose(1, 2).unsqueeze(-1)
        if self.normalize_xyz:
            local /= self.radius
        return local

    def forward(self, xyz, new_xyz, features=None):
        if features is None and not self.use_xyz:
            raise AssertionError("QueryAndGroup: no features given and use_xyz is off -- nothing to group")
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        unique_cnt = None
        if self.sample_uniformly:
            idx, unique_cnt = _resample_unique(idx, self.nsample)

        # the offsets cost a grouping launch and two passes: only when somebody consumes them
        local = self._offsets(xyz, new_xyz, idx) if (self.use_xyz or self.ret_grouped_xyz or features is None) else None
        if features is None:
            out = local
        else:
            grouped = grouping_operation(features, idx)
            out = torch.cat([local, grouped], dim=1) if self.use_xyz else grouped

        extras = ([local] if self.ret_grouped_xyz else []) + ([unique_cnt] if self.ret_unique_cnt else [])
        return (out, *extras) if extras else out


class GroupAll(nn.Module):
    """Single group holding every point.  Mirrors pointnet2_utils.py:364-411."""

    def __init__(self, use_xyz=True, ret_grouped_xyz=False):
        super().__init__()
        self.use_xyz = use_xyz
        self.ret_grouped_xyz = ret_grouped_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is not None:
            grouped_features = features.unsqueeze(2)
            new_features = torch.cat([grouped_xyz, grouped_features], dim=1) if self.use_xyz \
                else grouped_features
        else:
            new_features = grouped_xyz
        if self.ret_grouped_xyz:
            return new_features, grouped_xyz
        return new_features
