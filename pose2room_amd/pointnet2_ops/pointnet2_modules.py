"""PointNet++ set-abstraction / feature-propagation modules on the HIP ops.

Host-side mirror of the reference's
external/pointnet2_ops_lib/pointnet2_ops/pointnet2_modules.py: same class names,
constructor keywords, forward signatures, return tuples and `state_dict` keys
(`mlp_module.{0,2}.weight|bias` for bn=False, `mlps.{i}.*`, `mlp.*`), so
reference checkpoints load.

P2RNet instantiates exactly one of them -- `PointnetSAModuleVotes(npoint=128,
radius=0.3, nsample=16, mlp=[256,256,256], use_xyz=False, normalize_xyz=True,
bn=False)` (models/p2rnet/modules/proposal_net.py:63-71) -- the others are kept
for the pointnet2_ops call surface.

With autograd off (the evaluation path) and a configuration that allows it (`fused=True`,
bn=False, use_xyz=False, pooling='max', 256-wide two-layer MLP, nsample=16)
`PointnetSAModuleVotes` replaces the chain ball_query -> group(xyz) -> group(features) ->
2x(conv1x1 + ReLU) -> max-pool by one fused MFMA kernel (`pose2room_amd.pointnet2_ops.fused`),
which returns the same tensors.
"""
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pointnet2_utils


def build_shared_mlp(mlp_spec: List[int], bn: bool = True) -> nn.Sequential:
    """1x1 Conv2d (+BN) + ReLU stack; layer indices match pointnet2_modules.py:9-19."""
    layers = []
    for cin, cout in zip(mlp_spec[:-1], mlp_spec[1:]):
        layers.append(nn.Conv2d(cin, cout, kernel_size=1, bias=not bn))
        if bn:
            layers.append(nn.BatchNorm2d(cout))
        layers.append(nn.ReLU(True))
    return nn.Sequential(*layers)


def _sample_centres(xyz, npoint, inds=None):
    """FPS (or given indices) -> (inds, new_xyz (B,npoint,3))."""
    if inds is None:
        inds = pointnet2_utils.furthest_point_sample(xyz, npoint)
    new_xyz = pointnet2_utils.gather_operation(xyz.transpose(1, 2).contiguous(), inds)
    return inds, new_xyz.transpose(1, 2).contiguous()


def _pool_samples(x, pooling="max"):
    """(B,C,P,S) -> (B,C,P) over the sample axis."""
    if pooling == "max":
        x = F.max_pool2d(x, kernel_size=[1, x.size(3)])
    elif pooling == "avg":
        x = F.avg_pool2d(x, kernel_size=[1, x.size(3)])
    else:
        raise ValueError(pooling)
    return x.squeeze(-1)


class _PointnetSAModuleBase(nn.Module):
    """Multi-scale SA forward shared by the MSG variants (pointnet2_modules.py:22-73)."""

    def __init__(self):
        super().__init__()
        self.npoint = None
        self.groupers = None
        self.mlps = None

    def forward(self, xyz: torch.Tensor, features: Optional[torch.Tensor]
                ) -> Tuple[torch.Tensor, torch.Tensor]:
        new_xyz = _sample_centres(xyz, self.npoint)[1] if self.npoint is not None else None
        pooled = []
        for grouper, mlp in zip(self.groupers, self.mlps):
            pooled.append(_pool_samples(mlp(grouper(xyz, new_xyz, features))))
        return new_xyz, torch.cat(pooled, dim=1)


class PointnetSAModuleMSG(_PointnetSAModuleBase):
    """Set abstraction with multi-scale grouping (pointnet2_modules.py:76-118)."""

    def __init__(self, npoint, radii, nsamples, mlps, bn=True, use_xyz=True, sample_uniformly=False):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint = npoint
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, mlp_spec in zip(radii, nsamples, mlps):
            self.groupers.append(
                pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz,
                                              sample_uniformly=sample_uniformly)
                if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
            if use_xyz:
                mlp_spec[0] += 3
            self.mlps.append(build_shared_mlp(mlp_spec, bn))


class PointnetSAModule(PointnetSAModuleMSG):
    """Single-scale set abstraction (pointnet2_modules.py:121-148)."""

    def __init__(self, mlp, npoint=None, radius=None, nsample=None, bn=True, use_xyz=True):
        super().__init__(mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample], bn=bn,
                         use_xyz=use_xyz)


class PointnetSAModuleVotes(nn.Module):
    """Set abstraction that also returns the sampled indices
    (pointnet2_modules.py:152-261).  forward(xyz (B,N,3), features (B,C,N),
    inds=None) -> (new_xyz (B,npoint,3), new_features (B,mlp[-1],npoint),
    inds (B,npoint) int32[, unique_cnt])."""

    def __init__(self, *, mlp: List[int], npoint: int = None, radius: float = None,
                 nsample: int = None, bn: bool = True, use_xyz: bool = True, pooling: str = 'max',
                 sigma: float = None, normalize_xyz: bool = False, sample_uniformly: bool = False,
                 ret_unique_cnt: bool = False, fused: bool = True):
        super().__init__()
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        self.pooling = pooling
        self.use_xyz = use_xyz
        self.sigma = sigma if sigma is not None else self.radius / 2
        self.normalize_xyz = normalize_xyz
        self.ret_unique_cnt = ret_unique_cnt
        self.bn = bn
        self.sample_uniformly = sample_uniformly
        self.fused = fused

        if npoint is not None:
            self.grouper = pointnet2_utils.QueryAndGroup(
                radius, nsample, use_xyz=use_xyz, ret_grouped_xyz=True, normalize_xyz=normalize_xyz,
                sample_uniformly=sample_uniformly, ret_unique_cnt=ret_unique_cnt)
        else:
            self.grouper = pointnet2_utils.GroupAll(use_xyz, ret_grouped_xyz=True)

        mlp_spec = mlp
        if use_xyz and len(mlp_spec) > 0:
            mlp_spec[0] += 3
        self.mlp_module = build_shared_mlp(mlp_spec, bn=bn)

    def _can_fuse(self, features):
        from . import fused as fused_ops
        return (self.fused and fused_ops.available() and self.npoint is not None
                and not self.bn and not self.use_xyz
                and self.pooling == 'max' and not self.sample_uniformly and not self.ret_unique_cnt
                and features is not None and features.is_cuda and len(self.mlp_module) == 4
                and fused_ops.supports(self.mlp_module, self.nsample))

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None, inds: torch.Tensor = None):
        if inds is not None:
            assert inds.shape[1] == self.npoint
        if self.npoint is not None:
            inds, new_xyz = _sample_centres(xyz, self.npoint, inds)
        else:
            new_xyz = None

        if self._can_fuse(features):
            from . import fused as fused_ops
            new_features = fused_ops.sa_votes(xyz, new_xyz, features, self.radius, self.nsample,
                                              self.mlp_module)
            return new_xyz, new_features, inds

        grouped = self.grouper(xyz, new_xyz, features)
        if self.ret_unique_cnt:
            grouped_features, grouped_xyz, unique_cnt = grouped
        else:
            grouped_features, grouped_xyz = grouped

        new_features = self.mlp_module(grouped_features)  # (B, mlp[-1], npoint, nsample)
        if self.pooling in ('max', 'avg'):
            new_features = _pool_samples(new_features, self.pooling)
        elif self.pooling == 'rbf':
            # radial-basis weighting of the samples (pointnet2_modules.py:250-255)
            rbf = torch.exp(-1 * grouped_xyz.pow(2).sum(1, keepdim=False) / (self.sigma ** 2) / 2)
            new_features = (torch.sum(new_features * rbf.unsqueeze(1), -1, keepdim=True)
                            / float(self.nsample)).squeeze(-1)
        else:
            new_features = new_features.squeeze(-1)

        if self.ret_unique_cnt:
            return new_xyz, new_features, inds, unique_cnt
        return new_xyz, new_features, inds


class PointnetSAModuleMSGVotes(nn.Module):
    """Multi-scale SA returning the sampled indices (pointnet2_modules.py:264-343)."""

    def __init__(self, *, mlps: List[List[int]], npoint: int, radii: List[float],
                 nsamples: List[int], bn: bool = True, use_xyz: bool = True,
                 sample_uniformly: bool = False):
        super().__init__()
        assert len(mlps) == len(nsamples) == len(radii)
        self.npoint = npoint
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, mlp_spec in zip(radii, nsamples, mlps):
            self.groupers.append(
                pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz,
                                              sample_uniformly=sample_uniformly)
                if npoint is not None else pointnet2_utils.GroupAll(use_xyz))
            if use_xyz:
                mlp_spec[0] += 3
            self.mlps.append(build_shared_mlp(mlp_spec, bn=bn))

    def forward(self, xyz: torch.Tensor, features: torch.Tensor = None, inds: torch.Tensor = None):
        if self.npoint is not None:
            inds, new_xyz = _sample_centres(xyz, self.npoint, inds)
        else:
            new_xyz = None
        pooled = [_pool_samples(mlp(grouper(xyz, new_xyz, features)))
                  for grouper, mlp in zip(self.groupers, self.mlps)]
        return new_xyz, torch.cat(pooled, dim=1), inds


class PointnetFPModule(nn.Module):
    """Feature propagation by inverse-distance three-NN interpolation
    (pointnet2_modules.py:346-406)."""

    def __init__(self, mlp, bn=True):
        super().__init__()
        self.mlp = build_shared_mlp(mlp, bn=bn)

    def forward(self, unknown, known, unknow_feats, known_feats):
        if known is not None:
            dist, idx = pointnet2_utils.three_nn(unknown, known)
            dist_recip = 1.0 / (dist + 1e-8)
            weight = dist_recip / torch.sum(dist_recip, dim=2, keepdim=True)
            interpolated = pointnet2_utils.three_interpolate(known_feats, idx, weight)
        else:
            interpolated = known_feats.expand(*(list(known_feats.size()[0:2]) + [unknown.size(1)]))
        new_features = interpolated if unknow_feats is None else \
            torch.cat([interpolated, unknow_feats], dim=1)
        return self.mlp(new_features.unsqueeze(-1)).squeeze(-1)
