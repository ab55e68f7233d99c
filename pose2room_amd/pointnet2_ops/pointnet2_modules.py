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


# ---- spatial-transformer grouping (pointnet2_modules.py:408-538; not instantiated by P2RNet) ----------------------------
def weights_init(m):
    """`module.apply` hook of the reference (pointnet2_modules.py:408-419): Conv2d and Linear layers start from zero
    weights and biases -- so a fresh STN3d predicts the identity transform.  (Conv1d layers do not match the rule.)"""
    kind = type(m).__name__
    if 'Conv2d' in kind or 'Linear' in kind:
        for name in ('weight', 'bias'):
            p = getattr(m, name, None)
            if p is not None and hasattr(p, 'data'):
                nn.init.constant_(p.data, 0.0)


class STN3d(nn.Module):
    """PointNet-style T-Net over the `num_points` samples of each proposal: three pointwise Conv1d + BatchNorm + ReLU
    (3 -> 64 -> 128 -> 256), max over the samples, three Linear layers (256 -> 128 -> 64 -> 12, the first two with
    BatchNorm + ReLU); the 12 outputs + identity are a 3 x 4 matrix [R | t] applied to the proposal's points.
    Same attribute names (state_dict keys) and forward contract as pointnet2_modules.py:421-467:
    grouped_xyz (B, 3, P, num_points) -> (B, 3, P, num_points)."""

    def __init__(self, num_points=2500):
        super().__init__()
        self.num_points = num_points
        self.conv1, self.conv2, self.conv3 = nn.Conv1d(3, 64, 1), nn.Conv1d(64, 128, 1), nn.Conv1d(128, 256, 1)
        self.mp1 = nn.MaxPool1d(num_points)
        self.fc1, self.fc2, self.fc3 = nn.Linear(256, 128), nn.Linear(128, 64), nn.Linear(64, 12)
        self.relu = nn.ReLU(inplace=True)
        self.bn1, self.bn2, self.bn3 = nn.BatchNorm1d(64), nn.BatchNorm1d(128), nn.BatchNorm1d(256)
        self.bn4, self.bn5 = nn.BatchNorm1d(128), nn.BatchNorm1d(64)
        self.apply(weights_init)

    def forward(self, grouped_xyz):
        B, _, P, _ = grouped_xyz.shape
        pts = grouped_xyz.transpose(2, 1).contiguous().view(B * P, 3, self.num_points)
        h = pts
        for conv, bn in ((self.conv1, self.bn1), (self.conv2, self.bn2), (self.conv3, self.bn3)):
            h = self.relu(bn(conv(h)))
        h = self.mp1(h).squeeze(2)
        h = self.relu(self.bn4(self.fc1(h)))
        h = self.relu(self.bn5(self.fc2(h)))
        eye = torch.eye(3, 4, dtype=torch.float32, device=grouped_xyz.device).view(1, 12)
        m = (self.fc3(h) + eye).view(B * P, 3, 4)
        moved = torch.bmm(m[:, :, :3], pts) + m[:, :, 3:]
        return moved.view(B, P, 3, -1).transpose(1, 2)


class STN_Group(nn.Module):
    """Ball-query grouping around the proposals, the neighbour offsets turned by each proposal's heading into its
    canonical frame and then by the learned STN3d transform (pointnet2_modules.py:470-538).
    forward(xyz (B,N,3), features (B,C,N), new_xyz (B,P,3), orientations (B,P)) ->
    (grouped_xyz (B,3,P,nsample), grouped_features[, unique_cnt])."""

    def __init__(self, radius: float = None, nsample: int = None, use_xyz: bool = True, normalize_xyz: bool = False,
                 sample_uniformly: bool = False, ret_unique_cnt: bool = False):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.normalize_xyz, self.ret_unique_cnt = normalize_xyz, ret_unique_cnt
        self.grouper = pointnet2_utils.QueryAndGroup(radius, nsample, use_xyz=use_xyz, ret_grouped_xyz=True,
                                                     normalize_xyz=normalize_xyz, sample_uniformly=sample_uniformly,
                                                     ret_unique_cnt=ret_unique_cnt)
        self.stn3d = STN3d(num_points=nsample)

    def forward(self, xyz, features=None, new_xyz=None, orientations=None):
        got = self.grouper(xyz, new_xyz, features)
        grouped_features, local = got[0], got[1]
        B, P = orientations.shape
        c, s = torch.cos(orientations), torch.sin(orientations)
        zero, one = torch.zeros_like(c), torch.ones_like(c)
        # rotation about z by -heading: rows (c, s, 0), (-s, c, 0), (0, 0, 1)
        rot = torch.stack([c, s, zero, -s, c, zero, zero, zero, one], dim=-1).view(B * P, 3, 3)
        local = torch.bmm(rot, local.transpose(1, 2).contiguous().view(B * P, 3, -1))
        local = local.view(B, P, 3, -1).transpose(1, 2).contiguous()
        out = (self.stn3d(local), grouped_features)
        return out + (got[2],) if self.ret_unique_cnt else out
