"""Workload for the PMC passes over the pointnet2 / nn_distance / NMS / vote-aggregation kernels: each kernel a few
times at the P2RNet shapes (B=32, N=512, npoint=128, nsample=16, C=256) and at the stress shapes of SURVEY.md 8(d)
(B=8, N=54,272, npoint=2048, nsample=32, C=64; build-defined, NOT exercised by the model).  SHAPE=p2r|stress."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pose2room_amd.pointnet2_ops import _ext
from pose2room_amd.net_utils.nn_distance import nn_distance
dev = torch.device('cuda:0')
shape = os.environ.get('SHAPE', 'p2r')
B, N, P, S, C = (32, 512, 128, 16, 256) if shape == 'p2r' else (8, 54272, 2048, 32, 64)
g = torch.Generator().manual_seed(0)
xyz = (torch.randn(B, N, 3, generator=g) * (0.5 if shape == 'p2r' else 2.0)).to(dev)
feats = torch.randn(B, C, N, generator=g).to(dev)
for _ in range(3):
    inds = _ext.furthest_point_sampling(xyz, P)
    new_xyz = _ext.gather_points(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
    idx = _ext.ball_query(new_xyz, xyz, 0.3, S)
    grouped = _ext.group_points(feats, idx)
    _ext.group_points_grad(grouped, idx, N)
    q = xyz[:, :min(N, 8192)].contiguous()
    d, i3 = _ext.three_nn(q, new_xyz)
    w = torch.softmax(-d, 2)
    out = _ext.three_interpolate(feats[:, :, :P].contiguous(), i3, w)
    _ext.three_interpolate_grad(out, i3, w, P)
    if shape == 'p2r':
        nn_distance(torch.randn(B * 512, 3, 3, device=dev), torch.randn(B * 512, 53, 3, device=dev))
    else:
        nn_distance(torch.randn(B, 2048, 3, device=dev), torch.randn(B, 4096, 3, device=dev))
torch.cuda.synchronize()
