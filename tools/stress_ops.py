"""Dev: the pointnet2 / nn_distance / NMS kernels at the stress shapes of SURVEY.md §8(d) (NOT exercised by the
P2RNet model: N = T*J = 54,272 points, npoint = 2048, nsample = 32, C = 64, B = 8), event-timed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pose2room_amd.pointnet2_ops import _ext
from pose2room_amd.net_utils.nn_distance import nn_distance
dev = torch.device('cuda:0')
B, N, P, S, C = 8, 54272, 2048, 32, 64
g = torch.Generator().manual_seed(0)
xyz = (torch.randn(B, N, 3, generator=g) * 2.0).to(dev)
feats = torch.randn(B, C, N, generator=g).to(dev)
def t(fn, reps=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / reps
inds = _ext.furthest_point_sampling(xyz, P)
new_xyz = _ext.gather_points(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
idx = _ext.ball_query(new_xyz, xyz, 0.3, S)
grouped = _ext.group_points(feats, idx)
unknown = new_xyz
ms = t(lambda: _ext.furthest_point_sampling(xyz, P), 3)
print(f'fps               {ms * 1e3:9.1f} us   {B * (P - 1) * N / ms / 1e6:7.2f} G pairs/s ({8 * B * (P - 1) * N / ms / 1e6 / 78600 * 100:.2f} % of fp32 VALU peak at 8 FLOP/pair; one workgroup per cloud, {P - 1} dependent rounds)')
ms = t(lambda: _ext.ball_query(new_xyz, xyz, 0.3, S))
print(f'ball_query        {ms * 1e3:9.1f} us   <= {B * P * N / ms / 1e6:7.2f} G pairs/s (early exit once a ball is full)')
ms = t(lambda: _ext.group_points(feats, idx))
by = 4.0 * B * C * P * S * 2 + 4.0 * B * P * S
print(f'group_points      {ms * 1e3:9.1f} us   {by / ms / 1e6:7.1f} GB/s algorithmic')
ms = t(lambda: _ext.group_points_grad(grouped, idx, N))
by = 4.0 * B * C * (P * S + N)
print(f'group_points_grad {ms * 1e3:9.1f} us   {by / ms / 1e6:7.1f} GB/s algorithmic')
ms = t(lambda: _ext.three_nn(xyz[:, :8192].contiguous(), new_xyz))
print(f'three_nn (8192 queries x 2048 known) {ms * 1e3:9.1f} us   {B * 8192 * P / ms / 1e6:7.2f} G pairs/s')
pc1 = torch.randn(B, 2048, 3, generator=g).to(dev); pc2 = torch.randn(B, 4096, 3, generator=g).to(dev)
ms = t(lambda: nn_distance(pc1, pc2))
print(f'nn_distance (2048 x 4096)            {ms * 1e3:9.1f} us   {B * 2048 * 4096 / ms / 1e6:7.2f} G pairs/s')
