"""Dev: group_points_grad at the P2RNet shape (B=32, C=256, n=512, 128 balls x 16): time, run-to-run bit equality."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pose2room_amd.pointnet2_ops import _ext
dev = torch.device('cuda:0')
B, C, n, P, S = 32, 256, 512, 128, 16
g = torch.Generator().manual_seed(0)
xyz = (torch.randn(B, n, 3, generator=g) * 0.5).to(dev)
inds = _ext.furthest_point_sampling(xyz, P)
new_xyz = _ext.gather_points(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
idx = _ext.ball_query(new_xyz, xyz, 0.3, S)
go = torch.randn(B, C, P, S, generator=g).to(dev)
a = _ext.group_points_grad(go, idx, n)
for _ in range(5):
    assert torch.equal(_ext.group_points_grad(go, idx, n), a)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    _ext.group_points_grad(go, idx, n)
e1.record(); e1.synchronize()
print(f'group_points_grad P2R_GG_GC={os.environ.get("P2R_GG_GC", "")}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us')
