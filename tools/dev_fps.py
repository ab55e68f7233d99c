"""Dev: time furthest point sampling at the reference's stress shape (8 x 54272 -> 2048)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pose2room_amd.pointnet2_ops import _ext
dev = torch.device('cuda:0')
for (b, n, m) in [(8, 54272, 2048), (1, 54272, 2048), (32, 54272, 2048), (8, 20000, 1024), (8, 16384, 2048)]:
    xyz = torch.randn(b, n, 3, device=dev)
    for _ in range(2):
        _ext.furthest_point_sampling(xyz, m)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        _ext.furthest_point_sampling(xyz, m)
    e1.record(); e1.synchronize()
    print(f'fps b={b} n={n} m={m}: {e0.elapsed_time(e1) / 3:.3f} ms', flush=True)
