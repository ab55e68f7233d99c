"""Dev: time the BatchNorm kernels at the bench shape through the autograd op (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from pose2room_amd.p2rnet import bn_op
dev = torch.device('cuda:0')
N, T, V = 32, 1024, 53
torch.manual_seed(0)
x = torch.randn(N, 64, T, V, device=dev, requires_grad=True)
res = torch.randn(N, 64, T, V, device=dev, requires_grad=True)
go = torch.randn(N, 64, T, V, device=dev)
bn = torch.nn.BatchNorm2d(64).to(dev).train()
for _ in range(2):
    bn_op.fused_bn_act(x, bn, res, relu=True).backward(go)
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(5):
        bn_op.fused_bn_act(x, bn, res, relu=True).backward(go)
    torch.cuda.synchronize()
for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:5]:
    print(f'{e.key[:60]:60s} n={e.count:3d} avg {e.device_time_total / e.count:.1f} us')
