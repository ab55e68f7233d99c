"""Dev: time p2r_gather_frames / _grad at the bench shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pose2room_amd.p2rnet import seed_op
dev = torch.device('cuda:0')
torch.manual_seed(0)
B, C, T, J, S = 32, 64, 1024, 53, 512
x = torch.randn(B, C, T, J, device=dev, requires_grad=True)
inds = torch.stack([torch.sort(torch.randperm(T, device=dev)[:S])[0] for _ in range(B)])
go = torch.randn(B, S, C * J, device=dev)
def t(fn, reps=40):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
rows = seed_op.seed_rows(x, inds)
print('forward', t(lambda: seed_op.seed_rows(x, inds)))
def bwd():
    x.grad = None
    rows.backward(go, retain_graph=True)
print('backward', t(bwd))
