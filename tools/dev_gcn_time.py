"""Dev: time the three fused graph-conv kernels at the bench shape (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pose2room_amd.p2rnet.modules.stgcn_layers import Graph
from pose2room_amd.p2rnet import gcn_op
dev = torch.device('cuda:0')
A = Graph().A
K, V = A.shape[0], A.shape[1]
tables = gcn_op.GraphTables(A)
N, T = int(os.environ.get('N', 32)), int(os.environ.get('T', 1024))
reps = int(os.environ.get('REPS', 3))
torch.manual_seed(0)
x = torch.randn(N, 64, T, V, device=dev, requires_grad=True)
w = (torch.randn(K * 64, 64, device=dev) / 8).requires_grad_(True)
b = (torch.randn(K * 64, device=dev) * 0.1).requires_grad_(True)
imp = (1 + 0.1 * torch.randn(K, V, V, device=dev)).requires_grad_(True)
At = torch.tensor(A, dtype=torch.float32, device=dev)
go = torch.randn(N, 64, T, V, device=dev)
for _ in range(reps):
    z = gcn_op.graph_conv(x, w, b, At * imp, tables)
    z.backward(go)
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
e[0].record(); z = gcn_op.graph_conv(x, w, b, At * imp, tables); e[1].record(); z.backward(go); e[2].record(); e[2].synchronize()
fl = 2 * 64 * K * 64 * N * T * V
print(f'fwd {e[0].elapsed_time(e[1]):.3f} ms ({fl / e[0].elapsed_time(e[1]) / 1e9:.1f} TF dense)  bwd {e[1].elapsed_time(e[2]):.3f} ms')
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(4):
        z = gcn_op.graph_conv(x, w, b, At * imp, tables)
        z.backward(go)
    torch.cuda.synchronize()
for e in sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:4]:
    print(f'{e.key[:48]:48s} n={e.count:3d} avg {e.device_time_total / e.count / 1e3:.3f} ms')
