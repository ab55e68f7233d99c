"""Workload for the HBM-traffic PMC passes: the three graph-conv kernels at the bench shape plus one kernel of
known traffic for calibrating the counters (bn_stats reads a 444.6 MB tensor exactly once)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pose2room_amd.p2rnet.modules.stgcn_layers import Graph
from pose2room_amd.p2rnet import gcn_op, bn_op
dev = torch.device('cuda:0')
A = Graph().A
K, V = A.shape[0], A.shape[1]
tables = gcn_op.GraphTables(A)
N, T = 32, 1024
torch.manual_seed(0)
x = torch.randn(N, 64, T, V, device=dev, requires_grad=True)
w = (torch.randn(K * 64, 64, device=dev) / 8).requires_grad_(True)
b = (torch.randn(K * 64, device=dev) * 0.1).requires_grad_(True)
imp = (1 + 0.1 * torch.randn(K, V, V, device=dev)).requires_grad_(True)
At = torch.tensor(A, dtype=torch.float32, device=dev)
go = torch.randn(N, 64, T, V, device=dev)
from pose2room_amd.p2rnet import math_mode          # P2R_MATH=split16 (environment): the split16 graph-conv kernels
for _ in range(3):
    z, part = gcn_op.graph_conv(x, w, b, At * imp, tables, want_stats=True)
    z.backward(go)
    bn_op._stats_partial(go)
if math_mode.split16():      # the temporal conv's two split launches on the same tensors
    from pose2room_amd.p2rnet import tconv_op
    W3 = (torch.randn(3, 64, 64, device=dev) / 8)
    st = tconv_op.SplitTaps(*tconv_op.split_taps(W3))
    sc, sh = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev)
    fin = torch.stack([torch.zeros(64, device=dev), torch.ones(64, device=dev), sc, sh]).contiguous()
    xd = x.detach()
    for _ in range(3):
        tconv_op._tconvh(xd, sc, sh, st, None, True)
        tconv_op._tconvh(go, None, None, st, None, True, bwd=(xd, fin), x_word=math_mode.range_word(go))
torch.cuda.synchronize()
