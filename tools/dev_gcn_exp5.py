"""Dev: gcn forward with / without the statistics epilogue (main lib)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pose2room_amd.p2rnet.modules.stgcn_layers import Graph
from pose2room_amd.p2rnet import gcn_op, gcn_tables
dev = torch.device('cuda:0')
A = Graph().A
N, T, V, K = 32, 1024, 53, 11
tables = gcn_op.GraphTables(A)
t = tables.on(dev)
x = torch.randn(N, 64, T, V, device=dev); W = torch.randn(K * 64, 64, device=dev) / 8
bias = torch.randn(64, V, device=dev)
coef = gcn_tables.coefficients(torch.tensor(A, dtype=torch.float32, device=dev), t['gidx_c']).contiguous()
for ws in (False, True, False, True):
    fn = lambda: gcn_op._gcn_forward(x, W, t['nbr_c'], coef, tables.LkA_c, bias, tables, ws)
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); e1.synchronize()
    print('want_stats', ws, '%.3f ms' % (e0.elapsed_time(e1) / 10))
