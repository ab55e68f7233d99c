"""Dev: time ablation builds of the adjacency-gradient kernel (tools/ubench/dc/dc_*.so; ablation results are wrong by design)."""
import os, sys, ctypes, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pose2room_amd import _lib
from pose2room_amd.p2rnet.modules.stgcn_layers import Graph
from pose2room_amd.p2rnet import gcn_op, gcn_tables
dev = torch.device('cuda:0')
A = Graph().A
K, V = A.shape[0], A.shape[1]
tables = gcn_op.GraphTables(A)
t = tables.on(dev)
N, T = 32, 1024
x = torch.randn(N, 64, T, V, device=dev); dz = torch.randn(N, 64, T, V, device=dev)
W = torch.randn(K, 64, 64, device=dev) / 8
ltot = t['real_r'].shape[0]
part = torch.empty(256, ltot, V, device=dev)
for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ubench', 'dc', 'dc_*.so'))) * 2:
    lib = ctypes.CDLL(path)
    def call():
        rc = lib.p2r_stgcn_gcn_coef_grad(N, T, V, K, tables.LkA_r, _lib.ptr(dz), _lib.ptr(x), _lib.ptr(W), _lib.ptr(t['nbr_r']),
                                         _lib.ptr(t['real_r']), 256, _lib.ptr(part), _lib.current_stream(dev))
        assert rc == 0, rc
    for _ in range(3): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): call()
    e1.record(); e1.synchronize()
    print(f'{os.path.basename(path)[3:-3]:24s} {e0.elapsed_time(e1) / 10:.3f} ms', flush=True)
