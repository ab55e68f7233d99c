"""Dev: time ablation builds of csrc/stgcn_gcn3_dw.hip (tools/ubench/w3/w3_*.so; results of ablations are wrong by design)."""
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
N, T = int(os.environ.get('N', 32)), int(os.environ.get('T', 1024))
x = torch.randn(N, 64, T, V, device=dev)
dz = torch.randn(N, 64, T, V, device=dev)
coef_r = gcn_tables.coefficients(torch.tensor(A, dtype=torch.float32, device=dev), t['gidx_r']).contiguous()
part = torch.empty(256, K, 64, 64, device=dev)
bpart = torch.empty(256, 64, V, device=dev)
here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ubench', 'w3')
for path in sorted(glob.glob(os.path.join(here, 'w3_*.so'))):
    lib = ctypes.CDLL(path)
    def call():
        rc = lib.p2r_stgcn_gcn3_weight_grad(N, T, V, K, coef_r.shape[0], _lib.ptr(x), _lib.ptr(dz), _lib.ptr(coef_r), 256,
                                            _lib.ptr(part), _lib.ptr(bpart), _lib.current_stream(dev))
        assert rc == 0, rc
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        call()
    e1.record(); e1.synchronize()
    print(f'{os.path.basename(path)[3:-3]:16s} {e0.elapsed_time(e1) / 10:.3f} ms', flush=True)
