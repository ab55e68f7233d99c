"""Dev: weight-gradient kernels -- statically scheduled (csrc/stgcn_gcn3_dw.hip) vs first generation: values, time."""
import os, sys
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
g = torch.Generator().manual_seed(0)
x = torch.randn(N, 64, T, V, generator=g).to(dev)
dz = torch.randn(N, 64, T, V, generator=g).to(dev)
Aeff = (torch.tensor(A, dtype=torch.float32) * (1 + 0.1 * torch.randn(K, V, V, generator=g))).to(dev)
coef_r = gcn_tables.coefficients(Aeff, t['gidx_r']).contiguous()
lib = _lib.lib()
st = _lib.current_stream(dev)
NB = 256
res = {}
for name in ('first generation', 'static schedule'):
    part = torch.empty(NB, K, 64, 64, device=dev)
    bpart = torch.empty(NB, 64, V, device=dev)
    def fn():
        if name == 'first generation':
            _lib.check(lib.p2r_stgcn_gcn_weight_grad(N, T, V, K, tables.LkA_r, _lib.ptr(dz), _lib.ptr(x), _lib.ptr(t['nbr_r']),
                                                     _lib.ptr(coef_r), NB, _lib.ptr(part), _lib.ptr(bpart), 1, st), 'old')
        else:
            _lib.check(lib.p2r_stgcn_gcn3_weight_grad(N, T, V, K, coef_r.shape[0], _lib.ptr(x), _lib.ptr(dz), _lib.ptr(coef_r), NB,
                                                      _lib.ptr(part), _lib.ptr(bpart), st), 'new')
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); e1.synchronize()
    print(f'{name:18s} {e0.elapsed_time(e1) / 10:.3f} ms', flush=True)
    res[name] = (_lib.sum_leading(part, tr64=True).double(), _lib.sum_leading(bpart).double())
a, b = res['first generation'], res['static schedule']
print('dW  old vs new: max abs diff %.3e, scale %.3e' % ((a[0] - b[0]).abs().max().item(), a[0].abs().max().item()))
print('dbias old vs new: max abs diff %.3e, scale %.3e' % ((a[1] - b[1]).abs().max().item(), a[1].abs().max().item()))
if N * T <= 4096:
    U = torch.einsum('nctv,kvw->nkctw', x.double(), Aeff.double())           # x . A_k
    dW = torch.einsum('nctw,nkdtw->kcd', dz.double(), U)                      # [k][c][ci]
    print('new vs fp64: max abs diff %.3e, scale %.3e' % ((b[0] - dW).abs().max().item(), dW.abs().max().item()))
    print('dbias vs fp64: %.3e' % (b[1] - dz.double().sum((0, 2))).abs().max().item())
