"""Dev: adjacency-gradient kernels -- statically scheduled (csrc/stgcn_gcn3_grad.hip) vs first generation: values, time."""
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
W = (torch.randn(K, 64, 64, generator=g) / 8).to(dev)
Wp = gcn_op.permute_planes(W)
ltot = t['gidx_r'].shape[0]
lib = _lib.lib()
st = _lib.current_stream(dev)
NB = 256
p_old = torch.empty(NB, ltot, V, device=dev)
p_new = torch.empty(NB, ltot, V, device=dev)

def old():
    _lib.check(lib.p2r_stgcn_gcn_coef_grad(N, T, V, K, tables.LkA_r, _lib.ptr(dz), _lib.ptr(x), _lib.ptr(W.contiguous()),
                                           _lib.ptr(t['nbr_r']), _lib.ptr(t['real_r']), NB, _lib.ptr(p_old), st), 'old')
def new():
    _lib.check(lib.p2r_stgcn_gcn3_coef_grad(N, T, V, K, ltot, _lib.ptr(x), _lib.ptr(dz), _lib.ptr(Wp), NB, _lib.ptr(p_new), st), 'new')

for name, fn in (('first generation', old), ('static schedule', new)):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record(); e1.synchronize()
    print(f'{name:18s} {e0.elapsed_time(e1) / 10:.3f} ms', flush=True)
a, b = p_old.double().sum(0), p_new.double().sum(0)
print('old vs new: max abs diff %.3e, scale %.3e' % ((a - b).abs().max().item(), a.abs().max().item()))
if N * T <= 4096:
    # fp64 definition: dA[k][v][w] = sum Y_k[c,t,v] dz[c,t,w]
    Y = torch.einsum('kcd,ndtv->nkctv', W.double(), x.double())
    dA = torch.einsum('nkctv,nctw->kvw', Y, dz.double())
    gi = t['gidx_r']
    want = torch.where(gi >= 0, dA.reshape(-1)[gi.clamp(min=0)], torch.zeros((), dtype=torch.float64, device=dev))
    print('new vs fp64: max abs diff %.3e, scale %.3e' % ((b - want).abs().max().item(), want.abs().max().item()))
