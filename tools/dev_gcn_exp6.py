"""Dev: time the backward graph-conv kernels (row-list form, as gcn_op calls them) for each libp2r_exp_*.so."""
import os, sys, ctypes, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pose2room_amd.p2rnet.modules.stgcn_layers import Graph
from pose2room_amd.p2rnet import gcn_tables
dev = torch.device('cuda:0')
A = Graph().A
N, T, V, K = 32, 1024, 53, 11
nbr, gidx, Lk = gcn_tables.build(A, transpose=True)
torch.manual_seed(0)
x = torch.randn(N, 64, T, V, device=dev); dz = torch.randn(N, 64, T, V, device=dev)
W = torch.randn(K, 64, 64, device=dev) / 8
coef = gcn_tables.coefficients(torch.tensor(A, dtype=torch.float32, device=dev), gidx.to(dev)).contiguous()
real = (gidx >= 0).float().to(dev).contiguous()
nb = nbr.to(dev); ltot = sum(Lk)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def t(fn):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8): fn()
    e1.record(); e1.synchronize(); return e0.elapsed_time(e1) / 8
for so in sorted(glob.glob(os.path.join(root, 'pose2room_amd', 'libp2r_exp_*.so'))):
    lib = ctypes.CDLL(so)
    LkA = (ctypes.c_int * K)(*Lk)
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    p = lambda a: ctypes.c_void_p(a.data_ptr())
    pw = torch.empty(256, K, 64, 64, device=dev); pb = torch.empty(256, 64, V, device=dev); pc = torch.empty(256, ltot, V, device=dev)
    dw = lambda: lib.p2r_stgcn_gcn_weight_grad(N, T, V, K, LkA, p(dz), p(x), p(nb), p(coef), 256, p(pw), p(pb), 1, st)
    dc = lambda: lib.p2r_stgcn_gcn_coef_grad(N, T, V, K, LkA, p(dz), p(x), p(W), p(nb), p(real), 256, p(pc), st)
    assert dw() == 0 and dc() == 0
    print(os.path.basename(so), f"dW {t(dw):.3f} ms  dcoef {t(dc):.3f} ms", "dW checksum %.6e" % pw.sum(0).double().abs().sum().item())
