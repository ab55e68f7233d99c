"""Diagnostic: gradients of the fused graph conv at the bench shape against the plain formulation evaluated in
batch chunks (no tensor beyond 2^31 bytes on the reference side)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pose2room_amd.p2rnet.modules.stgcn_layers import Graph
from pose2room_amd.p2rnet import gcn_op

dev = torch.device('cuda:0')
A = Graph().A
K, V = A.shape[0], A.shape[1]
tables = gcn_op.GraphTables(A)
N, T = int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 1024
g = torch.Generator().manual_seed(1)
x = torch.randn(N, 64, T, V, generator=g).to(dev)
w = (torch.randn(K * 64, 64, generator=g) / 8).to(dev)
b = (torch.randn(K * 64, generator=g) * 0.1).to(dev)
At = torch.tensor(A, dtype=torch.float32, device=dev)
imp = (1 + 0.1 * torch.randn(K, V, V, generator=g)).to(dev)
go = torch.randn(N, 64, T, V, generator=g).to(dev)

xd, wd, bd, idv = (t.clone().requires_grad_(True) for t in (x, w, b, imp))
z = gcn_op.graph_conv(xd, wd, bd, At * idv, tables)
z.backward(go)


def ref(xc, wr, br, ir):
    y = torch.nn.functional.conv2d(xc, wr.view(K * 64, 64, 1, 1), br)
    n, kc, t, v = y.shape
    return torch.einsum('nkctv,kvw->nctw', y.view(n, K, kc // K, t, v), At * ir)


def chunked(chunk):
    wr, br, ir = (t.clone().requires_grad_(True) for t in (w, b, imp))
    gx, zs = [], []
    for i in range(0, N, chunk):
        xc = x[i:i + chunk].clone().requires_grad_(True)
        zc = ref(xc, wr, br, ir)
        zc.backward(go[i:i + chunk])
        gx.append(xc.grad)
        zs.append(zc.detach())
    return torch.cat(zs), torch.cat(gx), wr.grad, br.grad, ir.grad


for chunk in (2, N):
    zr, gx, gw, gb, gi = chunked(chunk)
    for name, a, r in (('z', z.detach(), zr), ('dx', xd.grad, gx), ('dW', wd.grad, gw), ('db', bd.grad, gb), ('dimp', idv.grad, gi)):
        s = r.abs().max().item()
        print(f'chunk {chunk:3d} {name:5s} max err / scale = {(a - r).abs().max().item() / s:.3e}   scale {s:.3e}')
