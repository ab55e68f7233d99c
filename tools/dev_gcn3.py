"""Dev: the statically scheduled graph-conv kernel (csrc/stgcn_gcn3.hip) against the second generation: values and
time, forward (column lists) and data-gradient (row lists, with and without the BatchNorm-backward epilogue)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pose2room_amd.p2rnet.modules.stgcn_layers import Graph
from pose2room_amd.p2rnet import gcn_op, gcn_tables
dev = torch.device('cuda:0')
A = Graph().A
K, V = A.shape[0], A.shape[1]
tables = gcn_op.GraphTables(A)
assert tables.gen3
t = tables.on(dev)
N, T = int(os.environ.get('N', 32)), int(os.environ.get('T', 1024))
g = torch.Generator().manual_seed(0)
x = torch.randn(N, 64, T, V, generator=g).to(dev)
W = (torch.randn(K, 64, 64, generator=g) / 8).to(dev)
Wp = gcn_op.permute_planes(W)
Aeff = (torch.tensor(A, dtype=torch.float32) * (1 + 0.1 * torch.randn(K, V, V, generator=g))).to(dev)
bias = torch.randn(64, V, generator=g).to(dev)
add = torch.randn(N, 64, T, V, generator=g).to(dev)
u = torch.randn(N, 64, T, V, generator=g).to(dev)
mask = (torch.rand(N, 64, T, V, generator=g) > 0.4).to(torch.uint8).to(dev)
fin = torch.randn(4, 64, generator=g).to(dev)
cc = gcn_tables.coefficients(Aeff, t['gidx_c']).contiguous()
cr = gcn_tables.coefficients(Aeff, t['gidx_r']).contiguous()
cases = {
    'forward+stats': dict(coef=cc, stream=t['stream_c'], bias_cv=bias, want_stats=True, form=0),
    'dgrad': dict(coef=cr, stream=t['stream_r'], bias_cv=None, form=1),
    'dgrad+addend': dict(coef=cr, stream=t['stream_r'], bias_cv=None, addend=add, form=1),
    'dgrad+addend+bnbwd': dict(coef=cr, stream=t['stream_r'], bias_cv=None, addend=add, want_stats=True, bwd=(u, mask, fin), form=1),
}
for name, kw in cases.items():
    kw = dict(kw)
    coef, stream, bias_cv = kw.pop('coef'), kw.pop('stream'), kw.pop('bias_cv')
    res = {}
    for gen3 in (False, True):
        gcn_op.USE_GEN3 = gen3
        fn = lambda: gcn_op._gcn2_forward(x, Wp, coef, stream, bias_cv, tables, **kw)
        out = fn()
        torch.cuda.synchronize()
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); e1.synchronize()
        res[gen3] = (out, e0.elapsed_time(e1) / 10)
    a, b = res[False][0], res[True][0]
    za, zb = (a[0], b[0]) if isinstance(a, tuple) else (a, b)
    err = (za - zb).abs().max().item() / za.abs().max().item()
    msg = f'{name:22s} gcn2 {res[False][1]:.3f} ms   gcn3 {res[True][1]:.3f} ms   z rel err {err:.2e}'
    if isinstance(a, tuple):
        if a[1].shape == b[1].shape:
            sa, sb = a[1].double().sum(0), b[1].double().sum(0)
        else:       # (sum, sum sq) pairs against (count, mean, M2) entries: compare the variances
            from pose2room_amd.p2rnet import bn_op
            M = x.shape[0] * x.shape[2] * x.shape[3]
            sa, sb = bn_op.moments(a[1], M)[1], bn_op.moments(b[1], M)[1]
        msg += f'   stats rel err {((sa - sb).abs().max() / sa.abs().max()).item():.2e}'
    print(msg, flush=True)
