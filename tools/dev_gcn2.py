"""Dev: second-generation graph-conv forward / data-gradient kernel against the first generation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pose2room_amd.p2rnet.modules.stgcn_layers import Graph
from pose2room_amd.p2rnet import gcn_op, gcn_tables
dev = torch.device('cuda:0')
A = Graph().A
K, V = A.shape[0], A.shape[1]
tables = gcn_op.GraphTables(A)
t = tables.on(dev)
At = torch.tensor(A, dtype=torch.float32, device=dev)


def run(N, T, timing=False):
    torch.manual_seed(N * 1000 + T)
    x = torch.randn(N, 64, T, V, device=dev)
    W = torch.randn(K, 64, 64, device=dev) / 8
    imp = 1 + 0.1 * torch.randn(K, V, V, device=dev)
    Aeff = At * imp
    bias_cv = torch.randn(64, V, device=dev)
    for form, nbr, gidx, LkA, sched in (('col', t['nbr_c'], t['gidx_c'], tables.LkA_c, t['stream_c']),
                                        ('row', t['nbr_r'], t['gidx_r'], tables.LkA_r, t['stream_r'])):
        coef = gcn_tables.coefficients(Aeff, gidx).contiguous()
        z1, s1 = gcn_op._gcn_forward(x, W.reshape(K * 64, 64), nbr, coef, LkA, bias_cv, tables, True)
        Wp = gcn_op.permute_planes(W)
        z2, s2 = gcn_op._gcn2_forward(x, Wp, coef, sched, bias_cv, tables, True)
        torch.cuda.synchronize()
        err = (z1 - z2).abs().max().item()
        serr = (s1.double().sum(0) - s2.double().sum(0)).abs().max().item() / s1.double().sum(0).abs().max().item()
        print(f'N={N} T={T} {form}: max|z1-z2| = {err:.3e} (scale {z1.abs().max().item():.2f}) stats rel {serr:.2e}', flush=True)
        if timing:
            for f, args in ((gcn_op._gcn_forward, (x, W.reshape(K * 64, 64), nbr, coef, LkA, bias_cv, tables, True)),
                            (gcn_op._gcn2_forward, (x, Wp, coef, sched, bias_cv, tables, True))):
                for _ in range(3):
                    f(*args)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    f(*args)
                e1.record(); e1.synchronize()
                print(f'   {f.__name__}: {e0.elapsed_time(e1) / 10:.3f} ms', flush=True)


for N, T in ((1, 16), (1, 1), (2, 7), (1, 20), (3, 33), (2, 130), (2, 64)):
    run(N, T)
run(int(os.environ.get('N', 32)), int(os.environ.get('T', 1024)), timing=True)
