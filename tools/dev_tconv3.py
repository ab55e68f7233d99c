"""Dev: temporal-conv kernels -- third generation (csrc/stgcn_tconv3.hip) vs second: values, time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pose2room_amd.p2rnet import tconv_op
dev = torch.device('cuda:0')
N, T, V = int(os.environ.get('N', 32)), int(os.environ.get('T', 1024)), 53
g = torch.Generator().manual_seed(0)
x = torch.randn(N, 64, T, V, generator=g).to(dev)
z = torch.randn(N, 64, T, V, generator=g).to(dev)
scale, shift = (torch.rand(64, generator=g) + 0.5).to(dev), torch.randn(64, generator=g).to(dev)
bias = torch.randn(64, generator=g).to(dev)
fin = torch.randn(4, 64, generator=g).to(dev)
for taps in (3, 1):
    W3 = (torch.randn(taps, 64, 64, generator=g) / 8).to(dev)
    cases = {'fwd+stats': dict(scale=scale, shift=shift, bias=bias, want_stats=True),
             'dgrad': dict(scale=None, shift=None, bias=None),
             'dgrad+bnbwd': dict(scale=None, shift=None, bias=None, want_stats=True, bwd=(z, fin))}
    for name, kw in cases.items():
        res = {}
        for gen3 in (False, True):
            tconv_op.USE_GEN3 = gen3
            fn = lambda: tconv_op._tconv(x, kw['scale'], kw['shift'], W3, kw['bias'], kw.get('want_stats', False), kw.get('bwd'))
            o = fn(); torch.cuda.synchronize()
            for _ in range(2):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); e1.synchronize()
            res[gen3] = (o, e0.elapsed_time(e1) / 10)
        a, b = res[False][0], res[True][0]
        za, zb = (a[0], b[0]) if isinstance(a, tuple) else (a, b)
        msg = f'taps {taps} {name:12s} tconv2 {res[False][1]:.3f} ms  tconv3 {res[True][1]:.3f} ms  equal {torch.equal(za, zb)}'
        if isinstance(a, tuple):
            from pose2room_amd.p2rnet import bn_op
            if a[1].shape == b[1].shape:
                sa, sb = a[1].double().sum(0), b[1].double().sum(0)
            else:
                sa, sb = bn_op.moments(a[1], N * T * V)[1], bn_op.moments(b[1], N * T * V)[1]
            msg += f'  stats rel err {((sa - sb).abs().max() / sa.abs().max()).item():.1e}'
        print(msg, flush=True)
