"""Dev: G4e in both math modes: which ReLU gates differ between the modes, and how large is the gradient behind them?"""
import contextlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from tests.test_model_cpu import build, run_g4e, GB
from pose2room_amd.p2rnet import math_mode
from pose2room_amd.p2rnet.synthetic import make_batch

dev = torch.device('cuda:0')
z = np.load(GB)
cap = {}
for m in ('exact', 'split16'):
    net, cfg = build('train', 256, device=dev)
    net = net.to(dev)
    c = cap[m] = {}
    for i, blk in enumerate(net.backbone.st_gcn_networks):
        def fg(mod, inp, out, i=i, c=c, blk=blk):
            zz = out[0] if torch.is_tensor(out[0]) else out[0][0]
            bn = blk.tcn[0]
            invstd = torch.rsqrt(bn.running_var + bn.eps)
            pre = (zz.detach() - bn.running_mean.view(1, -1, 1, 1)) * (invstd * bn.weight).view(1, -1, 1, 1) + bn.bias.view(1, -1, 1, 1)
            c[f'pre{i}'] = pre.clone()
        blk.gcn.register_forward_hook(fg)
        def fb(mod, inp, out, i=i, c=c):
            y = out[0]
            c[f'y{i}'] = y.detach().clone()
            y.register_hook(lambda g, i=i, c=c: c.__setitem__(f'dy{i}', g.detach().clone()))
        blk.register_forward_hook(fb)
    with math_mode.use(m):
        run_g4e(net, make_batch(2, 256, seed=356, device=dev), z, dev, contextlib.nullcontext, tol=1.0)
    math_mode.reset()
    c['gb2'] = dict(net.named_parameters())['backbone.st_gcn_networks.2.gcn.conv.bias'].grad.clone()
for i in range(6):
    a, b = cap['exact'], cap['split16']
    f1 = (a[f'pre{i}'] > 0) != (b[f'pre{i}'] > 0)
    f2 = (a[f'y{i}'] > 0) != (b[f'y{i}'] > 0)
    dy = a[f'dy{i}']
    print(f'block {i}: tcn.0 gate flips {int(f1.sum())} (|pre| there <= {a[f"pre{i}"][f1].abs().max().item() if f1.any() else 0:.2e}); '
          f'output gate flips {int(f2.sum())}; |dy| at flipped output gates / max |dy|: '
          f'{(dy[f2].abs().max() / dy.abs().max()).item() if f2.any() else 0:.3e}')
d = (cap['exact']['gb2'] - cap['split16']['gb2']).abs()
print('block 2 gcn.conv.bias grad: exact vs split16 max diff / max', (d.max() / cap['exact']['gb2'].abs().max()).item())
