"""Dev: where do the two math modes part?  Block outputs and block-input gradients of the ST-GCN stack, split16 against
exact, per block (max |difference| / max |exact|), BatchNorm in eval mode (EVAL_BN=1) or train mode."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from tests.test_model_cpu import build
from pose2room_amd.p2rnet import math_mode
from pose2room_amd.p2rnet.synthetic import make_batch

dev = torch.device('cuda:0')
T = int(os.environ.get('T', 256))
res = {}
for m in ('exact', 'split16'):
    net, cfg = build('train', T, device=dev)
    net = net.to(dev).train()
    if os.environ.get('EVAL_BN'):
        for mod in net.modules():
            if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
                mod.eval()
    cap = {}
    hooks = []
    for i, blk in enumerate(net.backbone.st_gcn_networks):
        def fh(mod, inp, out, i=i):
            y = out[0]
            cap[f'y{i}'] = y.detach().clone()
            y.register_hook(lambda g, i=i: cap.__setitem__(f'dy{i}', g.detach().clone()))
        hooks.append(blk.register_forward_hook(fh))
    data = make_batch(2, T, seed=356, device=dev)
    with math_mode.use(m):
        ep = net(data)
        loss = net.loss(ep, data)
        loss['total'].backward()
    math_mode.reset()
    cap['grads'] = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
    res[m] = cap
for k in sorted(k for k in res['exact'] if k != 'grads'):
    a, b = res['exact'][k], res['split16'][k]
    d = (a - b).abs()
    print(f'{k:6s} max diff / max {d.max().item() / a.abs().max().item():.3e}   mean diff / mean abs {d.mean().item() / a.abs().mean().item():.3e}'
          f'   elements off by > 1e-3 of max: {int((d > 1e-3 * a.abs().max()).sum())}')
worst = sorted(((float((res['exact']['grads'][n] - res['split16']['grads'][n]).abs().max() / (res['exact']['grads'][n].abs().max() + 1e-30)), n)
                for n in res['exact']['grads']), reverse=True)[:10]
for e, n in worst:
    print(f'   {n:60s} {e:.3e}')
