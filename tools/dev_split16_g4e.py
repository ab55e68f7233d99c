"""Dev: G4e (eval-BatchNorm whole step vs the reference's recorded gradients) in both math modes: worst tensors."""
import contextlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from tests.test_model_cpu import build, run_g4e, run_g4b, GB
from pose2room_amd.p2rnet import math_mode
from pose2room_amd.p2rnet.synthetic import make_batch

dev = torch.device('cuda:0')
z = np.load(GB)
for which in ('g4e', 'g4b'):
    res = {}
    for m in ('exact', 'split16'):
        net, cfg = build('train', 256, device=dev)
        net = net.to(dev)
        with math_mode.use(m):
            if which == 'g4e':
                res[m] = run_g4e(net, make_batch(2, 256, seed=356, device=dev), z, dev, contextlib.nullcontext, tol=1.0)
            else:
                res[m] = run_g4b(net, make_batch(2, 256, seed=356, device=dev), z, dev, contextlib.nullcontext, tol=1.0)
        math_mode.reset()
    names = sorted(res['split16'], key=lambda n: -res['split16'][n])[:12]
    print(which, 'worst (split16 | exact):')
    for n in names:
        print(f'   {n:60s} {res["split16"][n]:.3e}  {res["exact"][n]:.3e}')
