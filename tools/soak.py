"""Soak: N train steps at the bench shape; loss must fall, allocated / reserved memory must stay flat.
`P2R_MATH=split16 python tools/soak.py` runs the same steps (same weights, batch and noise seeds) in the split16 mode: the
two loss curves are what `profiles/r6_soak.txt` holds side by side."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pose2room_amd.p2rnet import math_mode
from pose2room_amd.p2rnet.synthetic import make_batch
dev = torch.device('cuda:0')
steps = int(os.environ.get('STEPS', 150))
trainer, cfg = bench.build_trainer(dev, 1024, 1)
batch = make_batch(32, 1024, seed=1234, device=dev)
eps = float(os.environ.get('PERTURB', 0))  # control for the comparison of the modes: how far does a relative perturbation
if eps:                                    # of the initial weights by eps (1e-7 = fp32 rounding) move the loss curve?
    g = torch.Generator(device=dev).manual_seed(99)
    with torch.no_grad():
        for p_ in trainer.net.parameters():
            p_.mul_(1 + eps * torch.randn(p_.shape, device=dev, generator=g))
torch.manual_seed(7)                      # the MDN noise of the steps
print('math mode:', math_mode.mode(), flush=True)
for i in range(steps):
    log = trainer.train_step(dict(batch))
    if i % 25 == 0 or i == steps - 1:
        torch.cuda.synchronize()
        print(i, 'loss %.3f' % float(log['total'] if 'total' in log else list(log.values())[0]),
              'allocated %.2f GB reserved %.2f GB' % (torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30), flush=True)
