"""Dev (timing only, wrong numerics): split16 train step at the bench shape with the BatchNorm-backward REDUCTION pass of
the ST-GCN blocks not launched at all -- the upper bound of what folding that pass into the data-gradient kernel's store
epilogue could gain."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from pose2room_amd import _lib
from pose2room_amd.p2rnet import math_mode
from pose2room_amd.p2rnet.synthetic import make_batch
dev = torch.device('cuda:0')
trainer, cfg = bench.build_trainer(dev, 1024, 1)
batch = make_batch(32, 1024, seed=1234, device=dev)
lib = _lib.lib()
real = lib.p2r_bn_bwd_reduce


def run(steps=10):
    for _ in range(3):
        trainer.train_step(dict(batch))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        trainer.train_step(dict(batch))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def fake(N, C, L, *rest):
    if L == 1024 * 53 and C == 64:
        return 0
    return real(N, C, L, *rest)


for mode in ('exact', 'split16'):
    math_mode.set_mode(mode)
    lib.p2r_bn_bwd_reduce = real
    a = run()
    lib.p2r_bn_bwd_reduce = fake
    b = run()
    lib.p2r_bn_bwd_reduce = real
    print(f'{mode}: {a:.2f} ms/step; without the ST-GCN reduction launches {b:.2f} ms/step', flush=True)
