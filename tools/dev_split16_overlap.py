"""Dev: split16 train step at the bench shape with the BatchNorm-backward passes on the side stream (default) vs inline
on the main stream, and with the sums epilogue variants: ms per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from pose2room_amd.p2rnet import bn_op, math_mode
from pose2room_amd.p2rnet.synthetic import make_batch
dev = torch.device('cuda:0')
trainer, cfg = bench.build_trainer(dev, 1024, 1)
batch = make_batch(32, 1024, seed=1234, device=dev)


def run(steps=10):
    for _ in range(3):
        trainer.train_step(dict(batch))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        trainer.train_step(dict(batch))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for mode in ('exact', 'split16'):
    math_mode.set_mode(mode)
    for ov in ((True, True), (True, False), (False, False)):
        bn_op.OVERLAP_APPLY, bn_op.OVERLAP_REDUCE = ov
        print(f'{mode}: overlap apply={ov[0]} reduce={ov[1]}: {run():.2f} ms/step', flush=True)
    bn_op.OVERLAP_APPLY, bn_op.OVERLAP_REDUCE = True, True
