"""Workload for whole-step PMC passes: STEPS identical train steps at the bench shape, nothing else."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from pose2room_amd.p2rnet.synthetic import make_batch
dev = torch.device('cuda:0')
steps = int(os.environ.get('STEPS', 3))
B, T = int(os.environ.get('N', 32)), int(os.environ.get('T', 1024))
from pose2room_amd.p2rnet import math_mode          # P2R_MATH=split16 selects the opt-in arithmetic (environment)
trainer, cfg = bench.build_trainer(dev, T, 1)
batch = make_batch(B, T, seed=1234, device=dev)
for _ in range(steps):
    trainer.train_step(dict(batch))
torch.cuda.synchronize()
