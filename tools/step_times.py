"""Dev: per-step wall time of the first 30 train steps of a fresh process (clock / cache warm-up behaviour)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pose2room_amd.p2rnet.synthetic import make_batch
dev = torch.device('cuda:0')
trainer, cfg = bench.build_trainer(dev, 1024, 1)
batch = make_batch(32, 1024, seed=1234, device=dev)
ts = []
for i in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    trainer.train_step(dict(batch))
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(' '.join(f'{t:.1f}' for t in ts))
