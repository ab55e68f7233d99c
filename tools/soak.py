"""Dev: 150 train steps on changing batches: loss stays finite, memory does not grow, step time is flat."""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pose2room_amd.p2rnet.synthetic import make_batch
dev = torch.device('cuda:0')
trainer, cfg = bench.build_trainer(dev, 1024, 1)
batches = [make_batch(32, 1024, seed=100 + i, device=dev) for i in range(4)]
for i in range(3): trainer.train_step(dict(batches[i % 4]))
gc.collect(); gc.freeze()
torch.cuda.synchronize(); mem0 = torch.cuda.memory_reserved()
ts, losses = [], []
for i in range(150):
    t0 = time.perf_counter()
    l = trainer.train_step(dict(batches[i % 4]))
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3); losses.append(l['total'])
import math
assert all(math.isfinite(x) for x in losses)
print('loss first/last 4-batch means: %.2f -> %.2f' % (sum(losses[:4]) / 4, sum(losses[-4:]) / 4))
print('step ms: first 10 mean %.2f, last 10 mean %.2f, max %.2f' % (sum(ts[:10]) / 10, sum(ts[-10:]) / 10, max(ts)))
print('reserved GB before/after: %.2f / %.2f' % (mem0 / 2**30, torch.cuda.memory_reserved() / 2**30))
