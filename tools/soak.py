"""Soak: N train steps at the bench shape; loss must fall, allocated / reserved memory must stay flat."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pose2room_amd.p2rnet.synthetic import make_batch
dev = torch.device('cuda:0')
steps = int(os.environ.get('STEPS', 150))
trainer, cfg = bench.build_trainer(dev, 1024, 1)
batch = make_batch(32, 1024, seed=1234, device=dev)
for i in range(steps):
    log = trainer.train_step(dict(batch))
    if i % 25 == 0 or i == steps - 1:
        torch.cuda.synchronize()
        print(i, 'loss %.3f' % float(log['total'] if 'total' in log else list(log.values())[0]),
              'allocated %.2f GB reserved %.2f GB' % (torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30), flush=True)
