import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pose2room_amd.p2rnet.synthetic import make_batch
dev = torch.device('cuda:0')
trainer, cfg = bench.build_trainer(dev, 1024, 1)
batch = make_batch(32, 1024, seed=1234, device=dev)
mode = sys.argv[1] if len(sys.argv) > 1 else 'plain'
if mode == 'nogc': gc.disable()
ts = []
for i in range(16):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    trainer.optimizer.zero_grad()
    loss = trainer.compute_loss(dict(batch)); torch.cuda.synchronize(); t1 = time.perf_counter()
    loss['total'].backward(); torch.cuda.synchronize(); t2 = time.perf_counter()
    trainer.optimizer.step(); torch.cuda.synchronize(); t3 = time.perf_counter()
    ts.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
print(mode, ' | '.join(f'{a:.0f}/{b:.0f}/{c:.1f}' for a, b, c in ts))
print('gc counts', gc.get_count(), 'alloc retries', torch.cuda.memory_stats().get('num_alloc_retries'), 'reserved GB', torch.cuda.memory_reserved() / 2**30)
