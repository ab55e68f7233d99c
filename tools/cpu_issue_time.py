"""Dev: how long the host needs to ENQUEUE one train step (launch-bound risk) vs the step's wall time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pose2room_amd.p2rnet.synthetic import make_batch
dev = torch.device('cuda:0')
trainer, cfg = bench.build_trainer(dev, 1024, 1)
batch = make_batch(32, 1024, seed=1234, device=dev)
for _ in range(3): trainer.train_step(dict(batch))
torch.cuda.synchronize()
issue, wall = [], []
for _ in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    trainer.optimizer.zero_grad()
    loss = trainer.compute_loss(dict(batch)); t1 = time.perf_counter()
    loss['total'].backward(); t2 = time.perf_counter()
    trainer.optimizer.step(); t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    issue.append((t1 - t0, t2 - t1, t3 - t2)); wall.append(t4 - t0)
f = sum(i[0] for i in issue) / len(issue) * 1e3; b = sum(i[1] for i in issue) / len(issue) * 1e3; o = sum(i[2] for i in issue) / len(issue) * 1e3
print(f'host enqueue: forward+loss {f:.1f} ms, backward {b:.1f} ms, optimizer {o:.1f} ms; step wall {sum(wall) / len(wall) * 1e3:.1f} ms')
# the bench way: K train_step calls (each ends with the loss-dict transfer), one synchronize at the end
for K in (8, 8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K): trainer.train_step(dict(batch))
    torch.cuda.synchronize(); print(f'train_step loop: {(time.perf_counter() - t0) / K * 1e3:.1f} ms/step')
