"""Dev: host-side cost of one train step by module region (record_function) and by op (self CPU time)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from pose2room_amd.p2rnet.synthetic import make_batch
dev = torch.device('cuda:0')
trainer, cfg = bench.build_trainer(dev, 1024, 1)
batch = make_batch(32, 1024, seed=1234, device=dev)
for _ in range(3): trainer.train_step(dict(batch))
torch.cuda.synchronize()
net = trainer.net.module
# region timing of the forward enqueue
def timed(fn):
    t0 = time.perf_counter(); r = fn(); return r, (time.perf_counter() - t0) * 1e3
data = trainer.to_device(dict(batch))
torch.cuda.synchronize()
ep, t_backbone = timed(lambda: net.backbone(data['input_joints'], {}))
(out), t_vote = timed(lambda: net.centervoting(ep['seed_skeleton'], ep['seed_features']) if hasattr(net, 'centervoting') else None)
torch.cuda.synchronize()
print(f'enqueue backbone fwd {t_backbone:.1f} ms')
est, t_fwd = timed(lambda: net(data))
torch.cuda.synchronize()
l, t_loss = timed(lambda: net.loss(est, data))
torch.cuda.synchronize()
print(f'enqueue whole forward {t_fwd:.1f} ms, loss {t_loss:.1f} ms')
with profile(activities=[ProfilerActivity.CPU]) as prof:
    trainer.optimizer.zero_grad()
    loss = trainer.compute_loss(dict(batch))
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.self_cpu_time_total)[:25]
for e in rows:
    print(f'{e.key[:50]:50s} n={e.count:5d} self cpu {e.self_cpu_time_total / 1e3:8.2f} ms')
