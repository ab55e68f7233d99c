"""Steady-state kernel breakdown of the train step with torch.profiler (GPU box only).
    python tools/profile_step.py [--batch 32 --frames 1024 --steps 2]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
import bench

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--frames', type=int, default=1024)
ap.add_argument('--steps', type=int, default=2)
ap.add_argument('--rows', type=int, default=40)
ap.add_argument('--ops', action='store_true', help='group by torch op instead of kernel')
args = ap.parse_args()
dev = torch.device('cuda:0')
from pose2room_amd.p2rnet.synthetic import make_batch
trainer, cfg = bench.build_trainer(dev, args.frames, 1)
batch = make_batch(args.batch, args.frames, seed=1234, device=dev)
for _ in range(3):
    trainer.train_step(dict(batch))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(args.steps):
        trainer.train_step(dict(batch))
    torch.cuda.synchronize()
ev = prof.key_averages()
if args.ops:
    rows = sorted(ev, key=lambda e: -e.self_device_time_total)
    tot = sum(e.self_device_time_total for e in rows)
    print(f'total self device time per step: {tot / args.steps / 1e3:.2f} ms')
    for e in rows[:args.rows]:
        print(f'{e.key[:70]:70s} n={e.count // args.steps:5d} {e.self_device_time_total / args.steps / 1e3:9.3f} ms')
    sys.exit(0)
rows = sorted([e for e in ev if e.device_time_total > 0 and e.device_type.name != 'CPU'] or
              [e for e in ev if e.device_time_total > 0], key=lambda e: -e.device_time_total)
tot = sum(e.device_time_total for e in rows)
print(f'total device time per step: {tot / args.steps / 1e3:.2f} ms over {args.steps} steps')
for e in rows[:args.rows]:
    print(f'{e.key[:90]:90s} n={e.count // args.steps:5d} {e.device_time_total / args.steps / 1e3:9.3f} ms  {100 * e.device_time_total / tot:5.1f}%')
