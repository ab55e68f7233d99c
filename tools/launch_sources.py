"""Where do the kernel launches of one train step come from?  Groups device kernels by the innermost
pose2room_amd source line (forward) or by the autograd node (backward) that launched them.
    python tools/launch_sources.py [--rows 60]"""
import argparse, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
import bench

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--frames', type=int, default=1024)
ap.add_argument('--rows', type=int, default=60)
args = ap.parse_args()
dev = torch.device('cuda:0')
from pose2room_amd.p2rnet.synthetic import make_batch
trainer, cfg = bench.build_trainer(dev, args.frames, 1)
batch = make_batch(args.batch, args.frames, seed=1234, device=dev)
for _ in range(3):
    trainer.train_step(dict(batch))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], with_stack=True) as prof:
    trainer.train_step(dict(batch))
    torch.cuda.synchronize()

count = collections.Counter()
time_us = collections.Counter()
for e in prof.events():
    ks = getattr(e, 'kernels', None)
    if not ks:
        continue
    # only leaf launchers: skip ops whose children also carry the kernels
    if any(getattr(c, 'kernels', None) for c in e.cpu_children):
        continue
    where = None
    for fr in (e.stack or []):
        if 'pose2room_amd' in fr:
            where = fr.split('pose2room_amd/')[-1]
            break
    if where is None:
        p = e
        while p is not None:
            if p.name.startswith('autograd::engine::evaluate_function') or p.name.startswith('Optimizer') or 'clip_grad' in p.name:
                where = p.name.replace('autograd::engine::evaluate_function: ', 'bwd ')
                break
            p = p.cpu_parent
    if where is None:
        where = 'other: ' + e.name
    count[where] += len(ks)
    time_us[where] += sum(k.duration for k in ks)
tot = sum(count.values())
print(f'{tot} launches, {sum(time_us.values()) / 1e3:.2f} ms device time')
for w, n in count.most_common(args.rows):
    print(f'{n:5d} {time_us[w] / 1e3:8.3f} ms  {w[:120]}')
