import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from pose2room_amd.p2rnet.synthetic import make_batch
dev = torch.device('cuda:0')
trainer, cfg = bench.build_trainer(dev, 1024, 1)
batch = make_batch(32, 1024, seed=1234, device=dev)
for _ in range(3):
    trainer.train_step(dict(batch))
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU], with_stack=True) as prof:
    trainer.train_step(dict(batch))
    torch.cuda.synchronize()
import collections
cnt = collections.Counter(); tim = collections.Counter()
for e in prof.events():
    ks = getattr(e, 'kernels', None)
    if not ks: continue
    if any(getattr(c, 'kernels', None) for c in e.cpu_children): continue
    for k in ks:
        if k.duration > 2 and ('at::native' in k.name or 'reduce_kernel' in k.name or 'elementwise' in k.name or 'copy' in k.name.lower() or 'index' in k.name):
            st = [f for f in (e.stack or []) if 'pose2room_amd' in f or 'bench' in f]
            p = e
            chain = []
            while p is not None and len(chain) < 4:
                chain.append(p.name[:40]); p = p.cpu_parent
            key = (k.name[:60], (st[0].split('pose2room_amd/')[-1] if st else ' <- '.join(chain)))
            cnt[key] += 1; tim[key] += k.duration
for key, t in tim.most_common(60):
    print(f'{t/1e3:7.3f} ms {cnt[key]:3d}  {key[0]}  @ {key[1][:110]}')
