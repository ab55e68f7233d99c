"""Dev: the whole train step (zero_grad, forward, loss, backward, clip, AdamW) captured in one HIP graph
and replayed, against the eager step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from pose2room_amd.p2rnet.synthetic import make_batch
dev = torch.device('cuda:0')
B, T = int(os.environ.get('N', 32)), int(os.environ.get('T', 1024))
trainer, cfg = bench.build_trainer(dev, T, 1)
batch = make_batch(B, T, seed=1234, device=dev)
opt0 = trainer.optimizer
groups = [{k: v for k, v in g.items() if k in ('params', 'lr', 'betas', 'eps', 'weight_decay')} for g in opt0.param_groups]
trainer.optimizer = torch.optim.AdamW(groups, capturable=True, foreach=True)


def step_body():
    trainer.optimizer.zero_grad(set_to_none=True)
    loss = trainer.compute_loss(dict(batch))
    loss['total'].backward()
    max_norm = cfg.config['optimizer']['clip_norm']
    if max_norm > 0:
        torch.nn.utils.clip_grad_norm_(trainer.net.parameters(), max_norm)
    trainer.optimizer.step()
    return loss


def timeit(fn, n=10):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n, (time.perf_counter() - t0) * 1e3 / n


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step_body()
torch.cuda.current_stream().wait_stream(s)
print('eager (capturable AdamW): %.3f ms gpu, %.3f ms wall' % timeit(step_body), flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = step_body()
print('captured', flush=True)
print('graph replay: %.3f ms gpu, %.3f ms wall' % timeit(g.replay), flush=True)
print('loss', float(loss['total']))
