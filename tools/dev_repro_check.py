"""Dev: are repeated forward/backward passes of the full model bit-reproducible (one process, no DDP), with the
BatchNorm-backward passes on the side stream and inline?"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pose2room_amd.p2rnet import bn_op
from pose2room_amd.p2rnet.synthetic import make_batch

dev = torch.device('cuda:0')
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 2, int(sys.argv[2]) if len(sys.argv) > 2 else 64
trainer, cfg = bench.build_trainer(dev, T, 1)
batch = make_batch(B, T, seed=99, device=dev)
trainer.train_step(dict(batch)); trainer.train_step(dict(batch))


def grads(inline):
    bn_op.SIDE_INLINE = inline
    trainer.net.zero_grad()
    torch.manual_seed(1000)
    est = trainer.net(dict(batch))
    loss = trainer.net.module.loss(est, batch)
    loss['total'].backward()
    torch.cuda.synchronize()
    bn_op.SIDE_INLINE = False
    out = {n: p.grad.detach().clone() for n, p in trainer.net.module.named_parameters()}
    out['_loss'] = loss['total'].detach().clone()
    for k in ('seed_features', 'vote_xyz', 'vote_features', 'aggregated_vote_inds', 'center'):
        out['_' + k] = est[k].detach().clone()
    return out


base = grads(True)
for rnd in range(3):
    for inline in (True, False):
        got = grads(inline)
        differ = [n for n in base if not torch.equal(base[n], got[n])]
        worst = max([((base[n].double() - got[n].double()).abs().max() / (base[n].double().abs().max() + 1e-30)).item() for n in differ] or [0.0])
        rel = lambda n: ((base[n].double() - got[n].double()).abs().max() / (base[n].double().abs().max() + 1e-30)).item()
        print(rnd, 'main' if inline else 'side', len(differ), f'{worst:.3e}', {n: f'{rel(n):.2e}' for n in differ if n.startswith('_')},
              {n: f'{rel(n):.2e}' for n in differ[:3]})
