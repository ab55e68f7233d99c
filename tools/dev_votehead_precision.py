"""Dev: precision of the vote head's backward (csrc/pw_layers.hip job-list kernels) on REAL activations (hash-filled
weights, backbone output at bs=8, T=1024) against the module chain in fp64; the module chain in fp32 next to it."""
import copy, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_model_cpu import build
from tests import cases
from pose2room_amd.p2rnet.modules import vote_center
from pose2room_amd.p2rnet.synthetic import make_batch

dev = torch.device('cuda:0')
B, T = 8, 1024
net, cfg = build('train', T, device=dev)
net = net.to(dev).train()
batch = make_batch(B, T, seed=2024, device=dev)
with torch.no_grad():
    ep = net.backbone(batch['input_joints'], {})
sk, sf = ep['seed_skeleton'].detach(), ep['seed_features'].detach()
print('seed_features mean/std per channel: |mean|/std max', float((sf.mean((0, 1)).abs() / sf.std((0, 1))).max()))
gx, gf = cases.seam_cotangents(B)
gx, gf = gx.to(dev), gf.to(dev)


def run(mod, dtype, fused):
    vote_center.USE_FUSED_HEAD = fused
    try:
        mod = copy.deepcopy(mod).to(dtype).train()
        x = sf.to(dtype).clone().requires_grad_(True)
        xyz, f = mod(sk.to(dtype), x)
        f = f.div(torch.norm(f, p=2, dim=2).unsqueeze(2))
        torch.autograd.backward([xyz, f], [gx.to(dtype), gf.to(dtype)])
        out = {n: p.grad.double() for n, p in mod.named_parameters()}
        out['d_seed_features'] = x.grad.double()
        # pre-BN activations' statistics
        return out
    finally:
        vote_center.USE_FUSED_HEAD = True


def run_tail(mod):
    from pose2room_amd.p2rnet import pw_op
    mod = copy.deepcopy(mod).train()
    x = sf.clone().requires_grad_(True)
    assert pw_op.votes_normalized_supported(mod, sk, x)
    xyz, f = pw_op.votes_normalized(mod, sk, x)
    torch.autograd.backward([xyz, f], [gx, gf])
    out = {n: p.grad.double() for n, p in mod.named_parameters()}
    out['d_seed_features'] = x.grad.double()
    return out


ref = run(net.centervoting, torch.float64, False)
got = run_tail(net.centervoting)
print('fused fp32 with the fused tail (votes_normalized)')
for n in ref:
    e = ((got[n] - ref[n]).abs().max() / ref[n].abs().max()).item()
    print(f'   {e:.3e}  {n}')
for name, (dt, fused) in {'fused fp32': (torch.float32, True), 'module chain fp32': (torch.float32, False)}.items():
    got = run(net.centervoting, dt, fused)
    print(name)
    for n in ref:
        e = ((got[n] - ref[n]).abs().max() / ref[n].abs().max()).item()
        print(f'   {e:.3e}  {n}')
