"""STN3d / STN_Group / weights_init (reference pointnet2_modules.py:408-538; not used by P2RNet) against vectors captured
from the imported reference (tests/golden/make_stn_golden.py -> g9_stn.npz)."""
import os

import numpy as np
import pytest
import torch

G9 = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g9_stn.npz')


def _load(mod, z, prefix='sd/'):
    sd = {k[len(prefix):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(prefix)}
    missing, unexpected = mod.load_state_dict(sd, strict=True)
    assert not missing and not unexpected


def _close(a, b, tol=2e-5):
    b = torch.as_tensor(b)
    assert a.shape == b.shape
    err = (a.detach().cpu().double() - b.double()).abs().max().item()
    assert err <= tol * max(b.abs().max().item(), 1.0), err


def test_weights_init_and_fresh_stn3d_is_the_identity():
    from pose2room_amd.pointnet2_ops.pointnet2_modules import STN3d, weights_init
    z = np.load(G9)
    net = STN3d(num_points=16)
    fc = sum(p.abs().sum().item() for n, p in net.named_parameters() if n.startswith('fc'))
    assert fc == float(z['fresh_fc_abs_sum'][0]) == 0.0              # Linear layers zeroed ...
    assert net.conv1.weight.abs().sum().item() > 0 and z['fresh_conv_nonzero'][0] == 1.0     # ... Conv1d left alone
    lin = torch.nn.Linear(4, 4); conv2 = torch.nn.Conv2d(2, 2, 1); conv1 = torch.nn.Conv1d(2, 2, 1)
    w1 = conv1.weight.clone()
    for m in (lin, conv2, conv1):
        weights_init(m)
    assert lin.weight.abs().sum() == 0 and lin.bias.abs().sum() == 0 and conv2.weight.abs().sum() == 0
    assert torch.equal(conv1.weight, w1)
    net.eval()
    x = torch.randn(2, 3, 5, 16)
    with torch.no_grad():
        _close(net(x), x, 1e-6)


def test_stn3d_matches_the_reference_on_cpu():
    from pose2room_amd.pointnet2_ops.pointnet2_modules import STN_Group
    z = np.load(G9)
    mod = STN_Group(radius=0.6, nsample=16, use_xyz=True, normalize_xyz=True)
    _load(mod, z)
    mod.train()
    out = mod.stn3d(torch.from_numpy(z['train/stn3d_input']))
    _close(out, z['train/grouped_xyz'])
    for k in z.files:
        if k.startswith('after/'):
            _close(mod.state_dict()[k[len('after/'):]], z[k], 1e-6)
    mod.eval()
    with torch.no_grad():
        _close(mod.stn3d(torch.from_numpy(z['eval/stn3d_input'])), z['eval/grouped_xyz'])


@pytest.mark.gpu
def test_stn_group_matches_the_reference(dev):
    from pose2room_amd.pointnet2_ops.pointnet2_modules import STN_Group
    z = np.load(G9)
    mod = STN_Group(radius=0.6, nsample=16, use_xyz=True, normalize_xyz=True)
    _load(mod, z)
    mod = mod.to(dev)
    args = [torch.from_numpy(z[k]).to(dev) for k in ('xyz', 'features', 'new_xyz', 'orientations')]
    mod.train()
    gx, gf = mod(*args)
    ref = torch.from_numpy(z['train/grouped_features'])
    assert torch.equal(gf.cpu()[:, 3:], ref[:, 3:])              # the grouped feature channels: index-exact
    _close(gf[:, :3], ref[:, :3], 1e-6)                          # offsets / radius: one division in fp32
    _close(gx, z['train/grouped_xyz'])
    mod.eval()
    with torch.no_grad():
        gx, gf = mod(*args)
    _close(gx, z['eval/grouped_xyz'])
    assert torch.equal(gf.cpu()[:, 3:], torch.from_numpy(z['eval/grouped_features'])[:, 3:])
