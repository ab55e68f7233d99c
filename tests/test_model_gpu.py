"""GPU twin of test_model_cpu.py: the same golden vectors (captured from the imported
reference), with the pointnet2 ops and nn_distance on the HIP path."""
import contextlib
import os

import numpy as np
import pytest
import torch

from tests import cases
from tests.test_model_cpu import build, check_endpoints, run_g4, G

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag,B,T", [('g3u', 1, 768), ('g3f', 2, 512)])
def test_g3_eval_path_gpu(dev, tag, B, T):
    from pose2room_amd.p2rnet.synthetic import make_batch
    z = np.load(G)
    net, cfg = build('test', T, device=dev, remove_far_box=False)
    net = net.to(dev).eval()
    data = make_batch(B, T, seed=100 + T, device=dev)
    with torch.no_grad():
        ep = net.generate_end_points(data)
    check_endpoints(z, tag, ep)


def test_g4_train_step_gpu(dev):
    from pose2room_amd.p2rnet.synthetic import make_batch
    z = np.load(G)
    net, cfg = build('train', 256, device=dev)
    net = net.to(dev)
    run_g4(net, cfg, make_batch(2, 256, seed=356, device=dev), z, dev, contextlib.nullcontext)


def test_smoke_train_step(dev):
    from pose2room_amd.p2rnet import smoke
    out = smoke.run(dev)
    assert out['total'] > 0


def test_sa_module_matches_oracle_chain(dev, oracle):
    """PointnetSAModuleVotes on the HIP ops vs the same module on the CPU oracle ops."""
    from oracle.cpu_backend import cpu_ops
    from pose2room_amd.pointnet2_ops.pointnet2_modules import PointnetSAModuleVotes
    torch.manual_seed(3)
    mod = PointnetSAModuleVotes(npoint=128, radius=0.3, nsample=16, mlp=[256, 256, 256], use_xyz=False,
                                normalize_xyz=True, bn=False)
    xyz = cases.cloud(2, 512, 5, "walk")
    feats = torch.randn(2, 256, 512)
    with cpu_ops():
        fc = feats.clone().requires_grad_(True)
        wx, wf, wi = mod(xyz, fc)
        wf.square().sum().backward()
    mod_d = mod.to(dev)
    fd = feats.to(dev).requires_grad_(True)
    for p in mod_d.parameters():
        p.grad = None
    gx, gf, gi = mod_d(xyz.to(dev), fd)
    gf.square().sum().backward()
    assert torch.equal(gi.cpu(), wi)
    assert torch.equal(gx.cpu(), wx)
    torch.testing.assert_close(gf.detach().cpu(), wf.detach(), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(fd.grad.cpu(), fc.grad, rtol=1e-3, atol=1e-3)
