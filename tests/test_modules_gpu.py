"""GPU: the PointNet++ modules of the pointnet2_ops call surface that P2RNet itself does not instantiate --
PointnetFPModule (three_nn / three_interpolate consumers), PointnetSAModuleMSG, PointnetSAModule with GroupAll,
PointnetSAModuleMSGVotes, QueryAndGroup(sample_uniformly=True, ret_unique_cnt=True) -- on the HIP ops against the
same host modules on CPU with the oracle behind the ops (reference:
external/pointnet2_ops_lib/pointnet2_ops/pointnet2_modules.py:77-147,264-343,346-406, pointnet2_utils.py:321-330).
Tolerances: indices exact, outputs 1e-4, gradients 1e-3 of the tensor's largest entry."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _cloud(B, N, seed, spread=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(B, N, 3, generator=g) * 2 - 1) * spread + 0.2, g


def _close(a, b, tol, what):
    scale = b.abs().max().item() + 1e-12
    err = (a.detach().cpu().double() - b.detach().double()).abs().max().item()
    assert err <= tol * scale, f'{what}: err {err:.3e} vs scale {scale:.3e}'


def _run_both(make_module, inputs, dev, call):
    """Build the module once, run a deep copy on the GPU (HIP ops) and the original on CPU (oracle ops); returns
    (gpu outputs, cpu outputs, gpu grads, cpu grads) with grads = inputs that require grad + parameters."""
    from oracle.cpu_backend import cpu_ops
    torch.manual_seed(0)
    mod_cpu = make_module().train()
    mod_gpu = copy.deepcopy(mod_cpu).to(dev)

    def run(mod, device, ctx):
        ins = [t.clone().to(device).requires_grad_(t.is_floating_point() and rg) if t is not None else None
               for t, rg in inputs]
        with ctx:           # the backward of the ops runs inside the context too
            outs = call(mod, *ins)
            outs = outs if isinstance(outs, tuple) else (outs,)
            loss = sum((o * torch.linspace(0.5, 1.5, o.numel(), device=o.device).view_as(o)).sum()
                       for o in outs if o is not None and o.is_floating_point() and o.requires_grad)
            loss.backward()
        grads = [t.grad for t in ins if t is not None and t.requires_grad] + [p.grad for p in mod.parameters()]
        return outs, grads

    import contextlib
    og, gg = run(mod_gpu, dev, contextlib.nullcontext())
    oc, gc = run(mod_cpu, torch.device('cpu'), cpu_ops())
    return og, oc, gg, gc


def _compare(og, oc, gg, gc, name):
    assert len(og) == len(oc)
    for i, (a, b) in enumerate(zip(og, oc)):
        if a is None:
            assert b is None
        elif a.is_floating_point():
            _close(a, b, 1e-4, f'{name} output {i}')
        else:
            assert torch.equal(a.cpu(), b), f'{name} index output {i}'
    assert len(gg) == len(gc)
    for i, (a, b) in enumerate(zip(gg, gc)):
        assert (a is None) == (b is None)
        if a is not None:
            _close(a, b, 1e-3, f'{name} grad {i}')


@pytest.mark.parametrize("n,m,bn", [(200, 64, True), (97, 2, False), (64, 64, True)])
def test_fp_module(dev, n, m, bn):
    """three_nn -> 1/(dist + 1e-8) weights -> three_interpolate -> concat with the skip features -> shared MLP.
    m = 2 < 3 known points exercises the inf-distance / index-0 slots of three_nn inside the module."""
    from pose2room_amd.pointnet2_ops.pointnet2_modules import PointnetFPModule
    B, C1, C2 = 2, 12, 20
    unknown, g = _cloud(B, n, 11)
    known = unknown[:, :m].clone() + 0.01 * torch.randn(B, m, 3, generator=g) if m <= n else _cloud(B, m, 12)[0]
    uf = torch.randn(B, C1, n, generator=g)
    kf = torch.randn(B, C2, m, generator=g)
    res = _run_both(lambda: PointnetFPModule(mlp=[C1 + C2, 32, 16], bn=bn),
                    [(unknown, False), (known, False), (uf, True), (kf, True)], dev,
                    lambda mod, u, k, a, b: mod(u, k, a, b))
    if m < 3:    # the padded slots carry weight 1/(inf) = 0 on both sides; outputs finite
        assert torch.isfinite(res[0][0]).all()
    _compare(*res, name=f'FP n={n} m={m}')


def test_fp_module_without_known_points(dev):
    """known is None: the known features are broadcast to every unknown point (pointnet2_modules.py:391-394)."""
    from pose2room_amd.pointnet2_ops.pointnet2_modules import PointnetFPModule
    unknown, g = _cloud(2, 50, 3)
    kf = torch.randn(2, 8, 1, generator=g)
    # bn=False: with one feature vector per cloud a train-mode BatchNorm sees two distinct values per channel
    res = _run_both(lambda: PointnetFPModule(mlp=[8, 16], bn=False), [(unknown, False), (kf, True)], dev,
                    lambda mod, u, b: mod(u, None, None, b))
    _compare(*res, name='FP broadcast')


@pytest.mark.parametrize("use_xyz", [True, False])
def test_sa_module_msg(dev, use_xyz):
    from pose2room_amd.pointnet2_ops.pointnet2_modules import PointnetSAModuleMSG
    xyz, g = _cloud(2, 300, 21, spread=0.6)
    feats = torch.randn(2, 6, 300, generator=g)
    res = _run_both(lambda: PointnetSAModuleMSG(npoint=40, radii=[0.15, 0.4], nsamples=[8, 16],
                                                mlps=[[6, 16, 16], [6, 16, 24]], bn=True, use_xyz=use_xyz),
                    [(xyz, False), (feats, True)], dev, lambda mod, x, f: mod(x, f))
    _compare(*res, name=f'SA-MSG use_xyz={use_xyz}')
    assert res[0][0].shape == (2, 40, 3) and res[0][1].shape == (2, 40, 40)


def test_sa_module_group_all(dev):
    """npoint=None: one group of all points through GroupAll, new_xyz is None."""
    from pose2room_amd.pointnet2_ops.pointnet2_modules import PointnetSAModule
    xyz, g = _cloud(3, 70, 5)
    feats = torch.randn(3, 5, 70, generator=g)
    res = _run_both(lambda: PointnetSAModule(mlp=[5, 16, 32], npoint=None, bn=True, use_xyz=True),
                    [(xyz, False), (feats, True)], dev, lambda mod, x, f: mod(x, f))
    assert res[0][0] is None and res[1][0] is None and res[0][1].shape == (3, 32, 1)
    _compare(*res, name='SA GroupAll')


def test_sa_module_msg_votes(dev):
    from pose2room_amd.pointnet2_ops.pointnet2_modules import PointnetSAModuleMSGVotes
    xyz, g = _cloud(2, 256, 31, spread=0.5)
    feats = torch.randn(2, 4, 256, generator=g)
    res = _run_both(lambda: PointnetSAModuleMSGVotes(npoint=32, radii=[0.2, 0.5], nsamples=[4, 12],
                                                     mlps=[[4, 8], [4, 16]], bn=True, use_xyz=True),
                    [(xyz, False), (feats, True)], dev, lambda mod, x, f: mod(x, f))
    assert res[0][2].dtype == torch.int32 and res[0][2].shape == (2, 32)
    _compare(*res, name='SA-MSG-Votes')
    # given indices are used as they are
    inds = torch.arange(0, 256, 8, dtype=torch.int32).repeat(2, 1)
    res = _run_both(lambda: PointnetSAModuleMSGVotes(npoint=32, radii=[0.3], nsamples=[8], mlps=[[4, 8]], bn=False,
                                                     use_xyz=False),
                    [(xyz, False), (feats, True), (inds, False)], dev, lambda mod, x, f, i: mod(x, f, i))
    assert torch.equal(res[0][2].cpu(), inds)
    _compare(*res, name='SA-MSG-Votes given inds')


def test_query_and_group_sample_uniformly(dev):
    """sample_uniformly re-draws the padded slots of each ball from its distinct neighbours with the device's
    generator, so GPU and CPU draws differ; what is pinned: unique_cnt == number of distinct ball-query indices
    (oracle), the first unique_cnt slots are those indices ascending, every other slot is one of them, and the
    grouped tensors are the gather of exactly the returned neighbourhood."""
    from oracle import cpu_ext
    from pose2room_amd.pointnet2_ops import pointnet2_utils as pu
    xyz, g = _cloud(2, 400, 9, spread=0.5)
    new_xyz = xyz[:, ::10].contiguous()
    feats = torch.randn(2, 7, 400, generator=g)
    grouper = pu.QueryAndGroup(0.12, 16, use_xyz=True, ret_grouped_xyz=True, sample_uniformly=True,
                               ret_unique_cnt=True)
    torch.manual_seed(1)
    out, local, cnt = grouper(xyz.to(dev), new_xyz.to(dev), feats.to(dev))
    ref_idx = cpu_ext.OracleExt.ball_query(new_xyz, xyz, 0.12, 16).long()
    assert out.shape == (2, 10, 40, 16) and local.shape == (2, 3, 40, 16) and cnt.shape == (2, 40)
    # recover the neighbourhood from the grouped features (distinct random rows -> exact match identifies the point)
    got = out[:, 3:].cpu()                                              # (B,7,P,S)
    for b in range(2):
        for p in range(40):
            uniq = torch.unique(ref_idx[b, p])
            assert int(cnt[b, p]) == len(uniq)
            cols = []
            for s in range(16):
                match = (feats[b].t() == got[b, :, p, s]).all(1).nonzero().flatten()
                assert len(match) == 1
                cols.append(int(match))
            assert cols[:len(uniq)] == uniq.tolist()
            assert set(cols[len(uniq):]) <= set(uniq.tolist())
            want_local = xyz[b, cols].t() - new_xyz[b, p].unsqueeze(-1)
            assert torch.allclose(local[b, :, p].cpu(), want_local, atol=1e-6)
            assert torch.equal(out[b, :3, p].cpu(), local[b, :, p].cpu())
    with pytest.raises(AssertionError):
        pu.QueryAndGroup(0.1, 4, ret_unique_cnt=True)                   # needs sample_uniformly
