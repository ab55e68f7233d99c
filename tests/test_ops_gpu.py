"""GPU parity: HIP kernels (through the C ABI / `_ext` shim) vs the CPU oracle.

Index outputs and forward values are compared bit for bit; scatter-add
gradients (order-free atomics, like the reference) with a tolerance."""
import numpy as np
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ext(dev):
    from pose2room_amd.pointnet2_ops import _ext
    return _ext


@pytest.mark.parametrize("b,n,m,kind,seed", cases.FPS_CASES)
def test_fps_index_exact(ext, oracle, dev, b, n, m, kind, seed):
    xyz = cases.cloud(b, n, seed, kind)
    want = oracle.OracleExt.furthest_point_sampling(xyz, m)
    got = ext.furthest_point_sampling(xyz.to(dev), m).cpu()
    assert got.dtype == torch.int32 and got.shape == (b, m)
    assert torch.equal(got, want), f"first mismatch at {(got != want).nonzero()[:3].tolist()}"


@pytest.mark.parametrize("b,n,m,radius,nsample,kind,seed", cases.BALL_CASES)
def test_ball_query_index_exact(ext, oracle, dev, b, n, m, radius, nsample, kind, seed):
    xyz = cases.cloud(b, n, seed, kind)
    new_xyz = cases.centres_from(xyz, m, seed)
    want = oracle.OracleExt.ball_query(new_xyz, xyz, radius, nsample)
    got = ext.ball_query(new_xyz.to(dev), xyz.to(dev), radius, nsample).cpu()
    assert torch.equal(got, want)


def test_ball_query_empty_balls(ext, oracle, dev):
    xyz = cases.cloud(2, 200, 3)
    new_xyz = xyz[:, :20].clone() + 100.0  # nothing in range -> rows of zeros
    got = ext.ball_query(new_xyz.to(dev), xyz.to(dev), 0.3, 16).cpu()
    assert torch.equal(got, torch.zeros(2, 20, 16, dtype=torch.int32))
    assert torch.equal(got, oracle.OracleExt.ball_query(new_xyz, xyz, 0.3, 16))


@pytest.mark.parametrize("b,c,n,p,s,seed", [(2, 256, 512, 128, 16, 1), (2, 3, 512, 128, 16, 2),
                                            (1, 5, 33, 7, 3, 3), (3, 64, 1000, 50, 32, 4),
                                            (1, 17, 20000, 9, 5, 5), (2, 1, 1, 1, 1, 6)])
def test_group_points_and_grad(ext, oracle, dev, b, c, n, p, s, seed):
    g = torch.Generator().manual_seed(seed)
    pts = torch.randn(b, c, n, generator=g)
    idx = torch.randint(0, n, (b, p, s), generator=g, dtype=torch.int32)
    want = oracle.OracleExt.group_points(pts, idx)
    got = ext.group_points(pts.to(dev), idx.to(dev)).cpu()
    assert torch.equal(got, want)
    go = torch.randn(b, c, p, s, generator=g)
    wg = oracle.OracleExt.group_points_grad(go, idx, n)
    gg = ext.group_points_grad(go.to(dev), idx.to(dev), n).cpu()
    # gather form (round 5): every destination sums its slots in ascending order, like the sequential CPU loop -- bit-exact
    # and the same from run to run (no atomics)
    assert torch.equal(gg, wg)
    assert torch.equal(ext.group_points_grad(go.to(dev), idx.to(dev), n).cpu(), gg)


def test_group_points_grad_long_index_list_falls_back(ext, oracle, dev):
    """an index list too long for LDS (S = 40000 slots) takes the order-free scatter forms: 1e-5 like the reference's atomics"""
    g = torch.Generator().manual_seed(8)
    b, c, n, p, s = 1, 6, 700, 2500, 16
    idx = torch.randint(0, n, (b, p, s), generator=g, dtype=torch.int32)
    go = torch.randn(b, c, p, s, generator=g)
    torch.testing.assert_close(ext.group_points_grad(go.to(dev), idx.to(dev), n).cpu(),
                               oracle.OracleExt.group_points_grad(go, idx, n), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("b,c,n,m,seed", [(2, 3, 512, 128, 1), (1, 7, 100, 100, 2), (2, 256, 64, 9, 3)])
def test_gather_points_and_grad(ext, oracle, dev, b, c, n, m, seed):
    g = torch.Generator().manual_seed(seed)
    pts = torch.randn(b, c, n, generator=g)
    idx = torch.randint(0, n, (b, m), generator=g, dtype=torch.int32)
    assert torch.equal(ext.gather_points(pts.to(dev), idx.to(dev)).cpu(),
                       oracle.OracleExt.gather_points(pts, idx))
    go = torch.randn(b, c, m, generator=g)
    assert torch.equal(ext.gather_points_grad(go.to(dev), idx.to(dev), n).cpu(),
                       oracle.OracleExt.gather_points_grad(go, idx, n))          # ascending-slot sums, no atomics


@pytest.mark.parametrize("b,n,m,kind,seed", [(2, 512, 128, "uniform", 1), (2, 100, 2, "uniform", 2),
                                             (1, 10, 1, "uniform", 3), (2, 300, 64, "lattice", 4),
                                             (1, 3000, 2500, "halflattice", 5), (1, 7, 0, "uniform", 6)])
def test_three_nn_exact(ext, oracle, dev, b, n, m, kind, seed):
    unknown = cases.cloud(b, n, seed, kind)
    known = cases.cloud(b, m, seed + 50, kind) if m > 0 else torch.zeros(b, 0, 3)
    wd, wi = oracle.OracleExt.three_nn(unknown, known)
    gd, gi = ext.three_nn(unknown.to(dev), known.to(dev))
    assert torch.equal(gi.cpu(), wi)
    assert torch.equal(gd.cpu(), wd)  # includes +inf slots when m < 3


@pytest.mark.parametrize("b,c,m,n,seed", [(2, 64, 128, 512, 1), (1, 3, 5, 9, 2), (2, 256, 2048, 3000, 3)])
def test_three_interpolate_and_grad(ext, oracle, dev, b, c, m, n, seed):
    g = torch.Generator().manual_seed(seed)
    pts = torch.randn(b, c, m, generator=g)
    idx = torch.randint(0, m, (b, n, 3), generator=g, dtype=torch.int32)
    w = torch.rand(b, n, 3, generator=g)
    w = (w / w.sum(-1, keepdim=True)).contiguous()
    assert torch.equal(ext.three_interpolate(pts.to(dev), idx.to(dev), w.to(dev)).cpu(),
                       oracle.OracleExt.three_interpolate(pts, idx, w))
    go = torch.randn(b, c, n, generator=g)
    torch.testing.assert_close(ext.three_interpolate_grad(go.to(dev), idx.to(dev), w.to(dev), m).cpu(),
                               oracle.OracleExt.three_interpolate_grad(go, idx, w, m),
                               rtol=1e-5, atol=1e-5)


def test_ext_argument_checks(ext, dev):
    x = torch.rand(1, 8, 3, device=dev)
    with pytest.raises(RuntimeError):
        ext.furthest_point_sampling(x.cpu(), 2)            # CPU not supported
    with pytest.raises(RuntimeError):
        ext.furthest_point_sampling(x.double(), 2)         # must be float32
    with pytest.raises(RuntimeError):
        ext.gather_points(x.transpose(1, 2), torch.zeros(1, 2, dtype=torch.int32, device=dev))  # contiguity
    with pytest.raises(RuntimeError):
        ext.gather_points(x.transpose(1, 2).contiguous(), torch.zeros(1, 2, dtype=torch.int64, device=dev))


NND_SHAPES = [(1, 5, 6, 3), (64, 3, 53, 3), (4, 128, 10, 3), (1, 128, 1, 3), (3, 17, 200, 3), (2, 9, 4, 5)]


@pytest.mark.parametrize("B,N,M,C", NND_SHAPES)
@pytest.mark.parametrize("kw", [{}, {"l1smooth": True}, {"l1": True}, {"l1smooth": True, "delta": 0.3}])
def test_nn_distance_bit_exact(oracle, dev, B, N, M, C, kw):
    from pose2room_amd.net_utils.nn_distance import nn_distance
    g = torch.Generator().manual_seed(B * 1000 + N * 10 + M)
    a = torch.randn(B, N, C, generator=g)
    q = torch.randn(B, M, C, generator=g)
    want = oracle.nn_distance(a, q, **kw)
    ad = a.to(dev).requires_grad_(True)
    qd = q.to(dev).requires_grad_(True)
    got = nn_distance(ad, qd, **kw)
    for w, t in zip(want, got):
        assert t.dtype == w.dtype
        assert torch.equal(t.detach().cpu(), w)
    g1 = torch.randn(B, N, generator=g)
    g2 = torch.randn(B, M, generator=g)
    (got[0] * g1.to(dev)).sum().add((got[2] * g2.to(dev)).sum()).backward()
    wa, wq = oracle.nn_distance_grad(a, q, want[1], want[3], g1, g2, **kw)
    assert torch.equal(ad.grad.cpu(), wa)
    assert torch.equal(qd.grad.cpu(), wq)


@pytest.mark.parametrize("K", [1, 2, 16, 128, 300, 1024])
@pytest.mark.parametrize("thr", [0.1, 0.25])
@pytest.mark.parametrize("old_type", [False, True])
def test_nms3d_matches_oracle(oracle, dev, K, thr, old_type):
    from pose2room_amd.net_utils import nms
    boxes = cases.random_boxes(K, seed=K)
    assert nms.nms_3d_faster(boxes[:, :7], thr, old_type) == oracle.nms_3d(boxes[:, :7], thr, old_type)
    assert nms.nms_3d_faster_samecls(boxes, thr, old_type) == oracle.nms_3d(boxes, thr, old_type, True)


def test_nms3d_batched_with_valid_mask(oracle, dev):
    from pose2room_amd.net_utils import nms
    B, K = 5, 128
    allb = np.stack([cases.random_boxes(K, seed=100 + i, stride=7) for i in range(B)])
    rng = np.random.default_rng(0)
    valid = rng.uniform(size=(B, K)) < 0.7
    valid[3] = False                       # an empty set
    keep, pick, npick = nms.nms_3d_batched(torch.from_numpy(allb).to(dev), 0.1,
                                           valid=torch.from_numpy(valid), return_pick=True)
    keep = keep.cpu().numpy(); pick = pick.cpu().numpy(); npick = npick.cpu().numpy()
    for i in range(B):
        sel = np.where(valid[i])[0]
        want = [int(sel[j]) for j in oracle.nms_3d(allb[i][sel], 0.1)] if len(sel) else []
        assert list(pick[i, :npick[i]]) == want
        mask = np.zeros(K, np.uint8); mask[want] = 1
        assert np.array_equal(keep[i], mask)


@pytest.mark.parametrize("B,N,M,kind", [(2, 512, 128, "walk"), (1, 300, 37, "uniform"), (3, 64, 6, "lattice")])
def test_sa_votes_fused_forward(ext, oracle, dev, B, N, M, kind):
    """Fused ball-query + group + MLP + max kernel vs the unfused HIP chain / oracle indices."""
    from pose2room_amd.pointnet2_ops import fused
    from pose2room_amd.pointnet2_ops.pointnet2_modules import PointnetSAModuleVotes
    torch.manual_seed(B * 7 + M)
    mod = PointnetSAModuleVotes(npoint=M, radius=0.3, nsample=16, mlp=[256, 256, 256], use_xyz=False,
                                normalize_xyz=True, bn=False).to(dev)
    xyz = cases.cloud(B, N, 11, kind)
    feats = torch.randn(B, 256, N)
    new_xyz = cases.centres_from(xyz, M, 11)
    out, idx = fused.sa_votes(xyz.to(dev), new_xyz.to(dev), feats.to(dev), 0.3, 16, mod.mlp_module, return_idx=True)
    assert torch.equal(idx.cpu(), oracle.OracleExt.ball_query(new_xyz, xyz, 0.3, 16))
    with torch.no_grad():
        grouped = ext.group_points(feats.to(dev), idx)
        want = mod.mlp_module(grouped).max(dim=3).values
    torch.testing.assert_close(out, want, rtol=1e-4, atol=1e-4)
    # the module takes the fused path when autograd is off and returns the same tensors
    with torch.no_grad():
        nx, nf, inds = mod(xyz.to(dev), feats.to(dev))
    mod.fused = False
    with torch.no_grad():
        nx2, nf2, inds2 = mod(xyz.to(dev), feats.to(dev))
    assert torch.equal(inds, inds2) and torch.equal(nx, nx2)
    torch.testing.assert_close(nf, nf2, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("B,N,M,kind", [(2, 512, 128, "walk"), (1, 300, 37, "uniform"), (3, 64, 6, "lattice")])
def test_sa_votes_fused_backward(ext, oracle, dev, B, N, M, kind):
    """Training path of PointnetSAModuleVotes on the fused kernels (forward that saves G / H / arg-max, the two
    transposed MFMA layers, split-K weight gradients, LDS-accumulated feature gradient) against (a) the unfused HIP
    op chain through torch autograd and (b) the same module on the CPU oracle ops: indices exact, outputs 1e-4,
    gradients 1e-3 of each tensor's largest entry.  'lattice' has duplicate neighbours in a ball (padding with the
    first hit), i.e. exact ties in the max-pool: the gradient must go to the first maximum like max_pool2d's."""
    import copy
    from oracle.cpu_backend import cpu_ops
    from pose2room_amd.pointnet2_ops.pointnet2_modules import PointnetSAModuleVotes
    torch.manual_seed(B * 5 + M)
    mod_c = PointnetSAModuleVotes(npoint=M, radius=0.3, nsample=16, mlp=[256, 256, 256], use_xyz=False,
                                  normalize_xyz=True, bn=False)
    xyz = cases.cloud(B, N, 21, kind)
    feats = torch.randn(B, 256, N)
    go = torch.randn(B, 256, M)

    def run(mod, device, ctx, fused):
        mod.fused = fused
        for p in mod.parameters():
            p.grad = None
        f = feats.detach().clone().to(device).requires_grad_(True)
        with ctx():
            nx, nf, inds = mod(xyz.to(device), f)
            nf.backward(go.to(device))
        return (nx.cpu(), nf.detach().cpu(), inds.cpu(), f.grad.cpu(),
                {n: p.grad.detach().cpu().clone() for n, p in mod.named_parameters()})

    want = run(mod_c, 'cpu', cpu_ops, False)                       # oracle chain, torch autograd
    mod_d = copy.deepcopy(mod_c).to(dev)
    import contextlib
    chain = run(mod_d, dev, contextlib.nullcontext, False)         # unfused HIP chain
    got = run(mod_d, dev, contextlib.nullcontext, True)            # fused forward + backward
    for ref in (want, chain):
        assert torch.equal(got[0], ref[0]) and torch.equal(got[2], ref[2])
        torch.testing.assert_close(got[1], ref[1], rtol=1e-4, atol=1e-4)
        scale = ref[3].abs().max().item()
        assert (got[3] - ref[3]).abs().max().item() <= 1e-3 * scale
        for n in ref[4]:
            scale = max(ref[4][n].abs().max().item(), 1e-6)
            err = (got[4][n] - ref[4][n]).abs().max().item()
            assert err <= 1e-3 * scale, f'{n}: {err:.3e} vs scale {scale:.3e}'
