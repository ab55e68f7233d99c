"""Point-wise convolution stacks (csrc/pw_layers.hip, p2rnet/pw_op.py) against plain torch on the same GPU tensors:
every kernel in isolation against an fp64 reference, then the vote head and the proposal head against the module chains
they replace (nn.Conv1d / nn.BatchNorm1d / torch mixture read-out) -- outputs, running statistics and every gradient.
The model-level goldens (G3 / G4 / G4e, tests/test_model_gpu.py) pin the same path to the reference's numbers."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


def _rand(shape, seed, dev, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


# ---- single kernels ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,L,K,R,x_nlc,out_nlc,tr,bias", [
    (2, 128, 256, 128, 0, 0, 0, 0), (2, 128, 128, 128, 0, 0, 1, 0), (3, 64, 128, 24, 0, 1, 1, 1),
    (2, 128, 128, 100, 0, 0, 1, 1), (2, 512, 256, 256, 1, 0, 0, 0), (1, 192, 256, 259, 0, 1, 1, 1),
    (1, 64, 16, 7, 0, 0, 0, 1), (2, 64, 48, 33, 1, 1, 1, 0)])
def test_pw_gemm_forward(dev, B, L, K, R, x_nlc, out_nlc, tr, bias):
    from pose2room_amd.p2rnet import pw_op
    from pose2room_amd import _lib
    x = _rand((B, L, K) if x_nlc else (B, K, L), 1, dev)
    W = _rand((R, K), 2, dev, 0.1)
    bvec = _rand((R,), 3, dev) if bias else None
    fin = torch.stack([_rand((K,), 4, dev), _rand((K,), 5, dev).abs() + 0.5, _rand((K,), 6, dev),
                       _rand((K,), 7, dev)]).contiguous()
    out = torch.empty((B, L, R) if out_nlc else (B, R, L), device=dev)
    stats = torch.empty((B * L // 64, R, 3), device=dev)
    job = dict(x=pw_op._at(x), w=pw_op._at(W), bias=pw_op._at(bvec), out=pw_op._at(out), stats=pw_op._at(stats), k=K, rows=R,
               x_ctot=K, x_nlc=x_nlc, out_ctot=R, out_nlc=out_nlc)
    if tr:
        job.update(tr=pw_op._at(fin, 2 * K), tr_mode=1, tr_ld=K)
    pw_op._gemm([job], B, L, _lib.current_stream(dev))
    xc = (x.transpose(1, 2) if x_nlc else x).double()                       # (B,K,L)
    if tr:
        xc = torch.relu(xc * fin[2].double()[None, :, None] + fin[3].double()[None, :, None])
    ref = torch.einsum('rk,bkl->brl', W.double(), xc)
    if bias:
        ref = ref + bvec.double()[None, :, None]
    got = out.transpose(1, 2) if out_nlc else out
    assert _rel(got, ref) < 2e-6
    # statistics entries: (64, mean, M2) of every row over each 64-column tile
    cols = ref.permute(1, 0, 2).reshape(R, B * L // 64, 64)
    assert torch.all(stats[..., 0] == 64)
    assert _rel(stats[..., 1].t(), cols.mean(-1)) < 1e-5
    m2 = ((cols - cols.mean(-1, keepdim=True)) ** 2).sum(-1)
    assert _rel(stats[..., 2].t(), m2) < 1e-5


@pytest.mark.parametrize("B,L,K,R,x_nlc,lazy,out_nlc", [
    (2, 128, 100, 128, 0, 0, 0), (2, 128, 128, 128, 0, 1, 0), (2, 64, 24, 128, 1, 0, 0), (1, 128, 259, 256, 1, 0, 0),
    (2, 128, 128, 256, 0, 1, 1), (1, 64, 33, 48, 1, 1, 0)])
def test_pw_gemm_data_gradient(dev, B, L, K, R, x_nlc, lazy, out_nlc):
    """transposed weights, BatchNorm-backward input form, ReLU mask + BatchNorm-backward sums epilogue"""
    from pose2room_amd.p2rnet import pw_op
    from pose2room_amd import _lib
    g = _rand((B, L, K) if x_nlc else (B, K, L), 1, dev)
    z = _rand((B, L, K) if x_nlc else (B, K, L), 2, dev)
    W = _rand((K, R), 3, dev, 0.1)             # the layer's weight [out = K][in = R]
    coef = torch.stack([_rand((K,), 4, dev), _rand((K,), 5, dev), _rand((K,), 6, dev)]).contiguous()
    mz = _rand((B, R, L), 7, dev)
    mfin = torch.stack([_rand((R,), 8, dev), _rand((R,), 9, dev).abs() + 0.5, _rand((R,), 10, dev),
                        _rand((R,), 11, dev)]).contiguous()
    out = torch.empty((B, L, R) if out_nlc else (B, R, L), device=dev)
    part = torch.empty((B * L // 64, R, 2), device=dev)
    job = dict(x=pw_op._at(g), x_nlc=x_nlc, x_ctot=K, w=pw_op._at(W), w_t=1, out=pw_op._at(out), out_ctot=R, out_nlc=out_nlc,
               stats=pw_op._at(part), k=K, rows=R, epilogue=1, mz=pw_op._at(mz), mz_ctot=R, mfin=pw_op._at(mfin), mfin_ld=R)
    if lazy:
        job.update(x2=pw_op._at(z), tr=pw_op._at(coef), tr_mode=2, tr_ld=K)
    pw_op._gemm([job], B, L, _lib.current_stream(dev))
    gd = (g.transpose(1, 2) if x_nlc else g).double()
    zd = (z.transpose(1, 2) if x_nlc else z).double()
    dz = gd
    if lazy:
        c = coef.double()
        dz = c[0][None, :, None] * gd + c[1][None, :, None] * zd + c[2][None, :, None]
    dA = torch.einsum('kr,bkl->brl', W.double(), dz)
    f = mfin.double()
    mask = (mz * mfin[2][None, :, None] + mfin[3][None, :, None]) > 0       # the kernel's own fp32 expression
    ref = dA * mask
    got = out.transpose(1, 2) if out_nlc else out
    assert _rel(got, ref) < 2e-6
    xhat = (mz.double() - f[0][None, :, None]) * f[1][None, :, None]
    tiles = lambda t: t.permute(1, 0, 2).reshape(R, B * L // 64, 64).sum(-1)
    assert _rel(part[..., 0].t(), tiles(ref)) < 1e-5
    assert _rel(part[..., 1].t(), tiles(ref * xhat)) < 1e-5


@pytest.mark.parametrize("B,L,R,K,x_nlc,y_nlc,lazy,ytr", [
    (2, 128, 128, 128, 0, 0, 1, 1), (2, 128, 100, 128, 0, 0, 0, 1), (2, 64, 24, 128, 1, 0, 0, 1),
    (1, 256, 259, 256, 1, 0, 0, 1), (2, 512, 256, 256, 0, 1, 1, 0), (2, 128, 128, 256, 0, 0, 1, 0), (1, 64, 33, 20, 1, 1, 0, 1)])
def test_pw_wgrad(dev, B, L, R, K, x_nlc, y_nlc, lazy, ytr):
    from pose2room_amd.p2rnet import pw_op
    from pose2room_amd import _lib
    g = _rand((B, L, R) if x_nlc else (B, R, L), 1, dev)
    z = _rand((B, L, R) if x_nlc else (B, R, L), 2, dev)
    y = _rand((B, L, K) if y_nlc else (B, K, L), 3, dev)
    coef = torch.stack([_rand((R,), 4, dev), _rand((R,), 5, dev), _rand((R,), 6, dev)]).contiguous()
    yfin = torch.stack([_rand((K,), 7, dev), _rand((K,), 8, dev)]).contiguous()
    chunks = B * L // 64
    for split in sorted({1, min(3, chunks), chunks}):
        pw = torch.full((split, R, K), float('nan'), device=dev)
        pb = torch.full((split, R), float('nan'), device=dev)
        job = dict(x=pw_op._at(g), x_nlc=x_nlc, x_ctot=R, rows=R, y=pw_op._at(y), y_nlc=y_nlc, y_ctot=K, k=K,
                   dw_part=pw_op._at(pw), db_part=pw_op._at(pb), split=split)
        if lazy:
            job.update(x2=pw_op._at(z), tr=pw_op._at(coef), tr_mode=2, tr_ld=R)
        if ytr:
            job.update(ytr=pw_op._at(yfin), ytr_ld=K)
        pw_op._wgrad([job], B, L, _lib.current_stream(dev))
        dW, db = torch.empty((R, K), device=dev), torch.empty((R,), device=dev)
        pw_op._reduce([(pw, dW), (pb, db)], _lib.current_stream(dev))
        gd = (g.transpose(1, 2) if x_nlc else g).double()
        zd = (z.transpose(1, 2) if x_nlc else z).double()
        yd = (y.transpose(1, 2) if y_nlc else y).double()
        dz = gd
        if lazy:
            c = coef.double()
            dz = c[0][None, :, None] * gd + c[1][None, :, None] * zd + c[2][None, :, None]
        if ytr:
            yd = torch.relu(yd * yfin[0].double()[None, :, None] + yfin[1].double()[None, :, None])
        assert _rel(dW, torch.einsum('brl,bkl->rk', dz, yd)) < 2e-6, split
        assert _rel(db, dz.sum((0, 2))) < 2e-6, split


def test_pw_bn_finalize_matches_batchnorm1d(dev):
    from pose2room_amd.p2rnet import pw_op
    from pose2room_amd import _lib
    torch.manual_seed(0)
    C, B, L = 96, 3, 128
    z = torch.randn(B, C, L, device=dev) * 3 + 500.0          # |mean| >> std
    bn = torch.nn.BatchNorm1d(C).to(dev)
    bn.weight.data.uniform_(0.5, 1.5), bn.bias.data.uniform_(-1, 1)
    ref = torch.nn.BatchNorm1d(C).to(dev).double()
    ref.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in bn.state_dict().items()})
    cols = z.permute(1, 0, 2).reshape(C, -1, 64).double()
    part = torch.stack([torch.full_like(cols[..., 0], 64.0), cols.mean(-1),
                        ((cols - cols.mean(-1, keepdim=True)) ** 2).sum(-1)], -1).permute(1, 0, 2).float().contiguous()
    fin = torch.empty((4, C), device=dev)
    pw_op._bn_finalize([bn], [part], fin, [0], _lib.current_stream(dev))
    out_ref = ref(z.double())
    got = z * fin[2][None, :, None] + fin[3][None, :, None]
    assert _rel(got, out_ref) < 1e-4
    assert _rel(bn.running_mean, ref.running_mean) < 1e-6 and _rel(bn.running_var, ref.running_var) < 1e-5
    assert int(bn.num_batches_tracked) == 1
    # evaluation mode: running statistics, nothing updated
    before = bn.running_mean.clone()
    pw_op._bn_finalize([bn], [None], fin, [0], _lib.current_stream(dev))
    ref.eval()
    assert _rel(z * fin[2][None, :, None] + fin[3][None, :, None], ref(z.double())) < 1e-5
    assert torch.equal(before, bn.running_mean) and int(bn.num_batches_tracked) == 1


@pytest.mark.parametrize("f64", [False, True])
@pytest.mark.parametrize("sample", [True, False])
def test_mdn_mix(dev, f64, sample):
    """mixture read-out and its gradient against the module's own torch expression (mdn.py:34-99)"""
    from pose2room_amd.p2rnet import pw_op
    from pose2room_amd.p2rnet.config import Struct
    from pose2room_amd.p2rnet.modules.mdn import MixtureDensityHead
    from pose2room_amd import _lib
    torch.manual_seed(1)
    B, G, L, D = 3, 100, 128, 2 if f64 else 3
    mu0 = torch.randn(G, D, dtype=torch.float64 if f64 else torch.float32)
    head = MixtureDensityHead(Struct(input_dim=128, num_gaussian=G, out_dim=D, mu_bias_init=mu0, n_samples=1,
                                     central_tendency='mean')).to(dev)
    head.log_sigma.data.uniform_(-1.5, 0.0)
    logits_all = torch.randn(B, 2 * G, L, device=dev)
    logit = logits_all[:, G:].clone().requires_grad_(True)
    eps = torch.randn(B * L, G, 1, D, device=dev, dtype=mu0.dtype)
    pi = torch.sigmoid(logit)
    ref = head.generate_point_predictions(pi, eps=eps) if sample else head.get_mean(pi)       # (B, D, L)
    pred, = pw_op._mix_forward(logits_all, [G], G, L, [head], [eps if sample else None])      # (B, L, D)
    assert pred.dtype == mu0.dtype
    assert _rel(pred.transpose(1, 2), ref) < (1e-12 if f64 else 2e-6)
    if not sample:
        return
    dpred = torch.randn(B, L, D, device=dev, dtype=mu0.dtype)
    gl, gmu, gls = torch.autograd.grad(ref, [logit, head.mu, head.log_sigma], dpred.transpose(1, 2))
    dlogit = torch.zeros(B, 2 * G, L, device=dev)
    (dmu,), (dls,) = pw_op._mix_backward(logits_all, dlogit, [G], G, L, [head], [eps], [dpred])
    assert torch.all(dlogit[:, :G] == 0)
    assert _rel(dlogit[:, G:], gl) < 2e-6
    assert _rel(dmu, gmu) < (1e-12 if f64 else 1e-5)
    assert _rel(dls, gls) < 1e-5


# ---- the heads against their module chains ---------------------------------------------------------------------------
def _clone_module(m):
    import copy
    return copy.deepcopy(m)


def _grads(params):
    return [p.grad.clone() if p.grad is not None else None for p in params]


@pytest.mark.parametrize("train", [True, False])
def test_vote_head_matches_module_chain(dev, train):
    from tests.test_model_cpu import build
    from pose2room_amd.p2rnet.modules import vote_center
    net, cfg = build('train', 256, device=dev)
    mod = net.centervoting.to(dev)
    ref = _clone_module(mod)
    mod.train(train), ref.train(train)
    B, S = 3, 512
    seed_xyz = _rand((B, S, 53, 3), 1, dev)
    feats = _rand((B, S, 256), 2, dev).requires_grad_(True)
    feats_r = feats.detach().clone().requires_grad_(True)
    gx, gf = _rand((B, S, 3), 3, dev), _rand((B, S, 256), 4, dev)
    assert vote_center.USE_FUSED_HEAD
    xyz, f = mod(seed_xyz, feats)
    (xyz * gx).sum().add((f * gf).sum()).backward()
    vote_center.USE_FUSED_HEAD = False
    try:
        xyz_r, f_r = ref(seed_xyz, feats_r)
        (xyz_r * gx).sum().add((f_r * gf).sum()).backward()
    finally:
        vote_center.USE_FUSED_HEAD = True
    assert _rel(xyz, xyz_r) < 1e-5 and _rel(f, f_r) < 1e-5
    assert _rel(feats.grad, feats_r.grad) < 2e-4
    for (n, p), (_, q) in zip(mod.named_parameters(), ref.named_parameters()):
        assert p.grad is not None, n
        assert _rel(p.grad, q.grad) < 2e-4, n
    for (n, a), (_, b) in zip(mod.named_buffers(), ref.named_buffers()):
        assert _rel(a.float(), b.float()) < 1e-5, n


@pytest.mark.parametrize("train", [True, False])
def test_proposal_heads_match_module_chain(dev, train):
    from tests.test_model_cpu import build
    from pose2room_amd.p2rnet.modules import proposal_net
    from pose2room_amd.p2rnet import pw_op
    net, cfg = build('train', 256, device=dev)
    mod = net.detection.to(dev)
    ref = _clone_module(mod)
    mod.train(train), ref.train(train)
    B, K, G = 4, 128, 100
    feats = _rand((B, 256, K), 1, dev).requires_grad_(True)
    feats_r = feats.detach().clone().requires_grad_(True)
    eps = {'center': _rand((B * K, G, 1, 3), 2, dev), 'size': _rand((B * K, G, 1, 3), 3, dev),
           'heading': _rand((B * K, G, 1, 2), 4, dev).double()}
    gouts = [_rand((B, 3, K), 5, dev), _rand((B, 3, K), 6, dev), _rand((B, 2, K), 7, dev).double(),
             _rand((B, 24, K), 8, dev)]
    assert pw_op.proposal_heads_supported(mod, feats)
    outs = pw_op.proposal_heads(mod, feats, eps)
    sum((o * g).sum().double() for o, g in zip(outs, gouts)).backward()
    outs_r = [ref.gmm_center.predict(ref.conv_center(feats_r), eps=eps['center']),
              ref.gmm_size.predict(ref.conv_size(feats_r), eps=eps['size']),
              ref.gmm_heading.predict(ref.conv_heading(feats_r), eps=eps['heading']), ref.conv_sem_obj(feats_r)]
    sum((o * g).sum().double() for o, g in zip(outs_r, gouts)).backward()
    for o, r, name in zip(outs, outs_r, ('center', 'size', 'heading', 'sem_obj')):
        assert o.shape == r.shape and o.dtype == r.dtype, name
        assert _rel(o, r) < 2e-5, name
    assert _rel(feats.grad, feats_r.grad) < 2e-4
    used = {id(p) for p in pw_op._proposal_params(mod)}
    worst = {}
    for (n, p), (_, q) in zip(mod.named_parameters(), ref.named_parameters()):
        if id(p) not in used:
            continue
        assert p.grad is not None and q.grad is not None, n
        assert p.grad.dtype == q.grad.dtype and p.grad.shape == q.grad.shape, n
        worst[n] = _rel(p.grad, q.grad)
    bad = {n: e for n, e in worst.items() if e > 3e-4}
    assert not bad, bad
    for (n, a), (_, b) in zip(mod.named_buffers(), ref.named_buffers()):
        assert _rel(a.float(), b.float()) < 1e-5, n
    print('proposal heads worst grad rel err', max(worst.values()), max(worst, key=worst.get))


def test_proposal_heads_generate_means_and_pi(dev):
    from tests.test_model_cpu import build
    from pose2room_amd.p2rnet import pw_op
    net, cfg = build('test', 256, device=dev)
    mod = net.detection.to(dev).eval()
    feats = _rand((2, 256, 128), 1, dev)
    with torch.no_grad():
        pc, ps, ph, sem, pis = pw_op.proposal_heads(mod, feats, False, return_pi=True)
        kw = dict(return_pi=True, multi_modes=False, n_samples=1)
        rc, pic = mod.gmm_center.generate(mod.conv_center(feats), **kw)
        rs, pis_ = mod.gmm_size.generate(mod.conv_size(feats), **kw)
        rh, pih = mod.gmm_heading.generate(mod.conv_heading(feats), **kw)
        rsem = mod.conv_sem_obj(feats)
    for a, b in ((pc, rc), (ps, rs), (ph, rh), (sem, rsem), (pis[0], pic), (pis[1], pis_), (pis[2], pih)):
        assert a.shape == b.shape and a.dtype == b.dtype
        assert _rel(a, b) < 2e-5


# ---- seams of the backbone (csrc/seed_ops.hip) ---------------------------------------------------------------------------
@pytest.mark.parametrize("B,C,T,J,S,kind", [(2, 64, 128, 53, 64, 'sorted'), (3, 8, 40, 5, 100, 'dup'), (2, 4, 16, 7, 300, 'many'),
                                            (1, 64, 256, 53, 512, 'random')])
def test_seed_rows_match_advanced_indexing(dev, B, C, T, J, S, kind):
    from pose2room_amd.p2rnet import seed_op
    g = torch.Generator().manual_seed(5)
    if kind == 'sorted':
        inds = torch.sort(torch.stack([torch.randperm(T, generator=g)[:S] for _ in range(B)]), dim=1)[0]
    elif kind == 'many':        # far more seeds than frames: more hits per frame than the kernel's list holds
        inds = torch.randint(0, 3, (B, S), generator=g)
    else:
        inds = torch.randint(0, T, (B, S), generator=g)
        if kind == 'dup':
            inds = torch.sort(inds, dim=1)[0]
    inds = inds.to(dev)
    x = _rand((B, C, T, J), 1, dev).requires_grad_(True)
    xr = x.detach().clone().requires_grad_(True)
    rows = seed_op.seed_rows(x, inds)
    ref = xr.permute(0, 2, 1, 3)[torch.arange(B, device=dev)[:, None], inds].reshape(B, S, -1)
    assert torch.equal(rows, ref)
    gout = _rand((B, S, C * J), 2, dev)
    rows.backward(gout)
    ref.backward(gout)
    assert _rel(x.grad, xr.grad) < 1e-6
    assert torch.equal(x.grad == 0, xr.grad == 0)


@pytest.mark.parametrize("shape", [(2, 64, 128, 20), (3, 5, 7, 53), (1, 1, 1, 1), (2, 3, 1000, 64)])
def test_short_row_reductions(dev, shape):
    from pose2room_amd.p2rnet import seed_op
    x = _rand(shape, 1, dev).requires_grad_(True)
    xr = x.detach().clone().requires_grad_(True)
    m, mr = seed_op.mean_last(x), xr.mean(-1)
    assert _rel(m, mr) < 1e-6
    gm = _rand(shape[:-1], 2, dev)
    m.backward(gm), mr.backward(gm)
    assert _rel(x.grad, xr.grad) < 1e-6
    b = _rand(shape[:-1], 3, dev).requires_grad_(True)
    br = b.detach().clone().requires_grad_(True)
    a = _rand(shape, 4, dev).requires_grad_(True)
    ar = a.detach().clone().requires_grad_(True)
    y, yr = seed_op.add_broadcast_last(a, b), ar + br.unsqueeze(-1)
    assert torch.equal(y, yr)
    gy = _rand(shape, 5, dev)
    y.backward(gy), yr.backward(gy)
    assert torch.equal(a.grad, ar.grad) and _rel(b.grad, br.grad) < 2e-6


def test_embed3_statistics_from_input_moments(dev):
    """p2r_embed3_forward_stats: the (count, mean, M2) entries of the 3 -> 64 layer's output, derived from the moments of
    its three input rows, against the statistics of the output itself (fp64), incl. inputs far from the origin."""
    from pose2room_amd.p2rnet import tconv_op, bn_op
    torch.manual_seed(3)
    for B, L, off in ((2, 1060, 0.0), (3, 5000, 50.0), (1, 7, 0.0)):
        x = (torch.randn(B, 3, L, device=dev) * torch.tensor([1.0, 0.3, 2.0], device=dev)[None, :, None] + off).contiguous()
        conv = torch.nn.Conv1d(3, 64, 1).to(dev)
        out, stats = tconv_op.embed3(x, conv, want_stats=True)
        ref = torch.nn.functional.conv1d(x.double(), conv.weight.double(), conv.bias.double())
        assert _rel(out, ref) < 1e-6
        assert stats.shape == (1, 64, 3) and float(stats[0, 0, 0]) == B * L
        mean, var, _ = bn_op.moments(stats, B * L)
        assert _rel(mean, ref.mean((0, 2))) < 1e-6
        assert _rel(var, ref.var((0, 2), unbiased=False)) < 1e-5


@pytest.mark.parametrize("train", [True, False])
def test_votes_normalized_match_module_chain(dev, train):
    """the whole voting stage of P2RNet._votes (conv_input, offset / residual adds, unit-length normalisation, channel-major
    re-layout) fused vs the module chain + torch ops: values, layout contract and every gradient"""
    from tests.test_model_cpu import build
    from pose2room_amd.p2rnet import pw_op
    from pose2room_amd.p2rnet.modules import vote_center
    net, cfg = build('train', 256, device=dev)
    mod = net.centervoting.to(dev)
    ref = _clone_module(mod)
    mod.train(train), ref.train(train)
    B, S = 3, 512
    seed_xyz = _rand((B, S, 53, 3), 1, dev).requires_grad_(True)
    seed_xyz_r = seed_xyz.detach().clone().requires_grad_(True)
    feats = _rand((B, S, 256), 2, dev).requires_grad_(True)
    feats_r = feats.detach().clone().requires_grad_(True)
    gx, gf = _rand((B, S, 3), 3, dev), _rand((B, 256, S), 4, dev)
    assert pw_op.votes_normalized_supported(mod, seed_xyz, feats)
    xyz, f = pw_op.votes_normalized(mod, seed_xyz, feats)
    assert f.shape == (B, S, 256) and f.transpose(1, 2).is_contiguous()
    (xyz * gx).sum().add((f.transpose(1, 2) * gf).sum()).backward()
    vote_center.USE_FUSED_HEAD = False
    try:
        xyz_r, f_r = ref(seed_xyz_r, feats_r)
        f_r = f_r.div(torch.norm(f_r, p=2, dim=2).unsqueeze(2))
        (xyz_r * gx).sum().add((f_r.transpose(1, 2) * gf).sum()).backward()
    finally:
        vote_center.USE_FUSED_HEAD = True
    assert _rel(xyz, xyz_r) < 1e-5 and _rel(f, f_r) < 1e-5
    assert _rel(feats.grad, feats_r.grad) < 2e-4 and _rel(seed_xyz.grad, seed_xyz_r.grad) < 1e-6
    for (n, p), (_, q) in zip(mod.named_parameters(), ref.named_parameters()):
        assert _rel(p.grad, q.grad) < 2e-4, n


def test_nearest_prefix_equals_argmin_of_abs_difference(dev):
    """seed selection by arc length: the kernel must return torch.argmin's index for every target, incl. plateaus of the
    cumulative arc length (repeated frames: exact ties, first index wins) and targets exactly between two prefixes."""
    from pose2room_amd.p2rnet import seed_op
    g = torch.Generator().manual_seed(9)
    for B, T, S in ((4, 256, 512), (3, 1024, 512), (2, 341, 100), (1, 2048, 512), (1, 20000, 64), (2, 40000, 16)):  # > 16384 frames: beyond the default 64 KB of LDS
        step = torch.rand(B, T - 1, generator=g)
        step[torch.rand(B, T - 1, generator=g) < 0.3] = 0.0              # plateaus
        step = (step * 8).round() / 8                                     # exactly representable: exact mid-point ties
        cum = torch.cumsum(torch.cat([torch.zeros(B, 1), step], 1).double(), 1).float().to(dev)
        stride = cum[:, -1] / (S - 1)
        target = stride.unsqueeze(-1) * torch.arange(S, dtype=torch.float, device=dev)
        want = torch.argmin(torch.abs(cum.unsqueeze(-1) - target.unsqueeze(1)), dim=1)
        # torch's own contract: the first minimum (checked on the host so that the test does not depend on the device reduce)
        d = torch.abs(cum.unsqueeze(-1) - target.unsqueeze(1)).cpu()
        first = torch.stack([torch.tensor([int((d[b, :, s] == d[b, :, s].min()).nonzero()[0]) for s in range(S)]) for b in range(B)])
        got = seed_op.nearest_prefix(cum, target)
        assert torch.equal(got.cpu(), first)
        assert torch.equal(got, want)
