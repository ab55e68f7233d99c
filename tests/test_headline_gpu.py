"""GPU: the BENCHMARKED shape (BASELINE configs[2]: bs=32, T=1024, J=53) as a parity test, not only as a bench.

At this size the persistent-workgroup kernels walk 2048 tiles with 256 workgroups and 32 sequences -- index
arithmetic the small test shapes never reach.  The ST-GCN backbone on the fused HIP kernels (train mode: graph conv
forward / dX / dW / dA, temporal conv, BatchNorm passes and epilogue statistics, embedding MLPs) is compared with
the SAME weights run through plain torch modules (the reference's formulation: conv1x1 to 704 channels + einsum,
nn.BatchNorm2d, nn.Conv2d) on the same GPU: seed features and every backbone parameter gradient.  A mis-indexed tile
shows as an O(1) error; the tolerances (1e-3 forward, 1.5e-2 backward of each tensor's largest entry = about twice
the measured worst, 6.9e-3 on a graph-conv weight of the fifth block) only absorb
train-mode BatchNorm's amplification of fp32 summation-order noise over six blocks."""
import contextlib
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _chunked_graph_conv(self, x, A, *a, **k):
    """The reference formulation (conv1x1 to K*64 channels, then the graph einsum) four samples at a time: at bs=32,
    T=1024 the 704-channel tensor is 4.9 GB, and torch's own conv / einsum backward on tensors beyond 2^31 bytes
    returns a wrong weight gradient on this stack (measured with tools/dev_dw_check.py: relative error 2.5 unchunked,
    1e-6 in chunks) -- the plain side of this test must not depend on that."""
    outs = []
    for i in range(0, x.shape[0], 4):
        y = self.conv(x[i:i + 4])
        n, kc, t, v = y.size()
        outs.append(torch.einsum('nkctv,kvw->nctw', (y.view(n, self.kernel_size, kc // self.kernel_size, t, v), A)))
    return torch.cat(outs).contiguous(), A


@contextlib.contextmanager
def _plain_torch(net):
    """Every fused dispatch of the backbone switched off: the host modules run as plain torch chains."""
    import types
    from pose2room_amd.p2rnet import bn_op, tconv_op
    from pose2room_amd.p2rnet.modules import proposal_net, vote_center
    saved = (bn_op.supported, tconv_op.supported_embed3)
    bn_op.supported = lambda *a, **k: False
    tconv_op.supported_embed3 = lambda *a, **k: False
    proposal_net.USE_FUSED_HEADS = vote_center.USE_FUSED_HEAD = False     # nn.Conv1d / nn.BatchNorm1d chains
    for b in net.backbone.st_gcn_networks:
        b.gcn.forward = types.MethodType(_chunked_graph_conv, b.gcn)
    for b in net.backbone.st_gcn_networks:
        b.gcn.fused = False
        b.fused_bn = False
        b.fused_tconv = False
    try:
        yield
    finally:
        bn_op.supported, tconv_op.supported_embed3 = saved
        proposal_net.USE_FUSED_HEADS = vote_center.USE_FUSED_HEAD = True


def test_backbone_at_bench_shape_fused_vs_plain(dev):
    from pose2room_amd.p2rnet import P2RConfig, default_config, METHODS
    from pose2room_amd.p2rnet.synthetic import make_batch
    B, T = 32, 1024
    cfg = P2RConfig(default_config('train', data={'num_frames': T}), device=dev)
    torch.manual_seed(42)
    net = METHODS.get('P2RNet')(cfg).to(dev).train()
    ref = copy.deepcopy(net)
    joints = make_batch(B, T, seed=1234, device=dev)['input_joints']
    g = torch.Generator().manual_seed(5)
    go = torch.randn(B, 512, 256, generator=g).to(dev)

    def run(model):
        ep = model.backbone(joints, {})
        sf = ep['seed_features']
        sf.backward(go)
        grads = {k: p.grad.detach().clone() for k, p in model.backbone.named_parameters() if p.grad is not None}
        return ep['seed_inds'].clone(), sf.detach().clone(), grads

    inds_a, sf_a, g_a = run(net)
    with _plain_torch(ref):
        inds_b, sf_b, g_b = run(ref)
    torch.cuda.synchronize()
    assert torch.equal(inds_a, inds_b)
    assert torch.isfinite(sf_a).all()
    scale = sf_b.abs().max().item()
    err = (sf_a - sf_b).abs().max().item()
    assert err <= 1e-3 * scale, f'seed_features: {err:.3e} vs scale {scale:.3e}'
    assert set(g_a) == set(g_b) and len(g_a) > 60
    worst = {}
    for k in g_b:
        # zero in exact arithmetic (conv bias in front of a train-mode BatchNorm): rounding noise on both sides
        if k.endswith('gcn.conv.bias') or k.endswith('tcn.2.bias') or (k.endswith('conv.bias') and 'embed' in k):
            continue
        s = g_b[k].abs().max().item()
        if s < 1e-8:
            continue
        worst[k] = (g_a[k] - g_b[k]).abs().max().item() / s
    bad = {k: v for k, v in worst.items() if not v <= 1.5e-2}
    print('bench-shape backbone: worst relative gradient error', max(worst.values()), max(worst, key=worst.get))
    for k in sorted(worst, key=worst.get, reverse=True):          # the per-tensor table goes to the log (pytest -s / -rP)
        print(f'  {worst[k]:.3e}  {k}')
    assert not bad, bad


def test_train_step_at_bench_shape(dev, mathmode):
    """One full train step at bs=32, T=1024: finite losses, parameters move, and the fused detection loss equals its
    torch composition on the very tensors of that step."""
    import math
    import bench
    from pose2room_amd.p2rnet.synthetic import make_batch
    trainer, cfg = bench.build_trainer(dev, 1024, 1)
    batch = make_batch(32, 1024, seed=1234, device=dev)
    before = {k: v.detach().clone() for k, v in trainer.net.module.named_parameters()}
    out = trainer.train_step(dict(batch))
    assert all(math.isfinite(v) for v in out.values()), out
    moved = sum(int(not torch.equal(before[k], v)) for k, v in trainer.net.module.named_parameters())
    assert moved > 100
    net = trainer.net.module
    with torch.no_grad():
        est = net(batch)
        fused = net.detection_loss(est, batch, None)
        comp = net.detection_loss.composed(est, batch, None)
    for k in comp:
        a, b = float(fused[k]), float(comp[k])
        assert abs(a - b) <= 2e-5 * max(1.0, abs(b)), (k, a, b)
    v = bench.verify_bench_shape(trainer, batch)
    assert v['losses_finite'] and v['seed_inds_equal']


def test_forward_and_loss_at_config1_shape(dev, mathmode):
    """BASELINE configs[1] at its full size (bs=8, T=512, J=53): P2RNet forward + detection loss on the HIP path
    (fused backbone, vote / proposal heads on the job-list kernels, fused vote aggregation, fused loss) against the same
    weights through the plain torch chain (reference formulation of the graph conv, nn.BatchNorm, nn.Conv1d heads, the
    composed loss), train mode, same mixture noise.  Votes to 1e-3 of range (train-mode BatchNorm amplification over six
    blocks); the proposal indices are a discrete function of them, so everything behind is compared when they agree
    (they do at this seed) and the agreement itself is asserted."""
    from pose2room_amd.p2rnet import P2RConfig, default_config, METHODS
    from pose2room_amd.p2rnet.synthetic import make_batch
    B, T = 8, 512
    cfg = P2RConfig(default_config('train', data={'num_frames': T}), device=dev)
    torch.manual_seed(7)
    net = METHODS.get('P2RNet')(cfg).to(dev).train()
    ref = copy.deepcopy(net)
    batch = make_batch(B, T, seed=512, device=dev)
    g = torch.Generator().manual_seed(11)
    K, G = 128, cfg.config['data']['num_gaussian']
    eps = {'center': torch.randn(B * K, G, 1, 3, generator=g).to(dev), 'size': torch.randn(B * K, G, 1, 3, generator=g).to(dev),
           'heading': torch.randn(B * K, G, 1, 2, generator=g, dtype=torch.float64).to(dev)}
    with torch.no_grad():
        est_a = net(dict(batch), eps=eps)
        loss_a = net.loss(est_a, batch)
        with _plain_torch(ref):
            est_b = ref(dict(batch), eps=eps)
            loss_b = ref.detection_loss.composed(est_b, batch, cfg.dataset_config)
    torch.cuda.synchronize()

    def rel(k):
        a, b = est_a[k].double(), est_b[k].double()
        return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()

    assert torch.equal(est_a['seed_inds'], est_b['seed_inds'])
    for k in ('seed_features', 'vote_xyz', 'vote_features'):
        assert rel(k) <= 1e-3, (k, rel(k))
    assert torch.equal(est_a['aggregated_vote_inds'], est_b['aggregated_vote_inds']), \
        'rounding differences of the votes flipped a proposal choice at this seed'
    worst = {k: rel(k) for k in ('aggregated_vote_xyz', 'center', 'size', 'heading', 'objectness_scores', 'sem_cls_scores')}
    assert max(worst.values()) <= 2e-3, worst
    for k in loss_b:
        a, b = float(loss_a[k]), float(loss_b[k])
        assert abs(a - b) <= 2e-3 * max(1.0, abs(b)), (k, a, b)
    print('configs[1] forward + loss: worst relative output error', worst)


# ---- G10: the BASELINE config sizes against vectors captured from the IMPORTED REFERENCE --------------------------
# (tests/golden/make_headline_golden.py; the twin tests above stay as the O(1) indexing check on the same GPU).
import os

import numpy as np

G10 = os.path.join(os.path.dirname(__file__), 'golden', 'g10_headline.npz')
# Tolerances are the north star's 1e-4 of each tensor's range; measured on MI355X against the reference's CPU run, train-
# mode BatchNorm: 2.4e-6 .. 5.1e-6 on every end point at bs=32, T=1024 (see the prints of each test).
TOL_TRAIN_FWD = 1e-4
# relative gap (of the squared distances compared) below which a discrete choice behind the votes -- an FPS pick, a
# ball membership -- may legitimately differ between two fp32 evaluations of the same network (votes agree to ~3e-6)
GAP_NOISE = 1e-4


def _ref_net(T, dev):
    from tests.test_model_cpu import build
    net, cfg = build('train', T, device=dev)
    return net.to(dev), cfg


def _mixture_noise(B, dev):
    """The CPU stream the reference drew its mixture noise from (torch.manual_seed(123) in the generator)."""
    g = torch.Generator().manual_seed(123)
    eps = {}
    for head, dt, D in (('center', torch.float32, 3), ('size', torch.float32, 3), ('heading', torch.float64, 2)):
        eps[head] = torch.empty(B * 128, 100, 1, D, dtype=dt).normal_(generator=g).to(dev)
    return eps


def _relmax(got, ref):
    got = np.asarray(got, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30))


def _ball_gaps(vote_xyz, inds, radius=0.3):
    """(B, 128): for every proposal centre of the REFERENCE run, the smallest relative distance of any
    |centre - vote|^2 to radius^2 -- a membership that close may flip under fp32 noise of the votes."""
    x = np.asarray(vote_xyz, dtype=np.float64)
    c = np.take_along_axis(x, np.asarray(inds)[:, :, None].astype(np.int64), 1)
    d2 = ((c[:, :, None, :] - x[:, None, :, :]) ** 2).sum(-1)
    return np.abs(d2 - radius * radius).min(-1) / (radius * radius)


def _check_forward_and_loss(z, tag, ep, loss, tol, tol_loss):
    """End points + 10 losses of one forward against the reference's.  Returns the measured errors."""
    m = {}
    assert np.array_equal(ep['seed_inds'].cpu().numpy(), z[f'{tag}_seed_inds'])
    m['vote_xyz'] = _relmax(ep['vote_xyz'].cpu().numpy(), z[f'{tag}_vote_xyz'])
    for k in ('seed_features', 'vote_features'):
        got = ep[k][:, ::16, ::8].cpu().numpy()
        m[k] = float(np.abs(got - z[f'{tag}_{k}_sub']).max() / z[f'{tag}_{k}_sum'][2])
    for k in ('vote_xyz', 'seed_features', 'vote_features'):
        assert m[k] <= tol, (k, m[k])
    # proposals: FPS on the votes, then ball grouping -- discrete functions of fp32 values
    ref_inds = z[f'{tag}_aggregated_vote_inds']
    got_inds = ep['aggregated_vote_inds'].cpu().numpy()
    same = (got_inds == ref_inds).all(1)
    fps_gap = z[f'{tag}_fps_gap']
    for b in np.nonzero(~same)[0]:
        assert fps_gap[b] < GAP_NOISE, f'sample {b}: proposal set differs although the FPS margin is {fps_gap[b]:.2e}'
    m['samples_with_other_proposals'] = int((~same).sum())
    # measured on MI355X at these seeds, both arithmetic modes: every sample picks the reference's proposals (1.00); a
    # sample may differ only through an FPS decision inside the noise band (asserted above), and only a few may
    assert same.mean() >= 0.95, same.mean()
    solid = _ball_gaps(z[f'{tag}_vote_xyz'], ref_inds) >= GAP_NOISE          # (B, 128)
    solid &= same[:, None]
    m['solid_proposals'] = float(solid.mean())
    assert solid.mean() >= 0.98, solid.mean()          # measured 0.993 .. 0.996
    for k in ('aggregated_vote_xyz', 'center', 'size', 'heading', 'objectness_scores', 'sem_cls_scores'):
        got = ep[k].cpu().numpy(); ref = z[f'{tag}_{k}']
        assert got.dtype == ref.dtype, (k, got.dtype)
        scale = np.abs(ref).max()
        err = np.abs(got.astype(np.float64) - ref).reshape(ref.shape[0], ref.shape[1], -1).max(-1) / scale
        m[k] = float(err[solid].max())
        assert m[k] <= tol, (k, m[k])
        # a proposal one of whose memberships sits inside the noise band is still the same proposal: bounded change
        m[k + '_ambiguous'] = float(err[same].max())
    for k, v in loss.items():
        assert str(v.dtype) == str(z[f'{tag}_lossdtype_{k}']), (k, v.dtype)
        a, b = float(v), float(z[f'{tag}_loss_{k}'])
        m['loss_' + k] = abs(a - b) / max(1.0, abs(b))
        # the losses are batch means: with every sample on the reference's proposals they hold tol_loss; a sample on
        # another proposal set (FPS decision inside the noise band) moves them by at most its share of the batch
        slack = tol_loss if same.all() else tol_loss + float((~same).mean())
        assert m['loss_' + k] <= slack, (k, a, b, int((~same).sum()))
    return m


@pytest.mark.parametrize('tag', ['g10a', 'g10b'])
def test_g10_forward_and_loss_vs_reference(dev, tag, mathmode):
    """BASELINE configs[1] (bs=8, T=512) and configs[2] (bs=32, T=1024): P2RNet.forward + BoxNetDetectionLoss on the HIP
    path, train-mode BatchNorm, against the REFERENCE's own run of the same weights / batch / mixture noise
    (/root/reference/models/p2rnet/modules/network.py:75-106, models/loss.py:152-189, imported in the build
    container by tests/golden/make_headline_golden.py)."""
    from pose2room_amd.p2rnet.synthetic import make_batch
    z = np.load(G10)
    B, T, seed = {'g10a': (8, 512, 612), 'g10b': (32, 1024, 1234)}[tag]
    net, cfg = _ref_net(T, dev)
    net.train()
    batch = make_batch(B, T, seed=seed, device=dev)
    with torch.no_grad():
        ep = net(dict(batch), eps=_mixture_noise(B, dev))
        loss = net.loss(ep, batch)
    torch.cuda.synchronize()
    m = _check_forward_and_loss(z, tag, ep, loss, TOL_TRAIN_FWD, 1e-4)
    print(tag, 'vs reference:', {k: (f'{v:.2e}' if isinstance(v, float) else v) for k, v in m.items()})


def _packed_err(z, key, got):
    """max |got - fixture| over the fixture's strided sample, absolute; plus the tensor's largest reference entry."""
    ref = z[key + '_val'].astype(np.float64)
    stride = int(z[key + '_stride'])
    g = got.detach().flatten()[::stride].double().cpu().numpy()
    assert g.shape == ref.shape, (key, g.shape, ref.shape)
    return np.abs(g - ref), float(z[key + '_sum'][2])


def test_g10c_backbone_backward_vs_reference(dev, mathmode):
    """bs=8, T=1024, train-mode BatchNorm: seeded cotangents on (vote_xyz, vote_features) back-propagated through the
    backbone + voting on the HIP kernels; EVERY backbone / voting parameter gradient against the reference's autograd
    (stgcn_layers.py:50-67,399-439) evaluated in FLOAT64.  Two fp32 evaluations of these gradients differ visibly, for
    two reasons the fixture lets the test quantify instead of guessing a tolerance:
      * train-mode BatchNorm backward cancels most of its input (mean and projection removed): the fixture says how far
        the reference's OWN float32 run is from its float64 run, per tensor (1e-3 .. 4e-3 of the largest entry through the
        six blocks) -- our error may be FACTOR times that;
      * a ReLU bit: the forward agrees to 3e-6, so of the 4096 x 256 pre-activations of a vote-head layer (and of the 28 M
        of an ST-GCN gate) a few sit inside the noise and take the other branch.  One such element moves a BatchNorm-bias
        gradient by up to 1/64 of its scale and perturbs everything upstream by a rank-one term (round 5 measured 5.2e-3
        on centervoting.conv_input.1.batchnorm.bias and covered it with a blanket bound of 1e-2 on ~78 tensors).  The
        fixture now carries the reference's near-zero pre-activations of all fourteen gate layers (`g10c_gate_*`,
        make_headline_golden.py g): GateForcer finds the gates that are in another state here, asserts each lies inside
        the recorded band, and pins it to the reference's state BEFORE the fused kernels evaluate it -- the backward then
        runs with the reference's gates and every tensor is held at FACTOR times the reference's own float32 error
        (floor: the north star's 1e-4 of the tensor's largest entry).
    As a whole our fp32 run must be as close to float64 as the reference's fp32 run: median over the tensors <= 2x.
    Gradients that vanish analytically (a conv bias in front of a train-mode BatchNorm) are compared on the scale of
    the largest gradient in the net."""
    from tests import cases
    from pose2room_amd.p2rnet.synthetic import make_batch
    from tests.test_model_cpu import GateForcer
    FACTOR, FLOOR_REL = 4.0, 1e-4
    z = np.load(G10)
    B, T = 8, 1024
    net, cfg = _ref_net(T, dev)
    net.train(); net.zero_grad()
    batch = make_batch(B, T, seed=2024, device=dev)
    with GateForcer(net, z, 'g10c') as gates:
        xyz, feats, ep = net._votes(batch)
        gx, gf = cases.seam_cotangents(B)
        torch.autograd.backward([xyz, feats], [gx.to(dev), gf.to(dev)])
    # fourteen gate layers were looked at; the pinned gates are few and all inside the reference's near-zero band
    # (GateForcer asserts the band); the count is bounded so that a real disagreement cannot hide behind the pinning
    ncand = sum(int(z[k].size) for k in z.files if k.startswith('g10c_gate_') and k.endswith('_idx'))
    print(f'g10c gates: {len(gates.forced)} of {ncand} near-zero candidates pinned:', gates.forced[:12])
    assert gates.layers == 14 and len(gates.seen) == 14, gates.seen
    assert len(gates.forced) <= max(8, ncand // 100), (len(gates.forced), ncand)
    assert np.array_equal(ep['seed_inds'].cpu().numpy(), z['g10c_seed_inds'])
    fwd = {'vote_xyz': _relmax(xyz.detach().cpu().numpy(), z['g10c_vote_xyz']),
           'vote_features': float(np.abs(feats.detach()[:, ::16, ::8].cpu().numpy() - z['g10c_vote_features_sub']).max()
                                  / z['g10c_vote_features_sum'][2])}
    assert max(fwd.values()) <= TOL_TRAIN_FWD, fwd
    params = dict(net.named_parameters())
    names = [str(n) for n in z['g10c_names']]
    assert len(names) >= 80 and set(names) == {n for n in params if n.startswith(('backbone.', 'centervoting.'))}
    floor = 1e-3 * max(float(z[f'g10c_grad_{n}_sum'][2]) for n in names)
    rows, bad = [], {}
    for n in names:
        err, scale = _packed_err(z, f'g10c_grad_{n}', params[n].grad)
        ref32 = float(z[f'g10c_grad_{n}_ref32']) * scale               # the reference's own fp32 error, absolute
        behind_every_relu = n.startswith('centervoting.conv_input.2.')
        bound = max(FACTOR * ref32, (2e-4 if behind_every_relu else FLOOR_REL) * max(scale, floor))
        rows.append((err.max() / max(scale, floor), ref32 / max(scale, floor), n))
        if not err.max() <= bound:
            bad[n] = f'err {err.max():.3e} > bound {bound:.3e} (reference fp32 {ref32:.3e}, largest entry {scale:.3e})'
    rows.sort(reverse=True)
    print('g10c forward', fwd, '(reference fp32 vs fp64 on vote_xyz:', float(z['g10c_vote_xyz_ref32']), ')')
    print('  ours-vs-fp64   reference-fp32-vs-fp64   (of the largest entry)')
    for e, r, n in rows:
        print(f'  {e:.3e}  {r:.3e}  {n}' + ('   <-- over the bound' if n in bad else ''))
    assert not bad, bad
    ours = float(np.median([e for e, _, _ in rows])); theirs = float(np.median([r for _, r, _ in rows]))
    print(f'median over {len(rows)} tensors: ours {ours:.3e}, reference fp32 {theirs:.3e}')
    assert ours <= 2 * theirs, (ours, theirs)


def test_g10e_eval_bn_step_vs_reference(dev, mathmode):
    """bs=8, T=1024 with every BatchNorm on running statistics: the whole step end to end at the north star's 1e-4
    -- end points, 10 losses, d total / d pred_center, ~130 parameter gradients.  The gradient at the seam is compared
    element-wise EXCEPT that the max-pool of the vote aggregation routes a ball's gradient to the arg-max vote: where
    two votes of a ball tie within fp32 noise (8 x 128 x 16 x 256 candidates) the gradient of that channel lands on the
    other vote, so up to 1e-3 of its entries may differ while its sums hold 1e-4."""
    from tests.test_model_cpu import check_packed
    from pose2room_amd.p2rnet.synthetic import make_batch
    z = np.load(G10)
    B, T = 8, 1024
    net, cfg = _ref_net(T, dev)
    net.train()
    for mod in net.modules():
        if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
            mod.eval()
    net.zero_grad()
    batch = make_batch(B, T, seed=2024, device=dev)
    batch['center_label'] = torch.from_numpy(z['g10e_center_label']).to(dev)
    ep = net(dict(batch), eps=_mixture_noise(B, dev))
    for k in ('vote_xyz', 'vote_features', 'center'):
        ep[k].retain_grad()
    loss = net.loss(ep, batch)
    loss['total'].backward()
    torch.cuda.synchronize()
    tol = 1e-4
    m = _check_forward_and_loss(z, 'g10e', {k: (v.detach() if torch.is_tensor(v) else v) for k, v in ep.items()},
                                {k: v.detach() for k, v in loss.items()}, tol, tol)
    assert m['samples_with_other_proposals'] == 0
    m['dcenter'] = _relmax(ep['center'].grad.cpu().numpy(), z['g10e_dcenter'])
    assert m['dcenter'] <= tol, m['dcenter']
    for k in ('vote_xyz', 'vote_features'):
        err, scale = _packed_err(z, f'g10e_d{k}', ep[k].grad)
        m[f'd{k}_max'] = float(err.max() / scale)
        m[f'd{k}_outliers'] = float((err > tol * scale).mean())
        assert m[f'd{k}_outliers'] <= 1e-3, (k, m)
        g = ep[k].grad.double()
        ref = z[f'g10e_d{k}_sum']
        m[f'd{k}_abssum'] = abs(g.abs().sum().item() - ref[1]) / ref[1]
        assert m[f'd{k}_abssum'] <= tol, (k, m)
    params = dict(net.named_parameters())
    names = [str(n) for n in z['g10e_names']]
    assert sum(n.startswith('detection.') for n in names) > 20
    floor = 1e-3 * max(float(z[f'g10e_grad_{n}_sum'][2]) for n in names)
    worst = {n: check_packed(z, f'g10e_grad_{n}', params[n].grad, float('inf'), floor, n) for n in names}
    print('g10e vs reference:', {k: (f'{v:.2e}' if isinstance(v, float) else v) for k, v in m.items()})
    for n in sorted(worst, key=worst.get, reverse=True)[:8]:
        print(f'  {worst[n]:.3e}  {n}')
    # Parameters BEHIND the max-pool (the proposal heads) hold 1e-4.  The max-pool routes a ball's gradient to the arg-max
    # vote; ~1e-3 of the routes differ between two fp32 evaluations (above: `dvote_features_outliers`, up to 1e-2 of the
    # largest entry each).  Two classes in front of it:
    #  * the shared MLP of the vote aggregation sits directly under the routing: its gradients move with WHICH routes
    #    differ (measured 9.0e-4 in exact mode, 3.5e-3 in split16 mode -- with FEWER differing routes, 7.3e-4 against
    #    9.8e-4 of the entries: the mode changes which ties fall which way, not the size of the effect).  These tensors
    #    are produced by the exact vote-aggregation kernels in BOTH modes: bound 5e-3;
    #  * voting and the backbone -- where the split16 kernels live -- receive the routed gradient through d vote_features,
    #    an L2 perturbation of ~3e-4 of the cotangent: bound 5e-4 (was 2e-3; measured 3.5e-4 exact, 1.8e-4 split16).
    behind = lambda n: n.startswith('detection.') and not n.startswith('detection.vote_aggregation.')   # noqa: E731
    routed = lambda n: n.startswith('detection.vote_aggregation.')                                      # noqa: E731
    bad = {n: v for n, v in worst.items() if not v <= (tol if behind(n) else (5e-3 if routed(n) else 5e-4))}
    assert not bad, bad
    assert max(v for n, v in worst.items() if behind(n)) <= tol


def test_g10f_long_sequence_eval_vs_reference(dev, oracle, mathmode):
    """BASELINE configs[4]'s stress length through the EVALUATION path (test_epoch.py:18-46 on one batch): bs=2, T=2048,
    `generate(data, eval=True)` -- network, prediction parsing, far-box filter, NMS on device, per-class lists -- the
    loss dict of `Tester.test_step` and `APCalculator.compute_metrics()` at IoU 0.25 / 0.5, against the imported
    reference's run (G10f).  End points 1e-4; keep mask by decisions (tests/test_eval_gpu.py); metrics equal."""
    from tests.test_model_cpu import build
    from tests.test_eval_gpu import _check_keep_masks, _far_box_margin
    from pose2room_amd.net_utils.ap_helper import APCalculator
    from pose2room_amd.p2rnet.synthetic import make_batch
    z = np.load(G10)
    far = bool(int(z['g10f_remove_far_box']))
    net, cfg = build('test', 2048, device=dev, remove_far_box=far)
    net = net.to(dev).eval()
    data = make_batch(2, 2048, seed=4096, device=dev)
    with torch.no_grad():
        ep, eval_dict, parsed = net.generate(data, eval=True)
        loss = net.loss(ep, data)
    assert np.array_equal(ep['seed_inds'].cpu().numpy(), z['g10f_seed_inds'])
    assert np.array_equal(ep['aggregated_vote_inds'].cpu().numpy(), z['g10f_aggregated_vote_inds'])
    worst = {}
    for k in ('vote_xyz', 'aggregated_vote_xyz', 'center', 'size', 'heading', 'objectness_scores', 'sem_cls_scores'):
        worst[k] = _relmax(ep[k].cpu().numpy(), z[f'g10f_{k}'])
        assert worst[k] <= 1e-4, (k, worst[k])
    for k, v in loss.items():
        a, b = float(v), float(z[f'g10f_loss_{k}'])
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (k, a, b)
    ne_r = ne_o = margin_r = None
    if far:
        ne_r, margin_r = _far_box_margin(cfg, data, z['g10f_center'], z['g10f_size'], z['g10f_heading'])
        ne_o, _ = _far_box_margin(cfg, data, ep['center'].cpu(), ep['size'].cpu(), ep['heading'].cpu())
    _check_keep_masks(oracle, 'generate at T=2048', cfg.eval_config['nms_iou'],
                      (parsed['pred_corners_3d'], parsed['obj_prob']), (z['g10f_pred_corners_3d'], z['g10f_obj_prob']),
                      eval_dict['pred_mask'], z['g10f_pred_mask'], ne_o, ne_r, margin_r)
    assert [len(l) for l in eval_dict['batch_pred_map_cls']] == z['g10f_n_pred'].tolist()
    for thr in (0.25, 0.5):
        calc = APCalculator(thr, getattr(cfg.dataset_config, 'class2type', None), False, device=dev)
        calc.step(eval_dict['batch_pred_map_cls'], eval_dict['batch_gt_map_cls'])
        metrics = calc.compute_metrics()
        keys = [str(k) for k in z[f'g10f_metric_keys_{int(thr * 100)}']]
        assert sorted(metrics) == keys
        got = np.array([float(metrics[k]) for k in keys])
        np.testing.assert_allclose(got, z[f'g10f_metric_vals_{int(thr * 100)}'], rtol=1e-6, atol=1e-9, equal_nan=True)
    print('g10f vs reference:', {k: f'{v:.2e}' for k, v in worst.items()})
