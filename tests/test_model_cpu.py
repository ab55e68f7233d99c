"""CPU: our host model (pose2room_amd.p2rnet) against golden vectors captured from
the imported reference (tests/golden/make_model_golden.py).  The pointnet2 ops
and nn_distance run on the CPU oracle here (oracle.cpu_backend.cpu_ops); the GPU
twin of this file (test_model_gpu.py) runs the same checks on the HIP path.

Tolerance: fp32 outputs within 1e-4 (north star); index outputs exact."""
import os

import numpy as np
import pytest
import torch

from tests import cases

G = os.path.join(os.path.dirname(__file__), "golden", "g345_model.npz")
TOL = dict(rtol=1e-4, atol=1e-4)


def build(mode, num_frames, device='cpu', **test_over):
    from pose2room_amd.p2rnet import P2RConfig, default_config, METHODS
    cfg = P2RConfig(default_config(mode, data={'num_frames': num_frames}, test=test_over), device=device)
    torch.manual_seed(0); np.random.seed(0)
    net = METHODS.get('P2RNet')(cfg)
    cases.fill_weights(net)
    return net, cfg


def check_endpoints(z, tag, ep):
    for k in ['seed_inds', 'aggregated_vote_inds']:
        assert np.array_equal(ep[k].cpu().numpy(), z[f'{tag}_{k}']), k
    for k in ['aggregated_vote_xyz', 'vote_xyz', 'center', 'size', 'heading', 'objectness_scores', 'sem_cls_scores']:
        got = ep[k].detach().cpu().numpy()
        assert got.dtype == z[f'{tag}_{k}'].dtype, (k, got.dtype)
        np.testing.assert_allclose(got, z[f'{tag}_{k}'], err_msg=k, **TOL)
    for k in ['seed_features', 'vote_features']:
        np.testing.assert_allclose(ep[k][:, ::8, ::4].detach().cpu().numpy(), z[f'{tag}_{k}_sub'], err_msg=k, **TOL)
    if 'pi' in ep:
        for k, v in ep['pi'].items():
            np.testing.assert_allclose(v.detach().cpu().numpy(), z[f'{tag}_pi_{k}'], err_msg=k, **TOL)


def test_state_dict_layout():
    net, _ = build('train', 64)
    sd = net.state_dict()
    assert len(sd) == 219
    assert sum(p.numel() for p in net.parameters()) == 2043833
    assert len(list(net.parameters())) == 131
    assert sd['detection.gmm_heading.mdn.mu'].dtype == torch.float64
    assert sd['backbone.A'].shape == (11, 53, 53)
    for k in ['backbone.st_gcn_networks.0.gcn.conv.weight', 'backbone.st_gcn_networks.5.tcn.2.weight',
              'backbone.conv_joint.weight', 'centervoting.conv_input.2.conv.bias',
              'detection.vote_aggregation.mlp_module.2.weight', 'detection.gmm_size.mdn.pi.conv.weight',
              'detection.conv_sem_obj.2.conv.bias', 'backbone.pos_embed.1.batchnorm.running_var']:
        assert k in sd, k
    assert sd['backbone.st_gcn_networks.0.gcn.conv.weight'].shape == (704, 64, 1, 1)
    assert sd['backbone.conv_joint.weight'].shape == (256, 3392, 1)


@pytest.mark.parametrize("tag,B,T", [('g3u', 1, 768), ('g3f', 2, 512)])
def test_g3_eval_path(tag, B, T):
    from oracle.cpu_backend import cpu_ops
    from pose2room_amd.p2rnet.synthetic import make_batch
    z = np.load(G)
    net, cfg = build('test', T, remove_far_box=False)
    net.eval()
    assert abs(cases.state_checksum(net) - float(z[f'{tag}_wsum'][0])) < 1e-6 * float(z[f'{tag}_wsum'][0])
    data = make_batch(B, T, seed=100 + T)
    with torch.no_grad(), cpu_ops():
        ep = net.generate_end_points(data)
    check_endpoints(z, tag, ep)


def run_g4(net, cfg, data, z, device, ops_ctx):
    """Shared by the CPU and GPU twins: (1) end-to-end forward up to the votes with the
    conditioning-aware tolerance, (2) detection head + loss + backward + AdamW step from
    the recorded seam tensors, against the reference's numbers."""
    from pose2room_amd.p2rnet.training import load_optimizer
    net.train()
    opt = load_optimizer(cfg.config, net)
    opt.zero_grad()
    with ops_ctx():
        # (1) backbone + voting, end to end.  Train-mode BatchNorm divides by small batch
        # deviations and amplifies fp32 rounding differences (measured x26 over the six blocks with the hash fill), hence 1e-3.
        xyz, feats, ep = net._votes(data)
        assert np.array_equal(ep['seed_inds'].cpu().numpy(), z['g4_seed_inds'])
        np.testing.assert_allclose(xyz.detach().cpu().numpy(), z['g4_vote_xyz_full'], rtol=1e-3, atol=1e-3)
        np.testing.assert_allclose(feats.detach().cpu().numpy(), z['g4_vote_features_full'], rtol=1e-3, atol=1e-3)
        # (2) head + loss from the reference's seam tensors
        vx = torch.from_numpy(z['g4_vote_xyz_full']).to(device).requires_grad_(True)
        vf = torch.from_numpy(z['g4_vote_features_full']).to(device).requires_grad_(True)
        ep2 = {'seed_inds': ep['seed_inds'], 'seed_skeleton': ep['seed_skeleton'].detach(),
               'seed_features': ep['seed_features'].detach(), 'vote_xyz': vx, 'vote_features': vf}
        eps = {}
        g = torch.Generator().manual_seed(123)   # the CPU stream the reference drew its noise from
        for head, dt, D in (('center', torch.float32, 3), ('size', torch.float32, 3), ('heading', torch.float64, 2)):
            eps[head] = torch.empty(vx.shape[0] * 128, 100, 1, D, dtype=dt).normal_(generator=g).to(device)
        ep2, _ = net.detection(vx, vf, ep2, False, eps=eps)
        loss = net.loss(ep2, data)
        loss['total'].backward()
    assert np.array_equal(ep2['aggregated_vote_inds'].cpu().numpy(), z['g4_aggregated_vote_inds'])
    for k in ['aggregated_vote_xyz', 'center', 'size', 'heading', 'objectness_scores', 'sem_cls_scores']:
        got = ep2[k].detach().cpu().numpy()
        assert got.dtype == z[f'g4_{k}'].dtype, (k, got.dtype)
        np.testing.assert_allclose(got, z[f'g4_{k}'], err_msg=k, **TOL)
    for k, v in loss.items():
        assert str(v.dtype) == str(z[f'g4_lossdtype_{k}']), (k, v.dtype)
        np.testing.assert_allclose(v.item(), float(z[f'g4_loss_{k}']), rtol=1e-4, atol=1e-5, err_msg=k)
    for k, t in (('vote_xyz', vx), ('vote_features', vf)):
        ref_abs = float(z[f'g4_dvote_{k}_sum'][1])
        np.testing.assert_allclose(t.grad.flatten()[:256].cpu().numpy(), z[f'g4_dvote_{k}_head'], rtol=1e-3,
                                   atol=1e-3 * ref_abs / t.numel() + 1e-9, err_msg=k)
        assert abs(t.grad.double().abs().sum().item() - ref_abs) <= 1e-3 * ref_abs
    params = dict(net.named_parameters())
    names = sorted({k[len('g4_grad_'):-len('_head')] for k in z.files if k.startswith('g4_grad_') and k.endswith('_head')})
    det = [n for n in names if n.startswith('detection.')]
    assert len(det) == 6
    # gradients here reach 1e3 and some (a bias in front of a train-mode BatchNorm) are
    # analytically zero, i.e. pure cancellation noise: tolerances are relative to the
    # gradient scale of the head, not to each tensor's own magnitude.
    gscale = max(float(np.abs(z[f'g4_grad_{n}_head']).max()) for n in det)
    for name in det:
        gr = params[name].grad
        ref_abs = float(z[f'g4_grad_{name}_sum'][1])
        np.testing.assert_allclose(gr.flatten()[:64].cpu().numpy(), z[f'g4_grad_{name}_head'], rtol=2e-3,
                                   atol=5e-4 * gscale, err_msg=name)
        assert abs(gr.double().abs().sum().item() - ref_abs) <= 2e-3 * ref_abs + 5e-4 * gscale * gr.numel(), name
    opt.step()   # AdamW's first step is ~lr*sign(grad): only weights with a well-defined sign are compared
    for name in det:
        ref_g = z[f'g4_grad_{name}_head']
        solid = np.abs(ref_g) > 1e-3 * gscale
        got = params[name].detach().flatten()[:64].cpu().numpy()
        np.testing.assert_allclose(got[solid], z[f'g4_post_{name}_head'][solid], rtol=1e-4, atol=1e-5, err_msg=name)


GB = os.path.join(os.path.dirname(__file__), "golden", "g4b_backward.npz")


def check_packed(z, key, got, tol, floor=0.0, what=None):
    """Gradient tensor against its fixture record (strided values + (sum, abs-sum, max-abs)).
    Error bound: tol * max(max-abs of the reference tensor, floor) on every recorded element."""
    ref = z[key + '_val']
    stride = int(z[key + '_stride'])
    scale = max(float(z[key + '_sum'][2]), floor)
    g = got.detach().flatten()[::stride].double().cpu().numpy()
    assert g.shape == ref.shape, (key, g.shape, ref.shape)
    err = float(np.abs(g - ref).max()) if ref.size else 0.0
    assert err <= tol * scale, f"{what or key}: max err {err:.3e} > {tol:g} * scale {scale:.3e}"
    return err / scale if scale > 0 else 0.0


class GateForcer(object):
    """Pins the near-zero ReLU gates of the ST-GCN stack to the reference's recorded state (fixture arrays
    `<tag>_gate_{t,o}<block>_{idx,val,lim}`, tests/golden/make_model_golden.py: record_gates).

    The reference records, per gate layer, every pre-activation within GATE_EPS of zero (relative to the layer's largest).
    Our fp32 forward agrees with the reference's to ~1e-6 of a layer's scale, so only those gates can legitimately be in
    another state here -- and ONE gate in another state changes the gradients upstream of it by up to 1e-2 of their
    scale (measured: G4e, one gate of block 2 at |pre| = 1.5e-8, 7e-4 on five tensors).  Through bn_op.GATE_HOOK this
    object looks at our pre-activation at the recorded positions right before the fused kernels evaluate the gate; where
    the state (pre > 0) differs from the reference's it moves the BatchNorm's input so that the pre-activation becomes
    +-lim, the recorded band's half-width (a change below GATE_EPS of the layer's scale).  `forced` lists what was moved;
    the tests bound its length, so a real disagreement cannot hide behind it."""

    def __init__(self, net, z, tag):
        self.forced = []
        self.seen = set()
        self.by_bn = {}
        layers = [(f't{i}', blk.tcn[0]) for i, blk in enumerate(net.backbone.st_gcn_networks)]
        layers += [(f'o{i}', blk.tcn[3]) for i, blk in enumerate(net.backbone.st_gcn_networks)]
        # the two ReLUs of the vote head (fixtures that carry them)
        layers += [(f'v{i}', net.centervoting.conv_input[i].batchnorm) for i in (0, 1) if f'{tag}_gate_v{i}_idx' in z.files]
        for name, bn in layers:
            key = f'{tag}_gate_{name}'
            self.by_bn[id(bn)] = (key, torch.from_numpy(z[key + '_idx']), torch.from_numpy(z[key + '_val']),
                                  float(z[key + '_lim']))
        self.layers = len(layers)

    def __call__(self, bn, x, scale, shift, res):
        ent = self.by_bn.get(id(bn))
        if ent is None:
            return
        key, idx, val, lim = ent
        self.seen.add(key)
        if idx.numel() == 0:
            return
        with torch.no_grad():
            idx = idx.to(x.device)
            C, inner = x.shape[1], x[0, 0].numel()
            ch = (idx // inner) % C
            flat = x.detach().view(-1)
            sc, sh = scale.detach()[ch], shift.detach()[ch]
            r = res.detach().reshape(-1)[idx] if res is not None else torch.zeros((), device=x.device)
            pre = flat[idx] * sc + sh + r
            # (our pre-activation at the reference's near-zero positions is near zero too: the forward agrees)
            assert float(pre.abs().max()) <= 100 * lim, (key, float(pre.abs().max()), lim)
            want = (val.to(x.device) > 0)
            differ = (pre > 0) != want
            if bool(differ.any()):
                tgt = torch.where(want[differ], torch.full_like(pre[differ], lim), torch.full_like(pre[differ], -lim))
                flat[idx[differ]] = (tgt - sh[differ] - (r[differ] if res is not None else 0.0)) / sc[differ]
                again = flat[idx[differ]] * sc[differ] + sh[differ] + (r[differ] if res is not None else 0.0)
                assert bool(((again > 0) == want[differ]).all()), key
                self.forced += [(key, int(j), float(p)) for j, p in zip(idx[differ].tolist(), pre[differ].tolist())]

    def __enter__(self):
        from pose2room_amd.p2rnet import bn_op
        assert bn_op.GATE_HOOK is None
        bn_op.GATE_HOOK = self
        return self

    def __exit__(self, *exc):
        from pose2room_amd.p2rnet import bn_op
        bn_op.GATE_HOOK = None
        return False


def run_g4b(net, data, z, device, ops_ctx, tol):
    """Backbone + voting backward, train-mode BatchNorm: forward to the votes, then back-propagate the
    REFERENCE's recorded seam gradients (d total / d vote_xyz, d vote_features) through our backbone and
    compare ~60 parameter gradients with the reference's.  Feeding the recorded seam gradient keeps the
    check independent of discrete flips (ball-query membership) in the head."""
    net.train()
    net.zero_grad()
    with ops_ctx():
        xyz, feats, ep = net._votes(data)
        gx = torch.from_numpy(z['g4b_dvote_xyz_full']).to(device)
        gf = torch.from_numpy(z['g4b_dvote_features_full']).to(device)
        torch.autograd.backward([xyz, feats], [gx, gf])
    assert np.array_equal(ep['seed_inds'].cpu().numpy(), z['g4b_seed_inds'])
    params = dict(net.named_parameters())
    names = [str(n) for n in z['g4b_names']]
    assert len(names) >= 50 and sum(n.startswith('backbone.st_gcn_networks') for n in names) == 24
    # floor: gradients that vanish analytically (a conv bias in front of a train-mode BatchNorm) are cancellation
    # noise of the summands; they are compared on the scale of the largest backbone gradient
    floor = 1e-3 * max(float(z[f'g4b_grad_{n}_sum'][2]) for n in names)
    worst = {}
    for n in names:
        worst[n] = check_packed(z, f'g4b_grad_{n}', params[n].grad, tol, floor, n)
    return worst


def run_g4e(net, data, z, device, ops_ctx, tol=1e-4):
    """Every BatchNorm in eval mode (running statistics), autograd on: the whole step end to end against
    the reference at the north star's 1e-4 -- outputs, the 10 losses (with dtypes), and the gradients of
    all recorded parameters (fused ST-GCN forward AND backward without batch-statistics amplification)."""
    net.train()
    for m in net.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.eval()
    net.zero_grad()
    data = dict(data)
    data['center_label'] = torch.from_numpy(z['g4e_center_label']).to(device)
    eps = {}
    g = torch.Generator().manual_seed(123)
    B = data['input_joints'].shape[0]
    for head, dt, D in (('center', torch.float32, 3), ('size', torch.float32, 3), ('heading', torch.float64, 2)):
        eps[head] = torch.empty(B * 128, 100, 1, D, dtype=dt).normal_(generator=g).to(device)
    with ops_ctx():
        ep = net(data, eps=eps)
        ep['vote_xyz'].retain_grad()
        ep['vote_features'].retain_grad()
        loss = net.loss(ep, data)
        loss['total'].backward()
    check_endpoints(z, 'g4e', ep)
    for k, v in loss.items():
        assert str(v.dtype) == str(z[f'g4e_lossdtype_{k}']), (k, v.dtype)
        np.testing.assert_allclose(v.item(), float(z[f'g4e_loss_{k}']), rtol=1e-4, atol=1e-5, err_msg=k)
    for k in ('vote_xyz', 'vote_features'):
        check_packed(z, f'g4e_d{k}', ep[k].grad, tol)
    params = dict(net.named_parameters())
    names = [str(n) for n in z['g4e_names']]
    assert sum(n.startswith('detection.') for n in names) > 20
    floor = 1e-3 * max(float(z[f'g4e_grad_{n}_sum'][2]) for n in names)
    worst = {}
    for n in names:
        worst[n] = check_packed(z, f'g4e_grad_{n}', params[n].grad, tol, floor, n)
    return worst


def test_g4b_backbone_backward():
    """CPU twin (torch modules + oracle ops).  Train-mode BatchNorm amplifies rounding differences between two
    fp32 evaluation orders; 2e-3 of each tensor's largest gradient (measured worst: see the print)."""
    from oracle.cpu_backend import cpu_ops
    from pose2room_amd.p2rnet.synthetic import make_batch
    z = np.load(GB)
    net, cfg = build('train', 256)
    worst = run_g4b(net, make_batch(2, 256, seed=356), z, torch.device('cpu'), cpu_ops, tol=2e-3)
    print('g4b worst rel err', max(worst.values()), max(worst, key=worst.get))


def test_g4e_eval_bn_step():
    from oracle.cpu_backend import cpu_ops
    from pose2room_amd.p2rnet.synthetic import make_batch
    z = np.load(GB)
    net, cfg = build('train', 256)
    worst = run_g4e(net, make_batch(2, 256, seed=356), z, torch.device('cpu'), cpu_ops)
    print('g4e worst rel err', max(worst.values()), max(worst, key=worst.get))


def test_g4_train_step():
    from oracle.cpu_backend import cpu_ops
    from pose2room_amd.p2rnet.synthetic import make_batch
    z = np.load(G)
    net, cfg = build('train', 256)
    run_g4(net, cfg, make_batch(2, 256, seed=356), z, torch.device('cpu'), cpu_ops)


def test_g4_backbone_train_mode_statistics():
    """Train-mode forward of the whole net on CPU: BatchNorm running statistics after one
    forward match the reference's (robust to the downstream discrete flips)."""
    from oracle.cpu_backend import cpu_ops
    from pose2room_amd.p2rnet.synthetic import make_batch
    z = np.load(G)
    net, cfg = build('train', 256)
    net.train()
    with cpu_ops():
        net(make_batch(2, 256, seed=356))
    np.testing.assert_allclose(net.state_dict()['backbone.st_gcn_networks.2.tcn.0.running_mean'].numpy(),
                               z['g4_bn_running_mean'], rtol=1e-3, atol=1e-4)


def test_optimizer_groups_and_bn_momentum_schedule():
    """models/optimizers.py:22-94,121-148: one AdamW parameter group per sub-module with its own optim_spec (per-phase
    overrides from `model.<phase>.optimizer`), and the epoch-indexed BatchNorm momentum written into every BN layer."""
    from pose2room_amd.p2rnet import P2RConfig, default_config, METHODS
    from pose2room_amd.p2rnet.training import load_optimizer, BNMomentumScheduler
    conf = default_config('train', data={'num_frames': 32})
    conf['model']['detection']['optimizer'] = {'lr': 5e-4, 'weight_decay': 0.01}
    net = METHODS.get('P2RNet')(P2RConfig(conf, device='cpu'))
    opt = load_optimizer(conf, net)
    assert len(opt.param_groups) == 3
    assert sum(len(g['params']) for g in opt.param_groups) == len(list(net.parameters())) == 131
    lrs = sorted(g['lr'] for g in opt.param_groups)
    assert lrs == [5e-4, 1e-3, 1e-3]
    det = [g for g in opt.param_groups if g['lr'] == 5e-4][0]
    assert det['weight_decay'] == 0.01 and det['betas'] == (0.9, 0.999)
    assert {id(p) for p in det['params']} == {id(p) for p in net.detection.parameters()}
    sched = BNMomentumScheduler(None, net, bn_lambda=lambda e: max(0.5 * 0.5 ** (e // 2), 0.01))
    bns = [m for m in net.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
    assert bns and all(m.momentum == 0.5 for m in bns)
    for _ in range(4):          # the constructor applies epoch 0 and keeps last_epoch = -1, like the reference's
        sched.step()
    assert all(m.momentum == 0.25 for m in bns) and sched.last_epoch == 3


def test_graph_tables_other_skeletons_cpu():
    """GraphTables (and with it STGCN) can be built for any adjacency: the static work stream of the second-generation
    kernels exists only for 53-joint patterns that fit its record budget (advisor finding, round 2)."""
    import numpy as np
    from pose2room_amd.p2rnet import gcn_op, gcn_tables
    from pose2room_amd.p2rnet.modules.stgcn_layers import Graph
    rng = np.random.RandomState(0)
    t = gcn_op.GraphTables((rng.rand(3, 25, 25) > 0.8).astype(np.float32))
    assert not t.gen2 and t.V == 25 and t.K == 3
    dense = gcn_op.GraphTables(np.ones((11, 53, 53), dtype=np.float32))     # 53 joints, far over the record budget
    assert not dense.gen2
    assert gcn_op.GraphTables(Graph().A).gen2                                # the P2RNet skeleton keeps its stream
    with pytest.raises(gcn_tables.StreamBudgetError):
        gcn_tables.deal_runs([1] * 25, 8, 7)


def test_g10a_config1_forward_and_loss_cpu():
    """BASELINE configs[1] at its full size (bs=8, T=512) on the CPU: our host model with the oracle ops behind it against
    the vectors the imported reference produced at that size (G10, tests/golden/make_headline_golden.py) -- forward end
    points and the ten losses at the north star's 1e-4, train-mode BatchNorm.  (The GPU twin, and configs[2] at bs=32,
    T=1024, are in tests/test_headline_gpu.py.)"""
    from oracle.cpu_backend import cpu_ops
    from pose2room_amd.p2rnet.synthetic import make_batch
    from tests.test_headline_gpu import _check_forward_and_loss, _mixture_noise, G10
    z = np.load(G10)
    net, cfg = build('train', 512)
    net.train()
    assert abs(cases.state_checksum(net) - float(z['g10a_wsum'][0])) < 1e-6 * float(z['g10a_wsum'][0])
    batch = make_batch(8, 512, seed=612)
    with torch.no_grad(), cpu_ops():
        ep = net(dict(batch), eps=_mixture_noise(8, torch.device('cpu')))
        loss = net.loss(ep, batch)
    m = _check_forward_and_loss(z, 'g10a', ep, loss, 1e-4, 1e-4)
    print('g10a cpu vs reference:', {k: (f'{v:.2e}' if isinstance(v, float) else v) for k, v in m.items()})
