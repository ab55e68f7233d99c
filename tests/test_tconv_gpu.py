"""GPU: fused BatchNorm->ReLU->temporal conv op against the torch module chain."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


# T % 16 == 0 with 53 joints: the statically scheduled third generation (csrc/stgcn_tconv3.hip); (5, 1008): more tiles
# than persistent workgroups; everything else: second generation / first generation (other joint counts)
@pytest.mark.parametrize("N,T,V", [(2, 40, 53), (1, 1, 53), (3, 9, 53), (2, 130, 53), (2, 17, 25), (5, 1000, 53), (2, 64, 53), (1, 16, 53),
                                   (5, 1008, 53)])
@pytest.mark.parametrize("train", [True, False])
def test_bn_relu_tconv(dev, N, T, V, train):
    from pose2room_amd.p2rnet import tconv_op
    torch.manual_seed(N * 10 + T)
    bn_ref = torch.nn.BatchNorm2d(64).to(dev)
    conv_ref = torch.nn.Conv2d(64, 64, (3, 1), (1, 1), (1, 0)).to(dev)
    with torch.no_grad():
        bn_ref.weight.uniform_(0.5, 1.5); bn_ref.bias.uniform_(-0.5, 0.5)
        bn_ref.running_mean.uniform_(-0.2, 0.2); bn_ref.running_var.uniform_(0.5, 2.0)
    bn_new, conv_new = copy.deepcopy(bn_ref), copy.deepcopy(conv_ref)
    # fp64 reference: MIOpen's fp32 (3,1) convolution uses Winograd here and is itself only ~1e-4 accurate
    bn_ref, conv_ref = bn_ref.double(), conv_ref.double()
    bn_ref.train(train); bn_new.train(train)
    z = torch.randn(N, 64, T, V, device=dev) * 1.5 + 0.3
    go = torch.randn(N, 64, T, V, device=dev)

    zr = z.double().clone().requires_grad_(True)
    pre = bn_ref(zr)
    act = torch.relu(pre)
    act.retain_grad()
    ur = conv_ref(act)
    ur.backward(go.double())
    zn = z.clone().requires_grad_(True)
    assert tconv_op.supported(zn, bn_new, conv_new)
    un = tconv_op.bn_relu_tconv(zn, bn_new, conv_new)
    un.backward(go)

    def close(a, b, what, tol=3e-5, where=None):
        scale = b.abs().max().item() + 1e-12
        err = (a.double() - b.double()).abs()
        if where is not None:
            err = err * where
        err = err.max().item()
        assert err <= tol * scale, f"{what}: {err:.3e} vs {scale:.3e}"

    close(un, ur, "u")
    # the ReLU gate of an element whose pre-activation is within rounding of zero may open in fp32 and not in fp64
    # (one of 17M elements at the largest shape): such elements are left out of the data-gradient comparison
    decided = (pre.detach().abs() > 1e-5).double()
    assert decided.mean().item() > 0.9999
    close(zn.grad, zr.grad, "dz", 1e-4, where=decided)
    close(conv_new.weight.grad, conv_ref.weight.grad, "dW", 1e-4)
    close(conv_new.bias.grad, conv_ref.bias.grad, "dbias", 1e-4)
    # an undecided gate (see above) moves its whole gradient term in or out of the per-channel sums: each channel is
    # allowed the terms of its undecided elements on top of the tolerance
    und = 1.0 - decided
    xhat = (pre.detach() - bn_ref.bias.view(1, -1, 1, 1)) / bn_ref.weight.view(1, -1, 1, 1)
    for got, want, term, what in ((bn_new.bias.grad, bn_ref.bias.grad, act.grad.abs(), "dbeta"),
                                  (bn_new.weight.grad, bn_ref.weight.grad, (act.grad * xhat).abs(), "dgamma")):
        slack = (und * term).sum(dim=(0, 2, 3))            # also in eval mode (running statistics)
        err = (got.double() - want).abs()
        assert bool((err <= 1e-4 * want.abs().max() + slack).all()), f"{what}: {err.max().item():.3e} vs {want.abs().max().item():.3e}"
    if train:
        close(bn_new.running_var, bn_ref.running_var, "running_var", 1e-5)


@pytest.mark.parametrize("N,T", [(2, 40), (1, 1), (3, 9), (2, 130), (3, 64), (5, 1000)])
@pytest.mark.parametrize("train", [True, False])
def test_fused_bn_backward_apply_equals_the_separate_pass(dev, N, T, train):
    """p2r_stgcn_tconv_weight_grad_dz (BatchNorm-backward apply inside the weight-gradient launch) against
    p2r_bn_bwd_apply + p2r_stgcn_tconv_weight_grad: same arithmetic per element, fp32 contraction aside."""
    from pose2room_amd.p2rnet import tconv_op
    torch.manual_seed(N * 7 + T)
    bn = torch.nn.BatchNorm2d(64).to(dev)
    conv = torch.nn.Conv2d(64, 64, (3, 1), (1, 1), (1, 0)).to(dev)
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
        bn.running_mean.uniform_(-0.2, 0.2); bn.running_var.uniform_(0.5, 2.0)
    bn.train(train)
    z = torch.randn(N, 64, T, 53, device=dev) * 1.5 + 0.3
    go = torch.randn(N, 64, T, 53, device=dev)
    res = {}
    for fused in (True, False):
        tconv_op.FUSE_DZ = fused
        try:
            b, c = copy.deepcopy(bn), copy.deepcopy(conv)
            zn = z.clone().requires_grad_(True)
            tconv_op.bn_relu_tconv(zn, b, c).backward(go)
            res[fused] = (zn.grad, c.weight.grad, c.bias.grad, b.weight.grad, b.bias.grad)
        finally:
            tconv_op.FUSE_DZ = True
    for a, b_, what in zip(res[True], res[False], ("dz", "dW", "dbias", "dgamma", "dbeta")):
        assert torch.isfinite(a).all()
        err = (a - b_).abs().max().item()
        assert err <= 2e-6 * (b_.abs().max().item() + 1e-12), f"{what}: {err:.3e}"
    assert torch.equal(res[True][1], res[False][1]) and torch.equal(res[True][2], res[False][2])   # the product itself is the same launch


@pytest.mark.parametrize("N,T,V", [(2, 40, 53), (1, 1, 20), (3, 9, 20), (2, 130, 53), (2, 17, 64), (2, 48, 53), (3, 160, 53)])
@pytest.mark.parametrize("train", [True, False])
def test_bn_relu_pointwise(dev, N, T, V, train):
    """Single-tap form: BatchNorm1d -> ReLU -> Conv1d(64, 64, 1) of the embedding MLPs (stgcn.py:46-63)."""
    from pose2room_amd.p2rnet import tconv_op
    torch.manual_seed(N * 10 + T)
    bn_ref = torch.nn.BatchNorm1d(64).to(dev)
    conv_ref = torch.nn.Conv1d(64, 64, 1).to(dev)
    with torch.no_grad():
        bn_ref.weight.uniform_(0.5, 1.5); bn_ref.bias.uniform_(-0.5, 0.5)
        bn_ref.running_mean.uniform_(-0.2, 0.2); bn_ref.running_var.uniform_(0.5, 2.0)
    bn_new, conv_new = copy.deepcopy(bn_ref), copy.deepcopy(conv_ref)
    bn_ref, conv_ref = bn_ref.double(), conv_ref.double()
    bn_ref.train(train); bn_new.train(train)
    z = torch.randn(N, 64, T * V, device=dev) * 1.5 + 0.3
    go = torch.randn(N, 64, T * V, device=dev)

    zr = z.double().clone().requires_grad_(True)
    ur = conv_ref(torch.relu(bn_ref(zr)))
    ur.backward(go.double())
    zn = z.clone().requires_grad_(True)
    z4 = zn.view(N, 64, T, V)
    assert tconv_op.supported_pointwise(z4, bn_new, conv_new)
    un = tconv_op.bn_relu_tconv(z4, bn_new, conv_new).view(N, 64, T * V)
    un.backward(go)

    def close(a, b, what, tol=3e-5):
        scale = b.abs().max().item() + 1e-12
        err = (a.double() - b.double()).abs().max().item()
        assert err <= tol * scale, f"{what}: {err:.3e} vs {scale:.3e}"

    close(un, ur, "u")
    close(zn.grad, zr.grad, "dz", 1e-4)
    assert conv_new.weight.grad.shape == conv_new.weight.shape
    close(conv_new.weight.grad, conv_ref.weight.grad, "dW", 1e-4)
    close(conv_new.bias.grad, conv_ref.bias.grad, "dbias", 1e-4)
    close(bn_new.weight.grad, bn_ref.weight.grad, "dgamma", 1e-4)      # also in eval mode (running statistics)
    close(bn_new.bias.grad, bn_ref.bias.grad, "dbeta", 1e-4)
    if train:
        close(bn_new.running_var, bn_ref.running_var, "running_var", 1e-5)


@pytest.mark.parametrize("N,T,V", [(2, 48, 53), (3, 160, 53), (2, 40, 53), (2, 17, 20)])
@pytest.mark.parametrize("train", [True, False])
def test_bn_relu_pointwise_with_broadcast_addend(dev, N, T, V, train):
    """The last layer of the joint embedding with the position embedding folded in (stgcn.py:126-130):
    conv(relu(bn(z))) + pe[..., None] in one kernel when the statically scheduled path applies (53 joints, T % 16 == 0),
    composed otherwise; gradients of everything incl. the addend (row sums over the joints)."""
    from pose2room_amd.p2rnet import tconv_op
    torch.manual_seed(N * 7 + T)
    bn_ref = torch.nn.BatchNorm1d(64).to(dev)
    conv_ref = torch.nn.Conv1d(64, 64, 1).to(dev)
    with torch.no_grad():
        bn_ref.weight.uniform_(0.5, 1.5); bn_ref.bias.uniform_(-0.5, 0.5)
        bn_ref.running_mean.uniform_(-0.2, 0.2); bn_ref.running_var.uniform_(0.5, 2.0)
    bn_new, conv_new = copy.deepcopy(bn_ref), copy.deepcopy(conv_ref)
    bn_ref, conv_ref = bn_ref.double(), conv_ref.double()
    bn_ref.train(train); bn_new.train(train)
    z = torch.randn(N, 64, T * V, device=dev) * 1.5 + 0.3
    pe = torch.randn(N, 64, T, device=dev)
    go = torch.randn(N, 64, T, V, device=dev)
    zr, per = z.double().clone().requires_grad_(True), pe.double().clone().requires_grad_(True)
    ur = conv_ref(torch.relu(bn_ref(zr))).view(N, 64, T, V) + per.unsqueeze(-1)
    ur.backward(go.double())
    zn, pen = z.clone().requires_grad_(True), pe.clone().requires_grad_(True)
    un = tconv_op.bn_relu_tconv(zn.view(N, 64, T, V), bn_new, conv_new, add_ct=pen)
    un.backward(go)

    def close(a, b, what, tol=3e-5):
        scale = b.abs().max().item() + 1e-12
        err = (a.double() - b.double()).abs().max().item()
        assert err <= tol * scale, f"{what}: {err:.3e} vs {scale:.3e}"

    close(un, ur, "u")
    close(zn.grad, zr.grad, "dz", 1e-4)
    close(pen.grad, per.grad, "dpe", 1e-5)
    close(conv_new.weight.grad, conv_ref.weight.grad, "dW", 1e-4)
    close(conv_new.bias.grad, conv_ref.bias.grad, "dbias", 1e-4)
    close(bn_new.weight.grad, bn_ref.weight.grad, "dgamma", 1e-4)


@pytest.mark.parametrize("B,L", [(2, 53 * 40), (1, 7), (3, 20 * 33 + 1), (2, 4096)])
@pytest.mark.parametrize("bias", [True, False])
def test_embed3(dev, B, L, bias):
    """Conv1d(3 -> 64, 1): forward and weight / bias gradients against torch in fp64."""
    from pose2room_amd.p2rnet import tconv_op
    torch.manual_seed(B + L)
    conv = torch.nn.Conv1d(3, 64, 1, bias=bias).to(dev)
    ref = copy.deepcopy(conv).double()
    x = torch.randn(B, 3, L, device=dev)
    go = torch.randn(B, 64, L, device=dev)
    assert tconv_op.supported_embed3(x, conv)
    out = tconv_op.embed3(x, conv)
    out.backward(go)
    want = ref(x.double())
    want.backward(go.double())
    assert out.shape == (B, 64, L)
    torch.testing.assert_close(out.double(), want, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(conv.weight.grad.double(), ref.weight.grad, rtol=1e-4, atol=1e-4 * ref.weight.grad.abs().max().item())
    if bias:
        torch.testing.assert_close(conv.bias.grad.double(), ref.bias.grad, rtol=1e-4, atol=1e-4 * ref.bias.grad.abs().max().item())


@pytest.mark.parametrize("N,T,V,taps", [(2, 40, 53, 3), (3, 9, 20, 1), (2, 130, 53, 3), (1, 300, 53, 1), (3, 48, 53, 3),
                                        (20, 256, 53, 3), (2, 64, 53, 1)])
def test_kernel_emitted_statistics(dev, N, T, V, taps):
    """The partial statistics written by the conv epilogue equal the statistics of its output ((count, mean, M2) entries
    from the third generation, (sum, sum of squares) pairs from the others; 20 x 256 frames: 320 tiles over 256
    workgroups, unequal counts)."""
    from pose2room_amd.p2rnet import tconv_op, bn_op
    torch.manual_seed(T)
    x = torch.randn(N, 64, T, V, device=dev)
    W = torch.randn(taps, 64, 64, device=dev) / 8
    b = torch.randn(64, device=dev)
    sc, sh = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
    out, part = tconv_op._tconv(x, sc, sh, W, b, want_stats=True)
    assert part.dim() == 3 and part.shape[1:] == (64, 3 if (T % 16 == 0 and V == 53) else 2)
    assert torch.equal(out, tconv_op._tconv(x, sc, sh, W, b))          # same output with and without the statistics
    mean, var, M = bn_op.moments(part, N * T * V)
    o = out.double()
    torch.testing.assert_close(mean, o.mean(dim=(0, 2, 3)), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(var, o.var(dim=(0, 2, 3), unbiased=False), rtol=1e-4, atol=1e-6)


def test_statistics_survive_a_large_mean(dev):
    """|mean| >> std: the conv bias puts every output channel 1e3+ standard deviations from zero; the third-generation
    epilogue's pivoted (count, mean, M2) entries keep the variance, and the BatchNorm that consumes them (finalize)
    normalises to unit variance."""
    from pose2room_amd.p2rnet import tconv_op, bn_op
    torch.manual_seed(3)
    N, T, V = 4, 64, 53
    x = torch.randn(N, 64, T, V, device=dev)
    W = torch.randn(3, 64, 64, device=dev) / 64
    b = torch.linspace(100.0, 1000.0, 64, device=dev)
    out, part = tconv_op._tconv(x, None, None, W, b, want_stats=True)
    assert part.shape[-1] == 3
    o = out.double()
    ref_mean, ref_var = o.mean(dim=(0, 2, 3)), o.var(dim=(0, 2, 3), unbiased=False)
    assert float((ref_mean.abs() / ref_var.sqrt()).min()) > 100.0
    mean, var, _ = bn_op.moments(part, N * T * V)
    torch.testing.assert_close(mean, ref_mean, rtol=1e-6, atol=0)
    torch.testing.assert_close(var, ref_var, rtol=1e-4, atol=0)
    bn = torch.nn.BatchNorm2d(64).to(dev).train()
    fin = bn_op.finalize(part, N * T * V, bn)
    torch.testing.assert_close(fin[0].double(), ref_mean, rtol=1e-6, atol=0)
    torch.testing.assert_close(fin[1].double(), 1.0 / (ref_var + bn.eps).sqrt(), rtol=1e-4, atol=0)


@pytest.mark.parametrize("N,T,taps", [(1, 16, 3), (3, 64, 3), (9, 480, 3), (2, 32, 1), (5, 1008, 1)])
def test_tconv3_equals_tconv2(dev, N, T, taps):
    """The statically scheduled kernel (csrc/stgcn_tconv3.hip) and the second generation run the same MFMA chain per
    output element: bit-identical outputs and epilogue sums -- forward with the input transform and statistics, data
    gradient, data gradient with the BatchNorm-backward epilogue; three taps and the single-tap form."""
    from pose2room_amd.p2rnet import tconv_op
    V = 53
    g = torch.Generator().manual_seed(N * 13 + T + taps)
    x = torch.randn(N, 64, T, V, generator=g).to(dev)
    z = torch.randn(N, 64, T, V, generator=g).to(dev)
    scale, shift = (torch.rand(64, generator=g) + 0.5).to(dev), torch.randn(64, generator=g).to(dev)
    bias = torch.randn(64, generator=g).to(dev)
    fin = torch.randn(4, 64, generator=g).to(dev)
    W3 = (torch.randn(taps, 64, 64, generator=g) / 8).to(dev)
    cases = [dict(scale=scale, shift=shift, bias=bias, want_stats=True), dict(scale=None, shift=None, bias=None),
             dict(scale=None, shift=None, bias=None, want_stats=True, bwd=(z, fin))]
    try:
        for kw in cases:
            outs = []
            for gen3 in (False, True):
                tconv_op.USE_GEN3 = gen3
                outs.append(tconv_op._tconv(x, kw['scale'], kw['shift'], W3, kw['bias'], kw.get('want_stats', False), kw.get('bwd')))
            torch.cuda.synchronize()
            a, b = outs
            if isinstance(a, tuple):
                assert torch.equal(a[0], b[0])
                if a[1].shape == b[1].shape:        # BatchNorm-backward sums: the same pairs up to the summation order
                    sa, sb = a[1].double().sum(0), b[1].double().sum(0)
                    torch.testing.assert_close(sb, sa, rtol=2e-5, atol=2e-5 * float(sa.abs().max()))
                else:                               # forward statistics: (sum, sum sq) against (count, mean, M2)
                    from pose2room_amd.p2rnet import bn_op
                    (m2, v2, _), (m3, v3, _) = bn_op.moments(a[1], N * T * V), bn_op.moments(b[1], N * T * V)
                    torch.testing.assert_close(m3, m2, rtol=1e-6, atol=1e-6)
                    torch.testing.assert_close(v3, v2, rtol=1e-5, atol=1e-7)
            else:
                assert torch.equal(a, b)
    finally:
        tconv_op.USE_GEN3 = True


@pytest.mark.parametrize("B,rows,inner,add", [(2, 64, 53, True), (3, 16, 20, False), (1, 128, 53, False), (2, 48, 20, True),
                                              (2, 64, 64, True)])
@pytest.mark.parametrize("train", [True, False])
def test_embed_mlp_one_pass_backward(dev, B, rows, inner, add, train):
    """embed_op.embed_mlp (forward on the tconv_op kernels, backward one fused pass per layer: csrc/embed_bwd.hip) against
    (1) the module chain in fp64 for the output and the running statistics, and (2) the layer-by-layer functions of tconv_op
    (data gradient, BatchNorm-backward apply, weight gradient as separate passes -- themselves pinned to fp64 chains by the
    tests above) for every gradient.  Both fp32 paths run the same forward kernels, so they agree on every ReLU mask bit; an
    fp64 chain does not: a pre-activation within fp32 rounding of zero (|y| = 6e-8 was seen at one of 434,176 positions)
    flips one bit and moves a per-channel gradient sum by a whole element (1 % of a BatchNorm bias gradient)."""
    from pose2room_amd.p2rnet import embed_op
    from pose2room_amd.p2rnet.modules.stgcn import _point_mlp, STGCN
    L = rows * inner
    torch.manual_seed(B * 100 + rows + inner)
    seq = _point_mlp(3, 64, 64).to(dev)
    with torch.no_grad():
        for m in seq.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.5, 0.5)
                m.running_mean.uniform_(-0.2, 0.2); m.running_var.uniform_(0.5, 2.0)
    ref64, lay = copy.deepcopy(seq).double(), copy.deepcopy(seq)
    seq.train(train); ref64.train(train); lay.train(train)
    x = torch.randn(B, 3, L, device=dev) * torch.tensor([1.0, 0.4, 2.0], device=dev)[None, :, None]
    pe = torch.randn(B, 64, rows, device=dev).requires_grad_(True) if add else None
    pel = pe.detach().clone().requires_grad_(True) if add else None
    go = torch.randn(B, 64, L, device=dev)
    assert embed_op.supported(seq, x, inner, pe)
    out = embed_op.embed_mlp(seq, x, inner, pe)
    out.backward(go)
    embed_op.USE_FUSED = False
    try:
        out_l = STGCN._mlp(lay, x, inner, add_ct=pel)
        out_l.backward(go)
    finally:
        embed_op.USE_FUSED = True
    with torch.no_grad():
        o64 = ref64(x.double())
        if add:
            o64 = (o64.view(B, 64, rows, inner) + pe.detach().double().unsqueeze(-1)).view(B, 64, L)

    def close(a, b, what, tol):
        scale = b.abs().max().item() + 1e-12
        err = (a.double() - b.double()).abs().max().item()
        assert err <= tol * scale, f"{what}: {err:.3e} vs {scale:.3e}"

    close(out, o64, "out vs fp64", 5e-5)
    if inner == 53:     # statically scheduled kernels: deterministic, the two paths agree bit for bit
        assert torch.equal(out, out_l), "same forward kernels: same bits"
    else:               # first-generation kernels merge their statistics with LDS float atomics: last-bit differences
        close(out, out_l, "out vs layered", 2e-6)
    if add:
        close(pe.grad, pel.grad, "d_add", 1e-6)
    for (n, p), (_, q) in zip(seq.named_parameters(), lay.named_parameters()):
        assert p.grad is not None and q.grad is not None, n
        close(p.grad, q.grad, n, 1e-4)
    for (n, a_), (_, b_) in zip(seq.named_buffers(), lay.named_buffers()):
        close(a_.float(), b_.float(), n, 0.0 if inner == 53 else 2e-6)
    if train:
        for (n, a_), (_, b_) in zip(seq.named_buffers(), ref64.named_buffers()):
            close(a_.float(), b_.float(), n + " vs fp64", 2e-5)
