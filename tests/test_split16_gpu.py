"""GPU: the opt-in split16 arithmetic of the ST-GCN kernels (csrc/split16.h, math_mode) against float64 and against the
exact-fp32 kernels: every split kernel must be at most 1.5x as far from float64 as the exact kernel is on the same inputs
(plus a floor of 2e-7 of range, the rounding of the stored fp32 result itself), over the range cases that fp16 operands
make interesting: gradients of magnitude 1e-6 and 1e+3, tiles whose fp16 residuals are all subnormal, exact fp16 ties."""
import copy
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

V = 53
FLOOR = 2e-7


def _rel(a, ref):
    return (a.double() - ref).abs().max().item() / (ref.abs().max().item() + 1e-300)


def _conv64(h, W3, bias=None):
    """float64 (3,1) convolution: h (N,64,T,V), W3 [tap][co][ci]"""
    w = W3.double().permute(1, 2, 0).unsqueeze(-1)
    return torch.nn.functional.conv2d(h.double(), w, bias.double() if bias is not None else None, padding=(1, 0))


@pytest.mark.parametrize("n", [0, 1, 3, 4, 1000, 4099, 1 << 20])
def test_absmax_bits(dev, n):
    from pose2room_amd import _lib
    from pose2room_amd.p2rnet import math_mode
    g = torch.Generator().manual_seed(n)
    buf = (torch.randn(n + 1, generator=g) * 3).to(dev)
    for x in (buf[:n], buf[1:]):                      # 16-byte aligned and not
        before = math_mode.FALLBACK_PASSES
        word = math_mode.range_word(x)
        assert math_mode.FALLBACK_PASSES == before + 1
        want = x.abs().max() if x.numel() else torch.zeros((), device=dev)
        assert word.view(torch.float32).item() == want.item()
    if n:
        x = buf[:n].clone(); x[n // 2] = float('nan')
        bits = math_mode.range_word(x).item() & 0x7fffffff
        assert (bits >> 23) == 255 and (bits & 0x7fffff) != 0          # a NaN pattern: "no scale"


def test_range_word_announced_only_for_the_very_tensor(dev):
    from pose2room_amd.p2rnet import math_mode
    x = torch.randn(1000, device=dev)
    w = math_mode.new_word(dev); w.fill_(123)
    math_mode.announce(x, w)
    assert math_mode.range_word(x, keep=True) is w
    assert math_mode.range_word(x) is w
    assert math_mode.range_word(x) is not w                              # consumed
    math_mode.announce(x, w)
    x.add_(1.0)                                                          # another version of the buffer
    assert math_mode.range_word(x) is not w
    math_mode.reset()


def _tconv_inputs(N, T, seed, gmag=None):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, 64, T, V, generator=g)
    if gmag is not None:
        x = x * gmag
    W3 = torch.randn(3, 64, 64, generator=g) / 8
    scale, shift = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    bias = torch.randn(64, generator=g)
    return x, W3, scale, shift, bias


@pytest.mark.parametrize("N,T", [(2, 16), (3, 64), (1, 32), (2, 48), (1, 80), (4, 256), (3, 1024)])
def test_tconvh_forward_vs_float64_and_exact(dev, N, T):
    """forward: BatchNorm affine + ReLU on the input, bias, (count, mean, M2) statistics of the result"""
    from pose2room_amd.p2rnet import bn_op, tconv_op
    x, W3, scale, shift, bias = (t.to(dev) for t in _tconv_inputs(N, T, 100 + T))
    ref = _conv64(torch.relu(x.double() * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)), W3, bias)
    exact, epart = tconv_op._tconv(x, scale, shift, W3, bias, want_stats=True)
    st = tconv_op.SplitTaps(*tconv_op.split_taps(W3))
    got, part = tconv_op._tconvh(x, scale, shift, st, bias, want_stats=True)
    e_split, e_exact = _rel(got, ref), _rel(exact, ref)
    assert e_split <= 1.5 * e_exact + FLOOR, (e_split, e_exact)
    assert _rel(got, exact.double()) <= 4e-6
    # statistics of the STORED values, as the exact kernel's epilogue gives them
    M = N * T * V
    mean, var, _ = bn_op.moments(part, M)
    want_mean, want_var = got.double().mean(dim=(0, 2, 3)), got.double().var(dim=(0, 2, 3), unbiased=False)
    assert (mean - want_mean).abs().max().item() <= 1e-6 * got.abs().max().item()
    assert ((var - want_var).abs() / want_var).max().item() <= 1e-5
    assert part.shape[0] == N * (T // (64 if T % 64 == 0 else 32 if T % 32 == 0 else 16))
    assert float(part[..., 0].sum(0)[0]) == M


@pytest.mark.parametrize("gmag", [1e-6, 1.0, 1e3])
@pytest.mark.parametrize("N,T", [(2, 16), (3, 64), (2, 1024)])
def test_tconvh_data_gradient_range(dev, N, T, gmag):
    """plain form (the data gradient's) with the sums of the BatchNorm + ReLU backward, on gradients of magnitude
    1e-6 .. 1e+3: the range word puts each of them at the same place in fp16's range"""
    from pose2room_amd.p2rnet import math_mode, tconv_op
    du, W3, scale, shift, _ = (t.to(dev) for t in _tconv_inputs(N, T, 7 + T, gmag))
    z = torch.randn(N, 64, T, V, device=dev)
    mean, invstd = torch.randn(64, device=dev) * 0.1, torch.rand(64, device=dev) + 0.5
    fin = torch.stack([mean, invstd, scale, shift]).contiguous()
    ref = _conv64(du, W3)
    exact, epart = tconv_op._tconv(du, None, None, W3, None, want_stats=True, bwd=(z, fin))
    st = tconv_op.SplitTaps(*tconv_op.split_taps(W3))
    got, part = tconv_op._tconvh(du, None, None, st, None, want_stats=True, bwd=(z, fin), x_word=math_mode.range_word(du))
    e_split, e_exact = _rel(got, ref), _rel(exact, ref)
    assert e_split <= 1.5 * e_exact + FLOOR, (gmag, e_split, e_exact)
    # the two sums, from the stored result
    gate = (z.double() * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)) > 0
    gq = got.double() * gate
    s1 = gq.sum(dim=(0, 2, 3))
    s2 = (gq * (z.double() - mean.double().view(1, -1, 1, 1)) * invstd.double().view(1, -1, 1, 1)).sum(dim=(0, 2, 3))
    tot = part.double().sum(0)
    scale_ = gq.abs().sum(dim=(0, 2, 3)).max().item()
    assert (tot[:, 0] - s1).abs().max().item() <= 1e-5 * scale_
    assert (tot[:, 1] - s2).abs().max().item() <= 3e-5 * scale_
    # without a range word a 1e-6 gradient sits in fp16's subnormals (a handful of bits in the leading part; the scaled
    # residual recovers eleven more): the word is what makes the split work
    if gmag == 1e-6:
        raw = tconv_op._tconvh(du, None, None, st, None)
        assert _rel(raw, ref) > 10 * e_split


def test_tconvh_subnormal_residuals_and_fp16_ties(dev):
    """(a) one large element fixes the scale, everything else is so small that its fp16 residual is subnormal: the error
    stays an ABSOLUTE 2^-25 of the range (fixed-point behaviour), below the exact kernel's.  (b) operands at exact fp16
    ties (x = k + 2^-11 patterns): p + q must reproduce x -- with identity taps the result equals the input bit for bit."""
    from pose2room_amd.p2rnet import math_mode, tconv_op
    N, T = 2, 64
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(N, 64, T, V, generator=g) * 1e-5).to(dev)
    x[0, 0, 0, 0] = 4.0
    W3 = (torch.randn(3, 64, 64, generator=g) / 8).to(dev)
    st = tconv_op.SplitTaps(*tconv_op.split_taps(W3))
    ref = _conv64(x, W3)
    got = tconv_op._tconvh(x, None, None, st, None, x_word=math_mode.range_word(x))
    exact = tconv_op._tconv(x, None, None, W3, None)
    assert _rel(got, ref) <= 1.5 * _rel(exact, ref) + FLOOR
    small = torch.ones_like(ref, dtype=torch.bool); small[0, :, 0:2, 0] = False     # outputs the large element does not reach
    assert (got.double() - ref)[small].abs().max().item() <= 1e-9                   # ~1e-5-sized outputs: absolute, not 2^-11 relative
    # (b)
    k = torch.randint(1024, 2048, (N, 64, T, V), generator=g).float()
    x = ((k + 0.5) / 1024).to(dev)               # in [1, 2): fp16 spacing 2^-10 -- every value halfway between two fp16 numbers
    ident = torch.zeros(3, 64, 64); ident[1] = torch.eye(64)
    st = tconv_op.SplitTaps(*tconv_op.split_taps(ident.to(dev)))
    got = tconv_op._tconvh(x, None, None, st, None, x_word=math_mode.range_word(x))
    assert torch.equal(got, x)


def _tail_profile(N, T, seed):
    """per-frame magnitudes 2^0, 2^-5, ... 2^-20 in runs of 8 frames: what a real gradient looks like (measured on the
    P2RNet backward, tools/dev_grad_stats.py: the median element is 2^-11 of the largest, 1 % of them below 2^-19), and
    what ONE scale per tensor has to cope with"""
    g = torch.Generator().manual_seed(seed)
    k = (torch.arange(T) // 8) % 5 * 5
    return torch.randn(N, 64, T, V, generator=g) * (2.0 ** -k.float()).view(1, 1, T, 1), k


def _per_run_error(got, ref, k, what):
    """worst relative error (of the run's own largest value) over the interior frames of each run of equal magnitude"""
    T = ref.shape[2]
    worst = {}
    for t0 in range(0, T, 8):
        sl = slice(t0 + 2, t0 + 6)
        e = (got[:, :, sl].double() - ref[:, :, sl]).abs().max().item() / ref[:, :, sl].abs().max().item()
        worst[int(k[t0])] = max(worst.get(int(k[t0]), 0.0), e)
    return worst


@pytest.mark.parametrize("N,T", [(2, 80), (2, 1024)])
def test_tconvh_heavy_tailed_gradient(dev, N, T):
    """a gradient whose frames differ by up to 2^20 in magnitude: every run of frames keeps the exact kernel's RELATIVE
    accuracy (the residual part is kept scaled by 2^11: without that the 2^-20 runs lose 2^-16 of their own size)"""
    from pose2room_amd.p2rnet import math_mode, tconv_op
    du, k = _tail_profile(N, T, 21)
    du = du.to(dev)
    W3 = (torch.randn(3, 64, 64, generator=torch.Generator().manual_seed(4)) / 8).to(dev)
    st = tconv_op.SplitTaps(*tconv_op.split_taps(W3))
    ref = _conv64(du, W3)
    got = tconv_op._tconvh(du, None, None, st, None, x_word=math_mode.range_word(du))
    exact = tconv_op._tconv(du, None, None, W3, None)
    ws, we = _per_run_error(got, ref, k, 'split16'), _per_run_error(exact, ref, k, 'exact')
    for kk in sorted(ws):
        assert ws[kk] <= 1.5 * we[kk] + FLOOR, (kk, ws, we)


@pytest.mark.parametrize("N,T", [(2, 64), (3, 48), (2, 1024)])
def test_bn_relu_tconv_module_in_split16_mode(dev, N, T):
    """the op the model calls, forward + backward under `math_mode.use('split16')`, against the float64 module chain and
    against the exact mode: every gradient within 1.5x the exact mode's distance from float64"""
    from pose2room_amd.p2rnet import math_mode, tconv_op
    torch.manual_seed(N * 10 + T)
    bn_ref = torch.nn.BatchNorm2d(64).to(dev)
    conv_ref = torch.nn.Conv2d(64, 64, (3, 1), (1, 1), (1, 0)).to(dev)
    with torch.no_grad():
        bn_ref.weight.uniform_(0.5, 1.5); bn_ref.bias.uniform_(-0.5, 0.5)
    z = torch.randn(N, 64, T, V, device=dev) * 1.5 + 0.3
    go = torch.randn(N, 64, T, V, device=dev) * 1e-4
    res = {}
    for m in ('exact', 'split16'):
        bn, conv = copy.deepcopy(bn_ref).train(), copy.deepcopy(conv_ref)
        zn = z.clone().requires_grad_(True)
        with math_mode.use(m):
            u = tconv_op.bn_relu_tconv(zn, bn, conv)
            u.backward(go)
        res[m] = (u.detach(), zn.grad, conv.weight.grad, conv.bias.grad, bn.weight.grad, bn.bias.grad)
    bn, conv = copy.deepcopy(bn_ref).double().train(), copy.deepcopy(conv_ref).double()
    zr = z.double().clone().requires_grad_(True)
    pre = bn(zr)
    ur = conv(torch.relu(pre))
    ur.backward(go.double())
    want = (ur.detach(), zr.grad, conv.weight.grad, conv.bias.grad, bn.weight.grad, bn.bias.grad)
    decided = (pre.detach().abs() > 1e-5)
    for i, what in enumerate(("u", "dz", "dW", "dbias", "dgamma", "dbeta")):
        a, b, r = res['split16'][i].double(), res['exact'][i].double(), want[i]
        if what == "dz":
            a, b, r = a * decided, b * decided, r * decided
        es, ee = _rel(a, r), _rel(b, r)
        assert es <= 1.5 * ee + 1e-6, (what, es, ee)
    math_mode.reset()


# ---- graph conv ---------------------------------------------------------------------------------------------------------
def _gcn_reference(x, weight, bias, Aeff):
    K = Aeff.shape[0]
    y = torch.nn.functional.conv2d(x, weight.view(K * 64, 64, 1, 1), bias)
    n, kc, t, v = y.shape
    return torch.einsum('nkctv,kvw->nctw', y.view(n, K, kc // K, t, v), Aeff)


def _gcn_case(N, T, seed, gmag=1.0):
    from pose2room_amd.p2rnet.modules.stgcn_layers import Graph
    A = Graph().A
    K = A.shape[0]
    g = torch.Generator().manual_seed(seed)
    x = torch.relu(torch.randn(N, 64, T, V, generator=g) + 0.3)
    w = torch.randn(K * 64, 64, generator=g) / 8
    b = torch.randn(K * 64, generator=g) * 0.1
    imp = 1 + 0.1 * torch.randn(K, V, V, generator=g)
    go = torch.randn(N, 64, T, V, generator=g) * gmag
    return A, x, w, b, imp, go


@pytest.mark.parametrize("gmag", [1e-6, 1.0, 1e3])
@pytest.mark.parametrize("N,T", [(1, 16), (3, 48), (2, 256), (5, 1008)])
def test_graph_conv_split16_vs_float64_and_exact(dev, N, T, gmag):
    """graph_conv forward + backward in both modes against the float64 formulation of the reference op
    (stgcn_layers.py:57-67): z and dx (the split kernels) within 1.5x the exact kernels' distance from float64 over
    gradient magnitudes 1e-6 .. 1e+3; the other gradients (exact kernels in both modes so far) unchanged."""
    from pose2room_amd.p2rnet import gcn_op, math_mode
    A, x, w, b, imp, go = _gcn_case(N, T, N * 100 + T, gmag)
    tables = gcn_op.GraphTables(A)
    assert tables.gen3h and len(tables.pairs_c) == 6 and len(tables.pairs_r) == 6
    At = torch.tensor(A, dtype=torch.float32)
    xr, wr, br, ir = (t.double().to(dev).requires_grad_(True) for t in (x, w, b, imp))
    zr = _gcn_reference(xr, wr, br, At.double().to(dev) * ir)
    zr.backward(go.double().to(dev))
    want = (zr.detach(), xr.grad, wr.grad, br.grad, ir.grad)
    res = {}
    for m in ('exact', 'split16'):
        xd, wd, bd, idv = (t.to(dev).requires_grad_(True) for t in (x, w, b, imp))
        with math_mode.use(m):
            z = gcn_op.graph_conv(xd, wd, bd, At.to(dev) * idv, tables)
            z.backward(go.to(dev))
        res[m] = (z.detach(), xd.grad, wd.grad, bd.grad, idv.grad)
    for i, what in enumerate(("z", "dx", "dW", "db", "d importance")):
        es, ee = _rel(res['split16'][i], want[i]), _rel(res['exact'][i], want[i])
        assert es <= 1.5 * ee + FLOOR, (what, gmag, es, ee)
    assert not torch.equal(res['split16'][0], res['exact'][0])          # the split kernels did run
    math_mode.reset()


def test_graph_conv_split16_masked_addend_and_statistics(dev):
    """the data gradient's masked addend (bit-identical to plain + where(mask, addend, 0)) and the forward's
    (count, mean, M2) statistics (the moments of the stored tensor), at a shape with more tiles than workgroups"""
    from pose2room_amd.p2rnet import bn_op, gcn_op, gcn_tables, math_mode
    N, T = 5, 1008
    A, x, w, b, imp, go = _gcn_case(N, T, 5, 1e-4)
    tables = gcn_op.GraphTables(A)
    t = tables.on(dev)
    K = A.shape[0]
    W = w.view(K, 64, 64).to(dev)
    Aeff = (torch.tensor(A, dtype=torch.float32) * imp).to(dev)
    cc = gcn_tables.coefficients(Aeff, t['gidx_c']).contiguous()
    cr = gcn_tables.coefficients(Aeff, t['gidx_r']).contiguous()
    spf = gcn_op.SplitPlanes(*gcn_op.split_planes(W, tables.pairs_c))
    spb = gcn_op.SplitPlanes(*gcn_op.split_planes(W.transpose(1, 2), tables.pairs_r))
    x, dz = x.to(dev), go.to(dev)
    bias_cv = torch.randn(64, V, device=dev)
    z, part = gcn_op._gcn3h_forward(x, spf, cc, bias_cv, tables, True, math_mode.range_word(x))
    assert torch.equal(z, gcn_op._gcn3h_forward(x, spf, cc, bias_cv, tables, False, math_mode.range_word(x)))
    mean, var, _ = bn_op.moments(part, N * T * V)
    z64 = z.double()
    assert (mean - z64.mean(dim=(0, 2, 3))).abs().max().item() <= 1e-6 * z.abs().max().item()
    assert ((var - z64.var(dim=(0, 2, 3), unbiased=False)).abs() / var).max().item() <= 1e-5
    word = math_mode.range_word(dz)
    plain = gcn_op._gcn3h_data_gradient(dz, spb, cr, tables, None, None, word)
    add = torch.randn(N, 64, T, V, device=dev) * 1e-4
    mask = (torch.rand(N, 64, T, V, device=dev) > 0.5).to(torch.uint8)
    assert torch.equal(gcn_op._gcn3h_data_gradient(dz, spb, cr, tables, add, None, word), plain + add)
    assert torch.equal(gcn_op._gcn3h_data_gradient(dz, spb, cr, tables, add, mask, word), plain + add * mask)
    assert torch.equal(gcn_op._gcn3h_data_gradient(dz, spb, cr, tables, None, None, word), plain)      # run to run


def test_graph_conv_split16_aggregate_headroom(dev):
    """the aggregate of a unit is a coefficient-weighted SUM of neighbours: with edge importances of 6 the sums reach
    several times max |x|, which the range word's 8x headroom has to absorb (no overflow to inf in the fp16 parts)"""
    from pose2room_amd.p2rnet import gcn_op, math_mode
    N, T = 2, 64
    A, x, w, b, imp, go = _gcn_case(N, T, 11)
    imp = torch.full_like(imp, 6.0)
    tables = gcn_op.GraphTables(A)
    At = torch.tensor(A, dtype=torch.float32)
    xr, wr, br = (t.double().to(dev) for t in (x, w, b))
    zr = _gcn_reference(xr, wr, br, (At * imp).double().to(dev))
    with math_mode.use('split16'):
        z = gcn_op.graph_conv(x.to(dev), w.to(dev), b.to(dev), (At * imp).to(dev), tables)
    with math_mode.use('exact'):
        ze = gcn_op.graph_conv(x.to(dev), w.to(dev), b.to(dev), (At * imp).to(dev), tables)
    assert torch.isfinite(z).all()
    assert _rel(z, zr) <= 1.5 * _rel(ze, zr) + FLOOR
    math_mode.reset()


@pytest.mark.parametrize("N,T", [(2, 80), (3, 1008)])
def test_graph_conv_split16_heavy_tailed_gradient(dev, N, T):
    """the graph conv's data gradient on a gradient whose frames differ by up to 2^20 in magnitude (see
    test_tconvh_heavy_tailed_gradient): per run of frames within 1.5x the exact kernel's relative error"""
    from pose2room_amd.p2rnet import gcn_op, gcn_tables, math_mode
    A, x, w, b, imp, go = _gcn_case(N, T, 13)
    tables = gcn_op.GraphTables(A)
    assert tables.gen3h
    t = tables.on(dev)
    K = A.shape[0]
    W = w.view(K, 64, 64).to(dev)
    Aeff = (torch.tensor(A, dtype=torch.float32) * imp).to(dev)
    cr = gcn_tables.coefficients(Aeff, t['gidx_r']).contiguous()
    dz, k = _tail_profile(N, T, 22)
    dz = dz.to(dev)
    ref = torch.einsum('kdc,nkdtv->nctv', W.double(), torch.einsum('nctw,kvw->nkctv', dz.double(), Aeff.double()))
    spb = gcn_op.SplitPlanes(*gcn_op.split_planes(W.transpose(1, 2), tables.pairs_r))
    got = gcn_op._gcn3h_data_gradient(dz, spb, cr, tables, None, None, math_mode.range_word(dz))
    exact = gcn_op._gcn2_forward(dz, gcn_op.permute_planes(W.transpose(1, 2).contiguous()), cr, t['stream_r'], None, tables, form=1)
    ws, we = _per_run_error(got, ref, k, 'split16'), _per_run_error(exact, ref, k, 'exact')
    for kk in sorted(ws):
        assert ws[kk] <= 1.5 * we[kk] + FLOOR, (kk, ws, we)


@pytest.mark.parametrize("N,T", [(2, 80), (3, 1008)])
def test_graph_conv_split16_weight_gradient_heavy_tailed(dev, N, T):
    """the split16 weight gradient (both operands runtime tensors, each with its range word; bias-table gradient from the
    same tile) on a heavy-tailed dz against float64: within 1.5x the exact kernel's error"""
    from pose2room_amd import _lib
    from pose2room_amd.p2rnet import gcn_op, gcn_tables, math_mode
    A, x, w, b, imp, go = _gcn_case(N, T, 17)
    tables = gcn_op.GraphTables(A)
    assert tables.gen3h
    t = tables.on(dev)
    K = A.shape[0]
    Aeff = (torch.tensor(A, dtype=torch.float32) * imp).to(dev)
    cr = gcn_tables.coefficients(Aeff, t['gidx_r']).contiguous()
    dz, k = _tail_profile(N, T, 23)
    x, dz = (x * 3.0).to(dev), dz.to(dev)
    U = torch.einsum('nitv,kvw->nkitw', x.double(), Aeff.double())
    ref = torch.einsum('nctw,nkitw->kci', dz.double(), U)                    # [k][c][ci]
    refb = dz.double().sum(dim=(0, 2))                                       # [c][v]
    lib, st = _lib.lib(), _lib.current_stream(dev)
    NB = 256
    out = {}
    for name in ('exact', 'split16'):
        part = torch.full((NB, K, 64, 64), float('nan'), device=dev)
        bpart = torch.full((NB, 64, V), float('nan'), device=dev)
        if name == 'exact':
            _lib.check(lib.p2r_stgcn_gcn3_weight_grad(N, T, V, K, cr.shape[0], _lib.ptr(x), _lib.ptr(dz), _lib.ptr(cr), NB,
                                                      _lib.ptr(part), _lib.ptr(bpart), st), 'gcn3_weight_grad')
        else:
            _lib.check(lib.p2r_stgcn_gcn3h_weight_grad(N, T, V, K, cr.shape[0], _lib.ptr(x), _lib.ptr(dz), _lib.ptr(cr), NB,
                                                       _lib.ptr(part), _lib.ptr(bpart), _lib.ptr(math_mode.range_word(x)),
                                                       _lib.ptr(math_mode.range_word(dz)), st), 'gcn3h_weight_grad')
        out[name] = (part.double().sum(0).transpose(1, 2), bpart.double().sum(0))
    for i, (what, r) in enumerate((('dW', ref), ('dbias table', refb))):
        es, ee = _rel(out['split16'][i], r), _rel(out['exact'][i], r)
        assert es <= 1.5 * ee + FLOOR, (what, es, ee)


@pytest.mark.parametrize("N,T", [(2, 40), (1, 7), (3, 130)])
def test_split16_mode_falls_back_to_the_exact_kernels(dev, N, T):
    """shapes the split kernels do not take (sequence lengths that are not whole 16-frame tiles) run the exact kernels in
    split16 mode: no split entry point is called (they are replaced by a tripwire for the duration), and the results equal
    the exact mode's up to the order-of-arrival noise of the first-generation kernels' LDS atomics"""
    from pose2room_amd import _lib
    from pose2room_amd.p2rnet import gcn_op, math_mode, tconv_op
    A, x, w, b, imp, go = _gcn_case(N, T, 31)
    tables = gcn_op.GraphTables(A)
    At = torch.tensor(A, dtype=torch.float32)
    bn = torch.nn.BatchNorm2d(64).to(dev).train()
    conv = torch.nn.Conv2d(64, 64, (3, 1), (1, 1), (1, 0)).to(dev)
    lib = _lib.lib()
    names = ('p2r_stgcn_gcn3h_forward', 'p2r_stgcn_gcn3h_data_gradient', 'p2r_stgcn_gcn3h_weight_grad',
             'p2r_stgcn_gcn3h_coef_grad', 'p2r_stgcn_tconvh_forward')
    orig = {n: getattr(lib, n) for n in names}

    def tripwire(*a):
        raise AssertionError('a split16 kernel was launched on a shape it does not take')
    res = {}
    try:
        for m in ('exact', 'split16'):
            if m == 'split16':
                for n in names:
                    setattr(lib, n, tripwire)
            xd, wd, bd, idv = (t.to(dev).requires_grad_(True) for t in (x, w, b, imp))
            bn_m, conv_m = copy.deepcopy(bn), copy.deepcopy(conv)
            with math_mode.use(m):
                z = gcn_op.graph_conv(xd, wd, bd, At.to(dev) * idv, tables)
                u = tconv_op.bn_relu_tconv(z, bn_m, conv_m)
                u.backward(go.to(dev))
            res[m] = (z.detach(), u.detach(), xd.grad, wd.grad, bd.grad, idv.grad, conv_m.weight.grad, bn_m.weight.grad)
    finally:
        for n in names:
            setattr(lib, n, orig[n])
    assert torch.equal(res['exact'][0], res['split16'][0]) and torch.equal(res['exact'][1], res['split16'][1])   # forward: deterministic kernels
    for a, b_ in zip(res['exact'], res['split16']):
        assert _rel(a, b_.double()) <= 1e-5
    math_mode.reset()


def test_split16_mode_other_skeletons_run_exact(dev):
    """the split kernels are generated for the P2RNet skeleton; any other adjacency runs the exact kernels in either mode"""
    import numpy as np
    from pose2room_amd.p2rnet import gcn_op, math_mode
    rng = np.random.RandomState(3)
    K, Vj = 11, 25
    A = np.zeros((K, Vj, Vj), dtype=np.float32)
    for k in range(K):
        for v in range(Vj):
            A[k, v, (v + k) % Vj] = rng.uniform(0.2, 1.0)
    tables = gcn_op.GraphTables(A)
    assert not tables.gen3h and tables.pairs_c is None
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 64, 32, Vj, generator=g).to(dev)
    w = (torch.randn(K * 64, 64, generator=g) / 8).to(dev)
    out = {}
    for m in ('exact', 'split16'):
        with math_mode.use(m):
            out[m] = gcn_op.graph_conv(x, w, None, torch.tensor(A).to(dev), tables)
    assert torch.equal(out['exact'], out['split16'])
    math_mode.reset()


@pytest.mark.parametrize("N,T", [(1, 16), (3, 48), (2, 256)])
def test_split16_kernels_write_only_their_outputs(dev, N, T):
    """every output of the split16 kernels (result tensors, statistics partials, weight / adjacency gradient partials)
    sits in the middle of a larger buffer of sentinels: each kernel fills its output completely and touches nothing
    around it (the entry points are called with the raw pointers, as a C caller would)"""
    from pose2room_amd import _lib
    from pose2room_amd.p2rnet import gcn_op, gcn_tables, math_mode, tconv_op
    lib = _lib.lib()
    st = _lib.current_stream(dev)
    GUARD, SENT = 4096, -12345.0

    def guarded(shape):
        n = 1
        for s in shape:
            n *= s
        buf = torch.full((n + 2 * GUARD,), SENT, device=dev)
        return buf, buf[GUARD:GUARD + n].view(shape)

    def check(what, buf, out):
        n = out.numel()
        assert bool((buf[:GUARD] == SENT).all()) and bool((buf[GUARD + n:] == SENT).all()), what + ": wrote outside"
        assert not bool((out == SENT).any()), what + ": output not filled"
        assert bool(torch.isfinite(out).all()), what

    A, x, w, b, imp, go = _gcn_case(N, T, 77 + T, 1e-3)
    tables = gcn_op.GraphTables(A)
    t = tables.on(dev)
    K = A.shape[0]
    W = w.view(K, 64, 64).to(dev)
    Aeff = (torch.tensor(A, dtype=torch.float32) * imp).to(dev)
    cc = gcn_tables.coefficients(Aeff, t['gidx_c']).contiguous()
    cr = gcn_tables.coefficients(Aeff, t['gidx_r']).contiguous()
    spf = gcn_op.SplitPlanes(*gcn_op.split_planes(W, tables.pairs_c))
    spb = gcn_op.SplitPlanes(*gcn_op.split_planes(W.transpose(1, 2), tables.pairs_r))
    spd = gcn_op.SplitPlanes(*gcn_op.split_planes_coef_grad(W))
    x, dz = x.to(dev), go.to(dev)
    xw, dw = math_mode.range_word(x), math_mode.range_word(dz)
    bias_cv = torch.randn(64, V, device=dev)
    shape = (N, 64, T, V)
    nb = min(N * (T // 16), 256)
    with torch.cuda.device(dev):
        # graph conv forward + statistics
        zb, z = guarded(shape)
        pb, part = guarded((nb, 64, 3))
        _lib.check(lib.p2r_stgcn_gcn3h_forward(N, T, V, K, cc.shape[0], _lib.ptr(x), _lib.ptr(spf.wh), _lib.ptr(spf.winv),
                                               _lib.ptr(cc), _lib.ptr(bias_cv), _lib.ptr(z), _lib.ptr(part), None,
                                               _lib.ptr(xw), st), "gcn3h_forward")
        check("gcn3h forward", zb, z)
        check("gcn3h forward statistics", pb, part)
        # data gradient with a masked addend
        add = torch.randn(shape, device=dev) * 1e-3
        mask = (torch.rand(shape, device=dev) > 0.5).to(torch.uint8)
        db, dx = guarded(shape)
        _lib.check(lib.p2r_stgcn_gcn3h_data_gradient(N, T, V, K, cr.shape[0], _lib.ptr(dz), _lib.ptr(spb.wh),
                                                     _lib.ptr(spb.winv), _lib.ptr(cr), _lib.ptr(add), _lib.ptr(mask),
                                                     _lib.ptr(dx), _lib.ptr(dw), st), "gcn3h_data_gradient")
        check("gcn3h data gradient", db, dx)
        # weight gradient + bias table, adjacency gradient (partials per workgroup)
        NB = 256
        wb, wpart = guarded((NB, K, 64, 64))
        bb, bpart = guarded((NB, 64, V))
        _lib.check(lib.p2r_stgcn_gcn3h_weight_grad(N, T, V, K, cr.shape[0], _lib.ptr(x), _lib.ptr(dz), _lib.ptr(cr), NB,
                                                   _lib.ptr(wpart), _lib.ptr(bpart), _lib.ptr(xw), _lib.ptr(dw), st),
                   "gcn3h_weight_grad")
        check("gcn3h weight gradient", wb, wpart)
        check("gcn3h bias-table gradient", bb, bpart)
        cb, cpart = guarded((NB, cr.shape[0], V))
        _lib.check(lib.p2r_stgcn_gcn3h_coef_grad(N, T, V, K, cr.shape[0], _lib.ptr(x), _lib.ptr(dz), _lib.ptr(spd.wh),
                                                 _lib.ptr(spd.winv), NB, _lib.ptr(cpart), _lib.ptr(xw), st),
                   "gcn3h_coef_grad")
        check("gcn3h adjacency gradient", cb, cpart)
        # temporal conv: forward + statistics, data gradient + BatchNorm-backward sums
        xs, W3, scale, shift, bias = (q.to(dev) for q in _tconv_inputs(N, T, 5 + T))
        taps = tconv_op.SplitTaps(*tconv_op.split_taps(W3))
        chunk = 64 if T % 64 == 0 else (32 if T % 32 == 0 else 16)
        ob, out = guarded(shape)
        sb, spart = guarded((N * (T // chunk), 64, 3))
        _lib.check(lib.p2r_stgcn_tconvh_forward(N, T, V, _lib.ptr(xs), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(taps.wh),
                                                _lib.ptr(taps.winv), _lib.ptr(bias), _lib.ptr(out), _lib.ptr(spart), None,
                                                None, None, None, st), "tconvh_forward")
        check("tconvh forward", ob, out)
        check("tconvh forward statistics", sb, spart)
        fin = torch.stack([torch.zeros(64, device=dev), torch.ones(64, device=dev), scale, shift]).contiguous()
        du = xs * 1e-3
        uw = math_mode.range_word(du)
        gb, gout = guarded(shape)
        s2b, s2 = guarded((N * (T // chunk), 64, 2))
        _lib.check(lib.p2r_stgcn_tconvh_forward(N, T, V, _lib.ptr(du), None, None, _lib.ptr(taps.wh), _lib.ptr(taps.winv),
                                                None, _lib.ptr(gout), _lib.ptr(s2), None, _lib.ptr(xs), _lib.ptr(fin),
                                                _lib.ptr(uw), st), "tconvh data gradient")
        check("tconvh data gradient", gb, gout)
        check("tconvh data gradient sums", s2b, s2)
    math_mode.reset()
