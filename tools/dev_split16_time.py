"""Dev: the split16 product kernels against the exact ones at the bench shape (32, 64, 1024, 53): time per launch.

    python tools/dev_split16_time.py [tconv] [gcn] [gcn_dw]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pose2room_amd.p2rnet import math_mode, tconv_op

dev = torch.device('cuda:0')
what = sys.argv[1:] or ['tconv', 'gcn', 'gcn_dw', 'gcn_dc']
N, T, V = int(os.environ.get('N', 32)), int(os.environ.get('T', 1024)), 53


def timed(fn, reps=20):
    if os.environ.get('REPS'):        # under a profiler: a couple of launches
        fn(); fn()
        torch.cuda.synchronize()
        return 0.0
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


g = torch.Generator().manual_seed(0)
x = torch.randn(N, 64, T, V, generator=g).to(dev)
if 'tconv' in what:
    W3 = (torch.randn(3, 64, 64, generator=g) / 8).to(dev)
    scale, shift = (torch.rand(64, generator=g) + 0.5).to(dev), torch.randn(64, generator=g).to(dev)
    bias = torch.randn(64, generator=g).to(dev)
    st = tconv_op.SplitTaps(*tconv_op.split_taps(W3))
    Wp = tconv_op._permute_taps(W3)
    fin = torch.stack([torch.zeros(64, device=dev), torch.ones(64, device=dev), scale, shift]).contiguous()
    du = x * 1e-4
    word = math_mode.range_word(du)
    rows = [
        ('forward + transform + statistics', lambda: tconv_op._tconv(x, scale, shift, None, bias, True, Wp=Wp),
         lambda: tconv_op._tconvh(x, scale, shift, st, bias, True)),
        ('forward + transform', lambda: tconv_op._tconv(x, scale, shift, None, bias, False, Wp=Wp),
         lambda: tconv_op._tconvh(x, scale, shift, st, bias, False)),
        ('data gradient + BatchNorm-backward sums', lambda: tconv_op._tconv(du, None, None, None, None, True, bwd=(x, fin), Wp=Wp),
         lambda: tconv_op._tconvh(du, None, None, st, None, True, bwd=(x, fin), x_word=word)),
        ('plain', lambda: tconv_op._tconv(du, None, None, None, None, False, Wp=Wp),
         lambda: tconv_op._tconvh(du, None, None, st, None, False, x_word=word)),
    ]
    for name, fe, fs in rows:
        a, b = fe(), fs()
        a, b = (a[0], b[0]) if isinstance(a, tuple) else (a, b)
        d = (a - b).abs().max().item() / a.abs().max().item()
        print(f'tconv {name}: exact {timed(fe):.3f} ms, split16 {timed(fs):.3f} ms, difference {d:.2e} of range', flush=True)
    print(f'absmax pass: {timed(lambda: math_mode.range_word(du)):.3f} ms')
if 'gcn' in what or 'gcn_dw' in what or 'gcn_dc' in what:
    from pose2room_amd.p2rnet import gcn_op, gcn_tables
    from pose2room_amd.p2rnet.modules.stgcn_layers import Graph
    A = Graph().A
    K = A.shape[0]
    tables = gcn_op.GraphTables(A)
    t = tables.on(dev)
    assert tables.gen3h
    W = (torch.randn(K, 64, 64, generator=g) / 8).to(dev)
    Aeff = (torch.tensor(A, dtype=torch.float32) * (1 + 0.1 * torch.randn(K, V, V, generator=g))).to(dev)
    cc = gcn_tables.coefficients(Aeff, t['gidx_c']).contiguous()
    cr = gcn_tables.coefficients(Aeff, t['gidx_r']).contiguous()
    bias = torch.randn(64, V, generator=g).to(dev)
    xa = torch.relu(x + 0.3)
    dz = x * 1e-4
if 'gcn' in what:
    spf = gcn_op.SplitPlanes(*gcn_op.split_planes(W, tables.pairs_c))
    spb = gcn_op.SplitPlanes(*gcn_op.split_planes(W.transpose(1, 2), tables.pairs_r))
    wpf, wpb = gcn_op.permute_planes(W), gcn_op.permute_planes(W.transpose(1, 2).contiguous())
    xw, dw_ = math_mode.range_word(xa), math_mode.range_word(dz)
    add = x * 1e-4
    mask = (torch.rand(N, 64, T, V, generator=g) > 0.5).to(torch.uint8).to(dev)
    rows = [
        ('forward + statistics', lambda: gcn_op._gcn2_forward(xa, wpf, cc, t['stream_c'], bias, tables, True, form=0),
         lambda: gcn_op._gcn3h_forward(xa, spf, cc, bias, tables, True, xw)),
        ('data gradient + masked addend',
         lambda: gcn_op._gcn2_forward(dz, wpb, cr, t['stream_r'], None, tables, addend=add, form=1, addend_mask=mask),
         lambda: gcn_op._gcn3h_data_gradient(dz, spb, cr, tables, add, mask, dw_)),
        ('data gradient, plain', lambda: gcn_op._gcn2_forward(dz, wpb, cr, t['stream_r'], None, tables, form=1),
         lambda: gcn_op._gcn3h_data_gradient(dz, spb, cr, tables, None, None, dw_)),
    ]
    for name, fe, fs in rows:
        a, b = fe(), fs()
        a, b = (a[0] if isinstance(a, tuple) else a), (b[0] if isinstance(b, tuple) else b)
        d = (a - b).abs().max().item() / a.abs().max().item()
        print(f'gcn {name}: exact {timed(fe):.3f} ms, split16 {timed(fs):.3f} ms, difference {d:.2e} of range', flush=True)
if 'gcn_dw' in what:
    import ctypes
    from pose2room_amd import _lib
    lib = _lib.lib()
    xw, dw_ = math_mode.range_word(xa), math_mode.range_word(dz)
    NB = 256
    part = torch.empty(NB, K, 64, 64, device=dev)
    bpart = torch.empty(NB, 64, V, device=dev)
    st = _lib.current_stream(dev)

    def fe():
        _lib.check(lib.p2r_stgcn_gcn3_weight_grad(N, T, V, K, cr.shape[0], _lib.ptr(xa), _lib.ptr(dz), _lib.ptr(cr), NB,
                                                  _lib.ptr(part), _lib.ptr(bpart), st), 'gcn3_weight_grad')
        return part.sum(0), bpart.sum(0)

    def fs():
        _lib.check(lib.p2r_stgcn_gcn3h_weight_grad(N, T, V, K, cr.shape[0], _lib.ptr(xa), _lib.ptr(dz), _lib.ptr(cr), NB,
                                                   _lib.ptr(part), _lib.ptr(bpart), _lib.ptr(xw), _lib.ptr(dw_), st), 'gcn3h_weight_grad')
        return part.sum(0), bpart.sum(0)
    (a, ab), (b, bb) = fe(), fs()
    ts = timed(lambda: (part.sum(0), bpart.sum(0)))
    print(f'gcn weight gradient (+ bias table): exact {timed(fe) - ts:.3f} ms, split16 {timed(fs) - ts:.3f} ms (kernels alone), difference '
          f'{((a - b).abs().max() / a.abs().max()).item():.2e} / {((ab - bb).abs().max() / ab.abs().max()).item():.2e} of range', flush=True)
if 'gcn_dc' in what:
    from pose2room_amd import _lib
    lib = _lib.lib()
    xw = math_mode.range_word(xa)
    NB = 256
    ltot = cr.shape[0]
    part = torch.empty(NB, ltot, V, device=dev)
    st = _lib.current_stream(dev)
    wpf = gcn_op.permute_planes(W)
    spd = gcn_op.SplitPlanes(*gcn_op.split_planes_coef_grad(W))

    def fe():
        _lib.check(lib.p2r_stgcn_gcn3_coef_grad(N, T, V, K, ltot, _lib.ptr(xa), _lib.ptr(dz), _lib.ptr(wpf), NB, _lib.ptr(part), st), 'coef_grad')
        return part.sum(0)

    def fs():
        _lib.check(lib.p2r_stgcn_gcn3h_coef_grad(N, T, V, K, ltot, _lib.ptr(xa), _lib.ptr(dz), _lib.ptr(spd.wh), _lib.ptr(spd.winv), NB,
                                                 _lib.ptr(part), _lib.ptr(xw), st), 'coef_gradh')
        return part.sum(0)
    a, b = fe(), fs()
    ts = timed(lambda: part.sum(0))
    print(f'gcn adjacency gradient: exact {timed(fe) - ts:.3f} ms, split16 {timed(fs) - ts:.3f} ms (kernels alone), difference '
          f'{((a - b).abs().max() / a.abs().max()).item():.2e} of range', flush=True)
