"""Dev: the split16 product kernels against the exact ones at the bench shape (32, 64, 1024, 53): time per launch.

    python tools/dev_split16_time.py [tconv] [gcn] [gcn_dw]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pose2room_amd.p2rnet import math_mode, tconv_op

dev = torch.device('cuda:0')
what = sys.argv[1:] or ['tconv', 'gcn', 'gcn_dw']
N, T, V = int(os.environ.get('N', 32)), int(os.environ.get('T', 1024)), 53


def timed(fn, reps=20):
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
if 'gcn' in what or 'gcn_dw' in what:
    import dev_split16_gcn  # noqa: F401
