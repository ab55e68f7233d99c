"""Dev: time the temporal-conv kernels at the bench shape (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pose2room_amd.p2rnet import tconv_op
dev = torch.device('cuda:0')
N, T, V = 32, 1024, 53
torch.manual_seed(0)
x = torch.randn(N, 64, T, V, device=dev)
sc, sh = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
b = torch.randn(64, device=dev)
def t(fn, reps=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
for taps in (3, 1):
    W = torch.randn(taps, 64, 64, device=dev) / 8
    print(f'taps={taps} fwd(bn,bias,stats) {t(lambda: tconv_op._tconv(x, sc, sh, W, b, True)):.3f} ms   '
          f'data-grad(plain) {t(lambda: tconv_op._tconv(x, None, None, W, None)):.3f} ms')
from pose2room_amd import _lib
import ctypes
for taps in (3, 1):
    part = torch.empty(256, taps, 64, 64, device=dev); bp = torch.empty(256, 64, device=dev)
    du = torch.randn(N, 64, T, V, device=dev)
    st = _lib.current_stream(dev)
    fn = lambda: _lib.check(_lib.lib().p2r_stgcn_tconv_weight_grad(N, T, V, taps, _lib.ptr(x), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(du), 256, _lib.ptr(part), _lib.ptr(bp), st), 'wg')
    print(f'taps={taps} weight_grad {t(fn):.3f} ms')
