"""Dev: bn_apply variants (tools/ubench/bn/bn_*.so: nontemporal loads / stores)."""
import os, sys, glob, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pose2room_amd import _lib
dev = torch.device('cuda:0')
N, C, L = 32, 64, 1024 * 53
n = N * C * L
x = torch.randn(N, C, L, device=dev); res = torch.randn_like(x); y = torch.empty_like(x)
mask = torch.empty(x.shape, dtype=torch.uint8, device=dev)
sc = torch.rand(C, device=dev) + 0.5; sh = torch.rand(C, device=dev)
st, P = _lib.current_stream(dev), _lib.ptr
for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ubench', 'bn', 'bn_*.so'))) * 2:
    lib = ctypes.CDLL(path)
    fn = lambda: lib.p2r_bn_apply(N, C, L, P(x), P(sc), P(sh), P(res), 1, P(y), P(mask), st)
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f'{os.path.basename(path)[3:-3]:12s} {ms * 1e3:7.1f} us  {13 * n / ms / 1e9:6.2f} TB/s', flush=True)
