"""Dev: time the BatchNorm streaming passes at the bench shape (N=32, C=64, L=1024*53) and report HBM rates."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pose2room_amd import _lib
dev = torch.device('cuda:0')
N, C, L = 32, 64, 1024 * 53
n = N * C * L
x = torch.randn(N, C, L, device=dev); res = torch.randn_like(x); dy = torch.randn_like(x)
y = torch.empty_like(x); dx = torch.empty_like(x); dres = torch.empty_like(x)
mask = torch.empty(x.shape, dtype=torch.uint8, device=dev)
v = [torch.rand(C, device=dev) + 0.5 for _ in range(6)]
part = torch.empty(N, C, 2, device=dev)
lib, st = _lib.lib(), _lib.current_stream(dev)
P = _lib.ptr


def t(fn, bytes_per_elt, name):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f'{name:34s} {ms * 1e3:7.1f} us  {bytes_per_elt * n / ms / 1e9:6.2f} TB/s', flush=True)


t(lambda: _lib.check(lib.p2r_bn_apply(N, C, L, P(x), P(v[0]), P(v[1]), P(res), 1, P(y), P(mask), st), 'a'), 13, 'bn_apply (res, relu, mask)')
t(lambda: _lib.check(lib.p2r_bn_bwd_reduce(N, C, L, P(dy), P(mask), P(x), P(v[0]), P(v[1]), 3, None, None, P(part), st), 'r'), 9, 'bn_bwd_reduce mask byte')
t(lambda: _lib.check(lib.p2r_bn_bwd_reduce(N, C, L, P(dy), None, P(x), P(v[0]), P(v[1]), 2, P(v[2]), P(v[3]), P(part), st), 'r'), 8, 'bn_bwd_reduce recomputed mask')
t(lambda: _lib.check(lib.p2r_bn_bwd_apply(N, C, L, P(dy), P(mask), P(x), P(v[0]), P(v[1]), P(v[2]), P(v[3]), P(v[4]), 3, None, None, P(dx), P(dres), st), 'b'), 17, 'bn_bwd_apply mask byte + dres')
t(lambda: _lib.check(lib.p2r_bn_bwd_apply(N, C, L, P(dy), None, P(x), P(v[0]), P(v[1]), P(v[2]), P(v[3]), P(v[4]), 2, P(v[2]), P(v[3]), P(dx), None, st), 'b'), 12, 'bn_bwd_apply recomputed mask')
t(lambda: y.copy_(x), 8, 'torch copy (reference rate)')
