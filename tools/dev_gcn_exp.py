"""Dev experiment: forward kernel time vs neighbour-list length (results are wrong for L != real)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from pose2room_amd import _lib
dev = torch.device('cuda:0')
N, T, V, K = 32, 1024, 53, 11
x = torch.randn(N, 64, T, V, device=dev); W = torch.randn(K * 64, 64, device=dev) / 8
z = torch.empty_like(x)
for Lval in (1, 2, 4, 8, 12):
    Lk = [Lval] * K
    ltot = sum(Lk)
    nbr = torch.randint(0, V, (ltot, V), dtype=torch.uint8, device=dev)
    coef = torch.rand(ltot, V, device=dev)
    LkA = (ctypes.c_int * K)(*Lk)
    def call():
        assert _lib.lib().p2r_stgcn_gcn_forward(N, T, V, K, LkA, _lib.ptr(x), _lib.ptr(W), _lib.ptr(nbr), _lib.ptr(coef), None, _lib.ptr(z), None, _lib.current_stream(dev)) == 0
    for _ in range(2): call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): call()
    e1.record(); e1.synchronize()
    print(f'L={Lval:2d}  {e0.elapsed_time(e1) / 5:.3f} ms')
