"""Dev: time ablation builds of the temporal-conv weight-gradient kernel (tools/ubench/tw/tw_*.so; results of the ablations are wrong by design)."""
import os, sys, ctypes, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pose2room_amd import _lib
dev = torch.device('cuda:0')
N, T, V = 32, 1024, 53
x = torch.randn(N, 64, T, V, device=dev); dz = torch.randn(N, 64, T, V, device=dev)
scale = torch.rand(64, device=dev) + 0.5; shift = torch.randn(64, device=dev) * 0.1
for taps in (3, 1):
    part = torch.empty(256, taps, 64, 64, device=dev); bpart = torch.empty(256, 64, device=dev)
    for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ubench', 'tw', 'tw_*.so'))):
        lib = ctypes.CDLL(path)
        def call():
            rc = lib.p2r_stgcn_tconv_weight_grad(N, T, V, taps, _lib.ptr(x), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(dz), 256,
                                                 _lib.ptr(part), _lib.ptr(bpart), _lib.current_stream(dev))
            assert rc == 0, rc
        for _ in range(3): call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): call()
        e1.record(); e1.synchronize()
        print(f'taps={taps} {os.path.basename(path)[3:-3]:24s} {e0.elapsed_time(e1) / 10:.3f} ms', flush=True)
