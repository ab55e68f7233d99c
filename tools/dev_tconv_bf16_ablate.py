import ctypes, os, sys, torch
sys.path.insert(0, '/root/repo')
from pose2room_amd import _lib
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
N, T, V = 32, 1024, 53
x = torch.randn(N, 64, T, V, generator=g).to(dev)
W = (torch.randn(3, 64, 64, generator=g) / 8).to(dev)
out = torch.empty_like(x)
for name in ['', '_NO_MFMA', '_NO_LOAD', '_NO_STORE', '_NO_LS']:
    lib = ctypes.CDLL(f'/root/repo/tools/ubench/libtconv_bf16_proto{name}.so')
    fn = lambda: lib.proto_tconv3b_forward(N, T, _lib.ptr(x), None, None, _lib.ptr(W), None, _lib.ptr(out), _lib.current_stream(dev))
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); e1.synchronize()
    print(f'{name or "full":10s} {e0.elapsed_time(e1) / 20:.3f} ms')
