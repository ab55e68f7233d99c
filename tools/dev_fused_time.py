"""Dev: time the first-generation forward kernel (tconv_fused_kernel) at the shapes the step uses it for."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pose2room_amd.p2rnet import tconv_op
dev = torch.device('cuda:0')
torch.manual_seed(0)
def t(fn, reps=40):
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
for (N, T, V, taps) in ((32, 1024, 20, 1), (32, 1000, 53, 3), (32, 1024, 25, 3)):
    x = torch.randn(N, 64, T, V, device=dev)
    sc, sh = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
    b = torch.randn(64, device=dev)
    W = torch.randn(taps, 64, 64, device=dev) / 8
    u3, u2 = tconv_op.USE_GEN3, getattr(tconv_op, 'USE_GEN2', None)
    tconv_op.USE_GEN3 = False
    if u2 is not None: tconv_op.USE_GEN2 = False
    print(f'N={N} T={T} V={V} taps={taps}: fwd(bn,bias,stats) {t(lambda: tconv_op._tconv(x, sc, sh, W, b, True)):.3f} ms')
    tconv_op.USE_GEN3 = u3
    if u2 is not None: tconv_op.USE_GEN2 = u2
