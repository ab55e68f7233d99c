"""Dev: time + check the temporal-conv weight gradient (p2r_stgcn_tconv_weight_grad) against an fp64 einsum."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pose2room_amd import _lib
dev = torch.device('cuda:0')
torch.manual_seed(0)
st = _lib.current_stream(dev)
def t(fn, reps=40):
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
for (N, T, V, taps) in ((32, 1024, 53, 3), (32, 1024, 53, 1), (32, 1024, 20, 1), (3, 7, 53, 3), (2, 16, 53, 3), (1, 1, 53, 3), (5, 9, 20, 1), (3, 11, 31, 3)):
    x = torch.randn(N, 64, T, V, device=dev)
    du = torch.randn(N, 64, T, V, device=dev)
    sc, sh = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
    part = torch.empty(256, 64, 64, taps, device=dev); bp = torch.empty(256, 64, device=dev)
    fn = lambda: _lib.check(_lib.lib().p2r_stgcn_tconv_weight_grad(N, T, V, taps, _lib.ptr(x), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(du), 256, _lib.ptr(part), _lib.ptr(bp), st), 'wg')
    ms = t(fn)
    got, gb = part.double().sum(0), bp.double().sum(0)
    h = torch.relu(x.double() * sc.double().view(1, 64, 1, 1) + sh.double().view(1, 64, 1, 1))
    halo = (taps - 1) // 2
    hp = torch.nn.functional.pad(h, (0, 0, halo, halo))
    ref = torch.stack([torch.einsum('nctv,nitv->ci', du.double(), hp[:, :, p:p + T]) for p in range(taps)], -1)
    e = (got - ref).abs().max().item() / ref.abs().max().item()
    rb = du.double().sum((0, 2, 3))
    eb = (gb - rb).abs().max().item() / rb.abs().max().item()
    print(f'N={N} T={T} V={V} taps={taps}: {ms:.3f} ms  rel err dW {e:.2e} db {eb:.2e}')
