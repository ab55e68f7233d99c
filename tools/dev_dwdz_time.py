"""Dev: time the temporal-conv backward tail (p2r_stgcn_tconv_weight_grad_dz) at the bench shape."""
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
N, T, V = 32, 1024, 53
x = torch.randn(N, 64, T, V, device=dev); du = torch.randn(N, 64, T, V, device=dev); dh = torch.randn(N, 64, T, V, device=dev)
dz = torch.empty_like(x)
fin = torch.stack([torch.randn(64, device=dev) * 0.1, torch.rand(64, device=dev) + 0.5, torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1]).contiguous()
m12 = (torch.randn(2, 64, device=dev) * 0.01).contiguous()
part = torch.empty(256, 64, 64, 3, device=dev); bp = torch.empty(256, 64, device=dev)
fn = lambda: _lib.check(_lib.lib().p2r_stgcn_tconv_weight_grad_dz(N, T, V, 3, _lib.ptr(x), _lib.ptr(fin), _lib.ptr(du), _lib.ptr(dh), _lib.ptr(m12), _lib.ptr(dz), 256, _lib.ptr(part), _lib.ptr(bp), st), 'wgdz')
print(f'weight_grad_dz {t(fn):.4f} ms')
fn2 = lambda: _lib.check(_lib.lib().p2r_stgcn_tconv_weight_grad(N, T, V, 3, _lib.ptr(x), _lib.ptr(fin[2]), _lib.ptr(fin[3]), _lib.ptr(du), 256, _lib.ptr(part), _lib.ptr(bp), st), 'wg')
print(f'weight_grad    {t(fn2):.4f} ms')
word = torch.zeros(1, dtype=torch.int32, device=dev)
fn3 = lambda: _lib.check(_lib.lib().p2r_stgcn_tconv_weight_grad_dz_amax(N, T, V, 3, _lib.ptr(x), _lib.ptr(fin), _lib.ptr(du), _lib.ptr(dh), _lib.ptr(m12), _lib.ptr(dz), 256, _lib.ptr(part), _lib.ptr(bp), _lib.ptr(word), st), 'wgdz_amax')
print(f'weight_grad_dz_amax (split16 mode: + range word of dz) {t(fn3):.4f} ms; word {word.view(torch.float32).item():.4f} vs max |dz| {dz.abs().max().item():.4f}')
