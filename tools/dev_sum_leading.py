"""Dev: p2r_sum_leading (sum of the per-workgroup partials of the weight / adjacency / bias gradients over the leading axis)
at the shapes of a train step: time per launch, HBM rate, value against float64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pose2room_amd import _lib
dev = torch.device('cuda:0')
torch.manual_seed(0)


def t(fn, reps=50):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, shape, tr in (('graph conv dW partials', (256, 11, 64, 64), True), ('temporal conv dW partials', (256, 64, 64, 3), False),
                        ('bias-table partials', (256, 64, 53), False), ('temporal conv bias partials', (256, 64), False),
                        ('few rows', (8, 11, 64, 64), True), ('ragged rows', (77, 3, 64, 64), True), ('one row', (1, 64, 64), True)):
    part = torch.randn(shape, device=dev)
    ref = part.double().sum(0)
    if tr:
        ref = ref.transpose(-1, -2)
    got = _lib.sum_leading(part, tr64=tr)
    err = (got.double() - ref).abs().max().item() / ref.abs().max().item()
    again = _lib.sum_leading(part, tr64=tr)
    us = t(lambda: _lib.sum_leading(part, tr64=tr))
    print(f'{name} {tuple(shape)}: {us:.1f} us, {part.numel() * 4 / us / 1e6:.2f} TB/s, error {err:.1e} of range, '
          f'bit-reproducible {torch.equal(got, again)}', flush=True)
