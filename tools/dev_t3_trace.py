"""Dev: cycle trace of one tile of tconv3: where do the cycles of a tile go, per wave?
    bash tools/build_trace_lib.sh && python tools/dev_t3_trace.py    (instrumented copy of the library: tools/ubench/libp2r_hip_trace.so)
The stamps are s_memtime (shader clock cycles): 16 MFMAs = 512 cycles."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from pose2room_amd.p2rnet import tconv_op
from pose2room_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "libp2r_hip_trace.so")
dev = torch.device('cuda:0')
N, T, V = 32, 1024, 53
g = torch.Generator().manual_seed(0)
x = torch.randn(N, 64, T, V, generator=g).to(dev)
z = torch.randn(N, 64, T, V, generator=g).to(dev)
scale, shift = (torch.rand(64, generator=g) + 0.5).to(dev), torch.randn(64, generator=g).to(dev)
bias = torch.randn(64, generator=g).to(dev)
fin = torch.randn(4, 64, generator=g).to(dev)
W3 = (torch.randn(3, 64, 64, generator=g) / 8).to(dev)
names = {0: 'tile start', 17: 'mfma done', 18: 'stats done', 19: 'tile end'}
for ph in range(4):
    names[1 + 4 * ph] = f'ph{ph} top'; names[2 + 4 * ph] = f'ph{ph} vm waited'; names[3 + 4 * ph] = f'ph{ph} xform done'
    names[20 + ph] = f'ph{ph} tap0 done'; names[24 + ph] = f'ph{ph} tap1 done'; names[4 + 4 * ph] = f'ph{ph} end'
order = [0] + sum([[1 + 4 * ph, 2 + 4 * ph, 3 + 4 * ph, 20 + ph, 24 + ph, 4 + 4 * ph] for ph in range(4)], []) + [17, 18, 19]
for name, kw in {'fwd+stats': dict(scale=scale, shift=shift, bias=bias, want_stats=True), 'dgrad': dict(scale=None, shift=None, bias=None),
                 'dgrad+bnbwd': dict(scale=None, shift=None, bias=None, want_stats=True, bwd=(z, fin))}.items():
    for _ in range(3):
        tconv_op._tconv(x, kw['scale'], kw['shift'], W3, kw['bias'], kw.get('want_stats', False), kw.get('bwd'))
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 256)()
    assert _lib.lib().p2r_debug_t3_trace(buf) == 0
    t = np.array(buf, dtype=np.int64).reshape(8, 32)
    t0 = t[:, 0].min()
    print(name)
    for w in (0, 4, 7):
        prev = t[w, 0]
        line = []
        for i in order:
            line.append(f'{names[i]} +{t[w, i] - prev}')
            prev = t[w, i]
        print(f' wave {w}: total {t[w, 19] - t[w, 0]} cycles  ' + ' | '.join(line))
