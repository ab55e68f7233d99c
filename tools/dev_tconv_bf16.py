"""Dev: the split-bf16 PROTOTYPE of the temporal convolution (tools/ubench/tconv_bf16_proto.hip; not in the product)
against the product's exact-fp32 kernel and a float64 convolution: values, time.

    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared -o tools/ubench/libtconv_bf16_proto.so tools/ubench/tconv_bf16_proto.hip
    python tools/dev_tconv_bf16.py
"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pose2room_amd import _lib
from pose2room_amd.p2rnet import tconv_op

dev = torch.device('cuda:0')
VARIANT = os.environ.get('VARIANT', 'bf16')          # bf16: six-term split-bf16; f16 (= f16f4), f16f8: three-term scaled two-part fp16, 4- / 8-frame tiles;
#                                                      f16w: the walking design (tconv_f16w_proto.hip), T % 64 == 0
proto = ctypes.CDLL(os.path.join(ROOT, 'tools', 'ubench', f'libtconv_{VARIANT}_proto.so'))
entry = getattr(proto, 'proto_tconv3b_forward' if VARIANT == 'bf16' else 'proto_tconv3h_forward')
entry.restype = ctypes.c_int
V = 53
print('variant', VARIANT)


def run_proto(x, scale, shift, W, bias):
    out = torch.empty_like(x)
    N, _, T, _ = x.shape
    rc = entry(N, T, _lib.ptr(x), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(W), _lib.ptr(bias),
                                     _lib.ptr(out), _lib.current_stream(dev))
    assert rc == 0, rc
    return out


def reference64(x, scale, shift, W, bias):
    h = x.double()
    if scale is not None:
        h = torch.relu(h * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1))
    w = W.double().permute(1, 2, 0).unsqueeze(-1)            # (c, ci, tap, 1)
    return torch.nn.functional.conv2d(h, w, bias.double() if bias is not None else None, padding=(1, 0))


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


g = torch.Generator().manual_seed(0)
TILE = 64 if VARIANT.startswith('f16w') else (8 if VARIANT.endswith('f8') else 4)
for N, T in (((32, 1024),) if os.environ.get('REPS') else ((2, 16), (3, 64), (1, 4), (32, 1024))):   # REPS: under a profiler, the bench shape only
    if T % TILE:
        continue
    xbuf = torch.zeros(N * 64 * T * V + 64, device=dev)          # (f16w's last 8-float loads reach past joint 52)
    x = xbuf[:N * 64 * T * V].view(N, 64, T, V)
    x.copy_(torch.randn(N, 64, T, V, generator=g))
    scale, shift = (torch.rand(64, generator=g) + 0.5).to(dev), torch.randn(64, generator=g).to(dev)
    bias = torch.randn(64, generator=g).to(dev)
    W = (torch.randn(3, 64, 64, generator=g) / 8).to(dev)
    for name, (sc, sh, b) in {'forward (BatchNorm + ReLU on the input)': (scale, shift, bias), 'plain (data gradient)': (None, None, None)}.items():
        got = run_proto(x, sc, sh, W, b)
        prod = tconv_op._tconv(x, sc, sh, W, b, False, None) if T % 16 == 0 else None
        msg = f'N={N} T={T} {name}: '
        if N * T <= 4096:
            ref = reference64(x, sc, sh, W, b)
            rng = ref.abs().max().item()
            msg += f'prototype vs fp64 {((got.double() - ref).abs().max().item() / rng):.2e} of range'
            if prod is not None:
                msg += f', product (fp32 MFMA) vs fp64 {((prod.double() - ref).abs().max().item() / rng):.2e}'
        elif prod is not None:
            msg += f'prototype vs product {((got - prod).abs().max().item() / prod.abs().max().item()):.2e} of range'
        if N * T >= 32768:
            tp = timed(lambda: run_proto(x, sc, sh, W, b))
            tq = timed(lambda: tconv_op._tconv(x, sc, sh, W, b, False, None))
            gb = 2 * x.numel() * 4 / 1e9
            msg += f'; prototype {tp:.3f} ms ({gb / tp * 1e3 / 1e3:.2f} TB/s algorithmic), product {tq:.3f} ms'
        print(msg, flush=True)
