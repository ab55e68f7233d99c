"""Dev: the PROTOTYPE temporal-conv weight gradient on two-part fp16 products (tools/ubench/tconv_dw_f16_proto.hip; not in
the product) against the product's exact-fp32 `p2r_stgcn_tconv_weight_grad` and a float64 reference: values, time.

    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared -o tools/ubench/libtconv_dw_f16_proto.so tools/ubench/tconv_dw_f16_proto.hip
    python tools/dev_tconv_dw_f16.py
"""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pose2room_amd import _lib

dev = torch.device('cuda:0')
proto = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", os.environ.get("PROTO_LIB", "libtconv_dw_f16_proto.so")))
proto.proto_tconv_dw_f16.restype = ctypes.c_int
lib = _lib.lib()
V, C = 53, 64


def padded(t):
    """the tensor in a buffer with 64 readable floats behind it (the kernel's last 8-float loads reach past joint 52)"""
    buf = torch.zeros(t.numel() + 64, dtype=t.dtype, device=t.device)
    buf[:t.numel()] = t.flatten()
    return buf[:t.numel()].view(t.shape), buf


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


g = torch.Generator().manual_seed(0)
for N, T in (((32, 1024),) if os.environ.get('REPS') else ((2, 64), (3, 128), (32, 1024))):       # REPS: under a profiler, the bench shape only
    z, zbuf = padded(torch.randn(N, C, T, V, generator=g).to(dev))
    dout, dbuf = padded((torch.randn(N, C, T, V, generator=g) * 1e-4).to(dev))          # a gradient's magnitude
    scale = (1 + 0.1 * torch.randn(C, generator=g)).to(dev)
    shift = (0.2 * torch.randn(C, generator=g)).to(dev)
    gs = 2.0 ** round(math.log2(64.0 / float(dout.abs().max())))
    st = _lib.current_stream(dev)
    nb = proto.proto_tconv_dw_f16(N, T, None, None, None, None, ctypes.c_float(gs), None, None)
    part = torch.empty(nb, C, C, 3, device=dev)

    def run_proto():
        rc = proto.proto_tconv_dw_f16(N, T, _lib.ptr(z), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(dout), ctypes.c_float(gs),
                                      _lib.ptr(part), st)
        assert rc == nb, rc
        return part.sum(0)

    NB = 256                                                   # tconv_op._N_BLOCKS
    ppart = torch.empty(NB, C, C, 3, device=dev)

    def run_prod():
        _lib.check(lib.p2r_stgcn_tconv_weight_grad(N, T, V, 3, _lib.ptr(z), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(dout), NB,
                                                   _lib.ptr(ppart), None, st), "tconv_weight_grad")
        return ppart.sum(0)

    got, prod = run_proto(), run_prod()
    msg = f'N={N} T={T}: '
    if N * T <= 4096:
        h = torch.relu(z.double() * scale.double()[None, :, None, None] + shift.double()[None, :, None, None])
        hp = torch.nn.functional.pad(h, (0, 0, 1, 1))
        ref = torch.stack([torch.einsum('nctj,nitj->ci', dout.double(), hp[:, :, p:p + T]) for p in range(3)], dim=-1)
        rng = ref.abs().max().item()
        msg += (f'prototype vs fp64 {((got.double() - ref).abs().max().item() / rng):.2e} of range, '
                f'product (fp32 MFMA) vs fp64 {((prod.double() - ref).abs().max().item() / rng):.2e}')
    else:
        msg += f'prototype vs product {((got - prod).abs().max().item() / prod.abs().max().item()):.2e} of range'
        tp, tq = timed(run_proto), timed(run_prod)
        ts = timed(lambda: part.sum(0)); tqs = timed(lambda: ppart.sum(0))
        msg += (f'; prototype {tp:.3f} ms, product {tq:.3f} ms (both with the sum of their {nb} / {NB} partials: {ts:.3f} / {tqs:.3f} ms)'
                f'  -> x{tq / tp:.2f}; kernels alone {tp - ts:.3f} vs {tq - tqs:.3f} ms -> x{(tq - tqs) / (tp - ts):.2f}')
    print(msg, flush=True)
