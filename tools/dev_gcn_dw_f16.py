"""Dev: the PROTOTYPE graph-conv weight gradient on two-part fp16 products (tools/ubench/gcn3dwh_proto.hip; not in the
product) against the product's exact-fp32 `p2r_stgcn_gcn3_weight_grad` and a float64 einsum: values, time.

    python tools/gen_gcn_dwh_sched.py
    hipcc -O3 -fno-slp-vectorize -std=c++17 -fPIC --offload-arch=gfx950 -shared -I tools/ubench -o tools/ubench/libgcn3dwh_proto.so tools/ubench/gcn3dwh_proto.hip
    python tools/dev_gcn_dw_f16.py
"""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pose2room_amd import _lib
from pose2room_amd.p2rnet.modules.stgcn_layers import Graph
from pose2room_amd.p2rnet import gcn_op, gcn_tables

dev = torch.device('cuda:0')
proto = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", os.environ.get("PROTO_LIB", "libgcn3dwh_proto.so")))
proto.proto_gcn3dwh.restype = ctypes.c_int
lib = _lib.lib()
A = Graph().A
K, V = A.shape[0], A.shape[1]
ncs = proto.proto_gcn3dwh_stream(None)
sbuf = (ctypes.c_int * ncs)()
proto.proto_gcn3dwh_stream(sbuf)
tables = gcn_op.GraphTables(A)
t = tables.on(dev)


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
for N, T in (((32, 1024),) if os.environ.get('REPS') else ((2, 16), (3, 64), (32, 1024))):
    xbuf = torch.zeros(N * 64 * T * V + 16, device=dev)            # (the last chunk load of a row reaches 3 floats past joint 52)
    x = xbuf[:N * 64 * T * V].view(N, 64, T, V)
    x.copy_(torch.relu(torch.randn(N, 64, T, V, generator=g) + 0.3))
    dz = (torch.randn(N, 64, T, V, generator=g) * 1e-4).to(dev)                        # a gradient's magnitude
    Aeff = (torch.tensor(A, dtype=torch.float32) * (1 + 0.1 * torch.randn(K, V, V, generator=g))).to(dev)
    coef_r = gcn_tables.coefficients(Aeff, t['gidx_r']).contiguous()                  # (ltot, V) row-form table
    ltot = coef_r.shape[0]
    gs = 2.0 ** round(math.log2(64.0 / float(dz.abs().max())))
    stream_idx = torch.tensor(list(sbuf), dtype=torch.long, device=dev)
    coef_s = (coef_r * gs).flatten()[stream_idx].contiguous()              # the coefficient stream in schedule order
    st = _lib.current_stream(dev)
    nb = proto.proto_gcn3dwh(N, T, ltot, None, None, None, ctypes.c_float(1.0 / gs), None, None)
    assert nb > 0, nb
    part = torch.empty(nb, K, 64, 64, device=dev)

    def run_proto():
        rc = proto.proto_gcn3dwh(N, T, ltot, _lib.ptr(x), _lib.ptr(dz), _lib.ptr(coef_s), ctypes.c_float(1.0 / gs), _lib.ptr(part), st)
        assert rc == nb, rc
        return part.sum(0).transpose(1, 2)                                            # -> [k][c][ci]

    NB = 256
    ppart = torch.empty(NB, K, 64, 64, device=dev)

    def run_prod():
        _lib.check(lib.p2r_stgcn_gcn3_weight_grad(N, T, V, K, ltot, _lib.ptr(x), _lib.ptr(dz), _lib.ptr(coef_r), NB, _lib.ptr(ppart),
                                                  None, st), "gcn3_weight_grad")
        return ppart.sum(0).transpose(1, 2)

    got, prod = run_proto(), run_prod()
    msg = f'N={N} T={T}: '
    if N * T <= 4096:
        U = torch.einsum('nitv,kvw->nkitw', x.double(), Aeff.double())
        ref = torch.einsum('nctw,nkitw->kci', dz.double(), U)
        rng = ref.abs().max().item()
        msg += (f'prototype vs fp64 {((got.double() - ref).abs().max().item() / rng):.2e} of range, '
                f'product (fp32 MFMA) vs fp64 {((prod.double() - ref).abs().max().item() / rng):.2e}')
    else:
        msg += f'prototype vs product {((got - prod).abs().max().item() / prod.abs().max().item()):.2e} of range'
        tp, tq = timed(run_proto), timed(run_prod)
        ts, tqs = timed(lambda: part.sum(0)), timed(lambda: ppart.sum(0))
        msg += (f'; kernels alone: prototype {tp - ts:.3f} ms, product {tq - tqs:.3f} ms -> x{(tq - tqs) / (tp - ts):.2f} '
                f'(sum of the {nb} / {NB} partials: {ts:.3f} / {tqs:.3f} ms)')
    print(msg, flush=True)
