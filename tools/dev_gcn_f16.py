"""Dev: the PROTOTYPE graph-conv forward on two-part fp16 products with plane-pair K (tools/ubench/gcn3h_proto.hip; not in
the product) against the product's exact-fp32 gcn3 kernel and a float64 einsum: values, time.

    python tools/gen_gcn_pair_sched.py
    hipcc -O3 -fno-slp-vectorize -DTRAIL_NOPS -std=c++17 -fPIC --offload-arch=gfx950 -shared -I tools/ubench -o tools/ubench/libgcn3h_proto.so tools/ubench/gcn3h_proto.hip
    python tools/dev_gcn_f16.py
"""
import ctypes, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from pose2room_amd import _lib
from pose2room_amd.p2rnet.modules.stgcn_layers import Graph
from pose2room_amd.p2rnet import gcn_op, gcn_tables

dev = torch.device('cuda:0')
FORM = os.environ.get("FORM", "c")      # "c": forward; "r": data gradient (libgcn3h_proto_r.so, built with -DH3_FORM_R from gcn3h_sched_r.inc)
proto = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", os.environ.get("PROTO_LIB", "libgcn3h_proto_r.so" if FORM == "r" else "libgcn3h_proto.so")))
proto.proto_gcn3h_forward.restype = ctypes.c_int
A = Graph().A
K, V = A.shape[0], A.shape[1]
tables = gcn_op.GraphTables(A)
t = tables.on(dev)
buf = (ctypes.c_int * 16)()
npairs = proto.proto_gcn3h_pairs(buf)
pairs = [(buf[2 * i], buf[2 * i + 1]) for i in range(npairs)]


def pack_weights(W):
    """W (K,64,64) [k][c][ci] -> (fp16 A operands [pair][phase][part][m][lane][8], scale 2^S)"""
    S = 10 - int(math.ceil(math.log2(float(W.abs().max()) + 1e-30)))          # 2^S max|W| in [2^9, 2^10]
    scale = 2.0 ** S
    Wsel = torch.zeros(npairs, 2, 64, 64, dtype=torch.float32, device=W.device)
    for pi, (a, b) in enumerate(pairs):
        Wsel[pi, 0] = W[a]
        if b >= 0:
            Wsel[pi, 1] = W[b]
    lane = torch.arange(64, device=W.device)
    kg, r = lane >> 4, lane & 15
    i = torch.arange(8, device=W.device)
    out = torch.empty(npairs, 4, 2, 4, 64, 8, dtype=torch.float16, device=W.device)
    for ph in range(4):
        for m in range(4):
            row = (16 * m + r)[:, None].expand(64, 8)
            ch = (16 * ph + kg)[:, None] + 4 * (i & 3)[None, :]             # k = 8 kg + i: channel kg + 4 (i & 3) ...
            half = (i >> 2)[None, :].expand(64, 8)                            # ... of plane a (i < 4) or b (i >= 4)
            wt = Wsel[:, half, row, ch] * scale                              # (P, 64, 8)
            p1 = wt.half()
            out[:, ph, 0, m] = p1
            out[:, ph, 1, m] = (wt - p1.float()).half()
    return out.contiguous(), scale


XS = float(os.environ.get("XS", "1"))     # (FORM=r sets it from the gradient's magnitude)
# Optional power-of-two pre-scale of the aggregate (carried by the coefficient table, undone by `scale`).  Kept as a knob
# because the fp16 residual x - fp16(x) of a small activation is a SUBNORMAL fp16 number; measured on gfx950 the error is
# the same at XS = 1, 32 and 1024 (9.3e-7 / 1.0e-6 of range), i.e. v_mfma_f32_16x16x32_f16 does not flush them.


def run_proto(x, Wp16, scale, coef1, bias):
    coef1 = (coef1 * XS).contiguous()
    scale = scale * XS
    z = torch.empty_like(x)
    N, _, T, _ = x.shape
    rc = proto.proto_gcn3h_forward(N, T, coef1.shape[0], _lib.ptr(x), _lib.ptr(Wp16), _lib.ptr(coef1), _lib.ptr(bias),
                                   ctypes.c_float(scale), _lib.ptr(z), _lib.current_stream(dev))
    assert rc == 0, rc
    return z


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
SIZES = ((32, 1024),) if os.environ.get('REPS') else ((2, 16), (3, 64), (32, 1024))     # REPS: under a profiler, the bench shape only
for N, T in SIZES:
    W = (torch.randn(K, 64, 64, generator=g) / 8).to(dev)
    Aeff = (torch.tensor(A, dtype=torch.float32) * (1 + 0.1 * torch.randn(K, V, V, generator=g))).to(dev)
    if FORM == "r":
        # dX = sum_k W_k^T (dZ . A_k^T): the same kernel over the row lists with the transposed weights; the input is a
        # GRADIENT (here ~1e-4): the coefficient table carries a power of two that lifts the aggregate into fp16's range
        x = (torch.randn(N, 64, T, V, generator=g) * 1e-4).to(dev)
        bias = torch.zeros(64, V, device=dev)
        cc = gcn_tables.coefficients(Aeff, t['gidx_r']).contiguous()
        Wk = W.transpose(1, 2).contiguous()                                       # [k][ci][c]
        XS = 2.0 ** round(math.log2(256.0 / float(x.abs().max())))
        stream, form = t['stream_r'], 1
        sub = 'nctw,kvw->nkctv'
    else:
        x = torch.relu(torch.randn(N, 64, T, V, generator=g) + 0.3).to(dev)       # like a BatchNorm + ReLU output
        bias = torch.randn(64, V, generator=g).to(dev)
        cc = gcn_tables.coefficients(Aeff, t['gidx_c']).contiguous()
        Wk = W
        stream, form = t['stream_c'], 0
        sub = 'nctv,kvw->nkctw'
    coef1 = torch.cat([cc, torch.zeros(1, V, device=dev)]).contiguous()
    Wp16, scale = pack_weights(Wk)
    got = run_proto(x, Wp16, scale, coef1, bias)
    Wp = gcn_op.permute_planes(Wk)
    prod = gcn_op._gcn2_forward(x, Wp, cc, stream, bias if FORM != "r" else None, tables, form=form)
    prod = prod[0] if isinstance(prod, tuple) else prod
    msg = f'{"data gradient" if FORM == "r" else "forward"} N={N} T={T}: '
    if N * T <= 4096:
        U = torch.einsum(sub, x.double(), Aeff.double())
        ref = torch.einsum('kdc,nkctw->ndtw', Wk.double(), U) + bias.double()[None, :, None, :]
        rng = ref.abs().max().item()
        msg += (f'prototype vs fp64 {((got.double() - ref).abs().max().item() / rng):.2e} of range, '
                f'product (fp32 MFMA) vs fp64 {((prod.double() - ref).abs().max().item() / rng):.2e}')
    else:
        msg += f'prototype vs product {((got - prod).abs().max().item() / prod.abs().max().item()):.2e} of range'
        tp = timed(lambda: run_proto(x, Wp16, scale, coef1, bias))
        tq = timed(lambda: gcn_op._gcn2_forward(x, Wp, cc, stream, bias if FORM != "r" else None, tables, form=form))
        msg += f'; prototype {tp:.3f} ms, product (no statistics) {tq:.3f} ms  -> x{tq / tp:.2f}'
    print(msg, flush=True)
    if FORM == "r":
        # the addend of the data gradient (the gradient of the residual branch), added in the store epilogue
        add = (torch.randn(N, 64, T, V, generator=g) * 1e-4).to(dev)
        proto.proto_gcn3h_addend(ctypes.c_void_p(add.data_ptr()))
        got_a = run_proto(x, Wp16, scale, coef1, bias); torch.cuda.synchronize()
        proto.proto_gcn3h_addend(None)
        prod_a = gcn_op._gcn2_forward(x, Wp, cc, stream, None, tables, addend=add, form=1)
        prod_a = prod_a[0] if isinstance(prod_a, tuple) else prod_a
        print(f'   with the addend: prototype - (plain prototype + addend) {(got_a - (got + add)).abs().max().item():.1e}, '
              f'vs product {((got_a - prod_a).abs().max() / prod_a.abs().max()).item():.2e} of range', flush=True)
    if FORM != "r":
        # the statistics epilogue: per-workgroup (count, mean, M2) per channel, merged (Chan) and compared with the
        # moments of the stored tensor and with what the product's gcn3 epilogue reports
        def merge(part):
            part = part.double()
            n, mean, m2 = part[:, :, 0], part[:, :, 1], part[:, :, 2]
            ntot = n.sum(0)
            mu = (n * mean).sum(0) / ntot
            return mu, (m2 + n * (mean - mu) ** 2).sum(0) / ntot
        nblk = min(N * (T // 16), 256)
        spart = torch.zeros(nblk, 64, 3, device=dev)
        proto.proto_gcn3h_stats(ctypes.c_void_p(spart.data_ptr()))
        got2 = run_proto(x, Wp16, scale, coef1, bias); torch.cuda.synchronize()
        if N * T > 4096:
            ts = timed(lambda: run_proto(x, Wp16, scale, coef1, bias))
            tqs = timed(lambda: gcn_op._gcn2_forward(x, Wp, cc, stream, bias, tables, want_stats=True, form=0))
            print(f'   with the statistics epilogue: prototype {ts:.3f} ms, product {tqs:.3f} ms -> x{tqs / ts:.2f}')
        proto.proto_gcn3h_stats(None)
        mu, var = merge(spart)
        g64 = got2.double()
        mu_t, var_t = g64.mean(dim=(0, 2, 3)), g64.var(dim=(0, 2, 3), unbiased=False)
        zp, ppart = gcn_op._gcn2_forward(x, Wp, cc, stream, bias, tables, want_stats=True, form=0)
        mu_p, var_p = merge(ppart.view(-1, 64, 3))
        print(f'   statistics epilogue: mean {((mu - mu_t).abs().max() / mu_t.abs().max()).item():.1e}, variance '
              f'{((var - var_t).abs() / var_t).max().item():.1e} (relative, vs the stored tensor in float64); product epilogue vs its '
              f'tensor: {((mu_p - zp.double().mean(dim=(0, 2, 3))).abs().max() / mu_t.abs().max()).item():.1e}, '
              f'{((var_p - zp.double().var(dim=(0, 2, 3), unbiased=False)).abs() / var_t).max().item():.1e}; outputs equal with / without: '
              f'{bool(torch.equal(got, got2))}', flush=True)
    if os.environ.get("PROFILE") and N * T > 4096:
        prof = torch.zeros(256, 8, 4, dtype=torch.int64, device=dev)
        proto.proto_gcn3h_profile(ctypes.c_void_p(prof.data_ptr()))
        run_proto(x, Wp16, scale, coef1, bias); torch.cuda.synchronize()
        proto.proto_gcn3h_profile(None)
        pm = prof.double().mean(0)                                 # (8 waves, 4)
        tot = pm.sum(1)
        print('profile (s_memtime ticks per wave, mean over the workgroups): wave  phase-start wait | bodies | to epilogue | epilogue')
        for w in range(8):
            print('   %d  %8.0f %8.0f %8.0f %8.0f   (%.0f %% / %.0f %% / %.0f %% / %.0f %%)' % ((w,) + tuple(pm[w].tolist()) + tuple((100 * pm[w] / tot[w]).tolist())))
