"""Fused spatial graph convolution op (HIP kernels in csrc/stgcn_gcn.hip).

`graph_conv(x, weight, bias, Aeff, tables)` computes exactly what the reference's
ConvTemporalGraphical.forward does (stgcn_layers.py:57-67):
    einsum('nkctv,kvw->nctw', conv1x1(x; weight, bias).view(N,K,C,T,V), Aeff)
for C = 64, without materialising the K*C-channel tensor, and is differentiable
w.r.t. x, weight, bias and Aeff (= A * edge_importance).

Only the sparse form of Aeff is used (its zero pattern is the skeleton's, fixed);
the bias enters as bias_cv[c,w] = sum_k b_k[c] * sum_v Aeff[k,v,w], built with torch
ops so autograd provides db and the bias share of dAeff.
"""
import ctypes

import torch
from torch.autograd import Function

from .. import _lib
from . import bn_op, gcn_tables, math_mode

_N_BLOCKS = 256     # persistent workgroups of the reduction kernels (one per CU)


class GraphTables:
    """Device-resident neighbour tables of one adjacency pattern (built once)."""

    def __init__(self, A):
        self.K, self.V = int(A.shape[0]), int(A.shape[1])
        self.nbr_c, self.gidx_c, self.Lk_c = gcn_tables.build(A, transpose=False)
        self.nbr_r, self.gidx_r, self.Lk_r = gcn_tables.build(A, transpose=True)
        self.LkA_c = (ctypes.c_int * self.K)(*self.Lk_c)
        self.LkA_r = (ctypes.c_int * self.K)(*self.Lk_r)
        # static work schedules of the second-generation kernels (unit = (plane, joint) with a non-empty list).  They
        # exist for the P2RNet skeleton size only (csrc/stgcn_gcn2.hip is a V = 53 kernel) and only when the pattern
        # fits the kernel's record budget; any other adjacency runs on the first-generation kernels.
        self.stream_c = self.stream_r = None
        if self.V == 53 and self.K < 15:
            try:
                sc = gcn_tables.build_stream(self.nbr_c, self.gidx_c, self.Lk_c)[0]
                sr = gcn_tables.build_stream(self.nbr_r, self.gidx_r, self.Lk_r)[0]
                self.stream_c, self.stream_r = torch.from_numpy(sc), torch.from_numpy(sr)
            except gcn_tables.StreamBudgetError:
                pass
        # third generation (csrc/stgcn_gcn3.hip): statically scheduled for ONE pattern, the P2RNet skeleton's; taken
        # when the signatures of both table forms equal the ones the library was generated for
        self._gen3 = None
        self._gen3h = None
        self._pairs = (None, None)
        self._dev = {}

    @property
    def gen3(self):
        if self._gen3 is None:
            ok = False
            if self.gen2:
                lib = _lib.lib()
                ok = (gcn_tables.pattern_signature(self.nbr_c, self.gidx_c, self.Lk_c) == lib.p2r_stgcn_gcn3_signature(0)
                      and gcn_tables.pattern_signature(self.nbr_r, self.gidx_r, self.Lk_r) == lib.p2r_stgcn_gcn3_signature(1))
            self._gen3 = ok
        return self._gen3

    @property
    def gen2(self):
        """True when the second-generation kernels (static work streams) serve this adjacency pattern."""
        return self.stream_c is not None

    @property
    def gen3h(self):
        """True when the split16 kernels (csrc/stgcn_gcn3h_body.h: schedules generated for ONE pattern, like the third
        generation) serve this adjacency pattern; then `pairs_c` / `pairs_r` hold their plane pairs."""
        if self._gen3h is None:
            ok = False
            if self.gen3:
                lib = _lib.lib()
                sig_r = gcn_tables.pattern_signature(self.nbr_r, self.gidx_r, self.Lk_r)
                ok = (gcn_tables.pattern_signature(self.nbr_c, self.gidx_c, self.Lk_c) == lib.p2r_stgcn_gcn3h_signature(0)
                      and sig_r == lib.p2r_stgcn_gcn3h_signature(1) and sig_r == lib.p2r_stgcn_gcn3h_weight_grad_signature())
                if ok:
                    buf = (ctypes.c_int * 32)()
                    n = lib.p2r_stgcn_gcn3h_pairs(0, buf)
                    pc = [(buf[2 * i], buf[2 * i + 1]) for i in range(n)]
                    n = lib.p2r_stgcn_gcn3h_pairs(1, buf)
                    self._pairs = (pc, [(buf[2 * i], buf[2 * i + 1]) for i in range(n)])
            self._gen3h = ok
        return self._gen3h

    @property
    def pairs_c(self):
        """plane pairs of the split16 forward schedule (None when `gen3h` is false)"""
        return self._pairs[0] if self.gen3h else None

    @property
    def pairs_r(self):
        return self._pairs[1] if self.gen3h else None

    def on(self, device):
        key = str(device)
        if key not in self._dev:
            self._dev[key] = dict(nbr_c=self.nbr_c.to(device), gidx_c=self.gidx_c.to(device),
                                  nbr_r=self.nbr_r.to(device), gidx_r=self.gidx_r.to(device),
                                  stream_c=self.stream_c.to(device) if self.gen2 else None,
                                  stream_r=self.stream_r.to(device) if self.gen2 else None,
                                  # 1 for a real list slot, 0 for padding (the adjacency gradient must also reach
                                  # real entries whose current coefficient happens to be zero)
                                  real_r=(self.gidx_r >= 0).to(torch.float32).to(device).contiguous())
        return self._dev[key]


def _gcn_forward(x, W, nbr, coef, LkA, bias_cv, tables, want_stats=False):
    N, C, T, V = x.shape
    z = torch.empty_like(x)
    part = None
    if want_stats:      # one (sum, sum of squares) row pair per workgroup = per (sample, tile of 384 // V frames)
        frames = min(384 // V, T)
        part = torch.empty((N * ((T + frames - 1) // frames), C, 2), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().p2r_stgcn_gcn_forward(
            N, T, V, tables.K, LkA, _lib.ptr(x), _lib.ptr(W), _lib.ptr(nbr), _lib.ptr(coef),
            _lib.ptr(bias_cv), _lib.ptr(z), _lib.ptr(part), _lib.current_stream(x.device)), "stgcn_gcn_forward")
    return (z, part) if want_stats else z


def permute_planes(W3):
    """W3 [K][64 rows][64 cols] -> the A-operand order of csrc/stgcn_gcn2.hip:
    Wp[k][ph][m][16 g + r][s] = W3[k][16 m + r][16 ph + 4 s + g]  (one contiguous 1 KB per wave load)."""
    K = W3.shape[0]
    return W3.reshape(K, 4, 16, 4, 4, 4).permute(0, 3, 1, 5, 2, 4).contiguous()     # (k, ph, m, g, r, s)


USE_GEN3 = True      # statically scheduled kernel for the P2RNet skeleton (tests switch it off to reach gcn2)
SPLIT_WEIGHT_GRAD = True   # split16 mode: the weight gradient on the split kernel too (tests / A-B timing switch it off)
SPLIT_COEF_GRAD = True     # ... and the adjacency gradient


def _gen3_able(x, z, addend, tables, bwd=None):
    """Everything p2r_stgcn_gcn3_forward checks before it launches (P2R_EINVAL otherwise), so that a shape or alignment
    it does not take falls back to the second generation instead of raising: whole 16-frame tiles, 16-byte aligned
    rows of x / z / addend / the saved activation of the BatchNorm-backward epilogue, 4-byte aligned mask bytes."""
    return (USE_GEN3 and tables.gen3 and x.shape[0] > 0 and x.shape[2] % 16 == 0 and x.data_ptr() % 16 == 0
            and z.data_ptr() % 16 == 0 and (addend is None or addend.data_ptr() % 16 == 0)
            and (bwd is None or (bwd[0].data_ptr() % 16 == 0 and bwd[1].data_ptr() % 4 == 0)))


def _gcn2_forward(x, Wp, coef, stream, bias_cv, tables, want_stats=False, addend=None, bwd=None, form=None,
                  addend_mask=None):
    """bwd = (u, mask, fin): data-gradient launch whose statistics epilogue is the reduction pass of the
    BatchNorm + residual + ReLU backward of the block in front (implies want_stats; see bn_op.BNLink).
    form: 0 column lists (forward) / 1 row lists (data gradient) when known -- the statically scheduled third-generation
    kernel then takes the launch if the tables carry the pattern it was generated for; `stream` is for gcn2."""
    N, C, T, V = x.shape
    z = torch.empty_like(x)
    lib = _lib.lib()
    part = None
    ltot = coef.shape[0]
    with torch.cuda.device(x.device):
        st = _lib.current_stream(x.device)
        gen3 = form is not None and _gen3_able(x, z, addend, tables, bwd)
        if addend_mask is not None and not (gen3 and form == 1 and addend_mask.data_ptr() % 4 == 0):
            # the masked-addend form exists only in the statically scheduled data-gradient kernel: anywhere else the
            # product is formed here (one elementwise launch)
            addend = addend * (addend_mask != 0)
            addend_mask = None
        if want_stats:      # one partial per persistent workgroup: min(tiles of 16 frames, 256); forward launches of
            #                 the third generation write (count, mean, M2) entries, everything else pairs of sums
            part = torch.empty((min(N * ((T + 15) // 16), 256), C, 3 if gen3 and bwd is None else 2),
                               dtype=torch.float32, device=x.device)
        if gen3 and addend_mask is not None:
            bu, bm, bf = (bwd[0], bwd[1], bwd[2].contiguous()) if bwd is not None else (None, None, None)
            _lib.check(lib.p2r_stgcn_gcn3_data_gradient_masked_addend(
                N, T, V, tables.K, ltot, _lib.ptr(x), _lib.ptr(Wp), _lib.ptr(coef), _lib.ptr(addend), _lib.ptr(addend_mask),
                _lib.ptr(z), _lib.ptr(part), _lib.ptr(bu), _lib.ptr(bm), _lib.ptr(bf), st),
                "stgcn_gcn3_data_gradient_masked_addend")
            return (z, part) if want_stats else z
        if gen3:
            bu, bm, bf = (bwd[0], bwd[1], bwd[2].contiguous()) if bwd is not None else (None, None, None)
            _lib.check(lib.p2r_stgcn_gcn3_forward(N, T, V, tables.K, ltot, int(form), _lib.ptr(x), _lib.ptr(Wp),
                                                  _lib.ptr(coef), _lib.ptr(bias_cv), _lib.ptr(addend), _lib.ptr(z),
                                                  _lib.ptr(part), None, _lib.ptr(bu), _lib.ptr(bm), _lib.ptr(bf), st),
                       "stgcn_gcn3_forward")
            return (z, part) if want_stats else z
        work = torch.empty_like(stream)       # the stream with this call's coefficients (scalar-loaded by the kernel)
        bu, bm, bf = (bwd[0], bwd[1], bwd[2].contiguous()) if bwd is not None else (None, None, None)
        _lib.check(lib.p2r_stgcn_gcn2_forward(N, T, V, tables.K, ltot, _lib.ptr(x), _lib.ptr(Wp), _lib.ptr(coef),
                                              _lib.ptr(stream), _lib.ptr(work), _lib.ptr(bias_cv), _lib.ptr(addend),
                                              _lib.ptr(z), _lib.ptr(part), None, _lib.ptr(bu), _lib.ptr(bm),
                                              _lib.ptr(bf), st), "stgcn_gcn2_forward")
    return (z, part) if want_stats else z


class SplitPlanes(object):
    """Operands of the split16 graph conv (csrc/stgcn_gcn3h_body.h): `wh` fp16 [6 pairs][4 phases][3 parts][4][64][8] =
    the parts (w1, w2, 2^-11 w1) of 2^S [W_a | W_b] in the kernel's A-operand order, `winv` device float [1] = 2^-S."""
    __slots__ = ('wh', 'winv')

    def __init__(self, wh, winv):
        self.wh, self.winv = wh, winv


def _pairs_index(pairs, K, transposed):
    """source positions (see math_mode.pack_parts; M = K * 4096) of the plane-pair operand layout
    out[pair][ph][part][m][16 kg + r][i] = part of W_{plane (i < 4 ? a : b)}[16 m + r][16 ph + kg + 4 (i & 3)]
    (transposed: of W_k^T, i.e. element [row][col] comes from W_k[col][row])"""
    import numpy as np
    M = K * 4096
    P = len(pairs)
    pa = np.array(pairs, dtype=np.int64)                                   # (P, 2)
    sh = (P, 4, 3, 4, 4, 16, 8)                                            # pair, ph, part, m, kg, r, i
    pi, ph, part, m, kg, r, i = np.meshgrid(*[np.arange(n) for n in sh], indexing='ij')
    plane = pa[pi, i >> 2]
    row, col = 16 * m + r, 16 * ph + kg + 4 * (i & 3)
    src = part * M + plane * 4096 + (col * 64 + row if transposed else row * 64 + col)
    return np.where(plane >= 0, src, 3 * M).reshape(-1).astype(np.int64)


def split_planes(W, pairs, transposed=False):
    """W [..., K][64 rows][64 cols] fp32 (leading batch dimensions allowed: one scale per leading index), pairs = the
    schedule's plane pairs -> (wh [..., P, 4, 3 parts, 4, 64, 8] fp16, winv [..., 1] fp32):
    wh[pair][ph][part][m][16 kg + r][i] = part of 2^S W_{plane (i < 4 ? a : b)}[16 m + r][16 ph + kg + 4 (i & 3)]
    (transposed=True: of W_k^T -- the data gradient's planes -- without materialising the transposed tensor)."""
    K = W.shape[-3]
    flat, inv = math_mode.pack_parts(W, W.dim() - 3)
    wh = math_mode.gather_layout(flat, ('gcn_pairs', tuple(pairs), K, bool(transposed)),
                                 lambda: _pairs_index(pairs, K, transposed))
    return wh.view(*W.shape[:-3], len(pairs), 4, 3, 4, 64, 8), inv


def _coef_grad_index(K):
    """source positions of out[k][ph][part (w1, w2)][ks][16 kg + r][i] = W_k[16 ph + r][32 ks + 16 (i >> 2) + 4 (i & 3) + kg]"""
    import numpy as np
    M = K * 4096
    sh = (K, 4, 2, 2, 4, 16, 8)                                            # k, ph, part, ks, kg, r, i
    k, ph, part, ks, kg, r, i = np.meshgrid(*[np.arange(n) for n in sh], indexing='ij')
    row, col = 16 * ph + r, 32 * ks + 16 * (i >> 2) + 4 * (i & 3) + kg
    return (part * M + k * 4096 + row * 64 + col).reshape(-1).astype(np.int64)


def split_planes_coef_grad(W, packed=None):
    """W [..., K][64 rows][64 cols] fp32 (forward planes; one scale per leading index) -> the A operands of the split16
    adjacency-gradient kernel (csrc/stgcn_gcn3h_grad.hip): (wd [..., K, 4 ph, 2 parts, 2 ks, 64, 8] fp16, winv [..., 1]):
    wd[k][ph][part][ks][16 kg + r][i] = part of 2^S W_k[16 ph + r][32 ks + 16 (i >> 2) + 4 (i & 3) + kg].
    packed: (flat, inv) of math_mode.pack_parts(W, ...) when the caller already has it."""
    K = W.shape[-3]
    flat, inv = packed if packed is not None else math_mode.pack_parts(W, W.dim() - 3)
    wd = math_mode.gather_layout(flat, ('gcn_coef_grad', K), lambda: _coef_grad_index(K))
    return wd.view(*W.shape[:-3], K, 4, 2, 2, 64, 8), inv


def split_unit_counts(tables):
    """(forward, data gradient): (plane pair, joint) units of the split16 schedules per 16-frame tile and channel phase --
    a unit is live when either plane of the pair has a non-empty neighbour list at the joint (12 MFMAs each)."""
    import numpy as np
    out = []
    for gidx, Lk, pairs in ((tables.gidx_c, tables.Lk_c, tables.pairs_c), (tables.gidx_r, tables.Lk_r, tables.pairs_r)):
        g = gidx.numpy()
        lofs = np.concatenate([[0], np.cumsum(Lk)])
        live = np.stack([(g[lofs[k]:lofs[k + 1]] >= 0).any(0) for k in range(tables.K)])
        out.append(int(sum((live[a] | (live[b] if b >= 0 else False)).sum() for a, b in pairs)))
    return tuple(out)


def split_weight_grad_units(tables):
    """live (plane, group of 8 joints) units of the split16 weight-gradient schedule per 4-frame tile (24 MFMAs per unit and
    wave, two waves -- the column halves -- per plane set)"""
    import numpy as np
    g = tables.gidx_r.numpy()
    lofs = np.concatenate([[0], np.cumsum(tables.Lk_r)])
    live = np.stack([(g[lofs[k]:lofs[k + 1]] >= 0).any(0) for k in range(tables.K)])      # [K][V]
    return int(sum(live[k, 8 * grp:8 * grp + 8].any() for k in range(tables.K) for grp in range((tables.V + 7) // 8)))


def _gen3h_able(x, tables):
    """shapes the split16 graph-conv kernels take (anything else runs on the exact kernels in either mode)"""
    return (USE_GEN3 and tables.gen3h and x.shape[0] > 0 and x.shape[2] % 16 == 0 and x.data_ptr() % 16 == 0)


def _gcn3h_forward(x, sp, coef, bias_cv, tables, want_stats, x_word):
    N, C, T, V = x.shape
    z = torch.empty_like(x)
    part = torch.empty((min(N * (T // 16), 256), C, 3), dtype=torch.float32, device=x.device) if want_stats else None
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().p2r_stgcn_gcn3h_forward(
            N, T, V, tables.K, coef.shape[0], _lib.ptr(x), _lib.ptr(sp.wh), _lib.ptr(sp.winv), _lib.ptr(coef),
            _lib.ptr(bias_cv), _lib.ptr(z), _lib.ptr(part), None, _lib.ptr(x_word), _lib.current_stream(x.device)),
            "stgcn_gcn3h_forward")
    return (z, part) if want_stats else z


def _gcn3h_data_gradient(dz, sp, coef, tables, addend, addend_mask, dz_word):
    N, C, T, V = dz.shape
    dx = torch.empty_like(dz)
    with torch.cuda.device(dz.device):
        _lib.check(_lib.lib().p2r_stgcn_gcn3h_data_gradient(
            N, T, V, tables.K, coef.shape[0], _lib.ptr(dz), _lib.ptr(sp.wh), _lib.ptr(sp.winv), _lib.ptr(coef),
            _lib.ptr(addend), _lib.ptr(addend_mask), _lib.ptr(dx), _lib.ptr(dz_word), _lib.current_stream(dz.device)),
            "stgcn_gcn3h_data_gradient")
    return dx


class _GraphConv(Function):
    @staticmethod
    def forward(ctx, x, weight, coef_c, coef_r, bias_cv, tables, want_stats=False, with_residual=False,
                bn_link=None, wp_f=None, wp_b=None, lazy_res=None, split=None):
        # weight (K*64, 64): plane k rows = output channels of plane k
        dev = x.device
        t = tables.on(dev)
        x = x.contiguous()
        W = weight.contiguous()
        ctx.split = ctx.x_word = None
        if split is None and math_mode.split16() and _gen3h_able(x, tables):
            W3 = W.view(tables.K, 64, 64)
            split = (SplitPlanes(*split_planes(W3, tables.pairs_c)),
                     SplitPlanes(*split_planes(W3, tables.pairs_r, transposed=True)),
                     SplitPlanes(*split_planes_coef_grad(W3)))
        if split is not None and _gen3h_able(x, tables):
            # split16 mode: two-part fp16 products (csrc/stgcn_gcn3h_body.h); x's range word stays with the op -- x is an
            # operand of the weight- and adjacency-gradient kernels again
            ctx.split = split
            ctx.x_word = math_mode.range_word(x)
            out = _gcn3h_forward(x, split[0], coef_c.contiguous(), bias_cv.contiguous(), tables, want_stats, ctx.x_word)
        elif tables.gen2:         # second-generation kernel (csrc/stgcn_gcn2.hip)
            out = _gcn2_forward(x, wp_f if wp_f is not None else permute_planes(W.view(tables.K, 64, 64)),
                                coef_c.contiguous(), t['stream_c'], bias_cv.contiguous(), tables, want_stats, form=0)
        else:
            out = _gcn_forward(x, W, t['nbr_c'], coef_c.contiguous(), tables.LkA_c, bias_cv.contiguous(), tables,
                               want_stats)
        ctx.save_for_backward(x, W, coef_c, coef_r)
        ctx.tables = tables
        ctx.bn_link = bn_link
        ctx.wp_b = wp_b            # planes of the data gradient, already in kernel order (prepare_chain), or None
        ctx.wp_f = wp_f            # forward planes in kernel order: the adjacency-gradient kernel multiplies by them
        ctx.n_out = 2 if want_stats else 1
        ctx.lazy_res = lazy_res if with_residual else None  # bn_op.ResLink: the identity branch's gradient arrives unmasked
        if want_stats:
            ctx.mark_non_differentiable(out[1])
        if with_residual:
            # the block's identity branch leaves through this op too, so that the backward sees the residual
            # gradient next to dz and adds it inside the data-gradient kernel (no separate accumulation pass)
            out = (out if want_stats else (out,)) + (x.view_as(x),)
        return out

    @staticmethod
    def backward(ctx, dz, *rest):
        x, W, coef_c, coef_r = ctx.saved_tensors
        dres = rest[ctx.n_out - 1] if len(rest) >= ctx.n_out else None     # gradient of the identity branch, if any
        dres_mask = None
        if ctx.lazy_res is not None and dres is not None:
            dres = dres.contiguous()
            dres_mask = ctx.lazy_res.take(dres)
            if dres_mask is None:
                raise RuntimeError("graph_conv: the identity branch's gradient was announced unmasked (lazy_res) but no "
                                   "mask is registered for it -- the residual has another consumer than this op")
        tables = ctx.tables
        dev = x.device
        t = tables.on(dev)
        dz = dz.contiguous()
        N, C, T, V = x.shape
        K = tables.K
        dx = dW = dcoef_r = dbias = None
        # split16 mode: dz is an operand of up to three split kernels below -- its range word is looked up once
        dz_word = math_mode.range_word(dz) if ctx.split is not None else None
        if ctx.needs_input_grad[0]:
            # dX = sum_k W_k^T (dZ . A_k^T): forward kernel with transposed planes + row lists
            if ctx.split is not None:
                link = ctx.bn_link
                use_link = link is not None and link.intact() and link.u.shape == x.shape
                # (the sums epilogue of the exact kernel does not exist here: the BatchNorm backward in front runs its
                # own reduction pass -- on the side stream under the gradient kernels below when the overlap is on)
                ad = dres.contiguous() if dres is not None else None
                if dres_mask is not None and dres_mask.data_ptr() % 4 != 0:
                    ad, dres_mask = ad * (dres_mask != 0), None
                dx = _gcn3h_data_gradient(dz, ctx.split[1], coef_r.contiguous(), tables, ad, dres_mask, dz_word)
                if use_link:
                    link.partials = None
                    link.grad_ptr, link.grad_version = dx.data_ptr(), dx._version
                    link.ready = torch.cuda.Event()
                    link.ready.record(torch.cuda.current_stream(dev))
                dres = None
            elif tables.gen2:
                link = ctx.bn_link
                use_link = link is not None and link.intact() and link.u.shape == x.shape
                # the sums either leave through this kernel's epilogue, or -- when the BatchNorm backward will run its
                # passes on a side stream under the gradient kernels launched below -- are left to its own reduction pass
                # (HBM-bound, hidden there, while the epilogue form costs 0.18 ms of this kernel with the matrix pipe idle)
                emit = use_link and not (bn_op.OVERLAP_APPLY and bn_op.OVERLAP_REDUCE and bool(ctx.needs_input_grad[1])
                                         and bool(ctx.needs_input_grad[3]))
                wp_b = ctx.wp_b if ctx.wp_b is not None else permute_planes(W.view(K, C, C).transpose(1, 2))
                dx = _gcn2_forward(dz, wp_b, coef_r.contiguous(),
                                   t['stream_r'], None, tables, addend=dres.contiguous() if dres is not None else None,
                                   want_stats=emit, bwd=(link.u, link.mask, link.fin) if emit else None,
                                   form=1, addend_mask=dres_mask)
                if use_link:
                    # dx is the whole gradient of the previous block's output: its BatchNorm backward takes the
                    # two per-channel sums from here instead of a pass over dx and its saved input
                    if emit:
                        dx, link.partials = dx
                    else:
                        link.partials = None
                    link.grad_ptr, link.grad_version = dx.data_ptr(), dx._version
                    # dx and the sums are complete HERE; the weight- and adjacency-gradient launches below do not
                    # touch them (bn_op._FusedBNAct.backward runs its apply pass under them, on a side stream)
                    link.ready = torch.cuda.Event()
                    link.ready.record(torch.cuda.current_stream(dev))
                dres = None
            else:
                if dres_mask is not None:
                    dres, dres_mask = dres * (dres_mask != 0), None
                Wt = W.view(K, C, C).transpose(1, 2).contiguous()            # [k][ci][c]
                dx = _gcn_forward(dz, Wt, t['nbr_r'], coef_r.contiguous(), tables.LkA_r, None, tables)
        lib = _lib.lib()
        st = _lib.current_stream(dev)
        with torch.cuda.device(dev):
            if ctx.needs_input_grad[1]:
                # dW_k[c][ci] = sum_cols dz[c] * (x . A_k)[ci] = sum_cols (dz . A_k^T)[c] * x[ci]: the second form
                # aggregates on the gradient side through the ROW lists, whose (plane, 4-joint group) units are
                # empty ~30 % of the time (skipped); the kernel then returns dW_k transposed
                part = torch.empty((_N_BLOCKS, K, C, C), dtype=torch.float32, device=dev)
                # the bias-table gradient (column sums of dz) rides on the same pass over dz
                bpart = torch.empty((_N_BLOCKS, C, V), dtype=torch.float32, device=dev) if ctx.needs_input_grad[4] else None
                if ctx.split is not None and SPLIT_WEIGHT_GRAD and dz.data_ptr() % 16 == 0:
                    # split16 mode (csrc/stgcn_gcn3dwh.hip): both operands are runtime tensors, each with its range word
                    _lib.check(lib.p2r_stgcn_gcn3h_weight_grad(
                        N, T, V, K, coef_r.shape[0], _lib.ptr(x), _lib.ptr(dz), _lib.ptr(coef_r.contiguous()), _N_BLOCKS,
                        _lib.ptr(part), _lib.ptr(bpart), _lib.ptr(ctx.x_word), _lib.ptr(dz_word), st),
                        "stgcn_gcn3h_weight_grad")
                elif (USE_GEN3 and tables.gen3 and N > 0 and T % 4 == 0 and x.data_ptr() % 16 == 0
                        and dz.data_ptr() % 16 == 0):      # (N == 0: the first-generation kernel returns zeros)
                    # statically scheduled kernel (csrc/stgcn_gcn3_dw.hip)
                    _lib.check(lib.p2r_stgcn_gcn3_weight_grad(
                        N, T, V, K, coef_r.shape[0], _lib.ptr(x), _lib.ptr(dz), _lib.ptr(coef_r.contiguous()), _N_BLOCKS,
                        _lib.ptr(part), _lib.ptr(bpart), st), "stgcn_gcn3_weight_grad")
                else:
                    _lib.check(lib.p2r_stgcn_gcn_weight_grad(
                        N, T, V, K, tables.LkA_r, _lib.ptr(dz), _lib.ptr(x), _lib.ptr(t['nbr_r']),
                        _lib.ptr(coef_r.contiguous()), _N_BLOCKS, _lib.ptr(part), _lib.ptr(bpart), 1, st),
                        "stgcn_gcn_weight_grad")
                dW = _lib.sum_leading(part, tr64=True).reshape(K * C, C)      # the kernel returns dW_k^T
                if bpart is not None:
                    dbias = _lib.sum_leading(bpart)                        # (C, V)
            if ctx.needs_input_grad[3]:
                # adjacency gradient in ROW-list form: Y_k = W_k . x on MFMA, reduced against dz gathered through
                # the row lists (same kernel as the column form with the roles of x and dz swapped); the row lists
                # leave ~20 % of the (plane, 16-column) units empty, which the kernel skips
                ltot = coef_r.shape[0]
                part = torch.empty((_N_BLOCKS, ltot, V), dtype=torch.float32, device=dev)
                if (ctx.split is not None and SPLIT_COEF_GRAD and T % 16 == 0 and dz.data_ptr() % 16 == 0
                        and x.data_ptr() % 16 == 0):
                    # split16 mode (csrc/stgcn_gcn3h_grad.hip): Y_k = W_k x on two-part fp16 operands; dz stays fp32
                    sp = ctx.split[2]
                    _lib.check(lib.p2r_stgcn_gcn3h_coef_grad(N, T, V, K, ltot, _lib.ptr(x), _lib.ptr(dz), _lib.ptr(sp.wh),
                                                             _lib.ptr(sp.winv), _N_BLOCKS, _lib.ptr(part),
                                                             _lib.ptr(ctx.x_word), st), "stgcn_gcn3h_coef_grad")
                elif USE_GEN3 and tables.gen3 and T % 16 == 0 and dz.data_ptr() % 16 == 0:
                    # statically scheduled kernel (csrc/stgcn_gcn3_grad.hip)
                    wp_f = ctx.wp_f if ctx.wp_f is not None else permute_planes(W.view(K, C, C))
                    _lib.check(lib.p2r_stgcn_gcn3_coef_grad(N, T, V, K, ltot, _lib.ptr(x), _lib.ptr(dz), _lib.ptr(wp_f),
                                                            _N_BLOCKS, _lib.ptr(part), st), "stgcn_gcn3_coef_grad")
                else:
                    _lib.check(lib.p2r_stgcn_gcn_coef_grad(
                        N, T, V, K, tables.LkA_r, _lib.ptr(dz), _lib.ptr(x), _lib.ptr(W), _lib.ptr(t['nbr_r']),
                        _lib.ptr(t['real_r']), _N_BLOCKS, _lib.ptr(part), st), "stgcn_gcn_coef_grad")
                dcoef_r = _lib.sum_leading(part)
        if ctx.needs_input_grad[4] and dbias is None:
            part = torch.empty((N * C, V), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _lib.check(lib.p2r_colsum(N * C, T, V, _lib.ptr(dz), _lib.ptr(part), st), "colsum")
            dbias = part.view(N, C, V).sum(0)                          # (C, V)
        if dres is not None:
            if dres_mask is not None:
                dres = dres * (dres_mask != 0)
            dx = dres if dx is None else dx + dres
        return dx, dW, None, dcoef_r, dbias, None, None, None, None, None, None, None, None


def grad_kernel_mfma_flops(tables, batch, frames):
    """MFMA FLOPs the weight- and adjacency-gradient kernels issue per launch (for the roofline accounting of
    bench.py; one v_mfma_f32_16x16x4_f32 = 2048 FLOP).  Kept next to the launches so that it changes with them.

    gcn_dw_kernel: 4-frame tiles; a (plane, group of four joints) unit is live when any of its joints has a non-empty
    row list, and then costs 16 MFMAs in each of the four ci-column waves.
    gcn_dcoef_kernel: tiles of F = 384 // V frames, joint-major n-tiles of 16 columns; a (plane, n-tile) unit is live
    when any of its columns' joints has a non-empty row list: 16 k-steps x 4 m-tiles."""
    import numpy as np
    K, V = tables.K, tables.V
    gidx = tables.gidx_r.numpy()
    lofs = np.concatenate([[0], np.cumsum(tables.Lk_r)])
    live = np.stack([(gidx[lofs[k]:lofs[k + 1]] >= 0).any(0) for k in range(K)])        # [K][V]
    groups = sum(int(live[k, 4 * g:4 * g + 4].any()) for k in range(K) for g in range((V + 3) // 4))
    dw = groups * 64 * 2048.0 * batch * ((frames + 3) // 4)
    F = min(384 // V, frames)
    units = 0
    for t in range(24):
        joints = sorted({c // F for c in range(16 * t, 16 * t + 16) if c < F * V})
        units += sum(int(live[k, joints].any()) for k in range(K)) if joints else 0
    dc = units * 64 * 2048.0 * batch * ((frames + F - 1) // F)
    if USE_GEN3 and tables.gen3 and frames % 16 == 0:
        # gcn3_dcoef_kernel: one 16-MFMA product per live (plane, joint) unit of the row lists, 16-frame tile and
        # output-row phase (the rest of a long list reuses the product)
        dc = int(live.sum()) * 4 * 16 * 2048.0 * batch * (frames // 16)
    if USE_GEN3 and tables.gen3 and frames % 4 == 0:
        # gcn3_dw_kernel: 16 MFMAs (4 m x 4 n tiles, one k-step) per live (plane, joint) unit and 4-frame tile
        dw = int(live.sum()) * 16 * 2048.0 * batch * (frames // 4)
    return {'gcn_weight_grad': dw, 'gcn_coef_grad': dc}


def supported(x, weight, A):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 64
            and weight.shape[1] == 64 and weight.shape[0] == 64 * A.shape[0] and A.shape[0] == 11
            and A.shape[1] <= 56)      # the weight-gradient kernel stages 4 frames x V <= 226 columns per row


def graph_conv(x, weight, bias, Aeff, tables, want_stats=False, with_residual=False, bn_link=None, prepared=None,
               lazy_res=None):
    """x (N,64,T,V); weight (K*64,64[,1,1]); bias (K*64) or None; Aeff (K,V,V).
    with_residual: additionally return x itself (last output) for the caller's identity branch; its gradient is then
    added inside the data-gradient kernel.
    want_stats: also return the kernel's per-workgroup (sum, sum of squares) partials of z per channel
    ([P,64,2], see bn_op.moments) -- the batch statistics of the BatchNorm that consumes z.
    bn_link: the bn_op.BNLink of the fused BatchNorm + residual + ReLU that produced x, when every use of x goes
    through this call (x and, with_residual, the identity branch): the data-gradient kernel then also emits the
    reduction pass of that BatchNorm's backward.
    prepared: this block's `BlockParams` from `prepare_chain` (coefficient tables, bias table and kernel-order
    planes computed for all blocks at once); Aeff is then not looked at.
    lazy_res (with_residual): a `bn_op.ResLink` shared with the `bn_op.fused_bn_act(..., lazy_res=link)` the identity
    branch goes into, whose backward hands its gradient over unmasked; this op's data-gradient kernel applies the mask
    while it adds."""
    K, V = tables.K, tables.V
    t = tables.on(x.device)
    w2 = weight.reshape(K * 64, 64)
    if prepared is not None:
        return _GraphConv.apply(x, w2, prepared.coef_c, prepared.coef_r, prepared.bias_cv, tables, want_stats,
                                with_residual, bn_link, prepared.gcn_wp_f, prepared.gcn_wp_b, lazy_res, prepared.gcn_split)
    coef_c = gcn_tables.coefficients(Aeff.detach(), t['gidx_c'])      # forward lists (values only)
    coef_r = gcn_tables.coefficients(Aeff, t['gidx_r'])               # backward lists; carries the gradient to Aeff
    if bias is not None:
        bias_cv = bias.view(K, 64).t() @ Aeff.sum(dim=1)               # (64,V) = sum_k b_k (x) colsum_k
    else:
        bias_cv = torch.zeros(64, V, dtype=x.dtype, device=x.device)
    return _GraphConv.apply(x, w2, coef_c, coef_r, bias_cv, tables, want_stats, with_residual, bn_link, None, None,
                            lazy_res)


class BlockParams(object):
    """One st_gcn_block's share of `prepare_chain`."""
    __slots__ = ('Aeff', 'coef_c', 'coef_r', 'bias_cv', 'gcn_wp_f', 'gcn_wp_b', 'tcn_wp_f', 'tcn_wp_b', 'gcn_split')


def prepare_chain(blocks, A, importances, tables, frames=None):
    """The small per-block parameter transforms of the fused path, done for ALL blocks of an ST-GCN stack at once:
    `A * edge_importance`, the two coefficient tables, the bias table, and the kernel-order copies of the graph-conv
    planes and temporal-conv taps (forward and data-gradient forms).  Per block these are ~25 launches of a few
    microseconds each way; batched they are ~20 for the whole stack.  Same arithmetic, same autograd graph up to
    stack / unbind nodes.  Requires every block to be 64 -> 64 with K planes of 64 x 64 and a (3,1) temporal conv."""
    B = len(blocks)
    K, V = tables.K, tables.V
    dev = A.device
    t = tables.on(dev)
    Aeff = A.unsqueeze(0) * torch.stack(list(importances))                      # (B,K,V,V)
    flat = Aeff.reshape(B, -1)

    def coef(flat_, gidx):
        safe = gidx.clamp(min=0).reshape(-1)
        vals = flat_.index_select(1, safe).view(B, *gidx.shape)
        return torch.where(gidx >= 0, vals, torch.zeros((), dtype=flat_.dtype, device=dev))

    coef_c = coef(flat.detach(), t['gidx_c'])                                   # values only
    coef_r = coef(flat, t['gidx_r'])                                            # carries the gradient to Aeff
    biases = [b.gcn.conv.bias for b in blocks]
    if all(b is not None for b in biases):
        bias_cv = torch.bmm(torch.stack(biases).view(B, K, 64).transpose(1, 2), Aeff.sum(dim=2))   # (B,64,V)
    else:
        assert all(b is None for b in biases)
        bias_cv = torch.zeros(B, 64, V, dtype=A.dtype, device=dev)
    with torch.no_grad():
        W = torch.stack([b.gcn.conv.weight.view(K, 64, 64) for b in blocks])   # (B,K,64 rows,64 cols)
        # forward: Wp[k][ph][m][16g+r][s] = W_k[16m+r][16ph+4s+g];  data gradient: the same of W_k^T
        gcn_f = W.view(B, K, 4, 16, 4, 4, 4).permute(0, 1, 4, 2, 6, 3, 5).contiguous()     # (b,k,ph,m,g,r,s)
        gcn_b = W.view(B, K, 4, 4, 4, 4, 16).permute(0, 1, 2, 5, 4, 6, 3).contiguous()     # rows = (ph,s,g), cols = (m,r)
        Wt = torch.stack([b.tcn[2].weight.view(64, 64, 3) for b in blocks])     # (B, c, ci, tap)
        tcn_f = Wt.view(B, 4, 16, 4, 4, 4, 3).permute(0, 6, 3, 1, 5, 2, 4).contiguous()    # (b,tap,ph,m,g,r,s)
        # data gradient: tap p' uses W[2 - p']^T
        tcn_b = Wt.flip(-1).view(B, 4, 4, 4, 4, 16, 3).permute(0, 6, 1, 4, 3, 5, 2).contiguous()   # rows = (ph,s,g)
        # split16 mode (math_mode; whole 16-frame tiles of the 53-joint skeleton only): the weights once per step as
        # two-part fp16 operands in the split kernels' lane order, one power-of-two scale per block
        split = (math_mode.split16() and tables.gen3h and V == 53 and frames is not None and frames % 16 == 0)
        if split:
            from . import tconv_op
            math_mode.reset()
            # one (scale, three fp16 planes) per weight tensor, then one gather per operand layout: a dozen launches
            gsf, gsf_inv = split_planes(W, tables.pairs_c)
            gsb, gsb_inv = split_planes(W, tables.pairs_r, transposed=True)
            gsd, gsd_inv = split_planes_coef_grad(W)
            tsf, tsf_inv = tconv_op.split_taps(Wt, layout='cit')                # Wt (B, c, ci, tap): the Conv2d weight itself
            tsb, tsb_inv = tconv_op.split_taps(Wt, layout='cit', gradient=True)  # data gradient: tap p' = W[2 - p']^T
    out = []
    A_b, cc, cr, bc = Aeff.unbind(0), coef_c.unbind(0), coef_r.unbind(0), bias_cv.unbind(0)
    for i in range(B):
        p = BlockParams()
        p.Aeff, p.coef_c, p.coef_r, p.bias_cv = A_b[i], cc[i], cr[i], bc[i]
        p.gcn_wp_f, p.gcn_wp_b, p.tcn_wp_f, p.tcn_wp_b = gcn_f[i], gcn_b[i], tcn_f[i], tcn_b[i]
        p.gcn_split = None
        if split:
            p.gcn_split = (SplitPlanes(gsf[i], gsf_inv[i]), SplitPlanes(gsb[i], gsb_inv[i]), SplitPlanes(gsd[i], gsd_inv[i]))
            p.tcn_wp_f, p.tcn_wp_b = tconv_op.SplitTaps(tsf[i], tsf_inv[i]), tconv_op.SplitTaps(tsb[i], tsb_inv[i])
        out.append(p)
    return out
