"""BatchNorm -> ReLU -> temporal (3,1) / pointwise convolution as one fused op (csrc/stgcn_tconv.hip).

`bn_relu_tconv(z, bn, conv)` equals `conv(relu(bn(z)))` for the `tcn.0 / tcn.1 / tcn.2` stage of
the reference's st_gcn_block (stgcn_layers.py:399-411) with a 64->64 (3,1) conv, stride 1,
padding (1,0).  The normalised activation is produced while the kernel stages its LDS tile
and never written to HBM; the backward recomputes the ReLU mask from z.  Differentiable
w.r.t. z, bn.weight, bn.bias, conv.weight, conv.bias; updates the BatchNorm running
statistics in training mode like nn.BatchNorm2d.

The same op with a single tap serves the `cbr -> cbr -> c` embedding MLPs (stgcn.py:46-63): the
BatchNorm + ReLU of stage i is folded into the pointwise 64->64 Conv1d of stage i+1
(`bn_relu_pointwise`), and their 3->64 first layer has its own streaming kernel (`embed3`).
"""
import ctypes

import torch
from torch.autograd import Function

from .. import _lib
from . import bn_op, math_mode

_N_BLOCKS = 256
FUSE_DZ = True       # BatchNorm-backward apply inside the weight-gradient launch (53 joints, 3 taps); tests switch it off
USE_GEN3 = True      # statically scheduled kernel (csrc/stgcn_tconv3.hip); tests switch it off to reach tconv2


def _permute_taps(W3):
    """W3 [taps][64 rows][64 cols] -> the A-operand order of csrc/stgcn_tconv2.hip (same as gcn_op.permute_planes)."""
    K = W3.shape[0]
    return W3.reshape(K, 4, 16, 4, 4, 4).permute(0, 3, 1, 5, 2, 4).contiguous()


def _tconv2_able(W3, V):
    return W3.shape[0] in (1, 3) and V == 53


class SplitTaps(object):
    """Operands of the split16 temporal conv (csrc/stgcn_tconvh.hip): `wh` fp16 [3 parts][3 taps][2][4][64][8] = the parts
    (w1, w2, 2^-11 w1) of 2^S W in the kernel's A-operand order, `winv` device float [1] = 2^-S."""
    __slots__ = ('wh', 'winv')

    def __init__(self, wh, winv):
        self.wh, self.winv = wh, winv


def _taps_index(layout, gradient):
    """source positions (math_mode.pack_parts; M = 3 * 4096) of out[part][tap][ks][w][16 kg + r][i] = part of
    A[tap][16 w + r][32 ks + 8 kg + i], A = W3 (forward) or the data gradient's taps A[p'] = W3[2 - p']^T;
    layout 'tci': the source is W3 [tap][co][ci]; 'cit': the Conv2d weight [co][ci][tap]"""
    import numpy as np
    M = 3 * 4096
    sh = (3, 3, 2, 4, 4, 16, 8)                                            # part, tap, ks, w, kg, r, i
    part, tap, ks, w, kg, r, i = np.meshgrid(*[np.arange(n) for n in sh], indexing='ij')
    arow, acol = 16 * w + r, 32 * ks + 8 * kg + i
    t, co, ci = (2 - tap, acol, arow) if gradient else (tap, arow, acol)
    pos = (t * 64 + co) * 64 + ci if layout == 'tci' else (co * 64 + ci) * 3 + t
    return (part * M + pos).reshape(-1).astype(np.int64)


def split_taps(W, layout='tci', gradient=False):
    """The (3,1) convolution's weights as operands of the split16 kernel.  W fp32 with leading batch dimensions allowed
    (one scale per leading index): layout 'tci' = [..., 3 taps][64 co][64 ci], 'cit' = [..., 64 co][64 ci][3 taps] (the
    Conv2d weight).  gradient: the data gradient's taps, tap p' = W[2 - p']^T.  ->
    (wh [..., 3 parts, 3, 2, 4, 64, 8] fp16, winv [..., 1] fp32), parts = (w1, w2, 2^-11 w1) (math_mode.split_parts):
    wh[part][tap][ks][w][16 kg + r][i] = part of 2^S A[tap][16 w + r][32 ks + 8 kg + i]."""
    flat, inv = math_mode.pack_parts(W, W.dim() - 3)
    wh = math_mode.gather_layout(flat, ('tconv_taps', layout, bool(gradient)), lambda: _taps_index(layout, gradient))
    return wh.view(*W.shape[:-3], 3, 3, 2, 4, 64, 8), inv


def _tconvh_able(taps, x):
    """shapes the split16 kernel takes (anything else runs on the exact kernels in either mode)"""
    return taps == 3 and x.shape[1] == 64 and x.shape[3] == 53 and x.shape[2] % 16 == 0 and x.shape[0] > 0


def _tconvh(x, scale, shift, st, bias, want_stats=False, bwd=None, x_word=None):
    """`_tconv` on the split16 kernel; st = SplitTaps; x_word = range word of x (data gradient) or None (forward)."""
    N, C, T, V = x.shape
    out = torch.empty_like(x)
    chunk = 64 if T % 64 == 0 else (32 if T % 32 == 0 else 16)
    part = None
    if want_stats:      # one partial per workgroup = per (sample, chunk of frames)
        part = torch.empty((N * (T // chunk), C, 3 if bwd is None else 2), dtype=torch.float32, device=x.device)
    bz, bfin = (bwd[0], bwd[1].contiguous()) if bwd is not None else (None, None)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().p2r_stgcn_tconvh_forward(
            N, T, V, _lib.ptr(x), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(st.wh), _lib.ptr(st.winv), _lib.ptr(bias),
            _lib.ptr(out), _lib.ptr(part), None, _lib.ptr(bz), _lib.ptr(bfin), _lib.ptr(x_word),
            _lib.current_stream(x.device)), "stgcn_tconvh_forward")
    return (out, part) if want_stats else out


def _tconv(x, scale, shift, W3, bias, want_stats=False, bwd=None, Wp=None):
    """bwd = (z, fin): data-gradient launch whose statistics epilogue is the reduction pass of the BatchNorm + ReLU
    backward in front (second-generation kernel only; implies want_stats).  Wp: W3 already in the kernel's
    A-operand order (`_permute_taps`), W3 may then be None."""
    N, C, T, V = x.shape
    out = torch.empty_like(x)
    lib = _lib.lib()
    part = None
    if Wp is not None or _tconv2_able(W3, V):          # second-generation kernel
        if Wp is None:
            Wp = _permute_taps(W3)
        bz, bfin = (bwd[0], bwd[1].contiguous()) if bwd is not None else (None, None)
        with torch.cuda.device(x.device):
            st = _lib.current_stream(x.device)
            # full tiles of aligned rows: the statically scheduled third generation; anything else the second
            gen3 = (USE_GEN3 and T % 16 == 0 and x.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0
                    and (bz is None or bz.data_ptr() % 16 == 0))
            if want_stats:      # one partial per persistent workgroup: min(tiles of 16 frames, 256); forward launches
                #                 of the third generation write (count, mean, M2) entries, everything else pairs of sums
                part = torch.empty((min(N * ((T + 15) // 16), 256), C, 3 if gen3 and bwd is None else 2),
                                   dtype=torch.float32, device=x.device)
            fn = lib.p2r_stgcn_tconv3_forward if gen3 else lib.p2r_stgcn_tconv2_forward
            _lib.check(fn(N, T, V, Wp.shape[0], _lib.ptr(x), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(Wp),
                          _lib.ptr(bias), _lib.ptr(out), _lib.ptr(part), None, _lib.ptr(bz), _lib.ptr(bfin), st),
                       "stgcn_tconv3_forward" if gen3 else "stgcn_tconv2_forward")
        return (out, part) if want_stats else out
    assert bwd is None
    with torch.cuda.device(x.device):
        st = _lib.current_stream(x.device)
        if want_stats:      # one (sum, sum of squares) partial per persistent workgroup: ask how many
            n = ctypes.c_int(0)
            _lib.check(lib.p2r_stgcn_tconv_forward(N, T, V, W3.shape[0], None, None, None, None, None, None, None,
                                                   ctypes.byref(n), st), "stgcn_tconv_forward(size)")
            part = torch.empty((n.value, C, 2), dtype=torch.float32, device=x.device)
        _lib.check(lib.p2r_stgcn_tconv_forward(N, T, V, W3.shape[0], _lib.ptr(x), _lib.ptr(scale), _lib.ptr(shift),
                                               _lib.ptr(W3), _lib.ptr(bias), _lib.ptr(out), _lib.ptr(part), None, st),
                   "stgcn_tconv_forward")
    return (out, part) if want_stats else out


class _BNReLUTConv(Function):
    """`fin` [4, 64] = (mean, invstd, scale, shift) of the BatchNorm in front (bn_op.finalize, or the eval-mode
    constants)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, fin, weight, bias, train, want_stats=False, wp_f=None, wp_b=None, add_ct=None):
        """wp_f / wp_b: the taps in kernel order for the forward and the data-gradient launch (gcn_op.prepare_chain);
        only for the (3,1) conv over 53 joints.  add_ct (N,64,T): addend broadcast over the joints (single tap only)."""
        z = z.contiguous()
        taps = weight.numel() // (64 * 64)
        ctx.has_add = add_ct is not None
        if add_ct is not None:
            assert taps == 1 and not want_stats and wp_f is None
            W3 = weight.reshape(64, 64, 1).permute(2, 0, 1).contiguous()
            N, C, T, V = z.shape
            add_c = add_ct.contiguous()
            if (USE_GEN3 and V == 53 and T % 16 == 0 and z.data_ptr() % 16 == 0 and add_c.shape == (N, C, T)
                    and add_c.dtype == torch.float32):
                out = torch.empty_like(z)
                with torch.cuda.device(z.device):
                    _lib.check(_lib.lib().p2r_stgcn_tconv3_forward_add(
                        N, T, V, _lib.ptr(z), _lib.ptr(fin[2]), _lib.ptr(fin[3]), _lib.ptr(_permute_taps(W3)),
                        _lib.ptr(bias.contiguous() if bias is not None else None), _lib.ptr(add_c), _lib.ptr(out),
                        _lib.current_stream(z.device)), "stgcn_tconv3_forward_add")
            else:
                out = _tconv(z, fin[2], fin[3], W3, bias.contiguous() if bias is not None else None) + add_c.unsqueeze(-1)
            ctx.save_for_backward(z, fin, W3)
        elif wp_f is not None:
            assert taps == 3 and z.shape[3] == 53 and wp_b is not None
            W3 = None
            if isinstance(wp_f, SplitTaps):        # split16 mode (gcn_op.prepare_chain handed over fp16 operand parts)
                out = _tconvh(z, fin[2], fin[3], wp_f, bias.contiguous() if bias is not None else None, want_stats)
            else:
                out = _tconv(z, fin[2], fin[3], None, bias.contiguous() if bias is not None else None, want_stats, Wp=wp_f)
            ctx.save_for_backward(z, fin)
        elif math_mode.split16() and _tconvh_able(taps, z):
            W3 = weight.reshape(64, 64, taps).permute(2, 0, 1).contiguous()         # [tap][c][ci]
            out = _tconvh(z, fin[2], fin[3], SplitTaps(*split_taps(W3)), bias.contiguous() if bias is not None else None,
                          want_stats)
            wp_b = SplitTaps(*split_taps(W3, gradient=True))                         # data gradient: tap p' = W[2 - p']^T
            ctx.save_for_backward(z, fin)
        else:
            W3 = weight.reshape(64, 64, taps).permute(2, 0, 1).contiguous()         # [tap][c][ci]
            out = _tconv(z, fin[2], fin[3], W3, bias.contiguous() if bias is not None else None, want_stats)
            ctx.save_for_backward(z, fin, W3)
        ctx.wp_b = wp_b
        ctx.taps = taps
        ctx.train = train
        ctx.has_bias = bias is not None
        ctx.wshape = weight.shape
        if want_stats:
            ctx.mark_non_differentiable(out[1])
        return out

    @staticmethod
    def backward(ctx, du, _dstats=None):
        if ctx.wp_b is not None:
            (z, fin), W3 = ctx.saved_tensors, None
        else:
            z, fin, W3 = ctx.saved_tensors
        mean, invstd, scale, shift = fin[0], fin[1], fin[2], fin[3]
        du = du.contiguous()
        N, C, T, V = z.shape
        L = T * V
        dev = z.device
        lib = _lib.lib()
        st = _lib.current_stream(dev)
        # dh[ci, t] = sum_p W[p][c][ci] du[c, t - (p-1)]  ->  same kernel, taps reversed + transposed
        wp_b = ctx.wp_b
        W3T = W3.flip(0).transpose(1, 2).contiguous() if wp_b is None else None
        need_sums = ctx.train or ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        part = None
        split = isinstance(wp_b, SplitTaps)
        if split:
            # split16: the incoming gradient is lifted into fp16's range by a power of two from its range word
            word = math_mode.range_word(du)
            if need_sums:
                dh, part = _tconvh(du, None, None, wp_b, None, want_stats=True, bwd=(z, fin), x_word=word)
            else:
                dh = _tconvh(du, None, None, wp_b, None, x_word=word)
        elif need_sums and (wp_b is not None or _tconv2_able(W3T, V)):
            # the reduction pass of the BatchNorm backward leaves through the data-gradient kernel's epilogue
            dh, part = _tconv(du, None, None, W3T, None, want_stats=True, bwd=(z, fin), Wp=wp_b)
        else:
            dh = _tconv(du, None, None, W3T, None, Wp=wp_b)
        dz = dgamma = dbeta = dW = dbias = None
        with torch.cuda.device(dev):
            if need_sums and part is None:
                part = torch.empty((N, C, 2), dtype=torch.float32, device=dev)
                _lib.check(lib.p2r_bn_bwd_reduce(N, C, L, _lib.ptr(dh), None, _lib.ptr(z), _lib.ptr(mean),
                                                 _lib.ptr(invstd), 2, _lib.ptr(scale), _lib.ptr(shift),
                                                 _lib.ptr(part), st), "bn_bwd_reduce")
            if ctx.train:
                tot = bn_op.bwd_finalize(part, N * L)              # (dbeta, dgamma, m1, m2)
                dbeta, dgamma, m1, m2, m12 = tot[0], tot[1], tot[2], tot[3], tot[2:]
            else:   # eval: statistics are constants, dz = scale * g
                m12 = torch.zeros((2, C), device=dev)
                m1, m2 = m12[0], m12[1]
                if need_sums:
                    # affine gradients as nn.BatchNorm2d gives them in eval mode: with mean / invstd = the running
                    # statistics the same reduction yields (sum g, sum g * xhat) = (dbeta, dgamma)
                    tot = bn_op.bwd_finalize(part, N * L)
                    dbeta, dgamma = tot[0], tot[1]
            dz = torch.empty_like(z)
            taps = ctx.taps
            part = torch.empty((_N_BLOCKS, 64, 64, taps), dtype=torch.float32, device=dev)
            bpart = torch.empty((_N_BLOCKS, 64), dtype=torch.float32, device=dev) if ctx.has_bias else None
            if FUSE_DZ and taps == 3 and V == 53 and split:
                # ... and, in split16 mode, leaves the range word of dz for the graph conv's gradient kernels
                word = math_mode.new_word(dev)
                _lib.check(lib.p2r_stgcn_tconv_weight_grad_dz_amax(
                    N, T, V, taps, _lib.ptr(z), _lib.ptr(fin), _lib.ptr(du), _lib.ptr(dh), _lib.ptr(m12), _lib.ptr(dz),
                    _N_BLOCKS, _lib.ptr(part), _lib.ptr(bpart), _lib.ptr(word), st), "stgcn_tconv_weight_grad_dz_amax")
                math_mode.announce(dz, word)
            elif FUSE_DZ and taps == 3 and V == 53:
                # the BatchNorm-backward apply pass rides on the weight-gradient kernel's tile staging (it holds z)
                _lib.check(lib.p2r_stgcn_tconv_weight_grad_dz(N, T, V, taps, _lib.ptr(z), _lib.ptr(fin), _lib.ptr(du),
                                                              _lib.ptr(dh), _lib.ptr(m12), _lib.ptr(dz), _N_BLOCKS,
                                                              _lib.ptr(part), _lib.ptr(bpart), st),
                           "stgcn_tconv_weight_grad_dz")
            else:
                _lib.check(lib.p2r_bn_bwd_apply(N, C, L, _lib.ptr(dh), None, _lib.ptr(z), _lib.ptr(mean),
                                                _lib.ptr(invstd), _lib.ptr(scale), _lib.ptr(m1), _lib.ptr(m2), 2,
                                                _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(dz), None, st),
                           "bn_bwd_apply")
                _lib.check(lib.p2r_stgcn_tconv_weight_grad(N, T, V, taps, _lib.ptr(z), _lib.ptr(scale),
                                                           _lib.ptr(shift), _lib.ptr(du), _N_BLOCKS, _lib.ptr(part),
                                                           _lib.ptr(bpart), st), "stgcn_tconv_weight_grad")
            dW = _lib.sum_leading(part).view(ctx.wshape)      # the kernel writes its partials in the weight's own (c, ci, tap) order
            if ctx.has_bias:        # row sums of du ride on the weight-gradient pass
                dbias = _lib.sum_leading(bpart)
        dadd = None
        if ctx.has_add and ctx.needs_input_grad[10]:       # the addend was broadcast over the joints: row sums of du
            from . import seed_op
            dadd = seed_op._rowsum(du, V, 1.0)
        return dz, dgamma, dbeta, None, dW, dbias, None, None, None, None, dadd


def supported(z, bn, conv):
    return (z.is_cuda and z.dtype == torch.float32 and z.dim() == 4 and z.shape[1] == 64 and z.shape[3] <= 64
            and conv.in_channels == 64 and conv.out_channels == 64 and tuple(conv.kernel_size) == (3, 1)
            and tuple(conv.stride) == (1, 1) and tuple(conv.padding) == (1, 0) and tuple(conv.dilation) == (1, 1)
            and conv.groups == 1 and bn_op.supported(z, bn))


def supported_pointwise(z, bn, conv):
    """z (B,64,T,V) view of a (B,64,L) activation; conv = Conv1d(64, 64, 1)."""
    return (z.is_cuda and z.dtype == torch.float32 and z.dim() == 4 and z.shape[1] == 64 and z.shape[3] <= 64
            and isinstance(conv, torch.nn.Conv1d) and conv.in_channels == 64 and conv.out_channels == 64
            and tuple(conv.kernel_size) == (1,) and tuple(conv.stride) == (1,) and tuple(conv.padding) == (0,)
            and conv.groups == 1 and bn_op.supported(z, bn))


class _Embed3(Function):
    """Conv1d(3 -> 64, kernel 1) on (B,3,L): streaming kernels of csrc/embed.hip."""

    @staticmethod
    def forward(ctx, x, weight, bias, want_stats=False):
        x = x.contiguous()
        B, _, L = x.shape
        out = torch.empty((B, 64, L), dtype=torch.float32, device=x.device)
        w2, b1 = weight.reshape(64, 3).contiguous(), (bias.contiguous() if bias is not None else None)
        stats = None
        with torch.cuda.device(x.device):
            if want_stats:      # batch statistics of the output from the moments of the three input rows
                stats = torch.empty((1, 64, 3), dtype=torch.float32, device=x.device)
                scratch = torch.empty((B * ((L + 1023) // 1024), 10), dtype=torch.float32, device=x.device)
                _lib.check(_lib.lib().p2r_embed3_forward_stats(B, L, _lib.ptr(x), _lib.ptr(w2), _lib.ptr(b1), _lib.ptr(out),
                                                               _lib.ptr(scratch), _lib.ptr(stats),
                                                               _lib.current_stream(x.device)), "embed3_forward_stats")
            else:
                _lib.check(_lib.lib().p2r_embed3_forward(B, L, _lib.ptr(x), _lib.ptr(w2), _lib.ptr(b1), _lib.ptr(out),
                                                         _lib.current_stream(x.device)), "embed3_forward")
        ctx.save_for_backward(x)
        ctx.has_bias = bias is not None
        ctx.wshape = weight.shape
        if want_stats:
            ctx.mark_non_differentiable(stats)
            return out, stats
        return out

    @staticmethod
    def backward(ctx, dout, _dstats=None):
        (x,) = ctx.saved_tensors
        assert not ctx.needs_input_grad[0], "embed3: the joint coordinates are inputs, no data gradient"
        dout = dout.contiguous()
        B, _, L = x.shape
        part = torch.empty((B * 64, 4), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().p2r_embed3_weight_grad(B, L, _lib.ptr(x), _lib.ptr(dout), _lib.ptr(part),
                                                         _lib.current_stream(x.device)), "embed3_weight_grad")
        tot = part.view(B, 64, 4).double().sum(0).float()
        return None, tot[:, :3].reshape(ctx.wshape).contiguous(), (tot[:, 3].contiguous() if ctx.has_bias else None), None


def supported_embed3(x, conv):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.shape[1] == 3 and not x.requires_grad
            and isinstance(conv, torch.nn.Conv1d) and conv.in_channels == 3 and conv.out_channels == 64
            and tuple(conv.kernel_size) == (1,) and tuple(conv.stride) == (1,) and tuple(conv.padding) == (0,)
            and conv.groups == 1)


def embed3(x, conv, want_stats=False):
    """want_stats: -> (out, statistics entries [1,64,3] of out for the BatchNorm that follows)"""
    return _Embed3.apply(x, conv.weight, conv.bias, want_stats)


def bn_relu_tconv(z, bn, conv, stats=None, want_stats=False, wp=None, add_ct=None):
    """stats: kernel partials [P,64,2] of z from its producer (bn_op.moments) instead of a statistics pass;
    want_stats: return (u, partials of u) for the BatchNorm that consumes u;
    wp: (forward, data-gradient) taps in kernel order from gcn_op.prepare_chain (train mode);
    add_ct (N,64,T): added to the result, broadcast over the joints (single-tap convolutions only)."""
    if bn.training:
        part = bn_op._stats_partial(z.contiguous()) if stats is None else stats
        fin = bn_op.finalize(part, z.numel() // z.shape[1], bn)     # also updates the running statistics
        if bn_op.GATE_HOOK is not None:
            bn_op.GATE_HOOK(bn, z, fin[2], fin[3], None)
        wp_f, wp_b = wp if wp is not None else (None, None)
        return _BNReLUTConv.apply(z, bn.weight, bn.bias, fin, conv.weight, conv.bias, True, want_stats, wp_f, wp_b, add_ct)
    invstd = torch.rsqrt(bn.running_var + bn.eps)
    scale = bn.weight * invstd
    fin = torch.stack([bn.running_mean, invstd, scale, bn.bias - bn.running_mean * scale]).detach()
    if bn_op.GATE_HOOK is not None:
        bn_op.GATE_HOOK(bn, z, fin[2], fin[3], None)
    return _BNReLUTConv.apply(z, bn.weight, bn.bias, fin, conv.weight, conv.bias, False, want_stats, None, None, add_ct)
