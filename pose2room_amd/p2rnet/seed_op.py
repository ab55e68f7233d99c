"""Seams of the ST-GCN backbone on the streaming kernels of csrc/seed_ops.hip: the frame gather in front of
`conv_joint` (reference models/p2rnet/modules/stgcn.py:142-149) and the two short-row reductions of the embedding
(stgcn.py:118-121,129-130).  Each replaces an ATen advanced-indexing / reduction chain by one launch each way."""
import ctypes

import torch
from torch.autograd import Function

from .. import _lib


def _rowsum(x, V, scale):
    """x: contiguous f32 with a trailing axis of length V -> x.sum(-1) * scale"""
    out = torch.empty(x.shape[:-1], dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().p2r_rowsum_short(ctypes.c_longlong(x.numel() // V), V, ctypes.c_float(scale), _lib.ptr(x),
                                               _lib.ptr(out), _lib.current_stream(x.device)), "rowsum_short")
    return out


def short_rows_supported(x):
    return x.is_cuda and x.dtype == torch.float32 and 1 <= x.shape[-1] <= 64


class _SeedRows(Function):
    """x (B,C,T,J), seed_inds (B,S) int64 -> rows (B,S,C*J) with rows[b,s,c*J+j] = x[b,c,seed_inds[b,s],j]."""

    @staticmethod
    def forward(ctx, x, seed_inds):
        x = x.contiguous()
        inds = seed_inds.long().contiguous()
        B, C, T, J = x.shape
        S = inds.shape[1]
        out = torch.empty((B, S, C * J), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().p2r_gather_frames(B, C, T, J, S, _lib.ptr(x), _lib.ptr(inds), _lib.ptr(out),
                                                    _lib.current_stream(x.device)), "gather_frames")
        ctx.save_for_backward(inds)
        ctx.dims = (B, C, T, J, S)
        return out

    @staticmethod
    def backward(ctx, dout):
        (inds,) = ctx.saved_tensors
        B, C, T, J, S = ctx.dims
        dout = dout.contiguous()
        dx = torch.empty((B, C, T, J), dtype=torch.float32, device=dout.device)
        with torch.cuda.device(dout.device):
            _lib.check(_lib.lib().p2r_gather_frames_grad(B, C, T, J, S, _lib.ptr(dout), _lib.ptr(inds), _lib.ptr(dx),
                                                         _lib.current_stream(dout.device)), "gather_frames_grad")
        return dx, None


def seed_rows_supported(x, seed_inds):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and seed_inds.is_cuda and seed_inds.dim() == 2
            and seed_inds.dtype in (torch.int64, torch.int32) and seed_inds.shape[0] == x.shape[0])


def seed_rows(x, seed_inds):
    return _SeedRows.apply(x, seed_inds)


class _MeanLast(Function):
    """x (..., V) -> x.mean(-1), V <= 64."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.shape = x.shape
        return _rowsum(x, x.shape[-1], 1.0 / x.shape[-1])

    @staticmethod
    def backward(ctx, g):
        return (g * (1.0 / ctx.shape[-1])).unsqueeze(-1).expand(ctx.shape)


def mean_last(x):
    return _MeanLast.apply(x)


class _AddBroadcastLast(Function):
    """a (..., V) + b (...).unsqueeze(-1); the gradient of b is the row sum of the incoming gradient (one streaming
    launch instead of an ATen reduction over an inner axis of 53), the gradient of a is the incoming tensor itself."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.V = a.shape[-1]
        return a + b.unsqueeze(-1)

    @staticmethod
    def backward(ctx, g):
        db = None
        if ctx.needs_input_grad[1]:
            gc = g.contiguous()
            db = _rowsum(gc, ctx.V, 1.0)
        return (g if ctx.needs_input_grad[0] else None), db


def add_broadcast_last(a, b):
    return _AddBroadcastLast.apply(a, b)


def nearest_prefix(cum, target):
    """cum (B,T), target (B,S) f32 on the GPU -> (B,S) int64: first index t minimising |cum[b,t] - target[b,s]| -- what
    `torch.argmin(torch.abs(cum.unsqueeze(-1) - target.unsqueeze(1)), dim=1)` returns, in one launch and without the
    (B,T,S) tensor."""
    cum, target = cum.contiguous(), target.contiguous()
    B, T = cum.shape
    S = target.shape[1]
    out = torch.empty((B, S), dtype=torch.int64, device=cum.device)
    with torch.cuda.device(cum.device):
        _lib.check(_lib.lib().p2r_nearest_prefix(B, T, S, _lib.ptr(cum), _lib.ptr(target), _lib.ptr(out),
                                                 _lib.current_stream(cum.device)), "nearest_prefix")
    return out
