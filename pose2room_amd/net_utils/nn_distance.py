"""Chamfer / nearest-neighbour distance on the HIP kernel.

Mirror of the reference's net_utils/nn_distance.py: `huber_loss(error, delta)`
(:15-32) and `nn_distance(pc1, pc2, l1smooth=False, delta=1.0, l1=False) ->
(dist1 (B,N) f32, idx1 (B,N) i64, dist2 (B,M) f32, idx2 (B,M) i64)` (:34-61),
differentiable through dist1 / dist2 w.r.t. both clouds.

The reference materialises (B,N,M,C) intermediates with `repeat`; here forward
and backward are one HIP launch each (pose2room_amd/csrc/nn_distance.hip) and
nothing of size N*M touches HBM.  CUDA/HIP tensors only -- no CPU fallback.
"""
import ctypes

import torch
from torch.autograd import Function

from .. import _lib

_MODE_L2, _MODE_L1SMOOTH, _MODE_L1 = 0, 1, 2


def huber_loss(error, delta=1.0):
    """0.5*min(|x|,d)^2 + d*(|x| - min(|x|,d)), element-wise (nn_distance.py:15-32)."""
    abs_error = torch.abs(error)
    quadratic = torch.clamp(abs_error, max=delta)
    linear = abs_error - quadratic
    return 0.5 * quadratic ** 2 + delta * linear


class _NNDistance(Function):
    @staticmethod
    def forward(ctx, pc1, pc2, mode, delta):
        if not (pc1.is_cuda and pc2.is_cuda):
            raise RuntimeError("nn_distance: GPU tensors required (no CPU fallback)")
        if pc1.dtype != torch.float32 or pc2.dtype != torch.float32:
            raise RuntimeError("nn_distance: float32 tensors required")
        a = pc1.contiguous()
        q = pc2.contiguous()
        B, N, C = a.shape
        M = q.shape[1]
        if q.shape[0] != B or q.shape[2] != C:
            raise RuntimeError("nn_distance: pc1 (B,N,C) and pc2 (B,M,C) shapes disagree")
        dev = a.device
        dist1 = torch.empty((B, N), dtype=torch.float32, device=dev)
        idx1 = torch.empty((B, N), dtype=torch.int64, device=dev)
        dist2 = torch.empty((B, M), dtype=torch.float32, device=dev)
        idx2 = torch.empty((B, M), dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().p2r_nn_distance(
                ctypes.c_int(B), ctypes.c_int(N), ctypes.c_int(M), ctypes.c_int(C), _lib.ptr(a),
                _lib.ptr(q), ctypes.c_int(mode), ctypes.c_float(delta), _lib.ptr(dist1),
                _lib.ptr(idx1), _lib.ptr(dist2), _lib.ptr(idx2), _lib.current_stream(dev)),
                "nn_distance")
        ctx.save_for_backward(a, q, idx1, idx2)
        ctx.mode, ctx.delta = mode, delta
        ctx.mark_non_differentiable(idx1, idx2)
        return dist1, idx1, dist2, idx2

    @staticmethod
    def backward(ctx, g1, gi1, g2, gi2):
        a, q, idx1, idx2 = ctx.saved_tensors
        B, N, C = a.shape
        M = q.shape[1]
        g1 = g1.contiguous().float() if g1 is not None else None
        g2 = g2.contiguous().float() if g2 is not None else None
        ga = torch.empty_like(a)
        gq = torch.empty_like(q)
        with torch.cuda.device(a.device):
            _lib.check(_lib.lib().p2r_nn_distance_grad(
                ctypes.c_int(B), ctypes.c_int(N), ctypes.c_int(M), ctypes.c_int(C), _lib.ptr(a),
                _lib.ptr(q), ctypes.c_int(ctx.mode), ctypes.c_float(ctx.delta), _lib.ptr(idx1),
                _lib.ptr(idx2), _lib.ptr(g1), _lib.ptr(g2), _lib.ptr(ga), _lib.ptr(gq),
                _lib.current_stream(a.device)), "nn_distance_grad")
        return ga, gq, None, None


def nn_distance(pc1, pc2, l1smooth=False, delta=1.0, l1=False):
    """pc1 (B,N,C), pc2 (B,M,C) -> dist1, idx1, dist2, idx2 (see module docstring)."""
    mode = _MODE_L1SMOOTH if l1smooth else (_MODE_L1 if l1 else _MODE_L2)
    return _NNDistance.apply(pc1, pc2, mode, float(delta))
