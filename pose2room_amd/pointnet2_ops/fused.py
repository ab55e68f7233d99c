"""Fused vote-aggregation op: ball query + feature grouping + shared MLP (2 x Conv2d 1x1 +
ReLU) + max-pool in one HIP launch on fp32 MFMA (csrc/sa_votes.hip), forward and backward.

`PointnetSAModuleVotes` uses it on GPU tensors for the P2RNet configuration (mlp = [256, 256, 256], bn off, xyz
features off, max pooling, nsample 16); anything else runs the differentiable op chain.
"""
import ctypes

import torch
import torch.nn as nn
from torch.autograd import Function

from .. import _lib

_SPLIT = 64      # column ranges of the split-K weight-gradient product (16 output tiles x 64 = 1024 workgroups, four per CU)


def available():
    try:
        return hasattr(_lib.lib(), 'p2r_sa_votes_forward')
    except _lib.P2RLibraryError:
        return False


def supports(mlp_module, nsample):
    if nsample != 16 or len(mlp_module) != 4:
        return False
    c1, r1, c2, r2 = mlp_module
    return (isinstance(c1, nn.Conv2d) and isinstance(c2, nn.Conv2d) and isinstance(r1, nn.ReLU)
            and isinstance(r2, nn.ReLU) and c1.kernel_size == (1, 1) and c2.kernel_size == (1, 1)
            and c1.in_channels == c1.out_channels == c2.in_channels == c2.out_channels == 256
            and c1.bias is not None and c2.bias is not None)


def _forward(xyz, new_xyz, features, radius, nsample, w1, b1, w2, b2, train):
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    dev = xyz.device
    out = torch.empty((B, 256, M), dtype=torch.float32, device=dev)
    idx = torch.empty((B, M, nsample), dtype=torch.int32, device=dev)
    G = H = amax = None
    if train:
        G = torch.empty((B, 256, M, nsample), dtype=torch.float32, device=dev)
        H = torch.empty((B, 256, M, nsample), dtype=torch.float32, device=dev)
        amax = torch.empty((B, 256, M), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().p2r_sa_votes_forward(
            B, N, M, nsample, ctypes.c_float(radius), 256, 256, 256, _lib.ptr(xyz), _lib.ptr(new_xyz),
            _lib.ptr(features), _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), _lib.ptr(b2), _lib.ptr(idx), _lib.ptr(out),
            _lib.ptr(G), _lib.ptr(H), _lib.ptr(amax), _lib.current_stream(dev)), "sa_votes_forward")
    return out, idx, G, H, amax


def _weight_grad(dZ, X):
    """dW (256,256) = sum over samples and positions of dZ . X^T for dZ, X (B,256,M,16)."""
    B, L = dZ.shape[0], dZ.shape[2] * dZ.shape[3]
    part = torch.empty((_SPLIT, 256, 256), dtype=torch.float32, device=dZ.device)
    with torch.cuda.device(dZ.device):
        _lib.check(_lib.lib().p2r_gemm_nt_256(B, L, _SPLIT, _lib.ptr(dZ), _lib.ptr(X), _lib.ptr(part),
                                              _lib.current_stream(dZ.device)), "gemm_nt_256")
    return _lib.sum_leading(part)


class _SAVotes(Function):
    @staticmethod
    def forward(ctx, xyz, new_xyz, features, w1, b1, w2, b2, radius, nsample):
        xyz, new_xyz, features = xyz.contiguous(), new_xyz.contiguous(), features.contiguous()
        w1c, w2c = w1.reshape(256, 256).contiguous(), w2.reshape(256, 256).contiguous()
        out, idx, G, H, amax = _forward(xyz, new_xyz, features, radius, nsample, w1c, b1.contiguous(), w2c,
                                        b2.contiguous(), True)
        ctx.save_for_backward(out, idx, G, H, amax, w1c, w2c)
        ctx.n_points = xyz.shape[1]
        ctx.shapes = (w1.shape, w2.shape)
        ctx.mark_non_differentiable(idx)
        return out, idx

    @staticmethod
    def backward(ctx, dout, _didx=None):
        out, idx, G, H, amax, w1c, w2c = ctx.saved_tensors
        dout = dout.contiguous()
        B, C, M = out.shape
        S = idx.shape[2]
        dev = out.device
        dZ2, dZ1, dG = (torch.empty((B, C, M, S), dtype=torch.float32, device=dev) for _ in range(3))
        lib = _lib.lib()
        w2t, w1t = w2c.t().contiguous(), w1c.t().contiguous()     # named: a temporary's block could be handed out again
        with torch.cuda.device(dev):
            st = _lib.current_stream(dev)
            _lib.check(lib.p2r_sa_votes_backward(B, M, S, C, _lib.ptr(dout), _lib.ptr(out), _lib.ptr(amax), _lib.ptr(H),
                                                 _lib.ptr(w2t), _lib.ptr(w1t), _lib.ptr(dZ2), _lib.ptr(dZ1),
                                                 _lib.ptr(dG), st), "sa_votes_backward")
            dfeat = None
            if ctx.needs_input_grad[2]:
                dfeat = torch.empty((B, C, ctx.n_points), dtype=torch.float32, device=dev)
                _lib.check(lib.p2r_group_points_grad(B, C, ctx.n_points, M, S, _lib.ptr(dG), _lib.ptr(idx),
                                                     _lib.ptr(dfeat), st), "group_points_grad")
        dw1 = _weight_grad(dZ1, G).view(ctx.shapes[0]) if ctx.needs_input_grad[3] else None
        dw2 = _weight_grad(dZ2, H).view(ctx.shapes[1]) if ctx.needs_input_grad[5] else None
        db1 = dZ1.sum(dim=(0, 2, 3)) if ctx.needs_input_grad[4] else None
        db2 = dZ2.sum(dim=(0, 2, 3)) if ctx.needs_input_grad[6] else None
        return None, None, dfeat, dw1, db1, dw2, db2, None, None


def sa_votes(xyz, new_xyz, features, radius, nsample, mlp_module, return_idx=False):
    """xyz (B,N,3), new_xyz (B,M,3), features (B,256,N) -> new_features (B,256,M); differentiable w.r.t. the
    features and the four MLP tensors (the sampled centres and the ball membership carry no gradient, as in the
    reference's ball_query / group_points)."""
    c1, _, c2, _ = mlp_module
    if torch.is_grad_enabled() and (features.requires_grad or c1.weight.requires_grad or c2.weight.requires_grad):
        out, idx = _SAVotes.apply(xyz.detach(), new_xyz.detach(), features, c1.weight, c1.bias, c2.weight, c2.bias,
                                  radius, nsample)
    else:
        out, idx, _, _, _ = _forward(xyz.contiguous(), new_xyz.contiguous(), features.contiguous(), radius, nsample,
                                     c1.weight.detach().reshape(256, 256).contiguous(), c1.bias.detach().contiguous(),
                                     c2.weight.detach().reshape(256, 256).contiguous(), c2.bias.detach().contiguous(),
                                     False)
    return (out, idx) if return_idx else out
