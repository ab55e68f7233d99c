"""Fused vote-aggregation op: ball query + feature grouping + shared MLP (2 x Conv2d 1x1 +
ReLU) + max-pool in one HIP launch on fp32 MFMA (csrc/sa_votes.hip).

Forward only: `PointnetSAModuleVotes` uses it when autograd is off (the `generate` /
evaluation path); training runs the differentiable op chain.
"""
import ctypes

import torch
import torch.nn as nn

from .. import _lib


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


def sa_votes(xyz, new_xyz, features, radius, nsample, mlp_module, return_idx=False):
    """xyz (B,N,3), new_xyz (B,M,3), features (B,256,N) -> new_features (B,256,M)."""
    xyz, new_xyz, features = xyz.contiguous(), new_xyz.contiguous(), features.contiguous()
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    c1, _, c2, _ = mlp_module
    w1 = c1.weight.detach().reshape(256, 256).contiguous()
    w2 = c2.weight.detach().reshape(256, 256).contiguous()
    out = torch.empty((B, 256, M), dtype=torch.float32, device=xyz.device)
    idx = torch.empty((B, M, nsample), dtype=torch.int32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        _lib.check(_lib.lib().p2r_sa_votes_forward(
            B, N, M, nsample, ctypes.c_float(radius), 256, 256, 256, _lib.ptr(xyz), _lib.ptr(new_xyz),
            _lib.ptr(features), _lib.ptr(w1), _lib.ptr(c1.bias.detach().contiguous()), _lib.ptr(w2),
            _lib.ptr(c2.bias.detach().contiguous()), _lib.ptr(idx), _lib.ptr(out),
            _lib.current_stream(xyz.device)), "sa_votes_forward")
    return (out, idx) if return_idx else out
