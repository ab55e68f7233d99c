"""BatchNorm (+ residual) (+ ReLU) as one fused op on the HIP kernels of csrc/bn_act.hip.

`fused_bn_act(x, bn, res=None, relu=True)` is numerically the reference's chain
`relu(bn(x) + res)` (stgcn_layers.py:399-439) for an `nn.BatchNorm2d` module `bn`, in train
mode (batch statistics, running-stat update with the module's momentum, unbiased running
variance) and eval mode (running statistics), differentiable w.r.t. x, bn.weight, bn.bias
and res.
"""
import torch
from torch.autograd import Function

from .. import _lib


def _rows(x):
    N, C = x.shape[0], x.shape[1]
    L = x.numel() // max(N * C, 1)
    return N, C, L


def _stats(x):
    """per-channel (mean, biased var) in fp64 from one HBM pass."""
    N, C, L = _rows(x)
    part = torch.empty((N * C, 2), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().p2r_bn_stats(N * C, L, _lib.ptr(x), _lib.ptr(part), _lib.current_stream(x.device)),
                   "bn_stats")
    tot = part.view(N, C, 2).double().sum(0)
    M = float(N * L)
    mean = tot[:, 0] / M
    var = (tot[:, 1] / M - mean * mean).clamp_(min=0.0)
    return mean, var, M


def moments(part, M):
    """(mean, biased var) in fp64 from kernel partials [P, C, 2] = (sum, sum of squares) over M elements
    per channel (the graph-conv / temporal-conv kernels emit these from their epilogues)."""
    tot = part.double().sum(0)
    mean = tot[:, 0] / M
    var = (tot[:, 1] / M - mean * mean).clamp_(min=0.0)
    return mean, var, float(M)


def _apply(x, scale, shift, res, relu):
    N, C, L = _rows(x)
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().p2r_bn_apply(N, C, L, _lib.ptr(x), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(res),
                                           int(relu), _lib.ptr(y), _lib.current_stream(x.device)), "bn_apply")
    return y


class _FusedBNAct(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, res, mean, invstd, relu):
        x = x.contiguous()
        res_c = res.contiguous() if res is not None else None
        scale = (weight * invstd).contiguous()
        shift = (bias - mean * scale).contiguous()
        y = _apply(x, scale, shift, res_c, relu)
        ctx.save_for_backward(x, y, weight, mean, invstd)
        ctx.relu = relu
        ctx.has_res = res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, weight, mean, invstd = ctx.saved_tensors
        dy = dy.contiguous()
        N, C, L = _rows(x)
        dev = x.device
        lib = _lib.lib()
        part = torch.empty((N * C, 2), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.p2r_bn_bwd_reduce(N, C, L, _lib.ptr(dy), _lib.ptr(y), _lib.ptr(x), _lib.ptr(mean),
                                             _lib.ptr(invstd), int(ctx.relu), None, None, _lib.ptr(part),
                                             _lib.current_stream(dev)), "bn_bwd_reduce")
        tot = part.view(N, C, 2).double().sum(0)
        dbias = tot[:, 0].float()
        dweight = tot[:, 1].float()
        M = float(N * L)
        m1 = (tot[:, 0] / M).float().contiguous()
        m2 = (tot[:, 1] / M).float().contiguous()
        kscale = (weight * invstd).contiguous()
        dx = torch.empty_like(x)
        dres = torch.empty_like(x) if ctx.has_res else None
        with torch.cuda.device(dev):
            _lib.check(lib.p2r_bn_bwd_apply(N, C, L, _lib.ptr(dy), _lib.ptr(y), _lib.ptr(x), _lib.ptr(mean),
                                            _lib.ptr(invstd), _lib.ptr(kscale), _lib.ptr(m1), _lib.ptr(m2),
                                            int(ctx.relu), None, None, _lib.ptr(dx), _lib.ptr(dres),
                                            _lib.current_stream(dev)), "bn_bwd_apply")
        return dx, dweight, dbias, dres, None, None, None


class _EvalBNAct(Function):
    """eval mode: fixed statistics, y = relu(x*scale + shift + res); gradient to x / res only."""

    @staticmethod
    def forward(ctx, x, scale, shift, res, relu):
        x = x.contiguous()
        y = _apply(x, scale.contiguous(), shift.contiguous(), res.contiguous() if res is not None else None, relu)
        ctx.save_for_backward(y, scale)
        ctx.relu, ctx.has_res = relu, res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        y, scale = ctx.saved_tensors
        g = dy * (y > 0) if ctx.relu else dy
        return g * scale.view(1, -1, *([1] * (dy.dim() - 2))), None, None, (g if ctx.has_res else None), None


def supported(x, bn):
    return x.is_cuda and x.dtype == torch.float32 and bn.affine and bn.track_running_stats and x.dim() >= 3


def fused_bn_act(x, bn, res=None, relu=True, stats=None):
    """stats: optional kernel partials [P, C, 2] of x (see `moments`) replacing the statistics pass."""
    if bn.training:
        mean64, var64, M = _stats(x.contiguous()) if stats is None else moments(stats, x.numel() // x.shape[1])
        with torch.no_grad():
            mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked + 1)
            bn.running_mean.mul_(1 - mom).add_(mom * mean64.float())
            unbiased = var64 * (M / max(M - 1.0, 1.0))
            bn.running_var.mul_(1 - mom).add_(mom * unbiased.float())
            bn.num_batches_tracked += 1
        mean = mean64.float()
        invstd = torch.rsqrt(var64 + bn.eps).float()
        return _FusedBNAct.apply(x, bn.weight, bn.bias, res, mean, invstd, relu)
    invstd = torch.rsqrt(bn.running_var + bn.eps)
    scale = bn.weight * invstd
    shift = bn.bias - bn.running_mean * scale
    return _EvalBNAct.apply(x, scale, shift, res, relu)
