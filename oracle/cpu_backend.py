"""Run the host model of pose2room_amd on CPU with the oracle behind its ops.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).
`cpu_ops()` is a context manager that swaps, for its duration,
  pose2room_amd.pointnet2_ops.pointnet2_utils._ext  -> OracleExt (p2r_oracle.c)
  pose2room_amd.p2rnet.loss.nn_distance             -> oracle-backed autograd op
so the unchanged host code (modules, loss, trainer) can be exercised without a
GPU and timed as the CPU baseline ("port").  The product never does this itself.
"""
import contextlib

import torch
from torch.autograd import Function

from . import cpu_ext


class _OracleNNDistance(Function):
    @staticmethod
    def forward(ctx, pc1, pc2, l1smooth, delta, l1):
        a = pc1.detach().contiguous().float()
        q = pc2.detach().contiguous().float()
        d1, i1, d2, i2 = cpu_ext.nn_distance(a, q, l1smooth=l1smooth, delta=delta, l1=l1)
        ctx.save_for_backward(a, q, i1, i2)
        ctx.kw = dict(l1smooth=l1smooth, delta=delta, l1=l1)
        ctx.mark_non_differentiable(i1, i2)
        return d1, i1, d2, i2

    @staticmethod
    def backward(ctx, g1, gi1, g2, gi2):
        a, q, i1, i2 = ctx.saved_tensors
        g1 = torch.zeros(a.shape[:2]) if g1 is None else g1
        g2 = torch.zeros(q.shape[:2]) if g2 is None else g2
        ga, gq = cpu_ext.nn_distance_grad(a, q, i1, i2, g1, g2, **ctx.kw)
        return ga, gq, None, None, None


def nn_distance(pc1, pc2, l1smooth=False, delta=1.0, l1=False):
    return _OracleNNDistance.apply(pc1, pc2, bool(l1smooth), float(delta), bool(l1))


@contextlib.contextmanager
def cpu_ops():
    from pose2room_amd.pointnet2_ops import pointnet2_utils
    from pose2room_amd.p2rnet import loss as loss_mod
    saved = (pointnet2_utils._ext, loss_mod.nn_distance)
    pointnet2_utils._ext = cpu_ext.OracleExt
    loss_mod.nn_distance = nn_distance
    try:
        yield
    finally:
        pointnet2_utils._ext, loss_mod.nn_distance = saved
