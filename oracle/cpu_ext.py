"""ctypes front end of the CPU oracle (oracle/p2r_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of p2r_oracle.c.  Importable from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from
pose2room_amd/.

`OracleExt` exposes the nine callables of the reference's pybind module
`pointnet2_ops._ext` (/root/reference/external/pointnet2_ops_lib/pointnet2_ops/
_ext-src/src/bindings.cpp:6-19) with the same positional signatures, operating
on CPU torch tensors, so it can stand in for `_ext` both under the imported
reference (golden-vector generation) and under our own host model (CPU tests,
CPU baseline).
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libp2r_oracle.so")


def build(force=False):
    """Compile the restatement with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "p2r_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libp2r_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.p2r_oracle_nms3d.restype = ctypes.c_int
        _lib.p2r_oracle_nms2d.restype = ctypes.c_int
        _lib.p2r_oracle_opt_n_threads.restype = ctypes.c_int
    return _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _chk(t, dtype, name):
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be a {dtype} tensor")
    if t.device.type != "cpu":
        raise RuntimeError(f"{name} must be a CPU tensor for the oracle")


class OracleExt:
    """CPU stand-in for `pointnet2_ops._ext` (bindings.cpp:6-19)."""

    @staticmethod
    def furthest_point_sampling(points, nsamples):  # sampling.cpp:66-87
        _chk(points, torch.float32, "points")
        b, n, _ = points.shape
        out = torch.zeros(b, nsamples, dtype=torch.int32)
        tmp = torch.full((b, n), 1e10, dtype=torch.float32)
        lib().p2r_oracle_furthest_point_sampling(b, n, nsamples, _p(points), _p(tmp), _p(out))
        return out

    @staticmethod
    def gather_points(points, idx):  # sampling.cpp:15-38
        _chk(points, torch.float32, "points"); _chk(idx, torch.int32, "idx")
        b, c, n = points.shape
        m = idx.shape[1]
        out = torch.zeros(b, c, m, dtype=torch.float32)
        lib().p2r_oracle_gather_points(b, c, n, m, _p(points), _p(idx), _p(out))
        return out

    @staticmethod
    def gather_points_grad(grad_out, idx, n):  # sampling.cpp:40-64
        _chk(grad_out, torch.float32, "grad_out"); _chk(idx, torch.int32, "idx")
        b, c, m = grad_out.shape
        out = torch.zeros(b, c, n, dtype=torch.float32)
        lib().p2r_oracle_gather_points_grad(b, c, n, m, _p(grad_out), _p(idx), _p(out))
        return out

    @staticmethod
    def ball_query(new_xyz, xyz, radius, nsample):  # ball_query.cpp:8-32
        _chk(new_xyz, torch.float32, "new_xyz"); _chk(xyz, torch.float32, "xyz")
        b, m, _ = new_xyz.shape
        n = xyz.shape[1]
        idx = torch.zeros(b, m, nsample, dtype=torch.int32)
        lib().p2r_oracle_ball_query(b, n, m, ctypes.c_float(radius), nsample,
                                    _p(new_xyz), _p(xyz), _p(idx))
        return idx

    @staticmethod
    def group_points(points, idx):  # group_points.cpp:12-36
        _chk(points, torch.float32, "points"); _chk(idx, torch.int32, "idx")
        b, c, n = points.shape
        _, npoints, nsample = idx.shape
        out = torch.zeros(b, c, npoints, nsample, dtype=torch.float32)
        lib().p2r_oracle_group_points(b, c, n, npoints, nsample, _p(points), _p(idx), _p(out))
        return out

    @staticmethod
    def group_points_grad(grad_out, idx, n):  # group_points.cpp:38-62
        _chk(grad_out, torch.float32, "grad_out"); _chk(idx, torch.int32, "idx")
        b, c, npoints, nsample = grad_out.shape
        out = torch.zeros(b, c, n, dtype=torch.float32)
        lib().p2r_oracle_group_points_grad(b, c, n, npoints, nsample, _p(grad_out), _p(idx), _p(out))
        return out

    @staticmethod
    def three_nn(unknowns, knows):  # interpolate.cpp:14-39
        _chk(unknowns, torch.float32, "unknowns"); _chk(knows, torch.float32, "knows")
        b, n, _ = unknowns.shape
        m = knows.shape[1]
        idx = torch.zeros(b, n, 3, dtype=torch.int32)
        dist2 = torch.zeros(b, n, 3, dtype=torch.float32)
        lib().p2r_oracle_three_nn(b, n, m, _p(unknowns), _p(knows), _p(dist2), _p(idx))
        return [dist2, idx]

    @staticmethod
    def three_interpolate(points, idx, weight):  # interpolate.cpp:41-69
        _chk(points, torch.float32, "points"); _chk(idx, torch.int32, "idx")
        _chk(weight, torch.float32, "weight")
        b, c, m = points.shape
        n = idx.shape[1]
        out = torch.zeros(b, c, n, dtype=torch.float32)
        lib().p2r_oracle_three_interpolate(b, c, m, n, _p(points), _p(idx), _p(weight), _p(out))
        return out

    @staticmethod
    def three_interpolate_grad(grad_out, idx, weight, m):  # interpolate.cpp:71-99
        _chk(grad_out, torch.float32, "grad_out"); _chk(idx, torch.int32, "idx")
        _chk(weight, torch.float32, "weight")
        b, c, n = grad_out.shape
        out = torch.zeros(b, c, m, dtype=torch.float32)
        lib().p2r_oracle_three_interpolate_grad(b, c, n, m, _p(grad_out), _p(idx), _p(weight), _p(out))
        return out


_MODES = {"l2": 0, "l1smooth": 1, "l1": 2}


def nn_distance(pc1, pc2, l1smooth=False, delta=1.0, l1=False):
    """Oracle of net_utils/nn_distance.py:34-61 (forward only, CPU tensors)."""
    pc1 = pc1.detach().contiguous().float()
    pc2 = pc2.detach().contiguous().float()
    B, N, C = pc1.shape
    M = pc2.shape[1]
    mode = 1 if l1smooth else (2 if l1 else 0)
    d1 = torch.empty(B, N); i1 = torch.empty(B, N, dtype=torch.int64)
    d2 = torch.empty(B, M); i2 = torch.empty(B, M, dtype=torch.int64)
    lib().p2r_oracle_nn_distance(B, N, M, C, _p(pc1), _p(pc2), mode, ctypes.c_float(delta),
                                 _p(d1), _p(i1), _p(d2), _p(i2))
    return d1, i1, d2, i2


def nn_distance_grad(pc1, pc2, idx1, idx2, g1, g2, l1smooth=False, delta=1.0, l1=False):
    pc1 = pc1.detach().contiguous().float(); pc2 = pc2.detach().contiguous().float()
    B, N, C = pc1.shape
    M = pc2.shape[1]
    mode = 1 if l1smooth else (2 if l1 else 0)
    ga = torch.empty(B, N, C); gb = torch.empty(B, M, C)
    lib().p2r_oracle_nn_distance_grad(B, N, M, C, _p(pc1), _p(pc2), mode, ctypes.c_float(delta),
                                      _p(idx1.contiguous()), _p(idx2.contiguous()),
                                      _p(g1.contiguous().float()), _p(g2.contiguous().float()),
                                      _p(ga), _p(gb))
    return ga, gb


def nms_3d(boxes, overlap_threshold, old_type=False, same_cls=False):
    """Oracle of net_utils/nms.py:41-77 / :79-119.  boxes (K,7|8) float64."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float64)
    K, stride = boxes.shape
    pick = np.zeros(max(K, 1), dtype=np.int32)
    n = lib().p2r_oracle_nms3d(K, stride, boxes.ctypes.data_as(ctypes.c_void_p),
                               ctypes.c_double(overlap_threshold), int(old_type), int(same_cls),
                               pick.ctypes.data_as(ctypes.c_void_p))
    return [int(v) for v in pick[:n]]


def nms_2d(boxes, overlap_threshold, old_type=False):
    """Oracle of net_utils/nms.py:7-39.  boxes (K,5) float64 rows [x1,y1,x2,y2,score]."""
    boxes = np.ascontiguousarray(boxes, dtype=np.float64)
    K, stride = boxes.shape
    assert stride == 5
    pick = np.zeros(max(K, 1), dtype=np.int32)
    n = lib().p2r_oracle_nms2d(K, boxes.ctypes.data_as(ctypes.c_void_p), ctypes.c_double(overlap_threshold),
                               int(old_type), pick.ctypes.data_as(ctypes.c_void_p))
    return [int(v) for v in pick[:n]]
