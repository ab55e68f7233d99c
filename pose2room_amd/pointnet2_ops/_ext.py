"""Drop-in for the reference's pybind module `pointnet2_ops._ext`.

Same nine callables, same positional signatures and return values as
/root/reference/external/pointnet2_ops_lib/pointnet2_ops/_ext-src/src/bindings.cpp:6-19
(implementations: ball_query.cpp:8-32, group_points.cpp:12-62,
interpolate.cpp:14-99, sampling.cpp:15-87), backed by libp2r_hip.so through the
C ABI of include/p2r_hip.h.

Behavioural contract kept from the reference wrappers:
  * inputs must be contiguous, float tensors float32, index tensors int32
    (CHECK_CONTIGUOUS / CHECK_IS_FLOAT / CHECK_IS_INT, include/utils.h:5-25)
    -> RuntimeError otherwise;
  * CPU tensors are rejected ("CPU not supported", e.g. ball_query.cpp:27-29);
  * outputs are freshly allocated, non-view tensors on the input's device;
  * work is enqueued on the current stream, no synchronisation.
Unlike the reference, a failed launch raises instead of calling exit(-1)
(include/cuda_utils.h:30-39).
"""
import ctypes

import torch

from .. import _lib

_c_int = ctypes.c_int
_c_float = ctypes.c_float


def _check(t, name, dtype):
    if not isinstance(t, torch.Tensor):
        raise RuntimeError(f"{name} must be a tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")
    if t.dtype != dtype:
        kind = "float" if dtype == torch.float32 else "int"
        raise RuntimeError(f"{name} must be a {kind} tensor")
    if not t.is_cuda:
        raise RuntimeError("CPU not supported")  # AT_ASSERT(false, "CPU not supported")


def _same_device(a, b, name):
    if a.device != b.device:
        raise RuntimeError(f"{name} must be on the same GPU as the first argument")


def _stream(t):
    return _lib.current_stream(t.device)


_FPS_COOP_MIN_N = 16384      # csrc/fps.hip: largest cloud one workgroup holds in registers


def furthest_point_sampling(points, nsamples):
    """points (B,N,3) f32 -> (B,nsamples) i32.  sampling.cpp:66-87."""
    _check(points, "points", torch.float32)
    b, n = points.size(0), points.size(1)
    output = torch.empty((b, nsamples), dtype=torch.int32, device=points.device)
    if b == 0 or nsamples == 0:
        return output
    # the register-resident kernel needs no scratch; the streaming one does
    tmp = None
    if n > 16384:
        tmp = torch.empty((b, n), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device):
        _lib.check(_lib.lib().p2r_furthest_point_sampling(
            _c_int(b), _c_int(n), _c_int(nsamples), _lib.ptr(points), _lib.ptr(tmp),
            _lib.ptr(output), _stream(points)), "furthest_point_sampling")
    if n > _FPS_COOP_MIN_N and b > 0 and nsamples > 0:
        # clouds beyond one workgroup's registers may run on several co-operating workgroups (csrc/fps.hip).  Should
        # a peer workgroup not show up within the kernel's spin bound the remaining picks come back as -1; surface
        # that here instead of letting gather_points read out of bounds.  (One sync, on this path only; the P2RNet
        # clouds of 512 points never take it.)
        if bool((output[:, -1] < 0).any()):
            raise RuntimeError("furthest_point_sampling: co-operating workgroups lost each other (device shared "
                               "with another stream / process?); no valid sample indices were produced")
    return output


def gather_points(points, idx):
    """points (B,C,N) f32, idx (B,M) i32 -> (B,C,M).  sampling.cpp:15-38."""
    _check(points, "points", torch.float32)
    _check(idx, "idx", torch.int32)
    _same_device(points, idx, "idx")
    b, c, n = points.shape
    m = idx.size(1)
    output = torch.empty((b, c, m), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device):
        _lib.check(_lib.lib().p2r_gather_points(
            _c_int(b), _c_int(c), _c_int(n), _c_int(m), _lib.ptr(points), _lib.ptr(idx),
            _lib.ptr(output), _stream(points)), "gather_points")
    return output


def gather_points_grad(grad_out, idx, n):
    """grad_out (B,C,M) f32, idx (B,M) i32 -> (B,C,n).  sampling.cpp:40-64."""
    _check(grad_out, "grad_out", torch.float32)
    _check(idx, "idx", torch.int32)
    _same_device(grad_out, idx, "idx")
    b, c, m = grad_out.shape
    output = torch.empty((b, c, n), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        _lib.check(_lib.lib().p2r_gather_points_grad(
            _c_int(b), _c_int(c), _c_int(n), _c_int(m), _lib.ptr(grad_out), _lib.ptr(idx),
            _lib.ptr(output), _stream(grad_out)), "gather_points_grad")
    return output


def ball_query(new_xyz, xyz, radius, nsample):
    """new_xyz (B,M,3), xyz (B,N,3) f32 -> idx (B,M,nsample) i32.  ball_query.cpp:8-32."""
    _check(new_xyz, "new_xyz", torch.float32)
    _check(xyz, "xyz", torch.float32)
    _same_device(new_xyz, xyz, "xyz")
    b, m = new_xyz.size(0), new_xyz.size(1)
    n = xyz.size(1)
    idx = torch.empty((b, m, nsample), dtype=torch.int32, device=new_xyz.device)
    with torch.cuda.device(new_xyz.device):
        _lib.check(_lib.lib().p2r_ball_query(
            _c_int(b), _c_int(n), _c_int(m), _c_float(radius), _c_int(nsample),
            _lib.ptr(new_xyz), _lib.ptr(xyz), _lib.ptr(idx), _stream(new_xyz)), "ball_query")
    return idx


def group_points(points, idx):
    """points (B,C,N) f32, idx (B,P,S) i32 -> (B,C,P,S).  group_points.cpp:12-36."""
    _check(points, "points", torch.float32)
    _check(idx, "idx", torch.int32)
    _same_device(points, idx, "idx")
    b, c, n = points.shape
    npoints, nsample = idx.size(1), idx.size(2)
    output = torch.empty((b, c, npoints, nsample), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device):
        _lib.check(_lib.lib().p2r_group_points(
            _c_int(b), _c_int(c), _c_int(n), _c_int(npoints), _c_int(nsample), _lib.ptr(points),
            _lib.ptr(idx), _lib.ptr(output), _stream(points)), "group_points")
    return output


def group_points_grad(grad_out, idx, n):
    """grad_out (B,C,P,S) f32, idx (B,P,S) i32 -> (B,C,n).  group_points.cpp:38-62."""
    _check(grad_out, "grad_out", torch.float32)
    _check(idx, "idx", torch.int32)
    _same_device(grad_out, idx, "idx")
    b, c, npoints, nsample = grad_out.shape
    output = torch.empty((b, c, n), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        _lib.check(_lib.lib().p2r_group_points_grad(
            _c_int(b), _c_int(c), _c_int(n), _c_int(npoints), _c_int(nsample), _lib.ptr(grad_out),
            _lib.ptr(idx), _lib.ptr(output), _stream(grad_out)), "group_points_grad")
    return output


def three_nn(unknowns, knows):
    """unknowns (B,n,3), knows (B,m,3) f32 -> [dist2 (B,n,3) f32, idx (B,n,3) i32].
    interpolate.cpp:14-39."""
    _check(unknowns, "unknowns", torch.float32)
    _check(knows, "knows", torch.float32)
    _same_device(unknowns, knows, "knows")
    b, n = unknowns.size(0), unknowns.size(1)
    m = knows.size(1)
    idx = torch.empty((b, n, 3), dtype=torch.int32, device=unknowns.device)
    dist2 = torch.empty((b, n, 3), dtype=torch.float32, device=unknowns.device)
    with torch.cuda.device(unknowns.device):
        _lib.check(_lib.lib().p2r_three_nn(
            _c_int(b), _c_int(n), _c_int(m), _lib.ptr(unknowns), _lib.ptr(knows), _lib.ptr(dist2),
            _lib.ptr(idx), _stream(unknowns)), "three_nn")
    return [dist2, idx]


def three_interpolate(points, idx, weight):
    """points (B,c,m) f32, idx (B,n,3) i32, weight (B,n,3) f32 -> (B,c,n).
    interpolate.cpp:41-69."""
    _check(points, "points", torch.float32)
    _check(idx, "idx", torch.int32)
    _check(weight, "weight", torch.float32)
    _same_device(points, idx, "idx")
    _same_device(points, weight, "weight")
    b, c, m = points.shape
    n = idx.size(1)
    output = torch.empty((b, c, n), dtype=torch.float32, device=points.device)
    with torch.cuda.device(points.device):
        _lib.check(_lib.lib().p2r_three_interpolate(
            _c_int(b), _c_int(c), _c_int(m), _c_int(n), _lib.ptr(points), _lib.ptr(idx),
            _lib.ptr(weight), _lib.ptr(output), _stream(points)), "three_interpolate")
    return output


def three_interpolate_grad(grad_out, idx, weight, m):
    """grad_out (B,c,n) f32, idx, weight (B,n,3) -> (B,c,m).  interpolate.cpp:71-99."""
    _check(grad_out, "grad_out", torch.float32)
    _check(idx, "idx", torch.int32)
    _check(weight, "weight", torch.float32)
    _same_device(grad_out, idx, "idx")
    _same_device(grad_out, weight, "weight")
    b, c, n = grad_out.shape
    output = torch.empty((b, c, m), dtype=torch.float32, device=grad_out.device)
    with torch.cuda.device(grad_out.device):
        _lib.check(_lib.lib().p2r_three_interpolate_grad(
            _c_int(b), _c_int(c), _c_int(n), _c_int(m), _lib.ptr(grad_out), _lib.ptr(idx),
            _lib.ptr(weight), _lib.ptr(output), _stream(grad_out)), "three_interpolate_grad")
    return output
