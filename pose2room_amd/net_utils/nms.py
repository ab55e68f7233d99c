"""2D / 3D NMS on the HIP kernel.

Mirror of the reference's net_utils/nms.py: `nms_2d_faster(boxes (K,5),
overlap_threshold, old_type=False) -> list[int]` (:7-39), `nms_3d_faster(boxes (K,7),
overlap_threshold, old_type=False) -> list[int]` (:41-77) and
`nms_3d_faster_samecls(boxes (K,8), ...)` (:79-119) keep their signatures and
return the picked indices in pick order.  They accept a NumPy array (copied to
the current GPU) or a CUDA float64 tensor.  `nms_3d_batched` is the
device-resident form used by the eval path: B box sets in one launch, returning
the keep mask without leaving the GPU (the reference loops over the batch on the
host, net_utils/ap_helper.py:216-232).  No CPU fallback.
"""
import ctypes

import numpy as np
import torch

from .. import _lib


def nms_3d_batched(boxes, overlap_threshold, old_type=False, same_cls=False, valid=None,
                   return_pick=False):
    """boxes (B,K,7|8) float64 CUDA; valid (B,K) bool/uint8 or None.
    Returns keep (B,K) uint8 [, pick (B,K) int32 padded with -1, npick (B) int32]."""
    if not boxes.is_cuda:
        raise RuntimeError("nms_3d_batched: GPU tensor required (no CPU fallback)")
    if boxes.dtype != torch.float64:
        raise RuntimeError("nms_3d_batched: float64 boxes required (the reference runs NMS in fp64)")
    boxes = boxes.contiguous()
    B, K, stride = boxes.shape
    dev = boxes.device
    v = None
    if valid is not None:
        v = valid.to(device=dev, dtype=torch.uint8).contiguous()
    keep = torch.zeros((B, K), dtype=torch.uint8, device=dev)
    pick = torch.full((B, K), -1, dtype=torch.int32, device=dev) if return_pick else None
    npick = torch.zeros((B,), dtype=torch.int32, device=dev) if return_pick else None
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().p2r_nms3d(
            ctypes.c_int(B), ctypes.c_int(K), ctypes.c_int(stride), _lib.ptr(boxes), _lib.ptr(v),
            ctypes.c_double(overlap_threshold), ctypes.c_int(int(old_type)),
            ctypes.c_int(int(same_cls)), _lib.ptr(keep), _lib.ptr(pick), _lib.ptr(npick),
            _lib.current_stream(dev)), "nms3d")
    if return_pick:
        return keep, pick, npick
    return keep


def _as_cuda_f64(boxes):
    if isinstance(boxes, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(boxes, dtype=np.float64)).cuda()
    return boxes.to(dtype=torch.float64)


def _pick_list(boxes, overlap_threshold, old_type, same_cls):
    t = _as_cuda_f64(boxes)
    if t.shape[0] == 0:
        return []
    _, pick, npick = nms_3d_batched(t.unsqueeze(0), overlap_threshold, old_type, same_cls,
                                    return_pick=True)
    n = int(npick[0].item())
    return [int(i) for i in pick[0, :n].tolist()]


def nms_3d_faster(boxes, overlap_threshold, old_type=False):
    """boxes (K,7) rows [x1,y1,z1,x2,y2,z2,score] -> picked indices, best score first."""
    return _pick_list(boxes, overlap_threshold, old_type, False)


def nms_3d_faster_samecls(boxes, overlap_threshold, old_type=False):
    """boxes (K,8) rows [...,score,cls]: only same-class boxes suppress each other."""
    return _pick_list(boxes, overlap_threshold, old_type, True)


def boxes_2d_as_3d(boxes):
    """(..., 5) rows [x1,y1,x2,y2,score] -> (..., 7) rows [x1,0,y1,x2,1,y2,score]: a unit extent along the middle axis.
    The 3-D kernel's arithmetic on these rows IS the 2-D arithmetic of nms.py:7-39, bit for bit: the volume
    (x2-x1)*(1-0)*(y2-y1) and the intersection l*max(0,1-0)*h multiply by exactly 1.0 in between, which changes no
    bit of an IEEE product."""
    if isinstance(boxes, np.ndarray):
        b = np.ascontiguousarray(boxes, dtype=np.float64)
        zeros, ones = np.zeros_like(b[..., :1]), np.ones_like(b[..., :1])
        return np.concatenate([b[..., 0:1], zeros, b[..., 1:2], b[..., 2:3], ones, b[..., 3:4], b[..., 4:5]], -1)
    b = boxes.to(dtype=torch.float64)
    zeros, ones = torch.zeros_like(b[..., :1]), torch.ones_like(b[..., :1])
    return torch.cat([b[..., 0:1], zeros, b[..., 1:2], b[..., 2:3], ones, b[..., 3:4], b[..., 4:5]], -1)


def nms_2d_faster(boxes, overlap_threshold, old_type=False):
    """boxes (K,5) rows [x1,y1,x2,y2,score] -> picked indices, best score first (nms.py:7-39; the `use_3d_nms: False`
    branch of ap_helper.py:198-214 calls it with the (x, z) extents of the predicted boxes)."""
    if boxes.shape[-1] != 5:
        raise ValueError("nms_2d_faster: boxes must be (K,5) [x1,y1,x2,y2,score]")
    return _pick_list(boxes_2d_as_3d(boxes), overlap_threshold, old_type, False)
