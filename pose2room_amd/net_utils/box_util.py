"""Oriented 3D box IoU, batched.

Same results as the reference's net_utils/box_util.py:90-118 `box3d_iou(corners1, corners2)`
-> (iou_3d, iou_2d) for boxes given as 8 corners in the order of utils/tools.py:33-51
(`get_box_corners`), evaluated for whole pair lists at once instead of one Python call per
(detection, ground truth) pair:

  * footprint = corners 3,2,6,7 projected on (x,z) (box_util.py:104-108: permutation
    [7,6,2,3,4,5,1,0], then rows 3..0) -- counter-clockwise for any heading;
  * footprint intersection by Sutherland-Hodgman clipping (box_util.py:22-69) with the same strict
    `inside` predicate and the same line-intersection expression, run in lock-step over all pairs on
    fixed-capacity vertex buffers (a convex quad clipped by four half-planes has <= 8 vertices);
  * intersection area by the shoelace formula (the reference asks SciPy's ConvexHull for the `volume`
    of the clipped polygon, box_util.py:71-81 -- the same number for the convex polygon the clip
    returns, without a qhull call per pair);
  * height overlap from corner 7 (top) and corner 4 (bottom), volumes from three edge lengths
    (box_util.py:83-88,110-117).

Everything is float64 tensor math, on whichever device the corners live on.  Degenerate inputs stay
degenerate: footprints with coincident edge lines (e.g. a box against itself) hit the strict predicate
and a zero denominator in the reference as well, so their IoU is unspecified on both sides; a clipped
polygon that collapses to a line makes the reference raise QhullError, here its area is 0.
"""
import numpy as np
import torch

_CAP = 16   # vertex-buffer capacity per pair (8 suffices for convex input; the slack absorbs rounding artefacts)


def _cross_inside(cp1, cp2, p):
    """box_util.py:37-38: p strictly left of the directed clip edge cp1->cp2."""
    return (cp2[..., 0] - cp1[..., 0]) * (p[..., 1] - cp1[..., 1]) > (cp2[..., 1] - cp1[..., 1]) * (p[..., 0] - cp1[..., 0])


def _line_hit(cp1, cp2, s, e):
    """box_util.py:40-46: intersection of line cp1-cp2 with line s-e."""
    dcx, dcy = cp1[..., 0] - cp2[..., 0], cp1[..., 1] - cp2[..., 1]
    dpx, dpy = s[..., 0] - e[..., 0], s[..., 1] - e[..., 1]
    n1 = cp1[..., 0] * cp2[..., 1] - cp1[..., 1] * cp2[..., 0]
    n2 = s[..., 0] * e[..., 1] - s[..., 1] * e[..., 0]
    n3 = 1.0 / (dcx * dpy - dcy * dpx)
    return torch.stack([(n1 * dpx - n2 * dcx) * n3, (n1 * dpy - n2 * dcy) * n3], dim=-1)


def clip_area(subject, clip):
    """subject, clip (P,4,2) f64 counter-clockwise quads -> area (P,) of subject clipped by clip."""
    P = subject.shape[0]
    dev = subject.device
    poly = torch.zeros((P, _CAP, 2), dtype=torch.float64, device=dev)
    poly[:, :4] = subject
    cnt = torch.full((P,), 4, dtype=torch.long, device=dev)
    ar = torch.arange(_CAP, device=dev)
    cp1 = clip[:, 3]
    for j in range(4):
        cp2 = clip[:, j]
        prev = torch.where(ar[None, :] == 0, (cnt - 1).clamp(min=0)[:, None], ar[None, :] - 1)      # (P,CAP)
        s = torch.gather(poly, 1, prev[..., None].expand(-1, -1, 2))
        e = poly
        valid = ar[None, :] < cnt[:, None]
        in_e = _cross_inside(cp1[:, None], cp2[:, None], e)
        in_s = _cross_inside(cp1[:, None], cp2[:, None], s)
        hit = _line_hit(cp1[:, None], cp2[:, None], s, e)
        # per input vertex the clip emits [hit if the edge s->e crosses the clip line][e if e is inside]
        cand = torch.stack([hit, e], dim=2).reshape(P, 2 * _CAP, 2)
        keep = torch.stack([valid & (in_e != in_s), valid & in_e], dim=2).reshape(P, 2 * _CAP)
        pos = torch.cumsum(keep, dim=1) - 1
        cnt = keep.sum(dim=1).clamp(max=_CAP)
        keep = keep & (pos < _CAP)
        dst = torch.where(keep, pos, torch.full_like(pos, _CAP))                                     # dropped -> spare slot
        new = torch.zeros((P, _CAP + 1, 2), dtype=torch.float64, device=dev)
        new.scatter_(1, dst[..., None].expand(-1, -1, 2), torch.where(keep[..., None], cand, torch.zeros_like(cand)))
        poly = new[:, :_CAP]
        cp1 = cp2
    # shoelace over the first cnt vertices (unused slots repeat vertex 0 -> no contribution)
    valid = ar[None, :] < cnt[:, None]
    pts = torch.where(valid[..., None], poly, poly[:, :1].expand(-1, _CAP, -1))
    nxt = torch.roll(pts, shifts=-1, dims=1)
    area = 0.5 * torch.abs((pts[..., 0] * nxt[..., 1] - pts[..., 1] * nxt[..., 0]).sum(dim=1))
    return torch.where(cnt >= 3, area, torch.zeros_like(area))


def _quad_area(q):
    """box_util.py:17-20 (shoelace on a quad)."""
    x, y = q[..., 0], q[..., 1]
    return 0.5 * torch.abs((x * torch.roll(y, 1, -1)).sum(-1) - (y * torch.roll(x, 1, -1)).sum(-1))


def box3d_iou_pairs(c1, c2):
    """c1, c2 (P,8,3) corners -> (iou_3d (P,), iou_2d (P,)) f64, pair i = (c1[i], c2[i])."""
    c1 = torch.as_tensor(c1).to(torch.float64)
    c2 = torch.as_tensor(c2).to(torch.float64)
    foot = [3, 2, 6, 7]
    r1 = c1[:, foot][..., [0, 2]]
    r2 = c2[:, foot][..., [0, 2]]
    a1, a2 = _quad_area(r1), _quad_area(r2)
    inter = clip_area(r1, r2)
    iou2d = inter / (a1 + a2 - inter)
    ymax = torch.minimum(c1[:, 7, 1], c2[:, 7, 1])
    ymin = torch.maximum(c1[:, 4, 1], c2[:, 4, 1])
    inter_vol = inter * (ymax - ymin).clamp(min=0.0)

    def vol(c):   # box_util.py:83-88 on the permuted corners: |c7-c6| |c6-c2| |c7-c4|
        return (torch.sqrt(((c[:, 7] - c[:, 6]) ** 2).sum(-1)) * torch.sqrt(((c[:, 6] - c[:, 2]) ** 2).sum(-1))
                * torch.sqrt(((c[:, 7] - c[:, 4]) ** 2).sum(-1)))

    iou = inter_vol / (vol(c1) + vol(c2) - inter_vol)
    return iou, iou2d


def box3d_iou_matrix(c1, c2):
    """c1 (n,8,3), c2 (m,8,3) -> iou_3d (n,m) f64 (every detection against every ground truth)."""
    c1 = torch.as_tensor(c1).to(torch.float64)
    c2 = torch.as_tensor(c2).to(torch.float64).to(c1.device)
    n, m = c1.shape[0], c2.shape[0]
    if n == 0 or m == 0:
        return torch.zeros((n, m), dtype=torch.float64, device=c1.device)
    iou, _ = box3d_iou_pairs(c1[:, None].expand(n, m, 8, 3).reshape(-1, 8, 3),
                             c2[None].expand(n, m, 8, 3).reshape(-1, 8, 3))
    return iou.view(n, m)


def box3d_iou(corners1, corners2):
    """Drop-in for box_util.py:90 `box3d_iou`: two (8,3) arrays -> (iou_3d, iou_2d) floats."""
    iou, iou2d = box3d_iou_pairs(np.asarray(corners1, dtype=np.float64)[None], np.asarray(corners2, dtype=np.float64)[None])
    return float(iou[0]), float(iou2d[0])
