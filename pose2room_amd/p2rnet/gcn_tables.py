"""Sparse neighbour tables of the skeleton adjacency for the fused graph-conv kernels.

For plane k of A (K,V,V) and output column w, the kernel needs the list of source
joints v with A[k,v,w] != 0 (column lists, forward) -- and for the data gradient the
row lists (for source row v, the w with A[k,v,w] != 0).  Lists are padded per plane to
the longest list of that plane; padded slots point at joint 0 with coefficient 0.

  nbr   uint8 [sum_k L_k, V]   joint index of the j-th neighbour
  gidx  int64 [sum_k L_k, V]   flat index into (K*V*V) of that entry, -1 when padded
  Lk    list[int]              list length per plane
`coefficients(Aeff, gidx)` gathers the current A*importance values (differentiable).
"""
import numpy as np
import torch


def build(A, transpose=False):
    A = np.asarray(A)
    K, V, _ = A.shape
    nz = A != 0
    Lk, nbr_rows, gidx_rows = [], [], []
    for k in range(K):
        lists = []
        for w in range(V):
            src = np.nonzero(nz[k, w, :])[0] if transpose else np.nonzero(nz[k, :, w])[0]
            lists.append(src)
        L = max(1, max(len(s) for s in lists))
        Lk.append(L)
        nb = np.zeros((L, V), dtype=np.uint8)
        gi = -np.ones((L, V), dtype=np.int64)
        for w, src in enumerate(lists):
            for j, v in enumerate(src):
                nb[j, w] = v
                # transpose: output column is the source row v_out = w, neighbour = destination joint
                gi[j, w] = (k * V + w) * V + v if transpose else (k * V + v) * V + w
        nbr_rows.append(nb)
        gidx_rows.append(gi)
    return torch.from_numpy(np.concatenate(nbr_rows)), torch.from_numpy(np.concatenate(gidx_rows)), Lk


def coefficients(Aeff, gidx):
    """Aeff (K,V,V) -> coef [sum L_k, V] with zeros at padded slots (differentiable)."""
    flat = Aeff.reshape(-1)
    safe = gidx.clamp(min=0)
    return torch.where(gidx >= 0, flat[safe], torch.zeros((), dtype=Aeff.dtype, device=Aeff.device))
