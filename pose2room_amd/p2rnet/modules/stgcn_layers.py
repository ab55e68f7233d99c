"""Skeleton graph and spatial-temporal graph-convolution block.

Host-side mirror of the reference's models/p2rnet/modules/stgcn_layers.py for
what P2RNet instantiates: `Graph('virtualroom', 'spatial', max_hop=5)` (:69-233),
`ConvTemporalGraphical` (:10-67) and `st_gcn_block` (:362-439).  Parameter names
(`gcn.conv`, `tcn.{0,2,3}`) match the reference `state_dict`.
"""
from collections import deque

import numpy as np
import torch
import torch.nn as nn

# 53-joint VirtualHome skeleton: (child, parent) bones, stgcn_layers.py:151-161
_VIRTUALROOM_BONES = [
    (0, 1), (1, 3), (3, 5), (5, 19), (0, 2), (2, 4), (4, 6), (6, 20), (0, 7), (7, 8), (8, 9),
    (9, 10), (10, 21), (10, 22), (8, 11), (11, 13), (13, 15), (15, 17), (8, 12), (12, 14),
    (14, 16), (16, 18), (17, 23), (23, 24), (24, 25), (17, 26), (26, 27), (27, 28), (17, 29),
    (29, 30), (30, 31), (17, 32), (32, 33), (33, 34), (17, 35), (35, 36), (36, 37), (18, 38),
    (38, 39), (39, 40), (18, 41), (41, 42), (42, 43), (18, 44), (44, 45), (45, 46), (18, 47),
    (47, 48), (48, 49), (18, 50), (50, 51), (51, 52)]

_LAYOUTS = {'virtualroom': dict(num_node=53, bones=_VIRTUALROOM_BONES, center=0)}


def hop_distance(num_node, edges, max_hop):
    """All-pairs hop count by BFS, inf beyond max_hop (same result as the
    reference's matrix-power construction, stgcn_layers.py:208-220)."""
    nbr = [[] for _ in range(num_node)]
    for i, j in edges:
        if i != j:
            nbr[i].append(j)
            nbr[j].append(i)
    hop = np.full((num_node, num_node), np.inf)
    for s in range(num_node):
        hop[s, s] = 0
        dq = deque([s])
        while dq:
            u = dq.popleft()
            if hop[s, u] >= max_hop:
                continue
            for v in nbr[u]:
                if np.isinf(hop[s, v]):
                    hop[s, v] = hop[s, u] + 1
                    dq.append(v)
    return hop


class Graph:
    """Adjacency stack `A (K, V, V)` float64 for a skeleton layout.

    strategy 'spatial': per hop h, entries with hop_dis == h are split by
    distance to the centre joint into root+closer and further sets (K = 1 + 2*max_hop);
    'uniform' and 'distance' as in the reference (stgcn_layers.py:163-205)."""

    def __init__(self, layout='virtualroom', strategy='spatial', max_hop=5, dilation=1):
        if layout not in _LAYOUTS:
            raise ValueError("Do Not Exist This Layout.")
        spec = _LAYOUTS[layout]
        self.max_hop, self.dilation = max_hop, dilation
        self.num_node, self.center = spec['num_node'], spec['center']
        self.edge = [(i, i) for i in range(self.num_node)] + list(spec['bones'])
        self.hop_dis = hop_distance(self.num_node, self.edge, max_hop)
        self.A = self._adjacency(strategy)

    def _adjacency(self, strategy):
        V = self.num_node
        hops = list(range(0, self.max_hop + 1, self.dilation))
        reach = np.zeros((V, V))
        for h in hops:
            reach[self.hop_dis == h] = 1
        col = reach.sum(0)
        inv = np.zeros(V)
        inv[col > 0] = col[col > 0] ** (-1)
        norm = reach * inv[None, :]          # A . D^-1 (column normalisation)
        if strategy == 'uniform':
            return norm[None]
        if strategy == 'distance':
            return np.stack([np.where(self.hop_dis == h, norm, 0.0) for h in hops])
        if strategy == 'spatial':
            dc = self.hop_dis[:, self.center]
            same = dc[:, None] == dc[None, :]        # [j, i]: j and i equally far from the centre
            closer = dc[:, None] > dc[None, :]       # row joint j further from centre than column joint i
            planes = []
            for h in hops:
                at_h = self.hop_dis == h             # symmetric, so [j,i] == [i,j]
                root = np.where(at_h & same, norm, 0.0)
                close = np.where(at_h & closer, norm, 0.0)
                further = np.where(at_h & ~same & ~closer, norm, 0.0)
                if h == 0:
                    planes.append(root)
                else:
                    planes.append(root + close)
                    planes.append(further)
            return np.stack(planes)
        raise ValueError("Do Not Exist This Strategy")


class ConvTemporalGraphical(nn.Module):
    """1x1 conv to K*C_out channels followed by the K-way graph contraction
    'nkctv,kvw->nctw' (stgcn_layers.py:10-67).  forward(x (N,C,T,V), A (K,V,V))."""

    def __init__(self, in_channels, out_channels, kernel_size, t_kernel_size=1, t_stride=1,
                 t_padding=0, t_dilation=1, bias=True):
        super().__init__()
        self.kernel_size = kernel_size
        self.out_channels = out_channels
        self.tables = None      # GraphTables of the adjacency pattern: enables the fused HIP path
        self.fused = True
        self.conv = nn.Conv2d(in_channels, out_channels * kernel_size, kernel_size=(t_kernel_size, 1),
                              padding=(t_padding, 0), stride=(t_stride, 1), dilation=(t_dilation, 1),
                              bias=bias)

    def forward(self, x, A, want_stats=False, with_residual=False, bn_link=None, prepared=None, lazy_res=None):
        """want_stats (fused GPU path only): return ((z, stats partials), A) -- the per-channel sums the
        following BatchNorm needs, produced by the kernel's epilogue (gcn_op.graph_conv).  with_residual (same
        path): x itself comes back as the last element of the tuple, for the caller's identity branch.
        bn_link, prepared: see gcn_op.graph_conv."""
        assert A.size(0) == self.kernel_size
        if self.tables is not None and self.fused:
            from .. import gcn_op
            if gcn_op.supported(x, self.conv.weight, A):
                return gcn_op.graph_conv(x, self.conv.weight, self.conv.bias, A, self.tables, want_stats,
                                         with_residual, bn_link, prepared, lazy_res), A
        assert not want_stats and not with_residual and prepared is None
        y = self.conv(x)
        n, kc, t, v = y.size()
        y = y.view(n, self.kernel_size, kc // self.kernel_size, t, v)
        z = torch.einsum('nkctv,kvw->nctw', (y, A))
        return z.contiguous(), A


def _zero(x):
    return 0


def _iden(x):
    return x


class st_gcn_block(nn.Module):
    """gcn -> BN -> ReLU -> temporal (k,1) conv -> BN -> dropout, plus residual,
    then ReLU (stgcn_layers.py:362-439)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dropout=0, residual=True):
        super().__init__()
        assert len(kernel_size) == 2 and kernel_size[0] % 2 == 1
        padding = ((kernel_size[0] - 1) // 2, 0)
        self.gcn = ConvTemporalGraphical(in_channels, out_channels, kernel_size[1])
        self.tcn = nn.Sequential(
            nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True),
            nn.Conv2d(out_channels, out_channels, (kernel_size[0], 1), (stride, 1), padding),
            nn.BatchNorm2d(out_channels),
            nn.Dropout(dropout, inplace=True),
        )
        if not residual:
            self.residual = _zero
        elif in_channels == out_channels and stride == 1:
            self.residual = _iden
        else:
            self.residual = nn.Sequential(
                nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=(stride, 1)),
                nn.BatchNorm2d(out_channels))
        self.relu = nn.ReLU(inplace=True)

    fused_bn = True   # BatchNorm + residual + ReLU on the fused HIP kernels (GPU tensors)
    fused_tconv = True   # BatchNorm + ReLU + temporal conv in one kernel
    lazy_residual_grad = True   # chain path: the residual gradient dout * mask is formed inside the graph-conv data gradient
    chain_input = False  # set by the owner when this block is the ONLY consumer of its input (the previous block's
                         # output): the data-gradient kernel then also serves that block's BatchNorm backward

    def chainable(self, x, A):
        """True when forward() takes the fully fused train-mode path for this input."""
        if not (self.fused_bn and x.is_cuda and self.tcn[4].p == 0):
            return False
        from .. import bn_op, gcn_op, tconv_op
        return (self.training and self.fused_tconv and self.gcn.tables is not None and self.gcn.fused
                and gcn_op.supported(x, self.gcn.conv.weight, A) and bn_op.supported(x, self.tcn[0])
                and tconv_op.supported(x, self.tcn[0], self.tcn[2]))

    def forward(self, x, A, prepared=None):
        """prepared: this block's gcn_op.BlockParams (only valid when `chainable`); A is then prepared.Aeff."""
        res = self.residual(x)
        if self.fused_bn and x.is_cuda and self.tcn[4].p == 0:
            from .. import bn_op, gcn_op, tconv_op
            # kernel epilogues hand the batch statistics to the BatchNorm that follows (train mode)
            chain = self.chainable(x, A)
            assert chain or prepared is None
            lazy = None
            if chain:
                if self.residual is _iden and x.requires_grad:
                    # identity branch routed through the graph-conv op: its gradient is added inside the
                    # data-gradient kernel instead of a separate accumulation pass over the activation -- and, being
                    # consumed there and nowhere else, it is handed over unmasked (bn_op._FusedBNAct, lazy_res): the
                    # kernel multiplies by the ReLU mask while it adds
                    in_link = getattr(x, '_p2r_bn_link', None) if self.chain_input else None
                    if self.lazy_residual_grad and self.tcn[3].training:      # (_FusedBNAct: train-mode BatchNorm only)
                        lazy = bn_op.ResLink()
                    (z, zstats, res), A = self.gcn(x, A, want_stats=True, with_residual=True, bn_link=in_link,
                                                   prepared=prepared, lazy_res=lazy)
                else:
                    (z, zstats), A = self.gcn(x, A, want_stats=True, prepared=prepared)
                wp = (prepared.tcn_wp_f, prepared.tcn_wp_b) if prepared is not None and z.shape[3] == 53 else None
                u, ustats = tconv_op.bn_relu_tconv(z, self.tcn[0], self.tcn[2], stats=zstats, want_stats=True, wp=wp)
                res_t = res if torch.is_tensor(res) else None
                return bn_op.fused_bn_act(u, self.tcn[3], res_t, relu=True, stats=ustats, link=bn_op.BNLink(),
                                          lazy_res=lazy), A
            x, A = self.gcn(x, A)
            if bn_op.supported(x, self.tcn[0]):
                if self.fused_tconv and tconv_op.supported(x, self.tcn[0], self.tcn[2]):
                    u = tconv_op.bn_relu_tconv(x, self.tcn[0], self.tcn[2])       # tcn.0 + tcn.1 + tcn.2
                else:
                    h = bn_op.fused_bn_act(x, self.tcn[0], None, relu=True)      # tcn.0 + tcn.1
                    u = self.tcn[2](h)                                             # temporal (3,1) conv
                res_t = res if torch.is_tensor(res) else None
                return bn_op.fused_bn_act(u, self.tcn[3], res_t, relu=True), A    # tcn.3 (+res) + relu
        else:
            x, A = self.gcn(x, A)
        x = self.tcn(x) + res
        return self.relu(x), A
