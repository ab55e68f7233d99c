"""ST-GCN pose-sequence backbone (mirror of the reference's
models/p2rnet/modules/stgcn.py:12-152; same `state_dict` keys).

forward(input_joints (B,T,53,3), end_points) adds
  seed_inds (B,S) i64, seed_skeleton (B,S,53,3), seed_features (B,S,256).
"""
import torch
import torch.nn as nn

from ..registers import MODULES
from .stgcn_layers import Graph, st_gcn_block
from .sub_modules import SingleConv


def _point_mlp(cin, width, cout):
    return nn.Sequential(SingleConv(cin, width, kernel_size=1, order='cbr', padding=0, ndim=1),
                         SingleConv(width, width, kernel_size=1, order='cbr', padding=0, ndim=1),
                         SingleConv(width, cout, kernel_size=1, order='c', padding=0, ndim=1))


@MODULES.register_module
class STGCN(nn.Module):
    def __init__(self, cfg, optim_spec=None):
        super().__init__()
        self.optim_spec = optim_spec

        self.graph = Graph(layout='virtualroom', strategy='spatial', max_hop=5)
        self.register_buffer('A', torch.tensor(self.graph.A, dtype=torch.float32, requires_grad=False))

        self.n_seeds = cfg.config['data']['num_seeds']
        self.origin_joint_id = cfg.dataset_config.origin_joint_id
        width, out_channels = 64, 256
        kernel_size = (3, self.A.size(0))   # (temporal, spatial partitions)
        self.knn = 20                        # temporal window of the position embedding

        self.pos_embed = _point_mlp(3, 64, width)
        self.sk_feat = _point_mlp(3, 64, width)
        self.st_gcn_networks = nn.ModuleList(
            [st_gcn_block(width, 64, kernel_size, 1, residual=False)] +
            [st_gcn_block(64, 64, kernel_size, 1) for _ in range(5)])
        from ..gcn_op import GraphTables
        tables = GraphTables(self.graph.A)     # sparse form of the skeleton adjacency for the fused kernels
        for i, blk in enumerate(self.st_gcn_networks):
            blk.gcn.tables = tables
            blk.chain_input = i > 0     # forward(): block i is the only consumer of block i-1's output
        self.conv_joint = nn.Conv1d(cfg.dataset_config.joint_num * 64, out_channels, kernel_size=1)
        self.edge_importance = nn.ParameterList(
            [nn.Parameter(torch.ones(self.A.size())) for _ in self.st_gcn_networks])

        n_frames = cfg.config['data']['num_frames']
        if self.n_seeds >= n_frames:   # stgcn.py:79-80: every frame (repeated) is a seed
            self.seed_inds = torch.round(torch.linspace(0, n_frames - 1, self.n_seeds)).long()
        else:
            self.seed_sampling = cfg.config['data']['seed_sampling']

    # -- seed selection (stgcn.py:88-103) ------------------------------------
    def _select_seeds(self, hip):
        n_batch, n_frames, _ = hip.shape
        device = hip.device
        if self.n_seeds >= n_frames:
            return self.seed_inds.repeat(n_batch, 1).to(device)
        if self.seed_sampling == 'random':
            inds = torch.argsort(torch.rand(size=(n_batch, n_frames)), dim=1)[:, :self.n_seeds]
            return torch.sort(inds, dim=1)[0].to(device)
        if self.seed_sampling == 'uniform':   # equal arc length along the hip trajectory
            step = torch.norm(torch.diff(hip, dim=1), dim=2)
            # Arc length accumulated in float64 and rounded once per prefix -- exactly what torch's CPU cumsum
            # of float32 does (the reference's CPU result), whereas a float32 parallel scan on the GPU rounds
            # the prefixes of a plateau (repeated frames: step == 0) differently and breaks their exact ties.
            cum = torch.cumsum(torch.cat([torch.zeros(size=(n_batch, 1), device=device), step], dim=1).double(),
                               dim=1).float()
            stride = cum[:, -1] / (self.n_seeds - 1)
            target = stride.unsqueeze(-1) * torch.arange(self.n_seeds, dtype=torch.float, device=device)
            if cum.is_cuda and cum.dtype == torch.float32 and n_frames <= 40000:
                from .. import seed_op
                return seed_op.nearest_prefix(cum, target)      # same fp32 expression, first minimum, one launch
            return torch.argmin(torch.abs(cum.unsqueeze(-1) - target.unsqueeze(1)), dim=1)
        raise NotImplementedError

    @staticmethod
    def _mlp(seq, x, inner, add_ct=None):
        """Run a `_point_mlp` stack (conv-BN-ReLU, conv-BN-ReLU, conv) on x (B,3,L), L = rows * inner.
        add_ct (B,64,rows): added to the result, broadcast over `inner` (folded into the last layer's kernel).
        On the GPU: the 3->64 layer is a streaming kernel, and each BatchNorm+ReLU is folded into the
        following pointwise 64->64 convolution (one pass per layer instead of GEMM + normalise +
        activate); elsewhere the plain module chain runs."""
        from .. import bn_op, tconv_op, embed_op
        if embed_op.supported(seq, x, inner, add_ct):
            # one autograd function for the stack: forward on the same kernels, backward one pass per layer
            return embed_op.embed_mlp(seq, x, inner, add_ct)
        s0, s1, s2 = seq
        fused = (x.is_cuda and len(seq) == 3 and hasattr(s0, 'batchnorm') and hasattr(s1, 'batchnorm')
                 and not hasattr(s2, 'batchnorm') and tconv_op.supported_embed3(x, s0.conv)
                 and x.shape[2] % inner == 0 and inner <= 64)
        if fused:
            z0s = None
            if s0.batchnorm.training:       # stage-0 statistics from the moments of the three input rows
                z, z0s = tconv_op.embed3(x, s0.conv, want_stats=True)
            else:
                z = tconv_op.embed3(x, s0.conv)
            z = z.view(x.shape[0], 64, x.shape[2] // inner, inner)
            fused = tconv_op.supported_pointwise(z, s0.batchnorm, s1.conv) and \
                tconv_op.supported_pointwise(z, s1.batchnorm, s2.conv)
            if fused:
                if s1.batchnorm.training:   # stage-1 statistics come out of the stage-0 kernel's epilogue
                    z, zs = tconv_op.bn_relu_tconv(z, s0.batchnorm, s1.conv, stats=z0s, want_stats=True)
                    z = tconv_op.bn_relu_tconv(z, s1.batchnorm, s2.conv, stats=zs, add_ct=add_ct)
                else:
                    z = tconv_op.bn_relu_tconv(z, s0.batchnorm, s1.conv)
                    z = tconv_op.bn_relu_tconv(z, s1.batchnorm, s2.conv, add_ct=add_ct)
                return z.view(x.shape[0], 64, x.shape[2])
        for stage in seq:
            if x.is_cuda and hasattr(stage, 'batchnorm') and hasattr(stage, 'ReLU') and \
                    bn_op.supported(x, stage.batchnorm):
                x = bn_op.fused_bn_act(stage.conv(x), stage.batchnorm, None, relu=True)
            else:
                x = stage(x)
        if add_ct is not None:
            x = (x.view(x.shape[0], x.shape[1], -1, inner) + add_ct.unsqueeze(-1)).view(x.shape)
        return x

    def embed(self, input_joints):
        """(B,T,J,3) -> (B,64,T,J): joint embedding + temporal-window position embedding
        (stgcn.py:105-130).  The reference builds the same tensor through
        (B,T,J,64) and a permute; the layouts are produced directly here."""
        n_batch, n_frames, n_joints, _ = input_joints.shape
        device = input_joints.device
        hip = input_joints[:, :, self.origin_joint_id]                       # (B,T,3)
        win = torch.arange(n_frames, device=device).unsqueeze(-1) + \
            torch.arange(-self.knn // 2, self.knn // 2, device=device).unsqueeze(0)
        win = win.clamp_(0, n_frames - 1)                                      # (T,knn)
        offs = hip[:, win] - hip.unsqueeze(2)                                  # (B,T,knn,3)
        pe = self._mlp(self.pos_embed, offs.reshape(n_batch, n_frames * self.knn, 3).transpose(1, 2).contiguous(), self.knn)
        from .. import seed_op
        pe = pe.view(n_batch, -1, n_frames, self.knn)
        fused = seed_op.short_rows_supported(pe)       # short-row reductions as streaming kernels (csrc/seed_ops.hip)
        pe = seed_op.mean_last(pe) if fused else pe.mean(dim=3)               # (B,64,T)
        rel = input_joints - input_joints[:, :, self.origin_joint_id:self.origin_joint_id + 1]
        # the position embedding is added (broadcast over the joints) inside the last layer of the joint embedding
        sk = self._mlp(self.sk_feat, rel.reshape(n_batch, n_frames * n_joints, 3).transpose(1, 2).contiguous(), n_joints,
                       add_ct=pe)
        return sk.view(n_batch, -1, n_frames, n_joints)

    def forward(self, input_joints, end_points=None):
        end_points = {} if end_points is None else end_points
        n_batch, n_frames, n_joints, n_dim = input_joints.size()
        seed_inds = self._select_seeds(input_joints[:, :, self.origin_joint_id])

        x = self.embed(input_joints)
        blocks = self.st_gcn_networks
        tables = blocks[0].gcn.tables
        if tables is not None and tables.gen2 and all(b.chainable(x, self.A) for b in blocks):
            # fused train-mode path: the per-block parameter transforms are computed for all blocks at once
            from ..gcn_op import prepare_chain
            for gcn, prep in zip(blocks, prepare_chain(blocks, self.A, self.edge_importance, tables, frames=x.shape[2])):
                x, _ = gcn(x, prep.Aeff, prepared=prep)
        else:
            for gcn, importance in zip(blocks, self.edge_importance):
                x, _ = gcn(x, self.A * importance)

        seed_skeleton = torch.gather(
            input_joints, 1, seed_inds[:, :, None, None].expand(n_batch, self.n_seeds, n_joints, n_dim))
        if self.n_seeds < n_frames:
            # conv_joint is pointwise in time (kernel 1), so the reference's conv-then-gather
            # (stgcn.py:142-149) equals gather-then-conv: only the seed frames go through the
            # 3392 -> 256 GEMM, and the frame gather doubles as the re-layout to (.., 64*J) rows.
            from .. import seed_op
            if seed_op.seed_rows_supported(x, seed_inds):               # one streaming launch each way
                rows = seed_op.seed_rows(x, seed_inds)                  # (B,S,64*J)
            else:
                rows = x.permute(0, 2, 1, 3)[torch.arange(n_batch, device=x.device)[:, None], seed_inds]   # (B,S,64,J)
            seed_features = torch.nn.functional.linear(                 # (B,S,256); feature order c*J + j as in
                rows.reshape(n_batch, self.n_seeds, -1),                # the reference's (B, 64*J, T) layout
                self.conv_joint.weight.squeeze(-1), self.conv_joint.bias)
        else:
            x = x.transpose(2, 3).reshape(n_batch, -1, n_frames)
            seed_features = self.conv_joint(x).transpose(1, 2)          # (B,T,256)
            seed_features = torch.gather(
                seed_features, 1, seed_inds.unsqueeze(-1).expand(n_batch, self.n_seeds, seed_features.size(-1)))
        end_points['seed_inds'] = seed_inds
        end_points['seed_skeleton'] = seed_skeleton[..., :3]
        end_points['seed_features'] = seed_features
        return end_points
