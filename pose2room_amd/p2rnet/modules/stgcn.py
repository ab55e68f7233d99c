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
        for blk in self.st_gcn_networks:
            blk.gcn.tables = tables
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
            cum = torch.cumsum(torch.cat([torch.zeros(size=(n_batch, 1)).to(device), step], dim=1), dim=1)
            stride = cum[:, -1] / (self.n_seeds - 1)
            target = stride.unsqueeze(-1) * torch.arange(self.n_seeds, dtype=torch.float).to(device)
            return torch.argmin(torch.abs(cum.unsqueeze(-1) - target.unsqueeze(1)), dim=1)
        raise NotImplementedError

    @staticmethod
    def _mlp(seq, x):
        """Run a `_point_mlp` stack; on the GPU each conv+BatchNorm+ReLU stage uses the fused
        BatchNorm/ReLU kernels instead of separate normalisation and activation passes."""
        from .. import bn_op
        for stage in seq:
            if x.is_cuda and hasattr(stage, 'batchnorm') and hasattr(stage, 'ReLU') and \
                    bn_op.supported(x, stage.batchnorm):
                x = bn_op.fused_bn_act(stage.conv(x), stage.batchnorm, None, relu=True)
            else:
                x = stage(x)
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
        pe = self._mlp(self.pos_embed, offs.reshape(n_batch, n_frames * self.knn, 3).transpose(1, 2))
        pe = pe.view(n_batch, -1, n_frames, self.knn).mean(dim=3)             # (B,64,T)
        rel = input_joints - input_joints[:, :, [self.origin_joint_id]]
        sk = self._mlp(self.sk_feat, rel.reshape(n_batch, n_frames * n_joints, 3).transpose(1, 2))
        return sk.view(n_batch, -1, n_frames, n_joints) + pe.unsqueeze(-1)

    def forward(self, input_joints, end_points=None):
        end_points = {} if end_points is None else end_points
        n_batch, n_frames, n_joints, n_dim = input_joints.size()
        seed_inds = self._select_seeds(input_joints[:, :, self.origin_joint_id])

        x = self.embed(input_joints)
        for gcn, importance in zip(self.st_gcn_networks, self.edge_importance):
            x, _ = gcn(x, self.A * importance)

        x = x.transpose(2, 3).reshape(n_batch, -1, n_frames)   # (B, 64*J, T), channel-major like the reference
        seed_features = self.conv_joint(x).transpose(1, 2)      # (B,T,256)

        seed_skeleton = torch.gather(
            input_joints, 1, seed_inds[:, :, None, None].expand(n_batch, self.n_seeds, n_joints, n_dim))
        seed_features = torch.gather(
            seed_features, 1, seed_inds.unsqueeze(-1).expand(n_batch, self.n_seeds, seed_features.size(-1)))
        end_points['seed_inds'] = seed_inds
        end_points['seed_skeleton'] = seed_skeleton[..., :3]
        end_points['seed_features'] = seed_features
        return end_points
