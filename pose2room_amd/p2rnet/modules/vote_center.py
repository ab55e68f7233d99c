"""Centre voting head (mirror of models/p2rnet/modules/vote_center.py:11-59):
three point-wise convs turn each seed feature into an xyz offset from the seed's
hip joint plus a residual feature."""
import torch.nn as nn

from ..registers import MODULES
from .sub_modules import SingleConv


USE_FUSED_HEAD = True       # tests switch it off to reach the module chain


@MODULES.register_module
class CenterVoteModule(nn.Module):
    def __init__(self, cfg, optim_spec=None):
        super().__init__()
        self.optim_spec = optim_spec
        self.origin_joint_id = cfg.dataset_config.origin_joint_id
        self.vote_factor = cfg.config['data']['vote_factor']
        in_dim = 256
        self.out_dim = in_dim   # residual connection: in == out
        self.conv_input = nn.Sequential(
            SingleConv(in_dim, 256, kernel_size=1, order='cbr', num_groups=8, padding=0, ndim=1),
            SingleConv(256, 256, kernel_size=1, order='cbr', num_groups=8, padding=0, ndim=1),
            SingleConv(256, (3 + self.out_dim) * self.vote_factor, kernel_size=1, order='c',
                       num_groups=8, padding=0, ndim=1))

    def forward(self, seed_xyz, seed_features):
        """seed_xyz (B,S,J,3), seed_features (B,S,C) -> vote_xyz (B,S*vf,3), vote_features (B,S*vf,C)."""
        hip = seed_xyz[:, :, self.origin_joint_id]
        b, s = hip.shape[0], hip.shape[1]
        from .. import pw_op
        if USE_FUSED_HEAD and pw_op.vote_head_supported(self, seed_features):
            net = pw_op.vote_head(self, seed_features)      # csrc/pw_layers.hip: one launch per layer
        else:
            net = self.conv_input(seed_features.transpose(1, 2))
        net = net.transpose(2, 1).view(b, s, self.vote_factor, 3 + self.out_dim)
        vote_xyz = (hip.unsqueeze(2) + net[..., 0:3]).contiguous().view(b, s * self.vote_factor, 3)
        vote_features = (seed_features.unsqueeze(2) + net[..., 3:]).contiguous()
        return vote_xyz, vote_features.view(b, s * self.vote_factor, self.out_dim).contiguous()
