"""Proposal / detection head (mirror of models/p2rnet/modules/proposal_net.py:15-252).

vote clustering (`PointnetSAModuleVotes` on the HIP ops: FPS -> gather -> ball query ->
grouping -> shared MLP -> max) -> proposals re-ordered by ascending FPS index ->
four point-wise conv stems -> three mixture-density heads -> `decode_scores`.
"""
import numpy as np
import torch
import torch.nn as nn

from ...pointnet2_ops.pointnet2_modules import PointnetSAModuleVotes
from ..config import Struct
from ..registers import MODULES
from .mdn import CategoryEmbeddingMDN
from .sub_modules import SingleConv


USE_FUSED_HEADS = True      # tests switch it off to reach the module chain (nn.Conv1d / nn.BatchNorm1d)


def decode_scores(pred_center, pred_size, pred_heading, sem_obj_feature, end_points):
    """(B,D,K) head outputs -> end_points entries in (B,K,D) layout (proposal_net.py:15-34)."""
    sem_obj = sem_obj_feature.transpose(2, 1)
    end_points['center'] = end_points['aggregated_vote_xyz'] + pred_center.transpose(2, 1)
    end_points['size'] = pred_size.transpose(2, 1)            # log-size
    end_points['heading'] = pred_heading.transpose(2, 1)      # (sin, cos)
    end_points['objectness_scores'] = sem_obj[..., 0:2]
    end_points['sem_cls_scores'] = sem_obj[..., 2:]
    return end_points


def _random_start_fps(xyz, npoint):
    """Pure-torch FPS with a random first point, as used ONLY to initialise the
    mixture means (net_utils/libs.py:152-173 of the reference); consumes the torch
    RNG the same way (one randint of shape (B,))."""
    B, N, _ = xyz.shape
    centroids = torch.zeros(B, npoint, dtype=torch.long)
    distance = torch.full((B, N), 1e10, dtype=xyz.dtype)
    farthest = torch.randint(0, N, (B,), dtype=torch.long)
    rows = torch.arange(B, dtype=torch.long)
    for i in range(npoint):
        centroids[:, i] = farthest
        d = torch.sum((xyz - xyz[rows, farthest, :].view(B, 1, 3)) ** 2, -1)
        distance = torch.where(d < distance, d, distance)
        farthest = torch.max(distance, -1)[1]
    return centroids


def _stem(cin, cout):
    return nn.Sequential(SingleConv(cin, 128, kernel_size=1, order='cbr', num_groups=8, padding=0, ndim=1),
                         SingleConv(128, cout, kernel_size=1, order='cbr', num_groups=8, padding=0, ndim=1))


@MODULES.register_module
class ProposalNet(nn.Module):
    def __init__(self, cfg, optim_spec=None):
        super().__init__()
        self.optim_spec = optim_spec
        self.cfg = cfg
        self.num_class = cfg.dataset_config.num_class
        self.num_proposals = cfg.config['data']['num_target']
        self.sampling = cfg.config['data']['cluster_sampling']
        vote_dim = 256
        if cfg.config['mode'] != 'train':
            self.multi_mode = cfg.eval_config['multi_mode']
            self.n_samples = np.random.choice(np.arange(1, 100), 1)[0]   # proposal_net.py:56-59

        self.vote_aggregation = PointnetSAModuleVotes(
            npoint=self.num_proposals, radius=0.3, nsample=16, mlp=[256, 256, vote_dim],
            use_xyz=False, normalize_xyz=True, bn=False)

        sem_obj_dim = 2 + self.num_class   # objectness (2) + classes
        gmm_dim = 128
        self.conv_center = _stem(vote_dim, gmm_dim)
        self.conv_heading = _stem(vote_dim, gmm_dim)
        self.conv_size = _stem(vote_dim, gmm_dim)
        self.conv_sem_obj = nn.Sequential(
            SingleConv(vote_dim, 128, kernel_size=1, order='cbr', num_groups=8, padding=0, ndim=1),
            SingleConv(128, 128, kernel_size=1, order='cbr', num_groups=8, padding=0, ndim=1),
            SingleConv(128, sem_obj_dim, kernel_size=1, order='c', num_groups=8, padding=0, ndim=1))

        G = cfg.config['data']['num_gaussian']
        self.gmm_center = self.load_gmm(G, gmm_dim, 3, 'center')
        self.gmm_size = self.load_gmm(G, gmm_dim, 3, 'size')
        self.gmm_heading = self.load_gmm(G, gmm_dim, 2, 'heading')

    # -- mixture-mean initialisation (proposal_net.py:96-134) --------------------
    def init_mu(self, num_gaussian, type):
        if type == 'center':       # points on a sphere of radius 0.1
            n_theta = np.ceil(np.sqrt(num_gaussian / 2)).astype(np.uint16)
            n_phi = 2 * n_theta
            width = np.pi / n_theta
            phi = [width * i - np.pi for i in range(0, n_phi)]
            theta = np.linspace(0, np.pi, n_theta + 2)[1:-1]
            grid = np.array(np.meshgrid(phi, theta)).reshape(2, -1).T
            pts = np.hstack([0.1 * np.sin(grid[:, [1]]) * np.cos(grid[:, [0]]),
                             0.1 * np.sin(grid[:, [1]]) * np.sin(grid[:, [0]]),
                             0.1 * np.cos(grid[:, [1]])])
            mu = torch.from_numpy(pts)
            if num_gaussian < mu.size(0):
                mu = self.get_farthest_points(mu, npoint=num_gaussian)
            return mu
        if type == 'size':         # log of a cubic grid of sizes in [0.05, 3]
            per_dim = np.ceil(num_gaussian ** (1 / 3)).astype(np.uint32)
            ticks = np.linspace(0.05, 3, per_dim)
            grid = np.log(np.array(np.meshgrid(ticks, ticks, ticks)).reshape(3, -1).T)
            return self.get_farthest_points(torch.from_numpy(grid), npoint=num_gaussian)
        if type == 'heading':      # unit circle, float64 (stays float64 in the state_dict)
            width = 2 * np.pi / num_gaussian
            thetas = [width * i - np.pi for i in range(0, num_gaussian)]
            return torch.from_numpy(np.array([[np.sin(t), np.cos(t)] for t in thetas]))
        return None

    def get_farthest_points(self, xyz, npoint):
        if xyz.dim() == 2:
            xyz = xyz.unsqueeze(0)
        xyz = xyz.float()
        inds = torch.sort(_random_start_fps(xyz, npoint), dim=-1)[0]
        out = torch.gather(xyz, 1, inds.unsqueeze(-1).expand(-1, -1, xyz.size(-1)))
        return out.squeeze(0) if out.size(0) == 1 else out

    def load_gmm(self, num_gaussian, in_dim, out_dim, type):
        mdn_config = Struct(num_gaussian=num_gaussian, out_dim=out_dim,
                            mu_bias_init=self.init_mu(num_gaussian, type), n_samples=1,
                            central_tendency='mean')
        config = Struct(embedding_dims=[], out_dim=3, continuous_dim=in_dim,
                        batch_norm_continuous_input=False, hidden_dim=128, mdn_config=mdn_config)
        return CategoryEmbeddingMDN(config)

    # -- shared front: cluster votes into proposals (proposal_net.py:158-178) ------
    def _aggregate(self, xyz, features, end_points):
        features = features.transpose(1, 2).contiguous()
        if self.sampling == 'vote_fps':
            xyz, features, fps_inds = self.vote_aggregation(xyz, features)
            sample_inds, order = torch.sort(fps_inds, dim=-1)
            xyz = torch.gather(xyz, 1, order.unsqueeze(-1).expand(-1, -1, xyz.size(2)))
            features = torch.gather(features, 2, order.unsqueeze(1).expand(-1, features.size(1), -1))
        elif self.sampling == 'seed_fps':
            # the reference reads end_points['seed_xyz'], which no module sets (dead branch there)
            seed_xyz = end_points['seed_xyz']
            step = torch.norm(torch.diff(seed_xyz, dim=1), dim=2)
            cum = torch.cumsum(torch.cat([torch.zeros(size=(xyz.shape[0], 1)).to(xyz.device), step], dim=1), dim=1)
            stride = cum[:, -1] / (self.num_proposals - 1)
            target = stride.unsqueeze(-1) * torch.arange(self.num_proposals, dtype=torch.float).to(xyz.device)
            sample_inds = torch.argmin(torch.abs(cum.unsqueeze(-1) - target.unsqueeze(1)), dim=1).type(torch.int32)
            xyz, features, _ = self.vote_aggregation(xyz, features, sample_inds)
        else:
            raise NotImplementedError('Undefined sampling strategy.')
        end_points['aggregated_vote_xyz'] = xyz
        end_points['aggregated_vote_inds'] = sample_inds.type(torch.int64)
        return features

    def forward(self, xyz, features, end_points, export_proposal_feature=False, eps=None):
        """xyz (B,S,3), features (B,S,C).  `eps` = optional dict(center=, size=, heading=)
        of explicit mixture noise (see mdn.py)."""
        features = self._aggregate(xyz, features, end_points)
        eps = eps or {}
        from .. import pw_op
        if USE_FUSED_HEADS and pw_op.proposal_heads_supported(self, features):
            # the four stems, the mixture backbones / pi convolutions / read-outs and conv_sem_obj on the job-list
            # kernels of csrc/pw_layers.hip (10 launches; same parameters, same noise draws in the same order)
            pred_center, pred_size, pred_heading, sem_obj_feature = pw_op.proposal_heads(self, features, eps)
        else:
            pred_center = self.gmm_center.predict(self.conv_center(features), eps=eps.get('center'))
            pred_size = self.gmm_size.predict(self.conv_size(features), eps=eps.get('size'))
            pred_heading = self.gmm_heading.predict(self.conv_heading(features), eps=eps.get('heading'))
            sem_obj_feature = self.conv_sem_obj(features)
        end_points = decode_scores(pred_center, pred_size, pred_heading, sem_obj_feature, end_points)
        return end_points, (features.transpose(1, 2).contiguous() if export_proposal_feature else None)

    def generate(self, xyz, features, end_points, export_proposal_feature=False):
        features = self._aggregate(xyz, features, end_points)
        from .. import pw_op
        if USE_FUSED_HEADS and not self.multi_mode and pw_op.proposal_heads_supported(self, features):
            pred_center, pred_size, pred_heading, sem_obj_feature, (pi_center, pi_size, pi_heading) = \
                pw_op.proposal_heads(self, features, False, return_pi=True)
        else:
            kw = dict(return_pi=True, multi_modes=self.multi_mode, n_samples=self.n_samples)
            pred_center, pi_center = self.gmm_center.generate(self.conv_center(features), **kw)
            pred_size, pi_size = self.gmm_size.generate(self.conv_size(features), **kw)
            pred_heading, pi_heading = self.gmm_heading.generate(self.conv_heading(features), **kw)
            sem_obj_feature = self.conv_sem_obj(features)
        end_points = decode_scores(pred_center, pred_size, pred_heading, sem_obj_feature, end_points)
        end_points['pi'] = {'center': pi_center, 'size': pi_size, 'heading': pi_heading}
        return end_points, (features.transpose(1, 2).contiguous() if export_proposal_feature else None)
