"""Mixture-density heads (mirror of models/p2rnet/modules/mdn.py:17-161).

pi = sigmoid(Conv1d(x)) over G Gaussians with learnable (mu, log_sigma) shared by
all proposals.  Training prediction = sum_g pi_g * (mu_g + sigma_g * eps_g) with
one standard-normal draw per (proposal, component, sample) (mdn.py:34-83);
evaluation = sum_g pi_g * mu_g (mdn.py:85-99).

`noise` hook: every sampling entry point accepts an explicit `eps` tensor of
shape (B*L, G, n_samples, D); when omitted it is drawn exactly like the
reference does (`new(...).normal_()` on the parameter's device/dtype), so a
recorded eps reproduces a reference step bit-for-bit in structure.
"""
from typing import Optional

import torch
from torch import nn
from torch.distributions.bernoulli import Bernoulli

from .sub_modules import SingleConv


class MixtureDensityHead(nn.Module):
    def __init__(self, config, **kwargs):
        super().__init__()
        self.hparams = config
        self.pi = SingleConv(config.input_dim, config.num_gaussian, kernel_size=1, order='c',
                             padding=0, ndim=1)
        self.log_sigma = nn.Parameter(torch.zeros(config.num_gaussian, config.out_dim))
        self.mu = nn.Parameter(config.mu_bias_init)
        self.noise_hook = None   # callable(shape, like) -> eps, used by parity tests

    def forward(self, x):
        return torch.sigmoid(self.pi(x))

    def _eps(self, n_rows, num_samples):
        shape = (n_rows, self.mu.size(0), num_samples, self.mu.size(1))
        if self.noise_hook is not None:
            return self.noise_hook(shape, self.mu)
        return self.mu.data.new_empty(shape).normal_()

    def sample(self, num_samples, n_batch, eps=None):
        """(n_batch, G, num_samples, D) draws from the G components."""
        sigma = torch.exp(self.log_sigma)[None, :, None, :]
        mu = self.mu[None, :, None, :]
        if eps is None:
            eps = self._eps(n_batch, num_samples)
        return eps * sigma + mu

    def generate_samples(self, pi, n_samples=None, sample_pi=False, eps=None):
        if n_samples is None:
            n_samples = self.hparams.n_samples
        n_batch, _, length = pi.size()
        pi_r = pi.transpose(1, 2).contiguous().view(n_batch * length, -1)        # (B*L, G)
        samples = self.sample(n_samples, pi_r.size(0), eps=eps)
        if sample_pi:
            gate = Bernoulli(pi_r).sample((n_samples,)).permute(1, 2, 0).unsqueeze(-1)  # (B*L,G,n,1)
        else:
            gate = pi_r[:, :, None, None]
        samples = torch.sum(samples * gate, dim=1)                               # (B*L, n, D)
        samples = samples.view(n_batch, length, n_samples, -1)
        return samples.transpose(1, 3).contiguous()                              # (B, D, n, L)

    def generate_point_predictions(self, pi, n_samples=None, sample_pi=False, eps=None):
        samples = self.generate_samples(pi, n_samples, sample_pi=sample_pi, eps=eps)
        if self.hparams.central_tendency == "mean":
            return torch.mean(samples, dim=2)
        if self.hparams.central_tendency == "median":
            return torch.median(samples, dim=2).values
        raise NotImplementedError

    def get_mean(self, pi):
        n_batch, _, length = pi.size()
        pi_r = pi.transpose(1, 2).contiguous().view(n_batch * length, -1)
        out = torch.sum(self.mu.unsqueeze(0) * pi_r.unsqueeze(-1), dim=1)         # (B*L, D)
        return out.view(n_batch, length, -1).transpose(1, 2).contiguous()


class BaseMDN(nn.Module):
    def __init__(self, config, **kwargs):
        super().__init__()
        self.config = config

    def forward(self, x):
        return self.mdn(self.backbone(self.unpack_input(x)))

    def predict(self, x, eps=None):
        return self.mdn.generate_point_predictions(self.forward(x), eps=eps)

    def generate(self, x, return_pi=False, multi_modes=False, n_samples=10):
        pi = self.forward(x)
        if multi_modes:
            pred = self.mdn.generate_point_predictions(pi, n_samples=n_samples, sample_pi=True)
        else:
            pred = self.mdn.get_mean(pi)
        return (pred, pi) if return_pi else pred

    def sample(self, x, n_samples: Optional[int] = None, ret_model_output=False):
        pi = self.forward(x)
        samples = self.mdn.generate_samples(pi, n_samples)
        return (samples, pi) if ret_model_output else samples

    def test_step(self, x, y):
        return self.mdn.generate_point_predictions(self(x)), y


class CategoryEmbeddingMDN(BaseMDN):
    def __init__(self, config, **kwargs):
        super().__init__(config, **kwargs)
        self.hparams = config
        if config.batch_norm_continuous_input:
            self.normalizing_batch_norm = nn.BatchNorm1d(config.continuous_dim)
        self.backbone = SingleConv(config.continuous_dim, config.hidden_dim, kernel_size=1, order='cbr',
                                   padding=0, ndim=1)
        config.mdn_config.update(input_dim=config.hidden_dim)
        self.mdn = MixtureDensityHead(config.mdn_config)

    def unpack_input(self, x):
        if self.hparams.batch_norm_continuous_input:
            x = self.normalizing_batch_norm(x)
        return x
