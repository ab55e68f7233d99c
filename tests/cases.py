"""Seeded input builders shared by the CPU (oracle/golden) and GPU (parity) tests."""
import numpy as np
import torch


def cloud(b, n, seed, kind="uniform"):
    g = torch.Generator().manual_seed(seed)
    if kind == "uniform":
        return (torch.rand(b, n, 3, generator=g) * 4 - 2).contiguous()
    if kind == "lattice":      # exact ties everywhere: integer grid, small range
        return torch.randint(-3, 4, (b, n, 3), generator=g).float().contiguous()
    if kind == "halflattice":  # ties + exactly representable distances
        return (torch.randint(-8, 9, (b, n, 3), generator=g).float() * 0.25).contiguous()
    if kind == "near_origin":  # many points inside the |p|^2 <= 1e-3 skip ball
        x = torch.rand(b, n, 3, generator=g) * 4 - 2
        m = torch.rand(b, n, generator=g) < 0.3
        x[m] = x[m] * 0.01
        # a few exactly on the threshold neighbourhood
        x[:, 1 % n] = torch.tensor([0.031622776, 0.0, 0.0])
        return x.contiguous()
    if kind == "all_skipped":
        return (torch.rand(b, n, 3, generator=g) * 0.01).contiguous()
    if kind == "duplicates":
        base = torch.rand(b, max(n // 4, 1), 3, generator=g) * 4 - 2
        idx = torch.randint(0, base.shape[1], (b, n), generator=g)
        return torch.gather(base, 1, idx.unsqueeze(-1).expand(b, n, 3)).contiguous()
    if kind == "walk":         # P2RNet-like: random-walk trajectory (vote_xyz-like clusters)
        steps = torch.randn(b, n, 3, generator=g) * 0.05
        return (torch.cumsum(steps, 1) + torch.tensor([0.0, 0.9, 0.0])).contiguous()
    raise ValueError(kind)


FPS_CASES = [  # (b, n, m, kind, seed)
    (2, 512, 128, "uniform", 1), (4, 512, 128, "walk", 2), (3, 64, 16, "uniform", 3),
    (2, 100, 30, "uniform", 4), (2, 1, 1, "uniform", 5), (2, 5, 5, "lattice", 6),
    (2, 512, 128, "lattice", 7), (2, 300, 300, "halflattice", 8), (2, 512, 64, "near_origin", 9),
    (1, 128, 16, "all_skipped", 10), (2, 256, 100, "duplicates", 11), (1, 1000, 200, "uniform", 12),
    (1, 2048, 256, "lattice", 13), (1, 5000, 128, "uniform", 14), (1, 9000, 64, "uniform", 15),
    (1, 16384, 32, "lattice", 16), (1, 20000, 48, "uniform", 17), (2, 33, 7, "lattice", 18),
    (1, 700, 64, "lattice", 19), (1, 3000, 64, "halflattice", 20),
    # beyond one workgroup's register file: several workgroups per cloud
    (8, 54272, 40, "uniform", 21), (3, 17000, 33, "lattice", 22), (16, 16500, 20, "halflattice", 23),
    (40, 16400, 6, "uniform", 24), (2, 70000, 300, "duplicates", 25),
]

BALL_CASES = [  # (b, n, m, radius, nsample, kind, seed)
    (2, 512, 128, 0.3, 16, "walk", 1), (2, 512, 128, 0.3, 16, "uniform", 2),
    (3, 100, 17, 0.8, 8, "uniform", 3), (2, 64, 64, 1.0, 32, "lattice", 4),
    (2, 300, 50, 0.5, 16, "halflattice", 5), (1, 3000, 40, 0.4, 16, "uniform", 6),
    (1, 5000, 100, 0.25, 64, "halflattice", 7), (2, 10, 5, 0.01, 4, "uniform", 8),
    (1, 512, 128, 5.0, 100, "uniform", 9), (2, 1, 1, 1.0, 3, "uniform", 10),
]


def centres_from(xyz, m, seed):
    g = torch.Generator().manual_seed(1000 + seed)
    b, n, _ = xyz.shape
    idx = torch.stack([torch.randperm(n, generator=g)[:m] if m <= n else torch.randint(0, n, (m,), generator=g)
                       for _ in range(b)])
    return torch.gather(xyz, 1, idx.unsqueeze(-1).expand(b, m, 3)).contiguous()


def random_boxes(K, seed, stride=8, ncls=3):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-3, 3, (K, 3))
    s = rng.uniform(0.2, 2.0, (K, 3))
    score = (rng.permutation(K) + rng.uniform(0.1, 0.9, K)) / max(K, 1)   # distinct
    cls = rng.integers(0, ncls, (K, 1)).astype(np.float64)
    b = np.concatenate([c - s / 2, c + s / 2, score[:, None], cls], 1)
    return np.ascontiguousarray(b[:, :stride])


def boxes_xz(boxes):
    """(K,>=7) rows [x1,y1,z1,x2,y2,z2,score,...] -> (K,5) rows [x1,z1,x2,z2,score]: the 2-D boxes the reference's
    `use_3d_nms: False` branch builds (ap_helper.py:201-207)."""
    return np.ascontiguousarray(boxes[:, [0, 2, 3, 5, 6]])


def _hash_uniform(n, k):
    """n deterministic pseudo-random numbers in [-1, 1) from integer arithmetic only
    (exact on every platform / torch version): a multiply-xorshift hash of (index, key)."""
    M = (1 << 32) - 1
    h = (torch.arange(n, dtype=torch.int64) * 2654435761 + (k + 1) * 2246822519) & M
    h = h ^ (h >> 15)
    h = (h * 1540483477) & M
    h = h ^ (h >> 13)
    h = (h * 1103515245 + 12345) & M
    h = h ^ (h >> 16)
    return h.double() / float(1 << 31) - 1.0


def fill_weights(net):
    """Deterministic weight rule applied identically to the reference model (when the
    fixtures are generated) and to ours (when they are checked): every floating tensor of
    the state_dict is filled from an integer hash of (element index, key index), with
    He-style fan-in scaling for conv weights so the net behaves like a freshly initialised
    one (a first version used sines of the element index: those weights are orthogonal to
    the smooth pose signal, the convs cancel it, and train-mode BatchNorm then amplifies
    fp32 rounding differences ~4000x -- measured -- which made the fixtures ill-conditioned).
    The adjacency buffer `A` and integer buffers are left alone."""
    sd = net.state_dict()
    with torch.no_grad():
        for k, name in enumerate(sorted(sd.keys())):
            t = sd[name]
            if not t.is_floating_point() or name == 'backbone.A':
                continue
            u = _hash_uniform(t.numel(), k)          # uniform [-1,1): std 0.577
            if name.endswith('running_var'):
                v = 1.0 + 0.5 * u.abs()
            elif name.endswith('running_mean'):
                v = 0.1 * u
            elif 'batchnorm.weight' in name or name.endswith('tcn.0.weight') or name.endswith('tcn.3.weight'):
                v = 1.0 + 0.1 * u
            elif 'edge_importance' in name:
                v = 1.0 + 0.1 * u
            elif name.endswith('log_sigma'):
                v = 0.1 * u - 1.0
            elif name.endswith('mdn.mu'):
                v = 0.5 * u
            elif name.endswith('bias'):
                v = 0.1 * u
            elif t.dim() > 1:
                v = u * (2.4 / t[0].numel() ** 0.5)  # std = 1.39/sqrt(fan_in) ~ He init
            else:
                v = 0.1 * u
            t.copy_(v.view_as(t).to(t.dtype))
    return net


def state_checksum(net):
    sd = net.state_dict()
    return float(sum(v.double().abs().sum() for k, v in sd.items() if v.is_floating_point()))


def seam_cotangents(B, seed=77):
    """Seeded cotangents on (vote_xyz, vote_features) at the backbone -> detection seam: fixture g10c back-propagates
    these through the reference's backbone, the GPU test through ours."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(B, 512, 3, generator=g), torch.randn(B, 512, 256, generator=g) * 0.1
