"""Seeded synthetic P2RNet batches: the reference's `.hdf5`
samples are not available, so benchmarks and tests use pose sequences with the
statistics of the demo clip and the batch-dict contract of the reference loader
(models/p2rnet/dataloader.py:138-146):

  input_joints (B,T,53,3) f32      box_label_mask (B,10) f32 (prefix of ones)
  sem_cls_label (B,10) i64         center_label (B,10,3) f32 (zeros when padded)
  size (B,10,3) f32 = log size     heading (B,10,2) f32 = (sin, cos)
  vote_label (B,T,53,9) f32        vote_label_mask (B,T,53) i64
"""
import math

import torch

MAX_GT = 10
N_JOINTS = 53
N_CLASS = 22
CONTACT_DIST = 1.0   # configs/dataset_config.py:56


def skeleton_template():
    g = torch.Generator().manual_seed(7)
    t = torch.randn(N_JOINTS, 3, generator=g) * 0.3
    t[0] = 0.0
    return t


def make_batch(batch_size, num_frames, seed=1234, rank=0, device=None):
    g = torch.Generator().manual_seed(seed + rank)
    B, T = batch_size, num_frames
    steps = torch.zeros(B, T, 3)
    steps[:, 1:, 0] = torch.randn(B, T - 1, generator=g) * 0.05
    steps[:, 1:, 2] = torch.randn(B, T - 1, generator=g) * 0.05
    hip = torch.cumsum(steps, 1).clamp_(-3.0, 3.0)
    hip[..., 1] = 0.9
    # articulated motion: every joint swings around its template offset with its own
    # frequency / phase / direction (0.15 m amplitude), plus 2 cm sensor jitter
    tt = torch.arange(T, dtype=torch.float32)[None, :, None]
    freq = torch.rand(B, 1, N_JOINTS, generator=g) * 0.25 + 0.05
    phase = torch.rand(B, 1, N_JOINTS, generator=g) * (2 * math.pi)
    axis = torch.randn(B, 1, N_JOINTS, 3, generator=g)
    axis = axis / axis.norm(dim=-1, keepdim=True)
    swing = 0.15 * torch.sin(freq * tt + phase)[..., None] * axis
    joints = hip[:, :, None, :] + skeleton_template()[None, None] + swing + \
        torch.randn(B, T, N_JOINTS, 3, generator=g) * 0.02
    joints[:, :, 0] = hip

    n_obj = torch.randint(1, MAX_GT + 1, (B,), generator=g)
    mask = (torch.arange(MAX_GT)[None, :] < n_obj[:, None]).float()
    centre = torch.empty(B, MAX_GT, 3)
    centre[..., 0] = torch.rand(B, MAX_GT, generator=g) * 6 - 3
    centre[..., 2] = torch.rand(B, MAX_GT, generator=g) * 6 - 3
    centre[..., 1] = torch.rand(B, MAX_GT, generator=g) * 1.3 + 0.2
    size = torch.rand(B, MAX_GT, 3, generator=g) * 1.7 + 0.3
    theta = (torch.rand(B, MAX_GT, generator=g) * 2 - 1) * math.pi
    cls = torch.randint(0, N_CLASS, (B, MAX_GT), generator=g)

    # votes: joints within CONTACT_DIST of an object's AABB vote for its centre (x3)
    lo = (centre - size / 2)[:, None, None]            # (B,1,1,G,3)
    hi = (centre + size / 2)[:, None, None]
    p = joints[:, :, :, None, :]                       # (B,T,J,1,3)
    gap = torch.maximum(torch.maximum(lo - p, p - hi), torch.zeros(()))
    dist = gap.norm(dim=-1)                            # (B,T,J,G)
    dist = dist.masked_fill(mask[:, None, None, :] == 0, float('inf'))
    near, which = dist.min(dim=-1)
    vote_mask = (near < CONTACT_DIST).long()
    target = torch.gather(centre[:, None, None].expand(B, T, N_JOINTS, MAX_GT, 3), 3,
                          which[..., None, None].expand(B, T, N_JOINTS, 1, 3)).squeeze(3)
    offset = (target - joints) * vote_mask[..., None].float()
    vote_label = offset.repeat(1, 1, 1, 3)

    m3 = mask[..., None]
    batch = {
        'input_joints': joints.float().contiguous(),
        'box_label_mask': mask,
        'sem_cls_label': (cls * mask.long()),
        'center_label': (centre * m3).contiguous(),
        'size': (torch.log(size) * m3).contiguous(),
        'heading': (torch.stack([torch.sin(theta), torch.cos(theta)], -1) * m3).contiguous(),
        'vote_label': vote_label.contiguous(),
        'vote_label_mask': vote_mask,
        'sample_idx': [f'synthetic_{seed}_{rank}_{i}' for i in range(B)],
    }
    if device is not None:
        batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
    return batch
