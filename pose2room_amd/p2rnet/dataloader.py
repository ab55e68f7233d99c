"""Sample loading for P2RNet: on-disk formats -> the batch dict the network consumes.

Mirror of the reference's models/p2rnet/dataloader.py (`P2RNet_VirtualHome`, `collate_fn`,
`P2RNet_dataloader`) and of the demo input path (demo.py:23-51 `Demo_DataSet`):

  * a training / test sample is an .hdf5 file with `skeleton_joints (T0,53,3)`,
    `skeleton_joint_votes (T0,53,10)` (column 0 = vote mask, 1..9 = three 3-vector votes) and
    `object_nodes/<id>/{class_id, centroid, R_mat, size}` (utils/virtualhome/3_generate_samples.py:176-193);
  * a demo input is a bare `(T0,53,3)` .npy pose sequence.

Frames are re-sampled to `num_frames` by index repetition (`np.linspace(0, T0-1, num_frames).round()`,
dataloader.py:131), boxes become `(centroid, log size, sin/cos heading)` padded to `max_gt_boxes`, and the
training augmentation is flip (x<->z) / rotation by a multiple of 90 degrees about y / x-z translation.

The numeric part is split from the file access so it can be pinned against the reference without the dataset
(`sample_to_tensors`, `augment_sample`; tests/golden/make_loader_golden.py).  Reading .hdf5 needs `h5py`,
which this image does not ship: `read_sample_hdf5` raises a clear ImportError then; nothing here is on the
timed path (bench.py feeds synthetic batches of the same layout, p2rnet/synthetic.py).
"""
import json
import os
import random

import numpy as np
import torch
import torch.utils.data
from torch.utils.data import DataLoader, Dataset
from torch.utils.data.distributed import DistributedSampler

FLIP_MATRIX = np.array([[0, 0, 1], [0, 1, 0], [1, 0, 0]])        # swaps x and z (dataloader.py:25)


def rot_y(theta):
    """dataloader.py:26-28."""
    return np.array([[np.cos(theta), 0., -np.sin(theta)], [0., 1., 0.], [np.sin(theta), 0, np.cos(theta)]])


def rot2head(R_mat):
    """utils/pc_utils.py:34-48: heading angle of a rotation matrix whose first row is the heading vector."""
    R_mat = np.array(R_mat)
    single = R_mat.ndim == 2
    if single:
        R_mat = R_mat[np.newaxis]
    heading = np.arctan2(-R_mat[:, 0, 2], R_mat[:, 0, 0])
    return heading[0] if single else heading


def resample_frames(n_src, num_frames):
    """Frame indices that stretch / thin a sequence of n_src frames to num_frames (dataloader.py:131)."""
    return np.linspace(0, n_src - 1, num_frames).round().astype(np.uint16)


def draw_augmentation():
    """The three random draws of `augment_data` in the reference's order and from the same generators
    (dataloader.py:33-35): python `random` for the flip and the offset, `np.random` for the angle."""
    if_flip = random.randint(0, 1)
    rot_angle = np.random.choice([-np.pi, -0.5 * np.pi, 0, 0.5 * np.pi])
    offset_scale = random.uniform(-1., 1.)
    return if_flip, rot_angle, offset_scale


def augment_sample(skeleton_joints, instances, skeleton_joint_votes, if_flip, rot_angle, offset_scale):
    """dataloader.py:31-80 with the draws passed in.  Returns new arrays / a new instance list; votes keep
    pointing at the same (transformed) object centres."""
    rot_mat = rot_y(rot_angle)
    offset = np.array([1., 0., 1.]) * offset_scale
    votes = np.array(skeleton_joint_votes, copy=True)
    joints = np.array(skeleton_joints, copy=True)
    nodes = [dict(class_id=n['class_id'], centroid=np.array(n['centroid']), R_mat=np.array(n['R_mat']),
                  size=np.array(n['size'])) for n in instances]
    n_frames, n_joints = votes.shape[:2]
    if if_flip:
        joints = np.dot(joints, FLIP_MATRIX)
        v = votes[..., 1:].reshape(n_frames, n_joints, 3, 3)
        votes[..., 1:] = np.dot(v, FLIP_MATRIX).reshape(n_frames, n_joints, 9)
        for node in nodes:
            node['centroid'] = np.dot(node['centroid'], FLIP_MATRIX)
            R = node['R_mat'].dot(FLIP_MATRIX)
            R[2] = np.cross(R[0], R[1])                    # keep the frame right-handed
            node['R_mat'] = R
    # rotate: vote end points and joints turn together, votes are re-expressed relative to the new joints
    ends = [np.dot(joints[..., 0:3] + votes[..., 1 + 3 * i:4 + 3 * i], rot_mat) for i in range(3)]
    joints = np.dot(joints, rot_mat)
    for i in range(3):
        votes[..., 1 + 3 * i:4 + 3 * i] = ends[i] - joints[..., 0:3]
    for node in nodes:
        node['centroid'] = np.dot(node['centroid'], rot_mat)
        node['R_mat'] = node['R_mat'].dot(rot_mat)
    # translate
    joints = joints + offset
    for node in nodes:
        node['centroid'] = node['centroid'] + offset
    return joints, nodes, votes


def sample_to_tensors(skeleton_joints, skeleton_joint_votes, instances, num_frames, max_num_obj=10,
                      use_height=False, sample_idx=''):
    """dataloader.py:100-147: one (possibly augmented) sample -> the per-sample dict of NumPy arrays."""
    boxes3D, classes = [], []
    for inst in instances:
        heading = rot2head(inst['R_mat'])
        boxes3D.append(np.hstack([inst['centroid'], np.log(inst['size']), np.sin(heading), np.cos(heading)]))
        classes.append(inst['class_id'])
    boxes3D = np.array(boxes3D)
    if use_height:
        floor_height = np.percentile(skeleton_joints[..., 1], 0.99)
        height = skeleton_joints[..., 1] - floor_height
        skeleton_joints = np.concatenate([skeleton_joints, np.expand_dims(height, -1)], -1)

    mask = np.zeros((max_num_obj))
    semcls = np.zeros((max_num_obj))
    centers = np.zeros((max_num_obj, 3))
    sizes = np.zeros((max_num_obj, 3))
    headings = np.zeros((max_num_obj, 2))
    n = boxes3D.shape[0]
    if n:
        mask[0:n] = 1
        semcls[0:n] = classes
        centers[0:n, :] = boxes3D[:, 0:3]
        sizes[0:n, :] = boxes3D[:, 3:6]
        headings[0:n, :] = boxes3D[:, 6:8]

    ids = resample_frames(skeleton_joints.shape[0], num_frames)
    return {'input_joints': skeleton_joints[ids].astype(np.float32),
            'box_label_mask': mask.astype(np.float32),
            'sem_cls_label': semcls.astype(np.int64),
            'center_label': centers.astype(np.float32),
            'size': sizes.astype(np.float32),
            'heading': headings.astype(np.float32),
            'vote_label': skeleton_joint_votes[ids, :, 1:].astype(np.float32),
            'vote_label_mask': skeleton_joint_votes[ids, :, 0].astype(np.int64),
            'sample_idx': sample_idx}


def read_sample_hdf5(path):
    """One sample file -> (skeleton_joints, skeleton_joint_votes, instances) (dataloader.py:85-97)."""
    try:
        import h5py
    except ImportError as e:  # pragma: no cover - h5py is absent from the build image
        raise ImportError("reading VirtualHome samples needs h5py (not installed here); the numeric part of the "
                          "loader (sample_to_tensors / augment_sample) does not") from e
    with h5py.File(path, "r") as f:
        joints = f['skeleton_joints'][:]
        votes = f['skeleton_joint_votes'][:]
        instances = [{'class_id': node['class_id'][0], 'centroid': node['centroid'][:], 'R_mat': node['R_mat'][:],
                      'size': node['size'][:]} for node in f['object_nodes'].values()]
    return joints, votes, instances


def load_demo_sample(path, num_frames, use_height=False):
    """demo.py:32-51: a bare (T0,53,3) pose sequence -> {'input_joints' (num_frames,53,3|4) f32, 'sample_idx'}."""
    skeleton_joints = np.load(str(path))
    if use_height:
        floor_height = np.percentile(skeleton_joints[..., 1], 0.99)
        height = skeleton_joints[..., 1] - floor_height
        skeleton_joints = np.concatenate([skeleton_joints, np.expand_dims(height, -1)], -1)
    ids = resample_frames(skeleton_joints.shape[0], num_frames)
    name = os.path.basename(str(path))
    return {'input_joints': skeleton_joints[ids].astype(np.float32), 'sample_idx': '.'.join(name.split('.')[:-1])}


def collate_fn(batch):
    """dataloader.py:149-161: default collation except `sample_idx`, which stays a list of names."""
    out = {}
    for key in batch[0]:
        if key == 'sample_idx':
            out[key] = [elem[key] for elem in batch]
        else:
            out[key] = torch.utils.data.dataloader.default_collate([elem[key] for elem in batch])
    return out


class P2RNet_VirtualHome(Dataset):
    """dataloader.py:17-147 + models/datasets.py:9-24: split json of sample paths, augmentation in train mode."""

    def __init__(self, cfg, mode):
        self.config = cfg.config
        self.dataset_config = cfg.dataset_config
        self.mode = mode
        with open(os.path.join(cfg.config['data']['split'], mode + '.json')) as f:
            self.split = json.load(f)
        self.aug = mode == 'train'
        self.num_frames = cfg.config['data']['num_frames']
        self.use_height = not cfg.config['data']['no_height']
        self.max_num_obj = cfg.config['data']['max_gt_boxes']

    def __len__(self):
        return len(self.split)

    def __getitem__(self, idx):
        path = self.split[idx]
        joints, votes, instances = read_sample_hdf5(path)
        if self.aug:
            joints, instances, votes = augment_sample(joints, instances, votes, *draw_augmentation())
        name = '.'.join(os.path.basename(path).split('.')[:-1])
        return sample_to_tensors(joints, votes, instances, self.num_frames, self.max_num_obj, self.use_height, name)


class Custom_Dataloader(object):
    def __init__(self, dataloader, sampler):
        self.dataloader = dataloader
        self.sampler = sampler


def my_worker_init_fn(worker_id):
    np.random.seed(np.random.get_state()[1][0] + worker_id)


class SyntheticPoseDataset(Dataset):
    """Seeded synthetic samples with the reference loader's per-sample contract (dataloader.py:138-146), held in host
    memory: what the benchmarks and the multi-rank tests feed through `P2RNet_dataloader` when no VirtualHome split is
    mounted.  Sample i is `synthetic.make_batch(1, T, seed + i)` without its batch axis."""

    def __init__(self, num_samples, num_frames, seed=1234):
        from .synthetic import make_batch
        self.samples = []
        for i in range(num_samples):
            b = make_batch(1, num_frames, seed=seed + i)
            self.samples.append({k: (v[0] if torch.is_tensor(v) else v[0]) for k, v in b.items()})

    def __len__(self):
        return len(self.samples)

    def __getitem__(self, idx):
        return self.samples[idx]


def P2RNet_dataloader(cfg, mode='train', dataset=None):
    """dataloader.py:172-199: DistributedSampler under DDP (one shard of the sample list per rank),
    random / sequential sampling otherwise; returns the (dataloader, sampler) pair the epoch loops expect.
    `dataset` (not in the reference): use this map-style dataset instead of the VirtualHome split on disk."""
    if cfg.config['data']['dataset'] != 'virtualhome':
        raise NotImplementedError
    if dataset is None:
        dataset = P2RNet_VirtualHome(cfg, mode)
    if cfg.config['device']['distributed']:
        sampler = DistributedSampler(dataset, shuffle=(mode == 'train'))
    elif mode == 'train':
        sampler = torch.utils.data.RandomSampler(dataset)
    else:
        sampler = torch.utils.data.SequentialSampler(dataset)
    batch_sampler = torch.utils.data.BatchSampler(sampler, batch_size=cfg.config[mode]['batch_size'], drop_last=False)
    loader = DataLoader(dataset=dataset, batch_sampler=batch_sampler, num_workers=cfg.config['device']['num_workers'],
                        collate_fn=collate_fn, worker_init_fn=my_worker_init_fn)
    return Custom_Dataloader(loader, sampler)
