"""Sample loader and the demo forward (BASELINE configs[0]) against G8: outputs of the reference's
`P2RNet_VirtualHome.__getitem__` and of `P2RNet.generate` on the reference's demo pose sequence
(tests/golden/make_loader_golden.py)."""
import os
import random

import numpy as np
import pytest
import torch

from pose2room_amd.p2rnet import dataloader as dl
from tests.test_model_cpu import build, check_endpoints

G8 = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'g8_loader_demo.npz'))
KEYS = ['input_joints', 'box_label_mask', 'sem_cls_label', 'center_label', 'size', 'heading', 'vote_label',
        'vote_label_mask']


def _sample():
    inst = [{'class_id': int(G8['s_class_id'][i]), 'centroid': G8['s_centroid'][i], 'R_mat': G8['s_R_mat'][i],
             'size': G8['s_size'][i]} for i in range(len(G8['s_class_id']))]
    return G8['s_joints'].copy(), G8['s_votes'].copy(), inst


@pytest.mark.parametrize("mode,seed", [('test', 0)] + [('train', s) for s in (1, 2, 3, 4, 5, 6)])
def test_dataset_item_matches_reference(mode, seed):
    joints, votes, inst = _sample()
    random.seed(seed); np.random.seed(seed)
    if mode == 'train':
        joints, inst, votes = dl.augment_sample(joints, inst, votes, *dl.draw_augmentation())
    item = dl.sample_to_tensors(joints, votes, inst, num_frames=64, max_num_obj=10, use_height=False,
                                sample_idx='3_0_364_Female2_0')
    tag = f'{mode}{seed}'
    for k in KEYS:
        want = G8[f'item_{tag}_{k}']
        assert item[k].dtype == want.dtype and item[k].shape == want.shape, k
        np.testing.assert_allclose(item[k], want, rtol=1e-6, atol=1e-6, err_msg=k)
    assert item['sample_idx'] == str(G8[f'item_{tag}_sample_idx'])


def test_augmentation_keeps_votes_on_targets():
    """Size-independent property: votes point at object centres before and after any augmentation."""
    joints, votes, inst = _sample()
    c = np.array([n['centroid'] for n in inst])
    votes[..., 1:4] = c[0] - joints            # make vote 1 of every joint point at object 0
    for flip in (0, 1):
        for ang in (-np.pi, -0.5 * np.pi, 0, 0.5 * np.pi):
            j2, i2, v2 = dl.augment_sample(joints, inst, votes, flip, ang, 0.37)
            np.testing.assert_allclose(j2 + v2[..., 1:4], np.broadcast_to(i2[0]['centroid'], j2.shape), atol=1e-5)
            R = i2[0]['R_mat']
            np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-6)
            assert np.linalg.det(R) > 0.99                       # still a proper rotation after the flip fix-up


def test_collate_and_resample():
    ids = dl.resample_frames(341, 768)
    assert ids.dtype == np.uint16 and ids[0] == 0 and ids[-1] == 340 and len(ids) == 768 and (np.diff(ids.astype(int)) >= 0).all()
    joints, votes, inst = _sample()
    items = [dl.sample_to_tensors(joints, votes, inst, 16, sample_idx=f's{i}') for i in range(3)]
    batch = dl.collate_fn(items)
    assert batch['input_joints'].shape == (3, 16, 53, 3) and batch['input_joints'].dtype == torch.float32
    assert batch['sem_cls_label'].dtype == torch.int64 and batch['sample_idx'] == ['s0', 's1', 's2']
    with pytest.raises(ImportError):
        dl.read_sample_hdf5('/nonexistent/sample.hdf5')            # h5py is not installed in this image


def _demo_data(tmp_path, device='cpu'):
    path = tmp_path / 'input_joints_1.npy'
    np.save(path, G8['demo_sequence'])
    item = dl.load_demo_sample(path, 768)
    assert item['sample_idx'] == 'input_joints_1' and item['input_joints'].shape == (768, 53, 3)
    return {'input_joints': torch.from_numpy(item['input_joints'])[None].to(device)}


def check_demo(net, data, parse=True):
    if not parse:           # prediction parsing + NMS exist on the GPU only (no CPU fallback in the product)
        check_endpoints(G8, 'demo', net.generate_end_points(data))
        return
    ep, eval_dict, parsed = net.generate(data, eval=False)
    check_endpoints(G8, 'demo', ep)
    assert np.array_equal(eval_dict['pred_mask'], G8['demo_pred_mask'])
    np.testing.assert_allclose(parsed['pred_corners_3d'], G8['demo_pred_corners_3d'], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(parsed['obj_prob'], G8['demo_obj_prob'], rtol=1e-4, atol=1e-5)
    assert np.array_equal(parsed['pred_sem_cls'], G8['demo_pred_sem_cls'])
    assert len(np.unique(G8['demo_seed_inds'])) < 512      # the case it is here for: repeated frames, duplicate seeds


def test_demo_forward_cpu(tmp_path):
    """configs[0]: the reference's demo sequence through the eval path, pointnet2 ops on the CPU oracle."""
    from oracle.cpu_backend import cpu_ops
    net, cfg = build('test', 768, remove_far_box=True)
    net.eval()
    with torch.no_grad(), cpu_ops():
        check_demo(net, _demo_data(tmp_path), parse=False)


@pytest.mark.gpu
def test_demo_forward_gpu(tmp_path, dev):
    net, cfg = build('test', 768, device=dev, remove_far_box=True)
    net = net.to(dev).eval()
    with torch.no_grad():
        check_demo(net, _demo_data(tmp_path, dev))


def test_read_sample_hdf5_through_in_memory_file(monkeypatch):
    """`read_sample_hdf5` cannot meet a real .hdf5 here (no h5py in the image): an in-memory stand-in with the h5py
    surface it touches -- File(path, 'r') as context manager, group['name'], dataset[:] / dataset[0], .values() --
    walks its eight lines over the sample layout of utils/virtualhome/3_generate_samples.py:176-193 and the result
    goes through the numeric loader like a file would (reference dataloader.py:85-97)."""
    import sys
    import types
    joints, votes, inst = _sample()

    class _DS(object):
        def __init__(self, a):
            self.a = np.asarray(a)

        def __getitem__(self, k):
            return self.a[k]

    class _Group(dict):
        pass

    opened = []

    class _File(_Group):
        def __init__(self, path, mode):
            opened.append((path, mode))
            self['skeleton_joints'] = _DS(joints)
            self['skeleton_joint_votes'] = _DS(votes)
            nodes = _Group()
            for i, n in enumerate(inst):
                g = _Group(class_id=_DS([n['class_id']]), centroid=_DS(n['centroid']), R_mat=_DS(n['R_mat']),
                           size=_DS(n['size']))
                nodes[str(i)] = g
            self['object_nodes'] = nodes

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

    fake = types.ModuleType('h5py')
    fake.File = _File
    monkeypatch.setitem(sys.modules, 'h5py', fake)
    j, v, nodes = dl.read_sample_hdf5('/data/samples/7_0_0.hdf5')
    assert opened == [('/data/samples/7_0_0.hdf5', 'r')]
    np.testing.assert_array_equal(j, joints)
    np.testing.assert_array_equal(v, votes)
    assert len(nodes) == len(inst)
    for a, b in zip(nodes, inst):
        assert a['class_id'] == b['class_id']
        for k in ('centroid', 'R_mat', 'size'):
            np.testing.assert_array_equal(a[k], b[k])
    want = dl.sample_to_tensors(joints, votes, inst, 16, sample_idx='s')
    got = dl.sample_to_tensors(j, v, nodes, 16, sample_idx='s')
    for k in want:
        if isinstance(want[k], np.ndarray):
            np.testing.assert_array_equal(got[k], want[k])
