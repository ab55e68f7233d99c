"""Configuration objects for the P2RNet hot path.

`P2RConfig` plays the role of the reference's `CONFIG` (configs/config_utils.py:42-138)
as far as the model code is concerned: `.config` (the yaml dict), `.dataset_config`
(the constants of configs/dataset_config.py the model reads), `.eval_config`
(configs/config_utils.py:146-159) and `.log_string`.  It creates no directories
and no log files.  `default_config()` reproduces the keys and values of
configs/config_files/p2rnet_train.yaml / p2rnet_test.yaml.
"""
import copy


class Struct:
    def __init__(self, **kwargs):
        self.update(**kwargs)

    def update(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)


class DatasetConfig:
    """The subset of Dataset_Config('virtualhome') the hot path reads
    (configs/dataset_config.py:11-56)."""
    joint_num = 53
    origin_joint_id = 0
    num_class = 22
    contact_dist_thresh = 1.0
    class_labels = ['bathtub', 'bed', 'bench', 'bookshelf', 'cabinet', 'chair', 'closet', 'desk',
                    'dishwasher', 'faucet', 'fridge', 'garbagecan', 'lamp', 'microwave', 'monitor',
                    'nightstand', 'sofa', 'stove', 'toilet', 'washingmachine', 'window', 'computer']
    type2class = {cls: index for index, cls in enumerate(class_labels)}       # dataset_config.py:82-83
    class2type = {index: cls for index, cls in enumerate(class_labels)}


_BASE = {
    'method': 'P2RNet', 'resume': False, 'finetune': False, 'weight': [], 'seed': 42,
    'device': {'use_gpu': True, 'gpu_ids': '0', 'num_workers': 0, 'world_size': 1,
               'dist_url': 'env://', 'gpu': 0, 'is_main_process': True, 'distributed': False},
    'data': {'dataset': 'virtualhome', 'split': 'datasets/virtualhome_22_classes/splits/script_level',
             'num_frames': 768, 'num_seeds': 512, 'seed_sampling': 'uniform', 'max_gt_boxes': 10,
             'num_target': 128, 'vote_factor': 1, 'cluster_sampling': 'vote_fps', 'no_height': True,
             'num_gaussian': 100},
    'model': {'backbone': {'method': 'STGCN', 'loss': 'Null'},
              'centervoting': {'method': 'CenterVoteModule', 'loss': 'Null'},
              'detection': {'method': 'ProposalNet', 'loss': 'BoxNetDetectionLoss'}},
    'optimizer': {'method': 'Adam', 'lr': 1e-3, 'betas': [0.9, 0.999], 'eps': 1e-08,
                  'weight_decay': 0, 'clip_norm': -1},
    'scheduler': {'milestones': [80, 120, 160], 'gamma': 0.1},
    'train': {'epochs': 180, 'phase': 'full', 'freeze': [], 'batch_size': 8},
    'val': {'phase': 'full', 'batch_size': 8},
    'test': {'phase': 'full', 'batch_size': 1, 'use_cls_nms': False, 'use_3d_nms': True,
             'ap_iou_thresholds': [0.25, 0.5], 'remove_far_box': True, 'nms_iou': 0.10,
             'use_old_type_nms': False, 'per_class_proposal': True, 'conf_thresh': 0.05,
             'multi_mode': False, 'sample_cls': False},
    'demo': {'phase': 'full'},
    'log': {'path': 'out/p2rnet', 'save_weight_step': 50, 'vis_step': 10, 'print_step': 10},     # p2rnet_train.yaml:55-59
}


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


def default_config(mode='train', **overrides):
    """yaml-equivalent dict; `overrides` are nested dicts merged on top, e.g.
    default_config('train', data={'num_frames': 1024})."""
    cfg = copy.deepcopy(_BASE)
    cfg['mode'] = mode
    return _merge(cfg, overrides)


class P2RConfig:
    def __init__(self, config=None, mode=None, device='cpu', logger=None):
        self.config = config if config is not None else default_config(mode or 'train')
        if mode is not None:
            self.config['mode'] = mode
        self.config.setdefault('device', {})['gpu'] = device
        self.dataset_config = DatasetConfig()
        self._logger = logger
        self.eval_config = None
        if self.config['mode'] != 'train':
            t = self.config['test']
            # configs/config_utils.py:146-159
            self.eval_config = {'remove_far_box': t['remove_far_box'], 'use_3d_nms': t['use_3d_nms'],
                                'nms_iou': t['nms_iou'], 'use_old_type_nms': t['use_old_type_nms'],
                                'cls_nms': t['use_cls_nms'], 'per_class_proposal': t['per_class_proposal'],
                                'conf_thresh': t['conf_thresh'], 'dataset_config': self.dataset_config,
                                'multi_mode': t['multi_mode'], 'sample_cls': t['sample_cls']}

    def log_string(self, content):
        if self._logger is not None:
            self._logger(content)
