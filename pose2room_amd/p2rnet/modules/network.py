"""P2RNet: backbone -> centre voting -> detection (mirror of the reference's
models/p2rnet/modules/network.py:10-106 and models/network.py:8-86)."""
import torch
import torch.nn as nn

from ..registers import METHODS, MODULES, LOSSES


def _multi_getattr(obj, dotted):
    for part in dotted.split("."):
        obj = getattr(obj, part)
    return obj


def _multi_hasattr(obj, dotted):
    for part in dotted.split("."):
        if not hasattr(obj, part):
            return False
        obj = getattr(obj, part)
    return True


class BaseNetwork(nn.Module):
    """Shared plumbing: per-phase optimiser spec, freezing, checkpoint key handling."""

    def freeze_modules(self, cfg):
        if cfg.config['mode'] == 'train':
            for layer in cfg.config['train']['freeze']:
                if not _multi_hasattr(self, layer):
                    continue
                for p in _multi_getattr(self, layer).parameters():
                    p.requires_grad = False
                cfg.log_string('The module: %s is fixed.' % layer)

    def set_mode(self):
        frozen = self.cfg.config['train']['freeze']
        for name, child in self.named_children():
            if name in frozen:
                child.train(False)

    def load_weight(self, pretrained_model):
        """Load a reference checkpoint's `net` dict: keys carry a leading 'module.'
        (DDP / DataParallel wrapper) which is dropped (models/network.py:59-67)."""
        own = self.state_dict()
        matched = {'.'.join(k.split('.')[1:]): v for k, v in pretrained_model.items()
                   if '.'.join(k.split('.')[1:]) in own}
        self.cfg.log_string(str({k.split('.')[0] for k in own if k not in matched}) + ' subnet missed.')
        own.update(matched)
        self.load_state_dict(own)

    def load_optim_spec(self, config, net_spec):
        if config['mode'] != 'train':
            return None
        if 'optimizer' in net_spec:
            spec = config['optimizer'].copy()
            for key in spec:
                spec[key] = net_spec['optimizer'].get(key, spec[key])
            return spec
        return config['optimizer']


@METHODS.register_module
class P2RNet(BaseNetwork):
    def __init__(self, cfg):
        nn.Module.__init__(self)
        self.cfg = cfg
        mode = cfg.config['mode']
        phase_names = ['backbone', 'centervoting', 'detection'] if cfg.config[mode]['phase'] in ['full'] else []
        if (not cfg.config['model']) or (not phase_names):
            cfg.log_string('No submodule found. Please check the phase name and model definition.')
            raise ModuleNotFoundError('No submodule found. Please check the phase name and model definition.')
        for phase_name, net_spec in cfg.config['model'].items():
            if phase_name not in phase_names:
                continue
            optim_spec = self.load_optim_spec(cfg.config, net_spec)
            self.add_module(phase_name, MODULES.get(net_spec['method'])(cfg, optim_spec))
            loss_cls = LOSSES.get(net_spec['loss'], 'Null')
            setattr(self, phase_name + '_loss',
                    loss_cls(net_spec.get('weight', 1), cfg.config['device']['gpu'], cfg))
        self.freeze_modules(cfg)

    def _votes(self, data):
        end_points = self.backbone(data['input_joints'], {})
        from .. import pw_op
        from . import vote_center
        if vote_center.USE_FUSED_HEAD and pw_op.votes_normalized_supported(self.centervoting, end_points['seed_skeleton'],
                                                                            end_points['seed_features']):
            # conv_input on the job-list kernels + offset / residual add / normalisation / re-layout in one launch
            xyz, features = pw_op.votes_normalized(self.centervoting, end_points['seed_skeleton'],
                                                   end_points['seed_features'])
        else:
            xyz, features = self.centervoting(end_points['seed_skeleton'], end_points['seed_features'])
            features = features.div(torch.norm(features, p=2, dim=2).unsqueeze(2))   # no epsilon, as the reference
        end_points['vote_xyz'] = xyz
        end_points['vote_features'] = features
        return xyz, features, end_points

    def forward(self, data, eps=None):
        """data['input_joints'] (B,T,J,3) -> end_points dict (network.py:75-96)."""
        xyz, features, end_points = self._votes(data)
        end_points, _ = self.detection(xyz, features, end_points, False, eps=eps)
        return end_points

    def generate_end_points(self, data):
        """The network part of `generate`: deterministic heads, no parsing / NMS."""
        xyz, features, end_points = self._votes(data)
        end_points, _ = self.detection.generate(xyz, features, end_points, False)
        return end_points

    def generate(self, data, eval=True):
        """Detection path with deterministic mixture means, prediction parsing and
        NMS (network.py:44-73)."""
        from ...net_utils.ap_helper import (parse_predictions, parse_groundtruths,
                                            assembly_pred_map_cls, assembly_gt_map_cls)
        end_points = self.generate_end_points(data)
        eval_dict, parsed_predictions = parse_predictions(end_points, data, self.cfg.eval_config)
        eval_dict = assembly_pred_map_cls(eval_dict, parsed_predictions, self.cfg.eval_config)
        if eval:
            parsed_gts = parse_groundtruths(data, self.cfg.eval_config)
            eval_dict['batch_gt_map_cls'] = assembly_gt_map_cls(parsed_gts)
        return end_points, eval_dict, parsed_predictions

    def loss(self, pred_data, gt_data):
        if isinstance(pred_data, tuple):
            pred_data = pred_data[0]
        return self.detection_loss(pred_data, gt_data, self.cfg.dataset_config)
