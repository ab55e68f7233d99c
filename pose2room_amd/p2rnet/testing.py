"""Test / evaluation loop: losses + detection AP over a stream of batches.

Mirror of the reference's models/p2rnet/testing.py:16-50 (`Tester.test_step`: generate -> loss ->
rank-averaged scalar dict, returns `(loss_dict, est_data)`) and test_epoch.py:10-76 (`test_func`,
`test`: one `APCalculator` per IoU threshold of `cfg.config[mode]['ap_iou_thresholds']`, fed with
`eval_dict['batch_pred_map_cls' / 'batch_gt_map_cls']` of every batch, loss meters synchronised over
ranks at the end).  Visualisation / result dumping (testing.py:52-130) is file export and not mirrored.
"""
from time import time

import torch
import torch.distributed as dist

from ..net_utils.ap_helper import APCalculator
from .training import Trainer, reduce_dict


class AverageMeter(object):
    """Running average of a scalar (net_utils/utils.py:295-327)."""

    def __init__(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        if isinstance(val, list):
            self.sum += sum(val)
            self.count += len(val)
        else:
            self.sum += val * n
            self.count += n
        self.avg = self.sum / self.count

    def synchronize_between_processes(self, device=None):
        if not (dist.is_available() and dist.is_initialized()):
            return
        t = torch.tensor([self.count, self.sum], dtype=torch.float64,
                         device=device if device is not None else ('cuda' if torch.cuda.is_available() else 'cpu'))
        dist.barrier()
        dist.all_reduce(t)
        self.count, self.sum = int(t[0].item()), t[1].item()
        self.avg = self.sum / max(self.count, 1)


class LossRecorder(object):
    """Per-key meters (net_utils/utils.py:329-355)."""

    def __init__(self, batch_size=1):
        self._batch_size = batch_size
        self._loss_recorder = {}

    @property
    def batch_size(self):
        return self._batch_size

    @property
    def loss_recorder(self):
        return self._loss_recorder

    def update_loss(self, loss_dict):
        for key, item in loss_dict.items():
            self._loss_recorder.setdefault(key, AverageMeter()).update(item, self._batch_size)

    def synchronize_between_processes(self, device=None):
        for meter in self._loss_recorder.values():
            meter.synchronize_between_processes(device)


class Tester(Trainer):
    def __init__(self, cfg, net, device=None):
        super().__init__(cfg, net, None, device)

    def test_step(self, data):
        data = self.to_device(data)
        est_data = self.net.module.generate(data)
        loss = self.net.module.loss(est_data, data)
        loss_reduced = reduce_dict(loss)
        return {k: v.item() for k, v in loss_reduced.items()}, est_data


def test_func(cfg, tester, batches, ap_device='cpu'):
    """batches: any iterable of data dicts (the reference passes `test_loader.dataloader`)."""
    mode = cfg.config['mode']
    recorder = LossRecorder(cfg.config[mode]['batch_size'])
    calculators = [APCalculator(thr, getattr(cfg.dataset_config, 'class2type', None), False, device=ap_device)
                   for thr in cfg.config[mode]['ap_iou_thresholds']]
    for data in batches:
        loss, est_data = tester.test_step(data)
        eval_dict = est_data[1]
        for calc in calculators:
            calc.step(eval_dict['batch_pred_map_cls'], eval_dict['batch_gt_map_cls'])
        recorder.update_loss(loss)
    recorder.synchronize_between_processes(tester.device)
    return recorder.loss_recorder, calculators


def test(cfg, tester, batches, ap_device='cpu'):
    """test_epoch.py:52-76 -> {'loss': {key: avg}, 'metrics': [{...} per IoU threshold]}; also logged."""
    mode = cfg.config['mode']
    tester.net.train(mode == 'train')
    start = time()
    with torch.no_grad():
        meters, calculators = test_func(cfg, tester, batches, ap_device)
    cfg.log_string('Test time elapsed: (%f).' % (time() - start))
    out = {'loss': {k: m.avg for k, m in meters.items()}, 'metrics': []}
    for key, avg in out['loss'].items():
        cfg.log_string('Test loss (%s): %f' % (key, avg))
    for thr, calc in zip(cfg.config[mode]['ap_iou_thresholds'], calculators):
        cfg.log_string(('-' * 10 + 'iou_thresh: %f' + '-' * 10) % thr)
        metrics = calc.compute_metrics()
        for key in metrics:
            cfg.log_string('eval %s: %f' % (key, metrics[key]))
        out['metrics'].append(metrics)
    return out
