"""Train / eval step (mirror of the reference's models/training.py:25-43 and
models/p2rnet/training.py:18-36,100-120): zero_grad -> forward -> loss ->
backward iff total.requires_grad -> optional clip -> optimizer step -> averaged
scalar dict.  `net` must expose `.module` (DDP or `ModuleWrapper`), as in the
reference where the loss is reached through `net.module.loss`."""
import torch
import torch.distributed as dist
from torch import nn


class ModuleWrapper(nn.Module):
    """Single-process stand-in for DDP/DataParallel: forwards to `.module`."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def reduce_dict(input_dict, average=True):
    """All-reduce a dict of scalar tensors (net_utils/utils.py:490-514): keys sorted,
    values stacked into one message, averaged over ranks."""
    world_size = get_world_size()
    if world_size < 2:
        return input_dict
    with torch.no_grad():
        names = sorted(input_dict.keys())
        values = torch.stack([input_dict[k] for k in names], dim=0)
        dist.all_reduce(values)
        if average:
            values /= world_size
        return {k: v for k, v in zip(names, values)}


def _scalars(loss_dict):
    """{name: python float} like the reference's `{k: v.item()}` (models/training.py:41-42), but with one
    device->host transfer for the whole dict instead of one synchronisation per entry."""
    names = list(loss_dict.keys())
    with torch.no_grad():     # float32 -> float64 is exact, so the numbers equal `.item()` of each entry
        vals = torch.stack([loss_dict[k].detach().double() for k in names]).tolist()
    return dict(zip(names, vals))


def load_optimizer(config, net):
    """AdamW with the yaml's Adam hyper-parameters (models/optimizers.py:90-94: the
    reference builds AdamW for `method: Adam`)."""
    spec = config['optimizer']
    params = [p for p in net.parameters() if p.requires_grad]
    return torch.optim.AdamW(params, lr=float(spec['lr']), betas=tuple(spec['betas']),
                             eps=float(spec['eps']), weight_decay=float(spec['weight_decay']))


def load_scheduler(config, optimizer):
    s = config['scheduler']
    return torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=s['milestones'], gamma=s['gamma'])


class Trainer(object):
    def __init__(self, cfg, net, optimizer, device=None):
        self.cfg = cfg
        self.net = net
        self.optimizer = optimizer
        self.device = device

    def to_device(self, data):
        for key in data:
            if key in ['sample_idx']:
                continue
            data[key] = data[key].to(self.device)
        return data

    def compute_loss(self, data):
        data = self.to_device(data)
        est_data = self.net(data)
        return self.net.module.loss(est_data, data)

    def train_step(self, data):
        self.optimizer.zero_grad()
        loss = self.compute_loss(data)
        if loss['total'].requires_grad:
            loss['total'].backward()
            max_norm = self.cfg.config['optimizer']['clip_norm']
            if max_norm > 0:
                torch.nn.utils.clip_grad_norm_(self.net.parameters(), max_norm)
            self.optimizer.step()
        return _scalars(reduce_dict(loss))

    def eval_step(self, data):
        data = self.to_device(data)
        est_data = self.net(data)
        loss = self.net.module.loss(est_data, data)
        return _scalars(reduce_dict(loss))
