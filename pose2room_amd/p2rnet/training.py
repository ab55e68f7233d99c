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
    return _ScalarFetch(loss_dict).result()


class _ScalarFetch(object):
    """The loss dict on its way to the host.  On a GPU the stacked float64 values are copied into pinned memory on the
    current stream as soon as they exist and an event is recorded behind the copy; `result()` waits for THAT event
    only.  `train_step` starts the fetch right after the loss (before backward and the optimiser are enqueued) and
    reads it at the end: the numbers are this step's, exactly what `{k: v.item()}` returns, but the host never waits
    for the backward pass, so the next step's launches are already queued when the device finishes this one.  (With
    the synchronisation at the very end of the step the device idled ~1.2 ms per 47 ms step at bs=32, T=1024 while the
    host woke up and re-filled the queue: rocprofv3 timeline, profiles/r4_*.)"""
    _pinned = {}

    def __init__(self, loss_dict):
        names = self._keys = list(loss_dict.keys())
        with torch.no_grad():     # float32 -> float64 is exact, so the numbers equal `.item()` of each entry
            # one stack per dtype and one conversion per group instead of one conversion launch per entry
            groups = {}
            for k in names:
                groups.setdefault(loss_dict[k].dtype, []).append(k)
            parts, order = [], []
            for dt, ks in groups.items():
                parts.append(torch.stack([loss_dict[k].detach().reshape(()) for k in ks]).double())
                order += ks
            vals = parts[0] if len(parts) == 1 else torch.cat(parts)
        self.names = order
        self.event = None
        if vals.is_cuda:
            key = (vals.device, len(self.names))
            host = self._pinned.get(key)
            if host is None:
                host = self._pinned[key] = torch.empty(len(self.names), dtype=torch.float64, pin_memory=True)
            host.copy_(vals, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record(torch.cuda.current_stream(vals.device))
            self.vals = host
        else:
            self.vals = vals

    def result(self):
        if self.event is not None:
            self.event.synchronize()
        got = dict(zip(self.names, self.vals.tolist()))
        return {k: got[k] for k in self._keys}


def _split_by_optim_spec(net):
    """Walk the module tree top-down: a module that carries a non-empty `optim_spec` (set from its
    `model.<phase>.optimizer` override, network.py load_optim_spec) becomes one parameter group with that spec;
    branches without any such module fall into the default group (models/optimizers.py:22-38)."""
    groups, default = [], []

    def owns_spec(m):
        return any(getattr(c, 'optim_spec', None) or owns_spec(c) for c in m.children())

    def visit(m):
        for child in m.children():
            if hasattr(child, 'optim_spec'):
                groups.append((child, child.optim_spec))
            elif owns_spec(child):
                visit(child)
            else:
                default.append(child)
    visit(net)
    return groups, default


def load_optimizer(config, net):
    """AdamW with the yaml's Adam hyper-parameters (models/optimizers.py:60-94: the reference builds AdamW for
    `method: Adam`, SGD with momentum 0.9 otherwise), one parameter group per sub-module that brings its own
    `optim_spec`, a default group for the rest."""
    spec = config['optimizer']
    adam = spec.get('method', 'Adam') == 'Adam'
    with_spec, default_modules = _split_by_optim_spec(net)
    groups = []
    for module, ms in with_spec:
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            continue
        ms = ms or spec
        g = {'params': params, 'lr': float(ms['lr'])}
        if adam:
            g.update(betas=tuple(ms['betas']), eps=float(ms['eps']), weight_decay=float(ms['weight_decay']))
        groups.append(g)
    rest = [p for m in default_modules for p in m.parameters() if p.requires_grad]
    if rest:
        groups.append({'params': rest})
    if not groups:       # a bare module without children / specs
        groups = [{'params': [p for p in net.parameters() if p.requires_grad]}]
    if adam:
        # parameters on the GPU: the fused implementation (one multi-tensor kernel per chunk of parameters instead of
        # ~10 elementwise passes over the parameter list; same update rule)
        on_gpu = all(p.is_cuda for g in groups for p in g['params'])
        return torch.optim.AdamW(groups, lr=float(spec['lr']), betas=tuple(spec['betas']), eps=float(spec['eps']),
                                 weight_decay=float(spec['weight_decay']), fused=True if on_gpu else None)
    return torch.optim.SGD(groups, lr=float(spec['lr']), momentum=0.9)


class BNMomentumScheduler(object):
    """Epoch-indexed BatchNorm momentum (models/optimizers.py:53-58,121-148):
    momentum(e) = max(init * decay_rate ** (e // decay_step), floor), written into every BatchNorm1d/2d/3d of `model`.
    The fused BatchNorm ops read `bn.momentum` at call time, so they follow the schedule like nn.BatchNorm does."""

    def __init__(self, cfg, model, bn_lambda=None, last_epoch=-1):
        if not isinstance(model, nn.Module):
            raise RuntimeError("Class '{}' is not a PyTorch nn Module".format(type(model).__name__))
        self.cfg, self.model = cfg, model
        if bn_lambda is None:
            b = cfg.config['bnscheduler']
            bn_lambda = lambda it: max(b['bn_momentum_init'] * b['bn_decay_rate'] ** int(it / b['bn_decay_step']),
                                       b['bn_momentum_max'])
        self.lmbd = bn_lambda
        self.step(last_epoch + 1)
        self.last_epoch = last_epoch

    def step(self, epoch=None):
        epoch = self.last_epoch + 1 if epoch is None else epoch
        self.last_epoch = epoch
        momentum = self.lmbd(epoch)
        for m in self.model.modules():
            if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)):
                m.momentum = momentum

    def show_momentum(self):
        self.cfg.log_string('Current BN decay momentum :%f.' % (self.lmbd(self.last_epoch)))


def load_bnm_scheduler(cfg, net, start_epoch):
    return BNMomentumScheduler(cfg, net, last_epoch=start_epoch - 1)


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
        # the logging scalars leave for the host now (same values as after the step: `loss` is not touched by the
        # backward pass); the wait for them comes after backward + optimiser have been enqueued
        fetch = _ScalarFetch(reduce_dict(loss))
        if loss['total'].requires_grad:
            loss['total'].backward()
            max_norm = self.cfg.config['optimizer']['clip_norm']
            if max_norm > 0:
                torch.nn.utils.clip_grad_norm_(self.net.parameters(), max_norm)
            self.optimizer.step()
        return fetch.result()

    def eval_step(self, data):
        data = self.to_device(data)
        est_data = self.net(data)
        loss = self.net.module.loss(est_data, data)
        return _scalars(reduce_dict(loss))

    def show_lr(self):
        """models/training.py:17-23"""
        lrs = [g['lr'] for g in self.optimizer.param_groups]
        self.cfg.log_string('Current learning rates are: ' + str(lrs) + '.')

    def eval_loss_parser(self, loss_recorder):
        """models/training.py:45-51: the scalar the best checkpoint is chosen by"""
        return loss_recorder['total'].avg

    def visualize_step(self, *args, **kwargs):
        """models/p2rnet/training.py:28-30: a no-op in the reference as well"""


def train_epoch(cfg, epoch, trainer, dataloaders, log_board=None):
    """One epoch over the 'train' and the 'val' loader -- mirror of the reference's train_epoch.py:8-58 (the caller of
    the hot path): network mode per phase (`net.train(phase == 'train')` + `module.set_mode()`), sampler epoch under
    DDP, `train_step` / `eval_step` per batch, loss meters synchronised over the ranks.  Returns the 'val' phase's
    recorder dict {name: AverageMeter} like the reference.  `log_board` (TensorBoard in the reference: out of scope) is
    any object with `update(loss, step_len, phase)`, or None."""
    from .testing import LossRecorder
    loss_recorder = None
    for phase in ['train', 'val']:
        dataloader = dataloaders[phase].dataloader
        sampler = dataloaders[phase].sampler
        batch_size = cfg.config[phase]['batch_size']
        loss_recorder = LossRecorder(batch_size)
        trainer.net.train(phase == 'train')
        trainer.net.module.set_mode()
        if cfg.config['device']['distributed']:
            sampler.set_epoch(epoch)
        cfg.log_string('-' * 100)
        cfg.log_string('Switch Phase to %s.' % (phase))
        cfg.log_string('-' * 100)
        for it, data in enumerate(dataloader):
            if phase == 'train':
                loss = trainer.train_step(data)
            else:
                with torch.no_grad():       # (the reference builds the graph here too and drops it; same values)
                    loss = trainer.eval_step(data)
            if (it % cfg.config['log']['vis_step']) == 0:
                trainer.visualize_step(epoch, phase, it, data)
            loss_recorder.update_loss(loss)
            if (it % cfg.config['log']['print_step']) == 0:
                cfg.log_string('Process: Phase: %s. Epoch %d: %d/%d. Current loss: %s.'
                               % (phase, epoch, it + 1, len(dataloader), str(loss)))
                if log_board is not None:
                    log_board.update(loss, cfg.config['log']['print_step'] * batch_size, phase)
        loss_recorder.synchronize_between_processes(trainer.device)
        cfg.log_string('=' * 100)
        for loss_name, loss_value in loss_recorder.loss_recorder.items():
            cfg.log_string('Currently the last %s loss (%s) is: %f' % (phase, loss_name, loss_value.avg))
        cfg.log_string('=' * 100)
    return loss_recorder.loss_recorder


def train(cfg, trainer, scheduler, checkpoint, train_loader, val_loader, log_board=None, bnm_scheduler=None):
    """Epoch loop -- mirror of train_epoch.py:60-105 (`bnm_scheduler`: the reference defines a BatchNorm-momentum
    scheduler, models/optimizers.py:54-58, but its train.py never builds one; stepped here only when the caller passes it).  `checkpoint` is any object with `get('min_loss')`, `register_modules(**kw)` and
    `save(name)` (the reference's CheckpointIO: file IO is out of scope, a dict-backed stand-in serves), or None."""
    import time
    start_epoch = scheduler.last_epoch
    total_epochs = cfg.config['train']['epochs']
    min_eval_loss = checkpoint.get('min_loss') if checkpoint is not None else None
    dataloaders = {'train': train_loader, 'val': val_loader}
    history = []
    for epoch in range(start_epoch, total_epochs):
        cfg.log_string('-' * 100)
        cfg.log_string('Epoch (%d/%s):' % (epoch + 1, total_epochs))
        trainer.show_lr()
        if bnm_scheduler is not None:
            bnm_scheduler.show_momentum()
        start = time.time()
        eval_loss_recorder = train_epoch(cfg, epoch + 1, trainer, dataloaders, log_board)
        eval_loss = trainer.eval_loss_parser(eval_loss_recorder)
        scheduler.step()
        if bnm_scheduler is not None:
            bnm_scheduler.step()
        cfg.log_string('Epoch (%d/%s) Time elapsed: (%f).' % (epoch + 1, total_epochs, time.time() - start))
        history.append(eval_loss)
        if checkpoint is not None:
            checkpoint.register_modules(epoch=epoch, min_loss=eval_loss)
            if ((epoch % cfg.config['log']['save_weight_step']) == 0) or (epoch == total_epochs - 1):
                checkpoint.save('last_%d' % (epoch))
                cfg.log_string('Saved the latest checkpoint.')
        if epoch == 0 or min_eval_loss is None or eval_loss < min_eval_loss:
            if checkpoint is not None:
                checkpoint.save('best')
            min_eval_loss = eval_loss
            cfg.log_string('Saved the best checkpoint.')
    return history
