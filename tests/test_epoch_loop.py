"""CPU: the epoch loop (p2rnet/training.py: train_epoch / train), mirror of the reference's train_epoch.py:8-105 -- the
caller of the hot path: phases, network modes, sampler epochs, meters, schedulers, checkpoint hand-off.  Tiny shapes,
the CPU oracle behind the ops."""
import torch


class _Checkpoint(dict):
    """dict-backed stand-in for the reference's CheckpointIO (file IO is out of scope)"""

    def __init__(self):
        super().__init__()
        self.saved = []

    def register_modules(self, **kw):
        self.update(kw)

    def save(self, name):
        self.saved.append(name)


class _Board(object):
    def __init__(self):
        self.calls = []

    def update(self, loss, step_len, phase):
        self.calls.append((phase, step_len, sorted(loss)))


def test_train_two_epochs_cpu():
    from oracle.cpu_backend import cpu_ops
    from pose2room_amd.p2rnet import P2RConfig, default_config, METHODS
    from pose2room_amd.p2rnet.dataloader import P2RNet_dataloader, SyntheticPoseDataset
    from pose2room_amd.p2rnet.training import Trainer, ModuleWrapper, load_optimizer, load_scheduler, train, train_epoch
    T = 64
    cfg = P2RConfig(default_config('train', data={'num_frames': T}, train={'epochs': 2, 'batch_size': 2},
                                   val={'batch_size': 1}, log={'print_step': 1, 'vis_step': 1, 'save_weight_step': 1}),
                    device='cpu')
    lines = []
    cfg.log_string = lines.append
    torch.manual_seed(0)
    net = ModuleWrapper(METHODS.get('P2RNet')(cfg))
    opt = load_optimizer(cfg.config, net)
    trainer = Trainer(cfg, net, opt, torch.device('cpu'))
    sched = load_scheduler(cfg.config, opt)
    tr = P2RNet_dataloader(cfg, 'train', dataset=SyntheticPoseDataset(4, T, seed=1))
    va = P2RNet_dataloader(cfg, 'val', dataset=SyntheticPoseDataset(2, T, seed=2))
    modes = []
    orig_train, orig_eval = trainer.train_step, trainer.eval_step
    trainer.train_step = lambda d: (modes.append(('train', net.training, torch.is_grad_enabled())), orig_train(d))[1]
    trainer.eval_step = lambda d: (modes.append(('val', net.training, torch.is_grad_enabled())), orig_eval(d))[1]
    ckpt, board = _Checkpoint(), _Board()
    w0 = next(net.parameters()).detach().clone()
    with cpu_ops():
        history = train(cfg, trainer, sched, ckpt, tr, va, log_board=board)
    assert len(history) == 2 and all(h == h and h > 0 for h in history)            # finite validation losses
    # per epoch: 2 train batches in train mode with grad, 2 val batches in eval mode without
    assert modes == ([('train', True, True)] * 2 + [('val', False, False)] * 2) * 2
    assert not torch.equal(w0, next(net.parameters()).detach())                       # the optimiser stepped
    assert sched.last_epoch == 2 and ckpt['epoch'] == 1 and ckpt['min_loss'] == history[-1]
    assert ckpt.saved[0:2] == ['last_0', 'best'] and 'last_1' in ckpt.saved
    assert [c[0] for c in board.calls] == (['train'] * 2 + ['val'] * 2) * 2 and board.calls[0][1] == 2 and board.calls[2][1] == 1
    assert any('Switch Phase to val.' in l for l in lines) and any('Current learning rates are' in l for l in lines)
    # train_epoch alone returns the validation meters, whose averages are the means of the per-batch losses
    seen = []
    trainer.eval_step = lambda d: (seen.append(orig_eval(d)), seen[-1])[1]
    with cpu_ops():
        rec = train_epoch(cfg, 3, trainer, {'train': tr, 'val': va})
    assert set(rec) == set(seen[0]) and 'total' in rec
    for k in rec:
        assert abs(rec[k].avg - sum(s[k] for s in seen) / len(seen)) <= 1e-6 * max(1.0, abs(rec[k].avg))
    assert trainer.eval_loss_parser(rec) == rec['total'].avg
