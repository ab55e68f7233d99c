"""One tiny P2RNet train step + one eval forward on a GPU (used by
__graft_entry__.smoke()).  Product path only: HIP ops through libp2r_hip.so."""
import torch


def run(device, batch=2, frames=128):
    from . import P2RConfig, default_config, METHODS
    from .synthetic import make_batch
    from .training import Trainer, ModuleWrapper, load_optimizer
    cfg = P2RConfig(default_config('train', data={'num_frames': frames}), device=device)
    torch.manual_seed(42)
    net = ModuleWrapper(METHODS.get('P2RNet')(cfg)).to(device)
    trainer = Trainer(cfg, net, load_optimizer(cfg.config, net), device)
    out = trainer.train_step(make_batch(batch, frames, seed=1))
    assert all(v == v for v in out.values()), f"NaN in loss dict: {out}"
    assert set(out) == {'total', 'vote_loss', 'objectness_loss', 'center_loss', 'size_loss', 'heading_loss',
                        'sem_cls_loss', 'pos_ratio', 'neg_ratio', 'obj_acc'}
    return out
