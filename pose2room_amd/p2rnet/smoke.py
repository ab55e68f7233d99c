"""One tiny P2RNet train step + one eval forward on a GPU (used by
__graft_entry__.smoke()).  Product path only: HIP ops through libp2r_hip.so."""
import torch


def run(device, batch=2, frames=128):
    from . import P2RConfig, default_config, METHODS
    from .synthetic import make_batch
    from .training import Trainer, ModuleWrapper, load_optimizer
    cfg = P2RConfig(default_config('train', data={'num_frames': frames}), device=device)
    def one_step():
        torch.manual_seed(42)
        net = ModuleWrapper(METHODS.get('P2RNet')(cfg)).to(device)
        trainer = Trainer(cfg, net, load_optimizer(cfg.config, net), device)
        torch.manual_seed(7)              # the mixture heads' noise
        return trainer.train_step(make_batch(batch, frames, seed=1))

    out = one_step()
    assert all(v == v for v in out.values()), f"NaN in loss dict: {out}"
    assert set(out) == {'total', 'vote_loss', 'objectness_loss', 'center_loss', 'size_loss', 'heading_loss',
                        'sem_cls_loss', 'pos_ratio', 'neg_ratio', 'obj_acc'}
    # the same step in the opt-in split16 arithmetic of the ST-GCN kernels (math_mode): same losses to fp32 accuracy
    from . import math_mode
    math_mode.set_mode('split16')
    try:
        out16 = one_step()
    finally:
        math_mode.set_mode('exact')
        math_mode.reset()
    for k in ('total', 'vote_loss', 'center_loss', 'size_loss', 'heading_loss', 'sem_cls_loss', 'objectness_loss'):
        assert abs(out16[k] - out[k]) <= 1e-3 * max(1.0, abs(out[k])), (k, out[k], out16[k])
    return out
