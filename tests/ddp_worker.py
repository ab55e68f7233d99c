"""Worker of the multi-rank GPU tests (launched by torch.distributed.run, one process per rank).

Builds the data-parallel trainer exactly as bench.py does (DistributedDataParallel over backend 'nccl' = RCCL, or
over gloo with every rank on cuda:0 when P2R_BENCH_SHARE_GPU=1), feeds each rank its DistributedSampler shard of a
seeded synthetic dataset through P2RNet_dataloader, runs two train steps and checks:
  * every parameter is bit-identical on all ranks afterwards (the gradients were all-reduced),
  * the ranks saw different samples,
  * DDP reduced the gradients in the bucket layout DESIGN.md section 7 states (f32 payload in one bucket),
  * the all-reduced gradients are bit-equal with the BatchNorm-backward passes on the side stream and inline, in the
    exact and in the split16 arithmetic mode.
Rank 0 prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    world = int(os.environ['WORLD_SIZE'])
    rank = int(os.environ['RANK'])
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    share = os.environ.get('P2R_BENCH_SHARE_GPU') == '1'
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if share:
        dist.init_process_group(backend='gloo', init_method='env://')
    else:
        dist.init_process_group(backend='nccl', init_method='env://', device_id=device)
    import bench
    from pose2room_amd.p2rnet.dataloader import P2RNet_dataloader, SyntheticPoseDataset
    frames, per_rank = 512, 2          # >= 512 frames: the seed frames are distinct (the real configuration; below that
    #                                    ATen's index backward adds duplicates with atomics and the step is not bit-reproducible)
    trainer, cfg = bench.build_trainer(device, frames, world)
    cfg.config['device']['distributed'] = True
    cfg.config['train']['batch_size'] = per_rank
    loader = P2RNet_dataloader(cfg, 'train', dataset=SyntheticPoseDataset(per_rank * world * 2, frames, seed=99))
    loader.sampler.set_epoch(0)
    names = []
    losses = None
    for i, batch in enumerate(loader.dataloader):
        names += batch['sample_idx']
        losses = trainer.train_step(batch)
    torch.cuda.synchronize()
    # shards are disjoint
    gathered = [None] * world
    dist.all_gather_object(gathered, names)
    flat = [n for g in gathered for n in g]
    assert len(set(flat)) == len(flat) == per_rank * world * 2, flat
    # parameters in lock-step: compare every tensor with rank 0's copy, bit for bit
    worst = 0
    for name, p in trainer.net.module.named_parameters():
        ref = p.detach().clone()
        dist.broadcast(ref, src=0)
        if not torch.equal(ref, p.detach()):
            worst += 1
    bad = torch.tensor([worst], device=device)
    dist.all_reduce(bad)
    assert int(bad.item()) == 0, f'{int(bad.item())} parameter tensors differ across ranks'
    log = trainer.net._get_ddp_logging_data()
    # the per-step exchange is what DESIGN.md section 7 states: 8,174,532 + 1,600 gradient bytes, 147,380 buffer bytes
    assert log['total_parameter_size_bytes'] == 8176132 and log['broadcast_buffers'] == 1, log
    assert sum(b.numel() * b.element_size() for b in trainer.net.module.buffers()) == 147380
    # one gradient exchange per step: every parameter took part (nothing unused, no second pass), the f32 payload left
    # in one bucket (+ the 1.6 KB f64 bucket), and DDP counted exactly the steps that ran
    assert int(log['iteration']) == i, log['iteration']          # DDP's counter is zero-based: i + 1 steps ran
    assert int(log['num_buckets_reduced']) <= 2 and int(log['unused_parameter_size']) == 0, log
    assert int(log['find_unused_parameters']) == 0 and int(log['has_sync_bn']) == 0
    # LOCAL_RANK -> device: module, DDP's device_ids / output_device and the current device all name this rank's GPU
    assert torch.cuda.current_device() == local_rank
    assert all(p.device.index == local_rank for p in trainer.net.module.parameters())
    assert str(log['device_ids']).strip() == str(local_rank) and int(log['output_device']) == local_rank, log
    if not share:
        # one rank per GPU: pinned staging memory (bench.py's H2D leg, the logging fetch of train_step) is allocated by
        # this process against ITS device, and the copy engine it feeds is this rank's
        probe = torch.empty(1 << 20, dtype=torch.uint8).pin_memory()
        assert probe.is_pinned()
        dst = probe.to(device, non_blocking=True)
        torch.cuda.current_stream(device).synchronize()
        assert dst.device.index == local_rank
    # The BatchNorm-backward passes that run on a side stream under gcn3_dw / gcn3_dcoef (bn_op.OVERLAP_APPLY +
    # OVERLAP_REDUCE) must have joined the main stream before DDP's hooks read the gradients they feed: the all-reduced
    # gradients are bit-equal with the passes on the side stream and with the very same launches issued on the main
    # stream (bn_op.SIDE_INLINE) -- same batch, same mixture noise, same weights, three rounds each.
    from pose2room_amd.p2rnet import bn_op
    assert bn_op.OVERLAP_APPLY and bn_op.OVERLAP_REDUCE
    fixed = trainer.to_device(dict(batch))

    def reduced_grads(inline):
        bn_op.SIDE_INLINE = inline
        try:
            trainer.net.zero_grad()
            torch.manual_seed(1000 + rank)
            est = trainer.net(dict(fixed))
            trainer.net.module.loss(est, fixed)['total'].backward()
            torch.cuda.synchronize()
            return {n: p.grad.detach().clone() for n, p in trainer.net.module.named_parameters()}
        finally:
            bn_op.SIDE_INLINE = False
    # ... in both arithmetic modes: in split16 mode the side-stream passes also EMIT the range words (max |dx| by atomicMax)
    # that scale the operands of the split kernels on the main stream -- a word read before its producer had joined would
    # change the scale, and with it the bits of every gradient upstream
    from pose2room_amd.p2rnet import math_mode
    report = []
    for mode in ('exact', 'split16'):
        with math_mode.use(mode):
            base = reduced_grads(True)
            for rnd in range(3):
                for inline in (True, False):
                    got = reduced_grads(inline)
                    differ = [n for n in base if not torch.equal(base[n], got[n])]
                    worst = max([((base[n] - got[n]).abs().max() / (base[n].abs().max() + 1e-30)).item() for n in differ] or [0.0])
                    report.append((mode, rnd, 'main' if inline else 'side', len(differ), worst, differ[:3]))
        math_mode.reset()
    assert all(r[3] == 0 for r in report), f'reduced gradients differ from the one-stream order: {report}'
    if rank == 0:
        print(json.dumps({'world': world, 'side_stream_equals_one_stream': True, 'modes': ['exact', 'split16'],
                          'backend': dist.get_backend(), 'steps': i + 1,
                          'loss_total': losses['total'],
                          'bucket_sizes': str(log.get('bucket_sizes', '')),
                          'num_buckets': int(log.get('num_buckets_reduced', -1)) if 'num_buckets_reduced' in log else None,
                          'comm_hook': log.get('comm_hook', ''), 'gradient_as_bucket_view': log.get('gradient_as_bucket_view')}),
              flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
