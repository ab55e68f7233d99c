#!/usr/bin/env python
"""P2RNet train-step benchmark on MI355X (contract in the task brief).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one full train step of P2RNet on one batch of synthetic pose sequences
already resident in HBM: zero_grad -> forward (ST-GCN backbone, centre voting, vote
aggregation on the HIP pointnet2 ops, proposal heads) -> loss (HIP nn_distance) ->
backward (+ RCCL gradient all-reduce for N>1) -> AdamW step -> the 10-scalar
reduce_dict / .item() of the reference's train_step (models/training.py:25-43).
Workload = BASELINE.json configs[2]: bs=32 per GPU, T=1024, J=53 (weak scaling).

The per-rank batch comes through the loader the epoch loops use (`P2RNet_dataloader`, with a
`DistributedSampler` shard per rank when N > 1) over a seeded in-memory synthetic dataset.  `value` is
measured with that batch resident in HBM (contract); the same K steps are then repeated with the
reference's `to_device` inside the step -- the batch (6.5 MB per sample) crossing PCIe from pinned
host memory each step, prefetched one step ahead on a side stream -- and reported under `h2d`.

Rank 0 prints ONE JSON line.  Besides the contract keys it carries
  step_ms      -- median / p10 / p90 of the per-step device time (events between steps, no host sync);
  h2d          -- the same run with the batch copied host -> device every step (pinned, prefetched);
  verify       -- what was checked at the bench shape BEFORE timing (finite losses of a train step; sample 0 of the
                  full batch against the same sample run alone with running-statistics BatchNorm);
  roofline     -- the dominant kernel (gcn3_kernel: graph-conv forward + data gradient) against the fp32 MFMA roof: MFMA
                  FLOPs the kernel ISSUES per launch / its average duration INSIDE the step (HIP events around every
                  launch of instrumented steps that follow the timed region);
  mfma_kernels -- the same for the other MFMA kernels of the ST-GCN blocks (in-step durations, issued FLOPs);
  step_roofline-- the whole step: MFMA FLOPs issued per step (stored PMC profile) over this run's step time;
  ddp          -- N > 1 only: per-rank step times, and the step time with the gradient all-reduce switched off
                  (`no_sync`) next to the one with it = the all-reduce time that is NOT hidden under the backward pass;
  split16      -- the SAME K steps in the opt-in split16 arithmetic of the ST-GCN kernels (pose2room_amd.p2rnet.math_mode:
                  two-part fp16 MFMA products, fp32 accumulation): value / ms_per_step / verify, and the split kernels
                  against the 16-bit MFMA roof (2.5 PFLOP/s dense) on the 16-bit MFMA FLOPs they issue.  Labelled,
                  never the headline: `value`, `dtype` and `roofline` above are the exact-fp32 path;
  kernels      -- event-timed durations of the pointnet2 / loss HIP kernels at the P2RNet shapes (`GBps_l2_assisted`:
                  algorithmic bytes / time of a gather whose 17 MB source stays in L2 / MALL -- not an HBM rate);
  cpu_baseline -- the same host model on the host cores with the CPU oracle behind
                  the ops ("port"), on a bounded sample.
"""
import argparse
import gc
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic fwd+bwd GFLOP per sample of the reference step (FlopCounterMode on the
# imported reference); linear in T.
_GFLOP_PER_SAMPLE = {512: 99.44, 768: 147.94, 1024: 196.44, 2048: 390.45}
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md, chip-level parameters
FP16_MFMA_PEAK_TFLOPS = 2500.0  # dense BF16 / FP16 MFMA (same table; the 5 PF headline figure includes 2:1 sparsity)


def gflop_per_sample(T):
    if T in _GFLOP_PER_SAMPLE:
        return _GFLOP_PER_SAMPLE[T]
    return 196.44 * T / 1024.0


def build_trainer(device, frames, world):
    from pose2room_amd.p2rnet import P2RConfig, default_config, METHODS
    from pose2room_amd.p2rnet.training import Trainer, ModuleWrapper, load_optimizer
    cfg = P2RConfig(default_config('train', data={'num_frames': frames}), device=device)
    torch.manual_seed(42)   # p2rnet_train.yaml: seed 42, identical weights on every rank
    net = METHODS.get('P2RNet')(cfg).to(device)
    if world > 1:
        from torch.nn.parallel import DistributedDataParallel as DDP
        # one ~8.2 MB gradient payload: a single bucket, one all-reduce per step over xGMI
        net = DDP(net, device_ids=[device.index], bucket_cap_mb=16, gradient_as_bucket_view=True)
    else:
        net = ModuleWrapper(net)
    return Trainer(cfg, net, load_optimizer(cfg.config, net), device), cfg


def _profile_json(name):
    """A measurement committed under profiles/ (written by the tools/pmc_*.sh scripts), or None."""
    path = os.path.join(ROOT, 'profiles', name)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)


class LaunchTimer:
    """HIP events around every call of the named C-ABI entry points (on the stream they launch on), to get a
    kernel's duration INSIDE the train step -- with the caches, clocks and neighbours it has there -- rather than
    in an isolated loop.  Used on extra steps after the timed region, never inside it."""

    def __init__(self, lib, names):
        self.lib, self.names, self.orig, self.records = lib, names, {}, []

    def __enter__(self):
        for name, tag_of in self.names.items():
            orig = getattr(self.lib, name)
            self.orig[name] = orig

            def wrapper(*args, _orig=orig, _name=name, _tag_of=tag_of):
                tag = _tag_of(args)
                if tag is None:
                    return _orig(*args)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = _orig(*args)
                e1.record()
                self.records.append((tag, e0, e1))
                return rc
            setattr(self.lib, name, wrapper)
        return self

    def __exit__(self, *exc):
        for name, orig in self.orig.items():
            setattr(self.lib, name, orig)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for tag, e0, e1 in self.records:
            out.setdefault(tag, []).append(e0.elapsed_time(e1))
        return {k: (sum(v) / len(v), len(v)) for k, v in out.items()}


def _null(a):
    return a is None or getattr(a, 'value', a) in (None, 0)


# entry point -> tag of the launch (None: not timed).  Argument positions as in include/p2r_hip.h.
_TIMED = {
    'p2r_stgcn_gcn3_forward': lambda a: None if _null(a[11]) else ('gcn_data_gradient' if a[5] == 1 else 'gcn_forward'),
    'p2r_stgcn_gcn3_data_gradient_masked_addend': lambda a: 'gcn_data_gradient',
    'p2r_stgcn_gcn3_coef_grad': lambda a: 'gcn_coef_grad',
    'p2r_stgcn_gcn3_weight_grad': lambda a: 'gcn_weight_grad',
    'p2r_stgcn_gcn2_forward': lambda a: None if _null(a[12]) else ('gcn_data_gradient' if _null(a[10]) else 'gcn_forward'),
    'p2r_stgcn_gcn_weight_grad': lambda a: None if _null(a[10]) else 'gcn_weight_grad',
    'p2r_stgcn_gcn_coef_grad': lambda a: 'gcn_coef_grad',
    'p2r_stgcn_tconv3_forward': lambda a: None if (_null(a[9]) or a[3] != 3) else ('tconv_data_gradient' if _null(a[5]) else 'tconv_forward'),
    'p2r_stgcn_tconv2_forward': lambda a: None if (_null(a[9]) or a[3] != 3) else ('tconv_data_gradient' if _null(a[5]) else 'tconv_forward'),
    'p2r_stgcn_tconv_weight_grad': lambda a: 'tconv_weight_grad' if a[3] == 3 else None,
    # the same launch with the BatchNorm-backward apply pass of the input BatchNorm riding on it (+0.9 GB of HBM traffic)
    'p2r_stgcn_tconv_weight_grad_dz': lambda a: 'tconv_weight_grad' if a[3] == 3 else None,
    # vote / proposal heads on the job-list kernels (csrc/pw_layers.hip): in-step time only
    'p2r_pw_gemm': lambda a: 'heads_pw_gemm',
    'p2r_pw_wgrad': lambda a: 'heads_pw_wgrad',
    'p2r_sa_votes_forward': lambda a: 'vote_aggregation_forward',
    'p2r_sa_votes_backward': lambda a: 'vote_aggregation_backward',
}


def issued_mfma_flops(batch, frames):
    """MFMA FLOPs each ST-GCN kernel ISSUES per launch at this shape (v_mfma_f32_16x16x4_f32 = 2048 FLOP), counted
    from the work tables the kernels run on -- units with an empty neighbour list are skipped, so this is less than
    the dense operator.  Cross-checked against the
    SQ_VALU_MFMA_BUSY_CYCLES x 64 of profiles/*_graphconv_mfma_util.json."""
    import numpy as np
    from pose2room_amd.p2rnet.modules.stgcn_layers import Graph
    from pose2room_amd.p2rnet import gcn_op, gcn_tables
    A = Graph().A
    K, V = A.shape[0], A.shape[1]
    tables = gcn_op.GraphTables(A)
    hdr = gcn_tables.STREAM_UMAX * gcn_tables.STREAM_REC
    tiles16 = batch * ((frames + 15) // 16)
    per_rec = 4 * 16 * 2048.0                       # one record = 16 MFMAs in each of the 4 channel phases
    # (the statically scheduled kernel and the run-time work stream split long lists the same way: same step count)
    if not tables.gen2:      # another skeleton: no work stream to count from (the first-generation kernels serve it)
        return {}, 0.0, 0.0
    out = {'gcn_forward': int(tables.stream_c[:, hdr].sum()) * per_rec * tiles16,
           'gcn_data_gradient': int(tables.stream_r[:, hdr].sum()) * per_rec * tiles16}
    out.update(gcn_op.grad_kernel_mfma_flops(tables, batch, frames))
    out['tconv_forward'] = out['tconv_data_gradient'] = 3 * V * per_rec * tiles16
    out['tconv_weight_grad'] = 3 * 2.0 * 64 * 64 * batch * frames * V
    cols = batch * frames * V
    nnz = int((A != 0).sum())
    dense = (2.0 * 64 * 64 * K + 2.0 * 64 * nnz / V) * cols
    ref = (2.0 * 64 * 64 * K + 2.0 * 64 * V * K) * cols
    return out, dense, ref


def mfma_rooflines(trainer, batch, batch_size, frames, steps=3):
    """`steps` instrumented train steps -> per kernel: in-step average duration, issued MFMA FLOPs, fraction of the
    fp32 MFMA peak.  Returns (roofline dict of the dominant kernel, table of the others)."""
    from pose2room_amd import _lib
    issued, dense, ref = issued_mfma_flops(batch_size, frames)
    with LaunchTimer(_lib.lib(), _TIMED) as lt:
        for _ in range(steps):
            trainer.train_step(dict(batch))
        per = lt.summary()

    def row(tag):
        ms, n = per[tag]
        tf = issued[tag] / ms / 1e9
        return {'ms_in_step': round(ms, 4), 'launches_per_step': n // steps, 'mfma_flops_issued': issued[tag],
                'tflops': round(tf, 2), 'frac': round(tf / FP32_MFMA_PEAK_TFLOPS, 4)}

    rows = {tag: row(tag) for tag in sorted(per) if tag in issued}
    for tag in sorted(per):          # kernels timed without a FLOP model: total in-step time per step
        if tag not in issued:
            ms_, n_ = per[tag]
            rows[tag] = {'ms_in_step_total': round(ms_ * n_ / steps, 4), 'launches_per_step': n_ // steps}
    from pose2room_amd.p2rnet import bn_op
    if bn_op.OVERLAP_APPLY:
        for tag in ('gcn_coef_grad', 'gcn_weight_grad'):
            if tag in rows:
                rows[tag]['hosting'] = ('timed with the BatchNorm-backward reduce + apply passes of the block in front '
                                        'co-resident on a side stream (5 of 6 launches); without guests 0.97 ms (weight) / 1.01 ms (coefficient '
                                        'gradient) per launch, profiles/r4q_bench_line.json')
    g2 = [t for t in ('gcn_forward', 'gcn_data_gradient') if t in per]
    n_g2 = sum(per[t][1] for t in g2)
    ms = sum(per[t][0] * per[t][1] for t in g2) / n_g2             # launch-weighted mean, as rocprofv3 --stats shows it
    fl = sum(issued[t] * per[t][1] for t in g2) / n_g2
    tf = fl / ms / 1e9
    traffic = (_profile_json('r5_gcn3_pmc_traffic.json') or _profile_json('r4_gcn3_pmc_traffic.json')
               or _profile_json('r3_gcn3_pmc_traffic.json'))
    cols = batch_size * frames * 53
    scale = cols / float(32 * 1024 * 53)
    from pose2room_amd.p2rnet import gcn_op
    kname = 'gcn3_kernel (statically scheduled)' if gcn_op.USE_GEN3 else 'gcn2_kernel'
    roof = {'bound': 'mfma', 'kernel': kname + ' (ST-GCN graph conv: 6 forward + 6 data-gradient launches/step)',
            'achieved': round(tf, 2), 'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(tf / FP32_MFMA_PEAK_TFLOPS, 4),
            'traffic': int(traffic['bytes_per_launch'] * scale) if traffic else None,
            'traffic_source': 'stored PMC measurement (profiles/r5_gcn3_pmc_traffic.json: FETCH_SIZE + WRITE_SIZE passes of this kernel at bs=32, T=1024, scaled by the column count), not a counter read in this run',
            'ms_per_launch': round(ms, 4), 'timing': f'HIP events around each of the {n_g2 // steps} launches per step inside '
                                                      f'{steps} instrumented train steps',
            'flops_per_launch': fl, 'flops': 'MFMA FLOPs issued (454 of 583 (plane, joint) units forward, 369 data gradient)',
            'ms_forward': round(per['gcn_forward'][0], 4) if 'gcn_forward' in per else None,
            'ms_data_gradient': round(per['gcn_data_gradient'][0], 4) if 'gcn_data_gradient' in per else None,
            'algorithmic_bytes_per_launch': 2 * 4 * 64 * cols}
    return roof, rows


# split16 mode: the entry points the ST-GCN blocks call instead (include/p2r_hip.h, "opt-in split16 arithmetic"); the
# weight gradient of the temporal conv (HBM-bound with the BatchNorm-backward apply pass on board) stays on the exact
# kernel in this mode
_TIMED16 = {
    'p2r_stgcn_gcn3h_forward': lambda a: None if _null(a[10]) else 'gcn_forward',
    'p2r_stgcn_gcn3h_data_gradient': lambda a: 'gcn_data_gradient',
    'p2r_stgcn_tconvh_forward': lambda a: None if _null(a[9]) else ('tconv_data_gradient' if _null(a[4]) else 'tconv_forward'),
    'p2r_stgcn_gcn3_coef_grad': lambda a: 'gcn_coef_grad (exact kernel)',
    'p2r_stgcn_gcn3h_coef_grad': lambda a: 'gcn_coef_grad',
    'p2r_stgcn_gcn3_weight_grad': lambda a: 'gcn_weight_grad (exact kernel)',
    'p2r_stgcn_gcn3h_weight_grad': lambda a: 'gcn_weight_grad',
    'p2r_stgcn_tconv_weight_grad_dz_amax': lambda a: 'tconv_weight_grad (exact kernel)',
    'p2r_absmax_bits': lambda a: 'range_word_fallback_pass',
}


def split16_rooflines(trainer, batch, batch_size, frames, steps=3):
    """In-step durations of the split16 kernels and their rate on the 16-bit MFMA FLOPs they ISSUE
    (v_mfma_f32_16x16x32_f16 = 16,384 FLOP; three per fp32-equivalent product) against the dense fp16 MFMA peak."""
    from pose2room_amd import _lib
    from pose2room_amd.p2rnet import gcn_op
    from pose2room_amd.p2rnet.modules.stgcn_layers import Graph
    tables = gcn_op.GraphTables(Graph().A)
    tiles16 = batch_size * (frames // 16)
    issued = {}
    if tables.gen3h:
        uc, ur = gcn_op.split_unit_counts(tables)
        issued['gcn_forward'] = uc * 12 * 4 * 16384.0 * tiles16          # (pair, joint) units x 12 MFMAs x 4 channel phases
        issued['gcn_data_gradient'] = ur * 12 * 4 * 16384.0 * tiles16
    issued['tconv_forward'] = issued['tconv_data_gradient'] = 4 * 72 * 16384.0 * batch_size * frames   # 4 waves x 72 per frame
    if tables.gen3h:        # adjacency gradient: 6 MFMAs per live (plane, joint) unit of the row lists, 16-frame tile and row phase
        import numpy as np
        g_ = tables.gidx_r.numpy()
        lofs_ = np.concatenate([[0], np.cumsum(tables.Lk_r)])
        live_ = sum(int((g_[lofs_[k]:lofs_[k + 1]] >= 0).any(0).sum()) for k in range(tables.K))
        issued['gcn_coef_grad'] = live_ * 4 * 6 * 16384.0 * tiles16
    if tables.gen3h:        # 56 live (plane, 8-joint group) units x 24 MFMAs x 2 column halves per 4-frame tile
        issued['gcn_weight_grad'] = gcn_op.split_weight_grad_units(tables) * 24 * 2 * 16384.0 * batch_size * (frames // 4)
    with LaunchTimer(_lib.lib(), _TIMED16) as lt:
        for _ in range(steps):
            trainer.train_step(dict(batch))
        per = lt.summary()
    rows = {}
    for tag in sorted(per):
        ms, n = per[tag]
        if tag in issued:
            tf = issued[tag] / ms / 1e9
            rows[tag] = {'ms_in_step': round(ms, 4), 'launches_per_step': n // steps, 'mfma16_flops_issued': issued[tag],
                         'tflops': round(tf, 1), 'frac': round(tf / FP16_MFMA_PEAK_TFLOPS, 4)}
        else:
            rows[tag] = {'ms_in_step': round(ms, 4), 'launches_per_step': round(n / steps, 2)}
    g = [t for t in ('gcn_forward', 'gcn_data_gradient') if t in per and t in issued]
    roof = None
    if g:
        n_g = sum(per[t][1] for t in g)
        ms = sum(per[t][0] * per[t][1] for t in g) / n_g
        fl = sum(issued[t] * per[t][1] for t in g) / n_g
        tf = fl / ms / 1e9
        roof = {'bound': 'mfma', 'kernel': 'gcn3h kernels (split16 graph conv: 6 forward + 6 data-gradient launches/step)',
                'achieved': round(tf, 1), 'peak': FP16_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(tf / FP16_MFMA_PEAK_TFLOPS, 4),
                'traffic': None, 'ms_per_launch': round(ms, 4), 'flops_per_launch': fl,
                'flops': '16-bit MFMA FLOPs issued (v_mfma_f32_16x16x32_f16; 3 per fp32-equivalent product)',
                'timing': f'HIP events around each launch inside {steps} instrumented split16 train steps'}
    return roof, rows


def step_mfma_issued(batch, frames):
    """MFMA FLOPs the step's kernels issue (SQ_VALU_MFMA_BUSY_CYCLES x 64 FLOP/cycle/SIMD summed over one step, PMC
    profile from tools/pmc_step_mfma.sh at bs=32, T=1024; linear in batch * frames).  (flops, source) or (None, None)."""
    for name in ('r5_step_mfma.json', 'r4_step_mfma.json', 'r3_step_mfma.json', 'r2_step_mfma.json'):
        prof = _profile_json(name)
        if prof:
            return prof['mfma_busy_cycles_per_step'] * 64.0 * (batch * frames) / float(32 * 1024), 'profiles/' + name
    return None, None


def verify_bench_shape(trainer, batch):
    """Before anything is timed: (1) one train step at the bench shape gives finite losses; (2) with every BatchNorm
    on its running statistics samples are independent, so sample 0 of the full batch must reproduce when it is run
    alone -- this walks the persistent-workgroup kernels through tile counts far beyond one wave of workgroups and
    catches batch- / tile-indexing faults that the small test shapes cannot.  Continuous tensors must agree to 1e-3
    of their range (different GEMM tilings per batch size), the seed selection exactly; the proposal indices are
    discrete functions of fp32 values (FPS over vote positions) and are REPORTED, not asserted."""
    import math
    out = trainer.train_step(dict(batch))
    bad = [k for k, v in out.items() if not math.isfinite(float(v))]
    assert not bad, f'non-finite losses at the bench shape: {bad}'
    net = trainer.net.module
    was_training = net.training
    net.eval()
    try:
        def front(data):      # backbone -> votes -> vote aggregation (no mixture heads: they draw noise in training)
            xyz, features, ep = net._votes(data)
            net.detection._aggregate(xyz, features, ep)
            return ep
        with torch.no_grad():
            full = front(batch)
            one = front({k: (v[:1].contiguous() if torch.is_tensor(v) else v) for k, v in batch.items()})
    finally:
        net.train(was_training)
    assert torch.equal(full['seed_inds'][:1], one['seed_inds']), 'seed_inds of sample 0 differ between B and 1'
    worst = 0.0
    for k in ('seed_features', 'vote_xyz', 'vote_features'):
        a, b = full[k][:1].float(), one[k].float()
        assert torch.isfinite(a).all() and torch.isfinite(b).all(), k
        err = (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)
        worst = max(worst, err)
        assert err <= 1e-3, f'{k} of sample 0: batched vs alone differ by {err:.2e} of range'
    same = torch.equal(full['aggregated_vote_inds'][:1], one['aggregated_vote_inds'])
    # Exact part, asserted: the proposal indices are a function of the sample's own votes -- FPS of sample 0's votes run
    # alone reproduces the batch's indices bit for bit, and the two runs can only pick different proposals when their
    # (fp32, differently batched GEMM) vote coordinates differ somewhere.  `aggregated_vote_inds_equal` then reports
    # whether such a rounding difference flipped a discrete choice in this run (never observed at this shape).
    from pose2room_amd.pointnet2_ops import _ext
    alone = torch.sort(_ext.furthest_point_sampling(full['vote_xyz'][:1].contiguous(), full['aggregated_vote_inds'].shape[1]).long(), dim=-1)[0]
    assert torch.equal(alone, full['aggregated_vote_inds'][:1]), 'FPS of sample 0 depends on the rest of the batch'
    assert same or not torch.equal(full['vote_xyz'][:1], one['vote_xyz']), \
        'identical votes gave different proposal indices'
    return {'losses_finite': True, 'seed_inds_equal': True, 'max_rel_err_sample0': float(f'{worst:.2e}'),
            'aggregated_vote_inds_equal': bool(same)}


def kernel_microbench(device):
    """Event-timed durations of the HIP kernels at the P2RNet shapes (B=32, N=512,
    npoint=128, nsample=16, C=256), on the stream they are launched on."""
    from pose2room_amd.pointnet2_ops import _ext
    from pose2room_amd.net_utils.nn_distance import nn_distance
    B, N, P, S, C = 32, 512, 128, 16, 256
    g = torch.Generator().manual_seed(0)
    xyz = (torch.randn(B, N, 3, generator=g) * 0.5).to(device)
    feats = torch.randn(B, C, N, generator=g).to(device)
    inds = _ext.furthest_point_sampling(xyz, P)
    new_xyz = _ext.gather_points(xyz.transpose(1, 2).contiguous(), inds).transpose(1, 2).contiguous()
    idx = _ext.ball_query(new_xyz, xyz, 0.3, S)
    grouped = _ext.group_points(feats, idx)
    pc1 = torch.randn(B * 512, 3, 3, generator=g).to(device)
    pc2 = torch.randn(B * 512, 53, 3, generator=g).to(device)
    ops = {
        'furthest_point_sampling': (lambda: _ext.furthest_point_sampling(xyz, P), None),
        'ball_query': (lambda: _ext.ball_query(new_xyz, xyz, 0.3, S), None),
        'group_points_c256': (lambda: _ext.group_points(feats, idx), 4.0 * B * C * P * S * 2 + 4.0 * B * P * S),
        'group_points_grad_c256': (lambda: _ext.group_points_grad(grouped, idx, N), 4.0 * B * C * (P * S + N)),
        'nn_distance_vote': (lambda: nn_distance(pc1, pc2), None),
    }
    out = {}
    stream = torch.cuda.current_stream(device)
    for name, (fn, bytes_) in ops.items():
        for _ in range(5):
            fn()
        reps = 50
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        out[name] = {'us': round(us, 2)}
        if bytes_:
            # the gathered source (B*C*N*4 = 17 MB) is L2 / MALL resident: an L2-assisted rate, not HBM bandwidth
            out[name]['GBps_l2_assisted'] = round(bytes_ / us / 1e3, 1)
    return out


def cpu_baseline(frames, budget_s=25.0):
    """The same host model on the host cores, CPU oracle behind the ops ("port")."""
    from oracle.cpu_backend import cpu_ops
    from pose2room_amd.p2rnet.synthetic import make_batch
    # oneDNN / OpenMP scale poorly past ~32 threads at this batch size: use at most 32 cores
    prev_threads = torch.get_num_threads()
    cores = min(os.cpu_count() or prev_threads, 32)
    torch.set_num_threads(cores)
    trainer, _ = build_trainer_cpu(frames)
    B = 2
    batch = make_batch(B, frames, seed=1234)
    with cpu_ops():
        t0 = time.time()
        trainer.train_step(dict(batch))          # warm-up (allocator, oneDNN primitives)
        warm = time.time() - t0
        n, t0 = 0, time.time()
        while n < 1 or (time.time() - t0 + warm) < budget_s and n < 8:
            trainer.train_step(dict(batch))
            n += 1
        dt = (time.time() - t0) / n
    torch.set_num_threads(prev_threads)
    return {'value': round(B / dt, 4), 'unit': 'samples/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n} train steps of bs={B}, T={frames}, J=53 after 1 warm-up (host model + CPU oracle ops)',
            # the reference's OWN Python (imported, CPU restatement of `_ext` behind it) cannot travel to this box; it was
            # timed in the survey container: same step, bs=2, T=1024
            'reference_python': {'value': 0.58, 'unit': 'samples/s', 'cores': 8, 'kind': 'reference',
                                 'where': 'survey container (8 host cores), BASELINE.md section 2; not measured in this run'}}


def build_trainer_cpu(frames):
    from pose2room_amd.p2rnet import P2RConfig, default_config, METHODS
    from pose2room_amd.p2rnet.training import Trainer, ModuleWrapper, load_optimizer
    cfg = P2RConfig(default_config('train', data={'num_frames': frames}), device='cpu')
    torch.manual_seed(42)
    net = ModuleWrapper(METHODS.get('P2RNet')(cfg))
    return Trainer(cfg, net, load_optimizer(cfg.config, net), torch.device('cpu')), cfg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=32, help='per-GPU batch (BASELINE: 32)')
    ap.add_argument('--frames', type=int, default=1024, help='T (BASELINE: 1024)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-microbench', action='store_true')
    ap.add_argument('--no-split16', action='store_true', help='skip the labelled split16 sub-measurement')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    assert torch.cuda.is_available(), 'bench.py needs a GPU (no CPU fallback for the product path)'
    # P2R_BENCH_SHARE_GPU=1 (testing only): all ranks on cuda:0 with gloo collectives, to exercise the multi-rank
    # code path (DDP hooks on the HIP autograd functions, barriers, max-over-ranks timing) on a 1-GPU box
    share = os.environ.get('P2R_BENCH_SHARE_GPU') == '1'
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        if share:
            dist.init_process_group(backend='gloo', init_method='env://')
        else:
            dist.init_process_group(backend='nccl', init_method='env://', device_id=device)
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'

    trainer, cfg = build_trainer(device, args.frames, world)
    # The batch arrives the way the reference's epoch loop gets it (models/p2rnet/dataloader.py:172-199): the loader
    # over a seeded synthetic dataset, one DistributedSampler shard per rank.
    from pose2room_amd.p2rnet.dataloader import P2RNet_dataloader, SyntheticPoseDataset
    cfg.config['device']['distributed'] = world > 1
    cfg.config['train']['batch_size'] = args.batch
    dataset = SyntheticPoseDataset(args.batch * world, args.frames, seed=1234)
    loader = P2RNet_dataloader(cfg, 'train', dataset=dataset)
    if world > 1:
        loader.sampler.set_epoch(0)
    host_batch = next(iter(loader.dataloader))
    assert host_batch['input_joints'].shape[0] == args.batch
    batch = trainer.to_device(dict(host_batch))                      # resident in HBM

    def step():
        return trainer.train_step(dict(batch))

    verify = verify_bench_shape(trainer, batch)      # not timed, not part of the warm-up count
    for _ in range(max(args.warmup - 1, 0)):
        step()
    # Python's cyclic GC: the first full (generation-2) collection of a process walks every object torch and the
    # model have created -- a ~100 ms host pause that lands around the 11th step (measured, tools/step_times2.py).
    # Collect now and freeze the survivors so later collections only look at the objects of the steps themselves.
    # The collection goes in front of the LAST warm-up step: the device sits idle under it and drops its clocks, and a
    # timed region that starts right behind it pays 2-3 ms on its first step (per-step events: 43.2, 40.6, 40.8, ... ms);
    # behind one more warm-up step it starts at the clocks the other steps run at.
    gc.collect()
    gc.freeze()
    if args.warmup > 0:
        step()

    def timed(run_step):
        """K steps between barrier + synchronize; also the per-step device time from events between the steps."""
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(args.steps):
            out = run_step()
            marks[i + 1].record()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
        q = lambda f: per_step[min(len(per_step) - 1, int(round(f * (len(per_step) - 1))))]
        return dt, out, {'median': round(q(0.5), 3), 'p10': round(q(0.1), 3), 'p90': round(q(0.9), 3)}

    elapsed, last, step_ms = timed(step)

    # ---- the same steps with the reference's to_device inside the step (models/p2rnet/training.py:100-107): the batch
    # sits in pinned host memory and is copied one step ahead on a side stream
    pinned = {k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in host_batch.items()}
    h2d_bytes = sum(v.numel() * v.element_size() for v in pinned.values() if torch.is_tensor(v))
    copy_stream = torch.cuda.Stream(device)
    state = {}

    def prefetch():
        with torch.cuda.stream(copy_stream):
            state['next'] = {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in pinned.items()}
            state['ready'] = torch.cuda.Event()
            state['ready'].record(copy_stream)

    def step_h2d():
        cur, ready = state['next'], state['ready']
        torch.cuda.current_stream(device).wait_event(ready)
        for v in cur.values():
            if torch.is_tensor(v):
                v.record_stream(torch.cuda.current_stream(device))
        prefetch()                                   # the next batch crosses PCIe under this step's kernels
        return trainer.train_step(cur)

    prefetch()
    step_h2d()
    elapsed_h2d, _, step_ms_h2d = timed(step_h2d)

    # instrumented steps (events around the MFMA kernels' launches): every rank runs them -- they contain the gradient
    # all-reduce -- rank 0 reports
    roof, rows = mfma_rooflines(trainer, batch, args.batch, args.frames)

    ddp_info = None
    if world > 1:
        # (1) spread of the ranks' own step times; (2) the same K steps with the gradient all-reduce switched off
        # (DDP.no_sync: hooks do not fire, gradients stay local) = what the all-reduce costs on top of a step, i.e. the
        # part of it that is not hidden under the backward pass.  Runs last: the ranks' weights diverge under no_sync.
        mine = torch.tensor([step_ms['median']], dtype=torch.float64, device=device)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [round(float(t.item()), 3) for t in every]

        def step_no_sync():
            with trainer.net.no_sync():
                return trainer.train_step(dict(batch))
        step_no_sync()
        # the two forms back to back, twice, in the same thermal / clock state (a difference against the headline block,
        # timed minutes earlier, is mostly noise: each number is a max over ranks with ~1 % spread)
        rounds = []
        for _ in range(2):
            e_sync, _, _ = timed(step)
            e_ns, _, _ = timed(step_no_sync)
            rounds.append((e_sync / args.steps * 1e3, e_ns / args.steps * 1e3))
        exposed = [a - b for a, b in rounds]
        ddp_info = {'rank_step_ms_median': per_rank, 'rank_spread_ms': round(max(per_rank) - min(per_rank), 3),
                    'ms_per_step_no_sync': round(min(b for _, b in rounds), 3),
                    'ms_per_step_sync_same_block': round(min(a for a, _ in rounds), 3),
                    'allreduce_exposed_ms_per_step': round(max(0.0, min(exposed)), 3),
                    'allreduce_exposed_rounds_ms': [round(e, 3) for e in exposed],
                    'how': '2 x (K steps with the gradient all-reduce, K steps under DDP.no_sync()) back to back; exposed = '
                           'the smaller difference, clamped at 0 (noise floor ~1 % of a step); the 80-byte logging '
                           'all-reduce is in both', 'gradient_bytes_per_step': 8176132,
                    'backend': dist.get_backend()}

    # ---- the same steps in the opt-in split16 arithmetic (every rank runs them: they contain the gradient all-reduce).
    # After everything the exact path needs; the mode is switched back before the line is printed.
    split16 = None
    if not args.no_split16 and args.frames % 16 == 0:
        from pose2room_amd.p2rnet import math_mode
        math_mode.set_mode('split16')
        try:
            verify16 = verify_bench_shape(trainer, batch)
            for _ in range(args.warmup):
                step()
            passes0 = math_mode.FALLBACK_PASSES
            elapsed16, last16, step_ms16 = timed(step)
            passes = (math_mode.FALLBACK_PASSES - passes0) / float(args.steps)
            roof16, rows16 = split16_rooflines(trainer, batch, args.batch, args.frames)
        finally:
            math_mode.set_mode('exact')
            math_mode.reset()
        split16 = {'value': round(world * args.batch * args.steps / elapsed16, 3), 'unit': 'samples/s',
                   'ms_per_step': round(elapsed16 / args.steps * 1e3, 3), 'step_ms': step_ms16, 'verify': verify16,
                   'dtype': 'f32 products as three fp16 MFMA products of two-part operands (22 of 24 significand bits), '
                            'fp32 accumulation -- ST-GCN graph conv forward / data gradient and temporal conv forward / data '
                            'gradient; every other kernel exact fp32',
                   'loss_total': round(float(last16['total']), 4), 'range_word_fallback_passes_per_step': passes,
                   'roofline': roof16, 'kernels': rows16, 'speedup_vs_exact': round(elapsed / elapsed16, 4),
                   'how': 'P2R_MATH=split16 / math_mode.set_mode; same trainer, same batch, same K and W as the headline'}

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        sps = world * args.batch * args.steps / elapsed
        line = {
            'metric': f'P2RNet train-step samples/sec (T={args.frames},J=53,bs={args.batch})', 'value': round(sps, 3),
            'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic', 'step_ms': step_ms,
            'h2d': {'value': round(world * args.batch * args.steps / elapsed_h2d, 3), 'unit': 'samples/s',
                    'ms_per_step': round(elapsed_h2d / args.steps * 1e3, 3), 'step_ms': step_ms_h2d,
                    'bytes_per_step': h2d_bytes,
                    'how': 'batch copied from pinned host memory every step (reference to_device), one step ahead on a '
                           'side stream'},
            'config': {'workload': f'P2RNet full train step, bs={args.batch}/GPU, T={args.frames}, J=53, '
                                   f'seeds=512, proposals=128'
                                   + (' (BASELINE configs[2])' if (args.batch, args.frames) == (32, 1024) else ''),
                       'input': 'P2RNet_dataloader over SyntheticPoseDataset' + (' + DistributedSampler' if world > 1 else ''),
                       'global_batch': world * args.batch, 'frames': args.frames,
                       'parallelism': f'dp{world}', 'loss_total': round(float(last['total']), 4)},
            'verify': verify,
        }
        if ddp_info:
            line['ddp'] = ddp_info
        if split16:
            line['split16'] = split16
        line['roofline'] = roof
        line['mfma_kernels'] = rows
        issued, src = step_mfma_issued(args.batch, args.frames)
        ex_tf = issued / (ms * 1e-3) / 1e12 if issued else None
        line['step_roofline'] = {
            'bound': 'mfma', 'achieved': round(ex_tf, 2) if issued else None, 'peak': FP32_MFMA_PEAK_TFLOPS * world,
            'unit': 'TFLOP/s', 'frac': round(ex_tf / FP32_MFMA_PEAK_TFLOPS, 4) if issued else None,
            'scope': f'whole train step of one rank: MFMA FLOPs issued per step ({src}) / step time',
            'derived': 'the FLOP count is a stored PMC measurement (SQ_VALU_MFMA_BUSY_CYCLES x 64 summed over one step of '
                       'the build that profile was taken on, at bs=32, T=1024, scaled by batch x frames), not a counter '
                       'read in this run; the step time is this run\'s',
            'reference_algorithmic_gflop_per_sample': gflop_per_sample(args.frames)}
        if not args.no_microbench:
            line['kernels'] = kernel_microbench(device)
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args.frames)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
