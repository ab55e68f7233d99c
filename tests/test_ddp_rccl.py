"""GPU, multi-rank: data-parallel P2RNet training across the GPUs of one node.

`test_ddp_rccl` is the real thing -- one process per GPU, backend 'nccl' (= RCCL over xGMI) -- and needs at least
two visible GPUs; on the 1-GPU test box it is skipped and `test_ddp_worker_shared_gpu` runs the very same worker with
both ranks on cuda:0 over gloo, so the worker itself (loader shards, lock-step parameters, bucket layout) is
exercised wherever a GPU exists.  Reference: net_utils/utils.py:235-255 (DDP wrap), :441-446 (all-reduce of the
logged scalars), models/p2rnet/dataloader.py:173-197 (DistributedSampler)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    return port


def _launch(script_args, nproc, extra_env):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', **extra_env)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc),
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port())] + script_args
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0]), out


def _check_worker_line(line, world):
    assert line['world'] == world and line['steps'] == 2
    # 2,043,833 parameters: 2,043,633 f32 (8.17 MB) + 200 f64 heading means (1.6 KB); with bucket_cap_mb=16 the f32
    # payload is ONE bucket = one all-reduce per step, the f64 tensors ride in their own small bucket
    sizes = [int(s) for s in line['bucket_sizes'].replace(',', ' ').split() if s.strip().isdigit()]
    assert sizes, line
    assert sum(sizes) == 2043633 * 4 + 200 * 8, sizes
    assert len(sizes) <= 2 and max(sizes) >= 2043633 * 4, sizes


def test_ddp_worker_shared_gpu():
    line, out = _launch([os.path.join(ROOT, 'tests', 'ddp_worker.py')], 2, {'P2R_BENCH_SHARE_GPU': '1'})
    assert line['backend'] == 'gloo'
    _check_worker_line(line, 2)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs >= 2 GPUs on the node (RCCL)')
def test_ddp_rccl():
    n = min(torch.cuda.device_count(), 8)
    line, out = _launch([os.path.join(ROOT, 'tests', 'ddp_worker.py')], n, {})
    assert line['backend'] == 'nccl'
    _check_worker_line(line, n)
    # the benchmark itself on RCCL: one JSON line, whole-job throughput over n GPUs
    bench_line, out = _launch([os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--steps', '3', '--warmup', '1', '--batch',
                               '8', '--frames', '256', '--no-cpu-baseline', '--no-microbench'], n, {})
    assert bench_line['n_gpus'] == n and bench_line['config']['global_batch'] == 8 * n
    assert bench_line['config']['parallelism'] == f'dp{n}' and bench_line['value'] > 0
    print(f"RCCL dp{n}: {bench_line['value']:.1f} samples/s at bs=8, T=256 per GPU (weak scaling; efficiency is the driver's to compute)")
