"""GPU: the multi-rank path of bench.py (DDP gradient hooks on the HIP autograd functions, barriers, max-over-ranks
timing, per-rank shards) with two ranks sharing the one GPU of the test box over gloo (P2R_BENCH_SHARE_GPU=1; the real
launch is one rank per GPU over RCCL, which a 1-GPU box cannot run)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_share_gpu():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, P2R_BENCH_SHARE_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
           '--batch', '4', '--frames', '128', '--no-cpu-baseline', '--no-microbench']
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]                 # rank 0 prints the one JSON line
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['config']['global_batch'] == 8 and line['scaling'] == 'weak'
    assert line['value'] > 0 and line['config']['parallelism'] == 'dp2'
    # multi-rank report: each rank's own step time and the cost of the gradient all-reduce (steps with / without it)
    ddp = line['ddp']
    assert len(ddp['rank_step_ms_median']) == 2 and all(v > 0 for v in ddp['rank_step_ms_median'])
    assert ddp['ms_per_step_no_sync'] > 0 and 'allreduce_exposed_ms_per_step' in ddp and ddp['backend'] == 'gloo'
    # the labelled split16 sub-measurement is part of rank 0's ONE line (every rank ran its steps: they contain the
    # gradient all-reduce), with its own verification
    s16 = line['split16']
    assert s16['value'] > 0 and s16['ms_per_step'] > 0 and s16['verify']['losses_finite']
    assert line['dtype'] == 'f32' and 'split16' not in line['metric']            # never the headline
    # gradients must arrive in the layout DDP's bucket views expect (no silent extra copies)
    assert 'strides' not in out.stderr, out.stderr[-2000:]
