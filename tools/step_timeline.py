"""Ordered kernel timeline of ONE steady-state train step from a rocprofv3 kernel trace.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $R/tools/step_workload.py
    python tools/step_timeline.py /tmp/tl/**/*kernel_trace.csv [--step -1] [--seq] [--groups]

The step boundary is the once-per-step kernel `det_loss_forward_kernel`: a step is cut at the optimiser kernel
(`multi_tensor_apply_kernel` ... FusedAdam) that follows it.  Prints the span of the step, the time the device
was busy, the idle gaps, a per-kernel table and (with --seq) the launch sequence with the gap in front of every launch.
`--groups` sums the launches by family: own HIP kernels / MIOpen / hipBLASLt (Cijk_*) / ATen / copies.
"""
import argparse
import collections
import csv
import glob
import re
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'at::native::(?:\w+::)*(\w+)<.*?at::native::(?:\(anonymous namespace\)::)?(\w+)', name)
    if name.startswith('at::native'):
        inner = re.findall(r'at::native::(?:\(anonymous namespace\)::|binary_internal::)?(\w+)', name)
        return 'aten:' + '/'.join(dict.fromkeys(inner[:3]))
    if name.startswith('Cijk_'):
        mt = re.search(r'MT(\d+x\d+x\d+)', name)
        return 'hipblaslt:' + name[:14] + (':MT' + mt.group(1) if mt else '')
    return name.split('(')[0][:70]


def family(name):
    if name.startswith('Cijk_'):
        return 'hipBLASLt'
    if name.startswith('void at::native') or name.startswith('at::native'):
        return 'ATen'
    if 'MIOpen' in name or name.startswith('igemm_') or name.startswith('batched_transpose') \
            or name.startswith('SubTensorOp') or 'miopen' in name:
        return 'MIOpen'
    if name.startswith('__amd_rocclr'):
        return 'runtime copy/fill'
    return 'own HIP'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('trace')
    ap.add_argument('--step', type=int, default=-1, help='which complete step (default: the last one)')
    ap.add_argument('--seq', action='store_true')
    ap.add_argument('--groups', action='store_true')
    ap.add_argument('--rows', type=int, default=60)
    args = ap.parse_args()
    files = glob.glob(args.trace, recursive=True)
    if not files:
        sys.exit('no trace file matches ' + args.trace)
    rows = []
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    # step cuts: first optimiser kernel after each det_loss_forward
    cuts, seen_loss = [], False
    for i, (s, e, n) in enumerate(rows):
        if 'det_loss_forward_kernel' in n:
            seen_loss = True
        elif seen_loss and 'FusedAdam' in n:
            # the last optimiser launch of the step: advance over the run of optimiser / clip kernels
            j = i
            while j + 1 < len(rows) and 'FusedAdam' in rows[j + 1][2]:
                j += 1
            cuts.append(j + 1)
            seen_loss = False
    if len(cuts) < 2:
        sys.exit(f'need two step boundaries, found {len(cuts)}')
    k = args.step if args.step >= 0 else len(cuts) - 2
    lo, hi = cuts[k], cuts[k + 1]
    step = rows[lo:hi]
    span = step[-1][1] - step[0][0]
    # kernels of two streams may overlap (the BatchNorm-backward apply pass on its side stream): the device is busy over
    # the UNION of the intervals, the gaps are what the union leaves of the span
    total = sum(e - s for s, e, _ in step)
    gaps, cover_end, busy = [], step[0][0], 0
    for s_, e_, _ in sorted((s, e, n) for s, e, n in step):
        if s_ > cover_end:
            gaps.append(s_ - cover_end)
            busy += e_ - s_
        elif e_ > cover_end:
            busy += e_ - cover_end
        cover_end = max(cover_end, e_)
    print(f'step {k}: {len(step)} launches, span {span / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, '
          f'idle {sum(gaps) / 1e6:.3f} ms (gaps > 20 us: {sum(1 for g in gaps if g > 20000)})'
          + (f', kernel time {total / 1e6:.3f} ms ({(total - busy) / 1e6:.3f} ms of it overlapped)' if total - busy > 1000 else ''))
    cnt, tim = collections.Counter(), collections.Counter()
    fam_c, fam_t = collections.Counter(), collections.Counter()
    for s, e, n in step:
        cnt[short(n)] += 1
        tim[short(n)] += e - s
        fam_c[family(n)] += 1
        fam_t[family(n)] += e - s
    if args.groups:
        for f_, t in fam_t.most_common():
            print(f'  {f_:20s} {fam_c[f_]:5d} launches {t / 1e6:8.3f} ms')
    for n, t in tim.most_common(args.rows):
        print(f'{cnt[n]:5d} {t / 1e6:8.3f} ms  {n}')
    if args.seq:
        t0 = step[0][0]
        prev = None
        for s, e, n in step:
            gap = 0 if prev is None else s - prev
            print(f'{(s - t0) / 1e3:10.1f} us  +{gap / 1e3:6.1f}  {(e - s) / 1e3:8.1f} us  {short(n)}')
            prev = e


if __name__ == '__main__':
    main()
