#!/bin/bash
# A/B on ONE box: bench.py's in-step kernel times with two builds of the library, alternating.
#   bash tools/ab_bench.sh tmp_ab/libp2r_base.so [rounds]   (B = the in-tree library)
A=$(realpath $1); N=${2:-2}
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
for i in $(seq $N); do
  P2R_LIB_PATH=$A python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-microbench --no-split16 > gpurun_out/ab_A$i.json 2>/dev/null
  python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-microbench --no-split16 > gpurun_out/ab_B$i.json 2>/dev/null
done
python - <<PY
import json, glob
def load(f): return json.loads(open(f).read().strip().splitlines()[-1])
for tag in ('A', 'B'):
    runs = [load(f) for f in sorted(glob.glob('gpurun_out/ab_%s*.json' % tag))]
    print(tag, 'samples/s', [r['value'] for r in runs], 'step median ms', [r['step_ms']['median'] for r in runs])
    keys = sorted(runs[0]['mfma_kernels'])
    for k in keys:
        v = [r['mfma_kernels'][k].get('ms_in_step', r['mfma_kernels'][k].get('ms_in_step_total')) for r in runs]
        print('   %-28s' % k, v)
PY
