#!/bin/bash
# Dev: SQ counters of the graph-conv / tconv kernels (two passes of 8 SQ counters), summarised per kernel.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/pmc
REPS=1 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmc/p1 -- python $R/tools/dev_gcn_time.py > $R/gpurun_out/pmc/p1.log 2>&1
REPS=1 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/pmc/p2 -- python $R/tools/dev_gcn_time.py > $R/gpurun_out/pmc/p2.log 2>&1
python - <<PY
import csv, glob, collections
for d in ('p1', 'p2'):
    fs = glob.glob('$R/gpurun_out/pmc/%s/**/*counter_collection.csv' % d, recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in fs:
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[1][:40] if r['Kernel_Name'].startswith('(anon') else r['Kernel_Name'][:40]
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:6]:
        print(d, k, {a: '%.3g' % b for a, b in v.items()})
PY
