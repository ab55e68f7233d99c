#!/bin/bash
# instruction mix / unit activity of gcn2_kernel (tools/dev_gcn2_exp.py ONLY=base)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pi1 /tmp/pi2
ONLY=base rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d /tmp/pi1 -- python $R/tools/dev_gcn2_exp.py > /dev/null 2>&1
ONLY=base rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_ANY --output-format csv -d /tmp/pi2 -- python $R/tools/dev_gcn2_exp.py > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for d in ('/tmp/pi1', '/tmp/pi2'):
    cc = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not cc: print('no counters in', d); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(cc[0])):
        if 'gcn2_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items(): print(k, sum(v) / len(v))
PY
