#!/bin/bash
# Wave-level stall anatomy of the prototype graph conv (tools/ubench/gcn3h_proto.hip) next to the product's gcn3 kernel:
# instruction cache, LDS / vector-memory queue levels, active and waiting cycles.  Separate --pmc passes.
#   bash tools/pmc_icache.sh   (GPU box)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
run() {
  rm -rf /tmp/pi
  REPS=1 timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pi -- python $R/tools/dev_gcn_f16.py > /tmp/pi.log 2>&1
  python - <<PY
import csv, glob, collections
cc = glob.glob('/tmp/pi/**/*counter_collection.csv', recursive=True)
if cc:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc[0])):
        n = r['Kernel_Name']
        key = 'gcn3h' if 'gcn3h_kernel' in n else ('gcn3' if 'gcn3_kernel' in n else None)
        if key: acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        print(k, {c: round(sum(v) / len(v)) for c, v in d.items()})
else:
    print(open('/tmp/pi.log').read()[-600:])
PY
}
run SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_IFETCH
run SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM
run SQ_WAVE_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16
