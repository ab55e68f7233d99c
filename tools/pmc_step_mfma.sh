#!/bin/bash
# MFMA work actually issued by one train step: SQ_VALU_MFMA_BUSY_CYCLES summed over every dispatch of
# tools/step_workload.py (STEPS identical steps, bs=32, T=1024) / STEPS.  One busy cycle = 64 fp32 MFMA FLOP on one SIMD.
#   bash tools/pmc_step_mfma.sh -> gpurun_out/step_mfma.json  (copy to profiles/)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out; rm -rf /tmp/pm_step
export STEPS=3
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/pm_step -- python $R/tools/step_workload.py > $R/gpurun_out/pm_step.log 2>&1
python - <<PY
import csv, glob, json, collections
cc = glob.glob('/tmp/pm_step/**/*counter_collection.csv', recursive=True)[0]
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(cc)):
    if r['Counter_Name'] != 'SQ_VALU_MFMA_BUSY_CYCLES': continue
    k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].split('<')[0].split('::')[-1].strip()[-60:] or r['Kernel_Name'][:60]
    tot[k] += float(r['Counter_Value']); n[k] += 1
steps = $STEPS
total = sum(tot.values()) / steps
top = sorted(tot.items(), key=lambda kv: -kv[1])[:12]
out = {'source': 'tools/pmc_step_mfma.sh: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES over tools/step_workload.py (%d train steps, bs=32, T=1024), MI355X' % steps,
       'mfma_busy_cycles_per_step': total, 'mfma_flops_per_step': total * 64.0,
       'by_kernel_per_step': {k: {'busy_cycles': v / steps, 'launches': n[k] / steps} for k, v in top}}
json.dump(out, open('$R/gpurun_out/step_mfma.json', 'w'), indent=1)
print(json.dumps(out, indent=1)[:1500])
PY
