#!/bin/bash
# Dev: effective shader clock of the graph-conv kernels = GRBM_GUI_ACTIVE / kernel duration.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pmc_clk; mkdir -p $R/gpurun_out
REPS=1 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d /tmp/pmc_clk -- python $R/tools/dev_gcn_time.py > $R/gpurun_out/pmc_clk.log 2>&1
python - <<PY
import csv, glob, collections
cc = glob.glob('/tmp/pmc_clk/**/*counter_collection.csv', recursive=True)[0]
kt = glob.glob('/tmp/pmc_clk/**/*kernel_trace.csv', recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[int(r['Dispatch_Id'])] = (int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name'])
agg = collections.defaultdict(list)
for r in csv.DictReader(open(cc)):
    if r['Counter_Name'] != 'GRBM_GUI_ACTIVE': continue
    d = int(r['Dispatch_Id'])
    if d in dur and dur[d][0] > 200000:
        name = dur[d][1].split('::')[1][:24] if '::' in dur[d][1] else dur[d][1][:24]
        agg[name].append(float(r['Counter_Value']) / dur[d][0])
for k, v in agg.items():
    print(k, 'n=%d' % len(v), 'GUI_ACTIVE cycles per ns: min %.3f max %.3f' % (min(v), max(v)))
PY
