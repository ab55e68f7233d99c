#!/bin/bash
# SQ counters of the second-generation graph-conv kernel (tools/dev_gcn2_exp.py, ONLY=<variant>, default base)
#   bash tools/pmc_gcn2.sh [variant] -> gpurun_out/gcn2_pmc_<variant>.json
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
VAR=${1:-base}
mkdir -p $R/gpurun_out; rm -rf /tmp/pm1 /tmp/pm2
ONLY=$VAR rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d /tmp/pm1 -- python $R/tools/dev_gcn2_exp.py > $R/gpurun_out/pm1.log 2>&1
ONLY=$VAR rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm2 -- python $R/tools/dev_gcn2_exp.py > $R/gpurun_out/pm2.log 2>&1
python - <<PY
import csv, glob, json, collections
def load(d):
    cc = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    kt = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
    dur = {int(r['Dispatch_Id']): (int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(kt))}
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(cc)):
        d_ = int(r['Dispatch_Id'])
        if d_ in dur and 'gcn2_kernel' in dur[d_][1]:
            acc[r['Counter_Name']].append(float(r['Counter_Value'])); acc['_ns_' + r['Counter_Name']].append(dur[d_][0])
    return acc
a, b = load('/tmp/pm1'), load('/tmp/pm2')
mean = lambda v: sum(v) / len(v)
# launches alternate col / row form: split even / odd halves by duration order is fragile -> report the mean of all
ghz = mean(b['GRBM_GUI_ACTIVE']) / 8.0 / mean(b['_ns_GRBM_GUI_ACTIVE'])
ns = mean(a['_ns_SQ_VALU_MFMA_BUSY_CYCLES']); busy = mean(a['SQ_VALU_MFMA_BUSY_CYCLES']); wave = mean(a['SQ_WAVE_CYCLES'])
out = {'variant': '$VAR', 'launches': len(a['SQ_WAVE_CYCLES']), 'duration_us_under_pmc': round(ns / 1e3, 1), 'shader_clock_GHz': round(ghz, 3),
       'mfma_busy_cycles': busy, 'mfma_pipe_utilisation': round(busy / (1024 * ns * ghz), 3), 'wave_cycles_quad': wave,
       'wait_any_frac': round(mean(a['SQ_WAIT_ANY']) / wave, 3), 'wait_inst_any_frac': round(mean(a['SQ_WAIT_INST_ANY']) / wave, 3),
       'wait_inst_lds_frac': round(mean(a['SQ_WAIT_INST_LDS']) / wave, 3),
       'active_inst_frac': round(mean(a['SQ_ACTIVE_INST_ANY']) / wave, 3), 'lds_active_cycles': mean(a['SQ_LDS_IDX_ACTIVE']),
       'lds_bank_conflict_frac': round(mean(a['SQ_LDS_BANK_CONFLICT']) / max(mean(a['SQ_LDS_IDX_ACTIVE']), 1), 3)}
json.dump(out, open('$R/gpurun_out/gcn2_pmc_$VAR.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
