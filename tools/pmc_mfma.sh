#!/bin/bash
# MFMA-pipe utilisation of the graph-conv / temporal-conv kernels: SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES-free
# estimate = MFMA busy cycles / (1024 SIMDs x kernel duration x shader clock from GRBM_GUI_ACTIVE).
#   bash tools/pmc_mfma.sh -> gpurun_out/mfma_util.json
#   WORKLOAD=tools/dev_tconv_time.py OUT=tconv_mfma_util.json bash tools/pmc_mfma.sh   (temporal-conv kernels)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
WORKLOAD=${WORKLOAD:-tools/dev_gcn_time.py}
OUT=${OUT:-mfma_util.json}
mkdir -p $R/gpurun_out; rm -rf /tmp/pm1 /tmp/pm2
REPS=1 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d /tmp/pm1 -- python $R/$WORKLOAD > $R/gpurun_out/pm1.log 2>&1
REPS=1 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm2 -- python $R/$WORKLOAD > $R/gpurun_out/pm2.log 2>&1
python - <<PY
import csv, glob, json, collections
KEYS = ('gcn3h_fwd_kernel', 'gcn3h_dx_kernel', 'gcn3dwh_kernel', 'gcn3h_dcoef_kernel', 'tconvh_kernel<true, false>', 'tconvh_kernel<false, true>', 'tconvh_kernel<false, false>',
        'gcn3h_kernel', 'tconv_dw_f16_kernel', 'tconv_f16w_kernel', 'gcn3_kernel', 'gcn3_dcoef_kernel', 'gcn3_dw_kernel', 'gcn2_kernel', 'gcn_fused_kernel', 'gcn_dw_kernel', 'gcn_dcoef_kernel',
        'tconv3_kernel<true, false, 3,', 'tconv3_kernel<false, false, 3,', 'tconv3_kernel<true, false, 1,', 'tconv3_kernel<false, false, 1,',
        'tconv_dw_kernel<3', 'tconv_dw_kernel<1')
def load(d):
    cc = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    kt = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
    dur = {int(r['Dispatch_Id']): (int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(kt))}
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc)):
        d_ = int(r['Dispatch_Id'])
        if d_ not in dur: continue
        n = dur[d_][1]
        for key in KEYS:
            if key in n:
                acc[key][r['Counter_Name']].append(float(r['Counter_Value'])); acc[key]['_ns_' + r['Counter_Name']].append(dur[d_][0])
    return acc
a, b = load('/tmp/pm1'), load('/tmp/pm2')
out = {'source': 'tools/pmc_mfma.sh over $WORKLOAD (N=32, T=1024, V=53), MI355X; counters averaged per launch', 'kernels': {}}
for k in KEYS:
    if not a[k].get('SQ_WAVE_CYCLES') or not b[k].get('GRBM_GUI_ACTIVE'): continue
    mean = lambda v: sum(v) / len(v)
    ghz = mean(b[k]['GRBM_GUI_ACTIVE']) / 8.0 / mean(b[k]['_ns_GRBM_GUI_ACTIVE'])        # 8 XCDs report separately
    ns = mean(a[k]['_ns_SQ_VALU_MFMA_BUSY_CYCLES'])
    busy = mean(a[k]['SQ_VALU_MFMA_BUSY_CYCLES'])
    wave = mean(a[k]['SQ_WAVE_CYCLES'])
    out['kernels'][k] = {
        'launches': len(a[k]['SQ_VALU_MFMA_BUSY_CYCLES']), 'duration_us_under_pmc': round(ns / 1e3, 1), 'shader_clock_GHz': round(ghz, 3),
        'mfma_busy_cycles': busy, 'mfma_pipe_utilisation': round(busy / (1024 * ns * ghz), 3),
        'wave_cycles_quad': wave, 'wait_any_frac': round(mean(a[k]['SQ_WAIT_ANY']) / wave, 3),
        'wait_inst_any_frac': round(mean(a[k]['SQ_WAIT_INST_ANY']) / wave, 3), 'active_inst_frac': round(mean(a[k]['SQ_ACTIVE_INST_ANY']) / wave, 3),
        'lds_active_cycles': mean(a[k]['SQ_LDS_IDX_ACTIVE']), 'lds_bank_conflict_frac': round(mean(a[k]['SQ_LDS_BANK_CONFLICT']) / mean(a[k]['SQ_LDS_IDX_ACTIVE']), 3)}
json.dump(out, open('$R/gpurun_out/$OUT', 'w'), indent=1)
print(json.dumps(out['kernels'], indent=1))
PY
