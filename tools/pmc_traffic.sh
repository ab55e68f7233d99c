#!/bin/bash
# HBM bytes per launch of the graph-conv kernels from TCC counters (separate passes, as MI355X_MICROARCH.md
# prescribes), with the gfx950 FETCH_SIZE correction calibrated on bn_stats_kernel (reads 444.6 MB once).
#   bash tools/pmc_traffic.sh  ->  gpurun_out/gcn_pmc_traffic.json
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out; rm -rf /tmp/pmc_f /tmp/pmc_w
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -- python $R/tools/pmc_traffic.py > $R/gpurun_out/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -- python $R/tools/pmc_traffic.py > $R/gpurun_out/pmc_w.log 2>&1
python - <<PY
import csv, glob, json, collections
def per_kernel(d, counter):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter: continue
        n = r['Kernel_Name']
        for key in ('gcn3h_fwd_kernel', 'gcn3h_dx_kernel', 'gcn3dwh_kernel', 'gcn3h_dcoef_kernel', 'tconvh_kernelILb1ELb0', 'tconvh_kernelILb0ELb1', 'tconvh_kernel<true, false>', 'tconvh_kernel<false, true>', 'gcn3_kernel', 'gcn3_dcoef_kernel', 'gcn3_dw_kernel', 'gcn2_kernel', 'gcn_fused_kernel', 'gcn_dw_kernel', 'gcn_dcoef_kernel', 'bn_stats_kernel'):
            if key in n: acc[key].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}
fe, nf = per_kernel('/tmp/pmc_f', 'FETCH_SIZE')
wr, nw = per_kernel('/tmp/pmc_w', 'WRITE_SIZE')
tensor = 32 * 64 * 1024 * 53 * 4
cal = tensor / (fe['bn_stats_kernel'] * 1024.0)          # bytes actually read / bytes the raw counter (KiB) reports
out = {'source': 'tools/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over tools/pmc_traffic.py (N=32, T=1024, V=53), MI355X',
       'units': 'raw counters are KiB per launch, averaged over launches (gcn3 / gcn2: forward and data-gradient launches mixed 1:1)',
       'fetch_calibration': {'kernel': 'bn_stats_kernel', 'bytes_read': tensor, 'raw_KiB': fe['bn_stats_kernel'], 'factor': cal},
       'launches': nf, 'raw_KiB': {'FETCH_SIZE': fe, 'WRITE_SIZE': wr}, 'hbm_bytes_per_launch': {}}
for k in ('gcn3h_fwd_kernel', 'gcn3h_dx_kernel', 'gcn3dwh_kernel', 'gcn3h_dcoef_kernel', 'tconvh_kernelILb1ELb0', 'tconvh_kernelILb0ELb1', 'tconvh_kernel<true, false>', 'tconvh_kernel<false, true>', 'gcn3_kernel', 'gcn3_dcoef_kernel', 'gcn3_dw_kernel', 'gcn2_kernel', 'gcn_fused_kernel', 'gcn_dw_kernel', 'gcn_dcoef_kernel'):
    if k not in fe: continue
    rd, wt = fe[k] * 1024.0 * cal, wr[k] * 1024.0
    out['hbm_bytes_per_launch'][k] = {'read': int(rd), 'write': int(wt), 'total': int(rd + wt), 'vs_algorithmic': round((rd + wt) / (2.0 * tensor), 3)}
for dom in ('gcn3_kernel', 'gcn2_kernel'):
    if dom in out['hbm_bytes_per_launch']:
        out['bytes_per_launch'] = out['hbm_bytes_per_launch'][dom]['total']; out['dominant_kernel'] = dom
        break     # read by bench.py (roofline.traffic)
json.dump(out, open('$R/gpurun_out/${OUT:-gcn_pmc_traffic.json}', 'w'), indent=1)
print(json.dumps(out['fetch_calibration'])); print(json.dumps(out['hbm_bytes_per_launch'], indent=1))
PY
