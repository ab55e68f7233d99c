#!/bin/bash
# Counters of the pointnet2 / nn_distance kernels at the P2RNet and the stress shapes (SURVEY 8d evidence):
# VALU activity, LDS bank conflicts, waits (one SQ pass + a clock pass), HBM bytes (FETCH_SIZE / WRITE_SIZE passes,
# FETCH_SIZE x2 as calibrated on gfx950: MI355X_MICROARCH.md / profiles/r2_gcn2_pmc_traffic.json).
#   bash tools/pmc_pointnet2.sh -> gpurun_out/r2_pointnet2_pmc.json  (copy to profiles/)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
for SH in p2r stress; do
  export SHAPE=$SH
  rm -rf /tmp/pn_sq_$SH /tmp/pn_clk_$SH /tmp/pn_f_$SH /tmp/pn_w_$SH
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/pn_sq_$SH -- python $R/tools/pn2_workload.py > $R/gpurun_out/pn_sq_$SH.log 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/pn_clk_$SH -- python $R/tools/pn2_workload.py > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pn_f_$SH -- python $R/tools/pn2_workload.py > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pn_w_$SH -- python $R/tools/pn2_workload.py > /dev/null 2>&1
done
python - <<PY
import csv, glob, json, collections
KEYS = ('fps', 'ball_query', 'group_points_kernel', 'group_grad', 'gather', 'three_nn', 'three_interpolate_kernel', 'three_interpolate_grad', 'interp_grad', 'nn_distance')
def load(d):
    cc = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    kt = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
    dur = {int(r['Dispatch_Id']): (int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(kt))}
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc)):
        d_ = int(r['Dispatch_Id'])
        if d_ not in dur: continue
        n = dur[d_][1]
        if 'at::native' in n or 'rocprim' in n: continue
        short = n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].split('<')[0].split('::')[-1].strip()
        acc[short][r['Counter_Name']].append(float(r['Counter_Value'])); acc[short]['_ns'].append(dur[d_][0])
    return acc
mean = lambda v: sum(v) / len(v)
out = {'source': 'tools/pmc_pointnet2.sh over tools/pn2_workload.py, MI355X; counters averaged per launch; FETCH_SIZE doubled (gfx950 calibration)',
       'peaks': {'fp32_valu_pairs_per_s': 19.7e12, 'hbm_bytes_per_s': 8.0e12}, 'shapes': {}}
for sh, desc in (('p2r', 'B=32 N=512 npoint=128 nsample=16 C=256'), ('stress', 'B=8 N=54272 npoint=2048 nsample=32 C=64 (build-defined, not on the model path)')):
    sq, clk, fe, wr = (load('/tmp/pn_%s_%s' % (t, sh)) for t in ('sq', 'clk', 'f', 'w'))
    ks = {}
    for k in sorted(sq):
        a = sq[k]
        if 'SQ_WAVE_CYCLES' not in a: continue
        ns = mean(a['_ns'])
        ghz = mean(clk[k]['GRBM_GUI_ACTIVE']) / 8.0 / mean(clk[k]['_ns']) if k in clk and 'GRBM_GUI_ACTIVE' in clk[k] else None
        rd = mean(fe[k]['FETCH_SIZE']) * 1024.0 * 2.0 if k in fe and 'FETCH_SIZE' in fe[k] else None
        wt = mean(wr[k]['WRITE_SIZE']) * 1024.0 if k in wr and 'WRITE_SIZE' in wr[k] else None
        wave = mean(a['SQ_WAVE_CYCLES'])
        e = {'launches': len(a['_ns']), 'duration_us': round(ns / 1e3, 2), 'shader_clock_GHz': round(ghz, 3) if ghz else None,
             'valu_busy_frac': round(mean(a['SQ_ACTIVE_INST_VALU']) * 4 / 1024.0 / (ns * ghz), 4) if ghz else None,
             'valu_insts': mean(a['SQ_INSTS_VALU']), 'wait_any_frac': round(mean(a['SQ_WAIT_ANY']) / wave, 3) if wave else None,
             'wait_inst_any_frac': round(mean(a['SQ_WAIT_INST_ANY']) / wave, 3) if wave else None,
             'lds_active_cycles': mean(a['SQ_LDS_IDX_ACTIVE']),
             'lds_bank_conflict_frac': round(mean(a['SQ_LDS_BANK_CONFLICT']) / max(mean(a['SQ_LDS_IDX_ACTIVE']), 1.0), 4),
             'hbm_read_bytes': rd, 'hbm_write_bytes': wt,
             'hbm_GBps': round(((rd or 0) + (wt or 0)) / ns, 1), 'hbm_frac_of_8TBps': round(((rd or 0) + (wt or 0)) / ns / 8000.0, 4)}
        ks[k] = e
    out['shapes'][sh] = {'shape': desc, 'kernels': ks}
json.dump(out, open('$R/gpurun_out/r2_pointnet2_pmc.json', 'w'), indent=1)
for sh in out['shapes']:
    for k, e in out['shapes'][sh]['kernels'].items():
        print(sh, k, e['duration_us'], 'us valu', e['valu_busy_frac'], 'lds_conf', e['lds_bank_conflict_frac'], 'hbm', e['hbm_GBps'], 'GB/s')
PY
