#!/bin/bash
# Round-5 PMC evidence (GPU box): HBM traffic and matrix-pipe utilisation of the graph-conv kernels, matrix-pipe
# utilisation of the temporal-conv kernels, MFMA work issued per train step.  Separate --pmc passes per counter group.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/pmc_traffic.sh > gpurun_out/r5_pmc_traffic.log 2>&1; cp gpurun_out/gcn_pmc_traffic.json gpurun_out/r5_gcn3_pmc_traffic.json
bash tools/pmc_mfma.sh > gpurun_out/r5_pmc_mfma.log 2>&1; cp gpurun_out/mfma_util.json gpurun_out/r5_graphconv_mfma_util.json
WORKLOAD=tools/dev_tconv_time.py OUT=tconv_mfma_util.json bash tools/pmc_mfma.sh > gpurun_out/r5_pmc_tconv.log 2>&1; cp gpurun_out/tconv_mfma_util.json gpurun_out/r5_tconv_mfma_util.json
bash tools/pmc_step_mfma.sh > gpurun_out/r5_pmc_step.log 2>&1; cp gpurun_out/step_mfma.json gpurun_out/r5_step_mfma.json
tail -40 gpurun_out/r5_pmc_tconv.log
