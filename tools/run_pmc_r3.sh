bash tools/pmc_traffic.sh > gpurun_out/r3_pmc_traffic.log 2>&1; cp gpurun_out/gcn_pmc_traffic.json gpurun_out/r3_gcn3_pmc_traffic.json
bash tools/pmc_mfma.sh > gpurun_out/r3_pmc_mfma.log 2>&1; cp gpurun_out/mfma_util.json gpurun_out/r3_graphconv_mfma_util.json
bash tools/pmc_step_mfma.sh > gpurun_out/r3_pmc_step.log 2>&1; cp gpurun_out/step_mfma.json gpurun_out/r3_step_mfma.json
tail -30 gpurun_out/r3_pmc_mfma.log
