#!/bin/bash
# GPU call E: gram kernel A/B (MFMA vs register-column form) and copy-engine experiments for the pipelined host entry.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
scripts/gpu_variants.sh r03e 2>&1 | tee gpurun_out/r03e_variants.txt
for envs in "X=1" "HSA_ENABLE_SDMA=1" "HSA_ENABLE_SDMA=0" "HSA_ENABLE_SDMA=1 ROC_USE_SDMA=1" "GPU_MAX_HW_QUEUES=8"; do
  echo "== $envs"
  env $envs timeout 300 python scripts/diag_pipeline_trace.py 2>&1 | grep pipelined
done | tee gpurun_out/r03e_pipeline_env.txt
