#!/bin/bash
# GPU call D of round 3: the -m gpu suite on the tree's library (curvature-row overflow path), a copy / kernel trace of the pipelined
# host entry, one rank's shard of BASELINE config 5 (8192 per-track centrelines, float increment rows).
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r03d_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03d_pytest.log
tail -16 gpurun_out/r03d_pytest.log
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/r03d_trace -- python $R/scripts/diag_pipeline_trace.py > $R/gpurun_out/r03d_trace.log 2>&1)
grep pipelined gpurun_out/r03d_trace.log
python scripts/diag_pipeline_trace.py --summarise gpurun_out/r03d_trace 2>&1 | tee gpurun_out/r03d_trace_summary.txt
timeout 900 python bench.py --config 5 --steps 2 --warmup 1 --no-extras > gpurun_out/r03d_config5_shard.json 2> gpurun_out/r03d_config5.err
echo "config5 rc $?"; cut -c1-700 gpurun_out/r03d_config5_shard.json
rm -rf gpurun_out/r03d_trace
