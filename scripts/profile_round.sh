#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root:  scripts/profile_round.sh <tag>
# Produces gpurun_out/<tag>_{stats,fetch,write}/ ; summarise afterwards with scripts/pmc_summary.py into profiles/.
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${TAG}_stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_fetch -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${TAG}_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_write -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${TAG}_write.log 2>&1
tail -n 1 $R/gpurun_out/${TAG}_stats.log | cut -c1-400
