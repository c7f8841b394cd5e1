#!/bin/bash
# GPU call A of round 3: same-box A/B of the kernel variants under build/variants, then the whole -m gpu suite on the tree's
# library (new goldens, full-size config 4, RCCL self-test, fp32 increment rows, pipelined host entry), then a full bench line.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
nproc > gpurun_out/r03a_nproc.txt
scripts/gpu_variants.sh r03a 2>&1 | tee gpurun_out/r03a_variants.txt
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r03a_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03a_pytest.log
tail -30 gpurun_out/r03a_pytest.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err
echo "bench rc $?"
cut -c1-1500 gpurun_out/r03a_bench.json
tail -5 gpurun_out/r03a_bench.err
