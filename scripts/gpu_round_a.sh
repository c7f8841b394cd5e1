#!/bin/bash
# GPU call A of round 2: the whole -m gpu suite (new parity tests), a bench line, the list of PMC counters of this box, clocks.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
(rocm-smi --showclocks --showpower --showtemp --showperflevel 2>&1 | head -60) > gpurun_out/r02a_smi.txt
nproc > gpurun_out/r02a_nproc.txt; lscpu | head -20 >> gpurun_out/r02a_nproc.txt
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r02a_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r02a_pytest.log
tail -25 gpurun_out/r02a_pytest.log
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.err
cut -c1-600 gpurun_out/r02a_bench.json
export TMPDIR=/tmp; cd /tmp
timeout 120 rocprofv3 -L > $R/gpurun_out/r02a_counters.txt 2>&1
grep -c . $R/gpurun_out/r02a_counters.txt
