#!/bin/bash
# Round 6: HBM traffic of the solver kernel when EVERY problem takes the Goldfarb-Idnani path (scripts/bench_gi_mode.py: one default launch, then
# launches in MCQ_ALG_GI) -- read requests by size class and write requests, one counter pass each (counters in their own runs)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
mkdir -p $R/gpurun_out
timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $R/gpurun_out/r06_gi_pmc_rd -- python $R/scripts/bench_gi_mode.py > $R/gpurun_out/r06_gi_pmc_rd.log 2>&1
echo "rd rc $?"
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $R/gpurun_out/r06_gi_pmc_wr -- python $R/scripts/bench_gi_mode.py > $R/gpurun_out/r06_gi_pmc_wr.log 2>&1
echo "wr rc $?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_gi_stats -- python $R/scripts/bench_gi_mode.py > $R/gpurun_out/r06_gi_stats.log 2>&1
echo "stats rc $?"
ls $R/gpurun_out/r06_gi_pmc_rd/*/ | head
