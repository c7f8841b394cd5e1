#!/bin/bash
# Run on the GPU box (via gpurun) from the repo root:  scripts/profile_round.sh <tag> [bench args]
# Produces gpurun_out/<tag>_stats/ (kernel trace + stats) and gpurun_out/<tag>_pmc_<set>/ (one counter pass each: counters in their
# own runs, never combined with trace domains); summarise afterwards with scripts/pmc_summary.py into profiles/.
# Counter sets (gfx950: 8 SQ slots, 4 TCC slots per pass; MI355X_MICROARCH.md section "rocprofv3 PMC slots"):
#   wave    wave occupancy / stall split        mfma   matrix-core and VALU activity      lds    LDS traffic, VMEM instructions
#   fetch   FETCH_SIZE                          write  WRITE_SIZE + write request classes rdreq  read requests by size class
#   l2      L2 hit / miss, DRAM reads
# and the same TCC sets over scripts/build/calib_stream (known byte counts, 8-B and 16-B per lane) for the corrections.
set -u
TAG=${1:-r02}
shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
export TMPDIR=/tmp
cd /tmp
BENCH="python $R/bench.py --no-extras $*"
# what was profiled: SHA-256 of the engine sources on THIS box (bench.py quotes a counter summary only for the sources it runs on)
(cd $R && python -c "import bench; print(bench.source_sha())") > $R/gpurun_out/${TAG}_source_sha.txt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- $BENCH --steps 3 --warmup 1 > $R/gpurun_out/${TAG}_stats.log 2>&1
declare -A SETS
SETS[wave]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS"
SETS[mfma]="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_FMA_F64 SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"
SETS[lds]="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_FLAT"
SETS[fetch]="FETCH_SIZE"
SETS[write]="WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
SETS[rdreq]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
SETS[l2]="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUBBLE_sum"
for S in wave mfma lds fetch write rdreq l2; do
  timeout 300 rocprofv3 --pmc ${SETS[$S]} --output-format csv -d $R/gpurun_out/${TAG}_pmc_$S -- $BENCH --steps 1 --warmup 1 > $R/gpurun_out/${TAG}_pmc_$S.log 2>&1
  echo "pmc $S rc $?"
done
if [ -x $R/scripts/build/calib_stream ]; then
  for S in fetch write rdreq; do
    timeout 120 rocprofv3 --pmc ${SETS[$S]} --output-format csv -d $R/gpurun_out/${TAG}_calib_$S -- $R/scripts/build/calib_stream > $R/gpurun_out/${TAG}_calib_$S.log 2>&1
    echo "calib $S rc $?"
  done
fi
tail -n 1 $R/gpurun_out/${TAG}_stats.log | cut -c1-300
