#!/bin/bash
# Builds scripts/kkt_check.hip (the saddle-point elimination in isolation, with its phase timers) into build/kc/:
#   kc        fp64 records            kc_f32   float records
#   kc_abl_*  ABLATION builds from a sed-modified TEMPORARY copy of the sources (results garbage, the phase times are what is looked at;
#             nothing of it lives in csrc/): nostore = the elimination's global stores removed
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
F="--offload-arch=gfx950 -O3 -std=c++17 -mllvm -enable-ipra=0 --gpu-max-threads-per-block=512 -DKKT_TIMERS=1"
mkdir -p $R/build/kc
$HIPCC $F -o $R/build/kc/kc $R/scripts/kkt_check.hip &
$HIPCC $F -DKC_F32=1 -o $R/build/kc/kc_f32 $R/scripts/kkt_check.hip &
T=$(mktemp -d)
mkdir -p $T/scripts $T/global_racetrajectory_optimization_amd/csrc $T/include
cp $R/global_racetrajectory_optimization_amd/csrc/*.h* $R/global_racetrajectory_optimization_amd/csrc/*.inc $T/global_racetrajectory_optimization_amd/csrc/
cp $R/include/mcq.h $T/include/; cp $R/scripts/kkt_check.hip $T/scripts/
sed -i 's|if (st_en\[r\]) o\[st_off\[r\]\] = (RT)nvv\[r\];|if (st_en[r] \&\& k < 0) o[st_off[r]] = (RT)nvv[r];|; s|if (cl < 5) o\[15 + cl\] = (RT)lov;|if (cl < 5 \&\& k < 0) o[15 + cl] = (RT)lov;|' $T/global_racetrajectory_optimization_amd/csrc/mcq_kkt.inc
grep -c "k < 0" $T/global_racetrajectory_optimization_amd/csrc/mcq_kkt.inc
$HIPCC $F -o $R/build/kc/kc_abl_nostore $T/scripts/kkt_check.hip &
wait
rm -rf $T
ls -la $R/build/kc
