#!/bin/bash
# Builds scripts/kkt_check.hip (the saddle-point elimination in isolation, with its phase timers) into build/kc/:
#   kc        fp64 records            kc_f32   float records
#   kc_abl_*  ABLATION builds from sed-modified TEMPORARY copies of the sources (results garbage, the phase times are what is looked at;
#             nothing of it lives in csrc/)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
F="--offload-arch=gfx950 -O2 -std=c++17 -mllvm -enable-ipra=0 -mllvm -amdgpu-schedule-metric-bias=0 --gpu-max-threads-per-block=512 -DKKT_TIMERS=1"
mkdir -p $R/build/kc
$HIPCC $F -o $R/build/kc/kc $R/scripts/kkt_check.hip &
$HIPCC $F -DKC_F32=1 -o $R/build/kc/kc_f32 $R/scripts/kkt_check.hip &
abl() {   # abl <name> <sed script applied to the temporary copy of mcq_kkt.inc>
  local T=$(mktemp -d)
  mkdir -p $T/scripts $T/global_racetrajectory_optimization_amd/csrc $T/include
  cp $R/global_racetrajectory_optimization_amd/csrc/*.h* $R/global_racetrajectory_optimization_amd/csrc/*.inc $T/global_racetrajectory_optimization_amd/csrc/
  cp $R/include/mcq.h $T/include/; cp $R/scripts/kkt_check.hip $T/scripts/
  sed -i "$2" $T/global_racetrajectory_optimization_amd/csrc/mcq_kkt.inc
  if cmp -s $T/global_racetrajectory_optimization_amd/csrc/mcq_kkt.inc $R/global_racetrajectory_optimization_amd/csrc/mcq_kkt.inc; then echo "ablation $1: pattern not found"; fi
  $HIPCC $F -o $R/build/kc/kc_abl_$1 $T/scripts/kkt_check.hip
  rm -rf $T
}
# nostore: the elimination's global stores removed;  noload: its coefficient loads of the chunks after the first removed (the first
# chunk's records reused);  noboth
abl nostore 's|if (st_en\[r\]) o\[st_off\[r\]\] = (RT)nvv\[r\];|if (st_en[r] \&\& k < 0) o[st_off[r]] = (RT)nvv[r];|; s|if (cl < 5) o\[15 + cl\] = (RT)lov;|if (cl < 5 \&\& k < 0) o[15 + cl] = (RT)lov;|' &
abl noload 's|if (k0 + KCKF < Lmax) kkt_point_load(c, sig, mk, wgt, fv, KKT_REC_POINT(k0 + KCKF), raw);|/* ablation: no further loads */|' &
abl noboth 's|if (st_en\[r\]) o\[st_off\[r\]\] = (RT)nvv\[r\];|if (st_en[r] \&\& k < 0) o[st_off[r]] = (RT)nvv[r];|; s|if (cl < 5) o\[15 + cl\] = (RT)lov;|if (cl < 5 \&\& k < 0) o[15 + cl] = (RT)lov;|; s|if (k0 + KCKF < Lmax) kkt_point_load(c, sig, mk, wgt, fv, KKT_REC_POINT(k0 + KCKF), raw);|/* ablation: no further loads */|' &
# round 6, the checkpoint idea (docs/NOTEBOOK.md R6.3) as an UPPER BOUND before building it: the forward-eliminated spike | y records (208 of the
# 368 B per waypoint the elimination writes and the spike pass reads back) not written / not read / neither -- what a spike pass that recomputes
# them from one record in six would save at most (its extra arithmetic not included)
abl noaystore 's|if (fl_on\[j\]) \*(gd2\*)(fl_dst\[j\] + (size_t)(k0 + kk) \* fl_rec\[j\]) = \*(const d2\*)fl_src\[j\];|if (fl_on[j] \&\& fl_rec[j] == KAD) *(gd2*)(fl_dst[j] + (size_t)(k0 + kk) * fl_rec[j]) = *(const d2*)fl_src[j];|' &
abl noayload 's|kkt_stage_issue<F32, KCK2 \* KAYR>(AY + (size_t)(lo_pt + (nq - 1) \* KCK2) \* KAYR, cl, ry);|/* ablation */|; s|kkt_stage_issue<F32, KCK2 \* KAYR>(AY + (size_t)(lo_pt + (q - 1) \* KCK2) \* KAYR, cl, ry);|/* ablation */|' &
abl noayboth 's|if (fl_on\[j\]) \*(gd2\*)(fl_dst\[j\] + (size_t)(k0 + kk) \* fl_rec\[j\]) = \*(const d2\*)fl_src\[j\];|if (fl_on[j] \&\& fl_rec[j] == KAD) *(gd2*)(fl_dst[j] + (size_t)(k0 + kk) * fl_rec[j]) = *(const d2*)fl_src[j];|; s|kkt_stage_issue<F32, KCK2 \* KAYR>(AY + (size_t)(lo_pt + (nq - 1) \* KCK2) \* KAYR, cl, ry);|/* ablation */|; s|kkt_stage_issue<F32, KCK2 \* KAYR>(AY + (size_t)(lo_pt + (q - 1) \* KCK2) \* KAYR, cl, ry);|/* ablation */|' &
wait
ls -la $R/build/kc
