#!/bin/bash
# Round 6: the elimination in isolation (scripts/kkt_check.hip) with the spike | y records not written / not read: the upper bound of the checkpoint idea
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
for k in kc kc_abl_noaystore kc_abl_noayload kc_abl_noayboth kc_abl_nostore; do
  for rep in 1 2; do
    echo "== $k fused=1 run $rep"; timeout 120 ./build/kc/$k 2000 6 2048 12 0 1 2>&1 | head -3
  done
done > gpurun_out/r06_ckpt_ablation.txt 2>&1
cat gpurun_out/r06_ckpt_ablation.txt
