#!/bin/bash
# Round 4: the elimination in isolation (phase timers; fp64 / float records / ablation without the elimination's stores), then the
# build variants under build/variants
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r04c}
cd $R; mkdir -p gpurun_out
for k in kc kc_f32; do
  for fused in 1 0; do
    timeout 120 ./build/kc/$k 2000 6 2048 12 0 $fused > gpurun_out/${T}_${k}_fused${fused}.txt 2>&1
    echo "$k fused=$fused: $(grep phases gpurun_out/${T}_${k}_fused${fused}.txt | head -1)"
  done
done
scripts/gpu_variants.sh ${T}v
