#!/bin/bash
# GPU box: the elimination in isolation (phase timers), then same-box A/B of the build variants, twice (run-to-run spread)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
for k in kc kc_f32; do
  for fused in 1 0; do
    timeout 120 ./build/kc/$k 2000 6 1024 12 0 $fused > gpurun_out/r04d_${k}_fused$fused.txt 2>&1
    echo "$k fused $fused: $(grep phases gpurun_out/r04d_${k}_fused$fused.txt | head -1)"
  done
done
scripts/gpu_variants.sh r04d "n2000_first_pass or reference_tracks_match_golden"
scripts/gpu_variants.sh r04e "n2000_first_pass"
