#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
scripts/gpu_variants.sh r04d "n2000_first_pass or reference_tracks_match_golden"
scripts/gpu_variants.sh r04e "n2000_first_pass"
