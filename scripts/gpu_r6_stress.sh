#!/bin/bash
# Round 6: the two-path stress runs (3000 mixed problems; 600 rings of 1000 .. 3000 waypoints) after working sets beyond MCQ_KMAX curvature rows
# go straight to the Goldfarb-Idnani path (no overflow-slot exchange in between)
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r06}
cd $R; mkdir -p gpurun_out
timeout 600 python scripts/stress_two_paths.py 3000 11 > gpurun_out/${T}_stress_two_paths.json 2> gpurun_out/${T}_stress.err
echo "stress rc $?"; cut -c1-600 gpurun_out/${T}_stress_two_paths.json
timeout 900 python scripts/stress_two_paths.py 600 23 1000 3000 > gpurun_out/${T}_stress_large_rings.json 2> gpurun_out/${T}_stress_large.err
echo "large rings rc $?"; cut -c1-600 gpurun_out/${T}_stress_large_rings.json
