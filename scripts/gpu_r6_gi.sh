#!/bin/bash
# Round 6: the Goldfarb-Idnani path after conditional re-orthogonalisation and slot growth in place: GI-mode bench, its tests, the two stress runs
# (default byte cap for full slots, and MCQ_GI_BYTES = 32 GB for the large rings)
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r06}
cd $R; mkdir -p gpurun_out
timeout 60 python scripts/bench_gi_mode.py > gpurun_out/${T}_gi_mode.json 2> gpurun_out/${T}_gi_mode.err
echo "gi mode rc $?"; cut -c1-500 gpurun_out/${T}_gi_mode.json
(timeout 600 python -m pytest tests/test_gpu_gi.py -m gpu -q -s 2>&1 | grep -E "GI mode|stadium|curvature-tight|the same through|rings above|passed|failed") > gpurun_out/${T}_gi_tests.txt
cat gpurun_out/${T}_gi_tests.txt
timeout 120 python scripts/stress_two_paths.py 3000 11 > gpurun_out/${T}_stress_two_paths.json 2> gpurun_out/${T}_stress.err
echo "stress rc $?"; cut -c1-600 gpurun_out/${T}_stress_two_paths.json
timeout 300 python scripts/stress_two_paths.py 600 23 1000 3000 > gpurun_out/${T}_stress_large_rings.json 2> gpurun_out/${T}_stress_large.err
echo "large rings rc $?"; cut -c1-600 gpurun_out/${T}_stress_large_rings.json
MCQ_GI_BYTES=34359738368 timeout 300 python scripts/stress_two_paths.py 600 23 1000 3000 > gpurun_out/${T}_stress_large_rings_32GB.json 2> gpurun_out/${T}_stress_large32.err
echo "large rings 32 GB rc $?"; cut -c1-600 gpurun_out/${T}_stress_large_rings_32GB.json
