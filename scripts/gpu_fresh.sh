#!/bin/bash
# the parity test in fresh processes, per build variant (uninitialised-memory / race hunting across boxes)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for so in global_racetrajectory_optimization_amd/csrc/variants/*.so; do
  name=$(basename $so .so | sed 's/^libmcq_//'); fails=0
  for i in $(seq 1 ${1:-8}); do
    MCQ_LIB=$R/$so python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "reference_tracks_match_golden" > /tmp/f_$i.log 2>&1 || { fails=$((fails+1)); grep -h "AssertionError: (" /tmp/f_$i.log | head -1; }
  done
  echo "$name: $fails failures of ${1:-8}"
done
