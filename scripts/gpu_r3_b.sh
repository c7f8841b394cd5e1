#!/bin/bash
# GPU call B of round 3: the factorisation in isolation with parts of a step removed (scripts/factor_bench.hip, build/fb/*), and the
# one GPU test that failed in call A.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
for f in build/fb/fb_*; do
  timeout 120 $f 1024 2000 12 1 2>&1 | sed "s/^/$(basename $f) fwd1 /"
done | tee gpurun_out/r03b_factor_bench.txt
for f in build/fb/fb_noexit build/fb/fb_now0 build/fb/fb_nodiag; do
  timeout 120 $f 1024 2000 12 0 2>&1 | sed "s/^/$(basename $f) fwd0 /"
done | tee -a gpurun_out/r03b_factor_bench.txt
timeout 120 build/fb/fb_noexit 256 2000 12 1 2>&1 | sed "s/^/noexit batch256 /" | tee -a gpurun_out/r03b_factor_bench.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config4_full or force_collective" > gpurun_out/r03b_pytest.log 2>&1
tail -5 gpurun_out/r03b_pytest.log
