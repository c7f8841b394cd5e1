#!/bin/bash
# GPU call C of round 3: base vs final kernel on the same box, the whole -m gpu suite, the full bench line, the profile passes.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
scripts/gpu_variants.sh r03c2 2>&1 | tee gpurun_out/r03c2_variants.txt
timeout 1500 python -m pytest tests -m gpu -x -q --durations=10 > gpurun_out/r03c2_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03c2_pytest.log
tail -22 gpurun_out/r03c2_pytest.log
MCQ_POISON=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or n2000 or berlin or fp32" > gpurun_out/r03c2_pytest_poison.log 2>&1
echo "poison rc $?"; tail -3 gpurun_out/r03c2_pytest_poison.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r03c2_bench.json 2> gpurun_out/r03c2_bench.err
echo "bench rc $?"
cut -c1-400 gpurun_out/r03c2_bench.json
scripts/profile_round.sh r03 2>&1 | tail -12
