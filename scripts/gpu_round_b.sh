#!/bin/bash
# GPU call B of round 2: -m gpu suite, the full bench line (host_to_host, iqp, CPU-A / CPU-B), the 1-GPU collective self-test,
# config 4, and the profile passes (kernel stats + 7 counter sets + calibration).
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r02b}
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/${T}_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/${T}_pytest.log
tail -12 gpurun_out/${T}_pytest.log
if [ -d scratch_ft/reference ]; then   # BASELINE config 1 on the real library: the untouched main_globaltraj.py, stdout kept
  timeout 600 python -m pytest tests/test_harness.py -m gpu -s -q > gpurun_out/${T}_harness_berlin.log 2>&1; tail -2 gpurun_out/${T}_harness_berlin.log
fi
timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc $?"; cut -c1-300 gpurun_out/${T}_bench.json
timeout 300 python bench.py --steps 5 --warmup 1 --force-collective --no-extras > gpurun_out/${T}_bench_fc.json 2> gpurun_out/${T}_bench_fc.err
echo "fc rc $?"
timeout 300 python bench.py --config 4 --steps 2 --warmup 1 --force-collective > gpurun_out/${T}_bench_c4.json 2> gpurun_out/${T}_bench_c4.err
echo "c4 rc $?"; cut -c1-300 gpurun_out/${T}_bench_c4.json
if [ "${2:-prof}" = "prof" ]; then scripts/profile_round.sh ${T}; fi
python __graft_entry__.py smoke > gpurun_out/${T}_smoke.log 2>&1; tail -2 gpurun_out/${T}_smoke.log
