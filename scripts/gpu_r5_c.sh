#!/bin/bash
# Round 5: GI path inside the solver kernel -- GI + comm tests, the pipelined entry with one / two compute streams, bench A/B against round 4's kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r05d}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_gi.py tests/test_gpu_comm.py -m gpu -q -x -s > gpurun_out/${T}_pytest_gi.log 2>&1
echo "gi pytest rc $?" >> gpurun_out/${T}_pytest_gi.log
tail -14 gpurun_out/${T}_pytest_gi.log
python scripts/diag_pipe.py 2>&1 | tail -5; MCQ_PIPE_ONE_STREAM=1 python scripts/diag_pipe.py 2>&1 | tail -3
timeout 1500 python -m pytest tests -m gpu -q --durations=4 --deselect tests/test_gpu_gi.py --deselect tests/test_gpu_comm.py > gpurun_out/${T}_pytest_gpu.log 2>&1
echo "pytest rc $?" >> gpurun_out/${T}_pytest_gpu.log
tail -8 gpurun_out/${T}_pytest_gpu.log
