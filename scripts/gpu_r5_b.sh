#!/bin/bash
# Round 5: the collective with two ranks on one GPU, the whole -m gpu suite, the bench line (host_to_host through two compute streams).
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r05c}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_comm.py -m gpu -q -x -s > gpurun_out/${T}_pytest_comm.log 2>&1
echo "comm pytest rc $?" >> gpurun_out/${T}_pytest_comm.log
tail -12 gpurun_out/${T}_pytest_comm.log
timeout 1500 python -m pytest tests -m gpu -q --durations=6 --deselect tests/test_gpu_comm.py > gpurun_out/${T}_pytest_gpu.log 2>&1
echo "pytest rc $?" >> gpurun_out/${T}_pytest_gpu.log
tail -12 gpurun_out/${T}_pytest_gpu.log
timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc $?"; python - <<PY
import json
d=json.load(open("gpurun_out/${T}_bench.json")); c=d["config"]
print("value %.0f ms/step %.3f kernel %s" % (d["value"], d["ms_per_step"], c.get("kernel_ms")))
print("host_to_host", c.get("host_to_host"))
print("roofline", d.get("roofline")); print("cpu", d.get("cpu_baseline")); print("iqp", c.get("iqp"))
PY
