#!/bin/bash
# Round 5, first GPU call: the Goldfarb-Idnani tests, the whole -m gpu suite, and a same-box A/B of the bench line (round 4's kernels against this tree's).
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r05a}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_gi.py -m gpu -q -x -s --durations=5 > gpurun_out/${T}_pytest_gi.log 2>&1
echo "gi pytest rc $?" >> gpurun_out/${T}_pytest_gi.log
tail -25 gpurun_out/${T}_pytest_gi.log
timeout 1500 python -m pytest tests -m gpu -q --durations=8 --deselect tests/test_gpu_gi.py > gpurun_out/${T}_pytest_gpu.log 2>&1
echo "pytest rc $?" >> gpurun_out/${T}_pytest_gpu.log
tail -14 gpurun_out/${T}_pytest_gpu.log
for rep in 1 2; do
for so in build/variants/libmcq_00base.so build/variants/libmcq_new.so build/variants/libmcq_outline.so; do
  name=$(basename $so .so | sed 's/^libmcq_//')
  MCQ_LIB=$R/$so timeout 300 python bench.py --no-extras --steps 10 --warmup 2 > gpurun_out/${T}_${name}_${rep}.json 2> gpurun_out/${T}_${name}_${rep}.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/${T}_${name}_${rep}.json")); c=d["config"]
    print("${name} ${rep}: %.0f solves/s, ms/step %.3f, kernel_ms %s, failed %d" % (d["value"], d["ms_per_step"], c.get("kernel_ms"), c["failed_problems"]))
except Exception as e:
    print("${name}: no result", e)
PY
done
done
