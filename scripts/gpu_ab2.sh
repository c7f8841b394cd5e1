#!/bin/bash
# same-box A/B of every library under build/variants/: two rounds of the bench line each (scripts/build_variants.sh builds them)
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-ab}
K=${2:-"n2000_first_pass or reference_tracks_match_golden"}
cd $R; mkdir -p gpurun_out
for so in build/variants/*.so; do
  name=$(basename $so .so | sed 's/^libmcq_//')
  MCQ_LIB=$R/$so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "$K" > gpurun_out/${T}_${name}_pytest.log 2>&1
  echo "$name pytest rc $? $(tail -1 gpurun_out/${T}_${name}_pytest.log)"
done
for rep in 1 2; do
for so in build/variants/*.so; do
  name=$(basename $so .so | sed 's/^libmcq_//')
  MCQ_LIB=$R/$so timeout 300 python bench.py --no-extras --steps 10 --warmup 2 > gpurun_out/${T}_${name}_${rep}.json 2> gpurun_out/${T}_${name}_${rep}.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/${T}_${name}_${rep}.json")); c=d["config"]
    print("${name} ${rep}: %.0f solves/s, kernel %.3f ms, ipm %.2f (f32 %s) as %.2f, failed %d 2nd %d, phases %s" % (d["value"], c["kernel_ms"]["solve"], c["mean_ipm_iters"], d["roofline"]["model"][d["roofline"]["model"].find("float_record_factorisations"):][:36], c["mean_as_iters"], c["failed_problems"], c["second_attempts"], {k: round(v,3) for k,v in c["solver_phase_ms_per_problem"].items()}))
except Exception as e:
    print("${name}: no result", e)
PY
done
done
