#!/bin/bash
# Runs on the GPU box: every build variant under build/variants/*.so (compiled in the build container by scripts/build_variants.sh
# with different -D switches; git-ignored, they travel with the snapshot) through a parity check and the bench line.   scripts/gpu_variants.sh <tag> [pytest -k expr]
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-var}
K=${2:-"reference_tracks_match_golden or n2000_first_pass or full_size_oval_properties"}
cd $R
mkdir -p gpurun_out
for so in build/variants/*.so; do
  name=$(basename $so .so | sed 's/^libmcq_//')
  export MCQ_LIB=$R/$so
  timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "$K" > gpurun_out/${T}_${name}_pytest.log 2>&1
  echo "$name pytest rc $? $(tail -1 gpurun_out/${T}_${name}_pytest.log)"
  timeout 300 python bench.py --no-extras --steps 5 --warmup 1 > gpurun_out/${T}_${name}.json 2> gpurun_out/${T}_${name}.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/${T}_${name}.json"))
    c=d["config"]
    print("${name}: %.0f solves/s, kernel solve %.2f ms, phases %s, iters ipm %.2f as %.2f 2nd %d failed %d, ticks %s" % (d["value"], c["kernel_ms"]["solve"], {k: round(v,3) for k,v in c["solver_phase_ms_per_problem"].items()}, c["mean_ipm_iters"], c["mean_as_iters"], c["second_attempts"], c["failed_problems"], [int(v) for v in c.get("ticks_mean", [])]))
except Exception as e:
    print("${name}: no result", e)
PY
done
unset MCQ_LIB
