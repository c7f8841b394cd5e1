#!/bin/bash
# Same-box comparison of the warm-started exchange's round cap (MCQ_WARM_ROUNDS; build/variants/libmcq_warm<cap>.so): per-pass kernel time,
# histogram of exchange rounds, fallbacks and the slowest problems of each IQP pass on the 1024 synthetic ovals (scripts/diag_iqp_rounds.py).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
for rep in 1; do
for so in build/variants/libmcq_warm*.so; do
  name=$(basename $so .so | sed 's/^libmcq_//')
  MCQ_LIB=$R/$so timeout 300 python scripts/diag_iqp_rounds.py > gpurun_out/r05_${name}_${rep}.jsonl 2> gpurun_out/r05_${name}_${rep}.err
  python - <<PY
import json
for l in open("gpurun_out/r05_${name}_${rep}.jsonl"):
    d = json.loads(l)
    print("${name} ${rep} pass %d: kernel %.2f ms, as_iters mean %.2f max %d, second attempts %d, per problem mean %.2f p90 %.2f max %.2f, hist %s" % (
        d["pass"], d["kernel_ms"].get("solve", 0.0) if isinstance(d["kernel_ms"], dict) else d["kernel_ms"], d["as_iters"][0], d["as_iters"][1], d["second_attempts"],
        d["ms_per_problem"]["kernel_mean"], d["ms_per_problem"]["kernel_p90"], d["ms_per_problem"]["kernel_max"], d["as_hist"]))
PY
done
done
