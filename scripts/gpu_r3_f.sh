#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 300 python scripts/diag_pipeline_trace.py 2>&1 | grep pipelined
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03f_bench.json 2> gpurun_out/r03f_bench.err
echo "bench rc $?"
python - <<PY
import json
d=json.load(open("gpurun_out/r03f_bench.json"))
print(d["value"], d["ms_per_step"], d["config"]["kernel_ms"])
print({k:v for k,v in d["host_to_host"].items() if k not in ("what",)})
print({k:v for k,v in d["iqp"].items() if k not in ("what",)})
PY
