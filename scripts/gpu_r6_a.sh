#!/bin/bash
# Round 6, first GPU call: the whole -m gpu suite on the round's first sources (new oracle, two slot pools, replay test) and the bench line with
# the latency_batch1 record.
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r06a}
mkdir -p $R/gpurun_out
cd $R
timeout 1500 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/${T}_pytest_gpu.log 2>&1
echo "pytest rc $?" >> gpurun_out/${T}_pytest_gpu.log
tail -14 gpurun_out/${T}_pytest_gpu.log
(timeout 300 python -m pytest tests/test_harness.py -m gpu -q -s 2>&1 | grep -E "replay|passed|failed|skipped") > gpurun_out/${T}_replay.txt
cat gpurun_out/${T}_replay.txt
timeout 300 python scripts/bench_gi_mode.py > gpurun_out/${T}_gi_mode.json 2> gpurun_out/${T}_gi_mode.err
echo "gi mode rc $?"; cut -c1-400 gpurun_out/${T}_gi_mode.json
timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc $?"; cut -c1-300 gpurun_out/${T}_bench.json
python - $T <<'PY'
import json
d=json.load(open("gpurun_out/%s_bench.json" % __import__("sys").argv[1]))
print(json.dumps(d.get("latency_batch1"), indent=1))
print("value", d["value"], "frac", d["roofline"]["frac"], "iqp", d.get("iqp",{}).get("value"), "h2h", d.get("host_to_host",{}).get("value"))
PY
