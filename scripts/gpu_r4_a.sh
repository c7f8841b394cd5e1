#!/bin/bash
# Round 4, first GPU call: the whole -m gpu suite on the single-translation-unit library (harness tests included when the reference tree
# was shipped: scripts/gpu_with_reference.sh), the bench line, the shortest-path bench, build variants under build/variants.
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=${1:-r04a}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/${T}_pytest_gpu.log 2>&1
echo "pytest rc $?" >> gpurun_out/${T}_pytest_gpu.log
tail -14 gpurun_out/${T}_pytest_gpu.log
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc $?"; cut -c1-400 gpurun_out/${T}_bench.json
timeout 300 python scripts/bench_shortest_path.py > gpurun_out/${T}_shortest_path.json 2> gpurun_out/${T}_shortest_path.err
echo "shortest path rc $?"; cut -c1-400 gpurun_out/${T}_shortest_path.json
if ls build/variants/*.so > /dev/null 2>&1; then scripts/gpu_variants.sh ${T}v; fi
timeout 300 python bench.py --force-collective --steps 5 --warmup 2 --no-extras > gpurun_out/${T}_bench_force_collective_1gpu.json 2> gpurun_out/${T}_bench_fc.err
echo "force-collective rc $?"; cut -c1-300 gpurun_out/${T}_bench_force_collective_1gpu.json; tail -3 gpurun_out/${T}_bench_fc.err
