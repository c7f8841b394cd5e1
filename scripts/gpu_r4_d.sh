#!/bin/bash
# GPU box: the write-pattern microbenchmark (two sizes) + same-box A/B of the build variants
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; mkdir -p gpurun_out
for kb in 736 2048; do timeout 120 build/tools/write_bw $kb 5 > gpurun_out/r04d_write_bw_$kb.jsonl 2>&1; cat gpurun_out/r04d_write_bw_$kb.jsonl; done
scripts/gpu_variants.sh r04d "n2000_first_pass"
